"""Seeded synthetic occupancy grids for BASELINE configs 3-5: iid obstacles from a counter-based
hash, obstacle border, one goal at the centre.  The same hash is implemented in C in the checker
(oracle/harmonic_oracle.c: oracle_synthetic); tests/test_synthetic.py holds the two bit-identical."""
import numpy as np

_GOLD = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
DEFAULT_SEED = 20240601


def _mix64(z):
    z = (z ^ (z >> np.uint64(30))) * _M1
    z = (z ^ (z >> np.uint64(27))) * _M2
    return z ^ (z >> np.uint64(31))


def synthetic_grid(m, seed=DEFAULT_SEED, density=0.05, chunk=1 << 24):
    """Returns (u float32, locked uint32), both flat with prod(m) cells (row-major, last dimension contiguous)."""
    m = [int(x) for x in m]
    cells = int(np.prod(np.asarray(m, dtype=np.int64)))
    thresh = np.uint64(int(density * 9007199254740992.0))
    locked = np.empty(cells, dtype=np.uint32)
    with np.errstate(over="ignore"):
        for lo in range(0, cells, chunk):
            hi = min(cells, lo + chunk)
            idx = np.arange(lo, hi, dtype=np.uint64)
            h = _mix64(np.uint64(seed) ^ (idx * _GOLD))
            obstacle = (h >> np.uint64(11)) < thresh
            border = np.zeros(hi - lo, dtype=bool)
            rem = idx.copy()
            for dim in reversed(m):
                x = rem % np.uint64(dim)
                rem //= np.uint64(dim)
                border |= (x == 0) | (x == np.uint64(dim - 1))
            locked[lo:hi] = obstacle | border
    u = np.full(cells, -1e6, dtype=np.float32)
    goal = 0
    for dim in m:
        goal = goal * dim + dim // 2
    u[goal] = 0.0
    locked[goal] = 1
    return u, locked


RAMP_RATE = 0.3   # per cell of Manhattan distance: about what a converged field on 5 % random obstacles falls by (512^2: -137 over ~500 cells)


def ramp_rows(m, row_lo, row_hi, u, locked, rate=RAMP_RATE):
    """A developed-LIKE start for timing legs that cannot afford tens of thousands of untimed iterations (32768^2, 512^3 on slabs):
    every unlocked cell of rows [row_lo, row_hi) of the first axis gets u = -rate x its Manhattan distance to the goal (the centre cell), so
    that every cell takes the general arithmetic path from the first iteration (on the all -1e6 start the same kernels run ~15 % faster).
    u / locked: the flat arrays of those rows, changed in place.  Not a reference workload: a leg that uses it says so."""
    m = [int(x) for x in m]
    inner = int(np.prod(m[1:]))
    goal = [d // 2 for d in m]
    dist_inner = np.zeros(m[1:], dtype=np.float32)
    for ax, d in enumerate(m[1:]):
        shape = [1] * (len(m) - 1)
        shape[ax] = d
        dist_inner = dist_inner + np.abs(np.arange(d, dtype=np.float32) - goal[ax + 1]).reshape(shape)
    dist_inner = dist_inner.ravel()
    for r in range(row_lo, row_hi):
        lo = (r - row_lo) * inner
        free = locked[lo:lo + inner] == 0
        u[lo:lo + inner][free] = (-rate * (dist_inner + abs(r - goal[0])))[free]
    return u
