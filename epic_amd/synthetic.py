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
