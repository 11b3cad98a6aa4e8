"""BASELINE.json configs[3] (32768 x 32768) and configs[4] (512^3) at FULL size on the device, through the C-ABI.

The checker cannot sweep a billion cells in test time, so full-size parity rests on a size-independent property of the
iteration: after K iterations only cells within K (Manhattan) of a goal can have moved, so the full-size field restricted
to a window around the goal must equal -- bit for bit, precise math -- the checker run on that window alone (whose own
border is further than K from the goal), and every cell outside the window must still hold its seed.  Same form as
tests/test_gpu_parity.py::test_full_size_8192_window_property (config 3's grid).  What only full size exercises: the
32-bit byte offsets of the buffer addressing at 2 strips x 16 chunks x 512 planes (3-D) and 128 strips x 2048 chunks
(2-D), the plane bases, the XCD block remap on grids of 10^5..10^6 tasks, the mask packing at that size.

The tall 3-D shapes are the 3-D twins of test_extreme_aspect_ratios: whole fields against the checker.

Reference behaviour matched: harmonic_cpu.cpp:81-133 (3-D update, colour rule (x0+x1+x2+iteration) even) and :38-79 (2-D).
"""
import ctypes as ct
import os

import numpy as np
import pytest

import _oracle as O
from epic_amd import epic_harmonic as eh
from epic_amd.harmonic import Harmonic
from epic_amd.synthetic import synthetic_grid

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(900)]

E = eh._epic
NT = 1024
SEED = np.float32(-1e6)


def make(m, u, locked):
    h = Harmonic()
    h.set_grid(m, u, locked)
    h.epsilon = 1e-6
    h.numIterationsToStaggerCheck = 100
    return h


def gpu_init(h):
    for fn in (E.harmonic_initialize_dimension_size_gpu, E.harmonic_initialize_potential_values_gpu,
               E.harmonic_initialize_locked_gpu):
        assert fn(h) == 0, fn.__name__
    assert E.harmonic_initialize_gpu(h, NT) == 0


def gpu_fini(h):
    for fn in (E.harmonic_uninitialize_gpu, E.harmonic_uninitialize_dimension_size_gpu,
               E.harmonic_uninitialize_potential_values_gpu, E.harmonic_uninitialize_locked_gpu):
        assert fn(h) == 0, fn.__name__


def run_gpu(m, u0, locked, k, scheme, track):
    """k iterations (first one a check sweep, so delta is iteration 0's) through the C-ABI; returns field, delta."""
    h = make(m, u0, locked)
    gpu_init(h)
    assert E.epic_hip_set_scheme(h, scheme) == 0
    assert E.epic_hip_set_activity_tracking(h, track) == 0
    assert E.harmonic_update_and_check_gpu(h, NT) in (0, 1)
    assert E.epic_hip_update_n_gpu(h, k - 1, 0) == 0
    assert h.currentIteration == k
    assert E.harmonic_get_potential_values_gpu(h) == 0
    gpu_fini(h)
    return h.u_array(), float(h.delta)


def run_oracle(m, u0, locked, k, scheme):
    p = O.Problem(m, u0, locked)
    lib = O.oracle()
    if scheme == eh.SCHEME_JACOBI:
        # delta of the FIRST sweep, as on the device side above
        assert lib.oracle_jacobi_run(ct.byref(p.h), 1) == 0
        first = float(p.h.delta)
        assert lib.oracle_jacobi_run(ct.byref(p.h), k - 1) == 0
        return p.u, first
    for i in range(k):
        (lib.oracle_update_and_check if i == 0 else lib.oracle_update)(ct.byref(p.h))
    return p.u, float(p.h.delta)


@pytest.mark.parametrize("scheme", [eh.SCHEME_JACOBI, eh.SCHEME_REDBLACK])
def test_config5_512_cubed_window_property(scheme):
    """BASELINE configs[4]: 512^3, 5 % obstacles, centre goal (256, 256, 256 -- on the strip seam 256 of x2 and on the
    chunk seam of x1).  K = 12 iterations; the window is the cube of half-width 32 around the goal."""
    n, K, W = 512, 12, 32
    m = [n, n, n]
    u0, locked = synthetic_grid(m)
    c = n // 2
    win = (slice(c - W, c + W),) * 3
    uw = u0.reshape(m)[win].copy()
    lw = locked.reshape(m)[win].copy()
    want, wdelta = run_oracle([2 * W] * 3, uw, lw, K, scheme)
    want = want.reshape([2 * W] * 3)
    fields = []
    for track in (0, 1):
        got, gdelta = run_gpu(m, u0, locked, K, scheme, track)
        got = got.reshape(m)
        assert np.array_equal(got[win], want), "window around the goal differs from the checker's run on the window"
        assert gdelta == wdelta
        outside = np.ones(m, dtype=bool)
        outside[win] = False
        lk = locked.reshape(m) != 0
        assert np.all(got[outside & ~lk] == SEED), "a cell the front cannot have reached has moved"
        assert np.array_equal(got[lk], u0.reshape(m)[lk]), "locked cells must be untouched"
        moved = int((got != u0.reshape(m)).sum())
        assert 0 < moved <= (2 * K + 1) ** 3
        fields.append(got.copy())
    assert np.array_equal(fields[0], fields[1]), "activity tracking changed the result"


@pytest.mark.parametrize("m", [[3, 70000, 300], [600, 3, 2100], [70000, 3, 260], [5, 300, 20000]])
def test_tall_3d_shapes_vs_checker(m):
    """Whole-field parity on 3-D grids that stretch one axis at a time: 70 000 rows in a plane (x1) with 2 strips,
    600 planes of 9 strips, 70 000 planes, 79 strips per row.  Goals at both ends and in the middle so every part moves."""
    u0, locked = synthetic_grid(m, 29, 0.05)
    free = np.flatnonzero(locked == 0)
    for idx in (free[0], free[free.size // 2], free[-1]):
        u0[idx] = 0.0
        locked[idx] = 1
    for scheme in (eh.SCHEME_JACOBI, eh.SCHEME_REDBLACK):
        got, gdelta = run_gpu(m, u0, locked, 6, scheme, 0)
        want, wdelta = run_oracle(m, u0, locked, 6, scheme)
        assert np.array_equal(got.ravel(), want), f"{m} scheme {scheme}"
        assert gdelta == wdelta


def _synthetic_big(m):
    """synthetic_grid through the checker's C generator (OpenMP): the numpy one needs minutes for 2^30 cells."""
    return O.oracle_synthetic(m)


def test_config4_32768_squared_window_property_one_gpu():
    """BASELINE configs[3]'s grid on ONE GPU (2 x 4.3 GB of u + 0.13 GB of lane masks out of 288 GB).  K = 16 Jacobi
    sweeps with activity tracking on (the default above 4 Mcell: the work lists over 2^18 tiles) -- window of half-width
    64 around the goal (16384, 16384), a strip seam and a chunk seam."""
    n, K, W = 32768, 16, 64
    m = [n, n]
    u0, locked = _synthetic_big(m)
    c = n // 2
    win = (slice(c - W, c + W), slice(c - W, c + W))
    uw = u0.reshape(m)[win].copy()
    lw = locked.reshape(m)[win].copy()
    want, wdelta = run_oracle([2 * W, 2 * W], uw, lw, K, eh.SCHEME_JACOBI)
    h = make(m, u0, locked)
    gpu_init(h)
    assert E.harmonic_update_and_check_gpu(h, NT) in (0, 1)
    assert E.epic_hip_update_n_gpu(h, K - 1, 0) == 0
    assert E.harmonic_get_potential_values_gpu(h) == 0
    gpu_fini(h)
    got = h.u_array().reshape(m)
    assert np.array_equal(got[win], want.reshape(2 * W, 2 * W))
    assert float(h.delta) == wdelta
    # outside the window nothing may have moved: compare with the seed / the locked values band by band (no 4 GB temporaries)
    u0 = u0.reshape(m)
    for r0 in range(0, n, 2048):
        band = got[r0:r0 + 2048]
        same = band == u0[r0:r0 + 2048]
        if r0 <= c < r0 + 2048 or r0 <= c - W < r0 + 2048 or r0 <= c + W - 1 < r0 + 2048:
            lo, hi = max(c - W, r0) - r0, min(c + W, r0 + 2048) - r0
            same[lo:hi, c - W:c + W] = True
        assert same.all(), f"rows {r0}..{r0 + 2047}: a cell outside the goal's reach has moved"


def test_config4_32768_squared_fused_tol_pairs_window_property():
    """The same grid with the benchmarked arithmetic and kernel: tol math, activity tracking off, so that the 15 plain
    iterations behind the check run as seven fused double sweeps (jacobi_fused2d_kernel: 133 strips of 248 columns, byte
    offsets up to 4.3 GB) and one single sweep.  Window around the goal against the tol checker on the window alone,
    everything else still at its seed."""
    n, K, W = 32768, 16, 64
    m = [n, n]
    u0, locked = _synthetic_big(m)
    c = n // 2
    win = (slice(c - W, c + W), slice(c - W, c + W))
    p = O.Problem([2 * W, 2 * W], u0.reshape(m)[win].copy(), locked.reshape(m)[win].copy())
    assert O.oracle().oracle_tol_run(ct.byref(p.h), 1, 0) == 0       # the check is the first iteration
    wdelta = float(p.h.delta)
    assert O.oracle().oracle_tol_run(ct.byref(p.h), K - 1, 0) == 0
    h = make(m, u0, locked)
    gpu_init(h)
    assert E.epic_hip_set_math_mode(h, eh.MATH_TOL) == 0 and E.epic_hip_set_activity_tracking(h, 0) == 0
    assert E.epic_hip_iterations_per_pass(h) == 2
    assert E.harmonic_update_and_check_gpu(h, NT) in (0, 1)
    assert float(h.delta) == wdelta
    assert E.epic_hip_update_n_gpu(h, K - 1, 0) == 0
    assert E.harmonic_get_potential_values_gpu(h) == 0
    gpu_fini(h)
    got = h.u_array().reshape(m)
    assert np.array_equal(got[win].ravel(), p.u)
    u0 = u0.reshape(m)
    for r0 in range(0, n, 2048):
        same = got[r0:r0 + 2048] == u0[r0:r0 + 2048]
        if r0 <= c < r0 + 2048 or r0 <= c - W < r0 + 2048 or r0 <= c + W - 1 < r0 + 2048:
            lo, hi = max(c - W, r0) - r0, min(c + W, r0 + 2048) - r0
            same[lo:hi, c - W:c + W] = True
        assert same.all(), f"rows {r0}..{r0 + 2047}: a cell outside the goal's reach has moved"
