"""CPU side of the `tol` math mode: the checker's statement of it (oracle/tol_checker.c) against the reference-generated
goldens, and the split e^u = q 2^n on its own.  No GPU.

What is claimed for the arithmetic (the device kernels are held to it bit for bit in tests/test_gpu_tol.py):
  * it is NOT the reference's arithmetic; converged at eps = 1e-6 it agrees with the reference's converged fields within
    1e-5 max(1, |u|) on the seeded grids and on basic.png (maze / umass need minutes on a CPU: GPU suite) -- and within 1e-6
    when the loop finishes with the reference's own iteration from delta < 10 eps on (the default: oracle_tol_complete);
  * Jacobi stops by the reference's own test (max |du| < eps), after about as many iterations as the reference's red-black;
  * the split is unbiased to a few 1e-3 ulp and within one ulp.
"""
import ctypes as ct
import os

import numpy as np
import pytest

import _oracle as O

SMALL_2D = ["g2d_16", "g2d_32", "g2d_64", "g2d_23x37", "g2d_5x7", "g2d_3x3", "g2d_8x300", "g2d_70x66_dense"]
SMALL_3D = ["g3d_8", "g3d_16", "g3d_7x9x11", "g3d_20x12x34"]


def rel_err(got, want, locked):
    free = (np.ravel(locked) == 0) & (np.ravel(want) > -9e5)
    e = np.abs(np.ravel(got).astype(np.float64) - np.ravel(want)) / np.maximum(1.0, np.abs(np.ravel(want)))
    return float(e[free].max()) if free.any() else 0.0


@pytest.fixture
def finish_rule():
    """Switches the checker's finishing rule (oracle_tol_complete: the reference's own iteration from the first check with
    delta < 10 eps) and puts it back on."""
    lib = O.oracle()
    lib.oracle_tol_set_finish.argtypes = (ct.c_int,)
    lib.oracle_tol_set_finish.restype = None
    yield lib.oracle_tol_set_finish
    lib.oracle_tol_set_finish(1)


@pytest.mark.parametrize("finish", [1, 0])
@pytest.mark.parametrize("scheme", [0, 1])
@pytest.mark.parametrize("name", SMALL_2D + SMALL_3D)
def test_tol_converges_on_the_seeded_grids_within_the_bar(goldens, name, scheme, finish, finish_rule):
    """finish = 1: the library's default for its "until converged" loops -- the tol iteration hands over to the reference's
    own iteration at the first check with delta < 10 eps; the converged field is then within 1e-6 (measured: <= 2.7e-7) of
    the reference's.  finish = 0: the tol iteration to the end (within the 1e-5 bar on these grids)."""
    finish_rule(finish)
    g, info = goldens["small"], goldens["manifest"]["small"][name]
    p = O.Problem(g[name + "/m"], g[name + "/u0"], g[name + "/locked"], info["epsilon"], info["stagger"])
    assert O.oracle().oracle_tol_complete(ct.byref(p.h), scheme) == 0, "did not stop by max |du| < eps"
    assert p.h.delta < info["epsilon"]
    assert p.h.currentIteration >= max(int(x) for x in g[name + "/m"])
    want = g[name + "/converged"]
    assert rel_err(p.u, want, p.locked) <= (1e-6 if finish else 1e-5)
    unreached = np.ravel(want) <= -9e5
    assert np.array_equal(p.u[unreached], np.ravel(want)[unreached]), "cells the front never reaches stay at the seed, exactly"
    # about as many iterations as the reference's own run (stagger-quantised; the finishing phase needs a check of its own)
    assert abs(int(p.h.currentIteration) - info["iterations"]) <= max((6 if finish else 2) * info["stagger"], 0.05 * info["iterations"])


@pytest.mark.timeout(600)
def test_tol_basic_map_jacobi_stops_and_agrees(goldens):
    """The reference's 256 x 256 map (tests/maps/basic.png), ill-conditioned enough to matter: |u| up to 33."""
    m, u0, locked = O.load_png_reference_rule(os.path.join(O.ROOT, "tests", "golden", "maps", "basic.png"))
    p = O.Problem(m, u0, locked, 1e-6, 100)
    assert O.oracle().oracle_tol_complete(ct.byref(p.h), 0) == 0
    run = goldens["manifest"]["maps"]["basic"]["runs"]["1e-06"]
    assert p.h.delta < 1e-6 and abs(int(p.h.currentIteration) - run["iterations"]) <= 0.02 * run["iterations"]
    # with the finishing rule (the default): 2.3e-7 (19 601 tol + 4 301 reference iterations against the reference's 23 801)
    assert rel_err(p.u, goldens["maps"]["basic/converged_1e-06"], p.locked) <= 1e-6


def test_split_is_unbiased_and_within_one_ulp():
    """e^u = q 2^n against exp() in double: over the values a field takes (|u| up to six hundred -- exp() in double underflows beyond --, on the f32 grid) the
    mean signed error stays within 1e-2 ulp in every band and the largest error below one ulp.  (The relaxation amplifies a
    systematic error of the update by the square of the domain's radius; a bias of 0.1 ulp -- the hardware's v_exp_f32 --
    moved the reference's 256 x 256 map by 2e-5.)"""
    rng = np.random.default_rng(7)
    lib = O.oracle()
    for lo, hi in ((-0.05, -1e-4), (-1.0, -0.05), (-10.0, -1.0), (-60.0, -10.0), (-600.0, -60.0)):
        u = rng.uniform(lo, hi, 400000).astype(np.float32)
        q = np.empty_like(u)
        e = np.empty(u.size, dtype=np.int32)
        lib.oracle_tol_split(u.ctypes.data_as(ct.POINTER(ct.c_float)), u.size, q.ctypes.data_as(ct.POINTER(ct.c_float)),
                             e.ctypes.data_as(ct.POINTER(ct.c_int)))
        assert np.all((q >= 0.70) & (q <= 1.42))
        ref = np.exp(u.astype(np.float64))
        got = np.ldexp(q.astype(np.float64), e)
        ulp = np.ldexp(1.0, e - 23) * np.where(q >= 1.0, 1.0, 0.5)
        err = (got - ref) / ulp
        assert abs(err.mean()) < 1.2e-2, (lo, hi, float(err.mean()))
        assert np.abs(err).max() < 1.0, (lo, hi, float(np.abs(err).max()))
    # the seed of obstacles / unreached cells: n stays inside the 22 bits the magic-number trick has
    u = np.array([-1e6], dtype=np.float32)
    q = np.empty_like(u)
    e = np.empty(1, dtype=np.int32)
    lib.oracle_tol_split(u.ctypes.data_as(ct.POINTER(ct.c_float)), 1, q.ctypes.data_as(ct.POINTER(ct.c_float)),
                         e.ctypes.data_as(ct.POINTER(ct.c_int)))
    assert e[0] == -1442695 and 0.96 < q[0] < 0.98


def test_unreached_and_enclosed_cells_keep_the_seed_exactly():
    """A free cell whose neighbours are all at the seed must come out at the seed again (the reference's sequence does:
    -1e6 + ln 4 - ln 4 rounds back), and a front cell takes its value from its one reached neighbour."""
    m = [7, 9]
    u0 = np.full(m, -1e6, dtype=np.float32)
    locked = np.zeros(m, dtype=np.uint32)
    locked[0, :] = locked[-1, :] = locked[:, 0] = locked[:, -1] = 1
    u0[3, 2] = 0.0
    locked[3, 2] = 1
    locked[2:5, 5] = locked[2, 6] = locked[4, 6] = 1   # (3, 6) is enclosed: its neighbours are obstacles and the border
    p = O.Problem(m, u0, locked)
    ref = O.Problem(m, u0, locked)
    lib = O.oracle()
    for k in range(1, 6):
        assert lib.oracle_tol_run(ct.byref(p.h), 1, 0) == 0 and lib.oracle_jacobi_run(ct.byref(ref.h), 1) == 0
        f, r = p.field(), ref.field()
        assert f[3, 6] == np.float32(-1e6) and r[3, 6] == np.float32(-1e6)
        assert np.array_equal(f <= -9e5, r <= -9e5), "the front moves one cell per sweep in both"
        reached = r > -9e5
        assert np.abs(f[reached].astype(np.float64) - r[reached]).max() <= 2e-6
