"""The checker itself: oracle/harmonic_oracle.c against (a) the compiled reference when it is available and
(b) the committed golden vectors that the reference produced (tests/golden/generate_goldens.py)."""
import ctypes as ct

import numpy as np
import pytest

import _oracle as O

SMALL = ["g2d_16", "g2d_32", "g2d_64", "g2d_23x37", "g2d_5x7", "g2d_3x3", "g2d_8x300", "g2d_70x66_dense",
         "g3d_8", "g3d_16", "g3d_7x9x11", "g3d_20x12x34"]


def _steps(lib_update, lib_check, prob, k):
    for i in range(k):
        (lib_check if i == k - 1 else lib_update)(ct.byref(prob.h))


@pytest.mark.parametrize("name", SMALL)
def test_oracle_matches_golden_half_sweeps(goldens, name):
    g = goldens["small"]
    m, u0, locked = g[name + "/m"], g[name + "/u0"], g[name + "/locked"]
    lib = O.oracle()
    for k in (1, 2, 3, 10):
        p = O.Problem(m, u0, locked, 1e-6, 100)
        _steps(lib.oracle_update, lib.oracle_update_and_check, p, k)
        assert np.array_equal(p.u, g[f"{name}/rb{k}"]), f"{name}: field after {k} half-sweeps"
        assert np.float32(p.h.delta) == g[f"{name}/rb{k}_delta"]


@pytest.mark.parametrize("name", SMALL)
def test_oracle_matches_golden_converged(goldens, name):
    g, info = goldens["small"], goldens["manifest"]["small"][name]
    p = O.Problem(g[name + "/m"], g[name + "/u0"], g[name + "/locked"], info["epsilon"], info["stagger"])
    assert O.oracle().oracle_complete(ct.byref(p.h)) == 0
    assert p.h.currentIteration == info["iterations"]
    assert float(p.h.delta) == info["delta"]
    assert np.array_equal(p.u, g[name + "/converged"])


def test_oracle_matches_golden_map_basic(goldens):
    """A reference PNG map end to end (256x256, ~1 s): loader rule, iteration count, final delta, full field."""
    info = goldens["manifest"]["maps"]["basic"]
    m, u0, locked = O.load_png_reference_rule(O.ROOT + "/tests/golden/maps/basic.png")
    assert m == info["m"]
    for eps in ("0.001", "1e-06"):
        p = O.Problem(m, u0, locked, float(eps), 100)
        assert O.oracle().oracle_complete(ct.byref(p.h)) == 0
        run = info["runs"][eps]
        assert p.h.currentIteration == run["iterations"] and float(p.h.delta) == run["delta"]
        assert np.array_equal(p.u[goldens["maps"]["basic/sample_idx"]], goldens["maps"][f"basic/samples_{eps}"])
    assert np.array_equal(p.u, goldens["maps"]["basic/converged_1e-06"])


def test_oracle_matches_compiled_reference_random():
    """Bit-for-bit against oracle/_ref (the reference's own sources) on fresh random cases, incl. n = 3 and the
    single-step entry points.  Skipped where the reference was never compiled."""
    ref = O.ref()
    if ref is None:
        pytest.skip("oracle/_ref/libepic_ref.so not built (no /root/reference)")
    lib = O.oracle()
    rng = np.random.default_rng(7)
    for trial in range(12):
        n = 2 if trial % 3 else 3
        m = rng.integers(3, 40 if n == 2 else 14, size=n)
        u0, locked = O.oracle_synthetic(m, seed=100 + trial, density=float(rng.uniform(0, 0.35)))
        # a few extra goals so the field is not trivially symmetric
        free = np.flatnonzero(locked == 0)
        if free.size:
            extra = rng.choice(free, size=min(3, free.size), replace=False)
            u0[extra] = 0.0
            locked[extra] = 1
        a = O.Problem(m, u0, locked, 1e-5, int(rng.integers(1, 20)))
        b = a.clone()
        assert ref.harmonic_complete_cpu(ct.byref(a.h)) == 0
        assert lib.oracle_complete(ct.byref(b.h)) == 0
        assert a.h.currentIteration == b.h.currentIteration and a.h.delta == b.h.delta
        assert np.array_equal(a.u, b.u)
        a, b = O.Problem(m, u0, locked), O.Problem(m, u0, locked)
        for i in range(7):
            ra = (ref.harmonic_update_and_check_cpu if i % 3 == 0 else ref.harmonic_update_cpu)(ct.byref(a.h))
            rb = (lib.oracle_update_and_check if i % 3 == 0 else lib.oracle_update)(ct.byref(b.h))
            assert ra == rb and np.array_equal(a.u, b.u)


def test_oracle_set_cells_matches_golden(goldens):
    g = goldens["small"]
    p = O.Problem(g["set_cells/m"], g["set_cells/u0"], g["set_cells/locked0"])
    v, t = g["set_cells/v"].astype(np.uint32), g["set_cells/types"].astype(np.uint32)
    rc = O.oracle().oracle_set_cells_2d(ct.byref(p.h), len(t), v.ctypes.data_as(ct.POINTER(ct.c_uint)),
                                        t.ctypes.data_as(ct.POINTER(ct.c_uint)))
    assert rc == 0
    assert np.array_equal(p.u, g["set_cells/u1"]) and np.array_equal(p.locked, g["set_cells/locked1"])


def test_jacobi_and_red_black_reach_the_same_field():
    """with the same per-cell arithmetic the sweep order does not move the f32 stagnation point.
    This is what lets a Jacobi GPU result be compared with the red-black reference at convergence."""
    lib = O.oracle()
    u0, locked = O.oracle_synthetic([96, 80], seed=5, density=0.05)
    a = O.Problem([96, 80], u0, locked, 1e-6, 10)
    b = a.clone()
    assert lib.oracle_complete(ct.byref(a.h)) == 0
    assert lib.oracle_jacobi_complete(ct.byref(b.h)) == 0
    reach = (a.locked == 0) & (a.u > -9e5)
    assert np.array_equal(a.u <= -9e5, b.u <= -9e5)
    rel = np.abs(a.u[reach] - b.u[reach]) / np.maximum(1.0, np.abs(a.u[reach]))
    assert rel.max() < 2e-6


@pytest.mark.parametrize("name", ["g2d_16", "g2d_23x37", "g2d_8x300", "g2d_70x66_dense", "g3d_8", "g3d_7x9x11", "g3d_20x12x34"])
def test_jacobi_checker_contains_the_reference_half_sweeps(goldens, name):
    """Pins the JACOBI form of the checker (what the default GPU path is compared with) to reference-generated vectors:
    the grid is bipartite, so a Jacobi run is two interleaved red-black chains, and the chain in phase with the reference
    (harmonic_cpu.cpp:46-51, :88-93) reproduces its half-sweeps -- after k sweeps the cells of the colour updated last
    equal the reference's field after k iterations, bit for bit."""
    g = goldens["small"]
    m = [int(v) for v in g[name + "/m"]]
    u0, locked = g[name + "/u0"], g[name + "/locked"]
    idx = np.indices(m).sum(axis=0)
    interior = np.ones(m, dtype=bool)
    for ax in range(len(m)):
        sl = [slice(None)] * len(m)
        sl[ax] = [0, m[ax] - 1]
        interior[tuple(sl)] = False
    free = interior & (np.asarray(locked).reshape(m) == 0)
    for k in (1, 2, 3, 10):
        p = O.Problem(m, u0, locked)
        assert O.oracle().oracle_jacobi_run(ct.byref(p.h), k) == 0
        last = ((idx + (k - 1)) % 2 == 1) if len(m) == 2 else ((idx + (k - 1)) % 2 == 0)
        sel = free & last
        assert sel.any() and np.array_equal(p.u.reshape(m)[sel], np.asarray(g[f"{name}/rb{k}"]).reshape(m)[sel])


def test_parallel_half_sweeps_equal_the_sequential_ones():
    """oracle_update_parallel_2d (the all-cores CPU figure of bench.py): rows of a red-black half-sweep are independent,
    so 1, 3 and 8 threads give the sequential result bit for bit."""
    from epic_amd.synthetic import synthetic_grid

    lib = O.oracle()
    lib.oracle_update_parallel_2d.argtypes = (ct.c_void_p, ct.c_int)
    m = [97, 203]
    u0, locked = synthetic_grid(m, 3, 0.07)
    seq = O.Problem(m, u0, locked)
    for _ in range(31):
        lib.oracle_update(ct.byref(seq.h))
    for threads in (1, 3, 8):
        par = O.Problem(m, u0, locked)
        for _ in range(31):
            assert lib.oracle_update_parallel_2d(ct.byref(par.h), threads) == 0
        assert par.h.currentIteration == 31 and np.array_equal(par.u, seq.u)


def test_parallel_complete_is_the_reference_loop_bit_for_bit(goldens):
    """oracle_complete_parallel_2d -- harmonic_complete_cpu's loop with the half-sweeps dealt to OpenMP threads, what states the reference's
    converged field at the benchmark's own size (tests/golden/generate_8192_golden.py) -- against a field the REFERENCE itself converged: the
    benchmark's grid family at 512^2 (tests/golden/generate_synthetic_goldens.py): iteration count, delta, sha256 of the field; and against the
    sequential checker on a ragged grid with another check interval."""
    import ctypes as ct
    import hashlib

    from epic_amd.synthetic import synthetic_grid

    lib = O.oracle()
    lib.oracle_complete_parallel_2d.argtypes = (ct.POINTER(O.CHarmonic), ct.c_int, ct.c_uint)
    lib.oracle_complete_parallel_2d.restype = ct.c_int
    info = goldens["manifest"]["synthetic"]["grids"]["512"]
    u0, locked = synthetic_grid([512, 512])
    p = O.Problem([512, 512], u0, locked, info["epsilon"], info["stagger"])
    assert lib.oracle_complete_parallel_2d(ct.byref(p.h), 4, 0) == 0
    assert p.h.currentIteration == info["iterations"] and float(p.h.delta) == info["delta"]
    assert hashlib.sha256(p.u.tobytes()).hexdigest() == info["sha_u"]
    u0, locked = synthetic_grid([97, 211], 5, 0.1)
    a, b = O.Problem([97, 211], u0, locked, 1e-4, 7), O.Problem([97, 211], u0, locked, 1e-4, 7)
    assert lib.oracle_complete_parallel_2d(ct.byref(a.h), 3, 0) == 0 and lib.oracle_complete(ct.byref(b.h)) == 0
    assert a.h.currentIteration == b.h.currentIteration and float(a.h.delta) == float(b.h.delta) and np.array_equal(a.u, b.u)


@pytest.mark.parametrize("name", ["g3d_8", "g3d_16", "g3d_7x9x11", "g3d_20x12x34"])
def test_parallel_complete_3d_is_the_reference_loop_bit_for_bit(goldens, name):
    """oracle_complete_parallel_3d (the 7-point half-sweeps dealt to threads; what states the reference's converged 512^3 field of BASELINE
    configs[4], tests/golden/synthetic_512cubed.json) against the fields the REFERENCE converged: iteration count, delta, every cell."""
    import ctypes as ct

    lib = O.oracle()
    lib.oracle_complete_parallel_3d.argtypes = (ct.POINTER(O.CHarmonic), ct.c_int, ct.c_uint)
    lib.oracle_complete_parallel_3d.restype = ct.c_int
    g, info = goldens["small"], goldens["manifest"]["small"][name]
    p = O.Problem([int(x) for x in g[name + "/m"]], g[name + "/u0"], g[name + "/locked"], info["epsilon"], info["stagger"])
    assert lib.oracle_complete_parallel_3d(ct.byref(p.h), 3, 0) == 0
    assert p.h.currentIteration == info["iterations"] and float(p.h.delta) == info["delta"]
    assert np.array_equal(p.u, np.ravel(g[name + "/converged"]))
