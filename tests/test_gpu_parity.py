"""Parity tests proper: the HIP path, called through the C-ABI, against the checker (oracle/) and the committed
golden vectors that the reference produced.  All need a real MI355X.

Tolerances (floating point; stated by BASELINE.json's north star as "match CPU to 1e-5", made relative because an
absolute 1e-5 is below one f32 ulp wherever |u| > 128):
  * fixed sweep count, HIP Jacobi vs oracle Jacobi (identical scheme and, in the default precise math mode, the same
    expf/logf algorithm as the host libm): bit-identical, tolerance 0
  * converged, HIP Jacobi vs the reference's red-black result at epsilon = 1e-6:
        |du| <= 1e-5 * max(1, |u|)   over reachable free cells; unreachable cells must be exactly -1e6 on both sides.
"""
import ctypes as ct
import os

import numpy as np
import pytest

import _oracle as O
from epic_amd import epic_harmonic as eh
from epic_amd.harmonic import Harmonic
from epic_amd.harmonic_map import HarmonicMap
from epic_amd.synthetic import synthetic_grid

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(300)]

E = eh._epic
NT = 1024
UP = ct.POINTER(ct.c_uint)
FIXED_TOL = 0.0   # precise math: bit-identical to the checker's Jacobi, sweep for sweep
CONVERGED_TOL = 1e-5


@pytest.fixture(autouse=True, params=["tracking_auto", "tracking_on"])
def tracking_mode(request):
    """Every test of this file runs twice: with the library default (activity tracking only above 4 Mcell, so not on
    these grids) and with EPIC_HIP_TRACK=1 (work lists on every 2-D grid).  Results must not depend on it."""
    if request.param == "tracking_on":
        os.environ["EPIC_HIP_TRACK"] = "1"
    yield request.param
    os.environ.pop("EPIC_HIP_TRACK", None)


def make(m, u, locked, eps=1e-6, stagger=100):
    h = Harmonic()
    h.set_grid(m, u, locked)
    h.epsilon = eps
    h.numIterationsToStaggerCheck = stagger
    return h


def gpu_init(h):
    assert E.harmonic_initialize_dimension_size_gpu(h) == 0
    assert E.harmonic_initialize_potential_values_gpu(h) == 0
    assert E.harmonic_initialize_locked_gpu(h) == 0
    assert E.harmonic_initialize_gpu(h, NT) == 0
    assert h.d_m and h.d_u and h.d_locked and h.d_delta


def gpu_fini(h):
    assert E.harmonic_uninitialize_gpu(h) == 0
    assert E.harmonic_uninitialize_dimension_size_gpu(h) == 0
    assert E.harmonic_uninitialize_potential_values_gpu(h) == 0
    assert E.harmonic_uninitialize_locked_gpu(h) == 0
    assert not h.d_m and not h.d_u and not h.d_locked and not h.d_delta


def gpu_sweeps(m, u0, locked, k, rows_per_task=0):
    """k Jacobi sweeps through the fine-grained ABI (the navigation node's flow); last one is a check sweep."""
    h = make(m, u0, locked)
    gpu_init(h)
    if rows_per_task:
        assert E.epic_hip_set_rows_per_task(h, rows_per_task) == 0
    for i in range(k):
        rc = (E.harmonic_update_and_check_gpu if i == k - 1 else E.harmonic_update_gpu)(h, NT)
        assert rc in (0, 1)
    assert h.currentIteration == k
    assert E.harmonic_get_potential_values_gpu(h) == 0
    delta = float(h.delta)
    gpu_fini(h)
    return h.u_array().ravel().copy(), delta


def oracle_jacobi(m, u0, locked, k):
    """k iterations of the SESSION's scheme by the checker (tests/conftest.py: Jacobi unless EPIC_TEST_SCHEME=default, then the
    library default, the reference's red-black) -- what gpu_sweeps(), which sets no scheme, must reproduce."""
    p = O.Problem(m, u0, locked)
    assert O.run_session(p, k) == 0
    return p.u, float(p.h.delta)


def assert_close(got, want, locked, tol, what):
    got, want, locked = np.ravel(got), np.ravel(want), np.ravel(locked)
    lk = locked != 0
    assert np.array_equal(got[lk], want[lk]), what + ": locked cells must be untouched"
    unreached = want <= -9e5
    assert np.array_equal(got[unreached], want[unreached]), what + ": unreached cells must stay at the seed"
    err = np.abs(got.astype(np.float64) - want) / np.maximum(1.0, np.abs(want))
    worst = float(err.max()) if err.size else 0.0
    assert worst <= tol, f"{what}: max rel err {worst:.3e} > {tol:g} at {int(err.argmax())}"
    return worst


GRIDS_2D = [
    ([16, 16], 1, 0.05), ([23, 37], 4, 0.10), ([3, 3], 6, 0.0), ([3, 70], 6, 0.0), ([70, 3], 6, 0.0),
    ([8, 300], 7, 0.05), ([70, 66], 8, 0.30), ([257, 513], 9, 0.05), ([64, 1030], 10, 0.05), ([130, 256], 11, 0.05),
    ([100, 255], 12, 0.05), ([41, 257], 13, 0.02),
]


@pytest.mark.parametrize("m,seed,dens", GRIDS_2D)
def test_fixed_sweeps_2d_vs_oracle_jacobi(m, seed, dens):
    u0, locked = synthetic_grid(m, seed, dens)
    # two more goals off-centre so every strip/seam sees a moving front
    free = np.flatnonzero(locked == 0)
    if free.size > 4:
        for idx in (free[0], free[-1]):
            u0[idx] = 0.0
            locked[idx] = 1
    for k in (1, 2, 5, 40):
        got, gdelta = gpu_sweeps(m, u0, locked, k)
        want, wdelta = oracle_jacobi(m, u0, locked, k)
        assert_close(got, want, locked, FIXED_TOL, f"{m} after {k} sweeps")
        assert gdelta == wdelta, (gdelta, wdelta)


@pytest.mark.parametrize("m,rpt", [([66000, 300], 0), ([6, 80000], 0), ([1200, 9000], 64), ([40000, 520], 5),
                                   ([200, 700], 16), ([96, 300], 8), ([33, 257], 4)])
def test_extreme_aspect_ratios(m, rpt):
    """More rows than a grid dimension may hold in y (65535), hundreds of strips per row, the tallest tasks: the buffer
    addressing (32-bit row offsets from a per-task base), the row-per-block mask packing and the scalar row sides must
    hold on all of them.  Goals at both ends so that every part of the grid sees moving values.  The small grids with
    multi-row tasks run the pipelined four-row path with a single wave per SIMD, i.e. with a wave's instructions issuing
    back to back -- the setting in which a missing wait state (EXEC write -> DPP, cell_update.h) shows; the tall ones
    make a wave of a list-driven launch walk several tiles."""
    u0, locked = synthetic_grid(m, 17, 0.05)
    free = np.flatnonzero(locked == 0)
    for idx in (free[0], free[free.size // 2], free[-1]):
        u0[idx] = 0.0
        locked[idx] = 1
    got, gdelta = gpu_sweeps(m, u0, locked, 6, rows_per_task=rpt)
    want, wdelta = oracle_jacobi(m, u0, locked, 6)
    assert_close(got, want, locked, FIXED_TOL, f"{m} after 6 sweeps")
    assert gdelta == wdelta, (gdelta, wdelta)


@pytest.mark.parametrize("rpt", [1, 3, 8, 13, 64, 1000])
def test_rows_per_task_is_only_a_tiling_choice(rpt):
    m = [77, 300]
    u0, locked = synthetic_grid(m, 5, 0.05)
    ref, _ = gpu_sweeps(m, u0, locked, 9, rows_per_task=8)
    got, _ = gpu_sweeps(m, u0, locked, 9, rows_per_task=rpt)
    assert np.array_equal(ref, got)


GRIDS_3D = [([8, 8, 8], 11, 0.05), ([7, 9, 11], 13, 0.10), ([20, 12, 34], 14, 0.05), ([3, 3, 3], 1, 0.0),
            ([6, 40, 300], 15, 0.05), ([9, 5, 64], 16, 0.05)]


@pytest.mark.parametrize("m,seed,dens", GRIDS_3D)
def test_fixed_sweeps_3d_vs_oracle_jacobi(m, seed, dens):
    u0, locked = synthetic_grid(m, seed, dens)
    for k in (1, 2, 7, 30):
        got, gdelta = gpu_sweeps(m, u0, locked, k)
        want, wdelta = oracle_jacobi(m, u0, locked, k)
        assert_close(got, want, locked, FIXED_TOL, f"{m} after {k} sweeps")
        assert gdelta == wdelta, (gdelta, wdelta)


SMALL = ["g2d_16", "g2d_32", "g2d_64", "g2d_23x37", "g2d_5x7", "g2d_3x3", "g2d_8x300", "g2d_70x66_dense",
         "g3d_8", "g3d_16", "g3d_7x9x11", "g3d_20x12x34"]


@pytest.mark.parametrize("name", SMALL)
def test_jacobi_sweeps_contain_the_reference_half_sweeps(goldens, name):
    """Link between the default Jacobi path and REFERENCE-generated vectors.  The grid is bipartite, so a Jacobi run is two
    interleaved red-black chains; the chain that updates, at sweep k, the colour the reference updates at iteration k
    (harmonic_cpu.cpp:46-51 / :88-93) starts from the same values and sees the same neighbours, so after k sweeps the
    cells of the colour last updated hold exactly what the reference's k-th half-sweep wrote: bit for bit, 2-D and 3-D."""
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
        got, _ = gpu_sweeps(m, u0, locked, k)
        # iteration k - 1 was the last one: 2-D updates (x0 + x1 + iteration) odd, 3-D (x0 + x1 + x2 + iteration) even
        last = ((idx + (k - 1)) % 2 == 1) if len(m) == 2 else ((idx + (k - 1)) % 2 == 0)
        ref = np.asarray(g[f"{name}/rb{k}"]).reshape(m)
        sel = free & last
        assert np.array_equal(np.asarray(got).reshape(m)[sel], ref[sel]), f"{name}: colour of half-sweep {k}"


@pytest.mark.parametrize("name", SMALL)
def test_complete_gpu_vs_reference_golden(goldens, name):
    """harmonic_complete_gpu (the plugin's one-shot call) converged at eps = 1e-6 vs the reference's field."""
    g, info = goldens["small"], goldens["manifest"]["small"][name]
    m = g[name + "/m"]
    h = make(m, g[name + "/u0"], g[name + "/locked"], info["epsilon"], info["stagger"])
    assert E.harmonic_complete_gpu(h, NT) == 0
    assert not h.d_m and not h.d_u and not h.d_locked and not h.d_delta
    assert h.currentIteration >= max(m) and h.currentIteration % info["stagger"] == 1 % info["stagger"]
    assert h.delta < info["epsilon"]
    assert_close(h.u_array(), g[name + "/converged"], g[name + "/locked"], CONVERGED_TOL, name)


@pytest.mark.parametrize("name", ["basic", "maze", "umass"])
def test_maps_converged_vs_reference_golden(goldens, name, record_property):
    """BASELINE configs 1-2: the reference's own maps, python flow (Harmonic.solve(process='gpu')), eps = 1e-6."""
    want = goldens["maps"][name + "/converged_1e-06"]
    h = HarmonicMap().load(os.path.join(O.ROOT, "tests", "golden", "maps", name + ".png"))
    h.solve(process="gpu", epsilon=1e-6)
    worst = assert_close(h.u_array(), want, h.locked_array(), CONVERGED_TOL, name)
    free = h.locked_array().ravel() == 0
    absmax = float(np.abs(h.u_array().ravel()[free] - want[free]).max())
    record_property("max_rel_err", worst)
    record_property("max_abs_err", absmax)
    print(f"{name}: sweeps {h.currentIteration}, delta {h.delta:.3e}, max rel {worst:.3e}, max abs {absmax:.3e}")
    # The Jacobi run contains the reference's red-black chain (see test_jacobi_sweeps_contain_the_reference_half_sweeps):
    # it stops after the same number of iterations, and the cells of the colour updated last are the reference's, exactly.
    run = goldens["manifest"]["maps"][name]["runs"]["1e-06"]
    assert h.currentIteration == run["iterations"]
    rows, cols = h.shape
    rr, cc = np.indices((rows, cols), sparse=True)
    last = ((rr + cc + (h.currentIteration - 1)) % 2 == 1) & (h.locked_array() == 0)
    assert np.array_equal(h.u_array()[last], want.reshape(rows, cols)[last])


def test_execute_gpu_validation_and_lifecycle(capfd):
    u0, locked = synthetic_grid([20, 20], 1, 0.05)
    h = make([20, 20], u0, locked, eps=1e-4, stagger=10)
    assert E.harmonic_execute_gpu(h, NT) == eh.EPIC_ERROR_INVALID_DATA          # nothing on the device yet
    assert E.harmonic_initialize_dimension_size_gpu(h) == 0
    assert E.harmonic_initialize_potential_values_gpu(h) == 0
    assert E.harmonic_initialize_locked_gpu(h) == 0
    assert E.harmonic_execute_gpu(h, 1000) == eh.EPIC_ERROR_INVALID_CUDA_PARAM  # 1000 % 32 != 0 (harmonic_gpu.cu:240)
    h.epsilon = 0.0
    assert E.harmonic_execute_gpu(h, NT) == eh.EPIC_ERROR_INVALID_DATA
    h.epsilon = 1e-4
    assert E.harmonic_initialize_gpu(h, NT) == 0
    assert E.harmonic_initialize_gpu(h, NT) == eh.EPIC_ERROR_INVALID_DATA       # d_delta already set (harmonic_gpu.cu:208)
    assert E.harmonic_uninitialize_gpu(h) == 0
    assert E.harmonic_execute_gpu(h, NT) == 0
    assert h.currentIteration % 10 == 1 and h.currentIteration >= 20 and not h.d_delta
    # re-initialising while initialised must be tolerated (python solve(): harmonic.py:67-71 then harmonic_gpu.cu:172)
    assert E.harmonic_initialize_potential_values_gpu(h) == 0
    assert E.harmonic_initialize_locked_gpu(h) == 0
    assert E.harmonic_initialize_dimension_size_gpu(h) == 0
    assert E.harmonic_uninitialize_dimension_size_gpu(h) == 0
    assert E.harmonic_uninitialize_potential_values_gpu(h) == 0
    assert E.harmonic_uninitialize_locked_gpu(h) == 0
    assert E.harmonic_uninitialize_locked_gpu(h) == 0                           # idempotent
    assert "Error[harmonic_execute_gpu]" in capfd.readouterr().err


def test_n4_is_held_on_the_device_and_counted_but_never_swept(capfd):
    """harmonic_gpu.cu:156-162, :327-336: the reference's n == 4 branches are empty -- the state goes to the device, every
    update advances currentIteration and sweeps nothing.  The same here (the CPU half: tests/test_cpu_abi.py::test_n4_...);
    harmonic_execute_gpu refuses n = 4, where the reference's loop would never return (no sweep ever lowers delta)."""
    h = Harmonic()
    u0 = np.linspace(-5.0, 0.0, 3 * 4 * 5 * 6, dtype=np.float32)
    h.set_grid([3, 4, 5, 6], u0, np.zeros(u0.size, np.uint32))
    for fn in (E.harmonic_initialize_dimension_size_gpu, E.harmonic_initialize_potential_values_gpu, E.harmonic_initialize_locked_gpu):
        assert fn(h) == 0, fn.__name__
    assert h.d_m and h.d_u and h.d_locked
    assert E.harmonic_initialize_gpu(h, NT) == 0 and h.d_delta
    h.delta = 7.0
    for k in range(1, 4):
        assert E.harmonic_update_gpu(h, NT) == 0 and h.currentIteration == k
    assert E.harmonic_update_and_check_gpu(h, NT) == 0 and h.currentIteration == 4 and h.delta == 7.0   # (delta untouched, as on the CPU)
    assert E.epic_hip_update_n_gpu(h, 10, 0) == 0 and h.currentIteration == 14
    h.u_array().ravel()[:] = 1.0
    assert E.harmonic_get_potential_values_gpu(h) == 0 and np.array_equal(h.u_array().ravel(), u0)       # what was uploaded
    assert E.harmonic_execute_gpu(h, NT) == eh.EPIC_ERROR_INVALID_DATA
    assert "n = 4 is a counting no-op" in capfd.readouterr().err
    for fn in (E.harmonic_uninitialize_gpu, E.harmonic_uninitialize_dimension_size_gpu, E.harmonic_uninitialize_potential_values_gpu,
               E.harmonic_uninitialize_locked_gpu):
        assert fn(h) == 0, fn.__name__
    assert not h.d_m and not h.d_u and not h.d_locked and not h.d_delta


def test_navigation_node_flow_set_cells_and_mid_solve_readback():
    """src/epic_navigation_node_harmonic.cpp:165-189, :357-380, :522-542: update(k) batches, live cell edits on the
    resident state, full-field readback between batches; then update_model re-upload."""
    m = [48, 300]
    u0, locked = synthetic_grid(m, 3, 0.05)
    h = make(m, u0, locked)
    gpu_init(h)
    p = O.Problem(m, u0, locked)
    lib = O.oracle()

    def batch(k):
        rc = E.harmonic_update_and_check_gpu(h, NT)
        assert rc in (0, 1)
        assert E.epic_hip_update_n_gpu(h, k - 1, 0) == 0
        O.run_session(p, k)

    batch(10)
    assert E.harmonic_get_potential_values_gpu(h) == 0
    assert_close(h.u_array(), p.u, p.locked, FIXED_TOL, "after first batch")

    # edits: new goal, new obstacle, freed obstacle, out-of-range (skipped), invalid type (skipped), border cell
    obst = np.argwhere(p.locked.reshape(m)[1:-1, 1:-1] == 1)[0] + 1
    v = np.array([[250, 40], [10, 5], [obst[1], obst[0]], [300, 3], [5, 5], [0, 7]], dtype=np.uint32)
    t = np.array([0, 1, 2, 0, 9, 2], dtype=np.uint32)
    args = (len(t), v.ctypes.data_as(UP), t.ctypes.data_as(UP))
    assert E.harmonic_utilities_set_cells_2d_cpu(h, *args) == 0
    assert E.harmonic_utilities_set_cells_2d_gpu(h, NT, *args) == 0
    assert lib.oracle_set_cells_2d(ct.byref(p.h), *args) == 0
    assert E.harmonic_utilities_set_cells_2d_gpu(h, NT, 0, args[1], args[2]) == eh.EPIC_ERROR_INVALID_DATA
    # the host arrays were edited by the _cpu call; overwrite the edited entries on the oracle side identically,
    # then carry the GPU's u over (node semantics: device state is authoritative between edits)
    batch(25)
    assert E.harmonic_get_potential_values_gpu(h) == 0
    assert_close(h.u_array(), p.u, p.locked, FIXED_TOL, "after edits")
    assert np.array_equal(h.locked_array().ravel(), p.locked)

    # update_model: push a modified host model (harmonic_model_gpu.cu:172-204)
    hu = h.u_array().ravel()
    hu[:] = u0
    h.locked_array().ravel()[:] = locked
    assert E.harmonic_update_model_gpu(h) == 0
    p2 = O.Problem(m, u0, locked)
    p2.h.currentIteration = h.currentIteration     # (red-black: the iteration number selects the colour)
    assert E.epic_hip_update_n_gpu(h, 6, 1) in (0, 1)
    O.run_session(p2, 6)
    assert E.harmonic_get_potential_values_gpu(h) == 0
    assert_close(h.u_array(), p2.u, p2.locked, FIXED_TOL, "after update_model")
    gpu_fini(h)


def test_full_size_8192_window_property():
    """BASELINE config 3 at full size.  After K sweeps only cells within K of the goal can have moved, so the
    8192^2 result restricted to a window around the goal must equal the checker run on that window alone, and every
    cell outside the window must still hold its seed.  The goal (4096, 4096) sits on a strip seam (4096 = 16 * 256)."""
    n, K, W = 8192, 24, 64
    u0, locked = synthetic_grid([n, n])
    h = make([n, n], u0, locked)
    gpu_init(h)
    rc = E.epic_hip_update_n_gpu(h, K, 1)
    assert rc in (0, 1)
    assert E.harmonic_get_potential_values_gpu(h) == 0
    gpu_fini(h)
    got = h.u_array()
    c = n // 2
    win = (slice(c - W, c + W), slice(c - W, c + W))
    lw = locked.reshape(n, n)[win].copy()
    uw = u0.reshape(n, n)[win].copy()
    p = O.Problem([2 * W, 2 * W], uw, lw)
    O.run_session(p, K)
    assert_close(got[win], p.u, lw, FIXED_TOL, "window around the goal")
    outside = np.ones((n, n), dtype=bool)
    outside[win] = False
    assert np.all(got[outside] == np.float32(-1e6))
    assert np.array_equal(got.ravel()[locked != 0], u0[locked != 0])
    moved = int((got != u0.reshape(n, n)).sum())
    assert 0 < moved <= (2 * K + 1) ** 2
    assert abs(h.delta - p.h.delta) <= 1e-5 * max(1.0, p.h.delta)


@pytest.mark.parametrize("scheme", [0, 1])
def test_full_size_8192_tracking_and_tiling_leave_no_trace(scheme, tracking_mode):
    """BASELINE config 3 at full size, properties that need no reference run: the field after 700 iterations, a live-map
    edit far from the front and 300 more iterations is the same array, bit for bit, (a) with work-list tracking and
    with every tile recomputed every time, (b) with 8 and with 24 rows per task; the maximum principle holds (u <= 0,
    goals stay 0, obstacles stay at the seed); and what the front has not reached has not moved."""
    if tracking_mode == "tracking_on":
        pytest.skip("the test sets the tracking mode itself")
    n = 8192
    u0, locked = synthetic_grid([n, n])
    v = np.array([700, 900, 7000, 7100], dtype=np.uint32)            # (x, y) pairs: a new goal and a new obstacle
    types = np.array([eh.EPIC_CELL_TYPE_GOAL, eh.EPIC_CELL_TYPE_OBSTACLE], dtype=np.uint32)
    fields = []
    for track, rpt in ((1, 0), (0, 0), (1, 24)):
        h = make([n, n], u0, locked)
        gpu_init(h)
        assert E.epic_hip_set_scheme(h, scheme) == 0 and E.epic_hip_set_activity_tracking(h, track) == 0
        if rpt:
            assert E.epic_hip_set_rows_per_task(h, rpt) == 0
        assert E.epic_hip_update_n_gpu(h, 700, 1) in (0, 1)
        d1 = float(h.delta)
        if track:
            a, t = ct.c_ulonglong(0), ct.c_ulonglong(0)
            assert E.epic_hip_activity_stats(h, ct.byref(a), ct.byref(t)) == 0
            assert 0 < a.value < t.value // 4, (a.value, t.value)     # the front is ~700 cells out: most tiles sleep
        assert E.harmonic_utilities_set_cells_2d_gpu(h, NT, 2, v.ctypes.data_as(eh._UP), types.ctypes.data_as(eh._UP)) == 0
        assert E.epic_hip_update_n_gpu(h, 300, 1) in (0, 1)
        assert E.harmonic_get_potential_values_gpu(h) == 0
        fields.append((h.u_array().copy(), d1, float(h.delta)))
        gpu_fini(h)
    for other in fields[1:]:
        assert np.array_equal(fields[0][0], other[0]) and fields[0][1:] == other[1:]
    got = fields[0][0]
    lk = locked.reshape(n, n).copy()
    lk[900, 700] = lk[7100, 7000] = 1
    assert got.max() == 0.0 and got[900, 700] == 0.0 and got[7100, 7000] == np.float32(-1e6)
    goals = (u0.reshape(n, n) == 0.0)
    assert np.all(got[goals] == 0.0)
    obstacles = (locked.reshape(n, n) != 0) & ~goals
    assert np.all(got[obstacles] == np.float32(-1e6))
    c = n // 2
    far = np.ones((n, n), dtype=bool)
    far[c - 1001:c + 1002, c - 1001:c + 1002] = False               # the first goal's reach after 1000 iterations
    far[900 - 301:900 + 302, 700 - 301:700 + 302] = False           # the new goal's reach after 300
    assert np.all(got[far & (lk == 0)] == np.float32(-1e6))
    moved = (got != np.float32(-1e6)) & (lk == 0)
    assert moved[c - 900:c + 900, c].any() and moved[900 - 200:900 + 200, 700].any()


def test_raw_operator_row_ranges_match_whole_sweep():
    """include/epic_hip.h: epic_hip_sweep_2d over [0, r) and [r, rows) == one launch over [0, rows) (slab mode)."""
    import torch

    rows, cols = 90, 500
    pitch = E.epic_hip_pitch_for_cols(cols)
    u0, locked = synthetic_grid([rows, cols], 2, 0.05)
    dev = torch.device("cuda:0")
    d_locked = torch.from_numpy(locked.astype(np.int32)).to(dev)
    maskw = torch.zeros(E.epic_hip_mask_words_2d(rows, pitch), dtype=torch.int32, device=dev)
    a = torch.full((rows, pitch), -1e6, dtype=torch.float32, device=dev)
    a[:, :cols] = torch.from_numpy(u0.reshape(rows, cols)).to(dev)
    b1, b2 = torch.zeros_like(a), torch.zeros_like(a)
    d1 = torch.zeros(1, dtype=torch.int32, device=dev)
    d2 = torch.zeros(1, dtype=torch.int32, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    assert E.epic_hip_pack_mask_2d(d_locked.data_ptr(), rows, cols, pitch, 0, 0, maskw.data_ptr(), s) == 0
    assert E.epic_hip_sweep_2d(a.data_ptr(), b1.data_ptr(), maskw.data_ptr(), rows, pitch, 0, rows, 16,
                               eh.MATH_PRECISE, d1.data_ptr(), s) == 0
    for lo, hi in ((0, 1), (1, 37), (37, 89), (89, 90)):
        assert E.epic_hip_sweep_2d(a.data_ptr(), b2.data_ptr(), maskw.data_ptr(), rows, pitch, lo, hi, 8,
                                   eh.MATH_PRECISE, d2.data_ptr(), s) == 0
    torch.cuda.synchronize()
    assert torch.equal(b1, b2) and int(d1.item()) == int(d2.item()) and int(d1.item()) != 0
    p = O.Problem([rows, cols], u0, locked)     # (epic_hip_sweep_2d IS the Jacobi sweep, whatever the session's scheme)
    assert O.oracle().oracle_jacobi_run(ct.byref(p.h), 1) == 0
    want, wdelta = p.u, float(p.h.delta)
    assert_close(b1[:, :cols].cpu().numpy(), want, locked, FIXED_TOL, "raw operator")
    assert abs(np.int32(d1.item()).view(np.float32) - wdelta) <= 1e-5 * max(1.0, wdelta)


def test_device_libm_replica_is_bit_identical_to_host_libm():
    """cell_update.h restates glibc's expf/logf (the arithmetic the reference CPU path runs on) in f64 on the device.
    Every float in [1, 6] for log (the whole range the 2-D / 3-D sums can take) and a dense sweep of [-104, 0] for exp
    must match the host's libm bit for bit (host = same glibc as the checker uses)."""
    import torch

    dev = torch.device("cuda:0")
    s = torch.cuda.current_stream().cuda_stream
    lo = np.float32(1.0).view(np.uint32)
    hi = np.float32(6.0).view(np.uint32)
    x = np.arange(lo, hi + 1, dtype=np.uint32).view(np.float32)
    d_in = torch.from_numpy(x).to(dev)
    d_out = torch.empty_like(d_in)
    assert E.epic_hip_eval_math(d_in.data_ptr(), d_out.data_ptr(), x.size, 1, s) == 0
    torch.cuda.synchronize()
    got = d_out.cpu().numpy()
    want = np.log(x.astype(np.float32))   # numpy float32 log is NOT libm; compare through the checker's libm instead
    import ctypes as ct2
    libm = ct2.CDLL("libm.so.6")
    libm.logf.restype = ct2.c_float
    libm.logf.argtypes = (ct2.c_float,)
    libm.expf.restype = ct2.c_float
    libm.expf.argtypes = (ct2.c_float,)
    # spot-check 200k inputs through ctypes (slow per call), all inputs against float64 rounding bounds
    rng = np.random.default_rng(0)
    pick = rng.choice(x.size, size=200000, replace=False)
    ref = np.array([libm.logf(float(v)) for v in x[pick]], dtype=np.float32)
    assert np.array_equal(got[pick].view(np.uint32), ref.view(np.uint32))
    err_ulp = np.abs(got.astype(np.float64) - np.log(x.astype(np.float64))) / np.spacing(np.abs(want).clip(1e-30))
    assert err_ulp.max() < 0.82   # glibc documents 0.818 ulp for logf

    xe = -np.abs(rng.standard_cauchy(4_000_000).astype(np.float32)).clip(0, 103.0)
    xe[:1000] = -np.linspace(0, 103, 1000, dtype=np.float32)
    d_in = torch.from_numpy(xe).to(dev)
    d_out = torch.empty_like(d_in)
    assert E.epic_hip_eval_math(d_in.data_ptr(), d_out.data_ptr(), xe.size, 0, s) == 0
    torch.cuda.synchronize()
    got = d_out.cpu().numpy()
    pick = rng.choice(xe.size, size=200000, replace=False)
    ref = np.array([libm.expf(float(v)) for v in xe[pick]], dtype=np.float32)
    assert np.array_equal(got[pick].view(np.uint32), ref.view(np.uint32))
    exact = np.exp(xe.astype(np.float64))
    normal = exact > 1.2e-38
    err_ulp = np.abs(got.astype(np.float64) - exact)[normal] / np.spacing(exact[normal].astype(np.float32)).astype(np.float64)
    assert err_ulp.max() < 0.51
    assert got[xe == 0.0].tolist() == [1.0] * int((xe == 0.0).sum())


def test_fast_math_mode_is_close_but_not_the_parity_mode():
    """EPIC_HIP_MATH=fast / epic_hip_set_math_mode(1): hardware v_exp_f32 / v_log_f32.  Documented as ~1e-7 per sweep."""
    m = [60, 300]
    u0, locked = synthetic_grid(m, 2, 0.05)
    h = make(m, u0, locked)
    gpu_init(h)
    assert E.epic_hip_set_math_mode(h, eh.MATH_FAST) == 0
    assert E.epic_hip_update_n_gpu(h, 20, 1) in (0, 1)
    assert E.harmonic_get_potential_values_gpu(h) == 0
    gpu_fini(h)
    want, _ = oracle_jacobi(m, u0, locked, 20)
    assert_close(h.u_array(), want, locked, 2e-6, "fast math, 20 sweeps")


def test_slab_solver_hip_backend_single_rank():
    """epic_amd/slab.py with its real (HIP) backend on one GPU: boundary-first ordering, second stream, event join.
    world = 1 has no neighbours, so this checks the kernel plumbing; the exchange logic is covered by the gloo tests."""
    import torch

    from epic_amd.slab import SlabSolver

    grid = [150, 520]
    s = SlabSolver(grid, 0, 1, device=torch.device("cuda:0"), stagger=10)
    s.load_synthetic(seed=4, density=0.05)
    for i in range(23):
        s.sweep(check=(i == 22))
    delta = s.reduce_delta()
    u0, locked = synthetic_grid(grid, 4, 0.05)
    p = O.Problem(grid, u0, locked)     # (SlabSolver's own default scheme is Jacobi, whatever the session's)
    assert O.oracle().oracle_jacobi_run(ct.byref(p.h), 23) == 0
    want, wdelta = p.u, float(p.h.delta)
    assert np.array_equal(s.owned().ravel(), want)
    assert delta == wdelta


def test_removed_and_unknown_math_modes_are_refused():
    """Mode 3 was round 1's df32 arithmetic (removed); anything outside the documented set is INVALID_DATA, never a silent
    fall-back to another arithmetic."""
    u0, locked = synthetic_grid([20, 300], 1, 0.05)
    h = make([20, 300], u0, locked)
    gpu_init(h)
    for mode in (3, 5, -1):
        assert E.epic_hip_set_math_mode(h, mode) == eh.EPIC_ERROR_INVALID_DATA
    for mode in (eh.MATH_PRECISE, eh.MATH_TOL, eh.MATH_FAST):
        assert E.epic_hip_set_math_mode(h, mode) == 0
    gpu_fini(h)


# ---- red-black scheme (EPIC_HIP_SCHEME=redblack / epic_hip_set_scheme(h, 1)): the reference's own iteration, in place.
# With the precise math every half-sweep is the reference CPU solver's bit for bit, so the comparison is with the
# REFERENCE-GENERATED goldens directly, tolerance 0, including iteration counts and the final delta.
@pytest.fixture
def redblack_env():
    from conftest import scheme_env

    with scheme_env("redblack"):
        yield


SMALL_2D = SMALL  # 2-D and 3-D: both have the reference colouring on the device


@pytest.mark.parametrize("name", SMALL_2D)
def test_redblack_half_sweeps_equal_reference_golden(goldens, name):
    g = goldens["small"]
    m, u0, locked = g[name + "/m"], g[name + "/u0"], g[name + "/locked"]
    for k in (1, 2, 3, 10):
        h = make(m, u0, locked)
        gpu_init(h)
        assert E.epic_hip_set_scheme(h, eh.SCHEME_REDBLACK) == 0
        for i in range(k):
            rc = (E.harmonic_update_and_check_gpu if i == k - 1 else E.harmonic_update_gpu)(h, NT)
            assert rc in (0, 1)
        assert E.harmonic_get_potential_values_gpu(h) == 0
        gpu_fini(h)
        assert np.array_equal(h.u_array().ravel(), g[f"{name}/rb{k}"]), f"{name}: field after {k} half-sweeps"
        assert np.float32(h.delta) == g[f"{name}/rb{k}_delta"]


@pytest.mark.parametrize("name", SMALL_2D)
def test_redblack_complete_gpu_is_the_reference_result(goldens, name, redblack_env):
    g, info = goldens["small"], goldens["manifest"]["small"][name]
    h = make(g[name + "/m"], g[name + "/u0"], g[name + "/locked"], info["epsilon"], info["stagger"])
    assert E.harmonic_complete_gpu(h, NT) == 0
    assert h.currentIteration == info["iterations"] and float(h.delta) == info["delta"]
    assert np.array_equal(h.u_array().ravel(), g[name + "/converged"])


@pytest.mark.parametrize("name", ["basic", "maze", "umass"])
def test_redblack_maps_are_the_reference_result(goldens, name, redblack_env):
    """BASELINE configs 1-2 with the reference's own scheme: same half-sweep count, same delta, same field, bit for bit,
    as harmonic_complete_cpu produced on the reference's maps (tests/golden/generate_goldens.py)."""
    run = goldens["manifest"]["maps"][name]["runs"]["1e-06"]
    h = HarmonicMap().load(os.path.join(O.ROOT, "tests", "golden", "maps", name + ".png"))
    h.solve(process="gpu", epsilon=1e-6)
    assert h.currentIteration == run["iterations"] and float(h.delta) == run["delta"]
    assert np.array_equal(h.u_array().ravel(), goldens["maps"][name + "/converged_1e-06"])


def test_redblack_random_vs_checker_and_tiling():
    m = [130, 300]
    u0, locked = synthetic_grid(m, 17, 0.07)
    p = O.Problem(m, u0, locked)
    lib = O.oracle()
    for rpt in (8, 13, 64):
        h = make(m, u0, locked)
        gpu_init(h)
        assert E.epic_hip_set_scheme(h, 1) == 0 and E.epic_hip_set_rows_per_task(h, rpt) == 0
        assert E.epic_hip_update_n_gpu(h, 37, 1) in (0, 1)
        assert E.harmonic_get_potential_values_gpu(h) == 0
        gpu_fini(h)
        if rpt == 8:
            for i in range(37):
                (lib.oracle_update_and_check if i == 36 else lib.oracle_update)(ct.byref(p.h))
        assert np.array_equal(h.u_array().ravel(), p.u) and h.delta == p.h.delta and h.currentIteration == 37


@pytest.mark.parametrize("m,rpt", [([2048, 2048], 0), ([2050, 2100], 16), ([1100, 4000], 5), ([4200, 1000], 64)])
def test_redblack_fused_pairs_equal_the_checker(m, rpt):
    """Grids >= 4 Mcell run two plain red-black iterations per launch (rb_fused2d_kernel: halo lanes, recomputed halo
    rows, ping-pong).  13 iterations = 1 check + 6 fused pairs, 14 = 1 check + 6 pairs + 1 in-place half-sweep; both
    must equal the checker's red-black half-sweeps bit for bit."""
    u0, locked = synthetic_grid(m, 23, 0.06)
    free = np.flatnonzero(locked == 0)
    for idx in (free[7], free[free.size // 3], free[-9]):   # extra goals away from the centre: every strip seam moves
        u0[idx] = 0.0
        locked[idx] = 1
    lib = O.oracle()
    for k in (13, 14):
        # (since round 6 the precise pass takes over from 5.5 Mcell, where it beats the single sweeps: asked for explicitly on these sizes)
        saved = os.environ.get("EPIC_HIP_FUSE_MIN_CELLS")
        os.environ["EPIC_HIP_FUSE_MIN_CELLS"] = "0"
        try:
            h = make(m, u0, locked)
            gpu_init(h)
        finally:
            os.environ.pop("EPIC_HIP_FUSE_MIN_CELLS", None) if saved is None else os.environ.__setitem__("EPIC_HIP_FUSE_MIN_CELLS", saved)
        assert E.epic_hip_set_scheme(h, 1) == 0
        assert E.epic_hip_set_activity_tracking(h, 0) == 0   # the fused pass is the tracking-off path
        assert E.epic_hip_iterations_per_pass(h) == 2
        if rpt:
            assert E.epic_hip_set_rows_per_task(h, rpt) == 0
        assert E.harmonic_update_and_check_gpu(h, NT) in (0, 1)
        assert E.epic_hip_update_n_gpu(h, k - 1, 0) == 0
        assert h.currentIteration == k
        assert E.harmonic_get_potential_values_gpu(h) == 0
        gpu_fini(h)
        p = O.Problem(m, u0, locked)
        for i in range(k):
            (lib.oracle_update_and_check if i == 0 else lib.oracle_update)(ct.byref(p.h))
        assert np.array_equal(h.u_array().ravel(), p.u), f"{m} after {k} iterations"


# ---- activity tracking (on by default, so every test above already runs with it): tiles whose inputs did not change
# are skipped.  It must be invisible: same bits, same delta, same iteration counts as with tracking off and as the
# checker, across uploads, set_cells edits and mode changes; and it must actually skip something.
def _activity(h):
    a, t = ct.c_ulonglong(0), ct.c_ulonglong(0)
    assert E.epic_hip_activity_stats(h, ct.byref(a), ct.byref(t)) == 0
    return a.value, t.value


@pytest.mark.parametrize("scheme", [0, 1])
def test_activity_tracking_is_invisible_and_skips(scheme):
    m = [260, 1100]
    u0, locked = synthetic_grid(m, 31, 0.05)
    lib = O.oracle()
    p = O.Problem(m, u0, locked)
    run = lib.oracle_jacobi_run if scheme == 0 else None
    fields = {}
    for track in (1, 0):
        h = make(m, u0, locked)
        gpu_init(h)
        assert E.epic_hip_set_scheme(h, scheme) == 0 and E.epic_hip_set_rows_per_task(h, 4) == 0
        assert E.epic_hip_set_activity_tracking(h, track) == 0
        assert E.epic_hip_update_n_gpu(h, 60, 1) in (0, 1)
        if track:
            act, tiles = _activity(h)
            assert tiles == 65 * 5 and 0 < act < tiles, (act, tiles)   # the front has not reached the far strips yet
        else:
            assert _activity(h) == (0, 0)
        # an edit in a quiet corner: a new goal far from the front must start spreading at once
        v = np.array([250, 1050], dtype=np.uint32)
        types = np.array([eh.EPIC_CELL_TYPE_GOAL], dtype=np.uint32)
        assert E.harmonic_utilities_set_cells_2d_gpu(h, NT, 1, v.ctypes.data_as(eh._UP), types.ctypes.data_as(eh._UP)) == 0
        assert E.epic_hip_update_n_gpu(h, 41, 1) in (0, 1)
        d1 = float(h.delta)
        # a mode change mid-run: every tile must be recomputed under the new update rule
        assert E.epic_hip_set_math_mode(h, eh.MATH_FAST) == 0
        assert E.epic_hip_update_n_gpu(h, 3, 0) == 0
        assert E.epic_hip_set_math_mode(h, eh.MATH_PRECISE) == 0
        assert E.epic_hip_update_n_gpu(h, 700, 1) in (0, 1)
        assert E.harmonic_get_potential_values_gpu(h) == 0
        fields[track] = (h.u_array().ravel().copy(), d1, float(h.delta), int(h.currentIteration))
        gpu_fini(h)
    assert np.array_equal(fields[1][0], fields[0][0]) and fields[1][1:] == fields[0][1:]
    # and against the checker up to the mode change (the fast mode has no CPU twin)
    h = make(m, u0, locked)
    gpu_init(h)
    assert E.epic_hip_set_scheme(h, scheme) == 0 and E.epic_hip_set_rows_per_task(h, 4) == 0
    assert E.epic_hip_update_n_gpu(h, 60, 1) in (0, 1)
    v = np.array([250, 1050], dtype=np.uint32)
    types = np.array([eh.EPIC_CELL_TYPE_GOAL], dtype=np.uint32)
    assert E.harmonic_utilities_set_cells_2d_gpu(h, NT, 1, v.ctypes.data_as(eh._UP), types.ctypes.data_as(eh._UP)) == 0
    assert E.epic_hip_update_n_gpu(h, 400, 1) in (0, 1)
    assert E.harmonic_get_potential_values_gpu(h) == 0
    gpu_fini(h)
    if scheme == 0:
        assert lib.oracle_jacobi_run(ct.byref(p.h), 60) == 0
    else:
        for _ in range(60):
            lib.oracle_update(ct.byref(p.h))
    assert lib.oracle_set_cells_2d(ct.byref(p.h), 1, v.ctypes.data_as(eh._UP), types.ctypes.data_as(eh._UP)) == 0
    if scheme == 0:
        assert lib.oracle_jacobi_run(ct.byref(p.h), 400) == 0
    else:
        for i in range(400):
            (lib.oracle_update_and_check if i == 399 else lib.oracle_update)(ct.byref(p.h))
    assert np.array_equal(h.u_array().ravel(), p.u) and float(h.delta) == float(p.h.delta)


def test_activity_tracking_survives_model_reupload_and_graph_replay():
    """A small grid runs its plain sweeps as captured hipGraphs (which bake in the flag buffers); re-uploading the model
    must force full sweeps again, and the replayed graphs must keep the flag ping-pong in step."""
    m = [96, 700]
    u0, locked = synthetic_grid(m, 77, 0.06)
    lib = O.oracle()
    h = make(m, u0, locked)
    gpu_init(h)
    assert E.epic_hip_set_activity_tracking(h, 1) == 0   # (automatic mode leaves a grid this small untracked)
    for rounds in range(2):
        for n in (100, 33, 100, 8, 100):   # odd and even batch lengths, repeated so that graphs are replayed
            assert E.epic_hip_update_n_gpu(h, n, 1) in (0, 1)
        assert E.harmonic_get_potential_values_gpu(h) == 0
        p = O.Problem(m, u0, locked)
        assert O.run_session(p, 341) == 0
        assert np.array_equal(h.u_array().ravel(), p.u) and float(h.delta) == float(p.h.delta)
        h.u_array().ravel()[:] = u0   # start over from the host copy
        h.currentIteration = 0
        assert E.harmonic_update_model_gpu(h) == 0
    # changing the tiling re-allocates the flags: graphs captured for the old tiling must not be replayed afterwards
    for rpt, n in ((2, 64), (3, 64), (2, 64), (0, 64)):
        assert E.epic_hip_set_rows_per_task(h, rpt) == 0
        assert E.epic_hip_update_n_gpu(h, n, 1) in (0, 1)
    assert E.harmonic_get_potential_values_gpu(h) == 0
    p = O.Problem(m, u0, locked)
    assert O.run_session(p, 256) == 0
    assert np.array_equal(h.u_array().ravel(), p.u) and float(h.delta) == float(p.h.delta)
    gpu_fini(h)


# ---- streamlines on the device-resident field (epic_amd/csrc/path_2d.hip) ----------------------
def _resident_reference_field(goldens, name):
    """The REFERENCE's converged field of a map, uploaded as is: what the walk sees is exactly what the golden paths saw."""
    m, _, locked = O.load_png_reference_rule(os.path.join(O.ROOT, "tests", "golden", "maps", name + ".png"))
    h = make(m, goldens["maps"][name + "/converged_1e-06"], locked)
    gpu_init(h)
    return h


@pytest.mark.parametrize("name", ["basic", "umass", "maze"])
def test_device_streamlines_match_reference_goldens(goldens, name):
    import hashlib
    paths = np.load(os.path.join(O.ROOT, "tests", "golden", "paths.npz"))
    h = _resident_reference_field(goldens, name)
    PF = ct.POINTER(ct.c_float)
    for j in range(6):
        sx, sy, step, cd = (float(v) for v in paths[f"{name}/path{j}_start"])
        k, raw = ct.c_uint(0), PF()
        rc = E.epic_hip_compute_path_2d_gpu(h, sx, sy, step, cd, 1000000, ct.byref(k), ct.byref(raw))
        assert rc == int(paths[f"{name}/path{j}_rc"])
        if rc != 0:
            assert not raw
            continue
        pts = np.ctypeslib.as_array(raw, shape=(2 * k.value,)).copy()
        assert E.harmonic_free_path_cpu(ct.byref(raw)) == 0 and not raw
        key = f"{name}/path{j}"
        assert pts.size // 2 == int(paths[key + "_k"])
        assert np.array_equal(np.frombuffer(hashlib.sha256(pts.tobytes()).digest(), dtype=np.uint8), paths[key + "_sha256"])
    # a non-null *path is refused, as by the host function
    junk = (ct.c_float * 2)()
    raw = ct.cast(junk, PF)
    assert E.epic_hip_compute_path_2d_gpu(h, 5.0, 5.0, 0.5, 0.5, 10, ct.byref(ct.c_uint(0)), ct.byref(raw)) == eh.EPIC_ERROR_INVALID_DATA
    gpu_fini(h)


@pytest.mark.parametrize("name", ["umass", "maze"])
def test_device_streamline_batch_equals_host_walk(goldens, name):
    """300 random start points in one launch (free cells, obstacles, off-grid points) against harmonic_compute_path_2d_cpu
    on the same field: same code, same number of way-points, same bits."""
    h = _resident_reference_field(goldens, name)
    rows, cols = h.locked_array().shape
    rng = np.random.default_rng(99)
    n, max_len = 300, 6000
    starts = np.empty((n, 2), dtype=np.float32)
    starts[:, 0] = rng.uniform(-3.0, cols + 2.0, n)
    starts[:, 1] = rng.uniform(-3.0, rows + 2.0, n)
    k = np.zeros(n, dtype=np.uint32)
    rc = np.full(n, -1, dtype=np.int32)
    out = np.full((n, 2 * max_len), np.nan, dtype=np.float32)
    PF = ct.POINTER(ct.c_float)
    step, cd = 0.45, 0.5
    assert E.epic_hip_compute_paths_2d_gpu(h, n, starts.ctypes.data_as(PF), step, cd, max_len, k.ctypes.data_as(eh._UP),
                                           rc.ctypes.data_as(ct.POINTER(ct.c_int)), out.ctypes.data_as(PF)) == 0
    ok = 0
    for i in range(n):
        kk, raw = ct.c_uint(0), PF()
        want = E.harmonic_compute_path_2d_cpu(h, float(starts[i, 0]), float(starts[i, 1]), step, cd, max_len,
                                              ct.byref(kk), ct.byref(raw))
        assert rc[i] == want, (i, starts[i], rc[i], want)
        if want != 0:
            assert k[i] == 0
            continue
        pts = np.ctypeslib.as_array(raw, shape=(2 * kk.value,)).copy()
        E.harmonic_free_path_cpu(ct.byref(raw))
        assert k[i] == kk.value
        assert out[i, :2 * k[i]].tobytes() == pts.tobytes(), f"path {i} from {starts[i]}"
        ok += 1
    assert ok > 50   # the batch really walked a good number of paths
    gpu_fini(h)


# ---- seeded random shapes: ragged strips, one-row and one-column interiors, many goals, dense obstacles ---------------
def test_random_shapes_both_schemes_equal_the_checker():
    """40 random 2-D grids (3..150 rows, 3..900 columns: 1-4 strips, partial last strip, partial last 8-row mask group),
    random obstacle density and goal count, random sweep counts and tilings; Jacobi and red-black against the checker."""
    rng = np.random.default_rng(20240603)
    lib = O.oracle()
    for case in range(40):
        rows, cols = int(rng.integers(3, 151)), int(rng.integers(3, 901))
        m = [rows, cols]
        u0, locked = synthetic_grid(m, int(rng.integers(1, 1 << 30)), float(rng.uniform(0.0, 0.3)))
        free = np.flatnonzero(locked == 0)
        if free.size:
            for idx in rng.choice(free, size=min(free.size, int(rng.integers(0, 6))), replace=False):
                u0[idx] = 0.0
                locked[idx] = 1
        k = int(rng.integers(1, 60))
        rpt = int(rng.choice([0, 1, 2, 3, 5, 8, 16, 64]))
        scheme = case % 2
        h = make(m, u0, locked)
        gpu_init(h)
        assert E.epic_hip_set_scheme(h, scheme) == 0
        if rpt:
            assert E.epic_hip_set_rows_per_task(h, rpt) == 0
        assert E.epic_hip_update_n_gpu(h, k, 1) in (0, 1)
        assert E.harmonic_get_potential_values_gpu(h) == 0
        gpu_fini(h)
        p = O.Problem(m, u0, locked)
        if scheme == 0:
            assert lib.oracle_jacobi_run(ct.byref(p.h), k) == 0
        else:
            for i in range(k):
                (lib.oracle_update_and_check if i == k - 1 else lib.oracle_update)(ct.byref(p.h))
        assert np.array_equal(h.u_array().ravel(), p.u), f"case {case}: {m}, {k} iterations, scheme {scheme}, rpt {rpt}"
        assert float(h.delta) == float(p.h.delta), f"case {case}: delta"


@pytest.mark.parametrize("scheme", [0, 1])
def test_activity_tracking_3d_is_invisible_and_skips(scheme):
    """3-D work lists: tiles are (plane, 32-row chunk, 256-column strip), woken across the six faces.  Same bits with
    tracking on and off, against the checker, across a re-upload; and the lists really are short while the front is near
    the goal."""
    m = [36, 70, 300]
    u0, locked = synthetic_grid(m, 41, 0.05)
    lib = O.oracle()
    fields = {}
    for track in (1, 0):
        h = make(m, u0, locked)
        gpu_init(h)
        assert E.epic_hip_set_scheme(h, scheme) == 0 and E.epic_hip_set_activity_tracking(h, track) == 0
        assert E.epic_hip_update_n_gpu(h, 9, 1) in (0, 1)
        if track:
            act, tiles = _activity(h)
            assert tiles == 36 * 3 * 2 and 0 < act < tiles, (act, tiles)
        assert E.epic_hip_update_n_gpu(h, 60, 1) in (0, 1)
        first = float(h.delta)
        h.u_array().ravel()[:] = u0           # start over from the host copy: the lists must not survive it
        h.currentIteration = 0
        assert E.harmonic_update_model_gpu(h) == 0
        assert E.epic_hip_update_n_gpu(h, 45, 1) in (0, 1)
        assert E.harmonic_get_potential_values_gpu(h) == 0
        fields[track] = (h.u_array().ravel().copy(), first, float(h.delta))
        gpu_fini(h)
    assert np.array_equal(fields[1][0], fields[0][0]) and fields[1][1:] == fields[0][1:]
    p = O.Problem(m, u0, locked)
    if scheme == 0:
        assert lib.oracle_jacobi_run(ct.byref(p.h), 45) == 0
    else:
        for i in range(45):
            (lib.oracle_update_and_check if i == 44 else lib.oracle_update)(ct.byref(p.h))
    assert np.array_equal(fields[1][0], p.u) and fields[1][2] == float(p.h.delta)
