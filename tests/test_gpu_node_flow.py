"""The fine-grained API as the navigation node drives it (src/epic_navigation_node_harmonic.cpp:165-189): one
harmonic_update_and_check_gpu and steps_per_update - 1 single harmonic_update_gpu calls per tick.

Since round 6 a plain update only COUNTS and the library enqueues the counted iterations as whole blocks -- LDS tiles with several
iterations per launch, fused pairs, a captured graph -- at the size the kernel family in use is built for and at every ordering
point of the boundary (epic_amd/csrc/driver_loop.hip: "harmonic_update_gpu counts").  What must hold:
  * every read-back shows the field of every iteration asked for so far, bit for bit the checker's (and EPIC_HIP_DEFER=0's);
  * a check's delta is that iteration's, and it follows the pending iterations in order;
  * edits, mode changes, update_model and a caller-set currentIteration are ordering points;
  * tearing the state down with iterations pending neither leaks nor runs them on freed memory;
  * the literal C++ loop of the node (tests/plugin_replay/node_flow.cpp) ends in harmonic_execute_gpu's field after the same count.
"""
import ctypes as ct
import os

import numpy as np
import pytest

import _oracle as O
import fuzz_gpu_parity as F
from epic_amd import epic_harmonic as eh
from epic_amd.harmonic import Harmonic
from epic_amd.synthetic import synthetic_grid

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(600)]

E = eh._epic
NT = 1024
ROOT = O.ROOT


def make(m, u, locked, eps=1e-6, stagger=100):
    h = Harmonic()
    h.set_grid(m, u, locked)
    h.epsilon = eps
    h.numIterationsToStaggerCheck = stagger
    return h


def gpu_init(h):
    for fn in (E.harmonic_initialize_dimension_size_gpu, E.harmonic_initialize_potential_values_gpu, E.harmonic_initialize_locked_gpu):
        assert fn(h) == 0
    assert E.harmonic_initialize_gpu(h, NT) == 0


def gpu_fini(h):
    for fn in (E.harmonic_uninitialize_gpu, E.harmonic_uninitialize_dimension_size_gpu, E.harmonic_uninitialize_potential_values_gpu,
               E.harmonic_uninitialize_locked_gpu):
        assert fn(h) == 0


def ticks(h, n_ticks, steps):
    for _ in range(n_ticks):
        rc = E.harmonic_update_and_check_gpu(h, NT)
        assert rc in (0, 1)
        for _ in range(steps - 1):
            assert E.harmonic_update_gpu(h, NT) == 0


@pytest.fixture
def env():
    saved = {}

    def set_(**kw):
        for k, v in kw.items():
            saved.setdefault(k, os.environ.get(k))
            os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, str(v))
        assert E.epic_hip_config_reload(None) == 0

    yield set_
    for k, v in saved.items():
        os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
    E.epic_hip_config_reload(None)


GRIDS = [([48, 300], 3, 0.05), ([310, 940], 5, 0.10), ([200, 130], 7, 0.02), ([12, 20, 70], 9, 0.05)]


@pytest.mark.parametrize("m,seed,dens", GRIDS)
@pytest.mark.parametrize("steps", [1, 2, 7, 50])
def test_ticks_equal_the_checker_and_the_undeferred_library(m, seed, dens, steps, env):
    """n ticks of (check + steps - 1 plain updates), the session's scheme: the field at a read-back in the middle of a tick (iterations
    pending) and after the last tick is the checker's and the one-launch-per-call library's, bit for bit; so is the last check's delta."""
    u0, locked = synthetic_grid(m, seed, dens)
    n_ticks = 3 if steps >= 7 else 9
    fields = {}
    for defer in (None, "0"):
        env(EPIC_HIP_DEFER=defer)
        h = make(m, u0, locked)
        gpu_init(h)
        ticks(h, n_ticks, steps)
        assert h.currentIteration == n_ticks * steps
        d_last_tick = float(h.delta)
        for _ in range(3):   # a tick cut short: three plain updates pending at the read-back
            assert E.harmonic_update_gpu(h, NT) == 0
        assert E.harmonic_get_potential_values_gpu(h) == 0
        mid = h.u_array().ravel().copy()
        assert E.harmonic_update_and_check_gpu(h, NT) in (0, 1)
        assert E.harmonic_get_potential_values_gpu(h) == 0
        fields[defer] = (mid, h.u_array().ravel().copy(), d_last_tick, float(h.delta), int(h.currentIteration))
        gpu_fini(h)
    assert fields[None][4] == fields["0"][4] == n_ticks * steps + 4
    for a, b in zip(fields[None][:4], fields["0"][:4]):
        assert np.array_equal(a, b)
    # the checker: the same iterations, the checks where the ticks put them
    p = O.Problem(m, u0, locked)
    for _ in range(n_ticks):
        assert O.run_session(p, 1) == 0          # the tick's check comes FIRST
        d = float(p.h.delta)
        if steps > 1:
            assert O.run_session(p, steps - 1) == 0
    assert d == fields[None][2]
    assert O.run_session(p, 3) == 0
    assert np.array_equal(fields[None][0], p.u)
    assert O.run_session(p, 1) == 0
    assert np.array_equal(fields[None][1], p.u) and float(p.h.delta) == fields[None][3]


def test_every_ordering_point_sees_the_pending_iterations(env):
    """set_cells, a mode change, update_n, update_model, a caller-set currentIteration and the path walk between single updates: each
    acts on (or discards, for update_model) exactly the iterations asked for before it."""
    m = [64, 300]
    u0, locked = synthetic_grid(m, 11, 0.05)
    UP = ct.POINTER(ct.c_uint)
    v = np.array([[250, 40], [10, 5], [7, 60]], dtype=np.uint32)
    t = np.array([0, 1, 2], dtype=np.uint32)
    out = {}
    for defer in (None, "0"):
        env(EPIC_HIP_DEFER=defer)
        h = make(m, u0, locked)
        gpu_init(h)
        assert E.epic_hip_set_scheme(h, eh.SCHEME_REDBLACK) == 0
        shots = []

        def shot():
            assert E.harmonic_get_potential_values_gpu(h) == 0
            shots.append(h.u_array().ravel().copy())

        for _ in range(5):
            assert E.harmonic_update_gpu(h, NT) == 0
        assert E.harmonic_utilities_set_cells_2d_gpu(h, NT, len(t), v.ctypes.data_as(UP), t.ctypes.data_as(UP)) == 0   # after 5 iterations
        for _ in range(4):
            assert E.harmonic_update_gpu(h, NT) == 0
        shot()
        for _ in range(3):
            assert E.harmonic_update_gpu(h, NT) == 0
        assert E.epic_hip_set_math_mode(h, eh.MATH_TOL) == 0        # the 3 pending run with the precise math, what follows with tol
        for _ in range(3):
            assert E.harmonic_update_gpu(h, NT) == 0
        assert E.epic_hip_set_math_mode(h, eh.MATH_PRECISE) == 0
        shot()
        for _ in range(2):
            assert E.harmonic_update_gpu(h, NT) == 0
        assert E.epic_hip_update_n_gpu(h, 6, 1) in (0, 1)           # behind the 2 pending
        shots.append(np.float32(h.delta))
        assert E.harmonic_update_gpu(h, NT) == 0
        h.currentIteration = 40                                     # the caller renumbers: the pending one keeps ITS colour
        for _ in range(3):
            assert E.harmonic_update_gpu(h, NT) == 0
        assert E.harmonic_update_and_check_gpu(h, NT) in (0, 1)
        assert h.currentIteration == 44
        shots.append(np.float32(h.delta))
        shot()
        k = ct.c_uint(0)
        for _ in range(2):
            assert E.harmonic_update_gpu(h, NT) == 0
        prc = ct.c_int(0)
        pts = np.zeros(2 * 4000, dtype=np.float32)
        start = np.array([20.0, 30.0], dtype=np.float32)
        assert E.epic_hip_compute_paths_2d_gpu(h, 1, start.ctypes.data_as(ct.POINTER(ct.c_float)), 0.5, 0.5, 4000, ct.byref(k), ct.byref(prc),
                                               pts.ctypes.data_as(ct.POINTER(ct.c_float))) == 0   # walks the field of 46 iterations
        shots.append(pts[: 2 * k.value].copy())
        for _ in range(6):
            assert E.harmonic_update_gpu(h, NT) == 0
        hu = h.u_array().ravel()
        hu[:] = u0
        assert E.harmonic_update_model_gpu(h) == 0                  # both field and mask replaced: the 6 pending have nothing to show
        shot()
        assert np.array_equal(shots[-1], u0)
        for _ in range(7):
            assert E.harmonic_update_gpu(h, NT) == 0
        gpu_fini(h)                                                 # torn down with iterations pending
        out[defer] = shots
    assert len(out[None]) == len(out["0"])
    for a, b in zip(out[None], out["0"]):
        assert np.array_equal(a, b)


def test_teardown_in_every_order_with_iterations_pending(env):
    """Whatever order the caller frees the device state in, iterations still pending are run while the state is whole or dropped with
    it; the field read back after uninitialize_locked (legal in the reference) is the field of every iteration asked for."""
    import itertools

    m = [40, 200]
    u0, locked = synthetic_grid(m, 4, 0.05)
    p = O.Problem(m, u0, locked)
    assert O.run_session(p, 1) == 0
    assert O.run_session(p, 4) == 0
    want = p.u.copy()
    calls = (E.harmonic_uninitialize_gpu, E.harmonic_uninitialize_dimension_size_gpu, E.harmonic_uninitialize_locked_gpu)
    for order in itertools.permutations(calls):
        h = make(m, u0, locked)
        gpu_init(h)
        assert E.harmonic_update_and_check_gpu(h, NT) in (0, 1)
        for _ in range(3):
            assert E.harmonic_update_gpu(h, NT) == 0
        assert E.harmonic_update_and_check_gpu(h, NT) in (0, 1)    # (run_session(p, 4) ends with a check)
        p2 = O.Problem(m, want, locked)
        p2.h.currentIteration = 5
        for _ in range(3):
            assert E.harmonic_update_gpu(h, NT) == 0
        assert O.run_session(p2, 3) == 0
        for fn in order:
            assert fn(h) == 0
        assert E.harmonic_get_potential_values_gpu(h) == 0          # d_u is still there
        assert np.array_equal(h.u_array().ravel(), p2.u)
        assert E.harmonic_uninitialize_potential_values_gpu(h) == 0
        assert not h.d_m and not h.d_u and not h.d_locked and not h.d_delta


@pytest.mark.parametrize("m,pairs", [([337, 299], "0"), ([337, 299], None), ([16, 24, 121], None)])
@pytest.mark.parametrize("math", [eh.MATH_PRECISE, eh.MATH_TOL])
def test_a_caller_that_renumbers_its_iterations_with_work_lists_on(m, pairs, math, env):
    """currentIteration is the caller's field, and the colour of a red-black iteration is its number's parity.  With work lists the lists
    in force were made for the colour that was to come next; a caller that renumbers so that one colour runs twice in a row used to get an
    empty list for the iteration after (nothing had changed in the repeated colour) and a field that stood still -- found by the script
    fuzz of round 6 (seed 61, cases 373, 564, 572, 591), fixed by epic_amd/csrc/driver_enqueue.hip: note_iterations.  Checked against
    the checker here, which takes the parity from the number like the reference (harmonic_cpu.cpp:46-51)."""
    env(EPIC_HIP_TRACK="1", EPIC_HIP_FUSE_MIN_CELLS="0", EPIC_HIP_TILE="0", EPIC_HIP_TRACK_PAIRS=pairs, EPIC_HIP_TRACK_SWITCH="2")
    u0, locked = synthetic_grid(m, 17, 0.05)
    h = make(m, u0, locked)
    gpu_init(h)
    assert E.epic_hip_set_math_mode(h, math) == 0 and E.epic_hip_set_scheme(h, eh.SCHEME_REDBLACK) == 0
    p = O.Problem(m, u0, locked)
    lib = O.oracle()

    def steps(n, check):
        for i in range(n):
            last = check and i == n - 1
            assert (E.harmonic_update_and_check_gpu if last else E.harmonic_update_gpu)(h, NT) in (0, 1)
        if math == eh.MATH_TOL:
            assert lib.oracle_tol_run(ct.byref(p.h), n, 1) == 0
        else:
            for i in range(n):
                (lib.oracle_update_and_check if check and i == n - 1 else lib.oracle_update)(ct.byref(p.h))

    steps(21, True)
    for renumber in (70, 71, 1000, 3):        # same colour again, the other colour, and back
        h.currentIteration = renumber
        p.h.currentIteration = renumber
        steps(2, True)
        assert float(h.delta) == float(p.h.delta), renumber
        steps(5, False)
        assert E.harmonic_get_potential_values_gpu(h) == 0
        assert np.array_equal(h.u_array().ravel(), p.u), renumber
    gpu_fini(h)


def test_update_without_state_is_refused_and_counts_nothing():
    m = [16, 16]
    u0, locked = synthetic_grid(m, 1, 0.0)
    h = make(m, u0, locked)
    assert E.harmonic_update_gpu(h, NT) == eh.EPIC_ERROR_INVALID_DATA
    assert h.currentIteration == 0


@pytest.fixture(scope="module")
def nodeflow():
    so = os.path.join(ROOT, "tests", "plugin_replay", "libnodeflow.so")
    if not os.path.exists(so):
        pytest.skip("tests/plugin_replay/libnodeflow.so is not built (__graft_entry__.build())")
    lib = ct.CDLL(so)
    lib.node_flow_run.restype = ct.c_int
    lib.node_flow_execute.restype = ct.c_int
    return lib


@pytest.mark.parametrize("name,steps", [("maze", 50), ("umass", 100), ("basic", 50)])
def test_the_nodes_cpp_loop_on_the_reference_maps_ends_in_executes_field(nodeflow, name, steps):
    """The literal C++ loop (tests/plugin_replay/node_flow.cpp) with the library's defaults (no environment: the reference's own
    iteration, bit-exact arithmetic) for exactly as many iterations as harmonic_execute_gpu takes at eps = 1e-3: same field, and that
    field is the reference's (sha256 in tests/golden/ref_maps.json is checked by test_gpu_callers_eps.py for execute)."""
    saved = {k: os.environ.pop(k, None) for k in ("EPIC_HIP_SCHEME", "EPIC_HIP_MATH")}
    assert E.epic_hip_config_reload(None) == 0
    try:
        m, u0, locked = O.load_png_reference_rule(os.path.join(ROOT, "tests", "golden", "maps", name + ".png"))
        u0, locked = np.ravel(u0), np.ravel(locked)
        h = make(list(m), u0, locked, eps=1e-3)
        for fn in (E.harmonic_initialize_dimension_size_gpu, E.harmonic_initialize_potential_values_gpu, E.harmonic_initialize_locked_gpu):
            assert fn(h) == 0
        sec = ct.c_double(0.0)
        assert nodeflow.node_flow_execute(ct.byref(h), NT, ct.byref(sec)) == 0
        its, want, d_exec = int(h.currentIteration), h.u_array().ravel().copy(), float(h.delta)
        h.u_array().ravel()[:] = u0
        assert E.harmonic_update_model_gpu(h) == 0
        assert E.harmonic_initialize_gpu(h, NT) == 0
        h.currentIteration = 0
        done, conv = ct.c_uint(0), ct.c_uint(0)
        assert nodeflow.node_flow_run(ct.byref(h), its, steps, NT, 1, ct.byref(sec), ct.byref(done), ct.byref(conv)) == 0
        assert done.value == its and h.currentIteration == its
        assert np.array_equal(h.u_array().ravel(), want)
        gpu_fini(h)
    finally:
        for k, v in saved.items():
            if v is not None:
                os.environ[k] = v
        E.epic_hip_config_reload(None)


@pytest.mark.parametrize("seed", [31])
def test_random_scripts_of_single_calls_edits_and_readbacks(seed):
    assert F.campaign_node(40, seed, verbose=False) == []
