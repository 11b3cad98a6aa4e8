"""Several devices behind the unchanged C-ABI (EPIC_HIP_DEVICES, include/epic_hip.h): one 2-D grid
cut into row slabs, one per listed device, inside libepic.so -- so that the ROS plugin's harmonic_complete_gpu(&h, 1024)
(/root/reference/src/epic_nav_core_plugin.cpp:256) uses a whole node without a line changed.

A 1-GPU box cannot give every slab its own device, but the list may name a device more than once: "0,0,0,0" runs four
slabs, with their ghost rows, second streams, events, halo copies and per-slab delta words, on the one GPU.  The bar is
the single-device one: the tests of tests/test_gpu_parity.py / test_gpu_tol.py are run again, unchanged, under that
environment -- bit-identical to the checker and to the reference's goldens for every slab count and every halo depth.
BASELINE configs[3] (32768^2, 4 and 8 slabs) runs at full size: seam rows and the window around the goal against the
single-device result and the checker."""
import ctypes as ct
import os

import numpy as np
import pytest

import _oracle as O
import test_gpu_parity as P
import test_gpu_tol as T
from epic_amd import epic_harmonic as eh
from epic_amd.harmonic import Harmonic

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(900)]

E = eh._epic
NT = 1024
UP = ct.POINTER(ct.c_uint)


ALL_LISTS = [("0,0", "8"), ("0,0,0,0", "3"), ("0,0,0", "1"), ("0,0,0,0,0,0,0,0", "8")]
ALL_IDS = ["2slabs_halo8", "4slabs_halo3", "3slabs_halo1", "8slabs_halo8"]
# On a node with several GPUs the same tests also run with one slab per REAL device (hipMemcpyPeerAsync between devices,
# per-device streams and events): every device once, and every device twice interleaved ("0,1,0,1"...).
# (automatic wherever the library sees two or more devices; EPIC_TEST_MULTI_GPU=0 switches it off)
_NDEV = E.epic_hip_device_count() if os.environ.get("EPIC_TEST_MULTI_GPU", "1") != "0" else 0
if _NDEV >= 2:
    _real = ",".join(str(d) for d in range(min(_NDEV, 8)))
    ALL_LISTS += [(_real, "8"), (_real + "," + _real, "3")]
    ALL_IDS += ["%ddevices_halo8" % min(_NDEV, 8), "%ddevices_twice_halo3" % min(_NDEV, 8)]


@pytest.fixture(params=ALL_LISTS[1:4:2] + ALL_LISTS[4:], ids=ALL_IDS[1:4:2] + ALL_IDS[4:])
def devices(request):
    """Most tests run with 4 slabs / 3 ghost rows and 8 slabs / 8 ghost rows; test_fixed_sweeps_equal_the_checker takes
    every list (2 and 3 slabs, the one-row halo of the north star's wording included)."""
    os.environ["EPIC_HIP_DEVICES"], os.environ["EPIC_HIP_HALO"] = request.param
    yield request.param[0].count(",") + 1
    del os.environ["EPIC_HIP_DEVICES"], os.environ["EPIC_HIP_HALO"]


@pytest.fixture(params=ALL_LISTS, ids=ALL_IDS)
def every_list(request):
    os.environ["EPIC_HIP_DEVICES"], os.environ["EPIC_HIP_HALO"] = request.param
    yield request.param[0].count(",") + 1
    del os.environ["EPIC_HIP_DEVICES"], os.environ["EPIC_HIP_HALO"]


def slabs_of(h):
    dev = (ct.c_int * 64)()
    lo, hi, g = (ct.c_uint * 64)(), (ct.c_uint * 64)(), (ct.c_uint * 64)()
    n = E.epic_hip_device_layout(h, 64, dev, lo, hi, g)
    return n, [(dev[i], lo[i], hi[i], g[i]) for i in range(min(n, 64))]


def test_layout_follows_the_device_list(devices):
    m = [403, 300]
    u0, locked = P.synthetic_grid(m, 5, 0.05)
    h = P.make(m, u0, locked)
    P.gpu_init(h)
    n, slabs = slabs_of(h)
    assert n == devices and slabs[0][1] == 0 and slabs[-1][2] == m[0]
    assert all(a[2] == b[1] for a, b in zip(slabs, slabs[1:])) and all(s[0] == 0 for s in slabs)
    assert all(0 <= (a[2] - a[1]) - (b[2] - b[1]) <= 1 for a, b in zip(slabs, slabs[1:]))   # near-equal, larger first
    P.gpu_fini(h)
    # too few rows (planes) for the list: one device
    for m2 in ([7, 300], [6, 10, 70]):
        u0, locked = P.synthetic_grid(m2, 5, 0.05)
        h = P.make(m2, u0, locked)
        P.gpu_init(h)
        assert slabs_of(h)[0] == 1
        P.gpu_fini(h)
    # a 3-D grid is cut along x0, planes for rows
    m3 = [40, 10, 70]
    u0, locked = P.synthetic_grid(m3, 5, 0.05)
    h = P.make(m3, u0, locked)
    P.gpu_init(h)
    n3, slabs3 = slabs_of(h)
    assert n3 == devices and slabs3[0][1] == 0 and slabs3[-1][2] == m3[0]
    P.gpu_fini(h)


@pytest.mark.parametrize("m,seed,dens", [g for g in P.GRIDS_2D if g[0][0] >= 64])
def test_fixed_sweeps_equal_the_checker(every_list, m, seed, dens):
    P.test_fixed_sweeps_2d_vs_oracle_jacobi(m, seed, dens)


@pytest.mark.parametrize("m,rpt", [([66000, 300], 0), ([1200, 9000], 64), ([200, 700], 16)])
def test_extreme_aspect_ratios(devices, m, rpt):
    P.test_extreme_aspect_ratios(m, rpt)


def test_navigation_node_flow_with_live_edits(devices):
    """update(k) batches, set_cells edits (some on and next to the slab seams), mid-solve readback, update_model."""
    P.test_navigation_node_flow_set_cells_and_mid_solve_readback()
    m = [120, 300]
    u0, locked = P.synthetic_grid(m, 9, 0.05)
    h = P.make(m, u0, locked)
    P.gpu_init(h)
    n, slabs = slabs_of(h)
    p = O.Problem(m, u0, locked)
    lib = O.oracle()
    assert E.epic_hip_update_n_gpu(h, 11, 0) == 0
    O.run_session(p, 11)
    edits, types = [], []
    for (_, lo, hi, _g) in slabs[1:]:        # a goal on the last row of the slab above, an obstacle on the first row of this one
        edits += [(37, lo - 1), (150, lo), (151, lo + 1)]
        types += [0, 1, 2]
    v = np.array(edits, dtype=np.uint32)
    t = np.array(types, dtype=np.uint32)
    args = (len(t), v.ctypes.data_as(UP), t.ctypes.data_as(UP))
    assert E.harmonic_utilities_set_cells_2d_gpu(h, NT, *args) == 0
    assert lib.oracle_set_cells_2d(ct.byref(p.h), *args) == 0
    assert E.epic_hip_update_n_gpu(h, 23, 1) in (0, 1)
    O.run_session(p, 23)
    assert E.harmonic_get_potential_values_gpu(h) == 0
    P.gpu_fini(h)
    assert np.array_equal(h.u_array().ravel(), p.u) and h.delta == p.h.delta


def test_lifecycle_and_validation(devices, capfd):
    P.test_execute_gpu_validation_and_lifecycle(capfd)


@pytest.mark.parametrize("name", ["g2d_32", "g2d_64", "g2d_70x66_dense"])
def test_complete_gpu_goldens(devices, goldens, name):
    P.test_complete_gpu_vs_reference_golden(goldens, name)


@pytest.mark.parametrize("name", ["basic", "umass"])
def test_maps_converged(devices, goldens, name, record_property):
    P.test_maps_converged_vs_reference_golden(goldens, name, record_property)


@pytest.mark.parametrize("name", ["g2d_64", "g2d_70x66_dense"])
def test_redblack_is_the_reference_iteration(devices, goldens, name):
    P.test_redblack_half_sweeps_equal_reference_golden(goldens, name)


def test_redblack_map_is_the_reference_result(devices, goldens):
    from conftest import scheme_env

    with scheme_env("redblack"):
        P.test_redblack_maps_are_the_reference_result(goldens, "basic", None)


@pytest.mark.parametrize("m,seed,dens,rpt", [g for g in T.GRIDS if g[0] in ([211, 530], [257, 513])])
def test_tol_iterations_equal_the_checker(devices, m, seed, dens, rpt):
    T.test_tol_iterations_equal_the_checker_bit_for_bit(m, seed, dens, rpt, eh.SCHEME_JACOBI)


@pytest.mark.parametrize("m,seed,dens", [g for g in T.FUSED_GRIDS if g[0] in ([211, 530], [257, 513], [1200, 3000])])
@pytest.mark.parametrize("rows", [0, 7])
def test_tol_fused_pairs_on_slabs_equal_the_checker(every_list, m, seed, dens, rows, monkeypatch):
    """Pairs of plain iterations as one fused pass per slab wherever neither iteration ends with an exchange (a pass leaves
    two more ghost rows stale): every device list, ghost depths 1 (never a pair), 3, 8."""
    T.test_tol_fused_double_sweeps_equal_the_checker_bit_for_bit(m, seed, dens, rows, monkeypatch)


def test_full_size_8192_window_property(devices):
    P.test_full_size_8192_window_property()


@pytest.mark.parametrize("devlist", ["0,0,0,0", "0,0,0,0,0,0,0,0"], ids=["4slabs", "8slabs"])
def test_config4_32768_squared_on_slabs(devlist):
    """BASELINE configs[3]: 32768 x 32768, 4- and 8-way slab decomposition, at full size.  K = 2 halo + 3 sweeps (two
    exchanges and a check in between): the rows on both sides of every seam and the window around the goal must equal the
    single-device run of the same library bit for bit, the window also the checker's run on the window alone; nothing
    outside the goal's reach may have moved."""
    n, halo = 32768, 8
    K, W = 2 * halo + 3, 64
    m = [n, n]
    u0, locked = O.oracle_synthetic(m)
    # a second goal right on the first seam, so that values cross it: the centre goal is far from every seam but one
    nslab = devlist.count(",") + 1
    seam = n // nslab
    c = n // 2
    idx = (seam - 1) * n + 1000
    u0[idx] = 0.0
    locked[idx] = 1

    def run(env):
        if env:
            os.environ["EPIC_HIP_DEVICES"], os.environ["EPIC_HIP_HALO"] = env, str(halo)
        try:
            h = P.make(m, u0, locked)
            P.gpu_init(h)
            slabs = slabs_of(h)
            assert E.epic_hip_set_activity_tracking(h, 0) == 0
            assert E.epic_hip_update_n_gpu(h, halo + 1, 1) in (0, 1)      # a check sweep on the sweep after an exchange
            d1 = float(h.delta)
            assert E.epic_hip_update_n_gpu(h, K - halo - 1, 1) in (0, 1)
            assert E.harmonic_get_potential_values_gpu(h) == 0
            P.gpu_fini(h)
            return h.u_array().reshape(n, n), d1, float(h.delta), slabs
        finally:
            if env:
                del os.environ["EPIC_HIP_DEVICES"], os.environ["EPIC_HIP_HALO"]

    got, d1, d2, (ns, slabs) = run(devlist)
    assert ns == nslab and [s[1] for s in slabs] == [k * seam for k in range(nslab)]
    bands = [slice(max(0, s[1] - 2 * K), s[1] + 2 * K) for s in slabs[1:]] + [slice(c - W, c + W)]
    kept = [got[b].copy() for b in bands]
    moved = int((got != u0.reshape(n, n)).sum())
    del got
    ref, r1, r2, _ = run(None)
    assert (d1, d2) == (r1, r2)
    for b, k in zip(bands, kept):
        assert np.array_equal(ref[b], k), "rows %s differ from the single-device result" % (b,)
    assert moved == int((ref != u0.reshape(n, n)).sum()) and 0 < moved <= 2 * (2 * K + 1) ** 2
    win = (slice(c - W, c + W), slice(c - W, c + W))
    p = O.Problem([2 * W, 2 * W], u0.reshape(n, n)[win].copy(), locked.reshape(n, n)[win].copy())
    O.run_session(p, K)
    assert np.array_equal(ref[win].ravel(), p.u)


# ---- round 3: work lists per slab, planes of a 3-D grid as slabs, the staged transport, the issuing threads -----------------

@pytest.fixture
def tracking_on():
    os.environ["EPIC_HIP_TRACK"] = "1"
    yield
    del os.environ["EPIC_HIP_TRACK"]


@pytest.mark.parametrize("m,seed,dens", [g for g in P.GRIDS_2D if g[0][0] >= 64])
def test_tracked_slabs_equal_the_checker(every_list, tracking_on, m, seed, dens):
    """Activity tracking per slab: a slab's sweep is one list-driven launch over its own tiles, the tiles around the ghost
    rows are woken after every exchange.  Same bits as the checker's Jacobi, every list, ghost depths 1, 3, 8."""
    P.test_fixed_sweeps_2d_vs_oracle_jacobi(m, seed, dens)


@pytest.mark.parametrize("name", ["g2d_64", "g2d_70x66_dense"])
def test_tracked_slabs_redblack_is_the_reference_iteration(devices, tracking_on, goldens, name):
    P.test_redblack_half_sweeps_equal_reference_golden(goldens, name)


def test_tracked_slabs_converge_to_the_reference_map(devices, tracking_on, goldens):
    from conftest import scheme_env

    with scheme_env("redblack"):     # the library default: iteration count and field are the reference's, bit for bit
        P.test_redblack_maps_are_the_reference_result(goldens, "basic", None)


@pytest.mark.parametrize("m,seed,dens,rpt", [g for g in T.GRIDS if g[0] in ([211, 530], [257, 513])])
def test_tracked_slabs_tol_equal_the_checker(devices, tracking_on, m, seed, dens, rpt):
    T.test_tol_iterations_equal_the_checker_bit_for_bit(m, seed, dens, rpt, eh.SCHEME_JACOBI)


def test_tracked_slabs_live_edits(devices, tracking_on):
    test_navigation_node_flow_with_live_edits(devices)


@pytest.mark.parametrize("m,seed,dens", [g for g in P.GRIDS_3D if g[0][0] >= 8] + [([40, 12, 70], 21, 0.05), ([33, 40, 300], 22, 0.05)])
@pytest.mark.parametrize("track", [False, True])
def test_3d_plane_slabs_equal_the_checker(every_list, m, seed, dens, track):
    """A 3-D grid cut into slabs of planes (grids with fewer than four planes per listed device stay on one device)."""
    if track:
        os.environ["EPIC_HIP_TRACK"] = "1"
    try:
        P.test_fixed_sweeps_3d_vs_oracle_jacobi(m, seed, dens)
    finally:
        os.environ.pop("EPIC_HIP_TRACK", None)


@pytest.mark.parametrize("name", ["g3d_16", "g3d_20x12x34"])
def test_3d_plane_slabs_goldens(devices, goldens, name):
    P.test_complete_gpu_vs_reference_golden(goldens, name)
    P.test_redblack_half_sweeps_equal_reference_golden(goldens, name)


@pytest.mark.parametrize("m,seed,dens,rpt", [([20, 12, 34], 14, 0.05, 0), ([40, 12, 70], 21, 0.05, 0)])
def test_3d_plane_slabs_tol(devices, m, seed, dens, rpt):
    for scheme in (eh.SCHEME_JACOBI, eh.SCHEME_REDBLACK):
        T.test_tol_iterations_equal_the_checker_bit_for_bit(m, seed, dens, rpt, scheme)


@pytest.mark.parametrize("knob", ["EPIC_HIP_NO_PEER", "EPIC_HIP_THREADS"])
def test_staged_transport_and_single_issuing_thread(every_list, knob, monkeypatch):
    """EPIC_HIP_NO_PEER=1: the halo units go through pinned host memory (the path a node without peer access takes);
    EPIC_HIP_THREADS=0: the caller's thread issues everything instead of one thread per slab.  Same bits either way."""
    monkeypatch.setenv(knob, "1" if knob == "EPIC_HIP_NO_PEER" else "0")
    P.test_fixed_sweeps_2d_vs_oracle_jacobi([257, 513], 9, 0.05)
    P.test_fixed_sweeps_3d_vs_oracle_jacobi([20, 12, 34], 14, 0.05)
    monkeypatch.setenv("EPIC_HIP_TRACK", "1")
    P.test_fixed_sweeps_2d_vs_oracle_jacobi([130, 256], 11, 0.05)


def test_two_slabs_relax_8192_squared_like_one_device(record_property):
    """BASELINE configs[2] relaxed to eps = 1e-6 by the library default (precise, red-black, work lists) on one device and
    on two slabs of the same device: field, iteration count and final delta bit-identical, and the slabs within 15 % of the
    single-device time.  Since round 6 the slabs take what one device takes: tracked PAIRS of list-driven fused passes
    (driver_multi.hip: multi_run_pairs); until then they ran list-driven half-sweeps (EPIC_HIP_TRACK_PAIRS=0, run beside it: same
    bits).  Measured in one run (profiles/r06_experiments.txt item 3): one device 2.28 s, two slabs 2.50 s, four 2.99 s, two slabs of
    half-sweeps 2.83 s.  What is left grows by ~0.23 s per extra slab = 22 500 passes x ~10 us: every slab's pass is a launch of its
    own, and on ONE GPU a list-driven launch (persistent waves over the whole chip) does not overlap its neighbour's -- the slabs'
    launches queue up behind one another here and run side by side on devices of their own."""
    import time

    from epic_amd.synthetic import synthetic_grid

    m = [8192, 8192]
    u0, locked = synthetic_grid(m)
    out = {}

    def relax(label, env):
        if env:
            os.environ["EPIC_HIP_DEVICES"] = env
        if label == "two_half_sweeps":
            os.environ["EPIC_HIP_TRACK_PAIRS"] = "0"
        try:
            h = P.make(m, u0, locked)
            for fn in (E.harmonic_initialize_dimension_size_gpu, E.harmonic_initialize_potential_values_gpu,
                       E.harmonic_initialize_locked_gpu):
                assert fn(h) == 0
        finally:
            os.environ.pop("EPIC_HIP_DEVICES", None)
            os.environ.pop("EPIC_HIP_TRACK_PAIRS", None)
        assert E.epic_hip_set_math_mode(h, eh.MATH_PRECISE) == 0 and E.epic_hip_set_scheme(h, eh.SCHEME_REDBLACK) == 0
        assert E.epic_hip_set_activity_tracking(h, 2) == 0
        path = eh.config_dump(h)["path"]["plain_batch"]
        assert ("tracked pairs" in path) == (label != "two_half_sweeps"), path
        t0 = time.perf_counter()
        assert E.harmonic_execute_gpu(h, NT) == 0
        dt = time.perf_counter() - t0
        for fn in (E.harmonic_uninitialize_dimension_size_gpu, E.harmonic_uninitialize_potential_values_gpu,
                   E.harmonic_uninitialize_locked_gpu):
            assert fn(h) == 0
        out[label] = (h.u_array().ravel().copy(), int(h.currentIteration), float(h.delta), dt)
        print("%s: %d iterations, delta %.3e, %.3f s" % (label, out[label][1], out[label][2], dt))
        return dt

    for label, env in (("one", None), ("two", "0,0"), ("one_again", None), ("two_half_sweeps", "0,0")):
        relax(label, env)
    assert out["one"][1:3] == out["two"][1:3] == out["two_half_sweeps"][1:3]
    assert np.array_equal(out["one"][0], out["two"][0]) and np.array_equal(out["one"][0], out["two_half_sweeps"][0])
    one = min(out["one"][3], out["one_again"][3])
    two, half = out["two"][3], out["two_half_sweeps"][3]
    if two > 1.15 * one or two > 0.97 * half:
        # a 2.5 s wall-clock figure of a shared box: one hiccup of the host is 15 % -- the faster of two runs before anything fails
        # (the slabs' run is the one a hiccup can only hurt)
        first = out["two"][0]
        two = min(two, relax("two", "0,0"))
        assert np.array_equal(out["two"][0], first)
    record_property("seconds_one_device", one)
    record_property("seconds_two_slabs", two)
    record_property("seconds_two_slabs_half_sweeps", half)
    assert two <= 1.15 * one, (two, one)
    assert two <= 0.97 * half, (two, half)   # what the pairs buy on slabs


# ---- round 6: tracked PAIRS on the slabs (driver_multi.hip: multi_run_pairs) ------------------------------------------------------
import test_gpu_tracked_pairs as TP  # noqa: E402


@pytest.fixture(params=[("0,0", "8"), ("0,0,0,0", "3"), ("0,0,0", "2"), ("0,0,0,0,0", "5")], ids=["2slabs_halo8", "4slabs_halo3", "3slabs_halo2", "5slabs_halo5"])
def pair_lists(request):
    """Even and odd ghost depths (an odd depth trades the rows one iteration early), the shallowest one a pass can live with (2)."""
    os.environ["EPIC_HIP_DEVICES"], os.environ["EPIC_HIP_HALO"] = request.param
    yield request.param[0].count(",") + 1
    del os.environ["EPIC_HIP_DEVICES"], os.environ["EPIC_HIP_HALO"]


@pytest.mark.parametrize("rows,switch", [(None, None), (4, "2"), (7, "0"), (33, "2")])
def test_tracked_pairs_on_slabs_benchmark_family_equals_the_reference(pair_lists, rows, switch):
    """The library's defaults (precise, red-black, work lists) on slabs: every slab runs list-driven fused passes over its own rows,
    the check is the second iteration of the last pass (owned rows only), the ghost rows are traded every halo / 2 passes and the tiles
    two rows deep around them woken.  Field, iteration count and delta of harmonic_complete_cpu, any task height, lists always / never
    bypassed / by the rule."""
    TP.test_benchmark_family_through_tracked_pairs_equals_the_reference(512, switch, rows)


@pytest.mark.parametrize("name,eps", [("maze", "1e-06"), ("umass", "0.001"), ("basic", "0.001")])
def test_tracked_pairs_on_slabs_maps_equal_the_reference(pair_lists, name, eps, goldens):
    TP.test_maps_through_tracked_pairs_equal_the_reference(name, eps, goldens)


@pytest.mark.parametrize("stagger", [7, 10, 2, 1, 33])
def test_tracked_pairs_on_slabs_any_check_interval(pair_lists, goldens, stagger):
    TP.test_any_check_interval_pairs_or_not(goldens, "g2d_70x66_dense", stagger)


@pytest.mark.parametrize("scheme", ["redblack", "jacobi"])
def test_tracked_tol_pairs_on_slabs_equal_the_half_sweep_path(pair_lists, scheme):
    TP.test_tol_relaxations_through_tracked_pairs_equal_the_half_sweep_path(512, scheme, 7, None)


@pytest.mark.parametrize("scheme", ["redblack", "jacobi"])
def test_tracked_tol_pairs_on_slabs_equal_the_checkers_loop(pair_lists, goldens, scheme):
    TP.test_tol_tracked_pairs_equal_the_checkers_loop(goldens, "g2d_70x66_dense", scheme)


def test_tracked_pairs_on_slabs_through_the_fine_grained_api(pair_lists):
    """The navigation node's ticks on slabs with work lists: deferred blocks of halo iterations as pairs, the check the second iteration
    of the last pair, edits and read-backs in between (tests/test_gpu_node_flow.py's script fuzz draws slabs too; this is the fixed case)."""
    import test_gpu_node_flow as NF

    saved = {k: os.environ.get(k) for k in ("EPIC_HIP_TRACK", "EPIC_HIP_FUSE_MIN_CELLS", "EPIC_HIP_TILE")}
    os.environ.update(EPIC_HIP_TRACK="1", EPIC_HIP_FUSE_MIN_CELLS="0", EPIC_HIP_TILE="0")
    try:
        m = [310, 940]
        u0, locked = P.synthetic_grid(m, 5, 0.10)
        h = NF.make(m, u0, locked)
        NF.gpu_init(h)
        assert E.epic_hip_set_scheme(h, eh.SCHEME_REDBLACK) == 0     # (the library default, whatever the session's scheme)
        assert "tracked pairs" in eh.config_dump(h)["path"]["plain_batch"]
        NF.ticks(h, 4, 23)
        assert E.harmonic_get_potential_values_gpu(h) == 0
        p = O.Problem(m, u0, locked)
        lib = O.oracle()
        for _ in range(4):
            lib.oracle_update_and_check(ct.byref(p.h))
            d = float(p.h.delta)
            for _ in range(22):
                lib.oracle_update(ct.byref(p.h))
        assert np.array_equal(h.u_array().ravel(), p.u) and float(h.delta) == d
        NF.gpu_fini(h)
    finally:
        for k, v in saved.items():
            os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)


@pytest.mark.parametrize("m,rpt", [([2050, 2100], 16), ([4200, 1000], 64)])
def test_redblack_fused_pairs_on_slabs(every_list, m, rpt):
    """Untracked red-black stretches between two exchanges run as fused pairs per slab (rb_fused2d_kernel), as on one device."""
    P.test_redblack_fused_pairs_equal_the_checker(m, rpt)


@pytest.mark.parametrize("m,seed,dens", [g for g in T.FUSED_GRIDS if g[0] in ([257, 513], [1200, 3000])])
def test_tol_fused_redblack_pairs_on_slabs(every_list, m, seed, dens, monkeypatch):
    T.test_tol_fused_redblack_pairs_equal_the_checker_bit_for_bit(m, seed, dens, 6, monkeypatch)


def test_multi_report_describes_the_seams_and_times_one_exchange(devices):
    """epic_hip_multi_report (include/epic_hip.h): the mode reporting on itself -- seams, transport, one exchange iteration under
    timing events -- and leaving the relaxation bit-identical to one that was never probed (it advances by whole iterations)."""
    import json

    m = [2048, 1100]
    u0, locked = P.synthetic_grid(m, 9, 0.05)
    h = P.make(m, u0, locked)
    P.gpu_init(h)
    assert E.epic_hip_set_math_mode(h, eh.MATH_TOL) == 0
    assert E.epic_hip_update_n_gpu(h, 37, 0) == 0
    buf = ct.create_string_buffer(1 << 16)
    n = E.epic_hip_multi_report(h, buf, len(buf))
    assert n > 0
    rep = json.loads(buf.value.decode())
    assert rep["slabs"] == devices and len(rep["seams"]) == devices - 1 and len(rep["exchange"]) == devices
    for seam in rep["seams"]:
        assert seam["transport"] in ("same-device", "peer", "staged")
        if seam["upper_device"] == seam["lower_device"]:
            assert seam["transport"] == "same-device"
    for ex in rep["exchange"]:
        assert ex["interior_us"][1] > ex["interior_us"][0] >= 0.0 and ex["bands_and_copies_us"][1] > ex["bands_and_copies_us"][0] >= 0.0
        assert min(ex["interior_us"][0], ex["bands_and_copies_us"][0]) == 0.0 and ex["overlap_us"] >= 0.0
        assert isinstance(ex["copies_hidden"], bool)
    done = int(h.currentIteration)
    assert 37 < done <= 37 + rep["halo"]
    assert E.epic_hip_update_n_gpu(h, 100 - done, 1) in (0, 1)
    assert E.harmonic_get_potential_values_gpu(h) == 0
    got, gdelta = h.u_array().ravel().copy(), float(h.delta)
    P.gpu_fini(h)
    p = O.Problem(m, u0, locked)
    assert O.oracle().oracle_tol_run(ct.byref(p.h), 100, 0) == 0
    assert np.array_equal(got, p.u) and gdelta == float(p.h.delta)
    # one device: nothing to report
    del os.environ["EPIC_HIP_DEVICES"]
    try:
        h = P.make(m, u0, locked)
        P.gpu_init(h)
        assert E.epic_hip_multi_report(h, buf, len(buf)) == 0
        P.gpu_fini(h)
    finally:
        os.environ["EPIC_HIP_DEVICES"] = "0,0"
