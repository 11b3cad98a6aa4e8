"""EPIC_HIP_JACOBI_CHECKS=reference on the device (opt-in; round 6): every CHECK iteration of a Jacobi run is the reference's red-black half-sweep
of that iteration's colour, in place (epic_amd/csrc/driver_loop.hip: run_block; the checkers' statement and why it works:
tests/test_jacobi_reference_checks.py, oracle/harmonic_oracle.c: oracle_set_jacobi_ref_checks).

  * precise arithmetic: harmonic_complete_gpu under the JACOBI scheme returns harmonic_complete_cpu's own field, delta and iteration count, bit for
    bit -- against the reference-generated goldens, on every kernel family a Jacobi block can take (LDS tiles, fused pairs, single sweeps, work
    lists, row / plane slabs);
  * tol arithmetic: the device's loop is the checker's (tolerance 0), on the campaign's seven cases that the plain Jacobi checks leave outside the bar;
  * the fine-grained entry points (the navigation node's ticks, deferred or not) and the timed batches take the same check iterations."""
import ctypes as ct
import json
import os

import numpy as np
import pytest

import _oracle as O
from epic_amd import epic_harmonic as eh
from epic_amd.harmonic import Harmonic
from epic_amd.harmonic_map import HarmonicMap
from epic_amd.synthetic import synthetic_grid

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(600)]

E = eh._epic
NT = 1024
HERE = os.path.dirname(os.path.abspath(__file__))
SMALL = ["g2d_16", "g2d_32", "g2d_64", "g2d_23x37", "g2d_5x7", "g2d_3x3", "g2d_8x300", "g2d_70x66_dense", "g3d_8", "g3d_16", "g3d_7x9x11", "g3d_20x12x34"]
# what a Jacobi block runs on (study knobs: tests/conftest.py sets EPIC_HIP_STUDY=1)
FAMILIES = {"defaults": {}, "work_lists": {"EPIC_HIP_TRACK": "1"}, "graphs_of_single_sweeps": {"EPIC_HIP_TILE": "0", "EPIC_HIP_NO_FUSE": "1"}, "single_sweeps": {"EPIC_HIP_TILE": "0", "EPIC_HIP_NO_FUSE": "1", "EPIC_HIP_NO_GRAPH": "1"},
            "fused_pairs": {"EPIC_HIP_TILE": "0", "EPIC_HIP_FUSE_MIN_CELLS": "0"}, "fused_pairs_lists": {"EPIC_HIP_TILE": "0", "EPIC_HIP_FUSE_MIN_CELLS": "0", "EPIC_HIP_TRACK": "1"},
            "three_slabs": {"EPIC_HIP_DEVICES": "0,0,0", "EPIC_HIP_HALO": "2"}, "two_slabs_lists": {"EPIC_HIP_DEVICES": "0,0", "EPIC_HIP_HALO": "3", "EPIC_HIP_TRACK": "1"}}


@pytest.fixture(autouse=True)
def jacobi_with_reference_checks(monkeypatch):
    monkeypatch.setenv("EPIC_HIP_SCHEME", "jacobi")
    monkeypatch.setenv("EPIC_HIP_JACOBI_CHECKS", "reference")
    lib = O.oracle()
    lib.oracle_set_jacobi_ref_checks.argtypes = (ct.c_int,)
    lib.oracle_set_jacobi_ref_checks.restype = None
    lib.oracle_set_jacobi_ref_checks(1)
    yield lib
    lib.oracle_set_jacobi_ref_checks(0)


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


@pytest.mark.parametrize("family", list(FAMILIES))
@pytest.mark.parametrize("name", SMALL)
def test_precise_jacobi_with_reference_checks_returns_the_reference_result(goldens, name, family, monkeypatch):
    g, info = goldens["small"], goldens["manifest"]["small"][name]
    m = [int(x) for x in g[name + "/m"]]
    if "slabs" in family and m[0] < 12:
        pytest.skip("too few rows for these slabs")
    for k, v in FAMILIES[family].items():
        monkeypatch.setenv(k, v)
    h = make(m, g[name + "/u0"], g[name + "/locked"], info["epsilon"], info["stagger"])
    assert E.harmonic_complete_gpu(h, NT) == 0
    assert h.currentIteration == info["iterations"] and float(h.delta) == info["delta"]
    assert np.array_equal(h.u_array().ravel(), g[name + "/converged"])     # harmonic_complete_cpu's field


def test_the_dump_names_the_knob(monkeypatch):
    m = [40, 300]
    u0, locked = synthetic_grid(m, 3, 0.05)
    h = make(m, u0, locked)
    gpu_init(h)
    assert eh.config_dump(h)["config"]["jacobi_checks"] == "reference"
    monkeypatch.delenv("EPIC_HIP_JACOBI_CHECKS")
    assert E.epic_hip_config_reload(h) == 0
    assert eh.config_dump(h)["config"]["jacobi_checks"] == "jacobi"
    gpu_fini(h)


@pytest.mark.parametrize("name,eps", [("basic", 1e-6), ("maze", 1e-6), ("umass", 1e-6), ("basic", 1e-3), ("umass", 1e-3)])
@pytest.mark.parametrize("devices", [None, "0,0,0,0"])
def test_the_reference_maps_under_the_jacobi_scheme_are_the_reference_result(goldens, name, eps, devices, monkeypatch):
    """BASELINE configs 1-2 (the LDS tiles' Jacobi steps, ten per launch, and one half-sweep per check): iteration count, delta and field of
    harmonic_complete_cpu (tests/golden/generate_goldens.py) -- what the plain Jacobi checks reach only within the bar."""
    if devices:
        monkeypatch.setenv("EPIC_HIP_DEVICES", devices)
    key = "1e-06" if eps == 1e-6 else "0.001"
    run = goldens["manifest"]["maps"][name]["runs"][key]
    h = HarmonicMap().load(os.path.join(O.ROOT, "tests", "golden", "maps", name + ".png"))
    h.solve(process="gpu", epsilon=eps)
    assert h.currentIteration == run["iterations"] and float(h.delta) == run["delta"]
    idx = goldens["maps"][name + "/sample_idx"]
    assert np.array_equal(h.u_array().ravel()[idx], goldens["maps"][name + "/samples_" + key])
    if eps == 1e-6:
        assert np.array_equal(h.u_array().ravel(), goldens["maps"][name + "/converged_1e-06"])


def campaign_misses():
    rec = json.load(open(os.path.join(HERE, "golden", "tol_campaign.json")))
    return [(c["family"], c["seed"]) for c in rec["cases"] if not c["within_bar"]]


@pytest.mark.parametrize("family,seed", campaign_misses() + [("rooms", 1004), ("maze", 1103)])
@pytest.mark.parametrize("devices", [None, "0,0,0"])
def test_tol_jacobi_with_reference_checks_is_the_checkers_loop(family, seed, devices, jacobi_with_reference_checks, monkeypatch):
    """The campaign's seven cases outside the bar (Jacobi, eps = 1e-2) and two more maps at every epsilon: field, delta, iteration count at tolerance 0
    against oracle_tol_complete with the same switch -- which tests/test_jacobi_reference_checks.py holds to < 1e-6 of harmonic_complete_cpu on the seven."""
    import tol_campaign as TC

    lib = jacobi_with_reference_checks
    monkeypatch.setenv("EPIC_HIP_MATH", "tol")
    if devices:
        monkeypatch.setenv("EPIC_HIP_DEVICES", devices)
    m, u0, locked = TC.make_case(family, seed)
    for eps in ((1e-2,) if (family, seed) in campaign_misses() else TC.EPSILONS):
        p = O.Problem(m, u0, locked, eps, 100)
        assert lib.oracle_tol_complete(ct.byref(p.h), 0) == 0
        h = make(m, u0, locked, eps, 100)
        assert E.harmonic_complete_gpu(h, NT) == 0
        assert h.currentIteration == p.h.currentIteration and np.float32(h.delta) == np.float32(p.h.delta), eps
        assert np.array_equal(h.u_array().ravel(), p.u), eps


def ticks_by_the_checker(lib, p, ticks, steps):
    """The navigation node's loop (src/epic_navigation_node_harmonic.cpp:165-189): per tick one check -- the reference's half-sweep -- and steps - 1
    plain Jacobi sweeps."""
    for _ in range(ticks):
        assert lib.oracle_update_and_check(ct.byref(p.h)) in (0, 1)
        assert lib.oracle_jacobi_run(ct.byref(p.h), steps - 1) == 0


@pytest.mark.parametrize("defer", ["1", "0"])
@pytest.mark.parametrize("family", ["defaults", "work_lists", "fused_pairs_lists", "single_sweeps", "graphs_of_single_sweeps", "three_slabs"])
@pytest.mark.parametrize("m", [[96, 300], [20, 12, 34]])
def test_the_navigation_nodes_ticks_take_the_reference_checks(m, family, defer, jacobi_with_reference_checks, monkeypatch):
    lib = jacobi_with_reference_checks
    for k, v in FAMILIES[family].items():
        monkeypatch.setenv(k, v)
    monkeypatch.setenv("EPIC_HIP_DEFER", defer)
    u0, locked = synthetic_grid(m, 11, 0.06)
    p = O.Problem(m, u0, locked, 1e-6, 100)
    ticks_by_the_checker(lib, p, 4, 25)
    h = make(m, u0, locked)
    gpu_init(h)
    for _ in range(4):
        assert E.harmonic_update_and_check_gpu(h, NT) in (0, 1)
        delta = float(h.delta)
        for _ in range(24):
            assert E.harmonic_update_gpu(h, NT) == 0
    assert E.harmonic_get_potential_values_gpu(h) == 0
    gpu_fini(h)
    assert h.currentIteration == p.h.currentIteration == 100
    assert np.array_equal(h.u_array().ravel(), p.u)
    # the last check's delta: the checker's Problem holds the delta of its own last sweep, so state it again
    q = O.Problem(m, u0, locked, 1e-6, 100)
    ticks_by_the_checker(lib, q, 3, 25)
    assert lib.oracle_update_and_check(ct.byref(q.h)) in (0, 1)
    assert delta == float(q.h.delta)


@pytest.mark.parametrize("family", ["defaults", "fused_pairs", "three_slabs"])
def test_timed_batches_and_update_n_take_the_reference_checks(family, jacobi_with_reference_checks, monkeypatch):
    lib = jacobi_with_reference_checks
    for k, v in FAMILIES[family].items():
        monkeypatch.setenv(k, v)
    m = [130, 300]
    u0, locked = synthetic_grid(m, 17, 0.07)
    p = O.Problem(m, u0, locked, 1e-6, 20)
    ticks_by_the_checker(lib, p, 3, 20)                       # checks at iterations 0, 20, 40
    assert lib.oracle_jacobi_run(ct.byref(p.h), 7) == 0       # 60 .. 66 ...
    h = make(m, u0, locked, 1e-6, 20)
    gpu_init(h)
    ms = ct.c_float(0.0)
    assert E.epic_hip_timed_sweeps_gpu(h, 40, 20, ct.byref(ms)) == 0
    assert E.epic_hip_update_n_gpu(h, 20, 0) == 0             # 40 .. 59: update_n's check flag is about its LAST iteration only
    assert E.epic_hip_update_n_gpu(h, 7, 0) == 0
    assert E.harmonic_get_potential_values_gpu(h) == 0
    gpu_fini(h)
    # (update_n without a check runs plain Jacobi sweeps: the checker's iteration 40 above was a check, so restate 40 .. 66 plainly)
    q = O.Problem(m, u0, locked, 1e-6, 20)
    ticks_by_the_checker(lib, q, 2, 20)
    assert lib.oracle_jacobi_run(ct.byref(q.h), 27) == 0
    assert h.currentIteration == 67 and np.array_equal(h.u_array().ravel(), q.u)


def test_8192_squared_under_the_jacobi_scheme_has_the_cpu_statements_sha256():
    """The timed grid at full size (BASELINE configs[2]), precise arithmetic, Jacobi sweeps as fused pairs with work lists and 451 half-sweep checks:
    45 001 iterations, the reference's delta, and the sha256 over all 67 108 864 cells of the CPU statement of harmonic_complete_cpu
    (tests/golden/synthetic_8192.json, generate_8192_golden.py) -- the field the default (red-black) relaxation has in tests/test_gpu_tracked_pairs.py."""
    import hashlib

    g = json.load(open(os.path.join(HERE, "golden", "synthetic_8192.json")))
    u0, locked = synthetic_grid([8192, 8192])
    assert g["sha_u0"] == hashlib.sha256(u0.tobytes()).hexdigest() and g["sha_locked"] == hashlib.sha256(locked.tobytes()).hexdigest()
    h = make([8192, 8192], u0, locked)
    assert E.harmonic_complete_gpu(h, NT) == 0
    assert (h.currentIteration, float(h.delta)) == (g["iterations"], g["delta"]) == (45001, 2.384185791015625e-07)
    assert hashlib.sha256(h.u_array().tobytes()).hexdigest() == g["sha_u"]
