"""harmonic_execute_gpu under the default scheme (Jacobi) must END wherever the reference's loop ends.

A Jacobi iteration is two interleaved red-black chains; in f32 they can stagnate one unit in the last place apart, every
cell then flips between the two for ever and max |du| stays at one ulp of |u| -- above eps = 1e-6 as soon as |u| > 8.  The
case that showed it is the nav_core plugin's SECOND makePlan (src/epic_nav_core_plugin.cpp:234-338: the goal moves, the
field of the first goal is the start): plain Jacobi sits at delta = 7.6e-6 for 400 000 iterations and counting, the
reference's red-black iteration stops after 1 901.  harmonic_execute_gpu therefore hands over to the reference's in-place
half-sweeps at the first check with delta < 1 that is not below the previous check's delta (driver_loop.hip, "Jacobi
handover"); the checkers state the same rule (oracle_jacobi_complete, oracle_tol_complete), so iteration counts, delta and
fields are compared bit for bit, one device and row slabs alike.
"""
import ctypes as ct

import numpy as np
import pytest

import _oracle as O
from epic_amd import epic_harmonic as eh
from epic_amd.harmonic import Harmonic
from jacobi_handover_case import GRID, two_goal_sequence, set_goal

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(600)]

E = eh._epic


def gpu_complete(u, locked, env, monkeypatch):
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    h = Harmonic()
    h.set_grid(GRID, u, locked)
    h.epsilon = 1e-6
    h.numIterationsToStaggerCheck = 100
    assert E.harmonic_complete_gpu(h, 1024) == 0
    return h.u_array().reshape(GRID).copy(), int(h.currentIteration), float(h.delta)


@pytest.fixture(autouse=True)
def jacobi_session(monkeypatch):
    """These tests are about the Jacobi scheme whatever the session runs (tests/conftest.py: EPIC_TEST_SCHEME)."""
    monkeypatch.setenv("EPIC_HIP_SCHEME", "jacobi")


@pytest.mark.parametrize("math", ["precise", "tol"])
@pytest.mark.parametrize("devices", [None, "0,0,0"])
def test_second_goal_terminates_and_equals_the_checker(math, devices, monkeypatch):
    lib = O.oracle()
    env = {"EPIC_HIP_MATH": math}
    if devices:
        env["EPIC_HIP_DEVICES"] = devices
        env["EPIC_HIP_HALO"] = "3"
    u, locked, goals = two_goal_sequence()
    cu, cl = u.copy(), locked.copy()
    counts = []
    for x, y in goals:
        set_goal(u, locked, x, y)
        set_goal(cu, cl, x, y)
        got, it, delta = gpu_complete(u, locked, env, monkeypatch)
        p = O.Problem(GRID, cu, cl)
        rc = lib.oracle_jacobi_complete(ct.byref(p.h)) if math == "precise" else lib.oracle_tol_complete(ct.byref(p.h), 0)
        assert rc == 0
        assert it == p.h.currentIteration and delta == p.h.delta and delta < 1e-6
        assert np.array_equal(got.ravel(), p.u), f"goal ({x}, {y})"
        u, cu = got, p.u.reshape(GRID).copy()
        counts.append(it)
    # the second call is the one plain Jacobi never finishes: it ended, and not at once
    assert counts[1] > 1000 and counts[1] % 100 == 1


def test_handover_leaves_the_scheme_as_it_was(monkeypatch):
    """harmonic_execute_gpu on a state that stays initialised (the ctypes wrapper's solve path): after a handover the next
    iterations requested through harmonic_update_gpu are Jacobi again."""
    monkeypatch.setenv("EPIC_HIP_MATH", "precise")
    u, locked, goals = two_goal_sequence()
    set_goal(u, locked, *goals[0])
    first, _, _ = gpu_complete(u, locked, {}, monkeypatch)
    set_goal(first, locked, *goals[1])
    h = Harmonic()
    h.set_grid(GRID, first, locked)
    h.epsilon = 1e-6
    h.numIterationsToStaggerCheck = 100
    for fn in (E.harmonic_initialize_dimension_size_gpu, E.harmonic_initialize_potential_values_gpu,
               E.harmonic_initialize_locked_gpu):
        assert fn(h) == 0
    assert E.harmonic_execute_gpu(h, 1024) == 0          # hands over on the way (previous test)
    done = h.u_array().reshape(GRID).copy()
    assert E.harmonic_initialize_gpu(h, 1024) == 0
    assert E.epic_hip_update_n_gpu(h, 3, 1) in (0, 1)     # three more iterations: Jacobi sweeps of the whole grid
    assert E.harmonic_get_potential_values_gpu(h) == 0
    p = O.Problem(GRID, done, locked)
    assert O.oracle().oracle_jacobi_run(ct.byref(p.h), 3) == 0
    assert np.array_equal(h.u_array().ravel(), p.u) and float(h.delta) == float(p.h.delta)
    for fn in (E.harmonic_uninitialize_gpu, E.harmonic_uninitialize_dimension_size_gpu,
               E.harmonic_uninitialize_potential_values_gpu, E.harmonic_uninitialize_locked_gpu):
        assert fn(h) == 0
