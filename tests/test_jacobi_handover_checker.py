"""CPU: the Jacobi checkers' handover rule (oracle/harmonic_oracle.c: oracle_jacobi_complete, oracle/tol_checker.c:
oracle_tol_complete) on the case that needs it -- tests/jacobi_handover_case.py.  Plain Jacobi is shown NOT to meet the
reference's test there (bounded run), the checkers with the rule end, and they end in the red-black oracle's field."""
import ctypes as ct

import numpy as np

import _oracle as O
from jacobi_handover_case import GRID, two_goal_sequence, set_goal


def first_goal_field(lib):
    u, locked, goals = two_goal_sequence()
    set_goal(u, locked, *goals[0])
    p = O.Problem(GRID, u, locked)
    assert lib.oracle_complete(ct.byref(p.h)) == 0
    u = p.u.reshape(GRID).copy()
    set_goal(u, locked, *goals[1])
    return u, locked


def test_plain_jacobi_flips_between_two_fields_for_ever():
    lib = O.oracle()
    u, locked = first_goal_field(lib)
    p = O.Problem(GRID, u, locked)
    assert lib.oracle_jacobi_run(ct.byref(p.h), 6000) == 0        # the reference's iteration needs 1 901 half-sweeps
    a = p.u.copy()
    assert lib.oracle_jacobi_run(ct.byref(p.h), 1) == 0
    b = p.u.copy()
    assert lib.oracle_jacobi_run(ct.byref(p.h), 1) == 0
    assert p.h.delta >= 1e-6                                       # the termination test does not fire ...
    assert np.array_equal(p.u, a) and not np.array_equal(a, b)     # ... and never will: an exact 2-cycle


def test_checkers_with_the_rule_end_in_the_reference_field():
    lib = O.oracle()
    u, locked = first_goal_field(lib)
    ref = O.Problem(GRID, u, locked)
    assert lib.oracle_complete(ct.byref(ref.h)) == 0
    reached = ref.u > -9e5
    for name, run in (("precise", lambda p: lib.oracle_jacobi_complete(ct.byref(p.h))),
                      ("tol", lambda p: lib.oracle_tol_complete(ct.byref(p.h), 0))):
        p = O.Problem(GRID, u, locked)
        assert run(p) == 0, name
        assert p.h.delta < 1e-6 and p.h.currentIteration % 100 == 1 and p.h.currentIteration < 10 * ref.h.currentIteration
        assert np.array_equal(p.u[~reached], ref.u[~reached])
        err = np.abs(p.u[reached].astype(np.float64) - ref.u[reached]) / np.maximum(1.0, np.abs(ref.u[reached]))
        assert err.max() <= 1e-5, (name, float(err.max()))
