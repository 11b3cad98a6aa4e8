"""EPIC_HIP_JACOBI_CHECKS=reference, stated by the checkers (CPU; the device side: tests/test_gpu_jacobi_reference_checks.py).

A Jacobi sweep advances two interleaved red-black chains; the reference computes one of them.  With every CHECK iteration of a Jacobi run taken as
the reference's half-sweep of that iteration's colour, in place (oracle_set_jacobi_ref_checks; the library: epic_amd/csrc/driver_loop.hip, run_block),
the state after every check is the reference's own, so
  * with the reference's arithmetic the whole loop IS harmonic_complete_cpu: field, delta and iteration count bit for bit;
  * with the tol arithmetic the campaign's seven cases outside the bar (tests/golden/tol_campaign.json: Jacobi at eps = 1e-2, explained by the
    second chain's lag) end inside it, at the reference's iteration."""
import ctypes as ct
import json
import os

import numpy as np
import pytest

import _oracle as O
import tol_campaign as T

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture
def ref_checks():
    lib = O.oracle()
    lib.oracle_set_jacobi_ref_checks.argtypes = (ct.c_int,)
    lib.oracle_set_jacobi_ref_checks.restype = None
    lib.oracle_set_jacobi_ref_checks(1)
    yield lib
    lib.oracle_set_jacobi_ref_checks(0)


@pytest.mark.parametrize("name", ["g2d_16", "g2d_64", "g2d_23x37", "g2d_8x300", "g2d_70x66_dense", "g2d_3x3", "g3d_8", "g3d_7x9x11", "g3d_20x12x34"])
def test_the_jacobi_loop_with_reference_checks_is_the_reference_loop_bit_for_bit(goldens, name, ref_checks):
    g, info = goldens["small"], goldens["manifest"]["small"][name]
    p = O.Problem(g[name + "/m"], g[name + "/u0"], g[name + "/locked"], info["epsilon"], info["stagger"])
    assert ref_checks.oracle_jacobi_complete(ct.byref(p.h)) == 0
    assert p.h.currentIteration == info["iterations"] and float(p.h.delta) == info["delta"]
    assert np.array_equal(p.u, g[name + "/converged"])   # harmonic_complete_cpu's own field (tests/golden/generate_goldens.py)


def test_without_the_switch_the_jacobi_loop_is_what_it_was(goldens):
    """(the switch is off by default: the Jacobi loop's second chain ends near, not on, the reference's field)"""
    name = "g2d_23x37"
    g, info = goldens["small"], goldens["manifest"]["small"][name]
    p = O.Problem(g[name + "/m"], g[name + "/u0"], g[name + "/locked"], info["epsilon"], info["stagger"])
    assert O.oracle().oracle_jacobi_complete(ct.byref(p.h)) == 0
    assert not np.array_equal(p.u, g[name + "/converged"])


def test_the_campaigns_seven_misses_end_inside_the_bar_with_reference_checks(ref_checks):
    rec = json.load(open(os.path.join(HERE, "golden", "tol_campaign.json")))
    missed = [c for c in rec["cases"] if not c["within_bar"]]
    assert len(missed) == 7 and all(c["scheme"] == "jacobi" and c["epsilon"] == 1e-2 for c in missed)
    for c in missed:
        m, u0, locked = T.make_case(c["family"], c["seed"])
        pr = O.Problem(m, u0, locked, c["epsilon"], 100)
        T.reference_complete(pr)
        p = O.Problem(m, u0, locked, c["epsilon"], 100)
        assert ref_checks.oracle_tol_complete(ct.byref(p.h), 0) == 0
        reached = (pr.u > -9e5) & (np.asarray(locked).ravel() == 0)
        rel = np.abs(p.u.astype(np.float64) - pr.u) / np.maximum(1.0, np.abs(pr.u))
        assert p.h.currentIteration == pr.h.currentIteration, (c["family"], c["seed"])
        assert float(rel[reached].max()) < 1e-6, (c["family"], c["seed"], float(rel[reached].max()), c["max_rel"])
