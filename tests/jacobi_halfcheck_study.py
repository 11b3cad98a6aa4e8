"""A study with the checkers, no product code involved (round 6): would running every CHECK iteration of a Jacobi run as a
red-black half-sweep remove the second-chain lag that explains the campaign's remaining misses (tests/tol_campaign.py: Jacobi at eps = 1e-2)?

Claim: after Jacobi sweeps 1 .. K-1 the cells of the other colour hold chain-1 values of sweep K-1, so a half-sweep of the reference's colour at
iteration K leaves exactly the reference's red-black state R_K.  Checked here in two steps:
  1. exactly, with the reference's arithmetic: K-1 Jacobi sweeps (oracle_jacobi_run) + one half-sweep (oracle_update) == K half-sweeps (oracle_update), bit for bit;
  2. on the campaign's missed cases, with the tol arithmetic: the tol loop of oracle_tol_complete restated block by block (oracle_tol_run) with
     half-sweep checks, against the reference's harmonic_complete_cpu.

    python tests/jacobi_halfcheck_study.py            (CPU, a minute; test infrastructure: uses oracle/)"""
import ctypes as ct
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import _oracle as O  # noqa: E402
import tol_campaign as T  # noqa: E402
from epic_amd.synthetic import synthetic_grid  # noqa: E402


def step_1():
    lib = O.oracle()
    for m, seed in (([37, 53], 3), ([64, 200], 5), ([12, 20, 31], 7)):
        u0, locked = synthetic_grid(m, seed, 0.08)
        O.scramble_free(m, u0, locked, seed=seed + 1, lo=-40.0, hi=0.0)
        for K in (1, 2, 7, 30):
            a = O.Problem(m, u0, locked)
            for _ in range(K):
                lib.oracle_update(ct.byref(a.h))
            b = O.Problem(m, u0, locked)
            if K > 1:
                assert lib.oracle_jacobi_run(ct.byref(b.h), K - 1) == 0
            lib.oracle_update(ct.byref(b.h))      # the half-sweep of iteration K (currentIteration == K - 1)
            assert np.array_equal(a.u, b.u), (m, K)
    print("step 1: K - 1 Jacobi sweeps + the half-sweep of iteration K == K half-sweeps of the reference, bit for bit (2-D and 3-D)")


def tol_jacobi_with_half_checks(m, u0, locked, eps, stagger=100):
    """oracle_tol_complete's loop for the Jacobi scheme, restated per block, with every check iteration a red-black half-sweep."""
    lib = O.oracle()
    p = O.Problem(m, u0, locked, eps, stagger)
    m_max = max(m)
    factor = np.float32(100.0) if np.float32(eps) <= np.float32(1e-5) else np.float32(10.0)
    below = float(factor * np.float32(eps))
    converged, finishing, checks = False, False, 0
    while not converged or p.h.currentIteration < m_max:
        check = p.h.currentIteration % stagger == 0
        if finishing:
            (lib.oracle_update_and_check if check else lib.oracle_update)(ct.byref(p.h))
            converged = bool(check and p.h.delta < p.h.epsilon)
            continue
        if check:
            assert lib.oracle_tol_run(ct.byref(p.h), 1, 1) == 0          # a half-sweep of the reference's colour, tol arithmetic
            d = float(p.h.delta)
            converged = d < eps
            if d < below and not (d == 0.0 and checks == 0):
                finishing = True
                if not np.float32(eps) > np.float32(1e-5):
                    converged = False
            checks += 1
        else:
            n = stagger - p.h.currentIteration % stagger
            assert lib.oracle_tol_run(ct.byref(p.h), n, 0) == 0          # plain Jacobi sweeps up to the next check
            converged = False
    return p


def step_2():
    rec = json.load(open(os.path.join(HERE, "golden", "tol_campaign.json")))
    missed = [c for c in rec["cases"] if not c["within_bar"]]
    print("step 2: the campaign's %d cases outside the bar (all Jacobi, eps = 1e-2), tol arithmetic, half-sweep checks:" % len(missed))
    worst = 0.0
    for c in missed:
        m, u0, locked = T.make_case(c["family"], c["seed"])
        pr = O.Problem(m, u0, locked, c["epsilon"], 100)
        T.reference_complete(pr)
        p = tol_jacobi_with_half_checks(m, u0, locked, c["epsilon"])
        reached = (pr.u > -9e5) & (locked == 0)
        rel = np.abs(p.u.astype(np.float64) - pr.u) / np.maximum(1.0, np.abs(pr.u))
        w = float(rel[reached].max())
        worst = max(worst, w)
        print("  %-6s %d %s: recorded %.2e -> %.2e with half-sweep checks; iterations %d (reference %d)" % (
            c["family"], c["seed"], c["m"], c["max_rel"], w, p.h.currentIteration, pr.h.currentIteration))
    print("worst: %.2e (bar 1e-5)" % worst)


if __name__ == "__main__":
    step_1()
    step_2()
