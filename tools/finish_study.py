#!/usr/bin/env python3
"""CPU study behind the finishing rule of tol relaxations ("Finish"; profiles/r03_experiments.txt item 11).

  python tools/finish_study.py <basic|maze|umass> <switch delta> [scheme 0 = Jacobi | 1 = red-black]

Runs the tol iteration (oracle/tol_checker.c, with its own finishing rule switched off) on one of the reference's maps until a
check finds delta < <switch delta>, then the reference's iteration (oracle/harmonic_oracle.c: oracle_complete's loop, i.e.
harmonic_complete_cpu restated) from that state until ITS test fires, and prints the distance of both end states from the
field harmonic_complete_cpu itself converged (tests/golden/maps_converged.npz).  umass takes ~10 minutes on a few cores.
Test infrastructure: uses oracle/, never the product."""
import ctypes as ct
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _oracle as O  # noqa: E402
from epic_amd.harmonic_map import HarmonicMap  # noqa: E402  (map loader only: host code, no device)

name, sw = sys.argv[1], float(sys.argv[2])
scheme = int(sys.argv[3]) if len(sys.argv) > 3 else 0
g = os.path.join(ROOT, "tests", "golden")
h = HarmonicMap().load(os.path.join(g, "maps", name + ".png"))
m = [int(h.m[i]) for i in range(2)]
u0, locked = h.u_array().ravel().copy(), h.locked_array().ravel().copy()
want = np.load(os.path.join(g, "maps_converged.npz"))[name + "/converged_1e-06"].ravel()


def dist(got):
    reached = (locked == 0) & (want > -9e5)
    d = np.abs(got[reached].astype(np.float64) - want[reached])
    return float((d / np.maximum(1.0, np.abs(want[reached]))).max()), float(d.max())


lib = O.oracle()
lib.oracle_tol_set_finish(0)
p = O.Problem(m, u0, locked, sw, 100)
t0 = time.time()
assert lib.oracle_tol_complete(ct.byref(p.h), scheme) == 0
it1 = int(p.h.currentIteration)
print("%s, switch at delta < %g: tol iteration %d iterations, delta %.3e, max rel %.3e / max abs %.3e from the reference's field (%.0f s)"
      % ((name, sw, it1, float(p.h.delta)) + dist(p.u) + (time.time() - t0,)), flush=True)
p.h.epsilon = 1e-6
t0 = time.time()
assert lib.oracle_complete(ct.byref(p.h)) == 0          # (restarts its iteration count at 0)
print("%s: + %d iterations of the reference's own, delta %.3e, max rel %.3e / max abs %.3e (%.0f s)"
      % ((name, int(p.h.currentIteration), float(p.h.delta)) + dist(p.u) + (time.time() - t0,)), flush=True)
