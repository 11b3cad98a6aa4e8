"""Quick device check of the tol math mode: bit-identity with the checker's CPU twin (oracle/tol_checker.c) on a few grids,
Jacobi and red-black, then the time per 8192^2 sweep next to precise / traffic.  python tools/tol_gpu_check.py [--time]"""
import ctypes as ct
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _oracle as O  # noqa: E402
from epic_amd import epic_harmonic as eh  # noqa: E402
from epic_amd.harmonic import Harmonic  # noqa: E402
from epic_amd.synthetic import synthetic_grid  # noqa: E402

E = eh._epic


def gpu(m, u0, locked, k, scheme, track=0, rpt=0):
    h = Harmonic()
    h.set_grid(m, u0, locked)
    h.epsilon = 1e-6
    for fn in (E.harmonic_initialize_dimension_size_gpu, E.harmonic_initialize_potential_values_gpu,
               E.harmonic_initialize_locked_gpu):
        assert fn(h) == 0
    assert E.harmonic_initialize_gpu(h, 1024) == 0
    assert E.epic_hip_set_math_mode(h, eh.MATH_TOL) == 0 and E.epic_hip_set_scheme(h, scheme) == 0
    assert E.epic_hip_set_activity_tracking(h, track) == 0
    if rpt:
        assert E.epic_hip_set_rows_per_task(h, rpt) == 0
    assert E.epic_hip_update_n_gpu(h, k, 1) in (0, 1)
    assert E.harmonic_get_potential_values_gpu(h) == 0
    d = float(h.delta)
    for fn in (E.harmonic_uninitialize_gpu, E.harmonic_uninitialize_dimension_size_gpu,
               E.harmonic_uninitialize_potential_values_gpu, E.harmonic_uninitialize_locked_gpu):
        assert fn(h) == 0
    return h.u_array().ravel().copy(), d


def main():
    bad = 0
    grids = (([16, 16], 1, 0.05, 0), ([23, 37], 4, 0.1, 0), ([70, 66], 8, 0.3, 0), ([257, 513], 9, 0.05, 0),
                               ([64, 1030], 10, 0.05, 0), ([200, 700], 3, 0.05, 16), ([1200, 3000], 5, 0.05, 0), ([96, 300], 6, 0.05, 8))
    if "--quick" in sys.argv:
        grids = (([96, 300], 6, 0.05, 8), ([200, 700], 3, 0.05, 16))
    for m, seed, dens, rpt in grids:
        u0, locked = synthetic_grid(m, seed, dens)
        free = np.flatnonzero(locked == 0)
        for idx in (free[0], free[-1]):
            u0[idx] = 0.0
            locked[idx] = 1
        for scheme in (0, 1):
            for k in (1, 2, 7, 40):
                p = O.Problem(m, u0, locked)
                assert O.oracle().oracle_tol_run(ct.byref(p.h), k, scheme) == 0
                for track in (0, 1):
                    got, d = gpu(m, u0, locked, k, scheme, track, rpt)
                    same = np.array_equal(got, p.u) and d == float(p.h.delta)
                    if not same:
                        bad += 1
                        diff = np.flatnonzero(got != p.u)
                        print("MISMATCH", m, "scheme", scheme, "k", k, "track", track, "cells", diff.size, "first", diff[:5],
                              "delta", d, float(p.h.delta))
        print("checked", m, flush=True)
    print("tol bit-identity mismatches:", bad, flush=True)
    if "--time" in sys.argv:
        n = 8192
        u0, locked = synthetic_grid([n, n])
        for math in ("tol", "precise", "traffic"):
            h = Harmonic()
            h.set_grid([n, n], u0, locked)
            h.epsilon = 1e-6
            for fn in (E.harmonic_initialize_dimension_size_gpu, E.harmonic_initialize_potential_values_gpu,
                       E.harmonic_initialize_locked_gpu):
                assert fn(h) == 0
            assert E.harmonic_initialize_gpu(h, 1024) == 0
            assert E.epic_hip_set_math_mode(h, {"precise": 0, "traffic": 2, "tol": 4}[math]) == 0
            assert E.epic_hip_set_activity_tracking(h, 0) == 0
            if math != "traffic":
                assert E.epic_hip_update_n_gpu(h, 20000, 0) == 0
            ms = ct.c_float(0)
            for rep in range(3):
                assert E.epic_hip_timed_sweeps_gpu(h, 500, 100, ct.byref(ms)) == 0
                print(math, "us per 8192^2 sweep: %.2f" % (ms.value * 1e3 / 500), flush=True)
            for fn in (E.harmonic_uninitialize_gpu, E.harmonic_uninitialize_dimension_size_gpu,
                       E.harmonic_uninitialize_potential_values_gpu, E.harmonic_uninitialize_locked_gpu):
                assert fn(h) == 0
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
