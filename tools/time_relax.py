#!/usr/bin/env python3
"""Time harmonic_execute_gpu on the synthetic N x N grid: tools/time_relax.py [--size 8192] [--scheme jacobi|redblack]
[--track 0|1] [--rows-per-task R] [--math precise|tol].  Prints one JSON line (seconds, iterations, final active-tile
share).  Tuning experiments: EPIC_HIP_LIST_WAVES is read from the environment by the library."""
import argparse
import ctypes as ct
import json
import os

os.environ.setdefault("EPIC_HIP_STUDY", "1")   # this tool steers the kernel plan with study knobs (epic_amd/csrc/driver_config.cpp)
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from epic_amd import epic_harmonic as eh  # noqa: E402
from epic_amd.harmonic import Harmonic  # noqa: E402
from epic_amd.synthetic import synthetic_grid  # noqa: E402

E = eh._epic
ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=8192)
ap.add_argument("--scheme", default="jacobi")
ap.add_argument("--track", type=int, default=1)
ap.add_argument("--rows-per-task", type=int, default=0)
ap.add_argument("--math", default="precise")
ap.add_argument("--repeat", type=int, default=1)
a = ap.parse_args()

m = [a.size, a.size]
u0, locked = synthetic_grid(m, 20240601, 0.05)
h = Harmonic()
h.set_grid(m, u0, locked)
h.epsilon = 1e-6
h.numIterationsToStaggerCheck = 100
for fn in (E.harmonic_initialize_dimension_size_gpu, E.harmonic_initialize_potential_values_gpu,
           E.harmonic_initialize_locked_gpu):
    assert fn(h) == 0
assert E.harmonic_initialize_gpu(h, 1024) == 0
assert E.epic_hip_set_scheme(h, 1 if a.scheme == "redblack" else 0) == 0
assert E.epic_hip_set_math_mode(h, {"precise": 0, "fast": 1, "tol": 4}[a.math]) == 0
assert E.epic_hip_set_activity_tracking(h, a.track) == 0
if a.rows_per_task:
    assert E.epic_hip_set_rows_per_task(h, a.rows_per_task) == 0
assert E.harmonic_uninitialize_gpu(h) == 0   # harmonic_execute_gpu allocates delta itself
for _ in range(a.repeat):
    h.u_array().ravel()[:] = u0
    assert E.harmonic_update_model_gpu(h) == 0
    t0 = time.perf_counter()
    rc = E.harmonic_execute_gpu(h, 1024)
    dt = time.perf_counter() - t0
    assert rc == 0, rc
    act, tiles = ct.c_ulonglong(0), ct.c_ulonglong(0)
    E.epic_hip_activity_stats(h, ct.byref(act), ct.byref(tiles))
    print(json.dumps({"scheme": a.scheme, "track": a.track, "rows_per_task": a.rows_per_task, "math": a.math,
                      "list_waves": os.environ.get("EPIC_HIP_LIST_WAVES"), "seconds": round(dt, 3),
                      "iterations": int(h.currentIteration), "delta": float(h.delta),
                      "active_tiles_at_end": act.value, "tiles": tiles.value}), flush=True)
