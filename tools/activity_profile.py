#!/usr/bin/env python3
"""Where the tracked relaxation spends its time: every 1000 sweeps, the device time of the last 100 sweeps (one check +
99 plain, epic_hip_timed_sweeps_gpu) against the share of active tiles.  tools/activity_profile.py [N] [scheme]"""
import ctypes as ct
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from epic_amd import epic_harmonic as eh  # noqa: E402
from epic_amd.harmonic import Harmonic  # noqa: E402
from epic_amd.synthetic import synthetic_grid  # noqa: E402

E = eh._epic
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
scheme = sys.argv[2] if len(sys.argv) > 2 else "jacobi"
u0, locked = synthetic_grid([n, n])
h = Harmonic()
h.set_grid([n, n], u0, locked)
h.epsilon = 1e-6
h.numIterationsToStaggerCheck = 100
for fn in (E.harmonic_initialize_dimension_size_gpu, E.harmonic_initialize_potential_values_gpu,
           E.harmonic_initialize_locked_gpu):
    assert fn(h) == 0
assert E.harmonic_initialize_gpu(h, 1024) == 0
assert E.epic_hip_set_scheme(h, 1 if scheme == "redblack" else 0) == 0
ms = ct.c_float(0)
act, due, tiles = ct.c_ulonglong(0), ct.c_ulonglong(0), ct.c_ulonglong(0)
total = 0.0
print("sweeps  changed_share  due_share  us_per_sweep  us_per_due_share")
for block in range(46):
    assert E.epic_hip_timed_sweeps_gpu(h, 900, 100, ct.byref(ms)) == 0
    total += ms.value
    assert E.epic_hip_timed_sweeps_gpu(h, 100, 100, ct.byref(ms)) == 0
    total += ms.value
    E.epic_hip_activity_stats2(h, ct.byref(act), ct.byref(due), ct.byref(tiles))
    share, dshare = act.value / max(1, tiles.value), due.value / max(1, tiles.value)
    us = ms.value * 10.0
    print(f"{(block + 1) * 1000:6d}  {share:8.4f}  {dshare:8.4f}  {us:9.2f}  {us / max(dshare, 1e-9):9.1f}", flush=True)
print(f"device time of {46000} sweeps: {total / 1e3:.3f} s")
