#!/usr/bin/env python3
"""us per iteration of the fused tol Jacobi pass on slab-shaped grids (rows x 8192: what one GPU of 8, 4, 2 holds) against the task
height (EPIC_HIP_FUSED_ROWS; 0 = the rule): python tools/exp_slab_heights.py   (one process per point: the knob is read per context)"""
import json, os, subprocess, sys

os.environ.setdefault("EPIC_HIP_STUDY", "1")   # EPIC_HIP_FUSED_ROWS is a study knob (epic_amd/csrc/driver_config.cpp); the children inherit it
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, ctypes as ct, os
sys.path.insert(0, %r)
import torch
from epic_amd import epic_harmonic as eh
from epic_amd.harmonic import Harmonic
from epic_amd.synthetic import synthetic_grid
E = eh._epic
rows = int(sys.argv[1])
grid = [rows, 8192]
u0, locked = synthetic_grid(grid)
h = Harmonic(); h.set_grid(grid, u0, locked); h.epsilon = 1e-6; h.numIterationsToStaggerCheck = 100
for fn in (E.harmonic_initialize_dimension_size_gpu, E.harmonic_initialize_potential_values_gpu, E.harmonic_initialize_locked_gpu):
    assert fn(h) == 0
assert E.harmonic_initialize_gpu(h, 1024) == 0
assert E.epic_hip_set_math_mode(h, 4) == 0 and E.epic_hip_set_scheme(h, 0) == 0 and E.epic_hip_set_activity_tracking(h, 0) == 0
assert E.epic_hip_update_n_gpu(h, 1500, 0) == 0
ms = ct.c_float(0)
for _ in range(2): E.epic_hip_timed_sweeps_gpu(h, 100, 100, ct.byref(ms))
dev = 0.0
for _ in range(6):
    E.epic_hip_timed_sweeps_gpu(h, 100, 100, ct.byref(ms)); dev += ms.value
print("%%.2f %%d" %% (dev / 6 * 10, E.epic_hip_fused_rows_per_task(h)))
''' % ROOT
for rows in (1024, 2048, 4096):
    line = []
    for r in (0, 8, 12, 16, 20, 24, 32, 40, 48):
        env = dict(os.environ, EPIC_HIP_FUSED_ROWS=str(r), EPIC_HIP_TUNE="0")
        if r == 0:
            env.pop("EPIC_HIP_FUSED_ROWS")
        out = subprocess.run([sys.executable, "-c", CHILD, str(rows)], env=env, capture_output=True, text=True).stdout.strip().splitlines()
        us, used = out[-1].split() if out else ("nan", "0")
        line.append("%s: %s%s" % ("rule" if r == 0 else r, us, " (%s rows)" % used if r == 0 else ""))
    print("%d x 8192, us per iteration:  %s" % (rows, "   ".join(line)), flush=True)
