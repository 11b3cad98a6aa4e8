#!/usr/bin/env python3
"""Path request after a relaxation, two ways: the reference's flow -- copy the field to the host
(harmonic_get_potential_values_gpu) and walk it with harmonic_compute_path_2d_cpu -- against the walk on the resident
field (epic_hip_compute_path(s)_2d_gpu).  Synthetic N x N grid, relaxed to 1e-6 first.  tools/time_paths.py [N]"""
import ctypes as ct
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from epic_amd import epic_harmonic as eh  # noqa: E402
from epic_amd.harmonic import Harmonic  # noqa: E402
from epic_amd.synthetic import synthetic_grid  # noqa: E402

E = eh._epic
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
m = [n, n]
u0, locked = synthetic_grid(m, 20240601, 0.05)
h = Harmonic()
h.set_grid(m, u0, locked)
h.epsilon = 1e-6
h.numIterationsToStaggerCheck = 100
for fn in (E.harmonic_initialize_dimension_size_gpu, E.harmonic_initialize_potential_values_gpu,
           E.harmonic_initialize_locked_gpu):
    assert fn(h) == 0
t0 = time.perf_counter()
assert E.harmonic_execute_gpu(h, 1024) == 0
relax_s = time.perf_counter() - t0          # (execute ends with the D2H of u: the host copy is current)

lk = locked.reshape(m)
rng = np.random.default_rng(3)
free = np.argwhere(lk == 0)
starts = free[rng.choice(len(free), 64, replace=False)][:, ::-1].astype(np.float32)   # (x, y)
PF = ct.POINTER(ct.c_float)
step, cd, max_len = 0.2, 0.4, 200000

# reference flow, one request: D2H of the field + host walk
t0 = time.perf_counter()
assert E.harmonic_get_potential_values_gpu(h) == 0
d2h_s = time.perf_counter() - t0
host_s, host_k = [], []
for i in range(4):
    k, raw = ct.c_uint(0), PF()
    t0 = time.perf_counter()
    rc = E.harmonic_compute_path_2d_cpu(h, float(starts[i, 0]), float(starts[i, 1]), step, cd, max_len, ct.byref(k), ct.byref(raw))
    host_s.append(time.perf_counter() - t0)
    host_k.append(int(k.value) if rc == 0 else -rc)
    ref = np.ctypeslib.as_array(raw, shape=(2 * k.value,)).copy() if rc == 0 else None
    if rc == 0:
        E.harmonic_free_path_cpu(ct.byref(raw))
    # the same request on the device
    k2, raw2 = ct.c_uint(0), PF()
    t0 = time.perf_counter()
    rc2 = E.epic_hip_compute_path_2d_gpu(h, float(starts[i, 0]), float(starts[i, 1]), step, cd, max_len, ct.byref(k2), ct.byref(raw2))
    dev = time.perf_counter() - t0
    same = None
    if rc2 == 0:
        got = np.ctypeslib.as_array(raw2, shape=(2 * k2.value,)).copy()
        E.harmonic_free_path_cpu(ct.byref(raw2))
        same = bool(rc == 0 and got.tobytes() == ref.tobytes())
    print(json.dumps({"start": starts[i].tolist(), "host_rc": rc, "dev_rc": rc2, "points": host_k[-1],
                      "host_walk_s": round(host_s[-1], 5), "device_request_s": round(dev, 5), "bit_identical": same}), flush=True)

# a batch of 64 requests in one launch
kk = np.zeros(64, dtype=np.uint32)
rcs = np.zeros(64, dtype=np.int32)
out = np.empty((64, 2 * max_len), dtype=np.float32)
t0 = time.perf_counter()
assert E.epic_hip_compute_paths_2d_gpu(h, 64, starts.ctypes.data_as(PF), step, cd, max_len, kk.ctypes.data_as(eh._UP),
                                       rcs.ctypes.data_as(ct.POINTER(ct.c_int)), out.ctypes.data_as(PF)) == 0
batch_s = time.perf_counter() - t0
print(json.dumps({"grid": m, "relax_s": round(relax_s, 3), "field_d2h_s": round(d2h_s, 4),
                  "batch64_device_s": round(batch_s, 4), "batch_points": int(kk.sum()), "batch_ok": int((rcs == 0).sum())}))
