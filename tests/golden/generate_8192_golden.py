"""The reference's converged field at the BENCHMARK'S OWN SIZE, stated on the CPU (round 6).

harmonic_complete_cpu needs ~15 hours for the 8192 x 8192 grid of BASELINE configs[2] on one core, so until round 6 the converged-field parity
of the timed grid was HIP against HIP (the tol mode against the library's bit-exact default in the same run).  The reference's iteration is a
red-black half-sweep: within one, every updated cell reads only cells of the other colour, so its rows can be dealt to threads and the result --
field, max |du|, iteration count -- is the sequential one bit for bit (oracle/harmonic_oracle.c: oracle_complete_parallel_2d; held to the
reference-generated 512^2 / 1024^2 goldens by tests/test_oracle.py).  This script runs that loop on the benchmark's grid (seed 20240601, 5 %
obstacles, centre goal, eps = 1e-6, check every 100) and writes what pins the result: iteration count, final delta, sha256 of the whole field, min /
max, and 16 384 seeded samples.  tests/test_gpu_bench_parity.py then holds the library's default relaxation of the same grid (2.3 s on the
device) to it, sha256 included.

    python tests/golden/generate_8192_golden.py [--threads 8]        (~2 h on 8 cores of the build container; progress on stderr)
    python tests/golden/generate_8192_golden.py --cube 512           (BASELINE configs[4]: 512^3, 3 801 half-sweeps of the 7-point stencil)
"""
import argparse
import ctypes as ct
import hashlib
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import _oracle as O  # noqa: E402
from epic_amd.synthetic import DEFAULT_SEED, synthetic_grid  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=os.cpu_count() or 1)
    ap.add_argument("--size", type=int, default=8192)
    ap.add_argument("--cube", type=int, default=0, help="N: the N x N x N grid of BASELINE configs[4] (512) instead of the 2-D one")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    n = a.cube or a.size
    m = [n, n, n] if a.cube else [n, n]
    u0, locked = synthetic_grid(m)
    lib = O.oracle()
    loop = lib.oracle_complete_parallel_3d if a.cube else lib.oracle_complete_parallel_2d
    loop.argtypes = (ct.POINTER(O.CHarmonic), ct.c_int, ct.c_uint)
    loop.restype = ct.c_int
    p = O.Problem(m, u0, locked, 1e-6, 100)
    t0 = time.time()
    rc = loop(ct.byref(p.h), a.threads, 100 if a.cube else 1000)
    secs = time.time() - t0
    assert rc == 0, rc
    rng = np.random.default_rng(20240601)
    idx = np.sort(rng.choice(int(np.prod(m)), size=16384, replace=False)).astype(np.int64)
    reached = (p.u > -9e5) & (locked == 0)
    doc = {
        "generator": "tests/golden/generate_8192_golden.py --threads %d%s" % (a.threads, " --cube %d" % a.cube if a.cube else ""),
        "what": "harmonic_complete_cpu's loop on the benchmark's grid with its half-sweeps dealt to threads (oracle_complete_parallel_%dd: the sequential result bit for bit)" % len(m),
        "m": m, "seed": DEFAULT_SEED, "density": 0.05, "epsilon": 1e-6, "stagger": 100,
        "iterations": int(p.h.currentIteration), "delta": float(p.h.delta), "seconds": round(secs, 1), "threads": a.threads,
        "sha_u0": hashlib.sha256(u0.tobytes()).hexdigest(), "sha_locked": hashlib.sha256(locked.tobytes()).hexdigest(),
        "sha_u": hashlib.sha256(p.u.tobytes()).hexdigest(),
        "free": int((locked == 0).sum()), "unreached_free": int(((locked == 0) & ~reached).sum()),
        "min": float(p.u[reached].min()), "max": float(p.u[reached].max()),
        "sample_index": idx.tolist(), "sample_u": [float(x) for x in p.u[idx]],
    }
    out = a.out or os.path.join(HERE, "synthetic_%dcubed.json" % n if a.cube else "synthetic_%d.json" % n)
    with open(out, "w") as f:
        json.dump(doc, f, separators=(",", ":"))
    print("wrote %s: %d iterations, delta %.3e, %.0f s" % (out, doc["iterations"], doc["delta"], secs))


if __name__ == "__main__":
    main()
