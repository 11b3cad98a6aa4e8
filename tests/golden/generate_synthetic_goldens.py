#!/usr/bin/env python3
"""Reference-generated converged fields of the BENCHMARK'S grid family at sizes the reference finishes in minutes.

bench.py times BASELINE configs[2]: the synthetic 8192 x 8192 grid (epic_amd/synthetic.py, seed 20240601, 5 % obstacles,
one goal).  The reference's harmonic_complete_cpu needs ~15 h for that size on one core, so the parity of the benchmarked
arithmetic on THAT family is pinned here on its smaller members, same generator, same seed, same eps = 1e-6 / stagger 100:
512 x 512 (16 s) and 1024 x 1024 (107 s) -- and at 8192 x 8192 on the device itself against the reference-identical
`precise` + `redblack` run (tests/test_gpu_bench_parity.py).

Runs the reference (this container only): oracle/_ref/libepic_ref.so = the reference's own harmonic_cpu.cpp compiled by
oracle/Makefile.  Output: tests/golden/synthetic_converged.npz (fields, float32) + an entry in manifest.json.
"""
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

SIZES = (512, 1024)
SEED = 20240601


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    ref = O.ref()
    if ref is None:
        sys.exit("reference sources unavailable: goldens can only be generated where /root/reference exists")
    from epic_amd.synthetic import synthetic_grid   # the generator bench.py uses (bit-identical to oracle_synthetic)

    out, entries = {}, {}
    for n in SIZES:
        m = [n, n]
        u0, locked = synthetic_grid(m, seed=SEED, density=0.05)
        p = O.Problem(m, u0, locked, epsilon=1e-6, stagger=100)
        t0 = time.time()
        rc = ref.harmonic_complete_cpu(ct.byref(p.h))
        dt = time.time() - t0
        assert rc == 0
        free = p.locked == 0
        reached = free & (p.u > -9e5)
        out["s%d/converged" % n] = p.u.copy()
        entries[str(n)] = dict(m=m, seed=SEED, density=0.05, epsilon=1e-6, stagger=100, iterations=int(p.h.currentIteration),
                               delta=float(p.h.delta), seconds=round(dt, 1), sha_u0=sha(u0), sha_locked=sha(locked),
                               sha_u=sha(p.u), free=int(free.sum()), unreached_free=int((free & ~reached).sum()),
                               min=float(p.u[reached].min()), max=float(p.u[reached].max()))
        print("  synthetic %dx%d: %d half-sweeps, delta %.3e, %.1fs, u in [%.3f, %.3f]"
              % (n, n, p.h.currentIteration, p.h.delta, dt, entries[str(n)]["min"], entries[str(n)]["max"]), flush=True)
    np.savez_compressed(os.path.join(HERE, "synthetic_converged.npz"), **out)
    mpath = os.path.join(HERE, "manifest.json")
    manifest = json.load(open(mpath))
    manifest["synthetic"] = dict(generator="tests/golden/generate_synthetic_goldens.py", grids=entries)
    json.dump(manifest, open(mpath, "w"), indent=1, sort_keys=True)
    print("wrote", mpath)


if __name__ == "__main__":
    main()
