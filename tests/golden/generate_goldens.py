#!/usr/bin/env python3
"""Generate the committed golden vectors by RUNNING THE REFERENCE (this container only).

The reference has no golden vectors or known-answer tests of its own, so parity is pinned by
executing its CPU solver -- /root/reference/libepic/src/harmonic/*.cpp compiled by oracle/Makefile into
oracle/_ref/libepic_ref.so -- on (a) seeded synthetic grids and (b) the reference's own PNG maps, and committing
inputs' hashes + outputs here.  The checked-in binary /root/reference/libepic/lib/libepic.so is used as a second
opinion on one case.  Nothing here is read at run time on the GPU box; only the .npz/.json outputs are.

Usage:  python tests/golden/generate_goldens.py [--skip-maps]      (maps take ~10 CPU-minutes)
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
import _oracle as O  # noqa: E402

SMALL_2D = [
    # name, m, seed, density
    ("g2d_16", [16, 16], 1, 0.05),
    ("g2d_32", [32, 32], 2, 0.05),
    ("g2d_64", [64, 64], 3, 0.05),
    ("g2d_23x37", [23, 37], 4, 0.10),
    ("g2d_5x7", [5, 7], 5, 0.0),
    ("g2d_3x3", [3, 3], 6, 0.0),
    ("g2d_8x300", [8, 300], 7, 0.05),
    ("g2d_70x66_dense", [70, 66], 8, 0.30),
]
SMALL_3D = [
    ("g3d_8", [8, 8, 8], 11, 0.05),
    ("g3d_16", [16, 16, 16], 12, 0.05),
    ("g3d_7x9x11", [7, 9, 11], 13, 0.10),
    ("g3d_20x12x34", [20, 12, 34], 14, 0.05),
]
HALF_SWEEPS = (1, 2, 3, 10)
MAPS = ("basic", "maze", "umass")


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def run_ref_steps(ref, prob, k):
    """k half-sweeps through the reference's single-step entry points; the last one is a check sweep."""
    for i in range(k):
        if i == k - 1:
            ref.harmonic_update_and_check_cpu(ct.byref(prob.h))
        else:
            ref.harmonic_update_cpu(ct.byref(prob.h))


def small_case(ref, name, m, seed, density, out, manifest):
    u0, locked = O.oracle_synthetic(m, seed, density)
    out[name + "/m"] = np.asarray(m, dtype=np.uint32)
    out[name + "/u0"] = u0
    out[name + "/locked"] = locked.astype(np.uint8)
    for k in HALF_SWEEPS:
        p = O.Problem(m, u0, locked, epsilon=1e-6, stagger=100)
        run_ref_steps(ref, p, k)
        out[f"{name}/rb{k}"] = p.u.copy()
        out[f"{name}/rb{k}_delta"] = np.float32(p.h.delta)
    p = O.Problem(m, u0, locked, epsilon=1e-6, stagger=10)
    rc = ref.harmonic_complete_cpu(ct.byref(p.h))
    assert rc == 0
    out[name + "/converged"] = p.u.copy()
    manifest["small"][name] = dict(m=list(map(int, m)), seed=seed, density=density, epsilon=1e-6, stagger=10,
                                   iterations=int(p.h.currentIteration), delta=float(p.h.delta),
                                   sha_u0=sha(u0), sha_locked=sha(locked))
    print(f"  {name}: {p.h.currentIteration} half-sweeps, delta {p.h.delta:.3e}")


def set_cells_case(ref, out):
    m = [8, 8]
    u0, locked = O.oracle_synthetic(m, 21, 0.1)
    v = np.array([[1, 1], [6, 2], [3, 3], [9, 1], [2, 9], [4, 4], [0, 0], [5, 5]], dtype=np.uint32)  # (x, y)
    types = np.array([0, 1, 2, 0, 1, 7, 2, 0], dtype=np.uint32)
    p = O.Problem(m, u0, locked)
    rc = ref.harmonic_utilities_set_cells_2d_cpu(ct.byref(p.h), len(types), v.ctypes.data_as(ct.POINTER(ct.c_uint)),
                                                 types.ctypes.data_as(ct.POINTER(ct.c_uint)))
    assert rc == 0
    out["set_cells/m"] = np.asarray(m, dtype=np.uint32)
    out["set_cells/u0"] = u0
    out["set_cells/locked0"] = locked.astype(np.uint8)
    out["set_cells/v"] = v
    out["set_cells/types"] = types
    out["set_cells/u1"] = p.u.copy()
    out["set_cells/locked1"] = p.locked.astype(np.uint8)


def map_case(ref, name, manifest, fields):
    m, u0, locked = O.load_png_reference_rule(os.path.join(HERE, "maps", name + ".png"))
    entry = dict(m=m, sha_u0=sha(u0), sha_locked=sha(locked), free=int((locked == 0).sum()),
                 goals=int((u0 == 0).sum()), runs={})
    rng = np.random.default_rng(12345)
    idx = np.sort(rng.choice(u0.size, size=4096, replace=False))
    fields[name + "/sample_idx"] = idx.astype(np.int64)
    for eps in (1e-3, 1e-6):
        p = O.Problem(m, u0, locked, epsilon=eps, stagger=100)
        t0 = time.time()
        rc = ref.harmonic_complete_cpu(ct.byref(p.h))
        dt = time.time() - t0
        assert rc == 0
        u = p.u
        free = p.locked == 0
        entry["runs"][f"{eps:g}"] = dict(rc=rc, iterations=int(p.h.currentIteration), delta=float(p.h.delta),
                                         sum=float(u[free].astype(np.float64).sum()), min=float(u[free].min()),
                                         max=float(u[free].max()), seconds=round(dt, 2), sha_u=sha(u))
        fields[f"{name}/samples_{eps:g}"] = u[idx].copy()
        if eps == 1e-6:
            fields[name + "/converged_1e-06"] = u.copy()
        print(f"  {name} eps={eps:g}: {p.h.currentIteration} half-sweeps, delta {p.h.delta:.3e}, {dt:.1f}s")
    manifest["maps"][name] = entry


def cross_check_shipped_binary(ref):
    """The checked-in libepic.so must agree bit for bit with the sources we compiled."""
    so = "/root/reference/libepic/lib/libepic.so"
    if not os.path.exists(so):
        return None
    lib = ct.CDLL(so)
    lib.harmonic_complete_cpu.argtypes = (ct.POINTER(O.CHarmonic),)
    u0, locked = O.oracle_synthetic([32, 32], 2, 0.05)
    a = O.Problem([32, 32], u0, locked, 1e-6, 100)
    b = O.Problem([32, 32], u0, locked, 1e-6, 100)
    assert lib.harmonic_complete_cpu(ct.byref(a.h)) == 0 and ref.harmonic_complete_cpu(ct.byref(b.h)) == 0
    ok = bool(np.array_equal(a.u, b.u) and a.h.currentIteration == b.h.currentIteration)
    assert ok, "shipped libepic.so and compiled sources disagree"
    return dict(iterations=int(a.h.currentIteration), bit_identical=ok)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--skip-maps", action="store_true")
    args = ap.parse_args()
    ref = O.ref()
    if ref is None:
        sys.exit("reference sources unavailable: goldens can only be generated where /root/reference exists")
    manifest = dict(generator="tests/golden/generate_goldens.py",
                    reference="oracle/_ref/libepic_ref.so = g++ -std=c++11 -O3 of libepic/src/harmonic/*.cpp",
                    small={}, maps={})
    manifest["shipped_binary_cross_check"] = cross_check_shipped_binary(ref)
    out = {}
    print("small grids")
    for name, m, seed, dens in SMALL_2D + SMALL_3D:
        small_case(ref, name, m, seed, dens, out, manifest)
    set_cells_case(ref, out)
    np.savez_compressed(os.path.join(HERE, "small_grids.npz"), **out)
    mpath = os.path.join(HERE, "manifest.json")
    if args.skip_maps and os.path.exists(mpath):
        manifest["maps"] = json.load(open(mpath)).get("maps", {})
    else:
        fields = {}
        print("maps")
        for name in MAPS:
            map_case(ref, name, manifest, fields)
        np.savez_compressed(os.path.join(HERE, "maps_converged.npz"), **fields)
    json.dump(manifest, open(mpath, "w"), indent=1, sort_keys=True)
    print("wrote", mpath)


if __name__ == "__main__":
    main()
