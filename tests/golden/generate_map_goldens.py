#!/usr/bin/env python3
"""Golden vectors for the reference's OTHER maps and for the epsilons its callers use (this container only).

`generate_goldens.py` pins basic / maze / umass at eps 1e-3 and 1e-6.  The reference ships ten more distinct
grids (libepic/tests/maps/*.png, libepic/tests/batch/*.png) and its callers relax at
eps = 1e-3 (src/epic_nav_core_plugin.cpp:61,85; src/epic_navigation_node_harmonic.cpp:64;
libepic/tests/maps/maps.py:67) or 1e-2 (libepic/python/epic/harmonic.py:45,54).  This script RUNS THE
REFERENCE -- harmonic_complete_cpu of oracle/_ref/libepic_ref.so, the reference's own sources compiled by
oracle/Makefile -- on every map at eps in {1e-2, 1e-3, 1e-6}, stagger 100 (the wrapper's default,
harmonic.py:50), and commits per run: rc, iteration count, final delta, sha256 of the whole field, f64 sum / min /
max over the free cells, and the field at 16 384 seeded cell indices.  The whole field is kept only for maze_4 at
1e-3 (the run the reference's own script makes on the GPU, maps.py:43-67).

One job per (map, eps) so that the long ones (maze_3 at 1e-6: hours of one core) run side by side:

    python tests/golden/generate_map_goldens.py --jobs 5            # everything missing, 5 processes
    python tests/golden/generate_map_goldens.py --only maze_4:0.001 # one job, in this process
    python tests/golden/generate_map_goldens.py --merge             # parts -> ref_maps.npz + ref_maps.json

The PNGs under tests/golden/maps/ are the reference's fixtures (data); copies are made by --copy-maps.
Nothing here is read at run time on the GPU box; only ref_maps.npz / ref_maps.json / the PNGs are.
"""
import argparse
import ctypes as ct
import hashlib
import json
import os
import shutil
import subprocess
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import _oracle as O  # noqa: E402

REF_TESTS = "/root/reference/libepic/tests"
# fixture name -> the reference file it is a copy of
SOURCES = {
    "basic": "maps/basic.png",
    "maze": "maps/maze_1.png",            # = /root/reference/maps/maze.png
    "umass": "../../maps/umass.png",
    "c_space": "maps/c_space.png",
    "maze_2": "maps/maze_2.png",          # = batch/large_maze.png
    "maze_3": "maps/maze_3.png",
    "maze_4": "maps/maze_4.png",          # = batch/small_maze.png
    "mine_1": "maps/mine_1.png",          # = batch/small_mine.png
    "mine_2": "maps/mine_2.png",          # = batch/large_mine.png
    "trivial": "maps/trivial.png",
    "umass_lpr": "maps/umass_lpr.png",    # other pixels, same grid as umass under the loader rule
    "batch_c_space": "batch/c_space.png",
    "batch_umass": "batch/umass.png",
    "willow_garage": "batch/willow_garage.png",
}
ALIASES = {"umass_lpr": "umass"}          # same (m, u0, locked): solved once
EPSILONS = (1e-2, 1e-3, 1e-6)
SAMPLES = 16384
FULL_FIELDS = {("maze_4", 1e-3)}
PARTS = os.path.join(HERE, "_parts")      # git-ignored scratch


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def sample_idx(cells):
    rng = np.random.default_rng(20240601)
    return np.sort(rng.choice(cells, size=min(SAMPLES, cells), replace=False)).astype(np.int64)


def part_path(name, eps):
    return os.path.join(PARTS, f"{name}_{eps:g}")


def run_job(name, eps):
    ref = O.ref()
    if ref is None:
        sys.exit("reference sources unavailable")
    m, u0, locked = O.load_png_reference_rule(os.path.join(HERE, "maps", name + ".png"))
    p = O.Problem(m, u0, locked, epsilon=eps, stagger=100)
    t0 = time.time()
    rc = ref.harmonic_complete_cpu(ct.byref(p.h))
    dt = time.time() - t0
    u = p.u
    free = p.locked == 0
    idx = sample_idx(u.size)
    entry = dict(rc=int(rc), iterations=int(p.h.currentIteration), delta=float(p.h.delta), sha_u=sha(u),
                 sum=float(u[free].astype(np.float64).sum()), min=float(u[free].min()), max=float(u[free].max()),
                 reached=int((u[free] > -9e5).sum()), seconds=round(dt, 1))
    os.makedirs(PARTS, exist_ok=True)
    arrays = {"samples": u[idx].copy()}
    if (name, eps) in FULL_FIELDS:
        arrays["field"] = u.copy()
    np.savez_compressed(part_path(name, eps) + ".npz", **arrays)
    json.dump(entry, open(part_path(name, eps) + ".json", "w"))
    print(f"{name} eps={eps:g}: {entry['iterations']} half-sweeps, delta {entry['delta']:.3e}, {dt:.0f}s", flush=True)


def jobs_missing():
    out = []
    for name in SOURCES:
        if name in ALIASES:
            continue
        for eps in EPSILONS:
            if not os.path.exists(part_path(name, eps) + ".json"):
                m, _, _ = O.load_png_reference_rule(os.path.join(HERE, "maps", name + ".png"))
                out.append((m[0] * m[1] * (3 if eps < 1e-4 else 1), name, eps))
    return [(n, e) for _, n, e in sorted(out, reverse=True)]       # longest first


def copy_maps():
    for name, rel in SOURCES.items():
        dst = os.path.join(HERE, "maps", name + ".png")
        if not os.path.exists(dst):
            shutil.copyfile(os.path.join(REF_TESTS, rel), dst)
            os.chmod(dst, 0o644)


def merge():
    manifest = dict(generator="tests/golden/generate_map_goldens.py",
                    reference="oracle/_ref/libepic_ref.so = g++ -std=c++11 -O3 of libepic/src/harmonic/*.cpp",
                    stagger=100, samples="numpy default_rng(20240601).choice(cells, 16384, replace=False), sorted",
                    sources={k: "libepic/tests/" + v for k, v in SOURCES.items()}, aliases=ALIASES, maps={})
    arrays = {}
    for name in SOURCES:
        m, u0, locked = O.load_png_reference_rule(os.path.join(HERE, "maps", name + ".png"))
        entry = dict(m=m, sha_u0=sha(u0), sha_locked=sha(locked), free=int((locked == 0).sum()),
                     goals=int((u0 == 0).sum()), runs={})
        if name in ALIASES:
            other = manifest["maps"][ALIASES[name]]
            assert (other["sha_u0"], other["sha_locked"], other["m"]) == (entry["sha_u0"], entry["sha_locked"], m)
            entry["same_grid_as"] = ALIASES[name]
            manifest["maps"][name] = entry
            continue
        arrays[name + "/sample_idx"] = sample_idx(u0.size)
        for eps in EPSILONS:
            pj = part_path(name, eps) + ".json"
            if not os.path.exists(pj):
                print("missing:", name, eps)
                continue
            entry["runs"][f"{eps:g}"] = json.load(open(pj))
            z = np.load(part_path(name, eps) + ".npz")
            arrays[f"{name}/samples_{eps:g}"] = z["samples"]
            if "field" in z.files:
                arrays[f"{name}/field_{eps:g}"] = z["field"]
        manifest["maps"][name] = entry
    np.savez_compressed(os.path.join(HERE, "ref_maps.npz"), **arrays)
    json.dump(manifest, open(os.path.join(HERE, "ref_maps.json"), "w"), indent=1, sort_keys=True)
    print("wrote ref_maps.npz / ref_maps.json")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--jobs", type=int, default=0)
    ap.add_argument("--only")
    ap.add_argument("--merge", action="store_true")
    ap.add_argument("--copy-maps", action="store_true")
    args = ap.parse_args()
    if args.copy_maps:
        copy_maps()
    if args.only:
        name, eps = args.only.split(":")
        run_job(name, float(eps))
    if args.jobs:
        todo = jobs_missing()
        running = []
        while todo or running:
            running = [p for p in running if p.poll() is None]
            while todo and len(running) < args.jobs:
                name, eps = todo.pop(0)
                running.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), "--only", f"{name}:{eps:g}"]))
            time.sleep(2)
    if args.merge:
        merge()


if __name__ == "__main__":
    main()
