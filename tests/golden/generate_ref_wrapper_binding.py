#!/usr/bin/env python3
"""Build container only: the REFERENCE's own, unchanged ctypes wrapper bound to THIS repository's libepic.so.

INTEGRATION.md section 2 claims that /root/reference/libepic/python/epic/epic_harmonic.py keeps working when the library
behind it is epic_amd/lib/libepic.so.  This script shows it: the wrapper's one `ct.CDLL(".../lib/libepic.so")` call
(epic_harmonic.py:38-39) is answered with our library -- nothing else is touched, nothing of the reference is copied --, the
module is imported as it is (every one of its thirty `_epic.<fn>.argtypes = ...` lines has to find its symbol, or the import
raises AttributeError), and the reference's own Harmonic class (harmonic.py:35-107) relaxes a golden grid with
solve(process='cpu') and, without a device, with solve(process='gpu') (its print-and-fall-back-to-the-CPU flow,
harmonic.py:67-99).  Writes tests/golden/ref_wrapper_binding.json (symbol list, result hashes); tests/test_ref_wrapper_binding.py
asserts that fixture against the goldens harmonic_complete_cpu of the reference's C sources produced.

    python tests/golden/generate_ref_wrapper_binding.py          (needs /root/reference; exits 0 with a note where it is absent)
"""
import ctypes as ct
import hashlib
import json
import os
import re
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF_PY = "/root/reference/libepic/python/epic"
OUR_LIB = os.path.join(ROOT, "epic_amd", "lib", "libepic.so")
OUT = os.path.join(ROOT, "tests", "golden", "ref_wrapper_binding.json")


def main():
    if not os.path.isdir(REF_PY):
        print("generate_ref_wrapper_binding: %s is absent (not the build container): nothing to do" % REF_PY)
        return 0
    import numpy as np

    opened = []
    real_cdll = ct.CDLL

    def redirected(path, *args, **kw):
        # the wrapper asks for <its directory>/../../lib/libepic.so: answer with this repository's build of that library
        if os.path.basename(str(path)) == "libepic.so":
            opened.append(str(path))
            return real_cdll(OUR_LIB, *args, **kw)
        return real_cdll(path, *args, **kw)

    ct.CDLL = redirected
    if not hasattr(time, "clock"):
        time.clock = time.process_time          # harmonic.py:80,96 time their solver with time.clock (gone since Python 3.8)
    sys.path.insert(0, REF_PY)                   # harmonic.py imports `epic_harmonic` as a top-level module (harmonic.py:30-31)
    try:
        import epic_harmonic as ref_eh           # the reference's file, unchanged: binds all thirty entry points or raises
        import harmonic as ref_h
    finally:
        ct.CDLL = real_cdll
    assert ref_eh.__file__.startswith(REF_PY) and ref_h.__file__.startswith(REF_PY), (ref_eh.__file__, ref_h.__file__)
    assert len(opened) == 1, opened
    names = re.findall(r"^_epic\.(\w+)\.argtypes", open(ref_eh.__file__).read(), flags=re.M)
    assert len(names) == len(set(names)) == 30, len(names)
    for n in names:                               # bound, and bound to OUR library
        fn = getattr(ref_eh._epic, n)
        assert fn.argtypes is not None, n
    assert ref_eh._epic._name == OUR_LIB

    g = np.load(os.path.join(ROOT, "tests", "golden", "small_grids.npz"))
    manifest = json.load(open(os.path.join(ROOT, "tests", "golden", "manifest.json")))["small"]
    results = {}
    for name in ("g2d_32", "g3d_8"):
        info = manifest[name]
        for process in ("cpu", "gpu"):
            m = np.ascontiguousarray(g[name + "/m"], dtype=np.uint32)
            u = np.ascontiguousarray(g[name + "/u0"], dtype=np.float32).copy()
            locked = np.ascontiguousarray(g[name + "/locked"], dtype=np.uint32).copy()
            h = ref_h.Harmonic()                  # the reference's class, its defaults (harmonic.py:38-52)
            h.n = len(m)
            h.m = m.ctypes.data_as(ct.POINTER(ct.c_uint))
            h.u = u.ctypes.data_as(ct.POINTER(ct.c_float))
            h.locked = locked.ctypes.data_as(ct.POINTER(ct.c_uint))
            h.numIterationsToStaggerCheck = info["stagger"]
            # process='gpu' without a device: the three initialize calls fail (rc 4 each), harmonic_complete_gpu fails, the
            # wrapper prints its messages and runs harmonic_complete_cpu instead (harmonic.py:73-99) -- same result
            timing = h.solve(process=process, epsilon=info["epsilon"])
            assert timing is not None
            want = np.asarray(g[name + "/converged"], dtype=np.float32).ravel()
            results["%s %s" % (name, process)] = {
                "iterations": int(h.currentIteration), "delta": float(h.delta),
                "sha256_u": hashlib.sha256(u.tobytes()).hexdigest(),
                "equals_reference_golden": bool(np.array_equal(u, want)),
                "golden_iterations": info["iterations"], "golden_delta": info["delta"],
                "device_pointers_null_afterwards": not (h.d_m or h.d_u or h.d_locked or h.d_delta)}
    import torch

    out = {
        "generator": "tests/golden/generate_ref_wrapper_binding.py",
        "wrapper": {"epic_harmonic": ref_eh.__file__, "harmonic": ref_h.__file__,
                    "sha256_epic_harmonic": hashlib.sha256(open(ref_eh.__file__, "rb").read()).hexdigest(),
                    "asked_for": opened[0], "answered_with": os.path.relpath(OUR_LIB, ROOT)},
        "symbols_bound": sorted(names), "n_symbols": len(names),
        "gpu_visible": bool(torch.cuda.is_available()),
        "results": results,
    }
    with open(OUT, "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
        fh.write("\n")
    print("wrote", os.path.relpath(OUT, ROOT))
    print(json.dumps(results, indent=1))
    return 0


if __name__ == "__main__":
    sys.exit(main())
