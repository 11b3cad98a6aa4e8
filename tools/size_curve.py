"""us per iteration over grid size (0.06-16 Mcell), per kernel family: where the planner's thresholds belong (round 6, VERDICT r05 item 5:
grids of 3-4 Mcell fell between the LDS tiles, <= 3 Mcell, and the fused passes, >= 4 Mcell, onto per-iteration kernels).

For every size: the library's choice (no knobs) and each family forced -- tiles (EPIC_HIP_TILE_MAX_CELLS huge), fused pairs
(EPIC_HIP_FUSE_MIN_CELLS=0, EPIC_HIP_TILE=0), single sweeps (EPIC_HIP_TILE=0, EPIC_HIP_NO_FUSE=1) -- for the library's default
arithmetic (precise red-black) and the benchmarked one (tol Jacobi), work lists off (the kernel's own rate) and on steps of 100
iterations with one check (what harmonic_execute_gpu's loop runs), on a developed field.

    python tools/size_curve.py [--quick]      -> a table on stdout (profiles/r06_size_curve.txt)
"""
import ctypes as ct
import os

os.environ.setdefault("EPIC_HIP_STUDY", "1")   # this tool steers the kernel plan with study knobs (epic_amd/csrc/driver_config.cpp)
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from epic_amd import epic_harmonic as eh
from epic_amd.harmonic import Harmonic
from epic_amd.synthetic import synthetic_grid

E = eh._epic
SIZES = [(256, 256), (512, 512), (1024, 1024), (1448, 1448), (1419, 1735), (1900, 1900), (2048, 2048), (2304, 2304), (3072, 3072), (4096, 4096)]
FAMILIES = {
    "library": {},
    "tiles": {"EPIC_HIP_TILE_MAX_CELLS": str(1 << 40), "EPIC_HIP_FUSE_MIN_CELLS": str(1 << 40)},
    "fused": {"EPIC_HIP_FUSE_MIN_CELLS": "0", "EPIC_HIP_TILE": "0"},
    "single": {"EPIC_HIP_TILE": "0", "EPIC_HIP_NO_FUSE": "1"},
}
MODES = [("precise redblack", eh.MATH_PRECISE, eh.SCHEME_REDBLACK), ("tol jacobi", eh.MATH_TOL, eh.SCHEME_JACOBI),
         ("tol redblack", eh.MATH_TOL, eh.SCHEME_REDBLACK), ("precise jacobi", eh.MATH_PRECISE, eh.SCHEME_JACOBI)]
KNOBS = sorted({k for f in FAMILIES.values() for k in f})


def measure(m, math, scheme, env):
    for k in KNOBS:
        os.environ.pop(k, None)
    os.environ.update(env)
    os.environ["EPIC_HIP_TRACK"] = "0"
    try:
        u0, locked = synthetic_grid(list(m))
        h = Harmonic()
        h.set_grid(list(m), u0, locked)
        h.epsilon, h.numIterationsToStaggerCheck = 1e-6, 100
        for fn in (E.harmonic_initialize_dimension_size_gpu, E.harmonic_initialize_potential_values_gpu, E.harmonic_initialize_locked_gpu):
            assert fn(h) == 0
        assert E.harmonic_initialize_gpu(h, 1024) == 0
        assert E.epic_hip_set_math_mode(h, math) == 0 and E.epic_hip_set_scheme(h, scheme) == 0
        dev = (max(m) * 3) // 100 * 100      # the front has crossed the grid
        assert E.epic_hip_update_n_gpu(h, dev, 0) == 0
        import time

        best = None
        for rep in range(4):     # four blocks of harmonic_execute_gpu's loop (99 plain iterations + the check, one read-back each); best of the last three
            t0 = time.perf_counter()
            for _ in range(4):
                assert E.epic_hip_update_n_gpu(h, 100, 1) in (0, 1)
            dt = (time.perf_counter() - t0) * 1e3
            if rep:
                best = dt if best is None else min(best, dt)
        path = (eh.config_dump(h) or {}).get("path", {}).get("plain_batch", "?")
        for fn in (E.harmonic_uninitialize_gpu, E.harmonic_uninitialize_dimension_size_gpu, E.harmonic_uninitialize_potential_values_gpu,
                   E.harmonic_uninitialize_locked_gpu):
            fn(h)
        return best * 1e3 / 400, path
    finally:
        for k in KNOBS + ["EPIC_HIP_TRACK"]:
            os.environ.pop(k, None)


def main():
    quick = "--quick" in sys.argv
    for name, math, scheme in MODES:
        print("== %s: us per iteration (blocks of 99 plain iterations + 1 check through epic_hip_update_n_gpu = one block of harmonic_execute_gpu's loop; wall time)" % name)
        print("%-12s %8s | %9s %9s %9s %9s | %s" % ("grid", "Mcell", "library", "tiles", "fused", "single", "library's path"))
        for m in SIZES[::2] if quick else SIZES:
            row, path = {}, "?"
            for fam, env in FAMILIES.items():
                try:
                    us, p = measure(m, math, scheme, env)
                    row[fam] = us
                    if fam == "library":
                        path = p
                except AssertionError:
                    row[fam] = float("nan")
            print("%-12s %8.2f | %9.2f %9.2f %9.2f %9.2f | %s" % ("%dx%d" % m, m[0] * m[1] / 1e6, row["library"], row["tiles"], row["fused"],
                                                                 row["single"], path), flush=True)


if __name__ == "__main__":
    main()
