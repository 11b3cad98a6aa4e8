"""Kernel time of the small-grid tile path (kernels_tile2d.hip) by halo and tile height: eager launches back to back on the
library's stream between two HIP events (epic_hip_timed_sweeps_gpu), so time / launches = kernel duration as long as a kernel
outlasts the host's launch rate (~4 us).

    python tools/tile_probe.py [--map maze] [--mode default|tol_rb|tol_jacobi|jacobi] [--halo 1,2,4,8,12] [--rows 0,16,22,32] [--steps 0]
--steps k: launches of k iterations each with the given halo (k <= halo): separates the per-launch cost from the per-step cost.
"""
import argparse
import ctypes as ct
import os

os.environ.setdefault("EPIC_HIP_STUDY", "1")   # this tool steers the kernel plan with study knobs (epic_amd/csrc/driver_config.cpp)
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

MODES = {"default": (0, 1), "jacobi": (0, 0), "tol_rb": (4, 1), "tol_jacobi": (4, 0)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--map", default="maze")
    ap.add_argument("--mode", default="default")
    ap.add_argument("--halo", default="1,2,4,8,12")
    ap.add_argument("--rows", default="0")
    ap.add_argument("--width", default="0", help="64 | 128: the LDS tile's width (0: the library's choice)")
    ap.add_argument("--iters", type=int, default=4800)
    ap.add_argument("--develop", type=int, default=3000)
    args = ap.parse_args()
    from epic_amd import epic_harmonic as eh
    from epic_amd.harmonic_map import HarmonicMap

    E = eh._epic
    math, scheme = MODES[args.mode]
    h = HarmonicMap().load(os.path.join(ROOT, "tests/golden/maps", args.map + ".png"))
    h.epsilon = 1e-6
    for fn in (E.harmonic_initialize_dimension_size_gpu, E.harmonic_initialize_potential_values_gpu, E.harmonic_initialize_locked_gpu):
        assert fn(h) == 0
    assert E.harmonic_initialize_gpu(h, 1024) == 0
    assert E.epic_hip_set_math_mode(h, math) == 0 and E.epic_hip_set_scheme(h, scheme) == 0
    ms = ct.c_float(0)
    os.environ["EPIC_HIP_TILE"] = "0"
    assert E.epic_hip_config_reload(h) == 0   # (the library reads its environment once per context)
    assert E.epic_hip_timed_sweeps_gpu(h, args.develop, 0, ct.byref(ms)) == 0   # a developed field (values vary: clocks)
    assert E.epic_hip_timed_sweeps_gpu(h, args.iters, 0, ct.byref(ms)) == 0
    print(f"{args.map} {list(h.shape)} {args.mode}: per-iteration kernels (eager) {ms.value / args.iters * 1e3:.3f} us/iteration", flush=True)
    os.environ["EPIC_HIP_TILE"] = "1"
    if args.width != "0":
        os.environ["EPIC_HIP_TILE_WIDTH"] = args.width
    for halo in [int(x) for x in args.halo.split(",")]:
        for rows in [int(x) for x in args.rows.split(",")]:
            os.environ["EPIC_HIP_TILE_HALO"] = str(halo)
            if rows:
                os.environ["EPIC_HIP_TILE_ROWS"] = str(rows)
            else:
                os.environ.pop("EPIC_HIP_TILE_ROWS", None)
            assert E.epic_hip_config_reload(h) == 0
            if E.epic_hip_tile_iterations(h) != halo:
                print(f"halo {halo} rows {rows}: no plan")
                continue
            iters = args.iters // halo * halo
            assert E.epic_hip_timed_sweeps_gpu(h, iters, 0, ct.byref(ms)) == 0
            assert E.epic_hip_timed_sweeps_gpu(h, iters, 0, ct.byref(ms)) == 0
            print(f"halo {halo:2d} rows {rows:2d}: {ms.value / (iters / halo) * 1e3:7.2f} us/launch  {ms.value / iters * 1e3:.3f} us/iteration", flush=True)


if __name__ == "__main__":
    main()
