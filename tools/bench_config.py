#!/usr/bin/env python3
"""Side measurements for the BASELINE configs that are not bench.py's headline line (1 GPU):
    python tools/bench_config.py --grid 512 512 512 --sweeps 200           # config 5: 3-D 7-point
    python tools/bench_config.py --grid 32768 32768 --sweeps 50            # config 4's grid on one GPU
Prints one JSON line: device time per sweep (HIP events on the library stream), cell-updates/s, algorithmic GB/s."""
import argparse
import ctypes as ct
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", type=int, nargs="+", required=True)
    ap.add_argument("--sweeps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--math", choices=("precise", "tol", "fast", "traffic"), default="precise")
    ap.add_argument("--scheme", choices=("jacobi", "redblack"), default="jacobi")
    ap.add_argument("--develop", type=int, default=0, help="untimed sweeps first, so that the timed ones run on a developed field (the constant initial field flatters a VALU-bound kernel)")
    ap.add_argument("--rows-per-task", type=int, default=0)
    ap.add_argument("--relax", action="store_true", help="also run harmonic_execute_gpu to eps=1e-6")
    args = ap.parse_args()
    import numpy as np
    import torch  # noqa: F401  (one HIP runtime per process)

    from epic_amd import epic_harmonic as eh
    from epic_amd.harmonic import Harmonic
    from epic_amd.synthetic import synthetic_grid

    E = eh._epic
    t0 = time.perf_counter()
    u0, locked = synthetic_grid(args.grid)
    gen_s = time.perf_counter() - t0
    free = int((locked == 0).sum())
    h = Harmonic()
    h.set_grid(args.grid, u0, locked)
    h.epsilon = 1e-6
    t0 = time.perf_counter()
    for fn in (E.harmonic_initialize_dimension_size_gpu, E.harmonic_initialize_potential_values_gpu,
               E.harmonic_initialize_locked_gpu):
        assert fn(h) == 0, fn.__name__
    up_s = time.perf_counter() - t0
    assert E.harmonic_initialize_gpu(h, 1024) == 0
    assert E.epic_hip_set_math_mode(h, {"precise": 0, "fast": 1, "traffic": 2, "tol": 4}[args.math]) == 0
    if args.rows_per_task:
        E.epic_hip_set_rows_per_task(h, args.rows_per_task)
    ms = ct.c_float(0)
    assert E.epic_hip_set_activity_tracking(h, 0) == 0   # timed sweeps recompute every cell; the relaxation below uses the default
    assert E.epic_hip_set_scheme(h, 1 if args.scheme == "redblack" else 0) == 0
    if args.develop:
        assert E.epic_hip_update_n_gpu(h, args.develop, 0) == 0
    assert E.epic_hip_timed_sweeps_gpu(h, args.warmup, 0, ct.byref(ms)) == 0
    assert E.epic_hip_timed_sweeps_gpu(h, args.sweeps, 100, ct.byref(ms)) == 0
    us = ms.value * 1e3 / args.sweeps
    cells = int(np.prod(args.grid))
    out = dict(grid=args.grid, math=args.math, scheme=args.scheme, developed_sweeps=args.develop, sweeps=args.sweeps, us_per_sweep=round(us, 2),
               Mcell_updates_per_s=round(free / us, 1), algorithmic_GBps=round(8.0 * cells / us / 1e3, 1),
               frac_of_8TBps=round(8.0 * cells / us / 1e3 / 8000.0, 4), free_cells=free, cells=cells,
               generate_s=round(gen_s, 2), h2d_s=round(up_s, 3))
    assert E.epic_hip_set_activity_tracking(h, 2) == 0
    if args.relax:
        assert E.harmonic_uninitialize_gpu(h) == 0
        h.u_array().ravel()[:] = u0
        assert E.harmonic_update_model_gpu(h) == 0
        t0 = time.perf_counter()
        assert E.harmonic_execute_gpu(h, 1024) == 0
        out["relax"] = dict(sweeps=int(h.currentIteration), seconds=round(time.perf_counter() - t0, 3),
                            delta=float(h.delta))
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
