#!/usr/bin/env python3
"""bench.py -- headline benchmark: log-space harmonic relaxation throughput on MI355X.

Metric (BASELINE.json): Mcell-updates/s + % of HBM peak on the synthetic 8192 x 8192 grid (5 % random obstacles,
one goal, seed 20240601), relaxing towards epsilon = 1e-6.

A *step* is one pass of the reference's driver loop over `stagger` = 100 iterations (harmonic_gpu.cu:266-290):
one check sweep (max |du| reduced on the device, read back) followed by 99 plain Jacobi sweeps, on u resident in
HBM.  A cell-update is one unlocked cell recomputed once.  With N > 1 (one process per GPU, launched by
torch.distributed.run) every rank owns an 8192-row slab of an (8192 N) x 8192 grid (weak scaling) and exchanges
one halo row with each neighbour per sweep over RCCL (epic_amd/slab.py).

Prints ONE JSON line on rank 0.  Extra objects:
  roofline      dominant kernel (sweep2d) vs the HBM roofline: 8 algorithmic bytes per grid cell per sweep
                (read u once, write u once) / mean launch-to-launch device time, measured with HIP
                events on the stream the kernels run on (epic_hip_timed_sweeps_gpu).
  cpu_baseline  the reference's own harmonic_cpu.cpp compiled by oracle/Makefile (oracle/_ref/libepic_ref.so, kind
                "reference") or, when that did not travel with the repo, its C restatement (oracle/liboracle.so, kind
                "port"); 1 thread -- the reference is single-threaded -- timed on this host for a bounded number of
                red-black half-sweeps of the same grid.  Rank 0, N = 1 only.  Its `all_cores` entry is the build's own
                OpenMP form of the same half-sweep on every host core (not the reference, which has no threading).
  relax*        (N = 1) the complete relaxation to epsilon = 1e-6 through harmonic_execute_gpu: iterations, seconds --
                Jacobi and red-black with the library defaults (activity tracking on), Jacobi also with tracking off.
The timed region itself runs with activity tracking OFF: every sweep recomputes every unlocked cell.
"""
import argparse
import ctypes as ct
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 TB/s achievable)
BYTES_PER_CELL_SWEEP = 8.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--size", type=int, default=8192, help="grid is size x size per GPU")
    ap.add_argument("--stagger", type=int, default=100, help="sweeps per step (numIterationsToStaggerCheck)")
    ap.add_argument("--rows-per-task", type=int, default=0)
    ap.add_argument("--cpu-half-sweeps", type=int, default=40, help="bounded CPU sample (about 0.35 s each at 8192^2)")
    ap.add_argument("--math", choices=("precise", "tol", "df32", "fast", "traffic"), default="precise",
                    help="precise = libm-equivalent exp/log (bit-exact parity mode, default); tol = one exp-class split per "
                         "cell shared by its neighbours (tolerance parity mode); fast = v_exp_f32/v_log_f32")
    ap.add_argument("--scheme", choices=("jacobi", "redblack"), default="jacobi",
                    help="jacobi = ping-pong sweep of every cell (default); redblack = the reference's in-place half-sweeps")
    ap.add_argument("--halo", type=int, default=8, help="N > 1: ghost rows per side = sweeps between two halo exchanges")
    ap.add_argument("--slab", action="store_true", help="use the slab-decomposition driver even on one GPU")
    ap.add_argument("--strong", action="store_true",
                    help="N > 1: ONE size x size grid cut into N row slabs (BASELINE configs[3]: --size 32768 --gpus 4|8) "
                         "instead of the default weak scaling (one size x size grid per GPU)")
    ap.add_argument("--develop", type=int, default=20000,
                    help="untimed sweeps before the timed region, so that it runs on a developed field: on the constant "
                         "initial field (u = -1e6 almost everywhere) the same VALU-bound kernel runs ~15 %% faster "
                         "(measured 152 vs 181 us per sweep; the traffic-only build is unaffected), which would flatter it")
    ap.add_argument("--track", action="store_true",
                    help="leave activity tracking (skipping of tiles whose inputs did not change) ON in the timed region; "
                         "by default it is OFF there, so that every sweep recomputes every unlocked cell as the metric "
                         "counts them -- the `relax` legs always run with the library default (ON)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-relax", action="store_true")
    return ap.parse_args()


def cpu_baseline(m, u0, locked, half_sweeps, free_by_colour):
    """Checker leg: the reference's red-black half-sweeps on one host thread, bounded sample.  When the compiled reference
    travelled with the repo (oracle/_ref/libepic_ref.so, built by oracle/Makefile from the reference's own
    harmonic_cpu.cpp) that is what is timed (kind "reference"); otherwise the C restatement in oracle/ (kind "port")."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _oracle as O

    p = O.Problem(m, u0, locked, 1e-6, 100)
    ref = O.ref()
    if ref is not None:
        kind, plain, check = "reference", ref.harmonic_update_cpu, ref.harmonic_update_and_check_cpu
    else:
        lib = O.oracle()
        kind, plain, check = "port", lib.oracle_update, lib.oracle_update_and_check
    t0 = time.perf_counter()
    for i in range(half_sweeps):
        (check if i % 100 == 0 else plain)(ct.byref(p.h))
    dt = time.perf_counter() - t0
    # iteration i recomputes the unlocked cells with (row + col + i) odd (harmonic_cpu.cpp:46-51)
    updates = sum(free_by_colour[i % 2] for i in range(half_sweeps))
    out = dict(value=round(updates / dt / 1e6, 3), unit="Mcell-updates/s", cores=1, host_cores=os.cpu_count(),
               kind=kind, seconds=round(dt, 2),
               sample="%d red-black half-sweeps (harmonic_update_cpu, one of them with the convergence check) of the "
                      "same %dx%d grid (full relaxation needs ~5e4, ~15 h on one core)" % (half_sweeps, m[0], m[1]))
    # Not the reference (which has no threading): the same half-sweep with its rows dealt to every host core by OpenMP
    # (oracle_update_parallel_2d, bit-identical to the sequential one) -- the best this host can do with that algorithm.
    lib = O.oracle()
    if hasattr(lib, "oracle_update_parallel_2d"):
        lib.oracle_update_parallel_2d.argtypes = (ct.c_void_p, ct.c_int)
        lib.oracle_update_parallel_2d.restype = ct.c_int
        allowed = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        best = None
        for threads in sorted({allowed, 128, 64, 32, 16, 8} & set(range(1, allowed + 1)), reverse=True):
            t_warm = time.perf_counter()                                # thread start-up, sleeping cores, page placement
            while time.perf_counter() - t_warm < 0.3:
                lib.oracle_update_parallel_2d(ct.byref(p.h), threads)
            first = int(p.h.currentIteration)
            n = max(4, half_sweeps // 4)
            t0 = time.perf_counter()
            for _ in range(n):
                lib.oracle_update_parallel_2d(ct.byref(p.h), threads)
            dt = time.perf_counter() - t0
            rate = sum(free_by_colour[(first + i) % 2] for i in range(n)) / dt / 1e6
            if best is None or rate > best["value"]:
                best = dict(value=round(rate, 1), unit="Mcell-updates/s", cores=threads, kind="port+openmp",
                            seconds=round(dt, 2), half_sweeps=n)
        if best:
            best["allowed_cores"] = allowed
            best["note"] = "best of a few thread counts; the host may cap CPU time below its core count"
            out["all_cores"] = best
    return out


def measured_traffic(n, math, scheme):
    """HBM bytes per launch from the PMC counters (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, gfx950
    corrections applied by tools/summarize_profile.py) of this same command; recorded under profiles/ because bench.py
    cannot run under the profiler by itself.  None when no measurement of this configuration is on file."""
    path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    try:
        return json.load(open(path)).get("%d_%s_%s" % (n, math, scheme))
    except (OSError, ValueError):
        return None


def main():
    args = parse()
    import numpy as np
    import torch  # first: one HIP runtime per process (epic_amd/epic_harmonic.py)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d"
                     % (args.gpus, args.gpus))
        args.gpus = world
    ndev = torch.cuda.device_count()
    backend = os.environ.get("EPIC_BENCH_BACKEND", "nccl")   # "gloo": ranks may share a GPU (1-GPU smoke of the N > 1 path)
    if ndev < 1:
        sys.exit("bench.py: no GPU visible")
    if backend == "nccl" and local >= ndev:
        sys.exit("bench.py: rank %d has no GPU of its own (%d visible)" % (local, ndev))
    local = local % ndev
    torch.cuda.set_device(local)
    red_dev = "cuda" if backend == "nccl" else "cpu"          # where the few scalar reductions of this script live
    if world > 1:
        import torch.distributed as dist

        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)

    from epic_amd import epic_harmonic as eh
    from epic_amd.synthetic import synthetic_grid

    E = eh._epic
    n = args.size
    grid = [n, n] if args.strong else [n * world, n]

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    use_abi = world == 1 and not args.slab
    if use_abi:
        from epic_amd.harmonic import Harmonic

        u0, locked = synthetic_grid(grid)
        free_cells = int((locked == 0).sum())
        lk2 = locked.reshape(grid) == 0
        rr, cc = np.indices(lk2.shape, sparse=True)
        free_by_colour = [int((lk2 & (((rr + cc) & 1) == 1)).sum()), int((lk2 & (((rr + cc) & 1) == 0)).sum())]
        del lk2
        h = Harmonic()
        h.set_grid(grid, u0, locked)
        h.epsilon = 1e-6
        h.numIterationsToStaggerCheck = args.stagger
        t0 = time.perf_counter()
        for fn in (E.harmonic_initialize_dimension_size_gpu, E.harmonic_initialize_potential_values_gpu,
                   E.harmonic_initialize_locked_gpu):
            if fn(h) != 0:
                sys.exit("bench.py: %s failed -- no usable GPU" % fn.__name__)
        upload_s = time.perf_counter() - t0
        assert E.harmonic_initialize_gpu(h, 1024) == 0
        if args.rows_per_task:
            E.epic_hip_set_rows_per_task(h, args.rows_per_task)
        assert E.epic_hip_set_math_mode(h, {"precise": 0, "fast": 1, "traffic": 2, "df32": 3, "tol": 4}[args.math]) == 0
        assert E.epic_hip_set_scheme(h, 1 if args.scheme == "redblack" else 0) == 0
        assert E.epic_hip_set_activity_tracking(h, 1 if args.track else 0) == 0
        ms = ct.c_float(0.0)

        def step():
            rc = E.epic_hip_timed_sweeps_gpu(h, args.stagger, args.stagger, ct.byref(ms))
            assert rc == 0, rc
            return ms.value

        solver = None
    else:
        from epic_amd.slab import SlabSolver

        solver = SlabSolver(grid, rank, world, device=torch.device("cuda", local), stagger=args.stagger,
                            rows_per_task=args.rows_per_task, math=args.math, halo=args.halo)
        free_cells = solver.load_synthetic()
        upload_s = None

        def step():
            return solver.timed_step()

    # let the wavefront from the goal cover the grid first (untimed); a multiple of stagger keeps the check cadence
    develop = max(0, args.develop) // args.stagger * args.stagger
    if develop:
        if use_abi:
            assert E.epic_hip_update_n_gpu(h, develop, 0) == 0
        else:
            for _ in range(develop):
                solver.sweep(False)
    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    dev_ms = 0.0
    for _ in range(args.steps):
        dev_ms += step()
    barrier()
    wall = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([wall, dev_ms], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall, dev_ms = float(t[0]), float(t[1])
        f = torch.tensor([free_cells], dtype=torch.int64, device=red_dev)
        dist.all_reduce(f)
        free_cells = int(f[0])

    sweeps = args.steps * args.stagger

    def updates_in(iterations, first=0):
        """Unlocked cells recomputed by `iterations` iterations: all of them per Jacobi sweep; per red-black half-sweep
        the cells with (row + col + iteration) odd (harmonic_cpu.cpp:46-51)."""
        if args.scheme == "jacobi" or not use_abi:
            return free_cells * iterations
        even_it = (iterations + (1 - first % 2)) // 2      # iterations with even index update (row + col) odd
        return even_it * free_by_colour[0] + (iterations - even_it) * free_by_colour[1]

    value = updates_in(sweeps, develop + args.warmup * args.stagger) / wall / 1e6
    launch_us = dev_ms * 1e3 / sweeps
    # algorithmic bytes per launch: 8 B per cell the launch recomputes-or-copies.  A Jacobi sweep touches every cell of
    # the grid; a red-black half-sweep recomputes one colour, i.e. half the grid (its row-major in-place layout still
    # moves both colours -- that surplus shows up in `traffic`, not in `achieved`).
    rows_per_rank = grid[0] // world   # the dominant kernel is one rank's sweep of its slab
    cells_per_launch = rows_per_rank * n if (args.scheme == "jacobi" or not use_abi) else rows_per_rank * n // 2
    achieved = BYTES_PER_CELL_SWEEP * cells_per_launch / (launch_us * 1e-6) / 1e9
    out = {
        "metric": "cell_updates_per_s_log_harmonic_relax_8192sq",
        "value": round(value, 1),
        "unit": "Mcell-updates/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(wall * 1e3 / args.steps, 4),
        "higher_is_better": True,
        "scaling": "strong" if (args.strong and world > 1) else "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": ("synthetic %dx%d occupancy grid cut into %d row slabs, 5%% random obstacles + 1 goal (BASELINE "
                         "configs[3]), log-space Jacobi relax towards eps=1e-6" % (n, n, world)) if (args.strong and world > 1)
                        else ("synthetic %dx%d occupancy grid per GPU, 5%% random obstacles + 1 goal (BASELINE configs[2]), "
                              "log-space Jacobi relax towards eps=1e-6" % (n, n)),
            "grid": grid,
            "sweeps_per_step": args.stagger,
            "check_every": args.stagger,
            "developed_sweeps": develop,
            "math": args.math,
            "scheme": args.scheme if use_abi else "jacobi",
            "activity_tracking": bool(args.track) if use_abi else False,
            "free_cells": free_cells,
            "parallelism": "1 GPU" if world == 1 else "row slabs x%d, %d halo rows exchanged every %d sweeps over RCCL" % (world, args.halo, args.halo),
        },
        "roofline": {
            "bound": "hbm",
            "kernel": "sweep2d_kernel",
            "achieved": round(achieved, 1),
            "peak": HBM_PEAK_GBPS,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBPS, 4),
            "traffic": measured_traffic(n, args.math, args.scheme),
            "launch_us": round(launch_us, 3),
            "bytes_per_launch": int(BYTES_PER_CELL_SWEEP * cells_per_launch),
            "note": "8 B x grid cells per launch / mean launch-to-launch device time (HIP events on the kernel's stream)",
        },
    }

    if use_abi:
        if upload_s is not None:
            out["config"]["h2d_seconds"] = round(upload_s, 3)
        assert E.harmonic_uninitialize_gpu(h) == 0
        if not args.no_relax:
            # the complete relaxation, exactly as the plugin runs it (harmonic_execute_gpu), from the initial state;
            # once with the benchmarked scheme
            # (library default: activity tracking on) and once with the other one; the benchmarked scheme also with
            # tracking off, i.e. every sweep recomputing every cell as in the timed region above
            other = "redblack" if args.scheme == "jacobi" else "jacobi"
            for scheme, track in ((args.scheme, 1), (other, 1), (args.scheme, 0)):
                h.u_array().ravel()[:] = u0
                assert E.harmonic_update_model_gpu(h) == 0
                assert E.epic_hip_set_scheme(h, 1 if scheme == "redblack" else 0) == 0
                assert E.epic_hip_set_activity_tracking(h, track) == 0
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                rc = E.harmonic_execute_gpu(h, 1024)
                dt = time.perf_counter() - t0
                assert rc == 0, rc
                its = int(h.currentIteration)
                keep = args.scheme
                args.scheme = scheme
                upd = updates_in(its)
                args.scheme = keep
                key = ("relax" if scheme == args.scheme else "relax_" + scheme) + ("" if track else "_untracked")
                out[key] = {
                    "scheme": scheme, "activity_tracking": bool(track), "epsilon": 1e-6, "iterations": its,
                    "seconds": round(dt, 3), "delta": float(h.delta),
                    "Mcell_updates_per_s": round(upd / dt / 1e6, 1),
                    "note": "harmonic_execute_gpu from the initial state, includes the final D2H of u; cell-updates "
                            "counted as iterations x unlocked cells of the colour"
                            + (" (tiles skipped by tracking count as updated: their values are what the update would "
                               "have produced)" if track else "")
                            + ("; red-black = the reference's scheme, result bit-identical to harmonic_complete_cpu"
                               if scheme == "redblack" and args.math == "precise" else "")}
        for fn in (E.harmonic_uninitialize_dimension_size_gpu, E.harmonic_uninitialize_potential_values_gpu,
                   E.harmonic_uninitialize_locked_gpu):
            fn(h)
        if not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(grid, u0, locked, args.cpu_half_sweeps, free_by_colour)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
