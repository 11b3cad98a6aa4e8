#!/usr/bin/env python3
"""bench.py -- headline benchmark: log-space harmonic relaxation throughput on MI355X.

Metric (BASELINE.json): Mcell-updates/s + % of HBM peak on the synthetic 8192 x 8192 grid (5 % random obstacles,
one goal, seed 20240601), relaxing towards epsilon = 1e-6.

A *step* is one pass of the reference's driver loop over `stagger` = 100 iterations (harmonic_gpu.cu:266-290):
one check sweep (max |du| reduced on the device, read back) followed by 99 plain Jacobi sweeps, on u resident in
HBM.  A cell-update is one unlocked cell recomputed once.  With N > 1 (one process per GPU, launched by
torch.distributed.run) the SAME 8192 x 8192 grid is cut into N row slabs (strong scaling: what the metric names),
halo rows traded over RCCL (epic_amd/slab.py); the line then also carries a `weak` object (one 8192 x 8192 grid per GPU),
`ranks` (what every rank saw: device, backend) and `in_library` (the same grid through EPIC_HIP_DEVICES: one process -- a child of rank 0, run first --,
all GPUs, hipMemcpyPeerAsync halos -- include/epic_hip.h).
The arithmetic is the library's `tol` mode by default (--math): one exp-class split per cell shared by its neighbours,
the reference's rounding stages kept.  It is a TOLERANCE mode, and the line says how far it is from the reference: the
`parity` object holds, per BASELINE config, the measured distance of the converged field from the reference's (fields the
reference itself converged, tests/golden/; at 8192^2 and 512^3 the library's reference-identical default mode, relaxed in this
same run) next to the 1e-5 bar.  A tol relaxation finishes with the reference's own iteration (from the first check with
delta < 10 eps on; harmonic_execute_gpu, "Finish"): with that every config is within the bar; what the tol iteration ALONE
does is recorded beside it (`tol_iteration_alone`: umass.png 1.6e-5, outside the bar) and timed (`relax_tol_alone`).  The bit-exact `precise` mode
is timed beside it (`kernels.precise`), and `relax_default` is the whole relaxation as the unchanged ROS plugin gets it
(no environment: precise + red-black, bit-identical to harmonic_complete_cpu).

Prints ONE JSON line on rank 0.  Extra objects:
  roofline      dominant kernel (sweep2d) vs the HBM roofline: 8 algorithmic bytes per grid cell per sweep
                (read u once, write u once) / mean launch-to-launch device time, measured with HIP
                events on the stream the kernels run on (epic_hip_timed_sweeps_gpu).  `traffic` = HBM bytes per launch
                from the PMC counters, measured in the run itself at N = 1: before this process touches the GPU it runs
                itself twice as a child under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (one short step each);
                `traffic_source` says so, or names the recorded fallback (profiles/hbm_traffic.json).
  cpu_baseline  the reference's own harmonic_cpu.cpp compiled by oracle/Makefile (oracle/_ref/libepic_ref.so, kind
                "reference") or, when that did not travel with the repo, its C restatement (oracle/liboracle.so, kind
                "port"); 1 thread -- the reference is single-threaded -- timed on this host for a bounded number of
                red-black half-sweeps of the same grid.  Rank 0, N = 1 only.  Its `all_cores` entry is the build's own
                OpenMP form of the same half-sweep on every host core (not the reference, which has no threading).
  relax*        (N = 1) the complete relaxation to epsilon = 1e-6 through harmonic_execute_gpu: iterations, seconds --
                Jacobi and red-black with activity tracking on (the library's automatic mode), Jacobi also with tracking off,
                and `relax_default` (precise + red-black: the library with no environment).  Two rates each: `recomputed_...`
                counts the cells the kernels actually recomputed (epic_hip_work_done), `effective_...` counts every unlocked
                cell once per iteration whether its tile ran or was skipped as unchanged.
  parity        see above.
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
    ap.add_argument("--math", choices=("precise", "tol", "fast", "traffic"), default="tol",
                    help="tol (default HERE; the library's own default is precise) = one exp-class split per cell shared by its "
                         "neighbours: a tolerance mode, see the `parity` object of the line; precise = expf/logf bit-identical to "
                         "glibc's (the bit-exact mode); fast = v_exp_f32/v_log_f32, no parity claim")
    ap.add_argument("--scheme", choices=("jacobi", "redblack"), default="jacobi",
                    help="jacobi = ping-pong sweep of every cell (default HERE: what BASELINE.json's metric names); redblack = the "
                         "reference's in-place half-sweeps (the library's own default)")
    ap.add_argument("--halo", type=int, default=0,
                    help="N > 1: ghost rows per side = sweeps between two halo exchanges (0 = by slab height: 8 from 4096 rows per "
                         "GPU up, 16 from 2048, 32 below -- an exchange costs a fixed few tens of microseconds, a sweep of a short "
                         "slab only ~15, and 2 x halo extra rows per slab are cheap)")
    ap.add_argument("--slab", action="store_true", help="use the slab-decomposition driver even on one GPU")
    ap.add_argument("--strong", action="store_true", help="(default for N > 1; kept for older command lines)")
    ap.add_argument("--weak", action="store_true",
                    help="N > 1: make the weak-scaling run (one size x size grid per GPU) the headline value instead of the "
                         "strong one (ONE size x size grid cut into N row slabs, the metric's workload)")
    ap.add_argument("--no-extra-legs", action="store_true", help="N > 1: skip the weak / in-library legs; N = 1: skip the precise kernel leg")
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
    ap.add_argument("--in-library-child", type=int, default=0,
                    help="(internal) run only the in-library multi-device leg on devices 0..N-1 and print its JSON object: the N > 1 "
                         "run starts it as a child process, so that a fault in that never-yet-run path cannot take the headline line with it")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="do not measure roofline.traffic in this run (two rocprofv3 --pmc child passes of this command before "
                         "the timed part, N = 1); use the value recorded in profiles/hbm_traffic.json")
    ap.add_argument("--no-parity", action="store_true", help="skip the parity object (maps, 512^2 / 1024^2, 512^3 relaxations)")
    ap.add_argument("--no-config4", action="store_true", help="skip the 32768^2 leg (BASELINE configs[3]'s grid on one GPU: ~1 minute of host-side grid generation)")
    ap.add_argument("--no-maps", action="store_true", help="skip the timing of the reference's maps (BASELINE configs[0] / [1])")
    ap.add_argument("--no-node-flow", action="store_true", help="skip the navigation node's call sequence (1 check + 49 / 99 single updates per tick) on maze, umass and the timed grid")
    ap.add_argument("--only-config5", action="store_true", help="(internal) run only the 512^3 leg and print its object: the child of the PMC passes")
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


def parity_object(args, E, MODES, relaxed, locked_8192):
    """How far the converged fields of the TIMED arithmetic and scheme are from the reference's, per BASELINE config, measured
    in this run: the relaxation runs here on the device through harmonic_complete_gpu / harmonic_execute_gpu, the yard-stick
    is data -- fields harmonic_complete_cpu itself converged, committed under tests/golden/ (maps; the benchmark's grid family
    at 512^2 and 1024^2) -- or, where the reference needs hours (8192^2, 512^3), the library's default mode (precise +
    red-black), which the GPU tests show to be the reference's iteration bit for bit.  bar = 1e-5 max(1, |u|), BASELINE.json."""
    import numpy as np

    from epic_amd.harmonic import Harmonic
    from epic_amd.harmonic_map import HarmonicMap
    from epic_amd.synthetic import synthetic_grid

    bar = 1e-5
    gdir = os.path.join(ROOT, "tests", "golden")

    def dist(got, want, locked):
        got, want, locked = np.ravel(got), np.ravel(want), np.ravel(locked)
        reached = (locked == 0) & (want > -9e5)
        d = np.abs(got[reached].astype(np.float64) - want[reached])
        rel = float((d / np.maximum(1.0, np.abs(want[reached]))).max())
        return {"max_rel": float("%.3e" % rel), "max_abs": float("%.3e" % float(d.max())), "bar": bar, "within_bar": bool(rel <= bar),
                "others_equal": bool(np.array_equal(got[~reached], want[~reached]))}

    def complete(h, math, scheme, eps=1e-6):
        h.epsilon = eps
        h.numIterationsToStaggerCheck = 100
        prev = {k: os.environ.get(k) for k in ("EPIC_HIP_MATH", "EPIC_HIP_SCHEME")}   # (a user may have exported them for the run)
        os.environ["EPIC_HIP_MATH"], os.environ["EPIC_HIP_SCHEME"] = math, scheme
        try:
            rc = E.harmonic_complete_gpu(h, 1024)
        finally:
            for k, v in prev.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        if rc != 0:
            raise RuntimeError("harmonic_complete_gpu returned %d" % rc)
        return h.u_array().ravel().copy(), int(h.currentIteration)

    out = {"mode": {"math": args.math, "scheme": args.scheme}, "bar": "|du| <= 1e-5 max(1, |u|) over reached free cells",
           "finish": ("tol relaxations leave the tol arithmetic at the first check with delta < 10 eps (100 eps for eps <= 1e-5) and finish with the reference's own "
                      "iteration (precise red-black); at eps > 1e-5 that check keeps its own verdict (round 6: tests/tol_campaign.py, 720 generated cases against the "
                      "reference -- config.tol_campaign_*): the library's default for harmonic_execute_gpu / harmonic_complete_gpu; "
                      "`tol_iteration_alone` repeats a config with EPIC_HIP_TOL_FINISH=0") if args.math == "tol" else None,
           "configs": {}}
    try:
        maps = np.load(os.path.join(gdir, "maps_converged.npz"))
        for cfg, name in (("configs[0] maze.yaml", "maze"), ("configs[1] umass.yaml", "umass"), ("tests/maps basic.png", "basic")):
            h = HarmonicMap().load(os.path.join(gdir, "maps", name + ".png"))
            got, its = complete(h, args.math, args.scheme)
            e = dist(got, maps[name + "/converged_1e-06"], h.locked_array())
            e.update(iterations=its, against="harmonic_complete_cpu's converged field (tests/golden/maps_converged.npz)")
            if args.math == "tol" and os.environ.get("EPIC_HIP_TOL_FINISH") is None:
                os.environ["EPIC_HIP_TOL_FINISH"] = "0"     # the tol iteration to the end, for the record
                try:
                    h = HarmonicMap().load(os.path.join(gdir, "maps", name + ".png"))
                    got0, its0 = complete(h, args.math, args.scheme)
                    e0 = dist(got0, maps[name + "/converged_1e-06"], h.locked_array())
                    e["tol_iteration_alone"] = {"max_rel": e0["max_rel"], "max_abs": e0["max_abs"], "within_bar": e0["within_bar"],
                                                "iterations": its0}
                finally:
                    del os.environ["EPIC_HIP_TOL_FINISH"]
            out["configs"][cfg] = e
        # ... and at the epsilon the reference's callers use (1e-3: src/epic_nav_core_plugin.cpp:61,85, the node, maps.py:67), where the
        # loop stops while the field still moves and the stop iteration decides the field: the timed mode against the samples
        # harmonic_complete_cpu left at 1e-3 (tests/golden/ref_maps.npz, 16 384 cells per map), and its iteration count
        try:
            rm = np.load(os.path.join(gdir, "ref_maps.npz"))
            rj = json.load(open(os.path.join(gdir, "ref_maps.json")))["maps"]
            for name in ("maze", "umass", "basic", "maze_4"):
                run = rj.get(name, {}).get("runs", {}).get("0.001")
                if run is None or name + "/samples_0.001" not in rm.files:
                    continue
                h = HarmonicMap().load(os.path.join(gdir, "maps", name + ".png"))
                got, its = complete(h, args.math, args.scheme, 1e-3)
                idx, want = rm[name + "/sample_idx"], rm[name + "/samples_0.001"]
                e = dist(got[idx], want, h.locked_array().ravel()[idx])
                e.update(iterations=its, reference_iterations=run["iterations"], same_iterations=bool(its == run["iterations"]),
                         against="harmonic_complete_cpu at eps = 1e-3, 16 384 sampled cells (tests/golden/ref_maps.npz)")
                if its != run["iterations"]:
                    e["within_bar"] = False
                out["configs"]["%s.png at the callers' eps = 1e-3" % name] = e
            # ... and the one map of the reference on which an arithmetic that is not bit-identical cannot promise the bar (tol only):
            # trivial.png at 1e-6, with the rule's hand-over and with round 3's factor -- both converged by the reference's test
            if args.math == "tol" and "trivial/samples_1e-06" in rm.files and "EPIC_HIP_TOL_FINISH_FACTOR" not in os.environ:
                tr = {"note": "maps/trivial.png (not a BASELINE config), eps = 1e-6: delta crosses eps in single ulps over tens of thousands of "
                              "iterations, so where the loop stops -- and with it the field, by ~1e-3 -- is decided by single ulps of single "
                              "cells; which hand-over factor ends inside the bar is chance (DESIGN.md section 2); the library's default "
                              "(bit-identical arithmetic) reproduces the reference on it",
                      "reference_iterations": rj["trivial"]["runs"]["1e-06"]["iterations"]}
                idx, want = rm["trivial/sample_idx"], rm["trivial/samples_1e-06"]
                for label, factor in (("rule (hand-over at 100 eps)", None), ("hand-over at 10 eps (round 3's rule)", "10")):
                    if factor:
                        os.environ["EPIC_HIP_TOL_FINISH_FACTOR"] = factor
                    try:
                        h = HarmonicMap().load(os.path.join(gdir, "maps", "trivial.png"))
                        got, its = complete(h, args.math, "redblack", 1e-6)
                    finally:
                        os.environ.pop("EPIC_HIP_TOL_FINISH_FACTOR", None)
                    e = dist(got[idx], want, h.locked_array().ravel()[idx])
                    tr[label] = {"iterations": its, "max_rel": e["max_rel"], "within_bar": e["within_bar"]}
                out["ill_conditioned_map"] = tr
        except (OSError, ValueError, KeyError) as exc:
            out["callers_eps_error"] = repr(exc)
        synth = np.load(os.path.join(gdir, "synthetic_converged.npz"))
        for n in (512, 1024):
            u0, locked = synthetic_grid([n, n])
            h = Harmonic()
            h.set_grid([n, n], u0, locked)
            got, its = complete(h, args.math, args.scheme)
            e = dist(got, synth["s%d/converged" % n], locked)
            e.update(iterations=its, against="harmonic_complete_cpu's converged field (tests/golden/synthetic_converged.npz)")
            out["configs"]["configs[2] family, %dx%d" % (n, n)] = e
        for key, label in (("relax_jacobi", "configs[2] 8192x8192 (the timed grid)"), ("relax", "configs[2] 8192x8192, red-black (the `relax` leg)")):
            if key in relaxed and "relax_default" in relaxed:
                e = dist(relaxed[key], relaxed["relax_default"], locked_8192)
                e.update(against="this run's relax_default (precise + red-black, the reference's iteration bit for bit)")
                # round 6: that field is no longer only the library's word -- the reference's loop was run on the CPU at this size once
                # (tests/golden/generate_8192_golden.py, half-sweeps dealt to threads: the sequential result bit for bit) and its sha256 committed
                try:
                    import hashlib

                    g8 = json.load(open(os.path.join(gdir, "synthetic_8192.json")))
                    e["relax_default_equals_cpu_statement_of_the_reference"] = bool(hashlib.sha256(np.ascontiguousarray(relaxed["relax_default"]).tobytes()).hexdigest() == g8["sha_u"])
                    e["cpu_statement"] = "tests/golden/synthetic_8192.json: %d iterations, delta %.3e, %d threads, %.0f s" % (g8["iterations"], g8["delta"], g8["threads"], g8["seconds"])
                except (OSError, ValueError, KeyError):
                    e["relax_default_equals_cpu_statement_of_the_reference"] = None
                out["configs"][label] = e
        out["configs"]["configs[3] 32768x32768 on 4 / 8 GPUs"] = {"max_rel": None, "note": "not relaxed at N = 1; same arithmetic and kernels, bit-identical across slab counts (tests/test_gpu_multi_device.py)"}
        if not args.no_extra_legs:
            g3 = [512, 512, 512]
            u3, l3 = synthetic_grid(g3)
            fields = {}
            for math, scheme in ((args.math, "jacobi"), ("precise", "redblack")):
                h = Harmonic()
                h.set_grid(g3, u3, l3)
                t3 = time.perf_counter()
                fields[(math, scheme)], its3 = complete(h, math, scheme)
                fields[(math, scheme, "s")] = (round(time.perf_counter() - t3, 3), its3)
            e = dist(fields[(args.math, "jacobi")], fields[("precise", "redblack")], l3)
            e.update(against="precise + red-black relaxed in this run (the reference's 3-D iteration bit for bit)")
            # whole relaxations through harmonic_complete_gpu (upload, iterations to eps = 1e-6 with automatic tracking, download)
            e.update(relax_seconds={"%s jacobi" % args.math: fields[(args.math, "jacobi", "s")][0],
                                    "precise redblack (library default)": fields[("precise", "redblack", "s")][0]},
                     iterations={"%s jacobi" % args.math: fields[(args.math, "jacobi", "s")][1],
                                 "precise redblack (library default)": fields[("precise", "redblack", "s")][1]})
            try:   # round 6: the reference's loop run on the CPU at this size once (tests/golden/generate_8192_golden.py --cube 512)
                import hashlib

                g5 = json.load(open(os.path.join(gdir, "synthetic_512cubed.json")))
                e["relax_default_equals_cpu_statement_of_the_reference"] = bool(
                    hashlib.sha256(np.ascontiguousarray(fields[("precise", "redblack")]).tobytes()).hexdigest() == g5["sha_u"])
                e["cpu_statement"] = "tests/golden/synthetic_512cubed.json: %d iterations, delta %.3e, %d threads, %.0f s" % (g5["iterations"], g5["delta"], g5["threads"], g5["seconds"])
            except (OSError, ValueError, KeyError):
                e["relax_default_equals_cpu_statement_of_the_reference"] = None
            out["configs"]["configs[4] 512x512x512"] = e
        missed = [k for k, v in out["configs"].items() if v.get("within_bar") is False]
        out["misses"] = missed
    except Exception as exc:   # evidence object: never lose the headline line
        out["error"] = repr(exc)
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


def live_traffic(args):
    """HBM bytes per launch of every sweep kernel of THIS command, measured now: two child processes, each this same script
    under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (counters in passes of their own, nothing else traced, the
    program itself behind `--`: MI355X_MICROARCH.md, HBM section) for one short step on a developed field, before this
    process has touched the GPU.  Returns ({kernel-name fragment: bytes per launch}, note) -- empty dict when the profiler
    is not there or a pass fails (the caller then falls back to the recorded value and says so)."""
    import csv
    import glob
    import shutil
    import statistics
    import subprocess
    import tempfile

    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return {}, "rocprofv3 not found"
    # never from a process that is itself being profiled: the profiler's preloaded library has initialised the GPU in it, and
    # a GPU-initialised process must not start another program
    if any("rocprof" in os.environ.get(k, "").lower() for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "ROCPROFILER_LIBRARY_PATH",
                                                                 "HSA_TOOLS_LIB")):
        return {}, "this process runs under a profiler"
    out, counts = {}, {}   # out: kernel-name fragment -> {"traffic", "launches", "SQ_INSTS_VALU", "SQ_BUSY_CYCLES", "cycles_per_valu"}
    tmp = tempfile.mkdtemp(prefix="epic_pmc_", dir="/tmp")
    # (--develop 5000: past the point where the library measures the task height of its fused passes)
    child = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps", "1", "--warmup", "1", "--develop", "5000",
             "--size", str(args.size), "--stagger", str(args.stagger), "--math", args.math, "--scheme", args.scheme,
             "--rows-per-task", str(args.rows_per_task), "--no-cpu", "--no-relax", "--no-extra-legs", "--no-parity",
             "--no-maps", "--no-node-flow", "--no-live-traffic"] + (["--track"] if args.track else [])
    env = dict(os.environ, TMPDIR="/tmp")
    try:
        # The library measures that height by timing, which a counter pass distorts.  So one plain child first, to learn the
        # height, and the counter passes with that height fixed.
        if "EPIC_HIP_FUSED_ROWS" not in env:
            r = subprocess.run(child, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=300)
            lines = [l for l in r.stdout.decode(errors="replace").splitlines() if l.startswith("{")]
            rows = json.loads(lines[-1])["config"].get("fused_rows_per_task", 0) if r.returncode == 0 and lines else 0
            if rows > 0:
                env["EPIC_HIP_FUSED_ROWS"] = str(rows)
                counts["fused_rows_per_task"] = rows
        kernels = ("jacobi_fused2d_kernel", "rb_tol_fused2d_kernel", "rb_fused2d_kernel", "sweep2d_kernel", "sweep3d_pair_kernel",
                   "sweep3d_kernel")
        # the 2-D step, then the 3-D sweeps of config5 (--only-config5: nothing but that leg), each under the same three passes:
        # the two HBM counters in passes of their own (the guide's rule), and one pass for the two SQ counters that say how
        # busy the VALU was (instructions issued against the cycles the shader engines were busy)
        for what, cmd_child in (("2d", child), ("3d", child + ["--only-config5"])):
            for counters in (("FETCH_SIZE",), ("WRITE_SIZE",), ("SQ_INSTS_VALU", "SQ_BUSY_CYCLES")):
                d = os.path.join(tmp, what + "_" + counters[0])
                cmd = [exe, "--pmc", *counters, "--output-format", "csv", "-d", d, "--"] + cmd_child
                r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=300)
                files = glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)
                if r.returncode != 0 or not files:
                    if what == "2d" and counters[0] in ("FETCH_SIZE", "WRITE_SIZE"):
                        return {}, "rocprofv3 --pmc %s pass failed (rc %d)" % (counters[0], r.returncode)
                    continue   # the SQ pass and the 3-D passes are extras: the line says what it has
                per = {}
                for row in csv.DictReader(open(files[0])):
                    if row["Counter_Name"] not in counters:
                        continue
                    for key in kernels:
                        if key in row["Kernel_Name"]:
                            per.setdefault((key, row["Counter_Name"]), []).append(float(row["Counter_Value"]))
                            break
                for (key, counter), vals in per.items():
                    vals = vals[-90:]   # the steady state: the last step's launches (the library settles its task height during the develop phase)
                    entry = out.setdefault(key, {})
                    if counter in ("FETCH_SIZE", "WRITE_SIZE"):
                        # counter unit KiB; FETCH_SIZE under-reports streaming reads by 2 on gfx950 (the guide's correction)
                        entry["traffic"] = entry.get("traffic", 0.0) + statistics.mean(vals) * 1024.0 * (2.0 if counter == "FETCH_SIZE" else 1.0)
                        entry["launches"] = min(entry.get("launches", 1 << 30), len(vals))
                    else:
                        entry[counter] = statistics.mean(vals)
    except (OSError, subprocess.SubprocessError, ValueError, KeyError) as exc:
        return {}, "live PMC measurement failed: %r" % (exc,)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    for entry in out.values():
        if "traffic" in entry:
            entry["traffic"] = int(round(entry["traffic"]))
        if entry.get("SQ_INSTS_VALU") and entry.get("SQ_BUSY_CYCLES"):
            # SQ_BUSY_CYCLES sums the 32 shader engines, SQ_INSTS_VALU the wave instructions of all 1024 SIMDs: cycles per VALU
            # instruction per SIMD against the 4 a wave64 instruction occupies the SIMD for (tools/summarize_profile.py: sq)
            entry["cycles_per_valu"] = (entry["SQ_BUSY_CYCLES"] / 32.0) / (entry["SQ_INSTS_VALU"] / 1024.0)
    return out, counts


def summarise_into_config(out):
    """BASELINE's metric is "relax to eps = 1e-6"; the timed steps are a steady-state rate.  The driver's record keeps `config`,
    `roofline` and `cpu_baseline` whole and only NAMES the other objects, so the time-to-solution, the parity verdict and the
    other configs' headline numbers are repeated inside `config`: relax_to_eps for the fastest parity-clean mode and for the
    library default (what an unchanged caller gets), parity_misses, config5_frac, maps_seconds."""
    cfg = out["config"]

    def leg(key):
        x = out.get(key)
        if not x:
            return None
        return {"mode": "%s %s%s" % (x["math"], x["scheme"], " + finishing iterations" if x.get("finishing_iterations") else ""),
                "seconds": x["seconds"], "iterations": x["iterations"], "finishing_iterations": x.get("finishing_iterations", 0),
                "recomputed_Mcell_updates_per_s": x["recomputed_Mcell_updates_per_s"],
                "effective_Mcell_updates_per_s": x["effective_Mcell_updates_per_s"]}

    legs = {k: leg(k) for k in ("relax", "relax_jacobi", "relax_default")}
    if any(legs.values()):
        # with a parity miss in the line only the bit-exact default counts as clean
        misses = (out.get("parity") or {}).get("misses")
        have = [(legs[k]["seconds"], k) for k in (("relax_default",) if misses else tuple(legs)) if legs.get(k)]
        best = min(have)[1] if have else None
        cfg["relax_to_eps"] = {
            "epsilon": 1e-6,
            "fastest_parity_clean": dict(legs[best], leg=best) if best else None,
            "library_default": legs["relax_default"],
            "timed_scheme": legs["relax_jacobi"],
            "note": "harmonic_execute_gpu from the initial field to the reference's stop, final D2H included; library_default = no "
                    "environment (bit-identical to harmonic_complete_cpu); full objects: the line's relax* keys"}
    if "parity" in out and "error" not in out["parity"]:
        cfg["parity_misses"] = out["parity"].get("misses")
    for key in ("config5", "config4", "maps", "node_flow", "parity", "cpu_baseline"):
        if isinstance(out.get(key), dict) and set(out[key]) == {"error"}:
            cfg["%s_error" % key] = out[key]["error"]
    if "config5" in out and "frac" in out["config5"]:
        cfg["config5_frac"] = out["config5"]["frac"]
        for key, leg3 in (out["config5"].get("relax_seconds") or {}).items():
            cfg["config5_relax_seconds_%s" % key.split(" (")[0].replace(" ", "_").replace("-", "")] = leg3["seconds"]
        if "precise" in out["config5"]:
            cfg["config5_frac_default_math"] = out["config5"]["precise"]["frac"]
    if "kernels" in out and "precise" in out["kernels"]:
        cfg["default_math_frac"] = out["kernels"]["precise"]["frac"]   # the bit-exact sweep an unchanged caller's kernels run at
    if "maps" in out and "error" not in out["maps"]:
        cfg["maps_seconds"] = {name: out["maps"]["%s default eps 1e-06" % name]["seconds"] for name in ("maze", "umass")
                               if "%s default eps 1e-06" % name in out["maps"]}
    # The driver's record keeps SCALARS of `config` only (round 5's nested relax_to_eps / parity_misses / maps_seconds were dropped from
    # BENCH_r05.parsed): everything BASELINE's metric ("relax to eps = 1e-6") and the review need is repeated flat, one number each.
    rte = cfg.get("relax_to_eps") or {}
    if rte.get("library_default"):
        cfg["relax_default_seconds"] = rte["library_default"]["seconds"]
        cfg["relax_default_iterations"] = rte["library_default"]["iterations"]
    if rte.get("fastest_parity_clean"):
        cfg["relax_fastest_seconds"] = rte["fastest_parity_clean"]["seconds"]
        cfg["relax_fastest_mode"] = rte["fastest_parity_clean"]["mode"]
        cfg["relax_fastest_iterations"] = rte["fastest_parity_clean"]["iterations"]
        cfg["relax_finishing_iterations"] = rte["fastest_parity_clean"]["finishing_iterations"]
    if "parity" in out and "error" not in out["parity"]:
        cfg["parity_miss_count"] = len(out["parity"].get("misses") or [])
        e8 = (out["parity"].get("configs") or {}).get("configs[2] 8192x8192 (the timed grid)") or {}
        if e8.get("relax_default_equals_cpu_statement_of_the_reference") is not None:
            cfg["default_field_8192_equals_cpu_reference"] = e8["relax_default_equals_cpu_statement_of_the_reference"]
            cfg["timed_mode_8192_max_rel_vs_reference"] = e8.get("max_rel")
        e5 = (out["parity"].get("configs") or {}).get("configs[4] 512x512x512") or {}
        if e5.get("relax_default_equals_cpu_statement_of_the_reference") is not None:
            cfg["default_field_512cubed_equals_cpu_reference"] = e5["relax_default_equals_cpu_statement_of_the_reference"]
            cfg["timed_mode_512cubed_max_rel_vs_reference"] = e5.get("max_rel")
    for name, secs in (cfg.get("maps_seconds") or {}).items():
        cfg["%s_seconds" % name] = secs
    roof = out.get("roofline") or {}
    for src, dst in (("hbm_frac_measured", "hbm_frac_measured"), ("valu_issue_frac", "valu_issue_frac")):
        if roof.get(src) is not None:
            cfg[dst] = roof[src]
    nf = out.get("node_flow") or {}
    if "error" in nf:
        nf = {}
    for name, key in (("maze", "maze"), ("umass", "umass"), ("8192^2", "8192")):
        e = nf.get(name)
        if e and e.get("node_flow"):
            cfg["node_flow_%s_us_per_iteration" % key] = e["node_flow"][0]["us_per_iteration"]          # 50 steps per tick
            cfg["node_flow_%s_ratio_to_execute" % key] = e["node_flow"][0]["ratio_to_execute"]
            cfg["node_flow_%s_undeferred_us_per_iteration" % key] = e["undeferred"][0]["us_per_iteration"]
    if nf:
        cfg["node_flow_bit_identical"] = all(x["bit_identical"] for e in nf.values() if isinstance(e, dict) and "node_flow" in e
                                             for x in e["node_flow"] + e.get("undeferred", []))
    # the tol mode's parity behind a miss RATE: the committed campaign on generated maps (tests/tol_campaign.py, CPU: the loop the device
    # runs bit for bit -- tests/test_gpu_tol.py -- against the reference's harmonic_complete_cpu)
    try:
        camp = json.load(open(os.path.join(ROOT, "tests", "golden", "tol_campaign.json")))["summary"]
        cfg["tol_campaign_cases"] = camp["cases"]
        cfg["tol_campaign_misses"] = camp["misses"]
        cfg["tol_campaign_worst_rel"] = camp["worst_rel"]
        cfg["tol_campaign_misses_with_warning"] = camp["misses_with_warning"]
        # (every miss is a Jacobi run stopped at a coarse epsilon; the same maps for the Jacobi scheme with EPIC_HIP_JACOBI_CHECKS=reference, opt-in)
        camp = json.load(open(os.path.join(ROOT, "tests", "golden", "tol_campaign_reference_checks.json")))["summary"]
        cfg["tol_campaign_jacobi_reference_checks_cases"] = camp["cases"]
        cfg["tol_campaign_jacobi_reference_checks_misses"] = camp["misses"]
        cfg["tol_campaign_jacobi_reference_checks_worst_rel"] = camp["worst_rel"]
    except (OSError, ValueError, KeyError):
        pass


def self_launch(args):
    """`python3 bench.py --gpus N` with N > 1 typed as is (no launcher): start the N ranks ourselves, as a CHILD process --
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py <same
    arguments>` -- relay the child's output (rank 0's one JSON line on stdout) and return its exit code.  This process has
    made no GPU call and has not imported torch at this point, and it never replaces itself (a child, not an exec)."""
    import socket
    import subprocess

    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:   # a free rendezvous port on the loopback interface
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL's P2P transport needs on this driver
    env.setdefault("OMP_NUM_THREADS", "4")
    child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)   # stderr goes straight through
    lines = []
    for line in child.stdout:
        (lines.append(line) if line.startswith("{") else sys.stderr.write(line))
    rc = child.wait()
    for line in lines[-1:]:    # ONE line: rank 0's
        sys.stdout.write(line)
    sys.stdout.flush()
    if rc == 0 and not lines:
        sys.stderr.write("bench.py: the %d ranks ended without a result line\n" % args.gpus)
        rc = 1
    return rc


T_START = time.perf_counter()
# The A/B legs of this script switch kernel families with STUDY knobs (EPIC_HIP_NO_FUSE, EPIC_HIP_FUSED_ROWS in the counter passes): the library reads those
# only when EPIC_HIP_STUDY=1 says the caller means them (epic_amd/csrc/driver_config.cpp).  With none of them set a context is the library's default.
os.environ.setdefault("EPIC_HIP_STUDY", "1")
LEG_SECONDS = {}   # wall seconds per leg of this run, reported in the line (`leg_seconds`): where a default run's minutes go


def timed_leg(name, fn, *a, **kw):
    """An evidence leg beside the headline: timed, and never allowed to take the line with it -- a leg that raises is reported as
    {"error": ...} under its key (and on stderr), the keys derived from it are simply absent."""
    t0 = time.perf_counter()
    try:
        return fn(*a, **kw)
    except Exception as exc:   # noqa: BLE001
        import traceback

        traceback.print_exc()
        return {"error": "%s leg failed: %r" % (name, exc)}
    finally:
        LEG_SECONDS[name] = round(time.perf_counter() - t0, 1)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not args.in_library_child:
        sys.exit(self_launch(args))
    import numpy as np

    live, live_note = {}, "not measured in this run"
    if (int(os.environ.get("WORLD_SIZE", "1")) == 1 and args.gpus == 1 and not args.no_live_traffic and not args.in_library_child
            and not args.slab):
        _t_live = time.perf_counter()
        live, live_note = live_traffic(args)   # child processes, before anything here initialises the GPU
        LEG_SECONDS["live_traffic (rocprofv3 --pmc child passes)"] = round(time.perf_counter() - _t_live, 1)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("EPIC_BENCH_BACKEND", "nccl")   # "gloo": ranks may share a GPU (1-GPU smoke of the N > 1 path)
    # N > 1: the same grid through the C-ABI in ONE process on all GPUs (EPIC_HIP_DEVICES; halos by the copy engines) -- FIRST,
    # as a child of rank 0 BEFORE this process makes any torch.cuda call (device_count() included: without amdsmi it is
    # hipGetDeviceCount, which initialises the runtime, and a process that has initialised the GPU must not start another
    # program) and while the other ranks wait in the rendezvous below with their GPUs idle.  Eligibility comes from the
    # environment alone; a child that finds too few devices says so in its line.  A child, so that a fault in a path no
    # hardware has run yet cannot take the headline line with it.
    in_library = None
    if (world > 1 and rank == 0 and not args.no_extra_legs and not args.in_library_child
            and (backend == "nccl" or os.environ.get("EPIC_BENCH_DEVLIST"))):
        import subprocess

        env = {k: v for k, v in os.environ.items()
               if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK", "ROLE_RANK",
                            "LOCAL_WORLD_SIZE", "ROLE_WORLD_SIZE", "TORCHELASTIC_RUN_ID")}
        try:
            r2 = subprocess.run([sys.executable, os.path.abspath(__file__), "--in-library-child", str(world), "--steps", str(args.steps),
                                 "--math", args.math, "--size", str(args.size), "--develop", str(min(args.develop, 6000)),
                                 "--stagger", str(args.stagger), "--no-live-traffic"],
                                env=env, capture_output=True, text=True, timeout=600 if world >= 4 else 300)   # (well inside the other ranks' wait below)
            lines = [l for l in r2.stdout.splitlines() if l.startswith("{")]
            in_library = json.loads(lines[-1]) if lines else {"error": "rc %d: %s" % (r2.returncode, r2.stderr[-400:])}
            if in_library is not None and r2.stderr:
                in_library["stderr_tail"] = r2.stderr[-600:]   # the library reports refused peer access here
        except BaseException as exc:   # evidence leg only: never lose the headline line
            in_library = {"error": repr(exc)}
        LEG_SECONDS["in_library child (one process, all devices)"] = round(time.perf_counter() - T_START, 1)
    import torch  # first: one HIP runtime per process (epic_amd/epic_harmonic.py)

    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d"
                     % (args.gpus, args.gpus))
        args.gpus = world
    ndev = torch.cuda.device_count()
    if ndev < 1:
        sys.exit("bench.py: no GPU visible")
    if backend == "nccl" and local >= ndev:
        sys.exit("bench.py: rank %d has no GPU of its own (%d visible)" % (local, ndev))
    local = local % ndev
    torch.cuda.set_device(local)
    red_dev = "cuda" if backend == "nccl" else "cpu"          # where the few scalar reductions of this script live
    if world > 1:
        import datetime

        import torch.distributed as dist

        wait = datetime.timedelta(minutes=20)   # the other ranks wait here while rank 0 runs the in-library leg (at most 10 minutes: above)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local), timeout=wait)
        else:
            dist.init_process_group(backend, timeout=wait)
    from epic_amd import epic_harmonic as eh
    from epic_amd.synthetic import synthetic_grid

    E = eh._epic
    n = args.size
    MODES = {"precise": 0, "fast": 1, "traffic": 2, "tol": 4}

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(*vals):
        if world == 1:
            return vals
        t = torch.tensor(vals, dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return tuple(float(x) for x in t)

    develop = max(0, args.develop) // args.stagger * args.stagger
    sweeps = args.steps * args.stagger

    # ---------------------------------------------------------------------------------------------------------
    # N = 1: the C-ABI on one device
    # ---------------------------------------------------------------------------------------------------------
    def abi_setup(grid, u0, locked, math, scheme, track, devices=None):
        from epic_amd.harmonic import Harmonic

        if devices:
            os.environ["EPIC_HIP_DEVICES"] = devices
        try:
            h = Harmonic()
            h.set_grid(grid, u0, locked)
            h.epsilon = 1e-6
            h.numIterationsToStaggerCheck = args.stagger
            t0 = time.perf_counter()
            for fn in (E.harmonic_initialize_dimension_size_gpu, E.harmonic_initialize_potential_values_gpu,
                       E.harmonic_initialize_locked_gpu):
                if fn(h) != 0:
                    raise RuntimeError("bench.py: %s failed -- no usable GPU" % fn.__name__)
            upload_s = time.perf_counter() - t0
        finally:
            if devices:
                del os.environ["EPIC_HIP_DEVICES"]
        assert E.harmonic_initialize_gpu(h, 1024) == 0
        if args.rows_per_task:
            E.epic_hip_set_rows_per_task(h, args.rows_per_task)
        assert E.epic_hip_set_math_mode(h, MODES[math]) == 0
        assert E.epic_hip_set_scheme(h, 1 if scheme == "redblack" else 0) == 0
        assert E.epic_hip_set_activity_tracking(h, 1 if track else 0) == 0
        return h, upload_s

    def abi_timed(h, steps, warmup, do_develop=True):
        """W untimed + K timed steps; returns (wall seconds, device milliseconds)."""
        ms = ct.c_float(0.0)
        if do_develop and develop:
            assert E.epic_hip_update_n_gpu(h, develop, 0) == 0
        for _ in range(warmup):
            assert E.epic_hip_timed_sweeps_gpu(h, args.stagger, args.stagger, ct.byref(ms)) == 0
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        dev_ms = 0.0
        for _ in range(steps):
            assert E.epic_hip_timed_sweeps_gpu(h, args.stagger, args.stagger, ct.byref(ms)) == 0
            dev_ms += ms.value
        torch.cuda.synchronize()
        return time.perf_counter() - t0, dev_ms

    def abi_release(h):
        for fn in (E.harmonic_uninitialize_gpu, E.harmonic_uninitialize_dimension_size_gpu,
                   E.harmonic_uninitialize_potential_values_gpu, E.harmonic_uninitialize_locked_gpu):
            fn(h)

    def roofline(cells_per_launch, launch_us, math, scheme, single_device_full_grid, per_pass=1, kernel=None, valu_bound=None):
        """cells_per_launch: grid cells one launch of the dominant kernel sweeps; per_pass: iterations it advances (2 for
        the fused passes: the algorithmic bytes of a launch are those of the iterations it performs, SURVEY.md section
        8(d): 8 B per cell per iteration).  `bound` names what the kernel is actually limited by, measured: "valu" where the
        SQ counters of this run (or the round's profile) show the vector ALU issuing near its rate while HBM moves well under
        its peak; `frac` stays the ALGORITHMIC-bytes fraction the metric defines, `hbm_frac_measured` is the physical one."""
        achieved = BYTES_PER_CELL_SWEEP * cells_per_launch * per_pass / (launch_us * 1e-6) / 1e9
        fused = per_pass == 2
        if kernel is None:
            kernel = (("jacobi_fused2d_kernel" if scheme == "jacobi" else "rb_tol_fused2d_kernel" if math == "tol" else "rb_fused2d_kernel")
                      if fused else "sweep2d_kernel")
        traffic, source, sq = None, None, {}
        mine = live.get(kernel, {}) if single_device_full_grid and math == args.math else {}
        if "traffic" in mine:
            traffic = mine["traffic"]
            source = ("measured in this run: rocprofv3 --pmc FETCH_SIZE (x2, gfx950) + --pmc WRITE_SIZE, separate child passes of "
                      "this command, mean over %d launches of the kernel" % mine.get("launches", 0)
                      + (" at %d rows per task (the height a plain child pass saw the library measure)" % live_note["fused_rows_per_task"]
                         if fused and isinstance(live_note, dict) and "fused_rows_per_task" in live_note else ""))
        elif single_device_full_grid and kernel.startswith(("jacobi_fused2d", "sweep2d", "rb_")):
            traffic = measured_traffic(n, math, scheme + ("_fused" if fused else ""))
            if traffic is not None:
                source = ("recorded: profiles/hbm_traffic.json (an earlier PMC run of this command; live measurement: %s)"
                          % (live_note if isinstance(live_note, str) else "not for this kernel"))
        if "cycles_per_valu" in mine:
            sq = {"valu_issue_frac": round(4.0 / mine["cycles_per_valu"], 4),
                  "valu_cycles_per_instruction": round(mine["cycles_per_valu"], 3),
                  "valu_insts_per_cell_update": round(mine["SQ_INSTS_VALU"] * 64.0 / (cells_per_launch * per_pass), 2),
                  "valu_source": "measured in this run: rocprofv3 --pmc SQ_INSTS_VALU SQ_BUSY_CYCLES, a child pass of this command; "
                                 "issue frac = 4 cycles (one wave64 instruction on a SIMD) / (SQ_BUSY_CYCLES / 32 engines) x (SQ_INSTS_VALU / 1024 SIMDs)"}
        hbm_measured = None if traffic is None else traffic / (launch_us * 1e-6) / 1e9
        if valu_bound is None:   # decide from what was measured; without counters: what the round's profiles established
            if sq and hbm_measured is not None:
                valu_bound = sq["valu_issue_frac"] > hbm_measured / 6300.0   # busier than the memory side against ITS achievable rate
            else:
                valu_bound = fused
        out_r = {
            "bound": "valu" if valu_bound else "hbm", "kernel": kernel, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic, "traffic_source": source, "launch_us": round(launch_us, 3),
            # `frac` prices the ALGORITHMIC bytes (8 B per cell per iteration).  What the kernel really
            # moves through HBM is `traffic`; that rate against the same peak is this:
            "hbm_frac_measured": None if hbm_measured is None else round(hbm_measured / HBM_PEAK_GBPS, 4),
            "hbm_GBps_measured": None if hbm_measured is None else round(hbm_measured, 1),
            "limiter": "valu" if valu_bound else "hbm",   # the same decision as `bound` (kept for readers of earlier rounds' lines)
            "iterations_per_launch": per_pass,
            "bytes_per_launch": int(BYTES_PER_CELL_SWEEP * cells_per_launch * per_pass),
            "note": "frac = 8 B x grid cells x iterations per launch / mean launch-to-launch device time (HIP events on the kernel's "
                    "stream, over a batch of plain iterations = launches of this kernel only) / 8 TB/s: an ALGORITHMIC-bytes figure "
                    "(achieved, peak and unit are those of that figure); bound says what limits the kernel"
                    + ("; the fused pass performs two iterations per launch while moving the field through HBM once, so its real "
                       "HBM rate (hbm_frac_measured) is about half of frac and the kernel is bound by VALU issue (valu_issue_frac), not by HBM"
                       if fused else "")
                    + ("; traffic = HBM bytes per launch from the PMC counters, FETCH_SIZE x2 + WRITE_SIZE (see traffic_source)"
                       if traffic is not None else "; traffic: no PMC measurement of this configuration"),
        }
        out_r.update(sq)
        return out_r

    def config5_leg(sweeps3=300):
        """BASELINE configs[4]: 3-D 512^3, 7-point, the timed arithmetic, on a developed field (1 500 of the ~3 800 sweeps the
        relaxation takes), tracking off; 8 algorithmic bytes per cell per sweep as in 2-D.  With its own roofline object."""
        g3 = [512, 512, 512]
        u3, l3 = synthetic_grid(g3)
        h3, _ = abi_setup(g3, u3, l3, args.math, "jacobi", False)
        ms3 = ct.c_float(0.0)
        assert E.epic_hip_update_n_gpu(h3, 1500, 0) == 0
        assert E.epic_hip_timed_sweeps_gpu(h3, 100, 100, ct.byref(ms3)) == 0
        assert E.epic_hip_timed_sweeps_gpu(h3, sweeps3, 100, ct.byref(ms3)) == 0
        us3 = ms3.value * 1e3 / sweeps3
        res = {"workload": "synthetic 512x512x512, 5% obstacles + 1 goal, 7-point log-space Jacobi (BASELINE configs[4])",
               "math": args.math, "us_per_sweep": round(us3, 2), "Mcell_updates_per_s": round(int((l3 == 0).sum()) / us3, 1),
               "frac": round(BYTES_PER_CELL_SWEEP * 512 ** 3 / (us3 * 1e-6) / 1e9 / HBM_PEAK_GBPS, 4),
               "kernel": "sweep3d_pair_kernel" if args.math == "tol" and os.environ.get("EPIC_HIP_3D_PAIR", "1")[:1] != "0" else "sweep3d_kernel",
               "developed_sweeps": 1600, "sweeps": sweeps3}
        res["roofline"] = roofline(512 ** 3, us3, args.math, "jacobi", True, 1, kernel=res["kernel"])
        res["roofline"]["note"] = ("8 B x 512^3 cells / mean device time per sweep of a step (99 plain sweeps + 1 check per 100, HIP events "
                                   "on the library's stream) / 8 TB/s; traffic and the VALU figures: PMC child passes of this command "
                                   "(--only-config5) when measured in this run")
        if args.math != "precise" and not args.only_config5:
            assert E.epic_hip_set_math_mode(h3, MODES["precise"]) == 0   # the bit-exact 3-D sweep beside it, same field, same run
            assert E.epic_hip_timed_sweeps_gpu(h3, 100, 100, ct.byref(ms3)) == 0
            assert E.epic_hip_timed_sweeps_gpu(h3, 100, 100, ct.byref(ms3)) == 0
            res["precise"] = {"us_per_sweep": round(ms3.value * 10.0, 2),
                              "frac": round(BYTES_PER_CELL_SWEEP * 512 ** 3 / (ms3.value * 1e-5) / 1e9 / HBM_PEAK_GBPS, 4),
                              "kernel": "sweep3d_kernel", "note": "the library's default arithmetic (expf / logf bit-identical to glibc's), Jacobi"}
        if not args.only_config5 and not args.no_relax:
            # whole relaxations of the 512^3 grid to eps = 1e-6 through harmonic_execute_gpu, from the initial field: the library default
            # (bit-exact red-black), the timed arithmetic with the timed scheme, and the timed arithmetic with the red-black scheme
            # (2-D's fastest route to a converged field)
            res["relax_seconds"] = {}
            for key, mth, sch in (("default (precise red-black)", "precise", "redblack"), ("%s jacobi" % args.math, args.math, "jacobi"),
                                  ("%s red-black" % args.math, args.math, "redblack")):
                assert E.harmonic_uninitialize_gpu(h3) == 0
                h3.u_array().ravel()[:] = u3
                assert E.harmonic_update_model_gpu(h3) == 0
                assert E.epic_hip_set_math_mode(h3, MODES[mth]) == 0 and E.epic_hip_set_scheme(h3, 1 if sch == "redblack" else 0) == 0
                assert E.epic_hip_set_activity_tracking(h3, 2) == 0
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                rc3 = E.harmonic_execute_gpu(h3, 1024)
                dt3 = time.perf_counter() - t0
                fin3 = int(E.epic_hip_finish_iteration(h3))
                res["relax_seconds"][key] = {"seconds": round(dt3, 3), "iterations": int(h3.currentIteration), "rc": rc3,
                                             "finishing_iterations": int(h3.currentIteration) - fin3 if fin3 else 0}
                assert E.harmonic_initialize_gpu(h3, 1024) == 0
        abi_release(h3)
        return res

    def config4_leg():
        """BASELINE configs[3]'s grid (32768^2) on ONE GPU: 2 x 4.3 GB of u and 0.13 GB of masks resident, the timed arithmetic,
        plain iterations on a developed field (past the iteration at which the library measures its task height)."""
        n4 = 32768
        t0 = time.perf_counter()
        u4, l4 = synthetic_grid([n4, n4])
        gen_s = time.perf_counter() - t0
        h4, up_s = abi_setup([n4, n4], u4, l4, args.math, args.scheme, False)
        del u4, l4
        ms4 = ct.c_float(0.0)
        dev4 = 17000
        assert E.epic_hip_update_n_gpu(h4, dev4, 0) == 0
        per4 = max(1, int(E.epic_hip_iterations_per_pass(h4)))
        assert E.epic_hip_timed_sweeps_gpu(h4, 40, 0, ct.byref(ms4)) == 0
        assert E.epic_hip_timed_sweeps_gpu(h4, 100, 0, ct.byref(ms4)) == 0
        l_us = ms4.value * 1e3 / (100 // per4)
        assert E.epic_hip_timed_sweeps_gpu(h4, 200, 100, ct.byref(ms4)) == 0   # whole steps: checks and the odd iteration included
        step_us = ms4.value * 1e3 / 200
        rows4 = int(E.epic_hip_fused_rows_per_task(h4))
        abi_release(h4)
        cells = n4 * n4 if args.scheme == "jacobi" else n4 * n4 // 2
        return {"workload": "synthetic 32768x32768 (BASELINE configs[3]'s grid) on ONE GPU, %s %s, developed field (%d iterations)" % (args.math, args.scheme, dev4),
                "kernel": "jacobi_fused2d_kernel" if per4 == 2 and args.scheme == "jacobi" else "sweep2d_kernel", "iterations_per_launch": per4,
                "launch_us": round(l_us, 2), "frac": round(BYTES_PER_CELL_SWEEP * cells * per4 / (l_us * 1e-6) / 1e9 / HBM_PEAK_GBPS, 4),
                "step_us_per_iteration": round(step_us, 2), "fused_rows_per_task": rows4,
                "host_grid_generation_seconds": round(gen_s, 1), "h2d_seconds": round(up_s, 2),
                "note": "frac as roofline.frac: 8 B x cells x iterations per launch / launch-to-launch device time / 8 TB/s (algorithmic bytes)"}

    def maps_leg():
        """BASELINE configs[0] / [1] and the reference's 256^2 test map, relaxed as the plugin does it (harmonic_complete_gpu through
        Harmonic.solve: initialise x 3, complete, uninitialise -- upload and download included; second of two runs).  These grids
        take the small-grid path (epic_amd/csrc/kernels_tile2d.hip: eight iterations per launch on LDS tiles)."""
        from epic_amd.harmonic_map import HarmonicMap

        ref = {}
        for f in ("manifest.json", "ref_maps.json"):
            try:
                for k, v in json.load(open(os.path.join(ROOT, "tests", "golden", f)))["maps"].items():
                    ref.setdefault(k, {}).update(v.get("runs", {}))
            except (OSError, ValueError, KeyError):
                pass
        res = {"note": "seconds: wall time of Harmonic.solve(process='gpu') incl. upload / download, best of two after a warm-up; "
                       "reference_cpu_seconds: harmonic_complete_cpu of the reference's own sources, one core of the build container "
                       "(tests/golden/*.json), not of this host; default = no environment (bit-identical to that CPU run)"}
        saved = {k: os.environ.get(k) for k in ("EPIC_HIP_MATH", "EPIC_HIP_SCHEME")}
        try:
            for mode, (mth, sch) in (("default", (None, None)), ("tol_redblack", ("tol", "redblack")), ("tol_jacobi", ("tol", "jacobi"))):
                for k, v in (("EPIC_HIP_MATH", mth), ("EPIC_HIP_SCHEME", sch)):
                    os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
                for name in ("maze", "umass", "basic"):
                    for eps in ((1e-3, 1e-6) if mode == "default" else (1e-6,)):
                        best = None
                        for rep in range(3):
                            hm = HarmonicMap().load(os.path.join(ROOT, "tests", "golden", "maps", name + ".png"))
                            t0 = time.perf_counter()
                            hm.solve(process="gpu", epsilon=eps)
                            dt = time.perf_counter() - t0
                            best = dt if rep == 1 else (min(best, dt) if rep == 2 else None)
                        r = ref.get(name, {}).get("%g" % eps, {})
                        res["%s %s eps %g" % (name, mode, eps)] = {
                            "seconds": round(best, 4), "iterations": int(hm.currentIteration),
                            "us_per_iteration": round(best / hm.currentIteration * 1e6, 3),
                            "reference_cpu_seconds": r.get("seconds"), "reference_iterations": r.get("iterations")}
        finally:
            for k, v in saved.items():
                os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
        return res

    def node_flow_leg(big=None):
        """The reference's SECOND caller, the navigation node: every tick one harmonic_update_and_check_gpu and steps_per_update - 1
        single harmonic_update_gpu calls (src/epic_navigation_node_harmonic.cpp:165-189; 50 steps at 10 Hz by default, 100 at 30 Hz in
        launch/epic_navigation_node_umass.launch).  Since round 6 a plain update only counts and the library enqueues whole blocks
        (epic_amd/csrc/driver_loop.hip: deferred iterations).  Per grid, library defaults (no environment), from the initial field:
          execute    harmonic_execute_gpu to eps = 1e-6 (N iterations; its own final D2H included),
          node_flow  exactly N iterations in ticks of 50 / 100 through the two fine-grained calls + the final
                     harmonic_get_potential_values_gpu, the field compared bit for bit with execute's,
          undeferred the same with EPIC_HIP_DEFER=0: one single-iteration launch per call, what the calls did before round 6.
        The call loop itself is C++ (tests/plugin_replay/node_flow.cpp: -lepic through the reference's header paths, like
        replay.cpp); grids: maze.png, umass.png (BASELINE configs[0] / [1]) and the timed 8192^2 grid (`big`)."""
        from epic_amd.harmonic import Harmonic
        from epic_amd.harmonic_map import HarmonicMap

        so = os.path.join(ROOT, "tests", "plugin_replay", "libnodeflow.so")
        if not os.path.exists(so):
            return {"error": "tests/plugin_replay/libnodeflow.so is not built (python -c 'import __graft_entry__ as g; g.build()')"}
        nf = ct.CDLL(so)
        nf.node_flow_run.restype = ct.c_int
        nf.node_flow_execute.restype = ct.c_int
        res = {"note": "us_per_iteration = wall seconds / iterations, final D2H of u included on both sides; ratio = node_flow / execute "
                       "(VERDICT r05 item 1 asks <= 1.25); undeferred = EPIC_HIP_DEFER=0, one launch per call as before round 6; "
                       "bit_identical: the field after the same number of iterations equals harmonic_execute_gpu's"}
        grids = []
        for name in ("maze", "umass"):
            hm = HarmonicMap().load(os.path.join(ROOT, "tests", "golden", "maps", name + ".png"))
            grids.append((name, list(hm.shape), hm.u_array().ravel().copy(), hm.locked_array().ravel().copy()))
        if big is not None:
            grids.append(big)
        for name, grid, u0, locked in grids:
            h = Harmonic()
            h.set_grid(grid, u0, locked)
            h.epsilon = 1e-6
            h.numIterationsToStaggerCheck = 100
            for fn in (E.harmonic_initialize_dimension_size_gpu, E.harmonic_initialize_potential_values_gpu, E.harmonic_initialize_locked_gpu):
                assert fn(h) == 0, fn.__name__
            sec = ct.c_double(0.0)
            done, conv = ct.c_uint(0), ct.c_uint(0)
            entry = {}
            try:
                ex_s, want, its = None, None, 0
                for rep in range(2):   # second of two
                    h.u_array().ravel()[:] = u0
                    assert E.harmonic_update_model_gpu(h) == 0
                    assert nf.node_flow_execute(ct.byref(h), 1024, ct.byref(sec)) == 0
                    ex_s, its = sec.value, int(h.currentIteration)
                want = h.u_array().ravel().copy()
                entry["execute"] = {"iterations": its, "seconds": round(ex_s, 4), "us_per_iteration": round(ex_s / its * 1e6, 3)}

                def flow(steps, reps):
                    best, same = None, None
                    for rep in range(reps):
                        h.u_array().ravel()[:] = u0
                        assert E.harmonic_update_model_gpu(h) == 0
                        if not h.d_delta:
                            assert E.harmonic_initialize_gpu(h, 1024) == 0
                        h.currentIteration = 0
                        rc = nf.node_flow_run(ct.byref(h), its, steps, 1024, 1, ct.byref(sec), ct.byref(done), ct.byref(conv))
                        assert rc == 0 and done.value == its, (rc, done.value, its)
                        best = sec.value if best is None or rep == 1 else min(best, sec.value)
                        same = bool(np.array_equal(h.u_array().ravel(), want))
                    return {"steps_per_tick": steps, "seconds": round(best, 4), "us_per_iteration": round(best / its * 1e6, 3),
                            "ratio_to_execute": round(best / ex_s, 3), "bit_identical": same, "converged_ticks": int(conv.value)}

                reps = 2 if len(u0) < (1 << 24) else 1
                entry["node_flow"] = [flow(50, reps + 1), flow(100, reps + 1)]
                os.environ["EPIC_HIP_DEFER"] = "0"
                assert E.epic_hip_config_reload(h) == 0
                try:
                    entry["undeferred"] = [flow(50, reps)]
                finally:
                    del os.environ["EPIC_HIP_DEFER"]
                    assert E.epic_hip_config_reload(h) == 0
                dump = eh.config_dump(h)
                if dump:
                    entry["kernel_path"] = dump["path"]["plain_batch"]
            finally:
                abi_release(h)
            res[name] = entry
        return res

    if args.only_config5:   # (internal: the child of the PMC passes for config5's roofline object)
        print(json.dumps({"config5": config5_leg(100)}), flush=True)
        return

    if args.in_library_child > 0:
        # the same 8192^2 grid through the C-ABI in ONE process on N GPUs (EPIC_HIP_DEVICES; halos by hipMemcpyPeerAsync, one
        # issuing thread per device); prints one JSON object
        nd = args.in_library_child
        grid = [n, n]
        u0, locked = synthetic_grid(grid)
        free_cells = int((locked == 0).sum())
        res = {}
        h = None
        try:
            devlist = os.environ.get("EPIC_BENCH_DEVLIST") or ",".join(str(d) for d in range(nd))   # (rehearsal on one GPU: "0,0,0,0")
            if not os.environ.get("EPIC_BENCH_DEVLIST") and E.epic_hip_device_count() < nd:
                print(json.dumps({"error": "%d devices asked for, %d visible to this process" % (nd, E.epic_hip_device_count())}), flush=True)
                return
            h, _ = abi_setup(grid, u0, locked, args.math, "jacobi", False, devlist)
            dev = (ct.c_int * 64)()
            nsl = E.epic_hip_device_layout(h, 64, dev, None, None, None)
            # BEFORE any timing: what the library decided per seam (peer access asked for and granted? transport; link type and
            # hops as the runtime reports them) and how one exchange iteration actually ran under timing events -- did the
            # second stream's bands and copies overlap the interior sweep?  (include/epic_hip.h: epic_hip_multi_report)
            assert E.epic_hip_update_n_gpu(h, 200, 0) == 0    # a front that has left the goal's slab
            rep = ct.create_string_buffer(1 << 16)
            nrep = E.epic_hip_multi_report(h, rep, len(rep))
            try:
                res["report"] = json.loads(rep.value.decode()) if nrep > 0 else {"error": "epic_hip_multi_report returned 0 (not in multi-device mode)"}
            except ValueError as exc:
                res["report"] = {"error": repr(exc)}
            steps = max(2, args.steps // 2)
            iw, ims = abi_timed(h, steps, 1)
            isw = steps * args.stagger
            res.update({"devices": [dev[i] for i in range(min(nsl, 64))], "slabs": nsl, "value": round(free_cells * isw / iw / 1e6, 1),
                        "unit": "Mcell-updates/s", "us_per_sweep": round(ims * 1e3 / isw, 3)})
            # and the whole relaxation as the unchanged plugin gets it on this node: library defaults (precise, red-black,
            # work lists per slab)
            assert E.harmonic_uninitialize_gpu(h) == 0
            h.u_array().ravel()[:] = u0
            assert E.harmonic_update_model_gpu(h) == 0
            assert E.epic_hip_set_math_mode(h, MODES["precise"]) == 0 and E.epic_hip_set_scheme(h, 1) == 0
            assert E.epic_hip_set_activity_tracking(h, 2) == 0
            t0 = time.perf_counter()
            rrc = E.harmonic_execute_gpu(h, 1024)
            rdt = time.perf_counter() - t0
            res["relax_default"] = {"rc": rrc, "seconds": round(rdt, 3), "iterations": int(h.currentIteration), "delta": float(h.delta)}
        finally:
            if h is not None:
                abi_release(h)
        LEG_SECONDS["imports, grid, report, timed steps, default relaxation"] = round(time.perf_counter() - T_START, 1)
        from epic_amd.synthetic import RAMP_RATE, ramp_rows

        def slabs_timed(grid3, label, steps3, dev_sweeps):
            """One more grid through the same in-library slabs: the timed arithmetic, Jacobi, developed-like start; host-clocked steps."""
            t0 = time.perf_counter()
            ug, lg = synthetic_grid(grid3)
            ramp_rows(grid3, 0, grid3[0], ug, lg, RAMP_RATE)
            gen_s = time.perf_counter() - t0
            hg = None
            try:
                hg, up_s = abi_setup(grid3, ug, lg, args.math, "jacobi", False, devlist)
                fc = int((lg == 0).sum())
                del ug, lg
                devs = (ct.c_int * 64)()
                nslg = E.epic_hip_device_layout(hg, 64, devs, None, None, None)
                assert E.epic_hip_update_n_gpu(hg, dev_sweeps, 0) == 0
                gw, gms = abi_timed(hg, steps3, 1, do_develop=False)
                cells = int(np.prod(grid3))
                us = gms * 1e3 / (steps3 * args.stagger)
                return {"workload": label, "grid": grid3, "slabs": nslg, "value": round(fc * steps3 * args.stagger / gw / 1e6, 1), "unit": "Mcell-updates/s",
                        "us_per_iteration": round(us, 2), "frac_per_device": round(BYTES_PER_CELL_SWEEP * cells / max(1, nslg) / (us * 1e-6) / 1e9 / HBM_PEAK_GBPS, 4),
                        "start": "ramp (epic_amd/synthetic.py: ramp_rows) + %d untimed iterations" % dev_sweeps,
                        "host_grid_generation_seconds": round(gen_s, 1), "h2d_seconds": round(up_s, 2),
                        "kernel_path": (eh.config_dump(hg) or {}).get("path", {}).get("plain_batch")}
            finally:
                if hg is not None:
                    abi_release(hg)

        if not args.no_extra_legs:
            _t5 = time.perf_counter()
            try:   # BASELINE configs[4] on slabs of PLANES (the 3-D form of the same decomposition)
                res["config5"] = slabs_timed([512, 512, 512], "synthetic 512^3 (BASELINE configs[4]) cut into plane slabs, %s Jacobi" % args.math,
                                             max(2, args.steps // 4), 100)
            except BaseException as exc:   # evidence legs: never lose the line
                res["config5"] = {"error": repr(exc)}
            LEG_SECONDS["config5"] = round(time.perf_counter() - _t5, 1)
        if not args.no_extra_legs and not args.no_config4 and nd >= 4:
            _t4 = time.perf_counter()
            try:   # BASELINE configs[3] through the unchanged ABI: 8.6 GB of host arrays, ~1 minute of host-side generation
                import psutil

                if psutil.virtual_memory().available < 24 * (1 << 30):
                    res["config4"] = {"skipped": "less than 24 GB of host memory available for the 32768^2 arrays"}
                else:
                    res["config4"] = slabs_timed([32768, 32768], "synthetic 32768x32768 (BASELINE configs[3]) cut into %d row slabs in ONE process, %s Jacobi" % (nd, args.math),
                                                 max(2, args.steps // 5), 200)
            except BaseException as exc:
                res["config4"] = {"error": repr(exc)}
            LEG_SECONDS["config4"] = round(time.perf_counter() - _t4, 1)
        LEG_SECONDS["whole run"] = round(time.perf_counter() - T_START, 1)
        res["leg_seconds"] = dict(LEG_SECONDS)
        res["note"] = ("harmonic_*_gpu on one Harmonic in ONE process, EPIC_HIP_DEVICES=0..N-1 (one issuing thread per device, halo rows by "
                       "hipMemcpyPeerAsync); host-clocked.  relax_default: harmonic_execute_gpu to eps = 1e-6 with the library defaults "
                       "(precise, red-black, work lists per slab), incl. the final D2H")
        print(json.dumps(res), flush=True)
        return

    out = {
        "metric": "cell_updates_per_s_log_harmonic_relax_8192sq", "value": None, "unit": "Mcell-updates/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True,
        "scaling": "none" if world == 1 else "strong",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
    }

    if world == 1 and not args.slab:
        grid = [n, n]
        u0, locked = synthetic_grid(grid)
        free_cells = int((locked == 0).sum())
        lk2 = locked.reshape(grid) == 0
        rr, cc = np.indices(lk2.shape, sparse=True)
        free_by_colour = [int((lk2 & (((rr + cc) & 1) == 1)).sum()), int((lk2 & (((rr + cc) & 1) == 0)).sum())]
        del lk2

        def updates_in(scheme, iterations, first=0):
            """Unlocked cells recomputed by `iterations` iterations: all of them per Jacobi sweep; per red-black half-sweep
            the cells with (row + col + iteration) odd (harmonic_cpu.cpp:46-51)."""
            if scheme == "jacobi":
                return free_cells * iterations
            even_it = (iterations + (1 - first % 2)) // 2      # iterations with even index update (row + col) odd
            return even_it * free_by_colour[0] + (iterations - even_it) * free_by_colour[1]

        h, upload_s = abi_setup(grid, u0, locked, args.math, args.scheme, args.track)
        wall, dev_ms = abi_timed(h, args.steps, args.warmup)
        step_us_per_sweep = dev_ms * 1e3 / sweeps
        # the dominant kernel on its own: a batch of plain iterations (no check) is launches of that kernel only -- the
        # fused pass where the library fuses (two iterations per launch), the single sweep otherwise
        per_pass = max(1, int(E.epic_hip_iterations_per_pass(h)))
        library_dump = eh.config_dump(h)   # what the timed context was configured with and which path it is on, in the library's own words
        batch = 2 * (args.stagger // 2)
        kms, ms1 = 0.0, ct.c_float(0.0)
        for _ in range(max(2, args.steps)):
            assert E.epic_hip_timed_sweeps_gpu(h, batch, 0, ct.byref(ms1)) == 0
            kms += ms1.value
        launch_us = kms * 1e3 / (max(2, args.steps) * batch // per_pass)
        cells_per_launch = n * n if args.scheme == "jacobi" else n * n // 2   # cells one ITERATION recomputes
        single_us = None
        if per_pass == 2:   # the single sweep of the same arithmetic, same run: what the fusion buys
            # (the library reads its environment once per context: a knob changed on a live one is announced -- include/epic_hip.h)
            os.environ["EPIC_HIP_NO_FUSE"] = "1"
            assert E.epic_hip_config_reload(h) == 0
            assert E.epic_hip_timed_sweeps_gpu(h, batch, 0, ct.byref(ms1)) == 0
            assert E.epic_hip_timed_sweeps_gpu(h, batch, 0, ct.byref(ms1)) == 0
            del os.environ["EPIC_HIP_NO_FUSE"]
            assert E.epic_hip_config_reload(h) == 0
            single_us = ms1.value * 1e3 / batch
        out.update({
            "value": round(updates_in(args.scheme, sweeps, develop + args.warmup * args.stagger) / wall / 1e6, 1),
            "ms_per_step": round(wall * 1e3 / args.steps, 4),
            "config": {
                "workload": "synthetic %dx%d occupancy grid, 5%% random obstacles + 1 goal (BASELINE configs[2]), log-space "
                            "Jacobi relax towards eps=1e-6" % (n, n),
                "grid": grid, "sweeps_per_step": args.stagger, "check_every": args.stagger, "developed_sweeps": develop,
                "math": args.math, "scheme": args.scheme, "activity_tracking": bool(args.track), "free_cells": free_cells,
                "parallelism": "1 GPU", "h2d_seconds": round(upload_s, 3),
                "fused_rows_per_task": int(E.epic_hip_fused_rows_per_task(h)),   # measured by the library on this grid
            },
            "roofline": roofline(cells_per_launch, launch_us, args.math, args.scheme, True, per_pass),
            "step_us_per_iteration": round(step_us_per_sweep, 3),
            "library": library_dump,   # epic_hip_config_dump of the timed context: every EPIC_HIP_* knob as read, state, kernel path
        })
        if library_dump:
            out["config"]["kernel_path"] = library_dump["path"]["plain_batch"]
        if single_us is not None:
            out.setdefault("kernels", {})["single_sweep"] = {
                "launch_us": round(single_us, 3),
                "frac": round(BYTES_PER_CELL_SWEEP * cells_per_launch / (single_us * 1e-6) / 1e9 / HBM_PEAK_GBPS, 4),
                "note": "sweep2d_kernel, one iteration per launch (EPIC_HIP_NO_FUSE=1), same arithmetic, same field, same run"}
        assert E.harmonic_uninitialize_gpu(h) == 0
        LEG_SECONDS["imports, grid, develop, timed steps, kernel batches"] = round(time.perf_counter() - T_START - sum(LEG_SECONDS.values()), 1)
        _t_relax = time.perf_counter()
        relaxed = {}   # converged 8192^2 fields of this run, for the parity object
        if not args.no_relax:
            # the complete relaxation, exactly as the plugin runs it (harmonic_execute_gpu), from the initial state: the
            # benchmarked arithmetic with each scheme (activity tracking in the library's automatic mode), with tracking off,
            # and the library as it is with NO environment (precise + red-black: bit-identical to harmonic_complete_cpu)
            # `relax` is the fastest way to a converged field with the timed ARITHMETIC -- the red-black scheme (a Jacobi run is
            # two red-black chains interleaved: twice the arithmetic for the same answer) --; the timed SCHEME's relaxation is
            # relax_jacobi.  A tol Jacobi relaxation with its finishing iterations is NOT faster to a converged field than the
            # bit-exact default (relax_default): tol is a kernel-throughput mode, and the line shows both.
            legs = [("relax", args.math, "redblack", 2), ("relax_jacobi", args.math, "jacobi", 2),
                    ("relax_untracked", args.math, args.scheme, 0), ("relax_default", "precise", "redblack", 2)]
            if args.math == "tol" and os.environ.get("EPIC_HIP_TOL_FINISH") is None:
                legs.append(("relax_tol_alone", args.math, "redblack", 2))   # `relax` without the finishing iterations, for the record
            work = ct.c_double(0.0)
            for key, math, scheme, track in legs:
                if key == "relax_tol_alone":
                    os.environ["EPIC_HIP_TOL_FINISH"] = "0"
                    assert E.epic_hip_config_reload(h) == 0
                h.u_array().ravel()[:] = u0
                assert E.harmonic_update_model_gpu(h) == 0
                assert E.epic_hip_set_math_mode(h, MODES[math]) == 0
                assert E.epic_hip_set_scheme(h, 1 if scheme == "redblack" else 0) == 0
                assert E.epic_hip_set_activity_tracking(h, track) == 0
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                rc = E.harmonic_execute_gpu(h, 1024)
                dt = time.perf_counter() - t0
                if key == "relax_tol_alone":
                    os.environ.pop("EPIC_HIP_TOL_FINISH", None)
                    assert E.epic_hip_config_reload(h) == 0
                assert rc == 0, rc
                its = int(h.currentIteration)
                assert E.epic_hip_work_done(h, ct.byref(work), 0) == 0    # whole-grid iterations' worth of tiles actually run
                # tol: the loop finishes with the reference's own iteration (red-black half-sweeps) from iteration `fin` on
                fin = int(E.epic_hip_finish_iteration(h))
                every = (updates_in(scheme, its) if fin == 0 else                 # every unlocked cell (of the colour) once per iteration
                         updates_in(scheme, fin) + updates_in("redblack", its - fin, fin))
                out[key] = {
                    "finishing_iterations": (its - fin) if fin else 0,
                    "math": math, "scheme": scheme, "activity_tracking": bool(track), "epsilon": 1e-6, "iterations": its,
                    "seconds": round(dt, 3), "delta": float(h.delta), "grid_iterations_run": round(work.value, 1),
                    "recomputed_Mcell_updates_per_s": round(every * (work.value / its) / dt / 1e6, 1),
                    "effective_Mcell_updates_per_s": round(every / dt / 1e6, 1),
                    "note": "harmonic_execute_gpu from the initial state, includes the final D2H of u.  recomputed = cells the "
                            "kernels recomputed (iterations x unlocked cells of the colour x the share of tiles that ran, "
                            "epic_hip_work_done); effective = the same with every tile counted, i.e. the rate an untracked "
                            "solver would need for this time-to-solution -- tiles skipped by activity tracking hold exactly the "
                            "values the update would have produced.  finishing_iterations: tol math -- the last iterations of the call "
                            "are the reference's own (precise red-black half-sweeps, from the first check with delta < 10 eps on), "
                            "which is what puts the converged field within the parity bar on every config; relax_tol_alone is the same "
                            "leg with EPIC_HIP_TOL_FINISH=0"}
                if key in ("relax", "relax_jacobi", "relax_default"):
                    relaxed[key] = h.u_array().ravel().copy()
            assert E.epic_hip_set_math_mode(h, MODES[args.math]) == 0
        abi_release(h)
        LEG_SECONDS["relax legs (whole 8192^2 relaxations)"] = round(time.perf_counter() - _t_relax, 1)
        if not args.no_extra_legs and args.math != "precise":
            # the bit-exact mode on the same workload, same box, same run: what the tol arithmetic buys
            hp, _ = abi_setup(grid, u0, locked, "precise", args.scheme, args.track)
            pw, pms = abi_timed(hp, max(2, args.steps // 4), 1)
            abi_release(hp)
            pl = pms * 1e3 / (max(2, args.steps // 4) * args.stagger)
            out.setdefault("kernels", {})["precise"] = {"launch_us": round(pl, 3), "frac": round(BYTES_PER_CELL_SWEEP * cells_per_launch / (pl * 1e-6) / 1e9 / HBM_PEAK_GBPS, 4),
                                          "note": "default math mode of the library: expf/logf bit-identical to glibc, f64; same grid, scheme and run"}
        if not args.no_extra_legs:
            out["config5"] = timed_leg("config5", config5_leg)
        if not args.no_extra_legs and not args.no_config4:
            out["config4"] = timed_leg("config4", config4_leg)
        if not args.no_maps:
            out["maps"] = timed_leg("maps", maps_leg)
        if not args.no_node_flow:
            out["node_flow"] = timed_leg("node_flow", node_flow_leg, ("8192^2" if n == 8192 else "%d^2" % n, grid, u0, locked))
        if not args.no_parity:
            out["parity"] = timed_leg("parity", parity_object, args, E, MODES, relaxed, locked)
        if not args.no_cpu:
            out["cpu_baseline"] = timed_leg("cpu_baseline", cpu_baseline, grid, u0, locked, args.cpu_half_sweeps, free_by_colour)
        LEG_SECONDS["whole run"] = round(time.perf_counter() - T_START, 1)
        out["leg_seconds"] = dict(LEG_SECONDS)
        summarise_into_config(out)
        print(json.dumps(out), flush=True)
        return

    # ---------------------------------------------------------------------------------------------------------
    # N > 1 (or --slab): one process per GPU, row slabs, halos over the process group
    # ---------------------------------------------------------------------------------------------------------
    from epic_amd.slab import SlabSolver

    def slab_run(grid, steps, warmup, develop=develop, ramp=0.0):
        rows_each = grid[0] // world
        halo = args.halo or (8 if rows_each >= 4096 else 16 if rows_each >= 2048 else 32)
        solver = SlabSolver(grid, rank, world, device=torch.device("cuda", local), stagger=args.stagger,
                            rows_per_task=args.rows_per_task, math=args.math, halo=halo)
        free = solver.load_synthetic(ramp=ramp)
        for _ in range(develop):
            solver.sweep(False)
        pair_rows = solver.tune_pairs() if develop >= min(grid) // 2 else 0   # on a developed field, as the library does
        probe = solver.probe_exchange()   # BEFORE any timing: did the second stream's exchange overlap the interior sweep?
        while solver.iteration % args.stagger:   # (back to a step boundary: a step = one check + stagger - 1 plain iterations)
            solver.sweep(False)
        for _ in range(warmup):
            solver.timed_step()
        barrier()
        t0 = time.perf_counter()
        dev_ms = 0.0
        for _ in range(steps):
            dev_ms += solver.timed_step()
        barrier()
        wall = time.perf_counter() - t0
        wall, dev_ms = max_over_ranks(wall, dev_ms)
        if world > 1:
            f = torch.tensor([free], dtype=torch.int64, device=red_dev)
            dist.all_reduce(f)
            free = int(f[0])
        rows_local = solver.hi - solver.lo
        res = dict(wall=wall, dev_ms=dev_ms, free=free, halo=solver.halo, rows_local=rows_local,
                   pairs=bool(getattr(solver.backend, "pairs", False)), pair_rows=int(pair_rows), probe=probe)
        del solver
        torch.cuda.empty_cache()
        return res

    weak_first = args.weak and world > 1
    main_grid = [n * world, n] if weak_first else [n, n]
    _t_leg = time.perf_counter()
    r = slab_run(main_grid, args.steps, args.warmup)
    LEG_SECONDS["headline: develop, tune, probe, timed steps"] = round(time.perf_counter() - _t_leg, 1)
    launch_us = r["dev_ms"] * 1e3 / sweeps
    actual_backend = dist.get_backend() if world > 1 else "none"
    transport = {"nccl": "RCCL send/recv of device rows (xGMI)", "gloo": "gloo, rows staged through host memory",
                 "none": "no exchange"}.get(actual_backend, actual_backend)
    out.update({
        "value": round(r["free"] * sweeps / r["wall"] / 1e6, 1),
        "ms_per_step": round(r["wall"] * 1e3 / args.steps, 4),
        "scaling": "none" if world == 1 else "weak" if weak_first else "strong",
        "config": {
            "workload": ("synthetic %dx%d occupancy grid per GPU (%dx%d in all), " % (n, n, n * world, n) if weak_first else
                         "ONE synthetic %dx%d occupancy grid cut into %d row slabs, " % (n, n, world))
                        + "5% random obstacles + 1 goal (BASELINE configs[2]), log-space Jacobi relax towards eps=1e-6",
            "grid": main_grid, "sweeps_per_step": args.stagger, "check_every": args.stagger, "developed_sweeps": develop,
            "math": args.math, "scheme": "jacobi", "activity_tracking": False, "free_cells": r["free"],
            "fused_rows_per_task": r["pair_rows"],   # rank 0's slab: measured by SlabSolver.tune_pairs (0: the library's rule)
            "parallelism": "row slabs x%d (one process per GPU), %d halo rows exchanged every %d sweeps: %s"
                           % (world, r["halo"], r["halo"], transport),
        },
        # the dominant kernel is one rank's sweep of its slab (rows_local x n cells per iteration, ghost rows not counted);
        # launch_us here is device time per ITERATION of the whole step (pairs of iterations run as one fused pass where
        # neither a check nor an exchange falls, the others singly, plus the exchanges)
        "roofline": roofline(r["rows_local"] * n, launch_us, args.math, "jacobi", False),
    })
    out["roofline"]["kernel"] = ("jacobi_fused2d_kernel (pairs of iterations) + sweep2d_kernel (checks, exchange iterations)"
                                 if r["pairs"] else "sweep2d_kernel")
    out["roofline"]["note"] = ("8 B x this rank's owned cells per iteration / mean device time per iteration over the timed steps "
                               "(HIP events on the compute stream; includes ghost rows, halo exchange waits and check iterations)")
    if world > 1:
        seen = [None] * world
        # per rank: its device, whether it may address its neighbours' devices directly (what RCCL's P2P transport needs; with one
        # process per GPU this is the runtime's answer for the two device ordinals, both visible to every rank of the node), and
        # the probed exchange of the headline run (taken before its timed region)
        peers = {}
        for nb in (rank - 1, rank + 1):
            if 0 <= nb < world and backend == "nccl" and nb < ndev:
                try:
                    peers[str(nb)] = bool(torch.cuda.can_device_access_peer(local, nb))
                except Exception as exc:   # evidence only
                    peers[str(nb)] = repr(exc)
        dist.all_gather_object(seen, {"rank": rank, "local_rank": int(os.environ.get("LOCAL_RANK", "0")), "device": local,
                                      "device_name": torch.cuda.get_device_name(local), "devices_visible": ndev,
                                      "pid": os.getpid(), "can_access_neighbour_device": peers, "exchange_probe": r.get("probe")})
        out["ranks"] = {"ranks_seen": dist.get_world_size(), "backend": actual_backend, "per_rank": seen}
        if not args.no_extra_legs:
            other_grid = [n, n] if weak_first else [n * world, n]
            _t_leg = time.perf_counter()
            w = slab_run(other_grid, max(2, args.steps // 2), 1)
            LEG_SECONDS["the other scaling mode"] = round(time.perf_counter() - _t_leg, 1)
            wsweeps = max(2, args.steps // 2) * args.stagger
            wl = w["dev_ms"] * 1e3 / wsweeps
            out["strong" if weak_first else "weak"] = {
                "grid": other_grid, "value": round(w["free"] * wsweeps / w["wall"] / 1e6, 1), "unit": "Mcell-updates/s",
                "launch_us": round(wl, 3),
                "frac_per_gpu": round(BYTES_PER_CELL_SWEEP * w["rows_local"] * n / (wl * 1e-6) / 1e9 / HBM_PEAK_GBPS, 4),
                "note": "the other scaling mode, same run, fewer steps"}
    if world >= 4 and not args.no_extra_legs and not args.no_config4:
        # BASELINE configs[3]: 32768 x 32768 cut `world` ways, one process per GPU, halos over the process group -- a few steps on a
        # developed-like start (the front of the all -1e6 start would need ~16 000 untimed iterations at ~0.35 ms each to cross this grid)
        n4 = 32768
        need = 2 * (n4 // world + 64) * n4 * 4 + (n4 // world + 64) * n4 * 4 + (1 << 30)   # two buffers of u, the int32 mask staging, slack
        free_b = torch.cuda.mem_get_info(local)[0]
        ok = torch.tensor([1 if free_b >= need else 0], dtype=torch.int64, device=red_dev)
        if world > 1:
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok[0]) == 0:
            out["config4"] = {"skipped": "a rank has %.1f GB of device memory free, the slab needs %.1f GB" % (free_b / 1e9, need / 1e9)}
        else:
            from epic_amd.synthetic import RAMP_RATE

            st4 = max(2, args.steps // 5)
            _t_leg = time.perf_counter()
            try:
                r4 = slab_run([n4, n4], st4, 1, develop=200, ramp=RAMP_RATE)
            except BaseException as exc:   # an evidence leg: never lose the headline line (a failure on one rank only surfaces on the others as a collective's timeout)
                r4 = None
                out["config4"] = {"error": repr(exc)}
            LEG_SECONDS["config4"] = round(time.perf_counter() - _t_leg, 1)
            sw4 = st4 * args.stagger
            l4 = r4["dev_ms"] * 1e3 / sw4 if r4 else 0.0
            if r4:
              out["config4"] = {
                  "workload": "synthetic 32768x32768 (BASELINE configs[3]) cut into %d row slabs, one process per GPU, %s %s" % (world, args.math, "jacobi"),
                  "grid": [n4, n4], "value": round(r4["free"] * sw4 / r4["wall"] / 1e6, 1), "unit": "Mcell-updates/s", "steps": st4,
                  "us_per_iteration": round(l4, 2), "halo": r4["halo"], "rows_per_gpu": r4["rows_local"],
                  "frac_per_gpu": round(BYTES_PER_CELL_SWEEP * r4["rows_local"] * n4 / (l4 * 1e-6) / 1e9 / HBM_PEAK_GBPS, 4),
                  "start": "ramp: u = -%.1f x Manhattan distance to the goal on unlocked cells + 200 untimed iterations (epic_amd/synthetic.py: ramp_rows)" % RAMP_RATE,
                  "exchange_probe": r4["probe"],
                  "note": "frac_per_gpu as roofline.frac: 8 B x this rank's owned cells per iteration / mean device time per iteration (checks, "
                          "exchange waits included) / 8 TB/s; value = unlocked cells of the whole grid x iterations / max-over-ranks wall time"}
    if in_library is not None:
        out["in_library"] = in_library
    if world > 1:
        LEG_SECONDS["whole run (rank 0)"] = round(time.perf_counter() - T_START, 1)
        out["leg_seconds"] = dict(LEG_SECONDS)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
