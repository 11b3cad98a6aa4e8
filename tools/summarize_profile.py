#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (gpurun_out/…) into the small text summaries kept under profiles/.

  python tools/summarize_profile.py stats  <dir-with-*_kernel_stats.csv>  > profiles/rNN_kernel_stats_<tag>.txt
  python tools/summarize_profile.py pmc    <fetch-dir> <write-dir>        > profiles/rNN_hbm_traffic_<tag>.txt
  python tools/summarize_profile.py sq     <dir> <cell-updates per dispatch>
pmc and sq look at the dispatches whose kernel name contains PROFILE_KERNEL (environment; default "sweep": the sweep2d /
sweep3d kernels; "jacobi_fused2d" for the fused double sweep, whose dispatch performs two updates per grid cell).

PMC handling follows /opt/skills/guides/MI355X_MICROARCH.md §HBM: FETCH_SIZE and WRITE_SIZE are collected in separate
passes; both are reported by rocprofv3 in KiB; on gfx950 FETCH_SIZE counts 128-B requests as 64 B for wide coalesced
streaming reads, so the read side is doubled; WRITE_SIZE is exact for 16-B-per-lane streaming stores.
"""
import csv
import glob
import os
import statistics
import sys

KERNEL = os.environ.get("PROFILE_KERNEL", "sweep")
# PROFILE_LAST=N: pmc and sq look at the last N matching dispatches only (the library measures the task height of its fused
# passes a few thousand iterations into a run; the dispatches after that are the steady state)
LAST = int(os.environ.get("PROFILE_LAST", "0"))


def find(d, pat):
    hits = glob.glob(os.path.join(d, "**", pat), recursive=True)
    if not hits:
        sys.exit("no %s under %s" % (pat, d))
    return hits[0]


def stats(d):
    rows = list(csv.DictReader(open(find(d, "*_kernel_stats.csv"))))
    print("# rocprofv3 --kernel-trace --stats  (%s)" % d)
    print("%-100s %8s %14s %12s %7s %10s %10s" % ("kernel", "calls", "total_ns", "avg_ns", "pct", "min_ns", "max_ns"))
    for r in rows:
        print("%-100s %8s %14s %12.1f %7s %10s %10s" % (r["Name"][:100], r["Calls"], r["TotalDurationNs"],
                                                        float(r["AverageNs"]), r["Percentage"], r["MinNs"], r["MaxNs"]))


def pmc_values(d, counter):
    rows = list(csv.DictReader(open(find(d, "*_counter_collection.csv"))))
    vals = [float(r["Counter_Value"]) for r in rows if r["Counter_Name"] == counter and KERNEL in r["Kernel_Name"]]
    if LAST > 0:
        vals = vals[-LAST:]
    names = sorted({r["Kernel_Name"] for r in rows if KERNEL in r["Kernel_Name"]})
    return vals, names


def pmc(fetch_dir, write_dir):
    f, names = pmc_values(fetch_dir, "FETCH_SIZE")
    w, _ = pmc_values(write_dir, "WRITE_SIZE")
    fmean, wmean = statistics.mean(f), statistics.mean(w)
    read_b = fmean * 1024 * 2      # KiB -> B, gfx950 streaming-read correction x2
    write_b = wmean * 1024
    print("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), per sweep dispatch")
    print("kernels: %s" % "; ".join(n[:90] for n in names))
    print("dispatches: fetch %d, write %d" % (len(f), len(w)))
    print("FETCH_SIZE mean %.1f KiB (min %.1f max %.1f)  -> read  %.1f MB after the gfx950 x2 correction" % (
        fmean, min(f), max(f), read_b / 1e6))
    print("WRITE_SIZE mean %.1f KiB (min %.1f max %.1f)  -> write %.1f MB" % (wmean, min(w), max(w), write_b / 1e6))
    print("HBM traffic per launch: %.1f MB" % ((read_b + write_b) / 1e6))
    return read_b + write_b


def sq(d, cells=8192 * 8192):
    """Mean of every SQ_* counter per sweep dispatch, plus the ratios the design notes quote."""
    rows = list(csv.DictReader(open(find(d, "*_counter_collection.csv"))))
    rows = [r for r in rows if KERNEL in r["Kernel_Name"]]
    names = sorted({r["Counter_Name"] for r in rows})
    if LAST > 0 and names:   # the last N dispatches (rows are in dispatch order, one row per counter and dispatch)
        ids = []
        for r in rows:
            if not ids or ids[-1] != r["Dispatch_Id"]:
                ids.append(r["Dispatch_Id"])
        keep = set(ids[-LAST:])
        rows = [r for r in rows if r["Dispatch_Id"] in keep]
    mean = {n: statistics.mean(float(r["Counter_Value"]) for r in rows if r["Counter_Name"] == n) for n in names}
    n_disp = len([r for r in rows if r["Counter_Name"] == names[0]]) if names else 0
    print("# rocprofv3 --pmc SQ_* (%s), mean per sweep dispatch (%d dispatches of %s)" % (
        d, n_disp, "; ".join(sorted({r["Kernel_Name"][:80] for r in rows}))))
    for n in names:
        print("%-24s %.4g" % (n, mean[n]))
    if "SQ_INSTS_VALU" in mean:
        print("VALU instructions per grid cell (wave instructions x 64 lanes / 4 cells per lane ... per cell): %.1f"
              % (mean["SQ_INSTS_VALU"] * 64 / cells))
    if "SQ_INSTS_LDS" in mean:
        print("LDS instructions per grid cell: %.2f" % (mean["SQ_INSTS_LDS"] * 64 / cells))
    if "SQ_INSTS_VALU" in mean and "SQ_BUSY_CYCLES" in mean:
        # SQ_BUSY_CYCLES sums the 32 shader engines; per SIMD (1024 of them) the VALU issued one instruction every ... cycles
        print("SQ_BUSY_CYCLES / 32 = %.0f cycles per dispatch; one VALU instruction per %.2f of them per SIMD" % (
            mean["SQ_BUSY_CYCLES"] / 32, mean["SQ_BUSY_CYCLES"] / 32 / (mean["SQ_INSTS_VALU"] / 1024)))
    if "SQ_WAIT_ANY" in mean and "SQ_WAVE_CYCLES" in mean:
        print("waves parked at s_waitcnt (SQ_WAIT_ANY / SQ_WAVE_CYCLES): %.3f" % (mean["SQ_WAIT_ANY"] / mean["SQ_WAVE_CYCLES"]))
    if "SQ_WAIT_INST_ANY" in mean and "SQ_WAVE_CYCLES" in mean:
        print("issue-stall share of wave cycles (SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES): %.3f"
              % (mean["SQ_WAIT_INST_ANY"] / mean["SQ_WAVE_CYCLES"]))


if __name__ == "__main__":
    if len(sys.argv) >= 3 and sys.argv[1] == "sq":
        sq(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 8192 * 8192)
    elif len(sys.argv) >= 3 and sys.argv[1] == "stats":
        stats(sys.argv[2])
    elif len(sys.argv) >= 4 and sys.argv[1] == "pmc":
        pmc(sys.argv[2], sys.argv[3])
    else:
        sys.exit(__doc__)
