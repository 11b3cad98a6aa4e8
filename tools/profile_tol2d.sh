#!/bin/bash
# Kernel stats + SQ counters of the benchmarked 2-D tol configuration only (subset of profile_round.sh): bash tools/profile_tol2d.sh TAG
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B="$ROOT/bench.py --no-cpu --no-relax --no-extra-legs --no-parity --no-live-traffic"
SQ="SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY GRBM_GUI_ACTIVE"
run() { name=$1; shift; "$@" > "$OUT/$name.log" 2>&1; echo "[$name] rc=$?"; }
run stats_tol_jacobi rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_tol_jacobi" -- python3 $B --steps 5 --warmup 1
run sq_tol rocprofv3 --pmc $SQ --output-format csv -d "$OUT/sq_tol_jacobi" -- python3 $B --steps 1 --warmup 1 --develop 2000
cd "$ROOT"
S="python3 tools/summarize_profile.py"
$S stats "$OUT/stats_tol_jacobi" > "$OUT/${TAG}_kernel_stats_tol_jacobi.txt"
PROFILE_KERNEL=jacobi_fused2d $S sq "$OUT/sq_tol_jacobi" 134217728 > "$OUT/${TAG}_sq_counters_tol_fused.txt" 2>&1
PROFILE_KERNEL=sweep2d $S sq "$OUT/sq_tol_jacobi" 67108864 > "$OUT/${TAG}_sq_counters_tol.txt" 2>&1
find "$OUT" -name "*kernel_trace.csv" -size +8M -delete
find "$OUT" -name "*counter_collection.csv" -size +8M -delete
cat "$OUT"/${TAG}_kernel_stats_tol_jacobi.txt "$OUT"/${TAG}_sq_counters_tol_fused.txt
