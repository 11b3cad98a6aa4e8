#!/bin/bash
# Profiles kept under profiles/ for a round (run on the GPU box through gpurun; TAG = r01, r02, ...):
#   bash tools/profile_round.sh r01
# rocprofv3 passes, each with the program itself after `--` (python3 <script>), counters in their own runs:
#   1. --kernel-trace --stats of the bench.py command (untracked precise Jacobi, developed field)
#   2. the same for the red-black scheme and for a complete tracked relaxation (tools/time_relax.py)
#   3. --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes, short bench run
#   4. --pmc SQ counters, short bench run
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B="$ROOT/bench.py --no-cpu --no-relax"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_precise_jacobi" -- python3 $B --steps 5 --warmup 1 > "$OUT/stats_precise_jacobi.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_precise_redblack" -- python3 $B --steps 5 --warmup 1 --scheme redblack > "$OUT/stats_precise_redblack.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_relax_tracked" -- python3 $ROOT/tools/time_relax.py > "$OUT/stats_relax_tracked.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch_precise_jacobi" -- python3 $B --steps 1 --warmup 1 --develop 200 > "$OUT/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write_precise_jacobi" -- python3 $B --steps 1 --warmup 1 --develop 200 > "$OUT/write.log" 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d "$OUT/sq_precise_jacobi" -- python3 $B --steps 1 --warmup 1 --develop 200 > "$OUT/sq.log" 2>&1
cd "$ROOT"
python3 tools/summarize_profile.py stats "$OUT/stats_precise_jacobi" > "$OUT/${TAG}_kernel_stats_precise_jacobi.txt"
python3 tools/summarize_profile.py stats "$OUT/stats_precise_redblack" > "$OUT/${TAG}_kernel_stats_precise_redblack.txt"
python3 tools/summarize_profile.py stats "$OUT/stats_relax_tracked" > "$OUT/${TAG}_kernel_stats_relax_tracked.txt"
python3 tools/summarize_profile.py pmc "$OUT/fetch_precise_jacobi" "$OUT/write_precise_jacobi" > "$OUT/${TAG}_hbm_traffic_precise_jacobi.txt"
python3 tools/summarize_profile.py sq "$OUT/sq_precise_jacobi" > "$OUT/${TAG}_sq_counters_precise.txt" 2>&1
# only the summaries travel back in full; the raw CSVs of the long runs are large
find "$OUT" -name "*kernel_trace.csv" -size +8M -delete
tail -n 3 "$OUT"/*.log
cat "$OUT"/${TAG}_*.txt
