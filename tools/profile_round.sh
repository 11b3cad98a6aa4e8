#!/bin/bash
export EPIC_HIP_STUDY=1   # the knobs below are study knobs: read only when asked for (epic_amd/csrc/driver_config.cpp)
# Profiles kept under profiles/ for a round (run on the GPU box through gpurun; TAG = r01, r02, ...):
#   bash tools/profile_round.sh r05
# rocprofv3 passes, each with the program itself after `--` (python3 <script>), counters in their own runs:
#   1. --kernel-trace --stats of the bench.py command (default math = tol, untracked Jacobi, developed field: pairs of
#      iterations as jacobi_fused2d_kernel, the check and the odd iteration as sweep2d_kernel) and of
#      the same with --math precise; of the 512^3 sweeps (tools/bench_config.py, developed field), tol and precise
#   2. --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes: 2-D tol, 3-D tol, 3-D precise
#   3. --pmc SQ counters: 2-D tol, 2-D precise, 3-D tol, 3-D precise
#   4. --kernel-trace --stats of whole relaxations with activity tracking (tools/time_relax.py): tol Jacobi, and the library
#      default (precise, red-black) -- the list-driven kernels and the bypassed batches
#   bash tools/profile_round.sh r03 3d      only the passes whose name contains "3d" (after a change to the 3-D kernel)
TAG=${1:-r06}
ONLY=${2:-}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B="$ROOT/bench.py --no-cpu --no-relax --no-extra-legs --no-parity --no-live-traffic --no-node-flow"
C="$ROOT/tools/bench_config.py --grid 512 512 512 --develop 1500"
SQ="SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY GRBM_GUI_ACTIVE"
run() { name=$1; shift; if [ -n "$ONLY" ] && ! [[ "$name" =~ $ONLY ]]; then return 0; fi; "$@" > "$OUT/$name.log" 2>&1; echo "[$name] rc=$?"; }   # ONLY: a regex over the pass names
# the library measures the task height of its fused passes by timing, which counter passes distort: learn it from a plain run
# and fix it for every 2-D tol pass of this script
ROWS=$(python3 $B --steps 1 --warmup 1 --develop 5000 2>/dev/null | python3 -c 'import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])["config"].get("fused_rows_per_task", 0))' 2>/dev/null)
if [ -n "$ROWS" ] && [ "$ROWS" -gt 0 ] 2>/dev/null; then export EPIC_HIP_FUSED_ROWS=$ROWS; echo "[fused rows per task: $ROWS]"; fi
run stats_tol_jacobi      rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_tol_jacobi" -- python3 $B --steps 5 --warmup 1
run stats_precise_jacobi  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_precise_jacobi" -- python3 $B --steps 5 --warmup 1 --math precise
run stats_3d_tol          rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_3d_tol" -- python3 $C --math tol --sweeps 300
run stats_3d_precise      rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_3d_precise" -- python3 $C --math precise --sweeps 300
R="$ROOT/tools/time_relax.py --track 2"
run stats_relax_tol      rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_relax_tol" -- python3 $R --math tol --scheme jacobi
run stats_relax_tol_rb   rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_relax_tol_rb" -- python3 $R --math tol --scheme redblack
run stats_relax_default  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_relax_default" -- python3 $R --math precise --scheme redblack
#   5. (round 4) the reference's maps through harmonic_complete_gpu with the library's defaults: tile2d_kernel (several iterations per
#      launch on LDS tiles) -- kernel stats, and the SQ counters of the same command
M="$ROOT/tools/time_maps.py --modes default --tile 1 --repeat 1"
#   6. (round 6) the navigation node's call loop on maze.png (tools/time_node_flow.py): the kernels the fine-grained API reaches with the
#      deferred blocks, and with one launch per call (EPIC_HIP_DEFER=0)
N="$ROOT/tools/time_node_flow.py --map maze --iterations 20000 --steps 50"
run stats_node_flow  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_node_flow" -- python3 $N
EPIC_HIP_DEFER=0 run stats_node_flow_undeferred  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_node_flow_undeferred" -- python3 $N
run stats_maps  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_maps" -- python3 $M
run sq_maps     rocprofv3 --pmc $SQ --output-format csv -d "$OUT/sq_maps" -- python3 $M --maps maze
run fetch_tol   rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch_tol_jacobi" -- python3 $B --steps 1 --warmup 1 --develop 5000
run write_tol   rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write_tol_jacobi" -- python3 $B --steps 1 --warmup 1 --develop 5000
run fetch_3d_tol   rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch_3d_tol" -- python3 $C --math tol --sweeps 100
run write_3d_tol   rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write_3d_tol" -- python3 $C --math tol --sweeps 100
run fetch_3d_precise   rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch_3d_precise" -- python3 $C --math precise --sweeps 100
run write_3d_precise   rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write_3d_precise" -- python3 $C --math precise --sweeps 100
run sq_tol      rocprofv3 --pmc $SQ --output-format csv -d "$OUT/sq_tol_jacobi" -- python3 $B --steps 1 --warmup 1 --develop 5000
run sq_precise  rocprofv3 --pmc $SQ --output-format csv -d "$OUT/sq_precise_jacobi" -- python3 $B --steps 1 --warmup 1 --develop 5000 --math precise
run sq_3d_tol      rocprofv3 --pmc $SQ --output-format csv -d "$OUT/sq_3d_tol" -- python3 $C --math tol --sweeps 100
run sq_3d_precise  rocprofv3 --pmc $SQ --output-format csv -d "$OUT/sq_3d_precise" -- python3 $C --math precise --sweeps 100
cd "$ROOT"
S="sum1"
sum1() { for a in "$@"; do case "$a" in "$OUT"/*) [ -d "$a" ] || { echo "(pass not run in this call)"; return 0; };; esac; done; python3 tools/summarize_profile.py "$@"; }
keep() { [ -s "$1" ] && ! grep -q "pass not run in this call" "$1" || rm -f "$1"; }
$S stats "$OUT/stats_tol_jacobi" > "$OUT/${TAG}_kernel_stats_tol_jacobi.txt"
$S stats "$OUT/stats_precise_jacobi" > "$OUT/${TAG}_kernel_stats_precise_jacobi.txt"
$S stats "$OUT/stats_3d_tol" > "$OUT/${TAG}_kernel_stats_3d_tol.txt"
$S stats "$OUT/stats_3d_precise" > "$OUT/${TAG}_kernel_stats_3d_precise.txt"
$S stats "$OUT/stats_relax_tol" > "$OUT/${TAG}_kernel_stats_relax_tol_tracked.txt"
$S stats "$OUT/stats_relax_default" > "$OUT/${TAG}_kernel_stats_relax_default.txt"
$S stats "$OUT/stats_relax_tol_rb" > "$OUT/${TAG}_kernel_stats_relax_tol_redblack.txt"
$S stats "$OUT/stats_maps" > "$OUT/${TAG}_kernel_stats_maps_tiles.txt"
$S stats "$OUT/stats_node_flow" > "$OUT/${TAG}_kernel_stats_node_flow_maze.txt"
$S stats "$OUT/stats_node_flow_undeferred" > "$OUT/${TAG}_kernel_stats_node_flow_maze_undeferred.txt"
PROFILE_KERNEL=tile2d $S sq "$OUT/sq_maps" 232324 > "$OUT/${TAG}_sq_counters_maze_tiles.txt" 2>&1
PROFILE_LAST=90 PROFILE_KERNEL=jacobi_fused2d $S pmc "$OUT/fetch_tol_jacobi" "$OUT/write_tol_jacobi" > "$OUT/${TAG}_hbm_traffic_tol_jacobi_fused.txt"
PROFILE_KERNEL=sweep2d $S pmc "$OUT/fetch_tol_jacobi" "$OUT/write_tol_jacobi" > "$OUT/${TAG}_hbm_traffic_tol_jacobi.txt"
$S pmc "$OUT/fetch_3d_tol" "$OUT/write_3d_tol" > "$OUT/${TAG}_hbm_traffic_3d_tol.txt"
$S pmc "$OUT/fetch_3d_precise" "$OUT/write_3d_precise" > "$OUT/${TAG}_hbm_traffic_3d_precise.txt"
PROFILE_LAST=90 PROFILE_KERNEL=jacobi_fused2d $S sq "$OUT/sq_tol_jacobi" 134217728 > "$OUT/${TAG}_sq_counters_tol_fused.txt" 2>&1
PROFILE_KERNEL=sweep2d $S sq "$OUT/sq_tol_jacobi" 67108864 > "$OUT/${TAG}_sq_counters_tol.txt" 2>&1
$S sq "$OUT/sq_precise_jacobi" 67108864 > "$OUT/${TAG}_sq_counters_precise.txt" 2>&1
$S sq "$OUT/sq_3d_tol" 134217728 > "$OUT/${TAG}_sq_counters_3d_tol.txt" 2>&1
$S sq "$OUT/sq_3d_precise" 134217728 > "$OUT/${TAG}_sq_counters_3d_precise.txt" 2>&1
for f in "$OUT"/${TAG}_*.txt; do keep "$f"; done
# only the summaries travel back in full; the raw CSVs of the long runs are large
find "$OUT" -name "*kernel_trace.csv" -size +8M -delete
find "$OUT" -name "*counter_collection.csv" -size +8M -delete
[ -f "$OUT"/stats_tol_jacobi.log ] && tail -n 2 "$OUT"/stats_tol_jacobi.log | cut -c1-1500
cat "$OUT"/${TAG}_*.txt
