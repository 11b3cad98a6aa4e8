#!/bin/bash
# SQ counters of the tile kernel on maps/maze.png for the shipped library and for variants in gpurun_alt/ (same call)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/exp_tile_r05; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
M="$ROOT/tools/time_maps.py --modes default --tile 1 --repeat 1 --maps maze"
for which in shipped "$@"; do
  lib=$ROOT/epic_amd/lib/libepic.so; [ "$which" != shipped ] && lib=$ROOT/gpurun_alt/$which/libepic.so
  i=0
  for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
             "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT"; do
    i=$((i+1)); d=$OUT/${which}_$i; rm -rf "$d"
    EPIC_LIB=$lib rocprofv3 --pmc $grp --output-format csv -d "$d" -- python3 $M > "$d.log" 2>&1
    echo "== $which, group $i"
    python3 - "$d" <<'PY'
import csv,glob,sys,statistics,collections
fs=glob.glob(sys.argv[1]+'/**/*_counter_collection.csv',recursive=True)
if not fs: print("   no output"); sys.exit(0)
v=collections.defaultdict(list)
for r in csv.DictReader(open(fs[0])):
    if 'tile2d_kernel' in r['Kernel_Name']: v[r['Counter_Name']].append(float(r['Counter_Value']))
for k,x in sorted(v.items()): print("   %-28s mean %.5g per launch (%d dispatches)" % (k, statistics.mean(x), len(x)))
PY
    find "$d" -name "*.csv" -size +2M -delete
  done
done
