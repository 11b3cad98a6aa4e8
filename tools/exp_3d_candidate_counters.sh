#!/bin/bash
# Round 5: the stall counters of the three-waves-per-SIMD candidate of the 3-D tol sweep (gpurun_alt/$1) beside the shipped kernel's:
# the wave-time split and the L2 round trip, two rocprofv3 --pmc passes per library.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/exp3d_r05; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
C="$ROOT/tools/bench_config.py --grid 512 512 512 --develop 600 --math tol --sweeps 60"
for which in shipped $1; do
  lib=$ROOT/epic_amd/lib/libepic.so; [ "$which" != shipped ] && lib=$ROOT/gpurun_alt/$which/libepic.so
  i=0
  for grp in "SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_LDS" \
             "TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
    i=$((i+1)); d=$OUT/cand_${which}_$i; rm -rf "$d"
    EPIC_LIB=$lib rocprofv3 --pmc $grp --output-format csv -d "$d" -- python3 $C > "$d.log" 2>&1
    echo "== $which, group $i"
    python3 - "$d" <<'PY'
import csv,glob,sys,statistics,collections
fs=glob.glob(sys.argv[1]+'/**/*_counter_collection.csv',recursive=True)
if not fs: print("   no output"); sys.exit(0)
v=collections.defaultdict(list)
for r in csv.DictReader(open(fs[0])):
    if 'sweep3d_pair_kernel<false' in r['Kernel_Name'] or 'sweep3d_pair_kernelILb0' in r['Kernel_Name']: v[r['Counter_Name']].append(float(r['Counter_Value']))
for k,x in sorted(v.items()): print("   %-34s mean %.5g per sweep (%d dispatches)" % (k, statistics.mean(x), len(x)))
PY
    find "$d" -name "*.csv" -size +4M -delete
  done
done
