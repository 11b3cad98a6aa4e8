#!/bin/bash
# Round 5: the 3-D tol sweep (sweep3d_pair_kernel, 512^3) -- where its cycles go, and what a third wave per SIMD is worth.
#   1. the counters this pool offers (rocprofv3 -L) -> gpurun_out/r05_counters_avail.txt
#   2. same-call timing of the build variants in gpurun_alt/ (tools/build_variant.sh kernels_3d) against the shipped library
#   3. the 3-D tol parity tests through the candidate variant (EPIC_LIB)
#   4. per-class stall counters of the shipped kernel and of the candidate: one rocprofv3 --pmc pass per group
#   bash tools/exp_3d_occupancy.sh <candidate variant name> <other variants ...>
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/exp3d_r05
mkdir -p "$OUT"
CAND=$1
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > "$OUT/counters_avail.txt" 2>&1
echo "[counters listed: $(grep -c . "$OUT/counters_avail.txt") lines]"
LIBS="epic_amd/lib/libepic.so"
for v in "$@"; do LIBS="$LIBS gpurun_alt/$v/libepic.so"; done
ROUNDS=${ROUNDS:-2} bash $ROOT/tools/exp_3d_time.sh $LIBS 2>&1 | tee "$OUT/timing.txt"
if [ -n "$CAND" ]; then
  (cd $ROOT && EPIC_LIB=$ROOT/gpurun_alt/$CAND/libepic.so timeout -k 10 600 python3 -m pytest tests/test_gpu_tol.py tests/test_gpu_full_configs.py tests/test_gpu_whole_field.py -k "3d or 512 or cubed or plane" -x -q 2>&1 | tail -5) | tee "$OUT/parity_$CAND.txt"
fi
C="$ROOT/tools/bench_config.py --grid 512 512 512 --develop 600 --math tol --sweeps 60"
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES SQ_WAVE_CYCLES" \
           "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" \
           "SQ_INST_CYCLES_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_THREAD_CYCLES_VALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_UNALIGNED_STALL" \
           "SQ_INSTS_VALU_TRANS SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32" \
           "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_BRANCH SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES SQ_LEVEL_WAVES" \
           "TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  for which in shipped $CAND; do
    lib=$ROOT/epic_amd/lib/libepic.so; [ "$which" != shipped ] && lib=$ROOT/gpurun_alt/$which/libepic.so
    d=$OUT/pmc_${which}_$i; rm -rf "$d"
    EPIC_LIB=$lib rocprofv3 --pmc $grp --output-format csv -d "$d" -- python3 $C > "$d.log" 2>&1
    echo "== group $i ($which): $grp" | tee -a "$OUT/counters_$which.txt"
    python3 - "$d" <<'PY' | tee -a "$OUT/counters_$which.txt"
import csv,glob,sys,statistics,collections
fs=glob.glob(sys.argv[1]+'/**/*_counter_collection.csv',recursive=True)
if not fs: print("   no output (a counter of this group is not offered here?)"); sys.exit(0)
v=collections.defaultdict(list)
for r in csv.DictReader(open(fs[0])):
    if 'sweep3d_pair_kernelILb0' in r['Kernel_Name'] or ('sweep3d_pair_kernel<false' in r['Kernel_Name']): v[r['Counter_Name']].append(float(r['Counter_Value']))
for k,x in sorted(v.items()): print("   %-34s mean %.5g per sweep (%d dispatches)" % (k, statistics.mean(x), len(x)))
PY
    tail -n 3 "$d.log" | grep -i "error\|invalid\|not" | head -3
    find "$d" -name "*.csv" -size +4M -delete
  done
done
