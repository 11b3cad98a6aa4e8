#!/bin/bash
export EPIC_HIP_STUDY=1   # the knobs below are study knobs: read only when asked for (epic_amd/csrc/driver_config.cpp)
# LDS counters of the fused tol pass (8192^2, developed field) for the shipped library and a variant (gpurun_alt/<name>): item 11 of
# profiles/r05_experiments.txt.   bash tools/exp_lds_reads.sh k2d_b96
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/lds_reads; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B="$ROOT/bench.py --no-cpu --no-relax --no-extra-legs --no-parity --no-live-traffic --steps 1 --warmup 1 --develop 5000"
export EPIC_HIP_FUSED_ROWS=40
for v in shipped "$@"; do
  lib=$ROOT/epic_amd/lib/libepic.so; [ "$v" != shipped ] && lib=$ROOT/gpurun_alt/$v/libepic.so
  EPIC_LIB=$lib rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d "$OUT/$v" -- python3 $B > "$OUT/$v.log" 2>&1
  echo "== $v"; (cd $ROOT && PROFILE_LAST=90 PROFILE_KERNEL=jacobi_fused2d python3 tools/summarize_profile.py sq "$OUT/$v" 134217728 2>&1 | tail -16)
done
