#!/bin/bash
export EPIC_HIP_STUDY=1   # the knobs below are study knobs: read only when asked for (epic_amd/csrc/driver_config.cpp)
# Experiment: task height of the fused tol pass (EPIC_HIP_FUSED_ROWS) on the bench workload; one gpurun call, same box.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/exp_fused_rows.txt
: > "$OUT"
for rows in ${ROWS_LIST:-0 23 35 69 70 0}; do
  if [ "$rows" = 0 ]; then unset EPIC_HIP_FUSED_ROWS; else export EPIC_HIP_FUSED_ROWS=$rows; fi
  line=$(python3 "$ROOT/bench.py" --no-cpu --no-relax --no-extra-legs --no-parity --no-live-traffic --steps 8 --warmup 2 2>/dev/null | tail -1)
  echo "rows=$rows $(echo "$line" | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["roofline"]["launch_us"], d["roofline"]["frac"], d["ms_per_step"])')" | tee -a "$OUT"
done
