#!/bin/bash
export EPIC_HIP_STUDY=1   # the knobs below are study knobs: read only when asked for (epic_amd/csrc/driver_config.cpp)
# does a task height that fills the last round exactly pay?  8190 rows = 210 chunks of 39 (6.97 blocks per CU) against 205 chunks of 40 (6.81)
for spec in "8190 39" "8190 40" "8190 39" "8190 40" "8188 46" "8188 45"; do
  set -- $spec
  EPIC_HIP_FUSED_ROWS=$2 python3 bench.py --size $1 --no-cpu --no-relax --no-extra-legs --no-parity --no-live-traffic --no-config4 --no-maps --steps 5 --warmup 1 --develop 8000 2>/dev/null | python3 -c "
import sys,json
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('size $1 rows $2: launch %.1f us  frac %.4f' % (r['roofline']['launch_us'], r['roofline']['frac']))"
done
