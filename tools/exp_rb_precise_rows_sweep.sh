#!/bin/bash
export EPIC_HIP_STUDY=1   # the knobs below are study knobs: read only when asked for (epic_amd/csrc/driver_config.cpp)
# launch time of the default arithmetic's red-black pair pass (rb_fused2d_kernel, untracked) against the task height
for r in ${ROWS_LIST:-16 20 24 28 32 36 40 44 48 56 64}; do
  EPIC_HIP_FUSED_ROWS=$r python3 bench.py --math precise --scheme redblack --no-cpu --no-relax --no-extra-legs --no-parity --no-live-traffic --no-config4 --no-maps --steps 3 --warmup 1 --develop 4000 2>/dev/null | python3 -c "
import sys,json
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('rows $r: launch %.1f us  frac %.4f  kernel %s  rows_per_task %s' % (r['roofline']['launch_us'], r['roofline']['frac'], r['roofline'].get('kernel'), r['config'].get('fused_rows_per_task')))"
done
