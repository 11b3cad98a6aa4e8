export EPIC_HIP_STUDY=1   # the knobs below are study knobs: read only when asked for (epic_amd/csrc/driver_config.cpp)
for r in ${ROWS_LIST:-30 32 34 35 36 38 40 41 42 44 46 48 56}; do
  EPIC_HIP_FUSED_ROWS=$r python3 bench.py --no-cpu --no-relax --no-extra-legs --no-parity --no-live-traffic --no-config4 --no-maps --steps 5 --warmup 1 --develop 8000 2>/dev/null | python3 -c "
import sys,json
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('rows $r: launch %.1f us  frac %.4f  rows_per_task %s' % (r['roofline']['launch_us'], r['roofline']['frac'], r['config'].get('fused_rows_per_task')))"
done
