#!/bin/bash
# Experiment: HBM read traffic and time of the 3-D tol sweep for builds in gpurun_alt/ (FETCH_SIZE PMC pass + a timed run each)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  export EPIC_LIB=$ROOT/$lib
  d=$ROOT/gpurun_out/exp3d_$(echo $lib | tr '/' '_'); rm -rf "$d"
  rocprofv3 --pmc ${PMC:-FETCH_SIZE} --output-format csv -d "$d" -- python3 $ROOT/tools/bench_config.py --grid 512 512 512 --math tol --develop 300 --sweeps 100 > /dev/null 2>&1
  rd=$(python3 - "$d" <<'PY'
import csv,glob,sys,statistics
f=glob.glob(sys.argv[1]+'/**/*_counter_collection.csv',recursive=True)[0]
v=[float(r['Counter_Value']) for r in csv.DictReader(open(f)) if r["Counter_Name"] in ("FETCH_SIZE","WRITE_SIZE") and "sweep3d" in r['Kernel_Name']]
print('%.1f' % (statistics.mean(v)*1024*2/1e6))
PY
)
  us=$(python3 $ROOT/tools/bench_config.py --grid 512 512 512 --math tol --develop 1500 --sweeps 300 | python3 -c 'import sys,json; print(json.loads(sys.stdin.read())["us_per_sweep"])')
  echo "$lib  read ${rd} MB per sweep   ${us} us per sweep"
  find "$d" -name "*.csv" -size +4M -delete
done
