#!/bin/bash
# Experiment: where the 3-D tol sweep's reads are served from (L1 = TCP, L2 = TCC, HBM) -- one rocprofv3 --pmc pass per counter group
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_READ_sum TCC_WRITE_sum"; do
  i=$((i+1)); d=$ROOT/gpurun_out/exp3d_cache_$i; rm -rf "$d"
  rocprofv3 --pmc $grp --output-format csv -d "$d" -- python3 $ROOT/tools/bench_config.py --grid 512 512 512 --math tol --develop 300 --sweeps 60 > "$d.log" 2>&1
  python3 - "$d" <<'PY'
import csv,glob,sys,statistics,collections
fs=glob.glob(sys.argv[1]+'/**/*_counter_collection.csv',recursive=True)
if not fs: print("no output for", sys.argv[1]); sys.exit(0)
v=collections.defaultdict(list)
for r in csv.DictReader(open(fs[0])):
    if 'sweep3d' in r['Kernel_Name']: v[r['Counter_Name']].append(float(r['Counter_Value']))
for k,x in v.items(): print("%-32s mean %.4g per sweep (%d dispatches)" % (k, statistics.mean(x), len(x)))
PY
  find "$d" -name "*.csv" -size +4M -delete
done
