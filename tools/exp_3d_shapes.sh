#!/bin/bash
# Experiment: does the 3-D sweep's memory side depend on the plane / row strides being powers of two?  ns per kcell by grid shape.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for m in tol traffic; do
for g in "512 512 512" "512 513 512" "512 509 512" "512 516 512" "512 512 768" "512 384 768" "504 520 512" "384 640 512"; do
  python3 $ROOT/tools/bench_config.py --grid $g --math $m --develop 600 --sweeps 200 | python3 -c '
import sys,json; r=json.loads(sys.stdin.read()); print("%-8s %-16s %8.2f us per sweep  %.4f of 8 TB/s  %.3f ns per kcell" % (r["math"], r["grid"], r["us_per_sweep"], r["frac_of_8TBps"], r["us_per_sweep"]*1e6/r["cells"]))'
done
done
