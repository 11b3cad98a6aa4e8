#!/bin/bash
# Experiment: time of the 3-D tol sweep (512^3, developed field) for builds in gpurun_alt/, several rounds so that drift shows
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
ROUNDS=${ROUNDS:-2}
for round in $(seq $ROUNDS); do
for lib in "$@"; do
  m=tol; l=$lib
  case "$lib" in *:*) m=${lib#*:}; l=${lib%%:*};; esac
  us=$(EPIC_LIB=$ROOT/$l python3 $ROOT/tools/bench_config.py --grid 512 512 512 --math $m --develop 1500 --sweeps 300 | python3 -c 'import sys,json; print(json.loads(sys.stdin.read())["us_per_sweep"])')
  echo "round $round  $l ($m)  ${us} us per sweep"
done
done
