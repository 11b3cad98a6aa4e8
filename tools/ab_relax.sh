#!/bin/bash
# A/B of builds of the library on whole relaxations of the 8192^2 benchmark grid (harmonic_execute_gpu to eps = 1e-6, activity tracking on):
#   bash tools/ab_relax.sh <libA.so> <libB.so> ...     (paths relative to the repo root; alternating rounds so that drift shows)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
ROUNDS=${ROUNDS:-2}
MODES=${MODES:-"precise:redblack tol:redblack"}
for round in $(seq $ROUNDS); do
for lib in "$@"; do
for mode in $MODES; do
  m=${mode%%:*}; s=${mode#*:}
  out=$(EPIC_LIB=$ROOT/$lib python3 $ROOT/tools/time_relax.py --track 2 --math $m --scheme $s 2>/dev/null | tail -1 | python3 -c 'import sys,json; r=json.loads(sys.stdin.read()); print("%.3f s  %d iterations  delta %.3g" % (r["seconds"], r["iterations"], r["delta"]))')
  printf "round %s  %-44s %-8s %-9s %s\n" "$round" "$lib" "$m" "$s" "$out"
done
done
done
