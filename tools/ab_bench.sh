#!/bin/bash
# A/B timing of library builds inside ONE gpurun call (boxes differ by several per cent, so only numbers from the same
# call compare):  bash tools/ab_bench.sh [bench args --] libA.so libB.so ...   prints launch_us per build, two rounds.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
ARGS="--no-cpu --no-relax --no-parity --no-live-traffic --steps 10 --warmup 2"
LIBS=()
for a in "$@"; do LIBS+=("$a"); done
for round in 1 2; do
  for lib in "${LIBS[@]}"; do
    out=$(EPIC_LIB="$ROOT/$lib" python3 "$ROOT/bench.py" $ARGS $AB_ARGS 2>/dev/null | tail -1)
    us=$(echo "$out" | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["roofline"]["launch_us"])' 2>/dev/null)
    echo "round $round  $lib  launch_us ${us:-FAILED}"
  done
done
