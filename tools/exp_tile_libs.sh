#!/bin/bash
# wall seconds of the reference's small maps through harmonic_complete_gpu (library defaults and tol red-black) for several builds of the
# library, alternating, same call:   bash tools/exp_tile_libs.sh <lib.so> <lib.so> ...   (paths relative to the repo root)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for round in 1 2; do
  for lib in "$@"; do
    EPIC_LIB=$ROOT/$lib python3 $ROOT/tools/time_maps.py --maps ${MAPS:-basic,maze,umass} --modes default,tol_rb --tile 1 --eps 1e-6 --repeat 2 2>/dev/null | grep -v "^\[{" | sed "s|^|round $round $lib: |" | cut -c1-200
  done
done
