#!/bin/bash
# Round 5: a wave's passes of a tile step two at a time (shipped) against one after the other (gpurun_alt/tile_nopairs), same call:
# the reference's maps through harmonic_complete_gpu with the library's defaults, wall seconds (tools/time_maps.py), alternating.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for round in 1 2; do
  for lib in epic_amd/lib/libepic.so gpurun_alt/tile_nopairs/libepic.so; do
    for eps in 1e-6 1e-3; do
      EPIC_LIB=$ROOT/$lib python3 $ROOT/tools/time_maps.py --maps basic,maze,umass,trivial,maze_2,willow_garage --modes default,tol_rb --tile 1 --eps $eps --repeat 2 2>/dev/null | grep -v "^{" | sed "s|^|round $round $lib eps $eps: |"
    done
  done
done
