#!/bin/bash
# Small-grid tile path: variants side by side in ONE gpurun call (boxes differ).  bash tools/exp_tile.sh lib1.so lib2.so ...
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
for lib in "$@"; do
  echo "== $lib"
  EPIC_LIB="$ROOT/$lib" python3 tools/time_maps.py --modes ${MODES:-default} --maps ${MAPS:-basic,maze,umass} --tile ${TILE:-1} --halo ${HALO:-auto} --rows ${ROWS:-0} 2>/dev/null | grep -v "^\[" 
done
