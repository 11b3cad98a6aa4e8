#!/bin/bash
export EPIC_HIP_STUDY=1   # the knobs below are study knobs: read only when asked for (epic_amd/csrc/driver_config.cpp)
# Tracked pairs with the tol passes: whole 8192^2 relaxations, pairs against list-driven single sweeps, ONE gpurun call.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
for scheme in redblack jacobi; do
  R="python3 tools/time_relax.py --scheme $scheme --math tol --track 2 --repeat 2"
  run() { echo "== $scheme $*"; env "$@" $R 2>/dev/null | tail -1; }
  run EPIC_HIP_TRACK_PAIRS=0
  run EPIC_HIP_TRACK_PAIRS=1
  for sw in ${SWITCH:-0.5 0.7 2}; do run EPIC_HIP_TRACK_SWITCH=$sw; done
done
