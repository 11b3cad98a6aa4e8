#!/bin/bash
export EPIC_HIP_STUDY=1   # the knobs below are study knobs: read only when asked for (epic_amd/csrc/driver_config.cpp)
# Tracked red-black pairs (round 4): whole 8192^2 default relaxations side by side in ONE gpurun call.
#   bash tools/exp_pairs.sh   -> seconds for: half-sweeps (round 3's path), pairs at several task heights and bypass thresholds
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
R="python3 tools/time_relax.py --scheme redblack --math precise --track 2 --repeat 2"
run() { echo "== $*"; env "$@" $R 2>/dev/null | tail -1; }
run EPIC_HIP_TRACK_PAIRS=0
for rows in ${ROWS:-8 12 16 24 32}; do run EPIC_HIP_TRACK_PAIR_ROWS=$rows; done
for sw in ${SWITCH:-0.6 0.75 0.95 2}; do run EPIC_HIP_TRACK_SWITCH=$sw; done
