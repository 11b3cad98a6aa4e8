#!/bin/bash
export EPIC_HIP_STUDY=1   # the knobs below are study knobs: read only when asked for (epic_amd/csrc/driver_config.cpp)
# Experiment: the share of due tiles above which harmonic_execute_gpu runs a batch without the work lists (EPIC_HIP_TRACK_SWITCH),
# whole 8192^2 relaxations, same box.  --track 2 = automatic (the library's default).
#   CONFIGS="tol:jacobi:0.6,0.7 precise:redblack:0.5" bash tools/exp_track_switch.sh
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for cfg in ${CONFIGS:-tol:jacobi:0.5,0.6,0.7,0.8,0.9 tol:redblack:0.5,0.6,0.7,0.8,0.9 precise:redblack:0.5,0.6,0.7,0.8,0.9}; do
  IFS=: read -r math scheme list <<< "$cfg"
  for sw in ${list//,/ }; do
    s=$(EPIC_HIP_TRACK_SWITCH=$sw python3 $ROOT/tools/time_relax.py --math $math --scheme $scheme --track 2 --repeat 2 2>/dev/null | python3 -c 'import sys,json; print(" ".join(str(json.loads(l)["seconds"]) for l in sys.stdin if l.startswith("{")))')
    echo "$math $scheme switch $sw: $s s"
  done
done
