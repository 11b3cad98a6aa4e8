#!/bin/bash
# timing of build variants (gpurun_alt/<name>) against the shipped library + the 3-D parity tests through the first one
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/exp3d_r05; mkdir -p "$OUT"
CAND=$1
LIBS="epic_amd/lib/libepic.so"; for v in "$@"; do LIBS="$LIBS gpurun_alt/$v/libepic.so"; done
ROUNDS=${ROUNDS:-2} bash $ROOT/tools/exp_3d_time.sh $LIBS 2>&1 | grep -v amdgpu.ids | tee "$OUT/timing_$CAND.txt"
(cd $ROOT && EPIC_LIB=$ROOT/gpurun_alt/$CAND/libepic.so timeout -k 10 900 python3 -m pytest tests/test_gpu_tol.py tests/test_gpu_full_configs.py tests/test_gpu_whole_field.py -k "3d or 512 or cubed or plane" -x -q 2>&1 | tail -5) | tee "$OUT/parity_$CAND.txt"
