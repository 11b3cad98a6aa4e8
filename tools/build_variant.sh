#!/bin/bash
# A variant of the library that differs in the build knobs of ONE kernel source (recompiled; every other object from the main build):
# gpurun_alt/<name>/libepic.so, for same-call A/Bs on the GPU box (EPIC_LIB=...; tools/exp_3d_time.sh, tools/time_maps.py).
#   bash tools/build_variant.sh <kernels_3d|kernels_2d|kernels_tile2d> name1 "flags1" name2 "flags2" ...
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CS=$ROOT/epic_amd/csrc
SRC=$1; shift
make -s -C "$CS" > /dev/null
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-honor-nans -mllvm -amdgpu-set-wave-priority"
while [ $# -ge 2 ]; do
  name=$1; extra=$2; shift 2
  d=$ROOT/gpurun_alt/$name; mkdir -p "$d"
  /opt/rocm/bin/hipcc $FLAGS $extra -Rpass-analysis=kernel-resource-usage -c "$CS/$SRC.hip" -o "$d/$SRC.o" 2> "$d/resource.txt"
  objs=$(ls "$CS"/build/*.o | grep -v "/$SRC.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$d/libepic.so" $objs "$d/$SRC.o"
  python3 -c "import ctypes; ctypes.CDLL('$d/libepic.so')"   # (an undefined kernel handle shows here, not on the GPU box)
  echo "$name [$extra]: built"
done
