#!/bin/bash
# Builds variants of the library that differ in the 3-D kernels' build knobs only (kernels_3d.hip recompiled, every other object
# taken from the main build): gpurun_alt/<name>/libepic.so, for same-call A/Bs on the GPU box (tools/exp_3d_time.sh).
#   bash tools/build_3d_variants.sh name1 "flags1" name2 "flags2" ...
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CS=$ROOT/epic_amd/csrc
make -s -C "$CS" > /dev/null
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-honor-nans -mllvm -amdgpu-set-wave-priority"
while [ $# -ge 2 ]; do
  name=$1; extra=$2; shift 2
  d=$ROOT/gpurun_alt/$name; mkdir -p "$d"
  /opt/rocm/bin/hipcc $FLAGS $extra -Rpass-analysis=kernel-resource-usage -c "$CS/kernels_3d.hip" -o "$d/kernels_3d.o" 2> "$d/resource.txt"
  objs=$(ls "$CS"/build/*.o | grep -v kernels_3d.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$d/libepic.so" $objs "$d/kernels_3d.o"
  echo "$name [$extra]: $(grep -A12 'sweep3d_pair_kernelILb0ELb0ELb0' "$d/resource.txt" | grep -E 'VGPRs:|Scratch|Occupancy|LDS Size' | sed 's/.*remark: *//' | tr '\n' ' ')"
done
