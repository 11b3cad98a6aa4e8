#!/bin/bash
# Per-rank kernel + copy traces of the multi-GPU bench, so that the overlap of the halo exchange (second stream) with the
# interior sweep can be read from a timeline:   bash tools/profile_scale.sh "2 4 8" [extra bench.py flags]
# Needs a node with that many GPUs -- NOT RUN ON HARDWARE YET (the build sessions have one GPU); the driver's scaling run is
# the first multi-GPU execution of bench.py.  Written to the pool's rules: every rank is its own `rocprofv3 ... -- python3
# bench.py` (the launcher, torch.distributed.run --no-python, touches no GPU before it spawns them), kernel / memory-copy
# traces only (no --pmc in the same run).
#   gpurun_out/scale_N/<pid>_kernel_stats.csv     per rank: sweep2d_kernel vs RCCL kernels (ncclDevKernel_*)
#   gpurun_out/scale_N/<pid>_kernel_trace.csv     start / end per dispatch and queue: second-stream kernels overlap the
#                                                 interior sweep iff their intervals intersect
#   gpurun_out/scale_N/<pid>_memory_copy_trace.csv  hipMemcpyPeerAsync of the in_library leg
NS=${1:-"2 4 8"}
shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for N in $NS; do
    OUT=$ROOT/gpurun_out/scale_$N
    mkdir -p "$OUT"
    python3 -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port $((29600 + N)) \
        --no-python rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d "$OUT" -o "%pid%" -- \
        python3 "$ROOT/bench.py" --gpus "$N" --steps 3 --warmup 1 --develop 1000 --no-cpu "$@" > "$OUT/bench.json" 2> "$OUT/bench.err"
    echo "[N=$N] rc=$?"
    tail -c 1500 "$OUT/bench.json"
    find "$OUT" -name "*kernel_trace.csv" -size +16M -delete
done
