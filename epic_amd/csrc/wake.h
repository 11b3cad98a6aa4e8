// wake.h -- work lists of the activity-tracking sweeps (device side), shared by the 2-D and 3-D kernels.
//
// A tile (one task of a sweep kernel) has to be recomputed in iteration k + 1 only if iteration k changed one of the
// values it reads.  Otherwise its update would reproduce, bit for bit, the values already in place (Jacobi: in BOTH
// ping-pong buffers, because the tile itself did not change either; red-black: in place).  So every task that changed
// something WAKES the tiles that read it -- itself and the neighbours across the faces it changed -- by appending them
// to the work lists of the next iteration (`queued` marks keep a tile from being listed twice), and iteration k + 1 is
// a fixed-size launch of persistent waves that walk those lists: no wave is spent on a tile that has nothing to do, and
// the listed tiles spread evenly over the chip however they cluster in space.
//
// Layout decisions, all measured on the 8192^2 relaxation (profiles/r01_experiments.txt):
//  * one list with one counter serialises on that counter (7 ns per atomicAdd: 460 us per sweep), so there are
//    kWakeLists = 256 lists, each counter in a 128-byte line of its own;
//  * tile t is always listed in list t / list_cap (list_cap = ceil(tiles / 256) consecutive tiles), so a list cannot
//    overflow and its tiles are neighbours in memory;
//  * the consumer sees the 256 lists as one sequence (entries of list 0, then list 1, ...), cut into eight equal
//    segments, one per XCD (blocks are dealt round-robin over the XCDs, blockIdx % 8 labels them): the waves of an XCD
//    take the elements of its segment in turn -- even work, and tiles of one band meet in one L2.
// list_in == nullptr: every tile runs (the first two iterations after any edit of u, mask or mode) and the launch
// covers the grid like an untracked one, still waking tiles for its successor.  Three counter sets rotate: a launch
// reads count_in, fills count_out and resets count_zero, which the launch after next will fill.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"

namespace epic_hip {

constexpr int kWakeLists = kWakeListCount;          // = threads per block of the sweep kernels (4 waves)
constexpr int kWakeStride = kWakeCounterStride;     // words between two list counters
constexpr int kWakeXcds = 8;

struct WakeArgs {
    const uint32_t *list_in;    // kWakeLists x list_cap tile ids; list i holds count_in[i * kWakeStride] of them
    const uint32_t *count_in;
    uint32_t *list_out;         // tiles woken for the next launch
    uint32_t *count_out;
    uint32_t *count_zero;
    uint32_t *queued_in;        // marks of the tiles listed for this launch: cleared as they are taken
    uint32_t *queued_out;       // 1 = already in list_out
    unsigned long long *total;  // += the number of tiles listed for this launch (list-driven launches only)
    int list_cap;
};

inline WakeArgs wake_args(const Activity *act, size_t tiles)
{
    WakeArgs w = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0};
    if (act && act->list_out) {
        w.list_in = act->list_in;
        w.count_in = act->count_in;
        w.list_out = act->list_out;
        w.count_out = act->count_out;
        w.count_zero = act->count_zero;
        w.queued_in = act->queued_in;
        w.queued_out = act->queued_out;
        w.total = act->total;
        w.list_cap = (int)sweep_2d_list_cap(tiles);
    }
    return w;
}

// Position of a wave in the sequence of listed tiles (all members wave-uniform except the per-lane counters).
struct WakeCursor {
    uint32_t c0, c1, c2, c3, incl;  // lane l: counters 4 l .. 4 l + 3 and the inclusive prefix sum over the lanes
    int g, end, step;               // current element, end of this XCD's segment, waves per XCD
};

// Reset the counter set of the launch after next.  Call from every thread of the kernel (block 0 does the work).
__device__ __forceinline__ void wake_reset_next(const WakeArgs &w)
{
    if (blockIdx.x == 0 && threadIdx.x < kWakeLists) w.count_zero[threadIdx.x * kWakeStride] = 0;  // blocks have >= kWakeLists threads
}

// List-driven launch: find this wave's first element.  false = nothing to do for this wave.
__device__ __forceinline__ bool wake_begin(const WakeArgs &w, int lane, int wave, int waves_per_block, WakeCursor &k)
{
    const int xcd = blockIdx.x % kWakeXcds;
    const uint32_t *mine = w.count_in + (size_t)(4 * lane) * kWakeStride;
    k.c0 = mine[0]; k.c1 = mine[kWakeStride]; k.c2 = mine[2 * kWakeStride]; k.c3 = mine[3 * kWakeStride];
    k.incl = k.c0 + k.c1 + k.c2 + k.c3;
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t up = (uint32_t)__shfl_up((int)k.incl, d, 64);
        if (lane >= d) k.incl += up;
    }
    const long long all = __builtin_amdgcn_readlane((int)k.incl, 63);
    if (w.total && blockIdx.x == 0 && wave == 0 && lane == 0) atomicAdd(w.total, (unsigned long long)all);
    k.step = (gridDim.x / kWakeXcds) * waves_per_block;
    k.g = (blockIdx.x / kWakeXcds) * waves_per_block + wave + (int)(all * xcd / kWakeXcds);
    k.end = (int)(all * (xcd + 1) / kWakeXcds);
    return k.g < k.end;
}

// The tile at the cursor: the first lane whose inclusive prefix exceeds g holds its list among its four.
__device__ __forceinline__ int wake_tile(const WakeArgs &w, const WakeCursor &k)
{
    const int L = __popcll(__ballot(k.incl <= (uint32_t)k.g));
    const uint32_t l0 = (uint32_t)__builtin_amdgcn_readlane((int)k.c0, L), l1 = (uint32_t)__builtin_amdgcn_readlane((int)k.c1, L);
    const uint32_t l2 = (uint32_t)__builtin_amdgcn_readlane((int)k.c2, L), l3 = (uint32_t)__builtin_amdgcn_readlane((int)k.c3, L);
    uint32_t r = (uint32_t)k.g - ((uint32_t)__builtin_amdgcn_readlane((int)k.incl, L) - (l0 + l1 + l2 + l3));
    int list = 4 * L;
    if (r >= l0) { r -= l0; list++; if (r >= l1) { r -= l1; list++; if (r >= l2) { r -= l2; list++; } } }
    return __builtin_amdgcn_readfirstlane((int)w.list_in[(size_t)list * w.list_cap + r]);
}

__device__ __forceinline__ bool wake_next(WakeCursor &k)
{
    k.g += k.step;
    return k.g < k.end;
}

// Per lane: wake tile t for the next launch if `want` (first waker appends it to its list).
__device__ __forceinline__ void wake_push(const WakeArgs &w, int t, bool want)
{
    if (want && atomicExch(&w.queued_out[t], 1u) == 0u) {
        const unsigned li = (unsigned)t / (unsigned)w.list_cap;
        w.list_out[(size_t)li * w.list_cap + atomicAdd(&w.count_out[(size_t)li * kWakeStride], 1u)] = (uint32_t)t;
    }
}

}  // namespace epic_hip
