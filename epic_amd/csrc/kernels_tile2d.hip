// kernels_tile2d.hip -- small 2-D grids: SEVERAL iterations per launch on tiles that live in LDS (gfx950).
//
// The maps the reference's callers actually relax (src/epic_nav_core_plugin.cpp:256 -> harmonic_gpu.cu:266-290: one
// launch and one device synchronisation per half-sweep) are 0.1-1 Mcell: a half-sweep is ~0.2 us of arithmetic for the
// whole chip and 2.5-3 us of launch latency, captured graph or not.  So the plain iterations between two convergence
// checks run here K at a time: a workgroup loads its tile plus a ring of H >= K ghost cells into LDS, performs K
// iterations on it without leaving the CU (one workgroup barrier per iteration), and writes the cells it owns into the
// other buffer.  Ghost cells are swept like owned ones; the outermost ring has no neighbours to be computed from, so one
// more ring goes stale per iteration from the outside in -- the scheme the slabs of the multi-device mode use between
// GPUs (driver_multi.hip), at CU scale: after K <= H iterations every owned cell holds exactly what K whole-grid
// iterations would have given it.  The arithmetic of a cell is cell_update.h's, the same functions on the same values
// as in kernels_2d.hip: bit-identical to the per-iteration kernels and, with the precise math and the red-black scheme,
// to harmonic_complete_cpu (tests/test_gpu_tile.py: goldens, iteration counts, delta, whole fields).
//
// Geometry.  A tile is 64 columns wide in LDS (one lane per column: every LDS access of a wave hits 64 different banks)
// and S_r = T_r + 2 H <= 64 rows tall; it owns the inner T_r x (64 - 2 H) cells.  Red-black: the cells of one colour in
// a PAIR of rows are exactly one per column, so a wave takes a row pair per pass, each lane the row of the pair that has
// its column's active cell.  Jacobi: a wave takes a row per pass, the tile ping-pongs between two LDS arrays.  Rows that
// have gone stale are skipped (wave-uniform), stale columns idle.  Ping-pong in global memory (in != out): a neighbour
// must still find the old values of the cells it loads as ghosts.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <algorithm>

#include "cell_update.h"
#include "kernels.h"

#ifndef EPIC_CONSTS_TILE   // build knob (A/B): where the precise routines' f64 constants live (cell_update.h: MathTab::consts)
#define EPIC_CONSTS_TILE(RB) kConstsPlain   // (kConstsKeep: -2.5 % VALU instructions, nothing measurable on the maps -- the tile kernel waits, it does not issue)
#endif
namespace epic_hip {

namespace {

#ifndef EPIC_TILE_WAVES  // build knob (A/B): waves per workgroup
#define EPIC_TILE_WAVES 16
#endif
constexpr int kTileWaves = EPIC_TILE_WAVES;
constexpr int kTileThreads = 64 * kTileWaves;
// A step is as long as its slowest wave, and a cell's update is one chain of ~80 dependent instructions with three LDS round
// trips in it (~0.8 us for a wave alone on its SIMD, measured): the tile's row pairs are therefore spread over SIXTEEN
// waves -- one or two passes each -- and the four waves of a SIMD cover each other's latencies (4 waves per workgroup: 18.3 us
// per launch of 8 iterations on maps/maze.png, 16 waves: 10.8; tools/tile_probe.py, profiles/r04_experiments.txt).
// Build knob, OFF: between two iterations a wave waits not for the whole workgroup but for the two waves that hold the rows next
// to its own (rows are dealt to the waves in turn, so the neighbours of wave w's rows belong to waves w - 1 and w + 1,
// cyclically; w may start iteration j + 1 once both have FINISHED iteration j -- then what it reads there is written and what it
// overwrites has been read).  The idea: with a barrier per iteration all sixteen waves read LDS, wait, compute and wait again in
// lockstep -- the SQ counters have the VALU issuing 45 % of the time and the waves at s_waitcnt for half of it
// (profiles/r04_sq_counters_maze_tiles.txt) -- and staggered waves would fill each other's waits.  Built, bit-identical (the
// whole tile suite), and SLOWER: maze 0.0823 s against 0.0762 s, umass 0.1564 against 0.1447, same call -- polling two LDS words
// per iteration costs more than the barrier, and neighbours one iteration apart do not stagger far.  Kept as the measured
// alternative (profiles/r04_experiments.txt item 1h).
#ifndef EPIC_TILE_WAVE_SYNC
#define EPIC_TILE_WAVE_SYNC 0
#endif
constexpr bool kTileWaveSync = EPIC_TILE_WAVE_SYNC != 0;
constexpr lmask kOddLanes = 0xaaaaaaaaaaaaaaaaull;

struct Tile2dArgs {
    const float *in;
    float *out;
    const uint32_t *maskw;   // lane masks (kernels.h), 1 = locked; border and padding are locked
    unsigned *delta_bits;    // null, or: max |du| of the LAST iteration of the launch over the owned cells (atomicMax on float bits)
    float *tile_delta;       // null, or: the same maximum per tile, one plain store each (no zeroing, no atomics; may be host memory)
    int rows, pitch;
    int halo, tile_rows;     // H, T_r; owned columns per tile = LDS tile width - 2 H
    int tiles_c;
    int steps;               // iterations of this launch, <= halo
    int parity;              // red-black: number of the first iteration & 1
};

// CW, RMAX: the LDS tile is 64 CW columns wide and at most RMAX rows tall.  (1, 64): the maps up to ~0.3 Mcell, where a tile of
// ~1000 owned cells gives every CU one.  (2, 128): grids of 1-4 Mcell (the reference's batch fixtures: willow_garage, the mines,
// maze_2 / maze_3) -- tiles of up to 100 x 100 owned cells, whose ghost rings cost 1.3-1.5 cells per owned cell instead of 2.3; a
// row (pair) is then two units of 64 lanes, dealt to the waves like the rest.
template <int MATH, bool RB, int CW, int RMAX>
__global__ __launch_bounds__(kTileThreads) void tile2d_kernel(Tile2dArgs a)
{
    constexpr bool TOL = MATH == kMathTol;
    constexpr int kBufs = RB ? 1 : 2;
    constexpr int kTileCols = 64 * CW;
    constexpr int kTileMaxRows = RMAX;
    constexpr int kUnits = (RB ? RMAX / 2 : RMAX) * CW;                 // (row pair | row) x column block
    constexpr int kPasses = (kUnits + kTileWaves - 1) / kTileWaves;    // units per wave
    constexpr int kSlots = RB ? 2 * kPasses : kPasses;                 // rows a lane may update
    constexpr int kPlane = (kTileMaxRows + 2) * kTileCols;   // one pad row above and below: the neighbours of a stale edge cell stay inside
    __shared__ float u_lds[kBufs * kPlane];
    __shared__ float q_lds[TOL ? kBufs * kPlane : 1];
    __shared__ uint32_t n_lds[TOL ? kBufs * kPlane : 1];
    __shared__ unsigned wg_delta;
    __shared__ int step_done[kTileWaves];   // iterations each wave has finished (kTileWaveSync)
    __shared__ __attribute__((aligned(16))) char math_lds_bytes[TOL ? TolLn<4>::kLdsBytes : kMathLdsDoubles * (int)sizeof(double)];
    const TolLnEntry *const tl = reinterpret_cast<const TolLnEntry *>(math_lds_bytes);
    const MathTab tab = math_tables_at(reinterpret_cast<double *>(math_lds_bytes), EPIC_CONSTS_TILE(RB));
    MathTabRegs tab_regs = {};
    if (MATH == kMathPrecise) tab_regs = math_tables_fetch();

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tr = blockIdx.x / a.tiles_c, tc = blockIdx.x % a.tiles_c;
    const int H = a.halo, Sr = a.tile_rows + 2 * H, Tc = kTileCols - 2 * H;
    const int R0 = tr * a.tile_rows - H, C0 = tc * Tc - H;   // global position of local cell (0, 0)
    auto at = [](int lr, int lc) { return (lr + 1) * kTileCols + lc; };
    // unit of pass i of this wave: wave + W i = (row pair | row) x CW + column block
    auto unit_row = [&](int i) { return (wave + kTileWaves * i) / CW; };   // row pair (red-black) or row (Jacobi)
    auto unit_blk = [&](int i) { return (wave + kTileWaves * i) % CW; };   // column block: local columns 64 blk .. 64 blk + 63
    // local row of slot k of this wave -- red-black: the two rows of pass i's row pair are slots 2 i, 2 i + 1; Jacobi: slot = pass
    auto slot_row = [&](int k) { return RB ? 2 * unit_row(k >> 1) + (k & 1) : unit_row(k); };
    auto slot_blk = [&](int k) { return unit_blk(RB ? k >> 1 : k); };

    // ---- load: the tile with its ghost ring; cells outside the grid never change and are never read by a cell that does
    // (the grid's border is locked).  The locks of the rows a wave will update stay with it as LANE MASKS in scalar registers.
    lmask lockm[kSlots];
#pragma unroll
    for (int k = 0; k < kSlots; ++k) {
        const int lr = slot_row(k), lc = 64 * slot_blk(k) + lane;
        lockm[k] = ~0ull;
        if (lr >= Sr) continue;   // wave-uniform
        const int gr = R0 + lr, gc = C0 + lc;
        const bool inside = gc >= 0 && gc < a.pitch && gr >= 0 && gr < a.rows;
        float v = -1e6f;
        uint32_t lk = 1;
        if (inside) {
            v = a.in[(size_t)gr * a.pitch + gc];
            lk = (a.maskw[mask_word_2d((unsigned)gr, (unsigned)gc, (unsigned)a.pitch)] >> mask_bit_2d((unsigned)gc)) & 1u;
        }
        lockm[k] = __builtin_amdgcn_ballot_w64(lk != 0);
        u_lds[at(lr, lc)] = v;
        if (TOL) {
            const Split1 sp = tol_split1(v);
            q_lds[at(lr, lc)] = sp.q;
            n_lds[at(lr, lc)] = f2u(sp.zm);
        }
    }
    if (threadIdx.x < 2 * kTileCols) {   // the pad rows: read by the (discarded) updates of the outermost ring only
        const int lr = threadIdx.x < kTileCols ? -1 : Sr, lc = threadIdx.x % kTileCols;
#pragma unroll
        for (int b = 0; b < kBufs; ++b) {
            u_lds[b * kPlane + at(lr, lc)] = -1e6f;
            if (TOL) { q_lds[b * kPlane + at(lr, lc)] = 1.0f; n_lds[b * kPlane + at(lr, lc)] = kTolMagicBits; }
        }
    }
    if (threadIdx.x == 0) wg_delta = 0;
    if (threadIdx.x < kTileWaves) step_done[threadIdx.x] = 0;
    if (MATH == kMathPrecise) math_tables_commit(tab_regs, reinterpret_cast<double *>(math_lds_bytes));
    if (TOL) TolLn<4>::stage(reinterpret_cast<TolLnEntry *>(math_lds_bytes));   // ends with a workgroup barrier
    else __syncthreads();

    // ---- K iterations on the tile
    const int own_r0 = H, own_r1 = H + a.tile_rows;   // owned local rows [own_r0, own_r1), columns [H, width - H)
    // lanes of column block blk whose local column lies in [lo, width - lo): the first block loses its first lo lanes, the last its last lo
    auto cols_from = [&](int blk, int lo) -> lmask { return (blk == 0 ? ~0ull << lo : ~0ull) & (blk == CW - 1 ? ~0ull >> lo : ~0ull); };
    const int odd_lane = lane & 1;
    const bool want_delta = a.delta_bits != nullptr || a.tile_delta != nullptr;
    float dmax = 0.0f;
    const int wave_up = (wave + kTileWaves - 1) % kTileWaves, wave_dn = (wave + 1) % kTileWaves;
    for (int j = 0; j < a.steps; ++j) {
        if (kTileWaveSync && j > 0) {   // the two waves next to this one have finished iteration j - 1
            while (__hip_atomic_load(&step_done[wave_up], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < j ||
                   __hip_atomic_load(&step_done[wave_dn], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < j)
                __builtin_amdgcn_s_sleep(1);
        }
        // cells that still have four valid neighbours: local rows [lo, hi_r], columns [lo, 63 - lo]
        const int lo = j + 1, hi_r = Sr - 2 - j;
        const bool check = want_delta && j == a.steps - 1;
        const int src = RB ? 0 : (j & 1) * kPlane, dst = RB ? 0 : ((j & 1) ^ 1) * kPlane;
        // red-black: the active cell of column gc in global row gr has (gr + gc + iteration) odd (harmonic_cpu.cpp:46-51); in a row
        // pair that starts at an even local row the lanes with (lane + b) odd take the first row, the others the second (a column
        // block starts at an even column)
        const int b = (R0 + C0 + a.parity + j) & 1;
        const lmask first_row = b ? ~kOddLanes : kOddLanes;
#pragma unroll
        for (int i = 0; i < kPasses; ++i) {
            lmask upd, own;
            int p;
            const int blk = unit_blk(i);
            const lmask cols_ok = cols_from(blk, lo), own_cols = cols_from(blk, H);
            if (RB) {
                const int r2 = 2 * unit_row(i);
                const bool v0 = r2 >= lo && r2 <= hi_r, v1 = r2 + 1 >= lo && r2 + 1 <= hi_r;
                if (!(v0 || v1)) continue;   // wave-uniform: the pair has gone stale (or lies beyond the tile)
                const lmask lock = (lockm[2 * i] & first_row) | (lockm[2 * i + 1] & ~first_row);
                upd = ((v0 ? first_row : 0ull) | (v1 ? ~first_row : 0ull)) & cols_ok & ~lock;
                own = ((r2 >= own_r0 && r2 < own_r1 ? first_row : 0ull) | (r2 + 1 >= own_r0 && r2 + 1 < own_r1 ? ~first_row : 0ull)) & own_cols;
                p = at(r2, 64 * blk + lane) + ((odd_lane ^ b) ? 0 : kTileCols);
            } else {
                const int lr = unit_row(i);
                if (lr < lo || lr > hi_r) continue;   // wave-uniform
                upd = cols_ok & ~lockm[i];
                own = lr >= own_r0 && lr < own_r1 ? own_cols : 0ull;
                p = at(lr, 64 * blk + lane);
            }
            const float c = u_lds[src + p];
            const float uu = u_lds[src + p - kTileCols], ud = u_lds[src + p + kTileCols], ul = u_lds[src + p - 1], ur = u_lds[src + p + 1];
            float nv;
            if (TOL) {
                nv = tol_update_2d(uu, ud, ul, ur, q_lds[src + p - kTileCols], n_lds[src + p - kTileCols], q_lds[src + p + kTileCols],
                                   n_lds[src + p + kTileCols], q_lds[src + p - 1], n_lds[src + p - 1], q_lds[src + p + 1],
                                   n_lds[src + p + 1], tl);
            } else {
                nv = cell_update_2d<MATH>(uu, ud, ul, ur, tab);
            }
            const float o = sel(upd, nv, c);
            u_lds[dst + p] = o;
            if (TOL) {
                const Split1 sp = tol_split1(o);
                q_lds[dst + p] = sp.q;
                n_lds[dst + p] = f2u(sp.zm);
            }
            if (check) dmax = max2(dmax, sel(own, fabsf(c - o), 0.0f));   // (wave-uniform branch)
        }
        if (kTileWaveSync) {
            if (lane == 0) __hip_atomic_store(&step_done[wave], j + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else {
            __syncthreads();
        }
    }
    if (kTileWaveSync) __syncthreads();   // every wave has finished: the owned cells may leave

    // ---- store the owned cells (locked ones included: `out` is the other buffer)
    const int fin = RB ? 0 : (a.steps & 1) * kPlane;
    for (int unit = own_r0 * CW + wave; unit < own_r1 * CW; unit += kTileWaves) {   // (row, column block) units of the owned rows
        const int lr = unit / CW, lc = 64 * (unit % CW) + lane;
        const int gr = R0 + lr, gc = C0 + lc;
        if (gr >= a.rows) break;   // wave-uniform
        if (lc >= H && lc < kTileCols - H && gc < a.pitch) a.out[(size_t)gr * a.pitch + gc] = u_lds[fin + at(lr, lc)];
    }
    if (want_delta) {
        dmax = wave_max(dmax);
        if (a.delta_bits != nullptr && lane == 0 && dmax > 0.0f) atomicMax(a.delta_bits, __float_as_uint(dmax));
        if (a.tile_delta != nullptr) {
            if (lane == 0 && dmax > 0.0f) atomicMax(&wg_delta, __float_as_uint(dmax));
            __syncthreads();
            if (threadIdx.x == 0) a.tile_delta[blockIdx.x] = __uint_as_float(wg_delta);
        }
    }
}

}  // namespace

hipError_t launch_tile_2d(const float *in, float *out, const uint32_t *maskw, int rows, int pitch, const TilePlan &plan, int steps,
                          int math, int parity, unsigned *delta_bits, hipStream_t stream, float *tile_delta)
{
    if (steps <= 0) return hipSuccess;
    if (!in || !out || in == out || !maskw || pitch <= 0 || (pitch % 256) != 0 || rows <= 0) return hipErrorInvalidValue;
    if (math != kMathPrecise && math != kMathFast && math != kMathTol) return hipErrorInvalidValue;
    const bool rb = parity >= 0;
    const int width = plan.tile_cols + 2 * plan.halo, max_rows = tile_2d_max_rows(math, rb, width);
    if (plan.halo < 1 || steps > plan.halo || plan.tile_rows < 2 || max_rows == 0 || plan.tile_rows + 2 * plan.halo > max_rows ||
        plan.tile_cols < 8 || plan.tiles_r < 1 || plan.tiles_c < 1 || (long long)plan.tiles_r * plan.tile_rows < rows)
        return hipErrorInvalidValue;
    Tile2dArgs a;
    a.in = in;
    a.out = out;
    a.maskw = maskw;
    a.delta_bits = delta_bits;
    a.tile_delta = tile_delta;
    a.rows = rows;
    a.pitch = pitch;
    a.halo = plan.halo;
    a.tile_rows = plan.tile_rows;
    a.tiles_c = plan.tiles_c;
    a.steps = steps;
    a.parity = parity < 0 ? 0 : (parity & 1);
    void (*kernel)(Tile2dArgs);
    if (width == kTile2dCols)
        kernel = math == kMathTol    ? (rb ? tile2d_kernel<kMathTol, true, 1, 64> : tile2d_kernel<kMathTol, false, 1, 64>)
                 : math == kMathFast ? (rb ? tile2d_kernel<kMathFast, true, 1, 64> : tile2d_kernel<kMathFast, false, 1, 64>)
                                     : (rb ? tile2d_kernel<kMathPrecise, true, 1, 64> : tile2d_kernel<kMathPrecise, false, 1, 64>);
    else
        kernel = math == kMathTol    ? tile2d_kernel<kMathTol, true, 2, 64>   // (tol Jacobi has no wide tile: tile_2d_max_rows)
                 : math == kMathFast ? (rb ? tile2d_kernel<kMathFast, true, 2, 128> : tile2d_kernel<kMathFast, false, 2, 128>)
                                     : (rb ? tile2d_kernel<kMathPrecise, true, 2, 128> : tile2d_kernel<kMathPrecise, false, 2, 128>);
    hipLaunchKernelGGL(kernel, dim3((unsigned)(plan.tiles_r * plan.tiles_c)), dim3(kTileThreads), 0, stream, a);
    return hipGetLastError();
}

}  // namespace epic_hip
