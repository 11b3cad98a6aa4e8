// kernels.h -- internal launchers (host side) for the gfx950 kernels.  Not part of the public ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>

#include "driver_config.h"   // LaunchKnobs: what the launchers take from the environment (read in driver_config.cpp only)

namespace epic_hip {

// ---- 2-D (kernels_2d.hip) -------------------------------------------------------------------
// One Jacobi sweep of rows [row_begin, row_end) of a pitched rows x pitch grid.  delta_bits == nullptr
// selects the plain kernel; otherwise max |du| is atomicMax'ed into *delta_bits (float bits, zero it first).
// math: 0 = precise (libm-equivalent exp/log in f64, the default), 1 = fast (v_exp_f32 / v_log_f32).
// Activity tracking of full-grid launches (kernels_2d.hip, Sweep2dArgs): tiles = sweep_2d_tiles(rows, pitch,
// rows_per_task).  Per direction kWakeListCount lists of sweep_2d_list_cap(tiles) uint32 each and one uint32 of queued
// mark per tile; counter sets of kWakeListCount uint32.
constexpr int kWakeListCount = 256;
constexpr int kWakeCounterStride = 32;  // uint32 words between two list counters (a 128-byte line each)
struct Activity {
    const uint32_t *list_in;   // nullptr: every tile runs (forced iteration)
    const uint32_t *count_in;
    uint32_t *list_out;
    uint32_t *count_out;
    uint32_t *count_zero;
    uint32_t *queued_in;
    uint32_t *queued_out;
    unsigned long long *total;   // running sum of the tiles the list-driven launches were handed (epic_hip_work_done)
};
inline size_t sweep_2d_tiles(int rows, int pitch, int rows_per_task)
{
    return (size_t)((pitch + 255) / 256) * (size_t)((rows + rows_per_task - 1) / rows_per_task);
}
inline size_t sweep_2d_list_cap(size_t tiles) { return (tiles + kWakeListCount - 1) / kWakeListCount; }
// Blocks of a list-driven launch: as many persistent waves as the chip holds of THIS kernel at a time (resident_blocks:
// the occupancy of the instantiation x the CUs -- 4096 waves for the tracked tol sweeps, 5120 / 7168 for the precise ones),
// or one per tile on small grids; a multiple of the 8 XCDs.  More than fit only start when the first ones have left
// (8192^2 tol relaxation: 2.74 s with 8192 waves, 2.66 with 4096, 2.97 with 2048).  EPIC_HIP_LIST_WAVES overrides.
inline int sweep_2d_list_blocks(size_t tiles, int resident_blocks)
{
    const size_t forced = process_launch_knobs().list_waves;
    const size_t waves = forced ? forced : (size_t)(resident_blocks > 0 ? resident_blocks : 2048) * 4;
    return (int)(((tiles < waves ? tiles : waves) + 31) / 32) * 8;
}
// blocks of 256 threads of `kernel` the current device holds at a time (cached per kernel)
int resident_blocks_of(const void *kernel);
// parity < 0: Jacobi (in != out); parity 0 / 1: red-black half-sweep of that colour, in place (in == out).
// check_begin / check_end: with delta_bits, only these rows count for max |du| (default: the rows swept) -- a slab sweeps its
// ghost rows in the same launch but must not let them into the convergence test.
hipError_t launch_sweep_2d(const float *in, float *out, const uint32_t *maskw, int rows, int pitch, int row_begin,
                           int row_end, int rows_per_task, int math, int parity, unsigned *delta_bits,
                           hipStream_t stream, const Activity *act = nullptr, int check_begin = -1, int check_end = -1);
// Wake tiles [t_lo, t_hi) for the launch that will consume the lists given as next_as_out->list_out / count_out / queued_out.
hipError_t launch_wake_tile_range(const Activity *next_as_out, size_t tiles, int t_lo, int t_hi, hipStream_t stream);
// two red-black iterations fused into one in -> out pass (first colour = parity); see kernels_2d.hip
// act (may be null): work lists of THIS pass's tiling, rb_fused_2d_tiles() tiles; delta_bits (may be null): max |du| of the second
// of the two iterations (zero it first).
hipError_t launch_rb_fused_2d(const float *in, float *out, const uint32_t *maskw, int rows, int pitch, int rows_per_task,
                              int math, int parity, hipStream_t stream, const uint32_t *maskf = nullptr, const Activity *act = nullptr,
                              unsigned *delta_bits = nullptr, int check_begin = -1, int check_end = -1);   // check_*: as launch_sweep_2d (slabs)
inline size_t rb_fused_2d_tiles(int rows, int pitch, int rows_per_task)
{
    return (size_t)((pitch + 247) / 248) * (size_t)((rows + rows_per_task - 1) / rows_per_task);
}
// Two iterations in one pass (tol math only): in = u_k, out = u_{k+2}; in != out.  parity < 0: Jacobi; 0 / 1: the reference's
// red-black scheme, parity = the first iteration's number & 1 (both colours are swept, the first one first).
// maskf (may be null): the masks in the fused layout below -- saves the pass a funnel shift of two mask words per row.
// act (may be null): work lists of THIS pass's tiling (rb_fused_2d_tiles() tiles; needs maskf); delta_bits (may be null): max |du| of
// the second of the two iterations (zero it first; needs maskf).
hipError_t launch_jacobi_fused_2d(const float *in, float *out, const uint32_t *maskw, int rows, int pitch, int rows_per_task,
                                  int math, hipStream_t stream, int parity = -1, const uint32_t *maskf = nullptr,
                                  const Activity *act = nullptr, unsigned *delta_bits = nullptr, int check_begin = -1, int check_end = -1);
// Fused layout: a fused pass cuts a row into strips of 248 columns, lane L of strip S holding columns 248 S - 4 + 4 L .. + 3
// (lanes 0 and 63 are halo lanes); per row and such strip four 64-bit words as in the standard layout (bit L of word j =
// cell 4 L + j of that mapping).  Derived from the standard masks after every upload and every edit.
inline size_t mask_words_fused_2d(int rows, int pitch) { return (size_t)rows * (size_t)((pitch + 247) / 248) * 8u; }
hipError_t launch_fuse_masks_2d(const uint32_t *maskw, int rows, int pitch, uint32_t *maskf, hipStream_t stream);
hipError_t launch_eval_math(const float *in, float *out, size_t n, int which, hipStream_t stream);
hipError_t launch_pack_mask_2d(const uint32_t *locked, int rows, int cols, int pitch, int ghost_top,
                               int ghost_bottom, uint32_t *maskw, hipStream_t stream);
hipError_t launch_fill(float *p, size_t n, float v, hipStream_t stream);
// row0 / grid_rows / pin_*: slab form -- the buffer holds global rows [row0, row0 + rows) of a grid of grid_rows rows
// (0 = the buffer is the grid) and its outermost rows are pinned ghost rows.
hipError_t launch_set_cells_2d(float *u, uint32_t *maskw, int rows, int cols, int pitch, unsigned k,
                               const unsigned *v, const unsigned *types, hipStream_t stream, int row0 = 0, int grid_rows = 0,
                               int pin_top = 0, int pin_bottom = 0);

// ---- small grids: several iterations per launch on LDS tiles (kernels_tile2d.hip) ----------------
// A tile owns tile_rows x tile_cols cells and carries `halo` ghost rings; a launch performs up to `halo` iterations.
struct TilePlan { int halo, tile_rows, tile_cols, tiles_r, tiles_c; };   // halo == 0: no plan (grid or halo out of range)
constexpr int kTile2dCols = 64, kTile2dMaxRows = 64;   // the narrow LDS tile: one lane per column, at most 64 rows with the ghost rings
constexpr int kTile2dWideCols = 128;                   // the wide one: two column blocks of 64 lanes (width of a plan: tile_cols + 2 halo)
// Tallest LDS tile (rows, ghost rings included) the kernels have for this arithmetic, scheme and width; 0: no such kernel.
// (What fits 160 KB of LDS: u, and for the tol math q and n beside it, once for red-black and twice for Jacobi.)
inline int tile_2d_max_rows(int math, bool redblack, int width)
{
    if (width == kTile2dCols) return kTile2dMaxRows;
    if (width != kTile2dWideCols) return 0;
    if (math == 4) return redblack ? 64 : 0;   // tol
    return 128;
}
// How a grid is cut into tiles for `halo` ghost rings: owned columns width - 2 halo; owned rows as tall as the LDS tile allows
// (max_rows, ghost rings included), but not taller than what gives every CU of the chip a tile (the tiles of a launch run side
// by side: its time is the time of ONE tile, so smaller tiles are faster until the chip is full).  tile_rows > 0: that height
// (clamped).
inline TilePlan tile_2d_plan(int rows, int cols, int halo, int tile_rows = 0, int width = kTile2dCols, int max_rows = kTile2dMaxRows)
{
    TilePlan p = {0, 0, 0, 0, 0};
    if (halo < 1 || 2 * halo >= width - 8 || 2 * halo >= max_rows - 1 || rows < 3 || cols < 3) return p;
    const int tc = width - 2 * halo;
    const int tiles_c = (cols + tc - 1) / tc;
    const int max_tr = max_rows - 2 * halo;
    int tr = tile_rows;
    if (tr <= 0) {
        static const int cus = [] {   // (asked once per process: the devices of a node are alike)
            int n = 256, dev = 0;
            if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1) {
                (void)hipGetLastError();
                n = 256;
            }
            return n;
        }();
        const int tiles_r_max = cus / tiles_c > 1 ? cus / tiles_c : 1;   // rows of tiles the chip takes in one round
        tr = (rows + tiles_r_max - 1) / tiles_r_max;
        tr = (tr + 1) / 2 * 2;                                           // (even: the LDS tile is walked in row pairs)
        if (tr < 8) tr = 8;
    }
    tr = (tr < max_tr ? tr : max_tr) / 2 * 2;
    if (tr < 2) return p;
    p.halo = halo;
    p.tile_rows = tr;
    p.tile_cols = tc;
    p.tiles_r = (rows + tr - 1) / tr;
    p.tiles_c = tiles_c;
    return p;
}
// `steps` (<= plan.halo) iterations in -> out (in != out, also for red-black: a neighbouring tile must find the old ghost
// values).  parity < 0: Jacobi; else red-black, parity = number of the first iteration & 1.  delta_bits (may be null): max |du|
// of the LAST of the iterations (zero it first); tile_delta (may be null): the same maximum per tile, tiles_r x tiles_c floats
// written with plain stores -- nothing to zero, and the array may be pinned host memory.  math: precise, fast or tol.
hipError_t launch_tile_2d(const float *in, float *out, const uint32_t *maskw, int rows, int pitch, const TilePlan &plan, int steps,
                          int math, int parity, unsigned *delta_bits, hipStream_t stream, float *tile_delta = nullptr);

// ---- streamlines on the resident field (path_2d.hip): one lane per start point ----------------
// d_pts: n_paths x 2 * max_points floats; d_k: points per path (0 on failure); d_rc: EPIC_* code per path.
hipError_t launch_follow_paths_2d(const float *u, const uint32_t *maskw, int rows, int cols, int pitch, unsigned n_paths,
                                  const float *d_starts, float step, float cd, unsigned max_points, float *d_pts,
                                  unsigned *d_k, int *d_rc, hipStream_t stream);

// tol math, 2-D: rows are loaded kTolRowsAhead steps ahead of their use through a ring of kTolRowsAhead + 3 register sets; the
// pipelined loop runs in trips of kTolTripRows rows (the rings close after lcm(ring, 2) steps), and the host picks rows per
// task in multiples of it (driver_plan.hip: auto_rows_per_task).
#ifndef EPIC_TOL_AHEAD  // build knob (A/B)
#define EPIC_TOL_AHEAD 2
#endif
constexpr int kTolRowsAhead = EPIC_TOL_AHEAD;
constexpr int kTolTripRows = (kTolRowsAhead + 3) % 2 == 0 ? kTolRowsAhead + 3 : 2 * (kTolRowsAhead + 3);
// Rows per task of the fused double sweep (jacobi_fused2d_kernel) for a grid -- or a slab -- of `rows` rows, before anything is
// measured (driver_plan.hip: tune_fused_rows measures on grids of 4 Mcell and more).  A task recomputes the first iteration of one row
// above and one below its chunk and fills its pipeline once, so taller is cheaper -- up to about 40 rows, where the gain levels off --,
// but the chip wants a round of waves: 4096 of them (256 CUs x 16), i.e. rows x strips / 4096 rows per task at most.  Since round 5 the
// launcher cuts the rows into as many chunks as fit the last round of blocks (kernels_2d.hip: tighten_chunks), so the height no longer
// has to hit a whole number of rounds itself (rounds 3-4 searched for that: 23 rows at 8192^2, 180 us per launch against 163 at 40).
// Measured with the tightening on, us per iteration of the pass at 8 / 12 / 16 / 20 / 24 / 32 / 40 / 48 rows (tools/exp_slab_heights.py):
// 1024 x 8192: 13.7 / 12.4 / 12.4 / 12.5 / 12.5 / 12.4 / 15.4 / 15.6;  2048 x 8192: 23.4 / 21.3 / 21.1 / 20.8 / 21.0 / 21.1 / 22.0 / 22.0;
// 4096 x 8192: 42.8 / 42.3 / 40.0 / 38.3 / 37.9 / 37.8 / 38.0 / 39.0;  8192 x 8192 (per launch of two): 35: 172, 40: 163, 44: 164, 48: 166, 64: 167.
inline int jacobi_fused_auto_rows(int rows, int pitch)
{
    const long long nstrips = (pitch + 247) / 248, slots = 4096;
    const long long r = (long long)rows * nstrips / slots;
    return (int)(r < 12 ? 12 : r > 40 ? 40 : r);
}
// rows are padded to whole wave-strips (256 floats = 1 KiB): every lane of every wave is in bounds, always
inline int pitch_for_cols(int cols) { return (cols + 255) / 256 * 256; }
// 2-D mask layout (device-private): LANE MASKS.  For every row and every 256-column strip four 64-bit words, word j
// holding in bit L the lock of cell (row, 256 strip + 4 L + j) -- i.e. exactly the SGPR-pair operand v_cndmask wants
// for the j-th of the four cells lane L of the sweep owns.  A wave fetches the 32 bytes of its row with ONE scalar
// load; no VALU instruction is spent on the mask except the select itself.  1 bit per cell, 32 B per (row, strip).
inline size_t mask_words_2d(int rows, int pitch) { return (size_t)rows * (size_t)(pitch / 256) * 8u; }
// 32-bit word and bit of cell (r, c) in that layout (little-endian halves of the 64-bit lane masks)
__host__ __device__ inline size_t mask_word_2d(unsigned r, unsigned c, unsigned pitch)
{
    return (((size_t)r * (pitch >> 8) + (c >> 8)) * 4u + (c & 3u)) * 2u + ((c >> 7) & 1u);
}
__host__ __device__ inline unsigned mask_bit_2d(unsigned c) { return (c >> 2) & 31u; }

// ---- 3-D (kernels_3d.hip) -------------------------------------------------------------------
hipError_t launch_sweep_3d(const float *in, float *out, const uint32_t *maskw, int m0, int m1, int pitch,
                           int plane_begin, int plane_end, int math, int parity, unsigned *delta_bits,
                           hipStream_t stream, const Activity *act = nullptr, int check_begin = -1, int check_end = -1,
                           const LaunchKnobs *knobs = nullptr);   // knobs: the caller's context's (null: the process-wide ones)
// tiles of the 3-D sweep: one per (x0-plane, 32-row x1-chunk, 256-column x2-strip)
inline size_t sweep_3d_tiles(int m0, int m1, int pitch)
{
    return (size_t)m0 * (size_t)((m1 + 31) / 32) * (size_t)((pitch + 255) / 256);
}
hipError_t launch_pack_mask_3d(const uint32_t *locked, int m0, int m1, int m2, int pitch, uint32_t *maskw,
                               hipStream_t stream);
inline size_t mask_words_3d(int m0, int m1, int pitch) { return (size_t)m0 * (size_t)m1 * (size_t)(pitch / 32); }

}  // namespace epic_hip
