// kernels_2d.hip -- 2-D log-space Jacobi sweep for gfx950 (MI355X), hand-written HIP.
//
// Replaces the reference CUDA kernels harmonic_update_2d_gpu / harmonic_update_and_check_2d_gpu /
// harmonic_compute_max_delta_gpu (libepic/src/harmonic/harmonic_gpu.cu:39-153).  Not a translation:
// the reference does an in-place red-black half-sweep with one block per row, stride-2 column access,
// a uint32 mask word per cell and a second kernel + host sync for the max-delta.  Here:
//
//  * Jacobi ping-pong (u_in -> u_out), every cell read once and written once per sweep: 8 B / cell-update.
//  * One wave (64 lanes) owns a 256-column strip and marches down `rows_per_task` rows.  Each lane holds
//    4 consecutive columns as one dwordx4, so a wave-row is one fully coalesced 1 KiB load and 1 KiB store.
//    The three live rows (up / centre / down) stay in registers while marching, so vertical neighbours cost
//    no extra traffic; horizontal neighbours inside the strip move with one full-wave DPP shift each
//    (v_mov_b32_dpp wave_shr:1 / wave_shl:1); the two strip-edge columns are wave-uniform: two scalar loads per row.
//  * The obstacle/goal mask is bit-packed as LANE MASKS (kernels.h): per row and strip four 64-bit words that are
//    the SGPR-pair operands of the four v_cndmask of a row, fetched with one s_load_dwordx8: +0.125 B/cell instead
//    of the reference's +4 B/cell, and no VALU instruction besides the select itself.
//  * Rows are addressed through buffer descriptors with the row / strip part of the address in an SGPR: loads and
//    stores cost no VALU instruction either.  The kernel is bound by VALU issue, so every instruction that is not
//    arithmetic of the update was moved to the scalar unit (precise math: 70 VALU instructions per cell, 41 of them f64 /
//    conversions; tol math: 41 -- profiles/r02_sq_counters_*.txt).
//  * max |u_new - u_old| is reduced in registers, across the wave with shuffles, and leaves the wave as a
//    single atomicMax on the float's bit pattern (valid order for non-negative floats).  No second kernel.
//  * blockIdx is remapped so that each XCD sweeps a contiguous band of rows: vertically adjacent tasks share
//    their two halo rows through that XCD's L2.
//
//  * Optional activity tracking: a sweep lists the tiles its successor has to recompute and the successor runs as
//    persistent waves over those lists (Sweep2dArgs); results do not depend on it.
//
// Roofline: the memory side is HBM-bound (8 B per cell, 94-98 us per 8192^2 sweep with trivial arithmetic).  With the
// fast math (v_exp_f32 / v_log_f32) the kernel stays there; with the default precise math (expf / logf bit-identical to
// glibc, evaluated in f64) it is bound by VALU issue: 41 f64 / conversion instructions per cell plus 29 others at ~4
// cycles per wave each, 138-146 us; with the tol math (one split per cell shared by its neighbours, cell_update.h) both
// pipes are nearly full: 107-109 us = 0.62 of 8 TB/s -- which is why pairs of plain tol
// iterations run through jacobi_fused2d_kernel below instead: two iterations per pass over the field, 4 B per
// cell-update, VALU-bound at 0.67-0.70 (section 4.2b).  rb_fused2d_kernel is its red-black counterpart (precise / fast math).
#include <type_traits>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <algorithm>
#include <map>
#include <mutex>

#include "cell_update.h"
#include "kernels.h"
#include "wake.h"

// where the precise routines' f64 constants live, per kernel (cell_update.h: MathTab::consts); build knobs for the A/B
#ifndef EPIC_CONSTS_SWEEP2D
#define EPIC_CONSTS_SWEEP2D(RB, TRACK) ((RB) && !(TRACK) ? kConstsKeep : kConstsPlain)
#endif
#ifndef EPIC_CONSTS_RBPAIR
#define EPIC_CONSTS_RBPAIR(TRACK, CHECK) kConstsKeep
#endif
namespace epic_hip {

namespace {

constexpr int kWave = 64;
constexpr int kColsPerLane = 4;
constexpr int kStripCols = kWave * kColsPerLane;  // 256
constexpr int kWavesPerBlock = 4;
constexpr int kNumXcd = 8;
static_assert(kWakeLists == kWave * kWavesPerBlock, "one thread per work list resets the counters");

struct Sweep2dArgs {
    const float *in;
    float *out;
    const uint32_t *maskw;  // lane masks (kernels.h): per row and strip four 64-bit words, bit L of word j = cell 4 L + j; 1 = locked
    const uint32_t *maskf;  // fused passes only, may be null: the same masks cut for THEIR lane -> column mapping (kernels.h: fused layout)
    unsigned *delta_bits;   // max |du| as float bits (atomicMax), used when CHECK
    int rows;               // rows of the (local) grid, including ghost rows in slab mode
    int pitch;              // floats per row, multiple of 256
    int row_begin, row_end; // rows swept by this launch
    int rows_per_task;      // rows of a task's chunk ...
    int chunk_rem;          // ... the first chunk_rem chunks have one row more (fused passes, tighten_chunks below; 0: all alike)
    int nstrips;            // ceil(pitch / 256)
    int ntasks;             // nstrips * nchunks
    int parity;             // red-black scheme only: currentIteration & 1 (which colour this half-sweep updates)
    int flags;              // tuning, never results: bit 0 = odd row-chunks march upwards
    // Activity tracking (full-grid launches only; TRACK kernels): wake.h.  A tile is one task (rows_per_task x 256
    // cells); it reads its own cells, the last column of its left neighbour, the first column of its right neighbour,
    // the last row of the tile above and the first row of the tile below (5-point stencil).
    WakeArgs wake;
    int nchunks;
    int nblocks;            // logical blocks (ceil(ntasks / 4)): a launch may hold fewer workgroups, which then walk them (tol math)
    int check_lo, check_hi; // CHECK: only rows [check_lo, check_hi) count for max |du| (a slab's ghost rows are swept but do not)
};

// Blocks are dealt round-robin over the 8 XCDs (b % 8 labels the XCD group).  Give each group a
// contiguous range of logical block ids; bijective for any grid size.  Speed only, never correctness.
__device__ __forceinline__ int xcd_contiguous_block(int b, int nblk)
{
    int x = b % kNumXcd, i = b / kNumXcd;
    int q = nblk / kNumXcd, rem = nblk % kNumXcd;
    return x * q + (x < rem ? x : rem) + i;
}

// RB = false: Jacobi, in -> out.  RB = true: the reference's red-black half-sweep, IN PLACE (in == out): only the
// cells with (row + col + currentIteration) odd are recomputed (harmonic_cpu.cpp:46-51), from neighbours that all have
// the other colour and therefore do not change during this launch -- no ordering between waves is needed, and with
// the precise math the result is the reference CPU solver's, bit for bit, half-sweep for half-sweep.
// (68 VGPRs = 7 waves per SIMD with the precise math, untracked; 6, 7 or 8 waves per SIMD time the same,
// profiles/r01_experiments.txt.)
#ifdef EPIC_SWEEP_WAVES  // experiment knob (make EXTRA=-DEPIC_SWEEP_WAVES=8): pin the waves per SIMD of the sweep
#define EPIC_SWEEP_OCC __attribute__((amdgpu_waves_per_eu(EPIC_SWEEP_WAVES, EPIC_SWEEP_WAVES)))
#else
#define EPIC_SWEEP_OCC
#endif
// (the untracked tol kernels are asked for 6 waves per SIMD -- 80 VGPRs --, which they reach without spilling; left
// alone the five-row ring takes 83.  The tracked and the red-black check variants would spill: they keep what they get.
// No sweep kernel may use scratch: tools/isa_hazards.py checks it.)
template <bool CHECK, int MATH, bool RB, bool TRACK> struct SweepOcc {
    static constexpr int kMinWaves = (MATH == kMathTol && !TRACK && !(RB && CHECK)) ? (CHECK ? 5 : 6) : 1;  // the check sweep (1 in 100) needs 3 registers more
};
template <bool CHECK, int MATH, bool RB, bool TRACK>
__global__ __launch_bounds__(kWave * kWavesPerBlock, (SweepOcc<CHECK, MATH, RB, TRACK>::kMinWaves)) EPIC_SWEEP_OCC void sweep2d_kernel(Sweep2dArgs a)
{
    constexpr bool TOL = MATH == kMathTol;  // one split (exp-class evaluation) per cell, shared by its neighbours (cell_update.h)
    // precise math: glibc's expf / logf tables (3 KiB, every wave writes them itself); tol math: the table of its own
    // logarithm (16 KiB, copied by the whole workgroup before anything else happens -- one barrier, no wave has left yet)
    __shared__ __attribute__((aligned(16))) char math_lds_bytes[TOL ? TolLn<4>::kLdsBytes : kMathLdsDoubles * (int)sizeof(double)];
    double *const math_lds = reinterpret_cast<double *>(math_lds_bytes);
    const TolLnEntry *const tl = reinterpret_cast<const TolLnEntry *>(math_lds_bytes);
    if (TOL) TolLn<4>::stage(reinterpret_cast<TolLnEntry *>(math_lds_bytes));
    // libm tables in LDS (precise math only): fetched here, written to LDS only after the first task's row loads are
    // under way, so that the fetch from constant memory hides behind them (the small ROS maps run one row per wave:
    // 3.45 -> 2.95 us per sweep of the 482 x 482 map)
    const MathTab lds = math_tables_at(math_lds, EPIC_CONSTS_SWEEP2D(RB, TRACK));
    MathTabRegs tab_regs = {};
    if (MATH == kMathPrecise) tab_regs = math_tables_fetch();
    bool tables_pending = MATH == kMathPrecise;
    const int lane = threadIdx.x & (kWave - 1);
    // wave-uniform quantities are forced into SGPRs: the row loop, its addresses and branches are scalar
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (TRACK) wake_reset_next(a.wake);
    // list-driven launch (wake.h): persistent waves walking the tiles listed by the previous iteration
    const bool listed = TRACK && a.wake.list_in != nullptr;
    WakeCursor cursor = {};
    if (listed && !wake_begin(a.wake, lane, wave, kWavesPerBlock, cursor)) return;
    float dmax = 0.0f;
    const int rlast = a.rows - 1;
    const size_t pitch = (size_t)a.pitch;

    int vb = blockIdx.x;  // logical block of this pass (launches with fewer workgroups than logical blocks walk them)
    for (;;) {  // one pass per task: exactly one unless the launch is list-driven or holds fewer workgroups than blocks
    int task;
    if (listed) task = wake_tile(a.wake, cursor);
    else task = xcd_contiguous_block(vb, a.nblocks) * kWavesPerBlock + wave;
    if (task >= a.ntasks) {  // the spare waves of the last logical block
        if (listed) break;
        vb += gridDim.x;
        if (vb >= a.nblocks) break;
        continue;
    }
    if (TRACK && lane == 0) a.wake.queued_in[task] = 0;
    const int strip = task % a.nstrips;
    const int chunk = task / a.nstrips;
    const int r0 = a.row_begin + chunk * a.rows_per_task;
    const int r1 = min(r0 + a.rows_per_task, a.row_end);

    const int col0 = strip * kStripCols;  // pitch % 256 == 0: every lane of every strip is in bounds (DPP and the masked adds want all lanes live)

    // Addressing costs no VALU instruction: rows are reached through buffer descriptors (base = a few rows above the
    // task, so that byte offsets stay far below 4 GiB on any grid) with the lane part of the address in a VGPR that
    // never changes (lane * 16 bytes) and the row / strip part in an SGPR (buffer_load ... offen with soffset).
    const int rlo = max(r0 - 8, 0);  // the march touches rows r0 - 6 .. r1 + 5 at most (prefetch past either end)
    const __amdgpu_buffer_rsrc_t rin = raw_buffer(a.in + (size_t)rlo * pitch), rout = raw_buffer(a.out + (size_t)rlo * pitch);
    const unsigned lane16 = (unsigned)lane * 16u;
    typedef unsigned vu4 __attribute__((ext_vector_type(4)));
    auto row_off = [&](int r) -> unsigned { return (unsigned)((r - rlo) * a.pitch + col0) * 4u; };  // r already clamped

    auto ld = [&](int r) -> float4 {
        r = min(max(r, 0), rlast);
        const vu4 q = __builtin_amdgcn_raw_buffer_load_b128(rin, lane16, row_off(r), 0);
        return make_float4(u2f(q.x), u2f(q.y), u2f(q.z), u2f(q.w));
    };
    // What a row needs besides its three rows of u is wave-uniform and arrives through SCALAR loads: the two strip-edge
    // values u[r][col0 - 1] and u[r][col0 + 256], and the four lane masks of the row (32 bytes, one s_load_dwordx8).
    // (The sweep never writes what it reads through them -- Jacobi reads u_in only, red-black reads cells of the other
    // colour, nothing writes the mask -- and every kernel launch starts with an invalidated scalar cache.)
    typedef const __attribute__((address_space(4))) float cfloat;
    typedef const __attribute__((address_space(4))) uint64_t cu64;
    const int hcol_l = max(col0 - 1, 0), hcol_r = min(col0 + kStripCols, a.pitch - 1);
    struct RowSide { float l, r; lmask m0, m1, m2, m3; };
    auto side = [&](int r) -> RowSide {
        r = min(max(r, 0), rlast);
        cfloat *row = (cfloat *)(a.in + (size_t)r * pitch);
        cu64 *mk = (cu64 *)a.maskw + ((size_t)r * a.nstrips + strip) * 4;
        return RowSide{row[hcol_l], row[hcol_r], mk[0], mk[1], mk[2], mk[3]};
    };
    // tol math: the splits of the row's two strip-edge cells, needed in lane 0 (left) and lane 63 (right) only.  The plain
    // loops split the two wave-uniform values as one packed pair in every lane (edge_split); the trips of the pipelined
    // loop split the edge cells of all their rows at once, one cell per lane (below).
    struct EdgeSplit { float ql, qr, zl, zr; };
    auto edge_split = [&](const RowSide &h) -> EdgeSplit {
        const Split2 hs = tol_split2(v2f{h.l, h.r});
        return EdgeSplit{hs.q.x, hs.q.y, hs.zm.x, hs.zm.y};
    };
    auto masks = [&](int r) -> RowSide {  // a RowSide without the edge values (the caller fills them in)
        r = min(max(r, 0), rlast);
        cu64 *mk = (cu64 *)a.maskw + ((size_t)r * a.nstrips + strip) * 4;
        return RowSide{0.0f, 0.0f, mk[0], mk[1], mk[2], mk[3]};
    };

    // March direction.  Vertically adjacent tasks share two halo rows; if every task marched downwards, task k would
    // read them at its end and task k+1 at its start, a whole task apart in time, and the XCD's 4 MiB L2 would have
    // turned over in between.  With odd chunks marching upwards both neighbours touch their common rows at the same end
    // of their march.  (up + down is a commutative f32 add and max is symmetric, so swapping them changes no bit.)
    const int dir = ((a.flags & 1) && (chunk & 1)) ? -1 : 1;
    const int rfirst = dir > 0 ? r0 : r1 - 1;
    const int nrows = r1 - r0;

    // activity tracking: cells of this lane rewritten with different bits -- anywhere, in its first / last column
    // (meaningful in lane 0 / lane 63), in the task's first / last row
    // (lane masks in SGPRs: four compares per row, everything else on the scalar unit)
    lmask chg_any = 0, chg_x = 0, chg_w = 0, chg_top = 0, chg_bot = 0;
    // One row: up / c / dn are rows r-1, r, r+1 of u_in, h the two strip-edge values of row r.
    // (tol math: su / sc / sd are the splits of the three rows, computed once per row as it enters the window)
    // (tol math: he = the splits of the two strip-edge cells of the row, see EdgeSplit below)
    auto row_step = [&](int r, const float4 &up, const float4 &c, const float4 &dn, const RowSide &h, const Split4 &su,
                        const Split4 &sc, const Split4 &sd, const EdgeSplit &he) {
        const float lf = wave_from_left(c.w, h.l);   // u[r][col-1]
        const float rt = wave_from_right(c.x, h.r);  // u[r][col+4]
        float4 o;
        if (TOL) {
            const float ql = wave_from_left(sc.qw, he.ql), qr = wave_from_right(sc.qx, he.qr);
            const uint32_t nl = f2u(wave_from_left(u2f(sc.nw), he.zl)), nr = f2u(wave_from_right(u2f(sc.nx), he.zr));
            o = c;
            // the cells of the row this iteration updates, two at a time in the three phases of cell_update.h (tol_pre2_2d,
            // tol_ln_issue / tol_ln_wait, tol_post2): the table reads of one pair are in flight while the other pair is worked on
            const bool odd_cols = !RB || ((r + a.parity) & 1) == 0, even_cols = !RB || !odd_cols;  // scalar
            auto pre_xz = [&] {
                return tol_pre2_2d(up.x, dn.x, lf, c.y, su.qx, su.nx, sd.qx, sd.nx, ql, nl, sc.qy, sc.ny,
                                   up.z, dn.z, c.y, c.w, su.qz, su.nz, sd.qz, sd.nz, sc.qy, sc.ny, sc.qw, sc.nw);
            };
            auto pre_yw = [&] {
                return tol_pre2_2d(up.y, dn.y, c.x, c.z, su.qy, su.ny, sd.qy, sd.ny, sc.qx, sc.nx, sc.qz, sc.nz,
                                   up.w, dn.w, c.z, rt, su.qw, su.nw, sd.qw, sd.nw, sc.qz, sc.nz, qr, nr);
            };
            float nx, ny, nz, nw;
            TolLnRaw ea, eb;
            if (!RB) {
                const TolPre2 pxz = pre_xz();
                TolLnPair f0 = tol_ln_issue<4>(pxz, tl);
                const TolPre2 pyw = pre_yw();
                TolLnPair f1 = tol_ln_issue<4>(pyw, tl);
                tol_ln_wait<2>(f0, ea, eb);
                tol_post2(pxz, ea, eb, kLn4, nx, nz);
                o.x = sel(h.m0, c.x, nx);
                o.z = sel(h.m2, c.z, nz);
                tol_ln_wait<0>(f1, ea, eb);
                tol_post2(pyw, ea, eb, kLn4, ny, nw);
                o.y = sel(h.m1, c.y, ny);
                o.w = sel(h.m3, c.w, nw);
            } else if (even_cols) {
                const TolPre2 pxz = pre_xz();
                TolLnPair f0 = tol_ln_issue<4>(pxz, tl);
                tol_ln_wait<0>(f0, ea, eb);
                tol_post2(pxz, ea, eb, kLn4, nx, nz);
                o.x = sel(h.m0, c.x, nx);
                o.z = sel(h.m2, c.z, nz);
            } else {
                const TolPre2 pyw = pre_yw();
                TolLnPair f1 = tol_ln_issue<4>(pyw, tl);
                tol_ln_wait<0>(f1, ea, eb);
                tol_post2(pyw, ea, eb, kLn4, ny, nw);
                o.y = sel(h.m1, c.y, ny);
                o.w = sel(h.m3, c.w, nw);
            }
        } else if (RB) {
            o = c;
            if (((r + a.parity) & 1) == 0) {  // scalar: this row's active cells sit in the odd columns (.y, .w)
                const float ny = cell_update_2d<MATH>(up.y, dn.y, c.x, c.z, lds);
                const float nw = cell_update_2d<MATH>(up.w, dn.w, c.z, rt, lds);
                o.y = sel(h.m1, c.y, ny);
                o.w = sel(h.m3, c.w, nw);
            } else {                           // even columns (.x, .z)
                const float nx = cell_update_2d<MATH>(up.x, dn.x, lf, c.y, lds);
                const float nz = cell_update_2d<MATH>(up.z, dn.z, c.y, c.w, lds);
                o.x = sel(h.m0, c.x, nx);
                o.z = sel(h.m2, c.z, nz);
            }
        } else {
            o.x = cell_update_2d<MATH>(up.x, dn.x, lf, c.y, lds);
            o.y = cell_update_2d<MATH>(up.y, dn.y, c.x, c.z, lds);
            o.z = cell_update_2d<MATH>(up.z, dn.z, c.y, c.w, lds);
            o.w = cell_update_2d<MATH>(up.w, dn.w, c.z, rt, lds);
        }
        if (!RB && !TOL) {
            o.x = sel(h.m0, c.x, o.x);
            o.y = sel(h.m1, c.y, o.y);
            o.z = sel(h.m2, c.z, o.z);
            o.w = sel(h.m3, c.w, o.w);
        }
        if (CHECK && r >= a.check_lo && r < a.check_hi) {  // scalar condition
            dmax = max2(dmax, fabsf(c.x - o.x));
            dmax = max2(dmax, fabsf(c.y - o.y));
            dmax = max2(dmax, fabsf(c.z - o.z));
            dmax = max2(dmax, fabsf(c.w - o.w));
        }
        if (TRACK) {
            const lmask cx = lanes_ne(o.x, c.x), cw = lanes_ne(o.w, c.w);
            const lmask rc = cx | cw | lanes_ne(o.y, c.y) | lanes_ne(o.z, c.z);
            chg_any |= rc;
            chg_x |= cx;
            chg_w |= cw;
            if (r == r0) chg_top = rc;      // scalar conditions
            if (r == r1 - 1) chg_bot = rc;
        }
        // non-temporal: the row is not read again before the next sweep (traffic-only build 115.6 -> 96.6 us with it)
        store_row(rout, o.x, o.y, o.z, o.w, lane16, row_off(r));
    };

    // Software pipeline, rotated by hand over a 4-row register ring so that no register moves (and hence no
    // vmcnt(0)) sit between a load and its use two rows later: while row r is computed, rows r+1 and r+2
    // are already in flight.  Whole groups of four rows run without an exit test in between (a loop with three breaks
    // costs four v_readfirstlane per row for the exit values alone); the ragged rest -- and the one-row tasks of the
    // small grids -- go through a plain loop.
    auto row_at = [&](int i) { return rfirst + dir * i; };  // i-th row of the march
    auto split = [&](const float4 &q) { return TOL ? tol_split4(q) : Split4{}; };
    // (the traffic-only diagnostic build marches like the tol kernel, so that it times THAT kernel's loads and stores)
    constexpr bool DEEP = TOL || MATH == kMathTraffic;
    if constexpr (DEEP) {
        // The tol arithmetic leaves the kernel close to the memory roofline, where what counts is the number of bytes in
        // flight: rows are loaded TWO steps ahead (two 1 KiB loads per wave outstanding instead of one; one step ahead
        // the chip ran at 4.9 TB/s with 6 waves per SIMD, 110 us per 8192^2 sweep).  Five rows rotate through five register
        // sets -- the arrays below are indexed by constants once the inner loop is unrolled, so nothing is moved --, the
        // splits ride the same ring (a row is split when it enters the window as the row below), the row sides alternate
        // between two sets of SGPRs as before; 10 = lcm(5, 2) steps per trip.
        constexpr int kAhead = kTolRowsAhead, kRing = kAhead + 3, kTrip = kTolTripRows;
        static_assert(kTrip % kRing == 0 && kTrip % 2 == 0, "the rings close after a trip");
        float4 q[kRing];
        Split4 s[kRing];
        RowSide h[2];
#pragma unroll
        for (int k = 0; k < kAhead + 2; ++k) q[k] = ld(row_at(k - 1));
        h[0] = TOL ? masks(row_at(0)) : side(row_at(0));
        if (tables_pending) {  // wave-uniform, once per wave
            math_tables_commit(tab_regs, math_lds);
            tables_pending = false;
        }
        const int ntrip = nrows / kTrip * kTrip;
        if (ntrip > 0) {
            // The strip-edge cells of a trip's ten rows -- u[r][col0 - 1] and u[r][col0 + 256], which belong to the
            // neighbouring strips -- are loaded and split ONCE per trip, one cell per lane: lane k holds the left edge cell
            // of the trip's k-th row, lane 32 + k the right one (one dword load per lane, issued a trip ahead; one
            // unpacked split per trip instead of a packed one per row in every lane: 11 VALU instructions per row less).
            // A row takes its two cells' u, q and zm from those lanes with ds_bpermute_b32 (all lanes read one lane; the
            // LDS crossbar, no VALU), as the `edge` operands of the wave shifts.
            const int hk = min(lane & 31, kTrip - 1);
            const int hcol = lane < 32 ? hcol_l : hcol_r;
            auto edge_ld = [&](int i0) -> float {  // rows of the march only: r0 <= r < r1, so r - rlo >= 0
                const int r = rfirst + dir * (i0 + hk);
                return u2f(__builtin_amdgcn_raw_buffer_load_b32(rin, (unsigned)((r - rlo) * a.pitch + hcol) * 4u, 0, 0));
            };
            int lane_zero;  // address operand of edge_from_lanes (cell_update.h); opaque so that it stays one VGPR
            asm("v_mov_b32 %0, 0" : "=v"(lane_zero));
            float ev_next = edge_ld(0);
            s[0] = split(q[0]); s[1] = split(q[1]);
            for (int i = 0; i < ntrip; i += kTrip) {
                const float ev = ev_next;
                const Split1 es = tol_split1(ev);
                if (i + kTrip < ntrip) ev_next = edge_ld(i + kTrip);
                auto trip_row = [&](auto jc) {  // j as a constant: the lane numbers are immediates of edge_from_lanes
                    constexpr int j = decltype(jc)::value;
                    q[(j + 2 + kAhead) % kRing] = ld(row_at(i + j + 1 + kAhead));
                    h[(j + 1) & 1] = TOL ? masks(row_at(i + j + 1)) : side(row_at(i + j + 1));
                    s[(j + 2) % kRing] = split(q[(j + 2) % kRing]);
                    RowSide hj = h[j & 1];
                    EdgeSplit he = {};
                    if constexpr (TOL) edge_from_lanes<j>(lane_zero, ev, es.q, es.zm, hj.l, hj.r, he.ql, he.qr, he.zl, he.zr);
                    row_step(row_at(i + j), q[j % kRing], q[(j + 1) % kRing], q[(j + 2) % kRing], hj, s[j % kRing],
                             s[(j + 1) % kRing], s[(j + 2) % kRing], he);
                };
                unrolled<kTrip>(trip_row);
            }
        }
        for (int i = ntrip; i < nrows; ++i) {  // ragged rest, and the short tasks of the small grids
            const int r = row_at(i);
            const float4 ru = ld(r - dir), rc = ld(r), rd = ld(r + dir);
            const RowSide hr = side(r);
            row_step(r, ru, rc, rd, hr, split(ru), split(rc), split(rd), TOL ? edge_split(hr) : EdgeSplit{});
        }
    } else {
    const int nfull = nrows & ~3;
    float4 q0 = ld(row_at(-1)), q1 = ld(row_at(0)), q2 = ld(row_at(1)), q3;
    RowSide sa = side(row_at(0)), sb;  // row sides run one row ahead, alternating between two sets of SGPRs
    if (tables_pending) {  // wave-uniform, once per wave
        math_tables_commit(tab_regs, math_lds);
        tables_pending = false;
    }
    const Split4 none = {};
    const EdgeSplit no_edge = {};
    if (nfull > 0) {
        for (int i = 0; i < nfull; i += 4) {
            q3 = ld(row_at(i + 2)); sb = side(row_at(i + 1));
            row_step(row_at(i), q0, q1, q2, sa, none, none, none, no_edge);
            q0 = ld(row_at(i + 3)); sa = side(row_at(i + 2));
            row_step(row_at(i + 1), q1, q2, q3, sb, none, none, none, no_edge);
            q1 = ld(row_at(i + 4)); sb = side(row_at(i + 3));
            row_step(row_at(i + 2), q2, q3, q0, sa, none, none, none, no_edge);
            q2 = ld(row_at(i + 5)); sa = side(row_at(i + 4));
            row_step(row_at(i + 3), q3, q0, q1, sb, none, none, none, no_edge);
        }
    }
    if (nfull == 0 && nrows > 0) row_step(row_at(0), q0, q1, q2, sa, none, none, none, no_edge);  // the one-row tasks of the small grids land here
    for (int i = nfull == 0 ? 1 : nfull; i < nrows; ++i) {
        const int r = row_at(i);
        row_step(r, ld(r - dir), ld(r), ld(r + dir), side(r), none, none, none, no_edge);
    }
    }  // !DEEP

    if (TRACK) {
        // wake the tiles that read what this task changed: itself, and the neighbour across each edge that changed
        const bool any = chg_any != 0;
        const bool first_col = (chg_x & 1ull) != 0;         // lane 0 holds column 0 of the strip
        const bool last_col = (chg_w >> 63) != 0;           // lane 63 holds column 255
        const bool first_row = chg_top != 0, last_row = chg_bot != 0;
        int t = task;
        bool want = any;                                                                     // lane 0: the tile itself
        if (lane == 1) { t = task - 1; want = strip > 0 && first_col; }                      // its left neighbour
        if (lane == 2) { t = task + 1; want = strip + 1 < a.nstrips && last_col; }           // its right neighbour
        if (lane == 3) { t = task - a.nstrips; want = chunk > 0 && first_row; }              // the tile above
        if (lane == 4) { t = task + a.nstrips; want = chunk + 1 < a.nchunks && last_row; }   // the tile below
        wake_push(a.wake, t, want && lane < 5);
    }
    if (listed) { if (!wake_next(cursor)) break; }
    else { vb += gridDim.x; if (vb >= a.nblocks) break; }
    }  // task loop

    if (CHECK) {
        dmax = wave_max(dmax);
        // thousands of waves end here: look first (the word only grows, so a smaller-looking value costs one atomic and a
        // stale one nothing else); same-address atomics serialise at ~7 ns each, 230 us per 8192^2 check sweep
        if (lane == 0 && dmax > 0.0f &&
            __float_as_uint(dmax) > __hip_atomic_load(a.delta_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
            atomicMax(a.delta_bits, __float_as_uint(dmax));
    }
}

// ---- fused red-black sweep: two reference iterations (both colours) in ONE pass over the data --------------------
// The in-place half-sweep above moves every cell through HBM to recompute half of them (16 B per cell-update).  This
// kernel reads u_in once and writes u_out once per FULL red-black sweep (4 B per cell-update): while a wave marches down
// its strip it first recomputes the first colour ("A") of row r+1 from old values, then the second colour ("B") of row
// r from the fresh A values of rows r-1, r, r+1 -- exactly the values the second half-sweep of the reference would
// read -- and stores row r.  Waves stay independent because each one recomputes the A cells its B cells need itself:
//   * columns: a wave loads 256 columns but owns only the 248 in lanes 1..62; lanes 0 and 63 are halo lanes whose
//     innermost A cell feeds the neighbouring owned lane through the same DPP shift (strip stride 248, ~3 % lanes);
//   * rows: a task recomputes A of the row above and below its chunk (4 extra row loads, 2 extra A rows per chunk).
// Ping-pong (in != out) is required: a neighbouring task must still find the OLD values of the rows/columns it
// recomputes.  Arithmetic per cell is the half-sweep's, so results are bit-identical to two in-place half-sweeps
// (and, with the precise math, to two iterations of the reference CPU solver).
constexpr int kFusedOut = 248;  // owned columns per wave

// TRACK (round 4): the pass with work lists, as the plain sweep has them (wake.h) -- a tile is one task of THIS kernel
// (rows_per_task rows x 248 owned columns); it runs in a pass only if the previous pass changed a value it reads, and two
// iterations reach two cells far: its own cells, the two nearest rows of the tiles above and below, the two nearest columns
// of the tiles left and right, and the one corner cell of each diagonal neighbour.  A tile that is skipped holds, in BOTH
// buffers, the values the pass would have written (it did not change in the previous pass, and nothing it reads did).  The
// lists of this tiling are separate from the plain sweep's (different tiles): driver_enqueue.hip runs every tile in the first
// pass after anything else has touched the field.
// CHECK: max |du| of the SECOND iteration of the pass (the one a check iteration is when the host makes it the last of a
// batch): |level B - level A| over the cells level B recomputes, owned lanes only.
template <int MATH, bool FMASK, bool TRACK, bool CHECK>
__global__ __launch_bounds__(kWave * kWavesPerBlock) void rb_fused2d_kernel(Sweep2dArgs a)
{
    __shared__ __attribute__((aligned(16))) double math_lds[kMathLdsDoubles];
    MathTab lds = {};
    if (MATH == kMathPrecise) lds = math_tables_load(math_lds, EPIC_CONSTS_RBPAIR(TRACK, CHECK));
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (TRACK) wake_reset_next(a.wake);
    const bool listed = TRACK && a.wake.list_in != nullptr;
    WakeCursor cursor = {};
    if (listed && !wake_begin(a.wake, lane, wave, kWavesPerBlock, cursor)) return;
    float dmax = 0.0f;
    const int rlast = a.rows - 1;
    const size_t pitch = (size_t)a.pitch;
    const int it = a.parity;  // colour A = cells with (row + col + it) odd; col is even for .x
    typedef unsigned vu4 __attribute__((ext_vector_type(4)));
    typedef const __attribute__((address_space(4))) uint64_t cu64;
    const int nstd = a.pitch >> 8;

    int vb = blockIdx.x;
    for (;;) {  // one pass per task: exactly one unless the launch is list-driven
    int task;
    if (listed) task = wake_tile(a.wake, cursor);
    else task = xcd_contiguous_block(vb, a.nblocks) * kWavesPerBlock + wave;
    if (task >= a.ntasks) {
        if (listed) break;
        vb += gridDim.x;
        if (vb >= a.nblocks) break;
        continue;
    }
    if (TRACK && lane == 0) a.wake.queued_in[task] = 0;
    const int strip = task % a.nstrips;
    const int chunk = task / a.nstrips;
    const int r0 = a.row_begin + chunk * a.rows_per_task + min(chunk, a.chunk_rem);
    const int r1 = min(r0 + a.rows_per_task + (chunk < a.chunk_rem ? 1 : 0), a.row_end);
    const int col = strip * kFusedOut - kColsPerLane + lane * kColsPerLane;  // lane 0 = left halo lane
    const int lcol = min(max(col, 0), a.pitch - kColsPerLane);
    const bool owner = lane >= 1 && lane <= kWave - 2 && col < a.pitch;
    const lmask owners = (TRACK || CHECK) ? __builtin_amdgcn_ballot_w64(owner) : 0;

    // rows through buffer descriptors with scalar row offsets, as in the plain sweep (no VALU address arithmetic)
    const int rlo = max(r0 - 2, 0);  // rows r0 - 2 .. r1 + 2 are touched
    const __amdgpu_buffer_rsrc_t rin = raw_buffer(a.in + (size_t)rlo * pitch), rout = raw_buffer(a.out + (size_t)rlo * pitch);
    const unsigned lane_off = (unsigned)lcol * 4u;
    // Stores: the halo lanes (and lanes past the last column) must not write.  A branch around the store would do -- and did,
    // until the ISA showed what it costs: with a store that may or may not have been issued the compiler can no longer count
    // the memory operations in flight behind the row it is waiting for, waits for vmcnt(0) / vmcnt(1) at the top of every step,
    // and with that for the acknowledgement of the store issued a moment ago (share of wave cycles at s_waitcnt 0.37 -> 0.28 with
    // this change alone, about 1 % of the time: profiles/r03_experiments.txt item 3a).
    // Instead every lane stores, the lanes that own nothing at an offset beyond the descriptor's range, where the hardware
    // drops the write (raw buffers check voffset against num_records = 2 GiB): one store per step, exact counts.
    const unsigned store_off = owner ? lane_off : 0x80000000u;
    auto row_off = [&](int r) -> unsigned { return (unsigned)((r - rlo) * a.pitch) * 4u; };  // r already clamped
    auto ld = [&](int r) -> float4 {
        r = min(max(r, 0), rlast);
        const vu4 q = __builtin_amdgcn_raw_buffer_load_b128(rin, lane_off, row_off(r), 0);
        return make_float4(u2f(q.x), u2f(q.y), u2f(q.z), u2f(q.w));
    };
    // Lane masks of row r for THIS kernel's lane -> column mapping (lane L holds quad strip * 62 - 1 + L of the row,
    // the stored masks are cut at multiples of 64 quads): a funnel shift of two neighbouring words, all scalar.
    // Lanes outside the row (lane 0 of strip 0, lanes past the last column) get arbitrary bits: nothing they compute
    // is stored or reaches an unlocked cell.
    // In two phases, as in tol_fused_pass below: the words are FETCHED a step before they are CUT (no scalar-memory round trip
    // in the middle of a step); FMASK: the library's second copy of the masks, already cut for this mapping (kernels.h).
    struct RowMask { lmask m0, m1, m2, m3; };
    struct RowMaskRaw { lmask lo0, lo1, lo2, lo3, hi0, hi1, hi2, hi3; };
    const int g0 = strip * (kFusedOut / kColsPerLane) - 1;
    const int sw = max(g0, 0) >> 6, sh = max(g0, 0) & 63, sw1 = min(sw + 1, nstd - 1);
    auto mask_fetch = [&](int r) -> RowMaskRaw {
        r = min(max(r, 0), rlast);
        if (FMASK) {
            cu64 *mk = (cu64 *)a.maskf + ((size_t)r * a.nstrips + strip) * 4;
            return RowMaskRaw{mk[0], mk[1], mk[2], mk[3], 0, 0, 0, 0};
        }
        cu64 *lo = (cu64 *)a.maskw + ((size_t)r * nstd + sw) * 4, *hi = (cu64 *)a.maskw + ((size_t)r * nstd + sw1) * 4;
        return RowMaskRaw{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    };
    auto mask_cut = [&](const RowMaskRaw &w) -> RowMask {
        if (FMASK) return RowMask{w.lo0, w.lo1, w.lo2, w.lo3};
        auto cut = [&](lmask lo, lmask hi) -> lmask {   // (lo >> sh) | (hi << (64 - sh)), branch-free for sh = 0
            const lmask m = (lo >> sh) | ((hi << 1) << (63 - sh));
            return g0 < 0 ? m << 1 : m;
        };
        return RowMask{cut(w.lo0, w.hi0), cut(w.lo1, w.hi1), cut(w.lo2, w.hi2), cut(w.lo3, w.hi3)};
    };
    auto row_mask = [&](int r) -> RowMask { return mask_cut(mask_fetch(r)); };
    // what this task changed, as the tiles around it need to know it (TRACK): lane masks and flags in scalar registers
    lmask chg_any = 0, chg_top = 0, chg_bot = 0;
    bool chg_left = false, chg_right = false, c_tl = false, c_tr = false, c_bl = false, c_br = false;
    constexpr lmask kFirstOwned = 2ull, kLastOwned = 1ull << (kWave - 2);   // lanes 1 and 62
    // One colour of one row.  second = false: colour A (iteration it), true: colour B (iteration it + 1).
    // owned: the row belongs to this task (level A is also computed for the row above and the row below the chunk).
    auto stage = [&](int r, bool second, bool owned, const float4 &up, const float4 &c, const float4 &dn, const RowMask &k) -> float4 {
        float4 o = c;
        const bool odd_cols = ((((r + it) & 1) == 0) != second);  // scalar
        const bool counts = CHECK && r >= a.check_lo && r < a.check_hi;   // scalar: a slab's ghost rows are swept but kept out of max |du|
        lmask e0 = 0, e1 = 0;   // lanes whose first / second recomputed cell changed its bits
        if (odd_cols) {
            const float rt = wave_from_right(c.x, 0.0f);
            const float ny = cell_update_2d<MATH>(up.y, dn.y, c.x, c.z, lds);
            const float nw = cell_update_2d<MATH>(up.w, dn.w, c.z, rt, lds);
            o.y = sel(k.m1, c.y, ny);
            o.w = sel(k.m3, c.w, nw);
            if (TRACK) { e0 = lanes_ne(o.y, c.y); e1 = lanes_ne(o.w, c.w); }
            if (CHECK && second && counts) dmax = max2(dmax, sel(owners, max2(fabsf(c.y - o.y), fabsf(c.w - o.w)), 0.0f));
        } else {
            const float lf = wave_from_left(c.w, 0.0f);
            const float nx = cell_update_2d<MATH>(up.x, dn.x, lf, c.y, lds);
            const float nz = cell_update_2d<MATH>(up.z, dn.z, c.y, c.w, lds);
            o.x = sel(k.m0, c.x, nx);
            o.z = sel(k.m2, c.z, nz);
            if (TRACK) { e0 = lanes_ne(o.x, c.x); e1 = lanes_ne(o.z, c.z); }
            if (CHECK && second && counts) dmax = max2(dmax, sel(owners, max2(fabsf(c.x - o.x), fabsf(c.z - o.z)), 0.0f));
        }
        if (TRACK && owned) {   // (scalar throughout)
            const lmask rc = (e0 | e1) & owners;
            chg_any |= rc;
            if (r < r0 + 2) chg_top |= rc;
            if (r >= r1 - 2) chg_bot |= rc;
            // the two owned columns next to the left neighbour are lane 1's x and y (e0 in either case), next to the right one
            // lane 62's z and w (e1); the corner cells are lane 1's x (even columns) and lane 62's w (odd columns)
            const bool l = (e0 & kFirstOwned) != 0, rr = (e1 & kLastOwned & owners) != 0;
            chg_left |= l;
            chg_right |= rr;
            if (r == r0) { c_tl |= l && !odd_cols; c_tr |= rr && odd_cols; }
            if (r == r1 - 1) { c_bl |= l && !odd_cols; c_br |= rr && odd_cols; }
        }
        return o;
    };

    // prologue: colour A of rows r0-1 and r0 (old neighbours only: the other colour has not moved yet)
    const float4 om2 = ld(r0 - 2), om1 = ld(r0 - 1), o0 = ld(r0);
    float4 oa = ld(r0 + 1), ob = ld(r0 + 2), oc;  // old rows r+1, r+2, r+3
    RowMask kcur = row_mask(r0), knext = row_mask(r0 + 1);
    float4 ma = stage(r0 - 1, false, false, om2, om1, o0, row_mask(r0 - 1)), mb = stage(r0, false, true, om1, o0, oa, kcur), mc;
    // One row r: `mp`, `mq` = colour A of rows r-1, r; `o1`, `o2` = old rows r+1, r+2.  Leaves colour A of row r+1 in
    // `mr` and old row r+3 in `o3`.  The three A rows and the three old rows rotate through fixed registers (the loop
    // is unrolled by three), so nothing is moved.
    auto step = [&](int r, const float4 &mp, const float4 &mq, float4 &mr, const float4 &o1, const float4 &o2, float4 &o3) {
        o3 = ld(r + 3);
        const RowMaskRaw kraw = mask_fetch(r + 2);             // cut at the end of the step
        mr = stage(r + 1, false, r + 1 < r1, mq, o1, o2, knext);           // colour A of row r+1: up = row r (its B cells still old)
        const float4 x = stage(r, true, true, mp, mq, mr, kcur);    // colour B of row r from the fresh A cells around it
        kcur = knext;
        knext = mask_cut(kraw);
        store_row(rout, x.x, x.y, x.z, x.w, store_off, row_off(r));  // non-temporal, as in the plain sweep
    };
    int r = r0;
    for (; r + 3 <= r1; r += 3) {
        step(r, ma, mb, mc, oa, ob, oc);
        step(r + 1, mb, mc, ma, ob, oc, oa);
        step(r + 2, mc, ma, mb, oc, oa, ob);
    }
    for (; r < r1; ++r) {  // at most two rows
        step(r, ma, mb, mc, oa, ob, oc);
        ma = mb; mb = mc; oa = ob; ob = oc;
    }

    if (TRACK) {
        // wake the tiles that read what this task changed: itself, the four across its edges, the four across its corners
        const bool up_ok = chunk > 0, dn_ok = chunk + 1 < a.nchunks, lf_ok = strip > 0, rt_ok = strip + 1 < a.nstrips;
        int t = task;
        bool want = chg_any != 0;                                                        // lane 0: the tile itself
        if (lane == 1) { t = task - 1; want = lf_ok && chg_left; }
        if (lane == 2) { t = task + 1; want = rt_ok && chg_right; }
        if (lane == 3) { t = task - a.nstrips; want = up_ok && chg_top != 0; }
        if (lane == 4) { t = task + a.nstrips; want = dn_ok && chg_bot != 0; }
        if (lane == 5) { t = task - a.nstrips - 1; want = up_ok && lf_ok && c_tl; }
        if (lane == 6) { t = task - a.nstrips + 1; want = up_ok && rt_ok && c_tr; }
        if (lane == 7) { t = task + a.nstrips - 1; want = dn_ok && lf_ok && c_bl; }
        if (lane == 8) { t = task + a.nstrips + 1; want = dn_ok && rt_ok && c_br; }
        wake_push(a.wake, t, want && lane < 9);
    }
    if (listed) { if (!wake_next(cursor)) break; }
    else { vb += gridDim.x; if (vb >= a.nblocks) break; }
    }  // task loop

    if (CHECK) {
        dmax = wave_max(dmax);
        if (lane == 0 && dmax > 0.0f &&
            __float_as_uint(dmax) > __hip_atomic_load(a.delta_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
            atomicMax(a.delta_bits, __float_as_uint(dmax));
    }
}

// ---- fused Jacobi sweep, tol math: TWO iterations in one pass over the data -----------------------------------------
// The single tol sweep runs at ~0.9 of what its own loads and stores cost with no arithmetic at all (the traffic-only
// build), and those at ~0.93 of a flat copy: 8 B of HBM traffic per cell-update is the bill, whatever the VALU does.  This
// kernel pays it once for two iterations: while a wave marches down its strip it computes iteration k+1 of row r+1 from
// rows r .. r+2 of u_k (level A, kept in registers only), then iteration k+2 of row r from rows r-1 .. r+1 of level A
// (level B), and stores row r.  4 B per cell-update; the price is arithmetic done twice where tasks meet:
//   * columns: a wave loads 256 columns and owns the 248 in lanes 1..62 (as rb_fused2d_kernel); the inner cell of each
//     halo lane is a correct level-A value (its own neighbours are all inside the 256), everything further out is unused;
//   * rows: a task computes level A for the row above and the row below its chunk as well (rows r0-1 .. r1 of level A
//     from rows r0-2 .. r1+1 of u_k).
// Every cell value is produced by the same tol_update_2d on the same inputs as in two single sweeps, so the result is
// bit-identical to them (tests/test_gpu_tol.py compares odd and even iteration counts with the checker).  Ping-pong
// (in != out) as always for Jacobi.  Rings: u_k rows over six register sets (loaded two steps ahead), their splits, the
// level-A rows and THEIR splits over three each, the lane masks of three rows in SGPRs; six steps per trip, every step
// behind a scalar test of its row number so that a task may have any number of rows.
#ifndef EPIC_FUSED_MIN_WAVES  // build knob (A/B)
#define EPIC_FUSED_MIN_WAVES 4
#endif
// RB = false: two Jacobi iterations.  RB = true: two iterations of the reference's red-black scheme (both colours) -- the
// same march with half of the cells recomputed at each level: level A updates the cells of row r+1 with (row + col + it)
// odd from the old rows (their neighbours all have the other colour and have not moved yet), level B the cells of row r
// with (row + col + it + 1) odd from the level-A rows; a cell that a level does not touch passes through it.  Every
// cell is recomputed once per pass (4 B of HBM traffic per cell-update against 16 for the in-place half-sweep), the rows
// are split twice (before level A, and after it for level B).  Bit-identical to two in-place half-sweeps.
template <bool RB, bool FMASK, bool TRACK = false, bool CHECK = false>
__device__ __forceinline__ void tol_fused_pass(const Sweep2dArgs &a, TolLnEntry *math_lds)
{
    TolLn<4>::stage(math_lds);  // the whole workgroup, one barrier: before any wave may find itself without a task
    const TolLnEntry *const tl = math_lds;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // TRACK (round 4): the pass with work lists of its own tiling, exactly as rb_fused2d_kernel has them (see there: which tiles
    // a task wakes, why a skipped tile holds the right values in both buffers); CHECK: max |du| of the pass's SECOND iteration.
    if (TRACK) wake_reset_next(a.wake);
    const bool listed = TRACK && a.wake.list_in != nullptr;
    WakeCursor cursor = {};
    float dmax = 0.0f;
    const bool have_work = !listed || wake_begin(a.wake, lane, wave, kWavesPerBlock, cursor);
    // the launch holds as many workgroups as the chip keeps resident; each walks its share of the logical blocks (or of the lists)
    if (have_work)
    for (int vb = blockIdx.x;;) {
    int task;
    if (listed) task = wake_tile(a.wake, cursor);
    else task = xcd_contiguous_block(vb, a.nblocks) * kWavesPerBlock + wave;
    if (task >= a.ntasks) {
        if (listed) break;
        vb += gridDim.x;
        if (vb >= a.nblocks) break;
        continue;
    }
    if (TRACK && lane == 0) a.wake.queued_in[task] = 0;
    const int strip = task % a.nstrips;
    const int chunk = task / a.nstrips;
    const int r0 = a.row_begin + chunk * a.rows_per_task + min(chunk, a.chunk_rem);
    const int r1 = min(r0 + a.rows_per_task + (chunk < a.chunk_rem ? 1 : 0), a.row_end);
    const int col = strip * kFusedOut - kColsPerLane + lane * kColsPerLane;  // lane 0 = left halo lane
    const int lcol = min(max(col, 0), a.pitch - kColsPerLane);
    const bool owner = lane >= 1 && lane <= kWave - 2 && col < a.pitch;
    const lmask owners = (TRACK || CHECK) ? __builtin_amdgcn_ballot_w64(owner) : 0;
    lmask chg_any = 0, chg_top = 0, chg_bot = 0;   // what this task changed (TRACK): lane masks and flags in scalar registers
    bool chg_left = false, chg_right = false, c_tl = false, c_tr = false, c_bl = false, c_br = false;
    constexpr lmask kFirstOwned = 2ull, kLastOwned = 1ull << (kWave - 2);   // lanes 1 and 62
    const int rlast = a.rows - 1;
    const size_t pitch = (size_t)a.pitch;
    typedef unsigned vu4 __attribute__((ext_vector_type(4)));

    const int rlo = max(r0 - 2, 0);  // rows r0 - 2 .. r1 + 4 are touched (clamped to the grid)
    const __amdgpu_buffer_rsrc_t rin = raw_buffer(a.in + (size_t)rlo * pitch), rout = raw_buffer(a.out + (size_t)rlo * pitch);
    const unsigned lane_off = (unsigned)lcol * 4u;
    // Stores: the halo lanes (and lanes past the last column) must not write.  A branch around the store would do -- and did,
    // until the ISA showed what it costs: with a store that may or may not have been issued the compiler can no longer count
    // the memory operations in flight behind the row it is waiting for, waits for vmcnt(0) / vmcnt(1) at the top of every step,
    // and with that for the acknowledgement of the store issued a moment ago (share of wave cycles at s_waitcnt 0.37 -> 0.28 with
    // this change alone, about 1 % of the time: profiles/r03_experiments.txt item 3a).
    // Instead every lane stores, the lanes that own nothing at an offset beyond the descriptor's range, where the hardware
    // drops the write (raw buffers check voffset against num_records = 2 GiB): one store per step, exact counts.
    const unsigned store_off = owner ? lane_off : 0x80000000u;
    auto row_off = [&](int r) -> unsigned { return (unsigned)((r - rlo) * a.pitch) * 4u; };  // r already clamped
    auto ld = [&](int r) -> float4 {
        r = min(max(r, 0), rlast);
        const vu4 q = __builtin_amdgcn_raw_buffer_load_b128(rin, lane_off, row_off(r), 0);
        return make_float4(u2f(q.x), u2f(q.y), u2f(q.z), u2f(q.w));
    };
    // lane masks of a row for this kernel's lane -> column mapping (lane L holds quad strip * 62 - 1 + L of the row, the stored
    // masks are cut at multiples of 64 quads): a funnel shift of two neighbouring mask words, all scalar.  In two phases: the
    // eight words are FETCHED (two s_load_dwordx8, nothing depends on them yet) at the top of a step and CUT at its end.
    // (As one function with a test of `sh` between the loads -- round 2 -- the wave made eight scalar-memory round trips in a
    // row at the end of every step: the larger part of the 28-37 % of wave cycles spent at s_waitcnt.)
    typedef const __attribute__((address_space(4))) uint64_t cu64;
    struct RowMask { lmask m0, m1, m2, m3; };
    struct RowMaskRaw { lmask lo0, lo1, lo2, lo3, hi0, hi1, hi2, hi3; };
    const int nstd = a.pitch >> 8;
    const int g0 = strip * (kFusedOut / kColsPerLane) - 1;
    const int sw = max(g0, 0) >> 6, sh = max(g0, 0) & 63, sw1 = min(sw + 1, nstd - 1);
    // (FMASK: the library keeps a second copy of the masks already cut for this mapping -- kernels.h, fused layout --, one
    // s_load_dwordx8 per row and no shifts; the funnel remains for callers that hold the standard layout only)
    auto mask_fetch = [&](int r) -> RowMaskRaw {
        r = min(max(r, 0), rlast);
        if (FMASK) {
            cu64 *mk = (cu64 *)a.maskf + ((size_t)r * a.nstrips + strip) * 4;
            return RowMaskRaw{mk[0], mk[1], mk[2], mk[3], 0, 0, 0, 0};
        }
        cu64 *lo = (cu64 *)a.maskw + ((size_t)r * nstd + sw) * 4, *hi = (cu64 *)a.maskw + ((size_t)r * nstd + sw1) * 4;
        return RowMaskRaw{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    };
    auto mask_cut = [&](const RowMaskRaw &w) -> RowMask {
        if (FMASK) return RowMask{w.lo0, w.lo1, w.lo2, w.lo3};
        // (lo >> sh) | (hi << (64 - sh)) without a branch for sh = 0: (hi << 1) << (63 - sh) is 0 there
        auto cut = [&](lmask lo, lmask hi) -> lmask {
            const lmask m = (lo >> sh) | ((hi << 1) << (63 - sh));
            return g0 < 0 ? m << 1 : m;
        };
        return RowMask{cut(w.lo0, w.hi0), cut(w.lo1, w.hi1), cut(w.lo2, w.hi2), cut(w.lo3, w.hi3)};
    };
    auto row_mask = [&](int r) -> RowMask { return mask_cut(mask_fetch(r)); };
    // One iteration of one row from its three rows and their splits.  The outer cells of the halo lanes have no neighbour
    // on one side: the shift gives them zero bits there (bound_ctrl: one instruction, no edge operand to set up) -- their
    // results are never stored and never read by an owned cell.
    auto shl = [](float v) { return u2f(__builtin_amdgcn_mov_dpp(f2u(v), 0x138 /* wave_shr:1 */, 0xf, 0xf, true)); };  // from the left
    auto shr = [](float v) { return u2f(__builtin_amdgcn_mov_dpp(f2u(v), 0x130 /* wave_shl:1 */, 0xf, 0xf, true)); };  // from the right
    // (red-black: `odd_cols` says which half of the row this level recomputes -- .y / .w or .x / .z; scalar)
    // row: the row this level recomputes; second: level B (the pass's second iteration); owned: the row belongs to this task
    auto level = [&](const float4 &up, const float4 &c, const float4 &dn, const Split4 &su, const Split4 &sc, const Split4 &sd,
                     const RowMask &k, bool odd_cols, int row, bool second, bool owned) -> float4 {
        float4 o = c;
        // the cells this level updates, two at a time in the three phases of cell_update.h: the table reads of one pair are in
        // flight while the other pair is worked on
        const bool even = !RB || !odd_cols;
        // (the neighbours in the next lanes: only q is shifted as a move, u and n ride on the instructions that use them)
        auto pre_xz = [&] {
            return tol_pre2_2d_left(up.x, dn.x, c.w, c.y, su.qx, su.nx, sd.qx, sd.nx, shl(sc.qw), sc.nw, sc.qy, sc.ny,
                                    up.z, dn.z, c.y, c.w, su.qz, su.nz, sd.qz, sd.nz, sc.qy, sc.ny, sc.qw, sc.nw);
        };
        auto pre_yw = [&] {
            return tol_pre2_2d_right(up.y, dn.y, c.x, c.z, su.qy, su.ny, sd.qy, sd.ny, sc.qx, sc.nx, sc.qz, sc.nz,
                                     up.w, dn.w, c.z, c.x, su.qw, su.nw, sd.qw, sd.nw, sc.qz, sc.nz, shr(sc.qx), sc.nx);
        };
        float nx, ny, nz, nw;
        TolLnRaw ea, eb;
        if (!RB) {
            const TolPre2 pxz = pre_xz();
            TolLnPair f0 = tol_ln_issue<4>(pxz, tl);
            const TolPre2 pyw = pre_yw();
            TolLnPair f1 = tol_ln_issue<4>(pyw, tl);
            tol_ln_wait<2>(f0, ea, eb);
            tol_post2(pxz, ea, eb, kLn4, nx, nz);
            o.x = sel(k.m0, c.x, nx);
            o.z = sel(k.m2, c.z, nz);
            tol_ln_wait<0>(f1, ea, eb);
            tol_post2(pyw, ea, eb, kLn4, ny, nw);
            o.y = sel(k.m1, c.y, ny);
            o.w = sel(k.m3, c.w, nw);
        } else if (even) {
            const TolPre2 pxz = pre_xz();
            TolLnPair f0 = tol_ln_issue<4>(pxz, tl);
            tol_ln_wait<0>(f0, ea, eb);
            tol_post2(pxz, ea, eb, kLn4, nx, nz);
            o.x = sel(k.m0, c.x, nx);
            o.z = sel(k.m2, c.z, nz);
        } else {
            const TolPre2 pyw = pre_yw();
            TolLnPair f1 = tol_ln_issue<4>(pyw, tl);
            tol_ln_wait<0>(f1, ea, eb);
            tol_post2(pyw, ea, eb, kLn4, ny, nw);
            o.y = sel(k.m1, c.y, ny);
            o.w = sel(k.m3, c.w, nw);
        }
        if (CHECK && second && row >= a.check_lo && row < a.check_hi) {   // (scalar; a slab's ghost rows do not count)
            const float d = max2(max2(fabsf(c.x - o.x), fabsf(c.y - o.y)), max2(fabsf(c.z - o.z), fabsf(c.w - o.w)));
            dmax = max2(dmax, sel(owners, d, 0.0f));
        }
        if (TRACK && owned) {   // (scalar throughout; a cell the level does not touch compares equal)
            const lmask e0 = lanes_ne(o.x, c.x) | lanes_ne(o.y, c.y), e1 = lanes_ne(o.z, c.z) | lanes_ne(o.w, c.w);
            const lmask rc = (e0 | e1) & owners;
            chg_any |= rc;
            if (row < r0 + 2) chg_top |= rc;
            if (row >= r1 - 2) chg_bot |= rc;
            // next to the left neighbour: lane 1's x and y; next to the right one: lane 62's z and w; corner cells: lane 1's x, lane 62's w
            chg_left |= (e0 & kFirstOwned) != 0;
            chg_right |= (e1 & kLastOwned & owners) != 0;
            if (row == r0 || row == r1 - 1) {
                const bool cl = (lanes_ne(o.x, c.x) & kFirstOwned) != 0, cr = (lanes_ne(o.w, c.w) & kLastOwned & owners) != 0;
                if (row == r0) { c_tl |= cl; c_tr |= cr; }
                if (row == r1 - 1) { c_bl |= cl; c_br |= cr; }
            }
        }
        return o;
    };
    // which half a level recomputes in row `row`: level A belongs to iteration a.parity, level B to the one after it; the
    // cells with (row + col + iteration) odd are the active ones (harmonic_cpu.cpp:46-51), col is even for .x / .z
    auto odd_cols_of = [&](int row, int level_index) { return ((row + a.parity + level_index) & 1) == 0; };

    // Ring slots (row numbers relative to r0, trips start at multiples of 6):  u_k row x -> u[(x + 2) % 6],
    // its split -> su[(x + 2) % 3],  level-A row x and its split -> m / sm[(x + 1) % 3],  masks of row x -> k[(x + 1) % 3].
    constexpr int kTrip = 6;
    float4 u[6], m[3];
    Split4 su[3], sm[3];
    RowMask k[3];
#pragma unroll
    for (int x = 0; x < 6; ++x) u[x] = ld(min(r0 - 2 + x, r1 + 1));
    k[0] = row_mask(r0 - 1); k[1] = row_mask(r0); k[2] = row_mask(r0 + 1);
    su[0] = tol_split4(u[0]); su[1] = tol_split4(u[1]); su[2] = tol_split4(u[2]);
    m[0] = level(u[0], u[1], u[2], su[0], su[1], su[2], k[0], odd_cols_of(r0 - 1, 0), r0 - 1, false, false);   // level A of row r0 - 1
    sm[0] = tol_split4(m[0]);
    su[0] = tol_split4(u[3]);                                    // the split of row r0 - 2 is done with
    m[1] = level(u[1], u[2], u[3], su[1], su[2], su[0], k[1], odd_cols_of(r0, 0), r0, false, true);   // level A of row r0
    sm[1] = tol_split4(m[1]);
    const int nrows = r1 - r0;
    // (red-black: the half a level recomputes is a scalar test per level.  Two straight-line versions of the trip, chosen by
    // the one task-uniform bit that fixes the whole pattern, were built and measured: the scheduler then interleaves so much
    // that the kernel needs ~170 VGPRs and still spills -- 82 us per iteration at three waves per SIMD against 76 like this.)
    for (int i = 0; i < nrows; i += kTrip) {
        auto step = [&](auto jc) {
            constexpr int j = decltype(jc)::value;
            if (i + j < nrows) {  // scalar
                const int r = r0 + i + j;
                u[j % 6] = ld(min(r + 4, r1 + 1));   // slot of row r - 2; past r1 + 1 (nothing needs those rows) the last row again: a cache hit
                const RowMaskRaw kraw = mask_fetch(r + 2);                       // cut at the end of the step
                su[(j + 1) % 3] = tol_split4(u[(j + 4) % 6]);                    // row r + 2 (slot of row r - 1's split)
                m[(j + 2) % 3] = level(u[(j + 2) % 6], u[(j + 3) % 6], u[(j + 4) % 6], su[(j + 2) % 3], su[j % 3],
                                       su[(j + 1) % 3], k[(j + 2) % 3], odd_cols_of(r + 1, 0), r + 1, false, r + 1 < r1);   // level A of row r + 1
                sm[(j + 2) % 3] = tol_split4(m[(j + 2) % 3]);
                const float4 x = level(m[j % 3], m[(j + 1) % 3], m[(j + 2) % 3], sm[j % 3], sm[(j + 1) % 3], sm[(j + 2) % 3],
                                       k[(j + 1) % 3], odd_cols_of(r, 1), r, true, true);       // level B of row r
                k[j % 3] = mask_cut(kraw);                                       // slot of row r - 1's masks
                store_row(rout, x.x, x.y, x.z, x.w, store_off, row_off(r));
            }
        };
        unrolled<kTrip>(step);
    }
    if (TRACK) {
        // wake the tiles that read what this task changed: itself, the four across its edges, the four across its corners
        const int nchunks = a.nchunks;
        const bool up_ok = chunk > 0, dn_ok = chunk + 1 < nchunks, lf_ok = strip > 0, rt_ok = strip + 1 < a.nstrips;
        int t = task;
        bool want = chg_any != 0;
        if (lane == 1) { t = task - 1; want = lf_ok && chg_left; }
        if (lane == 2) { t = task + 1; want = rt_ok && chg_right; }
        if (lane == 3) { t = task - a.nstrips; want = up_ok && chg_top != 0; }
        if (lane == 4) { t = task + a.nstrips; want = dn_ok && chg_bot != 0; }
        if (lane == 5) { t = task - a.nstrips - 1; want = up_ok && lf_ok && c_tl; }
        if (lane == 6) { t = task - a.nstrips + 1; want = up_ok && rt_ok && c_tr; }
        if (lane == 7) { t = task + a.nstrips - 1; want = dn_ok && lf_ok && c_bl; }
        if (lane == 8) { t = task + a.nstrips + 1; want = dn_ok && rt_ok && c_br; }
        wake_push(a.wake, t, want && lane < 9);
    }
    if (listed) { if (!wake_next(cursor)) break; }
    else { vb += gridDim.x; if (vb >= a.nblocks) break; }
    }  // tasks
    if (CHECK) {
        dmax = wave_max(dmax);
        if (lane == 0 && dmax > 0.0f &&
            __float_as_uint(dmax) > __hip_atomic_load(a.delta_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
            atomicMax(a.delta_bits, __float_as_uint(dmax));
    }
}

template <int MATH, bool FMASK>
__global__ __launch_bounds__(kWave * kWavesPerBlock, EPIC_FUSED_MIN_WAVES) void jacobi_fused2d_kernel(Sweep2dArgs a)
{
    static_assert(MATH == kMathTol, "the fused Jacobi pass exists for the tol math");
    __shared__ TolLnEntry math_lds[TolLn<4>::kEntries];
    tol_fused_pass<false, FMASK>(a, math_lds);
}

// two red-black iterations of the tol math in one pass (the precise / fast math: rb_fused2d_kernel above)
template <bool FMASK>
__global__ __launch_bounds__(kWave * kWavesPerBlock, EPIC_FUSED_MIN_WAVES) void rb_tol_fused2d_kernel(Sweep2dArgs a)
{
    __shared__ TolLnEntry math_lds[TolLn<4>::kEntries];
    tol_fused_pass<true, FMASK>(a, math_lds);
}

// the same passes with work lists (TRACK) and / or the max |du| of their second iteration (CHECK): instantiations of their own, so
// that the plain ones keep their registers (these ask for three waves per SIMD instead of four: the change masks and the running
// maximum need a few registers more, and no sweep kernel may use scratch)
template <bool RB, bool TRACK, bool CHECK>
__global__ __launch_bounds__(kWave * kWavesPerBlock, 3) void tol_fused2d_tracked_kernel(Sweep2dArgs a)
{
    __shared__ TolLnEntry math_lds[TolLn<4>::kEntries];
    tol_fused_pass<RB, true, TRACK, CHECK>(a, math_lds);   // (needs the fused mask layout: the library always has it)
}

// The masks of a grid cut for the fused passes' lane -> column mapping (kernels.h: fused layout), from the standard lane
// masks: one thread per (row, fused strip, word), the funnel shift the passes would otherwise do per row and wave.
__global__ void fuse_masks_2d_kernel(const uint32_t *maskw, int rows, int pitch, uint32_t *maskf)
{
    const int nstd = pitch >> 8, nfused = (pitch + kFusedOut - 1) / kFusedOut;
    const size_t n = (size_t)rows * nfused * 4, i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int j = (int)(i & 3), strip = (int)((i >> 2) % nfused);
    const size_t r = (i >> 2) / nfused;
    const int g0 = strip * (kFusedOut / kColsPerLane) - 1;
    const int sw = max(g0, 0) >> 6, sh = max(g0, 0) & 63, sw1 = min(sw + 1, nstd - 1);
    const unsigned long long *std64 = reinterpret_cast<const unsigned long long *>(maskw);
    const unsigned long long lo = std64[(r * nstd + sw) * 4 + j], hi = std64[(r * nstd + sw1) * 4 + j];
    unsigned long long m = (lo >> sh) | ((hi << 1) << (63 - sh));
    if (g0 < 0) m <<= 1;
    reinterpret_cast<unsigned long long *>(maskf)[i] = m;
}

// uint32-per-cell mask (the ABI's format, rows x cols, unpitched) -> lane masks (kernels.h).  One wave per (row,
// strip): lane L looks at its four cells, four ballots are the four words.  Border cells and the padding beyond
// `cols` are forced locked (harmonic.h:35-37 "assumes border values are locked").
__global__ __launch_bounds__(kWave * kWavesPerBlock) void pack_mask_2d_kernel(const uint32_t *locked, int rows, int cols,
                                                                              int pitch, int ghost_top, int ghost_bottom,
                                                                              uint32_t *maskw)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int nstrips = pitch >> 8;
    const int strip = blockIdx.y * kWavesPerBlock + (threadIdx.x >> 6);
    const int r = blockIdx.x;  // rows ride on grid.x: grid.y stops at 65535
    if (strip >= nstrips || r >= rows) return;  // wave-uniform
    const bool ghost = (ghost_top && r == 0) || (ghost_bottom && r == rows - 1);
    const bool border_row = (!ghost_top && r == 0) || (!ghost_bottom && r == rows - 1);
    unsigned long long *out = reinterpret_cast<unsigned long long *>(maskw) + ((size_t)r * nstrips + strip) * 4;
#pragma unroll
    for (int j = 0; j < kColsPerLane; ++j) {
        const int c = strip * kStripCols + lane * kColsPerLane + j;
        bool lk = true;
        if (c < cols && !border_row && !ghost && c != 0 && c != cols - 1) lk = locked[(size_t)r * cols + c] != 0;
        const unsigned long long m = __ballot(lk);
        if (lane == j) out[j] = m;
    }
}

__global__ void fill_kernel(float *p, size_t n, float v)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) p[i] = v;
}

// Sparse edits (harmonic_utilities_gpu.cu:38-63): one thread per edit, into the CURRENT buffer and the
// lane masks.  Border cells stay locked in the mask whatever the edit says (the sweep never updates them).
// Slab form (multi-device): the buffer holds global rows [row0, row0 + rows) of a grid of grid_rows rows; edits outside
// are somebody else's, the border test is the global one, and the outermost ghost rows stay pinned.
__global__ void set_cells_2d_kernel(float *u, uint32_t *maskw, int rows, int cols, int pitch, unsigned k,
                                    const unsigned *v, const unsigned *types, int row0, int grid_rows, int pin_top,
                                    int pin_bottom)
{
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= k) return;
    const unsigned x = v[2 * i], gy = v[2 * i + 1];
    if (gy >= (unsigned)grid_rows || x >= (unsigned)cols) return;
    const int y = (int)gy - row0;
    if (y < 0 || y >= rows) return;
    const unsigned t = types[i];
    if (t > 2u) return;
    const float val = (t == 0u) ? 0.0f : -1e6f;
    const bool lock = (t != 2u) || x == 0 || gy == 0 || x == (unsigned)cols - 1 || gy == (unsigned)grid_rows - 1 ||
                      (pin_top && y == 0) || (pin_bottom && y == rows - 1);
    u[(size_t)y * pitch + x] = val;
    uint32_t *w = maskw + mask_word_2d((unsigned)y, x, (unsigned)pitch);
    const uint32_t bit = 1u << mask_bit_2d(x);
    if (lock) atomicOr(w, bit);
    else atomicAnd(w, ~bit);
}

// libm-replica check: out[i] = which ? ln(in[i]) : exp(in[i]) with the PRECISE device routines (test hook).
__global__ void eval_math_kernel(const float *in, float *out, size_t n, int which)
{
    __shared__ __attribute__((aligned(16))) double math_lds[kMathLdsDoubles];
    const MathTab plain = math_tables_load(math_lds), keep = math_tables_at(math_lds, kConstsKeep);   // which & 2: the kept-addend form (MathTab::consts)
    // all 64 lanes stay active (as in the sweeps, whose update assumes it), so the loop count is wave-uniform and
    // out-of-range lanes work on a clamped index and skip the store
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t first = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t trips = (n + stride - 1) / stride;
    for (size_t t = 0; t < trips; t++) {
        const size_t i = first + t * stride;
        const float x = in[i < n ? i : n - 1];
        float r;
        if (which & 2) r = (which & 1) == 0 ? precise_exp(x, keep) : precise_ln(x, keep);
        else r = (which & 1) == 0 ? precise_exp(x, plain) : precise_ln(x, plain);
        if (i < n) out[i] = r;
    }
}

}  // namespace

// CUs of the CURRENT device (cached per device ordinal: the slabs of the multi-device mode may sit on devices that differ, and their
// issuing threads come here concurrently).  0: unknown.
static int current_device_cus(int *dev_out = nullptr)
{
    static std::mutex mu;
    static std::map<int, int> cache;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    if (dev_out) *dev_out = dev;
    std::lock_guard<std::mutex> lk(mu);
    auto it = cache.find(dev);
    if (it != cache.end()) return it->second;
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1) {
        (void)hipGetLastError();
        n = 0;
    }
    return cache[dev] = n;
}

int resident_blocks_of(const void *kernel)
{
    static std::mutex mu;
    static std::map<std::pair<int, const void *>, int> cache;   // per device and kernel
    int dev = 0;
    const int cus = current_device_cus(&dev);
    std::lock_guard<std::mutex> lk(mu);
    const auto key = std::make_pair(dev, kernel);
    auto it = cache.find(key);
    if (it != cache.end()) return it->second;
    int blocks = 0;
    if (cus < 1 || hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, kernel, kWave * kWavesPerBlock, 0) != hipSuccess || blocks < 1) {
        (void)hipGetLastError();
        return cache[key] = 0;   // unknown: the caller's default
    }
    return cache[key] = blocks * cus;
}

// The fused passes without work lists: how many chunks the rows are cut into.  A launch's workgroups (four waves, one per SIMD) are dealt
// over the CUs, so what a launch costs is the CU with the most blocks: ceil(blocks / CUs) of them, each as long as its chunk is high.
// 8192 rows at 40 per task are 205 chunks x 34 strips = 1743 blocks = 6.81 per CU -- seven on most CUs; 210 chunks still make seven
// (1785 blocks = 6.97) and are 39 rows high.  So: the most chunks that keep ceil(blocks / CUs) where the requested height puts it, the
// rows dealt evenly over them (heights q and q + 1, never above the requested one).  Measured (profiles/r05_experiments.txt item 15):
// the pass steps up by 7-10 % where one block more than a multiple of the CU count appears (54 -> 55, 68 -> 69 rows per task), and
// 8190 rows = 210 x 39 run 1.4 % faster than as 205 x 40.  EPIC_HIP_FLAGS bit 2 (default on) switches it.
void tighten_chunks(Sweep2dArgs &a, int rows)
{
    a.chunk_rem = 0;
    if (!(a.flags & 4)) return;
    const int cus = current_device_cus();   // (per device: see there)
    if (cus <= 0 || a.nstrips <= 0) return;
    const long long blocks = ((long long)a.nchunks * a.nstrips + kWavesPerBlock - 1) / kWavesPerBlock;
    const long long per_cu = (blocks + cus - 1) / cus;
    long long n = per_cu * cus * kWavesPerBlock / a.nstrips;          // chunks that still fit
    n = std::min<long long>(n, rows / std::max(8, a.rows_per_task / 2));   // (never less than half the requested height, or 8 rows)
    if (n <= a.nchunks) return;
    a.nchunks = (int)n;
    a.rows_per_task = rows / (int)n;
    a.chunk_rem = rows % (int)n;
}

hipError_t launch_eval_math(const float *in, float *out, size_t n, int which, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(eval_math_kernel, dim3(4096), dim3(256), 0, stream, in, out, n, which);
    return hipGetLastError();
}

namespace {
// EPIC_HIP_FLAGS: bit 0 = alternate march direction (default 1).  Stores are always non-temporal (bit 1 used to switch
// that; measured on 8192^2 when both were switchable: traffic-only build 115.6 -> 96.6 us, red-black 104.2 -> 100.5 us,
// precise Jacobi 161.3 -> 155.8 us).
int sweep_flags() { return process_launch_knobs().flags; }   // (the environment is read in driver_config.cpp only)

// Workgroups of a launch whose workgroups walk `nblocks` logical blocks: what the chip keeps resident of `kernel`, in whole
// groups of the 8 XCDs (blocks are dealt round-robin over them; a stride that is a multiple of 8 keeps a workgroup's blocks
// in its XCD's band).  Unknown occupancy: one workgroup per block, as before.
int resident_grid(int nblocks, const void *kernel)
{
    const int res = resident_blocks_of(kernel) / kNumXcd * kNumXcd;
    return (res >= kNumXcd && nblocks > res) ? res : nblocks;
}

template <bool CHECK, bool RB, bool TRACK>
void launch_sweep_2d_math(int math, int nblocks, hipStream_t stream, const Sweep2dArgs &a)
{
    void (*kernel)(Sweep2dArgs) = math == kMathFast      ? sweep2d_kernel<CHECK, kMathFast, RB, TRACK>
                                  : math == kMathTraffic ? sweep2d_kernel<CHECK, kMathTraffic, RB, TRACK>
                                  : math == kMathTol     ? sweep2d_kernel<CHECK, kMathTol, RB, TRACK>
                                                         : sweep2d_kernel<CHECK, kMathPrecise, RB, TRACK>;
    // a list-driven launch is persistent waves: as many as the chip holds of this instantiation
    if (TRACK && a.wake.list_in) nblocks = sweep_2d_list_blocks((size_t)a.ntasks, resident_blocks_of((const void *)kernel));
    else if (math == kMathTol) nblocks = resident_grid(nblocks, (const void *)kernel);  // every workgroup stages the 16 KiB table once
    const dim3 grid(nblocks), block(kWave * kWavesPerBlock);
    hipLaunchKernelGGL(kernel, grid, block, 0, stream, a);
}
template <bool CHECK, bool RB>
void launch_sweep_2d_track(int math, int nblocks, hipStream_t stream, const Sweep2dArgs &a)
{
    if (a.wake.list_out) launch_sweep_2d_math<CHECK, RB, true>(math, nblocks, stream, a);
    else launch_sweep_2d_math<CHECK, RB, false>(math, nblocks, stream, a);
}
}  // namespace

// parity < 0: Jacobi sweep in -> out.  parity = 0 / 1: red-black half-sweep in place (in == out required).
hipError_t launch_sweep_2d(const float *in, float *out, const uint32_t *maskw, int rows, int pitch, int row_begin,
                           int row_end, int rows_per_task, int math, int parity, unsigned *delta_bits,
                           hipStream_t stream, const Activity *act, int check_begin, int check_end)
{
    if (row_end <= row_begin) return hipSuccess;
    if (pitch <= 0 || (pitch % 256) != 0 || rows <= 0 || row_begin < 0 || row_end > rows || rows_per_task <= 0)
        return hipErrorInvalidValue;
    if ((parity >= 0) != (in == out)) return hipErrorInvalidValue;
    if (math != kMathPrecise && math != kMathFast && math != kMathTraffic && math != kMathTol) return hipErrorInvalidValue;
    // the kernel addresses a task's rows with 32-bit byte offsets from a base 8 rows above it (rows_per_task + 14 rows)
    const long long max_rows = 0x7fffffffLL / ((long long)pitch * 4) - 16;
    if (max_rows < 1) return hipErrorInvalidValue;
    rows_per_task = (int)std::min<long long>(rows_per_task, max_rows);
    Sweep2dArgs a;
    a.in = in;
    a.out = out;
    a.maskw = maskw;
    a.maskf = nullptr;
    a.delta_bits = delta_bits;
    a.rows = rows;
    a.pitch = pitch;
    a.row_begin = row_begin;
    a.row_end = row_end;
    a.rows_per_task = rows_per_task;
    a.chunk_rem = 0;
    a.nstrips = (pitch + kStripCols - 1) / kStripCols;
    const int nchunks = (row_end - row_begin + rows_per_task - 1) / rows_per_task;
    a.ntasks = a.nstrips * nchunks;
    a.parity = parity < 0 ? 0 : (parity & 1);
    a.flags = sweep_flags();
    a.nchunks = nchunks;
    const bool whole = row_begin == 0 && row_end == rows;
    a.wake = wake_args(whole ? act : nullptr, (size_t)a.ntasks);
    int nblocks = (a.ntasks + kWavesPerBlock - 1) / kWavesPerBlock;
    a.nblocks = nblocks;
    a.check_lo = check_begin < 0 ? row_begin : check_begin;
    a.check_hi = check_begin < 0 ? row_end : check_end;
    if (parity < 0) {
        if (delta_bits) launch_sweep_2d_track<true, false>(math, nblocks, stream, a);
        else launch_sweep_2d_track<false, false>(math, nblocks, stream, a);
    } else {
        if (delta_bits) launch_sweep_2d_track<true, true>(math, nblocks, stream, a);
        else launch_sweep_2d_track<false, true>(math, nblocks, stream, a);
    }
    return hipGetLastError();
}

// Two consecutive red-black iterations (first one = `parity`) in one pass, in -> out (in != out).
// act (may be null): the work lists of THIS kernel's tiling (tiles = rb_fused_2d_tiles(rows, pitch, rows_per_task)); delta_bits (may
// be null): max |du| of the SECOND of the two iterations (zero it first).
namespace {
template <int MATH>
void launch_rb_fused_2d_math(const Sweep2dArgs &a, bool fmask, hipStream_t stream)
{
    const bool track = a.wake.list_out != nullptr, check = a.delta_bits != nullptr;
    void (*kernel)(Sweep2dArgs);
    if (track) kernel = check ? (fmask ? rb_fused2d_kernel<MATH, true, true, true> : rb_fused2d_kernel<MATH, false, true, true>)
                              : (fmask ? rb_fused2d_kernel<MATH, true, true, false> : rb_fused2d_kernel<MATH, false, true, false>);
    else kernel = check ? (fmask ? rb_fused2d_kernel<MATH, true, false, true> : rb_fused2d_kernel<MATH, false, false, true>)
                        : (fmask ? rb_fused2d_kernel<MATH, true, false, false> : rb_fused2d_kernel<MATH, false, false, false>);
    int nblocks = a.nblocks;
    if (track && a.wake.list_in) nblocks = sweep_2d_list_blocks((size_t)a.ntasks, resident_blocks_of((const void *)kernel));   // persistent waves
    hipLaunchKernelGGL(kernel, dim3(nblocks), dim3(kWave * kWavesPerBlock), 0, stream, a);
}
}  // namespace

hipError_t launch_rb_fused_2d(const float *in, float *out, const uint32_t *maskw, int rows, int pitch, int rows_per_task,
                              int math, int parity, hipStream_t stream, const uint32_t *maskf, const Activity *act, unsigned *delta_bits,
                              int check_begin, int check_end)
{
    if (pitch <= 0 || (pitch % 256) != 0 || rows <= 0 || rows_per_task <= 0 || in == out) return hipErrorInvalidValue;
    if (math != kMathPrecise && math != kMathFast && math != kMathTraffic) return hipErrorInvalidValue;  // (tol: in-place half-sweeps)
    if (math == kMathTraffic && (act || delta_bits)) return hipErrorInvalidValue;
    // the kernel addresses a task's rows with 32-bit byte offsets from a base 2 rows above it (rows_per_task + 5 rows)
    const long long max_rows = 0x7fffffffLL / ((long long)pitch * 4) - 8;
    if (max_rows < 1) return hipErrorInvalidValue;
    rows_per_task = (int)std::min<long long>(rows_per_task, max_rows);
    Sweep2dArgs a;
    a.in = in;
    a.out = out;
    a.maskw = maskw;
    a.maskf = maskf;
    a.check_lo = check_begin < 0 ? 0 : check_begin;
    a.check_hi = check_begin < 0 ? rows : check_end;
    a.delta_bits = delta_bits;
    a.rows = rows;
    a.pitch = pitch;
    a.row_begin = 0;
    a.row_end = rows;
    a.rows_per_task = rows_per_task;
    a.chunk_rem = 0;   // (one workgroup per block here, dealt as they finish: with the tol passes' chunk tightening the measured heights lie
                       //  within 156-163 us instead of 147-160, and whole relaxations take the same 2.306 s -- not used; item 15)
    a.nstrips = (pitch + kFusedOut - 1) / kFusedOut;
    a.nchunks = (rows + rows_per_task - 1) / rows_per_task;
    a.flags = sweep_flags();
    a.ntasks = a.nstrips * a.nchunks;
    a.parity = parity & 1;
    a.wake = wake_args(act, (size_t)a.ntasks);
    a.nblocks = (a.ntasks + kWavesPerBlock - 1) / kWavesPerBlock;
    if (math == kMathFast) launch_rb_fused_2d_math<kMathFast>(a, maskf != nullptr, stream);
    else if (math == kMathTraffic) {
        void (*kernel)(Sweep2dArgs) = maskf ? rb_fused2d_kernel<kMathTraffic, true, false, false> : rb_fused2d_kernel<kMathTraffic, false, false, false>;
        hipLaunchKernelGGL(kernel, dim3(a.nblocks), dim3(kWave * kWavesPerBlock), 0, stream, a);
    } else launch_rb_fused_2d_math<kMathPrecise>(a, maskf != nullptr, stream);
    return hipGetLastError();
}

// wake the tiles [t_lo, t_hi) for the launch that will consume `next` (its list_in / count_in / queued_in, given here as the
// *_out fields): a halo exchange has rewritten rows they hold or read
__global__ void wake_tile_range_kernel(WakeArgs w, int t_lo, int t_hi)
{
    const int t = t_lo + (int)(blockIdx.x * blockDim.x + threadIdx.x);
    wake_push(w, t, t < t_hi);
}

hipError_t launch_wake_tile_range(const Activity *next_as_out, size_t tiles, int t_lo, int t_hi, hipStream_t stream)
{
    if (!next_as_out || !next_as_out->list_out || t_hi <= t_lo) return hipSuccess;
    WakeArgs w = wake_args(next_as_out, tiles);
    w.list_in = nullptr;
    hipLaunchKernelGGL(wake_tile_range_kernel, dim3((unsigned)((t_hi - t_lo + 255) / 256)), dim3(256), 0, stream, w, t_lo, t_hi);
    return hipGetLastError();
}

hipError_t launch_fuse_masks_2d(const uint32_t *maskw, int rows, int pitch, uint32_t *maskf, hipStream_t stream)
{
    if (!maskw || !maskf || rows <= 0 || pitch <= 0 || (pitch % 256) != 0) return hipErrorInvalidValue;
    const size_t n = mask_words_fused_2d(rows, pitch) / 2;
    hipLaunchKernelGGL(fuse_masks_2d_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, maskw, rows, pitch, maskf);
    return hipGetLastError();
}

hipError_t launch_jacobi_fused_2d(const float *in, float *out, const uint32_t *maskw, int rows, int pitch, int rows_per_task,
                                  int math, hipStream_t stream, int parity, const uint32_t *maskf, const Activity *act, unsigned *delta_bits,
                                  int check_begin, int check_end)
{
    if (pitch <= 0 || (pitch % 256) != 0 || rows <= 0 || rows_per_task <= 0 || in == out) return hipErrorInvalidValue;
    if (math != kMathTol) return hipErrorInvalidValue;
    if ((act || delta_bits) && !maskf) return hipErrorInvalidValue;   // the tracked / checked instantiations take the fused mask layout
    // 32-bit byte offsets from a base 2 rows above the task (rows_per_task + 7 rows): as launch_sweep_2d clamps its tasks
    const long long max_rows = 0x7fffffffLL / ((long long)pitch * 4) - 8;
    if (max_rows < 1) return hipErrorInvalidValue;
    rows_per_task = (int)std::min<long long>(rows_per_task, max_rows);
    Sweep2dArgs a;
    a.in = in;
    a.out = out;
    a.maskw = maskw;
    a.maskf = nullptr;
    a.check_lo = check_begin < 0 ? 0 : check_begin;
    a.check_hi = check_begin < 0 ? rows : check_end;
    a.delta_bits = delta_bits;
    a.rows = rows;
    a.pitch = pitch;
    a.row_begin = 0;
    a.row_end = rows;
    a.rows_per_task = rows_per_task;
    a.chunk_rem = 0;
    a.nstrips = (pitch + kFusedOut - 1) / kFusedOut;
    a.nchunks = (rows + rows_per_task - 1) / rows_per_task;
    a.flags = sweep_flags();
    if (!act) tighten_chunks(a, rows);   // (work lists number uniform tiles: those launches keep the requested height)
    a.ntasks = a.nstrips * a.nchunks;
    a.parity = parity < 0 ? 0 : parity & 1;
    a.wake = wake_args(act, (size_t)a.ntasks);
    a.nblocks = (a.ntasks + kWavesPerBlock - 1) / kWavesPerBlock;
    a.maskf = maskf;
    const bool track = a.wake.list_out != nullptr, check = delta_bits != nullptr, rb = parity >= 0;
    void (*kernel)(Sweep2dArgs);
    if (!track && !check)
        kernel = !rb ? (maskf ? jacobi_fused2d_kernel<kMathTol, true> : jacobi_fused2d_kernel<kMathTol, false>)
                     : (maskf ? rb_tol_fused2d_kernel<true> : rb_tol_fused2d_kernel<false>);
    else if (rb)
        kernel = track ? (check ? tol_fused2d_tracked_kernel<true, true, true> : tol_fused2d_tracked_kernel<true, true, false>)
                       : tol_fused2d_tracked_kernel<true, false, true>;
    else
        kernel = track ? (check ? tol_fused2d_tracked_kernel<false, true, true> : tol_fused2d_tracked_kernel<false, true, false>)
                       : tol_fused2d_tracked_kernel<false, false, true>;
    // resident workgroups walk the logical blocks (every workgroup stages the table once); with lists: persistent waves
    const int nblocks = track && a.wake.list_in ? sweep_2d_list_blocks((size_t)a.ntasks, resident_blocks_of((const void *)kernel))
                                                : resident_grid(a.nblocks, (const void *)kernel);
    hipLaunchKernelGGL(kernel, dim3(nblocks), dim3(kWave * kWavesPerBlock), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_pack_mask_2d(const uint32_t *locked, int rows, int cols, int pitch, int ghost_top,
                               int ghost_bottom, uint32_t *maskw, hipStream_t stream)
{
    const int nstrips = pitch / kStripCols;
    dim3 grid(rows, (nstrips + kWavesPerBlock - 1) / kWavesPerBlock);
    hipLaunchKernelGGL(pack_mask_2d_kernel, grid, dim3(kWave * kWavesPerBlock), 0, stream, locked, rows, cols, pitch,
                       ghost_top, ghost_bottom, maskw);
    return hipGetLastError();
}

hipError_t launch_fill(float *p, size_t n, float v, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(fill_kernel, dim3(blocks), dim3(256), 0, stream, p, n, v);
    return hipGetLastError();
}

hipError_t launch_set_cells_2d(float *u, uint32_t *maskw, int rows, int cols, int pitch, unsigned k,
                               const unsigned *v, const unsigned *types, hipStream_t stream, int row0, int grid_rows,
                               int pin_top, int pin_bottom)
{
    if (k == 0) return hipSuccess;
    if (grid_rows <= 0) grid_rows = rows;
    hipLaunchKernelGGL(set_cells_2d_kernel, dim3((k + 255) / 256), dim3(256), 0, stream, u, maskw, rows, cols, pitch, k,
                       v, types, row0, grid_rows, pin_top, pin_bottom);
    return hipGetLastError();
}

}  // namespace epic_hip
