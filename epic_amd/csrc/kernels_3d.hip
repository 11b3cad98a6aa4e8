// kernels_3d.hip -- 3-D (n = 3) log-space Jacobi sweep for gfx950, 7-point stencil.
//
// The reference has no GPU code for n = 3 (the branches are empty: libepic/src/harmonic/harmonic_gpu.cu:160-162,
// :334-336, :367-369); the arithmetic follows its CPU sweep (libepic/src/harmonic/harmonic_cpu.cpp:81-133):
// neighbours in the order x0-1, x0+1, x1-1, x1+1, x2-1, x2+1, constant log(6.0).
//
// Layout: u[x0][x1][x2] with x2 contiguous and padded to `pitch` (multiple of 256 floats).  A wave owns 256 x2-columns
// (4 per lane, one dwordx4) of one x0-plane and marches along x1, keeping rows x1-1 / x1 / x1+1 of its plane in
// registers; x2 neighbours are full-wave DPP shifts; the rows of planes x0-1 and x0+1 are loaded per step.  The four
// waves of a workgroup sweep four consecutive planes of the same (x1-chunk, strip), so those extra rows are mostly the
// sibling waves' centre rows and are served by the CU's L1 / the XCD's L2 rather than HBM (six planes loaded for four:
// 1.41x the algorithmic reads measured, against the 1.5x of no reuse between workgroups at all).
// Mask: 1 bit per cell as lane masks (kernels.h), fetched with one scalar load per row.
// Optional activity tracking (wake.h): a tile is one task -- 32 x1-rows x 256 x2-columns of one x0-plane; it reads its
// own cells, the adjacent column / row of its four in-plane neighbours and the whole tile of the planes x0 - 1 and x0 + 1.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "cell_update.h"
#include "kernels.h"
#include "wake.h"

#ifndef EPIC_CONSTS_SWEEP3D   // build knob (A/B): where the precise routines' f64 constants live (cell_update.h: MathTab::consts)
#define EPIC_CONSTS_SWEEP3D(RB, TRACK) kConstsPlain   // (kConstsKeep for the red-black sweeps: +0.2 ... 1 % instructions; the 3-D loops keep their addends already)
#endif
namespace epic_hip {

namespace {

constexpr int kWave = 64;
constexpr int kStripCols = 256;
#ifndef EPIC_SWEEP3D_BLOCK_WAVES  // build knob (A/B).  On paper 8 planes per workgroup load 10 (1.25x) where 4 load 6 (1.5x); measured
#define EPIC_SWEEP3D_BLOCK_WAVES 4  // (PMC FETCH_SIZE, 512^3 tol, same box, profiles/r03_experiments.txt) 4 planes read 755 MB per sweep
#endif                            // and 8 planes 883 MB for 1 % of time (295.7 / 292.7 us; 16 planes: 302 us): 4 stays
constexpr int kWavesPerBlock = EPIC_SWEEP3D_BLOCK_WAVES;  // = consecutive planes per workgroup
static_assert(kWave * kWavesPerBlock >= kWakeLists, "one thread per work list resets the counters");
constexpr int kRowsPerTask = 32;

struct Sweep3dArgs {
    const float *in;
    float *out;
    const uint32_t *maskw;
    unsigned *delta_bits;
    int m0, m1, pitch;
    int plane_begin, plane_end;
    int nstrips, nchunks, nplane_groups;
    int nblocks;  // logical blocks: nstrips * (chunks * plane groups padded to a multiple of 8) (a tol launch holds fewer workgroups, which walk them)
    int check_lo, check_hi;  // CHECK: only planes [check_lo, check_hi) count for max |du| (a slab's ghost planes do not)
    int parity;  // red-black scheme only: currentIteration & 1
    int rows;    // sweep3d_pair_kernel: x1-rows per task
    WakeArgs wake;  // TRACK kernels, whole-grid launches only
};

// One x2-row of one plane with the tol math: c / up / dn = rows x1, x1 - 1, x1 + 1 of the plane, pa / pb = row x1 of the planes
// x0 - 1 and x0 + 1, s* their splits, hl / hr the two strip-edge cells of the row, m0..m3 its lane masks.  RB: only the cells of
// one colour are recomputed (even_cols: the lane's .x and .z).  The cells go through the three phases of cell_update.h two at
// a time: the table reads of one pair are in flight while the other pair is worked on.
// (hs: the split of the row's two strip-edge cells hl / hr -- .x the left one's, .y the right one's)
// In two halves, so that a wave that updates two rows can have the table reads of both in flight before it waits for any:
// tol_row_3d_front -- everything up to the ISSUE of the reads (both pairs of cells; red-black: the one pair of this colour);
// tol_row_3d_back<W0, W1> -- waits until at most W0 (W1) younger table lookups (cells) are outstanding for the first (second) pair, and
// finishes the cells.
struct TolRowFront { TolPre2 pxz, pyw; TolLnPair f0, f1; };
template <bool RB>
__device__ __forceinline__ TolRowFront tol_row_3d_front(const float4 &pa, const float4 &pb, const float4 &up, const float4 &c, const float4 &dn,
                                                        const Split4 &sa, const Split4 &sb, const Split4 &su, const Split4 &sc, const Split4 &sd,
                                                        float hl, float hr, const Split2 &hs, bool even_cols, const TolLnEntry *tl)
{
    const float lf = wave_from_left(c.w, hl);
    const float rt = wave_from_right(c.x, hr);
    const float ql = wave_from_left(sc.qw, hs.q.x), qr = wave_from_right(sc.qx, hs.q.y);
    const uint32_t nl = f2u(wave_from_left(u2f(sc.nw), hs.zm.x)), nr = f2u(wave_from_right(u2f(sc.nx), hs.zm.y));
    auto pre_xz = [&] {
        return tol_pre2_3d(TolNb6{pa.x, pb.x, up.x, dn.x, lf, c.y, sa.qx, sb.qx, su.qx, sd.qx, ql, sc.qy, sa.nx, sb.nx, su.nx, sd.nx, nl, sc.ny},
                           TolNb6{pa.z, pb.z, up.z, dn.z, c.y, c.w, sa.qz, sb.qz, su.qz, sd.qz, sc.qy, sc.qw, sa.nz, sb.nz, su.nz, sd.nz, sc.ny, sc.nw});
    };
    auto pre_yw = [&] {
        return tol_pre2_3d(TolNb6{pa.y, pb.y, up.y, dn.y, c.x, c.z, sa.qy, sb.qy, su.qy, sd.qy, sc.qx, sc.qz, sa.ny, sb.ny, su.ny, sd.ny, sc.nx, sc.nz},
                           TolNb6{pa.w, pb.w, up.w, dn.w, c.z, rt, sa.qw, sb.qw, su.qw, sd.qw, sc.qz, qr, sa.nw, sb.nw, su.nw, sd.nw, sc.nz, nr});
    };
    TolRowFront fr;
    if (!RB) {
        fr.pxz = pre_xz();
        fr.f0 = tol_ln_issue<5>(fr.pxz, tl);
        fr.pyw = pre_yw();
        fr.f1 = tol_ln_issue<5>(fr.pyw, tl);
    } else {   // red-black: the one pair of this colour, kept in pxz / f0 whichever it is
        fr.pxz = even_cols ? pre_xz() : pre_yw();
        fr.f0 = tol_ln_issue<5>(fr.pxz, tl);
    }
    return fr;
}
template <bool RB, int W0, int W1>
__device__ __forceinline__ float4 tol_row_3d_back(TolRowFront &fr, const float4 &c, lmask m0, lmask m1, lmask m2, lmask m3, bool even_cols)
{
    float4 o = c;
    float nx, ny, nz, nw;
    TolLnRaw ea, eb;
    if (!RB) {
        tol_ln_wait<W0>(fr.f0, ea, eb);
        tol_post2(fr.pxz, ea, eb, kLn6, nx, nz);
        o.x = sel(m0, c.x, nx);
        o.z = sel(m2, c.z, nz);
        tol_ln_wait<W1>(fr.f1, ea, eb);
        tol_post2(fr.pyw, ea, eb, kLn6, ny, nw);
        o.y = sel(m1, c.y, ny);
        o.w = sel(m3, c.w, nw);
    } else {   // (one pair of reads per row: W1 counts the younger ones)
        tol_ln_wait<W1>(fr.f0, ea, eb);
        tol_post2(fr.pxz, ea, eb, kLn6, nx, nz);
        if (even_cols) {
            o.x = sel(m0, c.x, nx);
            o.z = sel(m2, c.z, nz);
        } else {
            o.y = sel(m1, c.y, nx);
            o.w = sel(m3, c.w, nz);
        }
    }
    (void)ny; (void)nw;
    return o;
}
template <bool RB>
__device__ __forceinline__ float4 tol_row_3d(const float4 &pa, const float4 &pb, const float4 &up, const float4 &c, const float4 &dn,
                                             const Split4 &sa, const Split4 &sb, const Split4 &su, const Split4 &sc, const Split4 &sd,
                                             float hl, float hr, const Split2 &hs, lmask m0, lmask m1, lmask m2, lmask m3, bool even_cols,
                                             const TolLnEntry *tl)
{
    TolRowFront fr = tol_row_3d_front<RB>(pa, pb, up, c, dn, sa, sb, su, sc, sd, hl, hr, hs, even_cols, tl);
    return tol_row_3d_back<RB, 2, 0>(fr, c, m0, m1, m2, m3, even_cols);
}

template <bool RB>
__device__ __forceinline__ float4 tol_row_3d(const float4 &pa, const float4 &pb, const float4 &up, const float4 &c, const float4 &dn,
                                             const Split4 &sa, const Split4 &sb, const Split4 &su, const Split4 &sc, const Split4 &sd,
                                             float hl, float hr, lmask m0, lmask m1, lmask m2, lmask m3, bool even_cols,
                                             const TolLnEntry *tl)
{
    const Split2 hs = tol_split2(v2f{hl, hr});  // the two strip-edge cells of the row
    return tol_row_3d<RB>(pa, pb, up, c, dn, sa, sb, su, sc, sd, hl, hr, hs, m0, m1, m2, m3, even_cols, tl);
}

// RB = true: the reference's 3-D red-black half-sweep in place (in == out): cells with (x0 + x1 + x2 + currentIteration)
// even are recomputed (harmonic_cpu.cpp:89-102), all six neighbours have the other colour.
// At least 5 waves per SIMD: left alone, the list-driven precise variant takes 146 VGPRs (3 waves per SIMD); held to
// 5 waves it relaxes the 512^3 benchmark in 1.285 s instead of 1.33 (7 waves: 1.31; no spills either way).
// (With the selects and lane shifts going through compiler builtins the list-driven variants need a few registers more:
// held to 5 waves the tracked precise Jacobi kernel spilled 228 registers, so the tracked variants are held to 4 and
// that one -- whose allocation the compiler blows up to 146 registers whatever it is told -- to 3.
// No sweep kernel may use scratch: tools/isa_hazards.py checks the generated ISA.)
#ifndef EPIC_SWEEP3D_WAVES
#define EPIC_SWEEP3D_WAVES 5
#endif
template <bool CHECK, int MATH, bool RB, bool TRACK> struct Sweep3dOcc {
    static constexpr int kMinWaves = MATH == kMathTol ? (TRACK ? EPIC_SWEEP3D_WAVES - 2 : EPIC_SWEEP3D_WAVES - 1)  // tol: 8 more registers per row in the window
                                     : !TRACK ? EPIC_SWEEP3D_WAVES
                                     : (!CHECK && !RB) ? EPIC_SWEEP3D_WAVES - 2 : EPIC_SWEEP3D_WAVES - 1;
};
template <bool CHECK, int MATH, bool RB, bool TRACK>
__global__ __launch_bounds__(kWave * kWavesPerBlock, (Sweep3dOcc<CHECK, MATH, RB, TRACK>::kMinWaves)) void sweep3d_kernel(Sweep3dArgs a)
{
    constexpr bool TOL = MATH == kMathTol;  // one split per cell, shared by the six cells it is a neighbour of (cell_update.h)
    // precise math: glibc's expf / logf tables (every wave writes them itself); tol math: the table of its own logarithm,
    // five binades, copied by the whole workgroup before any wave may leave (one barrier)
    __shared__ __attribute__((aligned(16))) char math_lds_bytes[TOL ? TolLn<5>::kLdsBytes : kMathLdsDoubles * (int)sizeof(double)];
    const TolLnEntry *const tl = reinterpret_cast<const TolLnEntry *>(math_lds_bytes);
    if (TOL) TolLn<5>::stage(reinterpret_cast<TolLnEntry *>(math_lds_bytes));
    MathTab lds = {};
    if (MATH == kMathPrecise) lds = math_tables_load(reinterpret_cast<double *>(math_lds_bytes), EPIC_CONSTS_SWEEP3D(RB, TRACK));
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (TRACK) wake_reset_next(a.wake);
    const bool listed = TRACK && a.wake.list_in != nullptr;
    WakeCursor cursor = {};
    if (listed && !wake_begin(a.wake, lane, wave, kWavesPerBlock, cursor)) return;
    float dmax = 0.0f;

    int vb = blockIdx.x;  // logical block of this pass
    for (;;) {  // one pass per task: exactly one unless the launch is list-driven or holds fewer workgroups than blocks
    int strip, chunk, x0;
    if (listed) {
        int t = wake_tile(a.wake, cursor);  // tile id = (x0 * nchunks + chunk) * nstrips + strip
        strip = t % a.nstrips;
        t /= a.nstrips;
        chunk = t % a.nchunks;
        x0 = t / a.nchunks;
    } else {
        // XCD-aware order (workgroup vb runs on XCD vb % 8): the strips of a row share one L2, the (plane group, chunk) pairs are
        // dealt over the XCDs in turn (see sweep3d_pair_kernel)
        const int j = vb >> 3;
        strip = j % a.nstrips;
        const int pair = (j / a.nstrips) * 8 + (vb & 7);
        chunk = pair % a.nchunks;
        x0 = a.plane_begin + (pair / a.nchunks) * kWavesPerBlock + wave;
        if (pair >= a.nchunks * a.nplane_groups) x0 = a.plane_end;  // padding of the pair count to a multiple of 8: a spare block
    }
    if (x0 >= a.plane_end) {  // wave-uniform: the spare waves of the last plane group
        if (listed) break;
        vb += gridDim.x;
        if (vb >= a.nblocks) break;
        continue;
    }
    const int tile = (x0 * a.nchunks + chunk) * a.nstrips + strip;
    if (TRACK && lane == 0) a.wake.queued_in[tile] = 0;
    const int r0 = chunk * kRowsPerTask;
    const int r1 = min(r0 + kRowsPerTask, a.m1);

    const int col0 = strip * kStripCols;  // pitch % 256 == 0: every lane is in bounds
    const size_t pitch = (size_t)a.pitch;
    const size_t plane = (size_t)a.m1 * pitch;
    const int rlast = a.m1 - 1;

    // As in the 2-D kernel, nothing but the arithmetic of the update is left to the VALU: rows are reached through
    // buffer descriptors (one per plane x0 - 1 / x0 / x0 + 1 and one for the output, based a few rows above the task) with
    // the lane part of the address in a VGPR that never changes and the row / strip part in an SGPR; the strip-edge
    // values of the centre row and the four lane masks of the row (kernels.h) come through scalar loads.
    const int rlo = max(r0 - 4, 0);
    // ONE descriptor for the three planes a wave reads, based at the lowest of them: the plane rides on the scalar offset (the
    // launcher keeps three planes within the 2 GiB a descriptor spans) -- eight SGPRs less than a descriptor per plane, in a
    // kernel that spills scalars
    const int xa = max(x0 - 1, 0), xb = min(x0 + 1, a.m0 - 1);
    const float *pc = a.in + (size_t)x0 * plane;
    const __amdgpu_buffer_rsrc_t rin = raw_buffer(a.in + (size_t)xa * plane + (size_t)rlo * pitch),
                                 rout = raw_buffer(a.out + (size_t)x0 * plane + (size_t)rlo * pitch);
    const unsigned rc = (unsigned)((size_t)(x0 - xa) * plane * sizeof(float)), ra = 0u, rb = (unsigned)((size_t)(xb - xa) * plane * sizeof(float));
    const unsigned lane16 = (unsigned)lane * 16u;
    typedef unsigned vu4 __attribute__((ext_vector_type(4)));
    auto row_off = [&](int r) -> unsigned { return (unsigned)((r - rlo) * a.pitch + col0) * 4u; };  // r already clamped
    auto ld = [&](unsigned plane_off, int r) -> float4 {
        r = min(max(r, 0), rlast);
        const vu4 q = __builtin_amdgcn_raw_buffer_load_b128(rin, lane16, plane_off + row_off(r), 0);
        return make_float4(u2f(q.x), u2f(q.y), u2f(q.z), u2f(q.w));
    };
    typedef const __attribute__((address_space(4))) float cfloat;
    typedef const __attribute__((address_space(4))) uint64_t cu64;
    const int hcol_l = max(col0 - 1, 0), hcol_r = min(col0 + kStripCols, a.pitch - 1);
    struct RowSide { float l, r; lmask m0, m1, m2, m3; };
    auto side = [&](int r) -> RowSide {
        r = min(max(r, 0), rlast);
        cfloat *row = (cfloat *)(pc + (size_t)r * pitch);
        cu64 *mk = (cu64 *)a.maskw + (((size_t)x0 * a.m1 + r) * a.nstrips + strip) * 4;
        return RowSide{row[hcol_l], row[hcol_r], mk[0], mk[1], mk[2], mk[3]};
    };

    lmask chg_any = 0, chg_x = 0, chg_w = 0, chg_top = 0, chg_bot = 0;  // lane masks, as in the 2-D kernel

    // One row: up / c / dn = rows x1 - 1, x1, x1 + 1 of the plane, pa / pb = row x1 of the planes x0 - 1 and x0 + 1.
    // (tol math: su / sc / sd are the splits of the plane's three rows, made once per row as it enters the window; the
    // rows of the two neighbouring planes are split where they are used)
    auto row_step = [&](int r, const float4 &up, const float4 &c, const float4 &dn, const float4 &pa, const float4 &pb,
                        const RowSide &h, const Split4 &su, const Split4 &sc, const Split4 &sd) {
        float4 o;
        const float lf = TOL ? 0.0f : wave_from_left(c.w, h.l);
        const float rt = TOL ? 0.0f : wave_from_right(c.x, h.r);
        if (TOL) {
            const Split4 sa = tol_split4(pa), sb = tol_split4(pb);
            const bool even_cols = !RB || ((x0 + r + a.parity) & 1) == 0;  // scalar
            o = tol_row_3d<RB>(pa, pb, up, c, dn, sa, sb, su, sc, sd, h.l, h.r, h.m0, h.m1, h.m2, h.m3, even_cols, tl);
        } else if (RB) {
            o = c;
            if (((x0 + r + a.parity) & 1) == 0) {  // scalar: even x2 columns (.x, .z) are this row's active cells
                const float nx = cell_update_3d<MATH>(pa.x, pb.x, up.x, dn.x, lf, c.y, lds);
                const float nz = cell_update_3d<MATH>(pa.z, pb.z, up.z, dn.z, c.y, c.w, lds);
                o.x = sel(h.m0, c.x, nx);
                o.z = sel(h.m2, c.z, nz);
            } else {
                const float ny = cell_update_3d<MATH>(pa.y, pb.y, up.y, dn.y, c.x, c.z, lds);
                const float nw = cell_update_3d<MATH>(pa.w, pb.w, up.w, dn.w, c.z, rt, lds);
                o.y = sel(h.m1, c.y, ny);
                o.w = sel(h.m3, c.w, nw);
            }
        } else {
            o.x = sel(h.m0, c.x, cell_update_3d<MATH>(pa.x, pb.x, up.x, dn.x, lf, c.y, lds));
            o.y = sel(h.m1, c.y, cell_update_3d<MATH>(pa.y, pb.y, up.y, dn.y, c.x, c.z, lds));
            o.z = sel(h.m2, c.z, cell_update_3d<MATH>(pa.z, pb.z, up.z, dn.z, c.y, c.w, lds));
            o.w = sel(h.m3, c.w, cell_update_3d<MATH>(pa.w, pb.w, up.w, dn.w, c.z, rt, lds));
        }
        if (CHECK && x0 >= a.check_lo && x0 < a.check_hi) {  // scalar condition
            dmax = max2(dmax, fabsf(c.x - o.x));
            dmax = max2(dmax, fabsf(c.y - o.y));
            dmax = max2(dmax, fabsf(c.z - o.z));
            dmax = max2(dmax, fabsf(c.w - o.w));
        }
        if (TRACK) {
            const lmask cx = lanes_ne(o.x, c.x), cw = lanes_ne(o.w, c.w);
            const lmask rc2 = cx | cw | lanes_ne(o.y, c.y) | lanes_ne(o.z, c.z);
            chg_any |= rc2;
            chg_x |= cx;
            chg_w |= cw;
            if (r == r0) chg_top = rc2;
            if (r == r1 - 1) chg_bot = rc2;
        }
        store_row(rout, o.x, o.y, o.z, o.w, lane16, row_off(r));  // non-temporal: 389.7 -> 384.4 us per 512^3 sweep
    };

    // Register rings rotated by hand (no moves between a load and its use): the plane's own rows and the rows of the
    // neighbouring planes run two rows ahead over four register sets, the scalar row sides one row ahead over two.
    const int nrows = r1 - r0, nfull = nrows & ~3;
    auto split = [&](const float4 &q) { return TOL ? tol_split4(q) : Split4{}; };
    if (nfull > 0) {
        float4 q0 = ld(rc, r0 - 1), q1 = ld(rc, r0), q2 = ld(rc, r0 + 1), q3;
        RowSide sa = side(r0), sb;
        Split4 s0 = split(q0), s1 = split(q1), s2 = {}, s3 = {};
        // (the neighbouring planes' rows run two rows ahead as well, over four register sets each: 300.8 -> 295.7 us at 512^3 against
        //  one row ahead, same box, 747 -> 755 MB read -- the kernel waits for memory almost half of its time,
        //  profiles/r03_sq_counters_3d_tol.txt)
        float4 pa4[4], pb4[4];
        pa4[0] = ld(ra, r0); pb4[0] = ld(rb, r0); pa4[1] = ld(ra, r0 + 1); pb4[1] = ld(rb, r0 + 1);
        for (int i = 0; i < nfull; i += 4) {
            const int r = r0 + i;
            q3 = ld(rc, r + 2); pa4[2] = ld(ra, r + 2); pb4[2] = ld(rb, r + 2); sb = side(r + 1);
            s2 = split(q2);
            row_step(r, q0, q1, q2, pa4[0], pb4[0], sa, s0, s1, s2);
            q0 = ld(rc, r + 3); pa4[3] = ld(ra, r + 3); pb4[3] = ld(rb, r + 3); sa = side(r + 2);
            s3 = split(q3);
            row_step(r + 1, q1, q2, q3, pa4[1], pb4[1], sb, s1, s2, s3);
            q1 = ld(rc, r + 4); pa4[0] = ld(ra, r + 4); pb4[0] = ld(rb, r + 4); sb = side(r + 3);
            s0 = split(q0);
            row_step(r + 2, q2, q3, q0, pa4[2], pb4[2], sa, s2, s3, s0);
            q2 = ld(rc, r + 5); pa4[1] = ld(ra, r + 5); pb4[1] = ld(rb, r + 5); sa = side(r + 4);
            s1 = split(q1);
            row_step(r + 3, q3, q0, q1, pa4[3], pb4[3], sb, s3, s0, s1);
        }
    }
    for (int r = r0 + nfull; r < r1; ++r) {  // ragged tail
        const float4 ru = ld(rc, r - 1), rm = ld(rc, r), rd = ld(rc, r + 1);
        row_step(r, ru, rm, rd, ld(ra, r), ld(rb, r), side(r), split(ru), split(rm), split(rd));
    }

    if (TRACK) {
        // wake the tiles that read what this task changed: itself, the in-plane neighbours across the edges that
        // changed, and the same tile of the two neighbouring planes (every cell has an x0 - 1 and an x0 + 1 neighbour)
        const bool any = chg_any != 0;
        const bool first_col = (chg_x & 1ull) != 0, last_col = (chg_w >> 63) != 0;
        const bool first_row = chg_top != 0, last_row = chg_bot != 0;
        const int per_plane = a.nchunks * a.nstrips;
        int t = tile;
        bool want = any;
        if (lane == 1) { t = tile - 1; want = strip > 0 && first_col; }
        if (lane == 2) { t = tile + 1; want = strip + 1 < a.nstrips && last_col; }
        if (lane == 3) { t = tile - a.nstrips; want = chunk > 0 && first_row; }
        if (lane == 4) { t = tile + a.nstrips; want = chunk + 1 < a.nchunks && last_row; }
        if (lane == 5) { t = tile - per_plane; want = x0 > 0 && any; }
        if (lane == 6) { t = tile + per_plane; want = x0 + 1 < a.m0 && any; }
        wake_push(a.wake, t, want && lane < 7);
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

// ---------------------------------------------------------------------------------------------------------------------
// Two planes per wave (tol math, launches without work lists).  In sweep3d_kernel every row is loaded and split by three
// waves: its own and those of the planes on either side.  Timing-only builds of that kernel priced the two halves of this
// (512^3, same box, profiles/r03_experiments.txt item 8): without the neighbouring planes' LOADS 292 -> 235 us, without
// their SPLITS 292 -> 247 us -- the sweep is bound by both at once.  Here a wave owns the planes x0 and x0 + 1 of its strip
// and marches both along x1: each is the other's neighbour, so of the four rows a step needs from outside the wave's own
// window two are already in registers with their splits.  Per updated row: 2 rows loaded and split instead of 3 (51.6 VALU
// instructions per cell against 56.8).  177-189 registers, two waves per SIMD (three -- 168 registers and a few spills -- time
// the same); the four waves of a workgroup hold eight consecutive planes.  512^3, same box: 295 -> 270-273 us per sweep.
// The loads and stores of this march alone (a traffic-only build of round 3) take 217-225 us whatever the occupancy and the
// prefetch depth: the floor of the pattern.  Build knobs below: measured alternatives (profiles/r03_experiments.txt item 8).
// Same arithmetic on the same inputs as sweep3d_kernel: bit-identical results (tests/test_gpu_full_configs.py and the
// whole 3-D parity suite run through it; EPIC_HIP_3D_PAIR=0 selects the one-plane kernel).
#ifndef EPIC_PAIR_MIN_BLOCKS
#define EPIC_PAIR_MIN_BLOCKS 2
#endif
#ifndef EPIC_PAIR_OUTER_AHEAD
#define EPIC_PAIR_OUTER_AHEAD 2
#endif
#ifndef EPIC_PAIR_BARRIER_ROWS
#define EPIC_PAIR_BARRIER_ROWS 0
#endif
#ifndef EPIC_PAIR_BLOCK_WAVES
#define EPIC_PAIR_BLOCK_WAVES 4
#endif
#ifndef EPIC_PAIR_SEQ
#define EPIC_PAIR_SEQ 0
#endif
// EPIC_PAIR_LDS_OUTER = 1 (build knob, A/B): the rows of the two OUTER planes (x0 - 1 of plane A, x0 + 1 of plane B) do not wait
// in registers while they are in flight: they are loaded two steps ahead by LDS-DMA (buffer_load_dwordx4 ... lds: 1 KiB per
// wave-instruction, no destination registers) into a ring of four 2 KiB slots per wave and read back with two ds_read_b128 in
// the step that splits them -- 16 VGPRs less for the rings.  One __shared__ array PER SLOT: the compiler orders a read behind
// a pending LDS-DMA by a counted vmcnt only where its alias analysis can tell the slots apart (one array for all slots: vmcnt(0)
// before every read, i.e. no prefetch at all); a wave's own reads need nothing but that wait (MI355X_MICROARCH.md item 7).
#ifndef EPIC_PAIR_LDS_OUTER
#define EPIC_PAIR_LDS_OUTER 0
#endif
// One row of a wave (64 lanes x 16 bytes) from memory straight into LDS at `lds_row` (wave-uniform; lane i lands at + 16 i).
// A __device__ function, not a lambda of the kernel: the host pass of hipcc drops a kernel template whose (host-and-device)
// lambdas name this device-only builtin -- silently, the kernel's handle is then an undefined symbol of the object file.
template <class T>
__device__ __forceinline__ void lds_dma_row(const __amdgpu_buffer_rsrc_t &rsrc, T *lds_row, unsigned lane_off, unsigned row_off)
{
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void *)lds_row, 16, lane_off, row_off, 0, 0);
}
constexpr int kPairWaves = EPIC_PAIR_BLOCK_WAVES;  // waves per workgroup = pairs of consecutive planes
constexpr int kPairMinBlocks = EPIC_PAIR_MIN_BLOCKS;
constexpr int kPairRows = 64;  // x1-rows per task (512^3: 32 rows 273.5 us, 64: 269.7, 128: 268.5, 256: 323 -- too few tasks)
constexpr int kPairOuterAhead = EPIC_PAIR_OUTER_AHEAD;  // rows the outer planes' loads run ahead of their use (1 or 2)
// X0M = false: the pair is two consecutive PLANES (x0, x0 + 1) and the wave marches along x1 (rows of a task's chunk);
// X0M = true: the pair is two consecutive ROWS (x1, x1 + 1) of a strip and the wave marches along x0, plane by plane -- all
// workgroups then move through memory together, plane after plane.  "c" below is the pair axis, "t" the march axis.
template <bool CHECK, bool RB, bool X0M>
__global__ __launch_bounds__(kWave * kPairWaves, kPairMinBlocks) void sweep3d_pair_kernel(Sweep3dArgs a)
{
    __shared__ __attribute__((aligned(16))) char math_lds_bytes[TolLn<5>::kLdsBytes];
    const TolLnEntry *const tl = reinterpret_cast<const TolLnEntry *>(math_lds_bytes);
    TolLn<5>::stage(reinterpret_cast<TolLnEntry *>(math_lds_bytes));
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float dmax = 0.0f;
#if EPIC_PAIR_LDS_OUTER
    // slot j of the ring: [wave][outer row a / b][lane]
    typedef unsigned vu4s __attribute__((ext_vector_type(4)));
    __shared__ vu4s outer_slot0[kPairWaves][2][kWave], outer_slot1[kPairWaves][2][kWave], outer_slot2[kPairWaves][2][kWave],
        outer_slot3[kPairWaves][2][kWave];
#endif
    typedef unsigned vu4 __attribute__((ext_vector_type(4)));
    typedef const __attribute__((address_space(4))) uint64_t cu64;
    struct RowSide { lmask m0, m1, m2, m3; };
    const size_t pitch = (size_t)a.pitch;
    const size_t plane = (size_t)a.m1 * pitch;
    const unsigned plane_bytes = (unsigned)(plane * sizeof(float)), row_bytes = (unsigned)(pitch * sizeof(float));
    const unsigned cstride = X0M ? row_bytes : plane_bytes, tstride = X0M ? plane_bytes : row_bytes;  // bytes per step of either axis
    const int c_begin = X0M ? 0 : a.plane_begin, c_end = X0M ? a.m1 : a.plane_end, c_max = (X0M ? a.m1 : a.m0) - 1;
    const int t_begin = X0M ? a.plane_begin : 0, t_end = X0M ? a.plane_end : a.m1, t_max = (X0M ? a.m0 : a.m1) - 1;

    for (int vb = blockIdx.x; vb < a.nblocks; vb += gridDim.x) {
        // XCD-aware order: workgroup vb runs on XCD vb % 8 (round-robin dispatch; the grid is a multiple of 8), and the strips of a
        // row go to ONE XCD's L2 (each fetches its neighbour's edge cell: with the strips on different XCDs that was a 128-byte
        // line from HBM per 4 bytes used, 12 % of the sweep's reads).  The (plane group, chunk) pairs are dealt over the XCDs in
        // turn -- even for any chunk count; with a multiple of 8 chunks the chunk alone selects the XCD and the neighbouring
        // plane groups share it too.
        const int j = vb >> 3;
        const int strip = j % a.nstrips;
        const int pair = (j / a.nstrips) * 8 + (vb & 7);
        if (pair >= a.nchunks * a.nplane_groups) continue;   // wave-uniform (padding of the pair count to a multiple of 8)
        const int chunk = pair % a.nchunks;
        const int cA_raw = c_begin + ((pair / a.nchunks) * kPairWaves + wave) * 2;
#if EPIC_PAIR_BARRIER_ROWS == 0
        if (cA_raw >= c_end) continue;                   // wave-uniform: the spare waves of the last group
#endif
        const bool has_a = cA_raw < c_end;               // (with barriers the spare waves march along and store nothing)
        const int cA = has_a ? cA_raw : c_end - 1;
        const bool has_b = cA_raw + 1 < c_end;           // an odd count: the last wave's second plane / row is swept and dropped
        const int cB = min(cA + 1, c_max);
        const int t0 = t_begin + chunk * a.rows;
        const int t1 = min(t0 + a.rows, t_end);
        const int col0 = strip * kStripCols;
        const int tlo = max(t0 - 4, 0);
        // one descriptor for everything the wave reads, based at its lowest address (the launcher keeps that within 2 GiB)
        const int ca = max(cA - 1, 0), cb = min(cA + 2, c_max);
        const __amdgpu_buffer_rsrc_t rin = raw_buffer(a.in + ((size_t)ca * cstride + (size_t)tlo * tstride) / sizeof(float)),
                                     rout = raw_buffer(a.out + ((size_t)cA * cstride + (size_t)tlo * tstride) / sizeof(float));
        const unsigned oa = 0u, ocA = (unsigned)(cA - ca) * cstride, ocB = (unsigned)(cB - ca) * cstride, ob = (unsigned)(cb - ca) * cstride;
        const unsigned lane16 = (unsigned)lane * 16u;
        const unsigned laneA = has_a ? lane16 : 0x80000000u;  // beyond the descriptor's range: the hardware drops the store
        const unsigned laneB = has_b ? lane16 : 0x80000000u;
        auto t_off = [&](int t) -> unsigned { return (unsigned)(t - tlo) * tstride + (unsigned)col0 * 4u; };  // t already clamped
        auto ld = [&](unsigned c_off, int t) -> float4 {
            t = min(max(t, 0), t_max);
            const vu4 q = __builtin_amdgcn_raw_buffer_load_b128(rin, lane16, c_off + t_off(t), 0);
            return make_float4(u2f(q.x), u2f(q.y), u2f(q.z), u2f(q.w));
        };
        // The two strip-edge cells of a row (they belong to the neighbouring strips) come through the VECTOR path like the rows
        // themselves: one dword load in which lane 0 fetches the cell left of the strip and lane 63 the one right of it -- exactly
        // the lanes whose wave shifts take them as the edge operand.  (As scalar loads -- two per row and plane, as in
        // sweep3d_kernel -- they cost 6 % of the sweep, measured: every wait for a table read, lgkmcnt-counted like them, also
        // waited for the scalar loads just issued for the next step.  profiles/r03_experiments.txt item 8.)
        const int hcol_l = max(col0 - 1, 0), hcol_r = min(col0 + kStripCols, a.pitch - 1);
        const unsigned lane_e = lane == 0 ? (unsigned)hcol_l * 4u : lane == 63 ? (unsigned)hcol_r * 4u : (unsigned)col0 * 4u + lane16;
        auto eld = [&](unsigned c_off, int t) -> float {
            t = min(max(t, 0), t_max);
            return u2f(__builtin_amdgcn_raw_buffer_load_b32(rin, lane_e, c_off + (unsigned)(t - tlo) * tstride, 0));
        };
        auto side = [&](int c, int t) -> RowSide {
            t = min(max(t, 0), t_max);
            const int x0 = X0M ? t : c, x1 = X0M ? c : t;
            cu64 *mk = (cu64 *)a.maskw + (((size_t)x0 * a.m1 + x1) * a.nstrips + strip) * 4;
            return RowSide{mk[0], mk[1], mk[2], mk[3]};
        };
        // planes [check_lo, check_hi) count for max |du|: a property of the pair (x1-march) or of the step (x0-march)
        const bool chkA = CHECK && has_a && (X0M || (cA >= a.check_lo && cA < a.check_hi)),
                   chkB = CHECK && has_b && (X0M || (cB >= a.check_lo && cB < a.check_hi));
        auto fold = [&](const float4 &c, const float4 &o) {
            dmax = max2(dmax, fabsf(c.x - o.x));
            dmax = max2(dmax, fabsf(c.y - o.y));
            dmax = max2(dmax, fabsf(c.z - o.z));
            dmax = max2(dmax, fabsf(c.w - o.w));
        };

#if EPIC_PAIR_LDS_OUTER
        auto slot = [&](const int j) -> vu4s(*)[kWave] {   // j: a constant in every expansion
            return j == 0 ? outer_slot0[wave] : j == 1 ? outer_slot1[wave] : j == 2 ? outer_slot2[wave] : outer_slot3[wave];
        };
        // the outer rows of step t into slot j (the same addresses as ld(): t clamped like there)
        auto ld_outer = [&](const int j, int t) {
            t = min(max(t, 0), t_max);
            vu4s(*s)[kWave] = slot(j);
            lds_dma_row(rin, &s[0][0], lane16, oa + t_off(t));
            lds_dma_row(rin, &s[1][0], lane16, ob + t_off(t));
        };
#endif
        // rings, rotated through constant indices of fully unrolled steps (no moves): the pair's rows 2 ahead over four sets
        // (steps j - 1, j, j + 1 in use, j + 2 in flight), their splits likewise, the outer rows 2 ahead
        float4 qA[4], qB[4], pa[4], pb[4];
        float eA[4], eB[4];   // strip-edge cells of the owned rows (lane 0: left, lane 63: right)
        Split4 sA[4], sB[4];
        RowSide hA[2], hB[2];
        // slot of step t0 + j: j & 3 (sides: j & 1); step t0 - 1 sits in slot 3
        qA[3] = ld(ocA, t0 - 1); qB[3] = ld(ocB, t0 - 1);
        qA[0] = ld(ocA, t0); qB[0] = ld(ocB, t0);
        eA[0] = eld(ocA, t0); eB[0] = eld(ocB, t0);
        qA[1] = ld(ocA, t0 + 1); qB[1] = ld(ocB, t0 + 1);
        eA[1] = eld(ocA, t0 + 1); eB[1] = eld(ocB, t0 + 1);
#if EPIC_PAIR_LDS_OUTER
        ld_outer(0, t0);
        if (kPairOuterAhead > 1) ld_outer(1, t0 + 1);
#else
        pa[0] = ld(oa, t0); pb[0] = ld(ob, t0);
        if (kPairOuterAhead > 1) { pa[1] = ld(oa, t0 + 1); pb[1] = ld(ob, t0 + 1); }
#endif
        hA[0] = side(cA, t0); hB[0] = side(cB, t0);
        sA[3] = tol_split4(qA[3]); sB[3] = tol_split4(qB[3]);
        sA[0] = tol_split4(qA[0]); sB[0] = tol_split4(qB[0]);

        auto step = [&](int t, const int k) {  // k = (t - t0) & 3, a constant in every expansion
            const int km = (k + 3) & 3, kp = (k + 1) & 3, kn = (k + 2) & 3;
            const int ko = (k + kPairOuterAhead) & 3;
#if EPIC_PAIR_LDS_OUTER
            qA[kn] = ld(ocA, t + 2); qB[kn] = ld(ocB, t + 2); ld_outer(ko, t + kPairOuterAhead);
            {   // (the compiler waits for exactly this slot's two transfers: counted vmcnt)
                const vu4s ra = slot(k)[0][lane], rb = slot(k)[1][lane];
                pa[k] = make_float4(u2f(ra.x), u2f(ra.y), u2f(ra.z), u2f(ra.w));
                pb[k] = make_float4(u2f(rb.x), u2f(rb.y), u2f(rb.z), u2f(rb.w));
            }
#else
            qA[kn] = ld(ocA, t + 2); qB[kn] = ld(ocB, t + 2); pa[ko] = ld(oa, t + kPairOuterAhead); pb[ko] = ld(ob, t + kPairOuterAhead);
#endif
            eA[kn] = eld(ocA, t + 2); eB[kn] = eld(ocB, t + 2);
            hA[kp & 1] = side(cA, t + 1); hB[kp & 1] = side(cB, t + 1);
            sA[kp] = tol_split4(qA[kp]); sB[kp] = tol_split4(qB[kp]);
            const Split4 so_a = tol_split4(pa[k]), so_b = tol_split4(pb[k]);
            // the strip-edge cells of both owned rows in ONE packed split (.x: plane A's -- lane 0 left, lane 63 right --, .y: plane B's)
            const Split2 es = tol_split2(v2f{eA[k], eB[k]});
            const Split2 esA = Split2{v2f{es.q.x, es.q.x}, v2f{es.zm.x, es.zm.x}}, esB = Split2{v2f{es.q.y, es.q.y}, v2f{es.zm.y, es.zm.y}};
            const bool evenA = !RB || ((cA + t + a.parity) & 1) == 0;  // scalar; B has the other colour pattern
            const RowSide &ha = hA[k & 1], &hb = hB[k & 1];
            // the reference's order of the six neighbours: x0 - 1, x0 + 1, x1 - 1, x1 + 1, (x2 - 1, x2 + 1 inside tol_row_3d)
            // both rows' table reads are issued before either is waited for (Jacobi: 8 reads in flight; red-black: 4): the row
            // B's front end covers row A's round trip to LDS and the other way round
            // (X0M = false: pair axis = x0, ring = x1; X0M = true: ring = x0, pair axis = x1)
            auto frontA = [&] {
                return !X0M ? tol_row_3d_front<RB>(pa[k], qB[k], qA[km], qA[k], qA[kp], so_a, sB[k], sA[km], sA[k], sA[kp], eA[k], eA[k], esA, evenA, tl)
                            : tol_row_3d_front<RB>(qA[km], qA[kp], pa[k], qA[k], qB[k], sA[km], sA[kp], so_a, sA[k], sB[k], eA[k], eA[k], esA, evenA, tl);
            };
            auto frontB = [&] {
                return !X0M ? tol_row_3d_front<RB>(qA[k], pb[k], qB[km], qB[k], qB[kp], sA[k], so_b, sB[km], sB[k], sB[kp], eB[k], eB[k], esB, !evenA, tl)
                            : tol_row_3d_front<RB>(qB[km], qB[kp], qA[k], qB[k], pb[k], sB[km], sB[kp], sA[k], sB[k], so_b, eB[k], eB[k], esB, !evenA, tl);
            };
#if EPIC_PAIR_SEQ   // build knob (A/B): row A is finished before row B is begun -- lower register pressure inside the step, the two rows' table reads do not overlap
            TolRowFront fA = frontA();
            const float4 oA = tol_row_3d_back<RB, 2, 0>(fA, qA[k], ha.m0, ha.m1, ha.m2, ha.m3, evenA);
            TolRowFront fB = frontB();
            const float4 oB = tol_row_3d_back<RB, 2, 0>(fB, qB[k], hb.m0, hb.m1, hb.m2, hb.m3, !evenA);
#else
            TolRowFront fA = frontA(), fB = frontB();
            const float4 oA = tol_row_3d_back<RB, 6, (RB ? 2 : 4)>(fA, qA[k], ha.m0, ha.m1, ha.m2, ha.m3, evenA);
            const float4 oB = tol_row_3d_back<RB, 2, 0>(fB, qB[k], hb.m0, hb.m1, hb.m2, hb.m3, !evenA);
#endif
            const bool chk_t = !X0M || (t >= a.check_lo && t < a.check_hi);  // scalar
            if (chkA && chk_t) fold(qA[k], oA);
            if (chkB && chk_t) fold(qB[k], oB);
            store_row(rout, oA.x, oA.y, oA.z, oA.w, laneA, t_off(t));
            store_row(rout, oB.x, oB.y, oB.z, oB.w, laneB, cstride + t_off(t));
        };
        const int nsteps = t1 - t0, nfull = nsteps & ~3;
        for (int i = 0; i < nfull; i += 4) {
#if EPIC_PAIR_BARRIER_ROWS > 0  // experiment: the waves of a workgroup kept within a few steps of each other (they share rows through the caches)
            if ((i & (EPIC_PAIR_BARRIER_ROWS - 1)) == 0) __builtin_amdgcn_s_barrier();
#endif
            step(t0 + i, 0);
            step(t0 + i + 1, 1);
            step(t0 + i + 2, 2);
            step(t0 + i + 3, 3);
        }
        // ragged tail: up to three more steps behind scalar tests (the rings are in phase: nfull % 4 == 0)
        if (nsteps - nfull > 0) step(t0 + nfull, 0);
        if (nsteps - nfull > 1) step(t0 + nfull + 1, 1);
        if (nsteps - nfull > 2) step(t0 + nfull + 2, 2);
    }

    if (CHECK) {
        dmax = wave_max(dmax);
        if (lane == 0 && dmax > 0.0f &&
            __float_as_uint(dmax) > __hip_atomic_load(a.delta_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
            atomicMax(a.delta_bits, __float_as_uint(dmax));
    }
}

// uint32-per-cell mask (m0 x m1 x m2, unpitched) -> lane masks (kernels.h: per x2-row and strip four 64-bit words);
// faces and padding locked.  One wave per (row, strip), four ballots.
__global__ __launch_bounds__(kWave * kWavesPerBlock) void pack_mask_3d_kernel(const uint32_t *locked, int m0, int m1, int m2,
                                                                              int pitch, uint32_t *maskw)
{
    const int lane = threadIdx.x & 63;
    const int nstrips = pitch >> 8;
    const int strip = blockIdx.y * kWavesPerBlock + (threadIdx.x >> 6);
    const size_t row = blockIdx.x;  // x0 * m1 + x1 (rows ride on grid.x: grid.y stops at 65535)
    if (strip >= nstrips) return;   // wave-uniform
    const int x1 = (int)(row % m1), x0 = (int)(row / m1);
    const bool face01 = x0 == 0 || x0 == m0 - 1 || x1 == 0 || x1 == m1 - 1;
    unsigned long long *out = reinterpret_cast<unsigned long long *>(maskw) + (row * nstrips + strip) * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int x2 = strip * kStripCols + lane * 4 + j;
        bool lk = true;
        if (!face01 && x2 > 0 && x2 < m2 - 1) lk = locked[row * m2 + x2] != 0;
        const unsigned long long m = __ballot(lk);
        if (lane == j) out[j] = m;
    }
}

}  // namespace

namespace {
template <bool CHECK, bool RB, bool TRACK>
void launch_sweep_3d_track(int math, dim3 grid, dim3 block, hipStream_t stream, const Sweep3dArgs &a, size_t tiles)
{
    void (*kernel)(Sweep3dArgs) = math == kMathFast  ? sweep3d_kernel<CHECK, kMathFast, RB, TRACK>
                                  : math == kMathTol ? sweep3d_kernel<CHECK, kMathTol, RB, TRACK>
                                                     : sweep3d_kernel<CHECK, kMathPrecise, RB, TRACK>;
    // a list-driven launch is persistent waves: as many as the chip holds of this instantiation (kernels.h)
    if (TRACK && a.wake.list_in) grid = dim3((unsigned)sweep_2d_list_blocks(tiles, resident_blocks_of((const void *)kernel)));
    else if (math == kMathTol) {  // every workgroup stages the 20 KiB table once: as many as the chip keeps resident, in whole groups of 8
        const int res = resident_blocks_of((const void *)kernel) / 8 * 8;
        if (res >= 8 && (int)grid.x > res) grid = dim3((unsigned)res);
    }
    hipLaunchKernelGGL(kernel, grid, block, 0, stream, a);
}
template <bool CHECK, bool RB>
void launch_sweep_3d_math(int math, dim3 grid, dim3 block, hipStream_t stream, const Sweep3dArgs &a, size_t tiles)
{
    if (a.wake.list_out) launch_sweep_3d_track<CHECK, RB, true>(math, grid, block, stream, a, tiles);
    else launch_sweep_3d_track<CHECK, RB, false>(math, grid, block, stream, a, tiles);
}
template <bool CHECK, bool RB, bool X0M> void launch_sweep_3d_pair_axis(dim3 block, hipStream_t stream, const Sweep3dArgs &a)
{
    auto kernel = sweep3d_pair_kernel<CHECK, RB, X0M>;
    // resident workgroups walk the logical blocks (each stages the 20 KiB table once)
    const int res = resident_blocks_of((const void *)kernel);
    const dim3 grid((unsigned)(res >= 8 && a.nblocks > res ? res / 8 * 8 : a.nblocks));   // a multiple of 8: vb % 8 is the XCD
    hipLaunchKernelGGL(kernel, grid, block, 0, stream, a);
}
template <bool CHECK, bool RB> void launch_sweep_3d_pair(dim3 block, hipStream_t stream, const Sweep3dArgs &a, bool x0m)
{
    if (x0m) launch_sweep_3d_pair_axis<CHECK, RB, true>(block, stream, a);
    else launch_sweep_3d_pair_axis<CHECK, RB, false>(block, stream, a);
}
// (LaunchKnobs, driver_config.h: EPIC_HIP_3D_MARCH=x0|x1 -- the axis the pair kernel marches along, x1: pairs of planes, x0: pairs of
//  rows; EPIC_HIP_3D_PAIR=0 -- the one-plane-per-wave kernel for every launch (A/B, tests); EPIC_HIP_3D_PAIR_ROWS overrides the rule below)
// x1-rows per task of the pair kernel
int sweep_3d_pair_rows(int m1, const LaunchKnobs &knobs)
{
    const int v = knobs.pair3d_rows;
    if (v > 0) return v < 4 ? 4 : v > 4096 ? 4096 : v;
    // a multiple of 8 chunks of about kPairRows rows where the grid allows it: the chunk alone then selects the XCD (see the kernel)
    const int rounds = (m1 + 8 * kPairRows / 2) / (8 * kPairRows) > 0 ? (m1 + 8 * kPairRows / 2) / (8 * kPairRows) : 1;
    const int rows = (m1 + 8 * rounds - 1) / (8 * rounds);
    return rows < 4 ? 4 : rows;
}
}  // namespace

// parity < 0: Jacobi (in != out); parity 0 / 1: red-black half-sweep in place (in == out).
hipError_t launch_sweep_3d(const float *in, float *out, const uint32_t *maskw, int m0, int m1, int pitch,
                           int plane_begin, int plane_end, int math, int parity, unsigned *delta_bits,
                           hipStream_t stream, const Activity *act, int check_begin, int check_end, const LaunchKnobs *knobs_in)
{
    const LaunchKnobs &knobs = knobs_in ? *knobs_in : process_launch_knobs();
    if (plane_end <= plane_begin) return hipSuccess;
    if (pitch <= 0 || (pitch % 256) != 0 || m0 <= 0 || m1 <= 0 || plane_begin < 0 || plane_end > m0)
        return hipErrorInvalidValue;
    if ((parity >= 0) != (in == out)) return hipErrorInvalidValue;
    if (math != kMathPrecise && math != kMathFast && math != kMathTol) return hipErrorInvalidValue;  // no df32 / traffic build in 3-D
    // 32-bit byte offsets from one descriptor: a task's rows of three consecutive planes
    if ((long long)pitch * 4 * (kRowsPerTask + 12) + 2LL * m1 * pitch * 4 > 0x7fffffffLL) return hipErrorInvalidValue;
    Sweep3dArgs a;
    a.in = in;
    a.out = out;
    a.maskw = maskw;
    a.delta_bits = delta_bits;
    a.m0 = m0;
    a.m1 = m1;
    a.pitch = pitch;
    a.plane_begin = plane_begin;
    a.plane_end = plane_end;
    a.nstrips = (pitch + kStripCols - 1) / kStripCols;
    a.nchunks = (m1 + kRowsPerTask - 1) / kRowsPerTask;
    a.nplane_groups = (plane_end - plane_begin + kWavesPerBlock - 1) / kWavesPerBlock;
    const long long nblocks = ((long long)a.nchunks * a.nplane_groups + 7) / 8 * 8 * a.nstrips;   // (group, chunk) pairs padded to whole rounds of the 8 XCDs
    if (nblocks > 0x7fffffffLL) return hipErrorInvalidValue;
    a.nblocks = (int)nblocks;
    a.check_lo = check_begin < 0 ? plane_begin : check_begin;
    a.check_hi = check_begin < 0 ? plane_end : check_end;
    a.parity = parity < 0 ? 0 : (parity & 1);
    const bool whole = plane_begin == 0 && plane_end == m0;
    const size_t tiles = sweep_3d_tiles(m0, m1, pitch);
    a.wake = wake_args(whole ? act : nullptr, tiles);
    a.rows = kRowsPerTask;
    if (math == kMathTol && !a.wake.list_out && !a.wake.list_in && knobs.pair3d) {  // two planes per wave (no work lists)
        bool x0m = knobs.march_x0;
        const long long plane_b = (long long)m1 * pitch * 4, row_b = (long long)pitch * 4;
        a.rows = sweep_3d_pair_rows(x0m ? plane_end - plane_begin : m1, knobs);
        // 32-bit byte offsets from one descriptor: the task's steps along the march axis and four steps of the pair axis
        if (x0m && plane_b * (a.rows + 12) + 3 * row_b > 0x7fffffffLL) x0m = false, a.rows = sweep_3d_pair_rows(m1, knobs);
        if (!x0m && row_b * (a.rows + 12) + 3 * plane_b > 0x7fffffffLL) return hipErrorInvalidValue;
        const int march = x0m ? plane_end - plane_begin : m1, pairs = x0m ? m1 : plane_end - plane_begin;
        a.nchunks = (march + a.rows - 1) / a.rows;
        a.nplane_groups = (pairs + 2 * kPairWaves - 1) / (2 * kPairWaves);
        const long long nb = ((long long)a.nchunks * a.nplane_groups + 7) / 8 * 8 * a.nstrips;   // (group, chunk) pairs padded to whole rounds of the 8 XCDs
        if (nb > 0x7fffffffLL) return hipErrorInvalidValue;
        a.nblocks = (int)nb;
        const dim3 block(kWave * kPairWaves);
        if (parity < 0) {
            if (delta_bits) launch_sweep_3d_pair<true, false>(block, stream, a, x0m);
            else launch_sweep_3d_pair<false, false>(block, stream, a, x0m);
        } else {
            if (delta_bits) launch_sweep_3d_pair<true, true>(block, stream, a, x0m);
            else launch_sweep_3d_pair<false, true>(block, stream, a, x0m);
        }
        return hipGetLastError();
    }
    const dim3 grid((unsigned)nblocks), block(kWave * kWavesPerBlock);   // (list-driven launches: resized in launch_sweep_3d_track)
    if (parity < 0) {
        if (delta_bits) launch_sweep_3d_math<true, false>(math, grid, block, stream, a, tiles);
        else launch_sweep_3d_math<false, false>(math, grid, block, stream, a, tiles);
    } else {
        if (delta_bits) launch_sweep_3d_math<true, true>(math, grid, block, stream, a, tiles);
        else launch_sweep_3d_math<false, true>(math, grid, block, stream, a, tiles);
    }
    return hipGetLastError();
}

hipError_t launch_pack_mask_3d(const uint32_t *locked, int m0, int m1, int m2, int pitch, uint32_t *maskw,
                               hipStream_t stream)
{
    const int nstrips = pitch / kStripCols;
    const size_t nrows = (size_t)m0 * (size_t)m1;
    if (nrows > 0x7fffffffu) return hipErrorInvalidValue;
    hipLaunchKernelGGL(pack_mask_3d_kernel, dim3((unsigned)nrows, (unsigned)((nstrips + kWavesPerBlock - 1) / kWavesPerBlock)),
                       dim3(kWave * kWavesPerBlock), 0, stream, locked, m0, m1, m2, pitch, maskw);
    return hipGetLastError();
}

}  // namespace epic_hip
