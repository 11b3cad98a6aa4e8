// kernels_3d.hip -- 3-D (n = 3) log-space Jacobi sweep for gfx950, 7-point stencil.
//
// The reference has no GPU code for n = 3 (the branches are empty: libepic/src/harmonic/harmonic_gpu.cu:160-162,
// :334-336, :367-369); the arithmetic follows its CPU sweep (libepic/src/harmonic/harmonic_cpu.cpp:81-133):
// neighbours in the order x0-1, x0+1, x1-1, x1+1, x2-1, x2+1, constant log(6.0).
//
// Layout: u[x0][x1][x2] with x2 contiguous and padded to `pitch` (multiple of 256 floats).  A wave owns 256 x2-columns
// (4 per lane, one dwordx4) of one x0-plane and marches along x1, keeping rows x1-1 / x1 / x1+1 of its plane in
// registers; x2 neighbours are full-wave DPP shifts; the rows of planes x0-1 and x0+1 are loaded per step.  The four
// waves of a workgroup sweep four consecutive planes of the same (x1-chunk, strip), so those extra rows are the
// sibling waves' centre rows and are served by the CU's L1 / the XCD's L2 rather than HBM.
// Mask: 1 bit per cell, 32 consecutive x2 cells per word.
// Optional activity tracking (wake.h): a tile is one task -- 32 x1-rows x 256 x2-columns of one x0-plane; it reads its
// own cells, the adjacent column / row of its four in-plane neighbours and the whole tile of the planes x0 - 1 and x0 + 1.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cell_update.h"
#include "kernels.h"
#include "wake.h"

namespace epic_hip {

namespace {

constexpr int kWave = 64;
constexpr int kStripCols = 256;
constexpr int kWavesPerBlock = 4;  // = consecutive planes per workgroup
constexpr int kRowsPerTask = 32;

struct Sweep3dArgs {
    const float *in;
    float *out;
    const uint32_t *maskw;
    unsigned *delta_bits;
    int m0, m1, pitch;
    int plane_begin, plane_end;
    int nstrips, nchunks, nplane_groups;
    int parity;  // red-black scheme only: currentIteration & 1
    WakeArgs wake;  // TRACK kernels, whole-grid launches only
};

// RB = true: the reference's 3-D red-black half-sweep in place (in == out): cells with (x0 + x1 + x2 + currentIteration)
// even are recomputed (harmonic_cpu.cpp:89-102), all six neighbours have the other colour.
template <bool CHECK, int MATH, bool RB, bool TRACK>
__global__ __launch_bounds__(kWave * kWavesPerBlock) void sweep3d_kernel(Sweep3dArgs a)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (TRACK) wake_reset_next(a.wake);
    const bool listed = TRACK && a.wake.list_in != nullptr;
    WakeCursor cursor = {};
    if (listed && !wake_begin(a.wake, lane, wave, kWavesPerBlock, cursor)) return;
    MathTab lds = {};  // libm tables, one entry per lane (precise math only)
    if (MATH == kMathPrecise) lds = math_tables_load();
    float dmax = 0.0f;

    for (;;) {  // one pass per task: exactly one unless the launch is list-driven
    int strip, chunk, x0;
    if (listed) {
        int t = wake_tile(a.wake, cursor);  // tile id = (x0 * nchunks + chunk) * nstrips + strip
        strip = t % a.nstrips;
        t /= a.nstrips;
        chunk = t % a.nchunks;
        x0 = t / a.nchunks;
    } else {
        int b = blockIdx.x;
        strip = b % a.nstrips;
        b /= a.nstrips;
        chunk = b % a.nchunks;
        x0 = a.plane_begin + (b / a.nchunks) * kWavesPerBlock + wave;
    }
    if (x0 >= a.plane_end) break;  // wave-uniform
    const int tile = (x0 * a.nchunks + chunk) * a.nstrips + strip;
    if (TRACK && lane == 0) a.wake.queued_in[tile] = 0;
    const int r0 = chunk * kRowsPerTask;
    const int r1 = min(r0 + kRowsPerTask, a.m1);

    const int col0 = strip * kStripCols;
    const int col = col0 + lane * 4;
    const int lcol = col;  // pitch % 256 == 0: every lane is in bounds
    const int hcol = (lane == 0) ? max(col0 - 1, 0) : min(col0 + kStripCols, a.pitch - 1);
    const bool edge_lane = (lane == 0) | (lane == 63);
    const size_t pitch = (size_t)a.pitch;
    const size_t plane = (size_t)a.m1 * pitch;
    const int wpitch = a.pitch >> 5;

    const float *pc = a.in + (size_t)x0 * plane;
    const float *pa = a.in + (size_t)max(x0 - 1, 0) * plane;
    const float *pb = a.in + (size_t)min(x0 + 1, a.m0 - 1) * plane;
    const int rlast = a.m1 - 1;

    auto ld = [&](const float *p, int r) -> float4 {
        r = min(max(r, 0), rlast);
        return *reinterpret_cast<const float4 *>(p + (size_t)r * pitch + lcol);
    };
    auto ldh = [&](int r) -> float {
        float h = 0.0f;
        r = min(max(r, 0), rlast);
        if (edge_lane) h = pc[(size_t)r * pitch + hcol];
        return h;
    };
    auto ldm = [&](int r) -> uint32_t {
        r = min(r, rlast);
        return a.maskw[((size_t)x0 * a.m1 + r) * wpitch + (lcol >> 5)];
    };

    float4 up = ld(pc, r0 - 1), c = ld(pc, r0), d1 = ld(pc, r0 + 1);
    float4 a1 = ld(pa, r0), b1 = ld(pb, r0);
    float hc = ldh(r0), h1 = ldh(r0 + 1);
    uint32_t mw = ldm(r0);
    bool chg_any = false, chg_x = false, chg_w = false, chg_top = false, chg_bot = false;  // as in the 2-D kernel

    for (int r = r0; r < r1; ++r) {
        const float4 d2 = ld(pc, r + 2);
        const float4 a2 = ld(pa, r + 1), b2 = ld(pb, r + 1);
        const float h2 = ldh(r + 2);
        const uint32_t mw2 = ldm(r + 1);

        const float lf = wave_from_left(c.w, hc);
        const float rt = wave_from_right(c.x, hc);
        const uint32_t nib = mw >> (lcol & 31);

        float4 o;
        if (RB) {
            o = c;
            if (((x0 + r + a.parity) & 1) == 0) {  // scalar: even x2 columns (.x, .z) are this row's active cells
                const float nx = cell_update_3d<MATH>(a1.x, b1.x, up.x, d1.x, lf, c.y, lds);
                const float nz = cell_update_3d<MATH>(a1.z, b1.z, up.z, d1.z, c.y, c.w, lds);
                o.x = (nib & 1u) ? c.x : nx;
                o.z = (nib & 4u) ? c.z : nz;
            } else {
                const float ny = cell_update_3d<MATH>(a1.y, b1.y, up.y, d1.y, c.x, c.z, lds);
                const float nw = cell_update_3d<MATH>(a1.w, b1.w, up.w, d1.w, c.z, rt, lds);
                o.y = (nib & 2u) ? c.y : ny;
                o.w = (nib & 8u) ? c.w : nw;
            }
        } else {
            o.x = cell_update_3d<MATH>(a1.x, b1.x, up.x, d1.x, lf, c.y, lds);
            o.y = cell_update_3d<MATH>(a1.y, b1.y, up.y, d1.y, c.x, c.z, lds);
            o.z = cell_update_3d<MATH>(a1.z, b1.z, up.z, d1.z, c.y, c.w, lds);
            o.w = cell_update_3d<MATH>(a1.w, b1.w, up.w, d1.w, c.z, rt, lds);
            o.x = (nib & 1u) ? c.x : o.x;
            o.y = (nib & 2u) ? c.y : o.y;
            o.z = (nib & 4u) ? c.z : o.z;
            o.w = (nib & 8u) ? c.w : o.w;
        }
        if (CHECK) {
            dmax = max2(dmax, fabsf(c.x - o.x));
            dmax = max2(dmax, fabsf(c.y - o.y));
            dmax = max2(dmax, fabsf(c.z - o.z));
            dmax = max2(dmax, fabsf(c.w - o.w));
        }
        if (TRACK) {
            const bool cx = f2u(o.x) != f2u(c.x), cw = f2u(o.w) != f2u(c.w);
            const bool rc = cx | cw | (f2u(o.y) != f2u(c.y)) | (f2u(o.z) != f2u(c.z));
            chg_any |= rc;
            chg_x |= cx;
            chg_w |= cw;
            if (r == r0) chg_top = rc;
            if (r == r1 - 1) chg_bot = rc;
        }
        *reinterpret_cast<float4 *>(a.out + (size_t)x0 * plane + (size_t)r * pitch + col) = o;

        up = c; c = d1; d1 = d2;
        a1 = a2; b1 = b2;
        hc = h1; h1 = h2;
        mw = mw2;
    }

    if (TRACK) {
        // wake the tiles that read what this task changed: itself, the in-plane neighbours across the edges that
        // changed, and the same tile of the two neighbouring planes (every cell has an x0 - 1 and an x0 + 1 neighbour)
        const bool any = __ballot(chg_any) != 0;
        const bool first_col = (__ballot(chg_x) & 1ull) != 0, last_col = (__ballot(chg_w) >> 63) != 0;
        const bool first_row = __ballot(chg_top) != 0, last_row = __ballot(chg_bot) != 0;
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
    if (!listed || !wake_next(cursor)) break;
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

// uint32-per-cell mask (m0 x m1 x m2, unpitched) -> 1 bit per cell, 32 x2-cells per word; faces and padding locked.
__global__ void pack_mask_3d_kernel(const uint32_t *locked, int m0, int m1, int m2, int pitch, uint32_t *maskw)
{
    const int wpitch = pitch >> 5;
    const size_t nwords = (size_t)m0 * m1 * wpitch;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nwords) return;
    const int w = (int)(i % wpitch);
    const size_t row = i / wpitch;
    const int x1 = (int)(row % m1);
    const int x0 = (int)(row / m1);
    const bool face01 = x0 == 0 || x0 == m0 - 1 || x1 == 0 || x1 == m1 - 1;
    uint32_t bits = 0;
    for (int k = 0; k < 32; ++k) {
        const int x2 = w * 32 + k;
        bool lk = true;
        if (!face01 && x2 > 0 && x2 < m2 - 1) lk = locked[((size_t)x0 * m1 + x1) * m2 + x2] != 0;
        bits |= (lk ? 1u : 0u) << k;
    }
    maskw[i] = bits;
}

}  // namespace

namespace {
template <bool CHECK, bool RB, bool TRACK>
void launch_sweep_3d_track(int math, dim3 grid, dim3 block, hipStream_t stream, const Sweep3dArgs &a)
{
    if (math == kMathFast) hipLaunchKernelGGL((sweep3d_kernel<CHECK, kMathFast, RB, TRACK>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((sweep3d_kernel<CHECK, kMathPrecise, RB, TRACK>), grid, block, 0, stream, a);
}
template <bool CHECK, bool RB>
void launch_sweep_3d_math(int math, dim3 grid, dim3 block, hipStream_t stream, const Sweep3dArgs &a)
{
    if (a.wake.list_out) launch_sweep_3d_track<CHECK, RB, true>(math, grid, block, stream, a);
    else launch_sweep_3d_track<CHECK, RB, false>(math, grid, block, stream, a);
}
}  // namespace

// parity < 0: Jacobi (in != out); parity 0 / 1: red-black half-sweep in place (in == out).
hipError_t launch_sweep_3d(const float *in, float *out, const uint32_t *maskw, int m0, int m1, int pitch,
                           int plane_begin, int plane_end, int math, int parity, unsigned *delta_bits,
                           hipStream_t stream, const Activity *act)
{
    if (plane_end <= plane_begin) return hipSuccess;
    if (pitch <= 0 || (pitch % 256) != 0 || m0 <= 0 || m1 <= 0 || plane_begin < 0 || plane_end > m0)
        return hipErrorInvalidValue;
    if ((parity >= 0) != (in == out)) return hipErrorInvalidValue;
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
    const long long nblocks = (long long)a.nstrips * a.nchunks * a.nplane_groups;
    if (nblocks > 0x7fffffffLL) return hipErrorInvalidValue;
    a.parity = parity < 0 ? 0 : (parity & 1);
    const bool whole = plane_begin == 0 && plane_end == m0;
    const size_t tiles = sweep_3d_tiles(m0, m1, pitch);
    a.wake = wake_args(whole ? act : nullptr, tiles);
    const dim3 grid(a.wake.list_in ? (unsigned)sweep_2d_list_blocks(tiles) : (unsigned)nblocks), block(kWave * kWavesPerBlock);
    if (parity < 0) {
        if (delta_bits) launch_sweep_3d_math<true, false>(math, grid, block, stream, a);
        else launch_sweep_3d_math<false, false>(math, grid, block, stream, a);
    } else {
        if (delta_bits) launch_sweep_3d_math<true, true>(math, grid, block, stream, a);
        else launch_sweep_3d_math<false, true>(math, grid, block, stream, a);
    }
    return hipGetLastError();
}

hipError_t launch_pack_mask_3d(const uint32_t *locked, int m0, int m1, int m2, int pitch, uint32_t *maskw,
                               hipStream_t stream)
{
    const size_t nwords = mask_words_3d(m0, m1, pitch);
    hipLaunchKernelGGL(pack_mask_3d_kernel, dim3((unsigned)((nwords + 255) / 256)), dim3(256), 0, stream, locked, m0, m1,
                       m2, pitch, maskw);
    return hipGetLastError();
}

}  // namespace epic_hip
