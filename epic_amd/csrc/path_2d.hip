// path_2d.hip -- streamline extraction on the DEVICE-RESIDENT field (gfx950).
//
// The consumer of the relaxation: the navigation node answers every path request by copying the
// whole field to the host and walking it there (src/epic_navigation_node_harmonic.cpp:614-674 -> harmonic_compute_path_2d_cpu,
// libepic/src/harmonic/harmonic_path_cpu.cpp:154-221).  For an 8192 x 8192 grid that copy is 268 MB per request.  Here
// the walk runs where the field lives: one lane per path, any number of start points per launch, only the way-points come
// back.  The walk is serial by nature (each step needs the previous point), so the win is the avoided copy and the batch,
// not the arithmetic.
//
// The arithmetic follows the host implementation (harmonic_path_cpu.cpp in this directory, itself a restatement of the
// reference file above) operation by operation -- f32 products and sums uncontracted, correctly rounded f32 division, the
// norm evaluated in f64 and narrowed -- so way-points are bit-identical to harmonic_compute_path_2d_cpu on the same field
// (tests/test_gpu_parity.py).  Field access goes through the solver's private layout: pitched u, bit-packed tiled mask.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"

namespace epic_hip {
namespace {

struct PathField {
    const float *u;
    const uint32_t *maskw;
    unsigned w, h;  // columns, rows of the caller's grid
    unsigned pitch;
};

// x86-64 converts float -> unsigned through a 64-bit signed truncation and keeps the low word: negative values below -1
// wrap to huge indices (and then fail the bounds test) instead of saturating to 0 as v_cvt_u32_f32 would.
__device__ __forceinline__ unsigned to_index(float v) { return (unsigned)(long long)v; }

// (indices are clamped: where the host code would read outside its arrays -- a start on a goal cell of the border row
// with a large cdPrecision -- the device reads the nearest cell instead of faulting)
__device__ __forceinline__ float at(const PathField &f, unsigned x, unsigned y)
{
    x = min(x, f.w - 1u);
    y = min(y, f.h - 1u);
    return f.u[(size_t)y * f.pitch + x];
}

__device__ __forceinline__ bool locked(const PathField &f, unsigned x, unsigned y)
{
    x = min(x, f.w - 1u);
    y = min(y, f.h - 1u);
    return (f.maskw[mask_word_2d(y, x, f.pitch)] >> mask_bit_2d(x)) & 1u;
}

// harmonic_path_cpu.cpp:55-57: outside the grid, or a locked cell with u < 0 (an obstacle; goals are locked with u == 0).
// Both loads are issued unconditionally (indices are clamped, so they are always safe): a walk is one long chain of
// dependent memory round trips, and every load that does not wait for a branch shortens it.
__device__ __forceinline__ bool usable(const PathField &f, float x, float y)
{
    const unsigned cx = to_index(x + 0.5f), cy = to_index(y + 0.5f);
    const bool lk = locked(f, cx, cy);
    const float v = at(f, cx, cy);
    return (cx < f.w) & (cy < f.h) & (!lk | !(v < 0.0f));
}

// harmonic_path_cpu.cpp:63-80
__device__ __forceinline__ float bilinear(const PathField &f, float x, float y)
{
    const unsigned x0 = to_index(x - 0.5f), x1 = to_index(x + 0.5f);
    const unsigned y0 = to_index(y - 0.5f), y1 = to_index(y + 0.5f);
    const float alpha = x - (float)x0, beta = y - (float)y0;
    const float top = (1.0f - alpha) * at(f, x0, y0) + alpha * at(f, x1, y0);
    const float bottom = (1.0f - alpha) * at(f, x0, y1) + alpha * at(f, x1, y1);
    return (1.0f - beta) * top + beta * bottom;
}

// harmonic_path_cpu.cpp:85-118.  Returns false where the host returns EPIC_ERROR_INVALID_GRADIENT.
__device__ __forceinline__ bool gradient(const PathField &f, float x, float y, float cd, float &gx, float &gy)
{
    const float xl = x - cd, xr = x + cd, yu = y - cd, yd = y + cd;
    // the sample points sit within cd of a usable point, but bilinear() reads up to one cell further: stay inside.
    // (All 24 loads of a step -- four usability tests, four bilinear samples -- are independent of one another and of
    // the test's outcome, so they travel together: one memory round trip per step instead of three.)
    const bool ok = usable(f, xl, y) & usable(f, xr, y) & usable(f, x, yu) & usable(f, x, yd);
    const float v0 = bilinear(f, xl, y), v1 = bilinear(f, xr, y), v2 = bilinear(f, x, yu), v3 = bilinear(f, x, yd);
    if (!ok) return false;
    gx = (v1 - v0) / (2.0f * cd);
    gy = (v3 - v2) / (2.0f * cd);
    const double dx = (double)gx, dy = (double)gy;
    const float norm = (float)sqrt(dx * dx + dy * dy);
    gx /= norm;
    gy /= norm;
    return true;
}

// One lane per path.  pts: n_paths x 2 * max_points floats; k / rc: per path.
__global__ void follow_paths_kernel(PathField f, unsigned n_paths, const float *starts, float step, float cd,
                                    unsigned max_points, float *pts_all, unsigned *k_out, int *rc_out)
{
    const unsigned id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= n_paths) return;
    float x = starts[2 * id], y = starts[2 * id + 1];
    float *pts = pts_all + (size_t)id * 2 * max_points;
    k_out[id] = 0;
    if (!usable(f, x, y)) {
        rc_out[id] = 10;  // EPIC_ERROR_INVALID_LOCATION
        return;
    }
    unsigned n = 0;  // points so far
    const float half = step / 2.0f;
    if (max_points > 0) { pts[0] = x; pts[1] = y; }
    n = 1;
    // the (up to) five points before the newest one, newest first, in registers: the stuck test must not wait for the
    // way-points to come back from memory
    float hx0 = 0.0f, hy0 = 0.0f, hx1 = 0.0f, hy1 = 0.0f, hx2 = 0.0f, hy2 = 0.0f, hx3 = 0.0f, hy3 = 0.0f, hx4 = 0.0f, hy4 = 0.0f;
    auto near = [&](float qx, float qy) {
        const double dx = (double)(x - qx), dy = (double)(y - qy);
        return (float)sqrt(dx * dx + dy * dy) < half;
    };
    // the host loop runs while size() < 2 * maxLength values, i.e. n < max_points points
    for (;;) {
        const unsigned cx = to_index(x + 0.5f), cy = to_index(y + 0.5f);
        // everything this step reads depends on (x, y) only: the lock of the current cell and the gradient samples
        const bool lk = locked(f, cx, cy);
        float gx = 0.0f, gy = 0.0f;
        const bool grad_ok = gradient(f, x, y, cd, gx, gy);
        if (!(n < max_points && cx < f.w && cy < f.h && !lk)) break;
        // stuck: the newest point is within step / 2 of one of the (up to) five before it (harmonic_path_cpu.cpp:121-151)
        const unsigned back = n - 1 < 5u ? n - 1 : 5u;
        const bool is_stuck = (back >= 1 && near(hx0, hy0)) | (back >= 2 && near(hx1, hy1)) | (back >= 3 && near(hx2, hy2)) |
                              (back >= 4 && near(hx3, hy3)) | (back >= 5 && near(hx4, hy4));
        if (is_stuck) break;
        if (!grad_ok) {
            rc_out[id] = 12;  // EPIC_ERROR_INVALID_GRADIENT
            return;
        }
        hx4 = hx3; hy4 = hy3; hx3 = hx2; hy3 = hy2; hx2 = hx1; hy2 = hy1; hx1 = hx0; hy1 = hy0; hx0 = x; hy0 = y;
        x += gx * step;
        y += gy * step;
        pts[2 * n] = x;
        pts[2 * n + 1] = y;
        n++;
    }
    if (n <= 2) {
        rc_out[id] = 13;  // EPIC_ERROR_INVALID_PATH
        return;
    }
    k_out[id] = n;
    rc_out[id] = 0;
}

}  // namespace

hipError_t launch_follow_paths_2d(const float *u, const uint32_t *maskw, int rows, int cols, int pitch, unsigned n_paths,
                                  const float *d_starts, float step, float cd, unsigned max_points, float *d_pts,
                                  unsigned *d_k, int *d_rc, hipStream_t stream)
{
    if (n_paths == 0) return hipSuccess;
    PathField f{u, maskw, (unsigned)cols, (unsigned)rows, (unsigned)pitch};
    hipLaunchKernelGGL(follow_paths_kernel, dim3((n_paths + 63) / 64), dim3(64), 0, stream, f, n_paths, d_starts, step, cd,
                       max_points, d_pts, d_k, d_rc);
    return hipGetLastError();
}

}  // namespace epic_hip
