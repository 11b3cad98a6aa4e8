// stream_lab.hip -- what the HBM side of the 2-D sweep's access pattern can reach on gfx950, variant by variant.
// 8192 x 8192 f32 in -> out (268 MB each way, 537 MB per launch, same as one sweep).
//   hipcc --offload-arch=gfx950 -O3 tools/stream_lab.hip -o tools/stream_lab && tools/stream_lab
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

constexpr int N = 8192;
typedef float vf4 __attribute__((ext_vector_type(4)));

// V0: flat grid-stride dwordx4 copy
__global__ __launch_bounds__(256) void flat_copy(const float4 *in, float4 *out, size_t n4)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n4; i += stride) out[i] = in[i];
}

// strip-march variants: one wave = 256 columns x RPT rows
template <int MODE, bool NT>
__global__ __launch_bounds__(256) void strip(const float *in, float *out, const uint32_t *mask, int rpt, int nstrips, int ntasks)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int task = blockIdx.x * 4 + wave;
    if (task >= ntasks) return;
    const int strip_i = task % nstrips, chunk = task / nstrips;
    const int r0 = chunk * rpt, r1 = min(r0 + rpt, N);
    const int col = strip_i * 256 + lane * 4;
    const int hcol = lane == 0 ? max(strip_i * 256 - 1, 0) : min(strip_i * 256 + 256, N - 1);
    auto ld = [&](int r) { r = min(max(r, 0), N - 1); const vf4 *p = reinterpret_cast<const vf4 *>(in + (size_t)r * N + col);
                           vf4 v = NT ? __builtin_nontemporal_load(p) : *p; return make_float4(v.x, v.y, v.z, v.w); };
    auto ldh = [&](int r) { float h = 0; r = min(max(r, 0), N - 1); if (lane == 0 || lane == 63) h = in[(size_t)r * N + hcol]; return h; };
    auto st = [&](int r, float4 v) { vf4 *p = reinterpret_cast<vf4 *>(out + (size_t)r * N + col); vf4 w = {v.x, v.y, v.z, v.w};
                                     if (NT) __builtin_nontemporal_store(w, p); else *p = w; };
    if (MODE == 1) {  // plain strip copy, one row in flight ahead
        float4 c = ld(r0);
        for (int r = r0; r < r1; ++r) { float4 n = ld(r + 1); st(r, c); c = n; }
        return;
    }
    float4 up = ld(r0 - 1), c = ld(r0), d1 = ld(r0 + 1);
    float hc = MODE >= 3 ? ldh(r0) : 0.f, h1 = MODE >= 3 ? ldh(r0 + 1) : 0.f;
    uint32_t mw = 0;
    for (int r = r0; r < r1; ++r) {
        float4 d2 = ld(r + 2);
        float h2 = MODE >= 3 ? ldh(r + 2) : 0.f;
        if (MODE >= 4 && (r & 7) == 0) mw = mask[(size_t)(r >> 3) * (N / 4) + (col >> 2)];
        float4 o;
        o.x = up.x + d1.x + c.y + hc; o.y = up.y + d1.y + c.x + c.z; o.z = up.z + d1.z + c.y + c.w; o.w = up.w + d1.w + c.z + hc;
        if (MODE >= 4 && (mw >> ((r & 7) * 4) & 1)) o.x = c.x;
        st(r, o);
        up = c; c = d1; d1 = d2; hc = h1; h1 = h2;
    }
}

template <typename F>
static float time_it(F launch, int reps = 20)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; i++) launch();
    hipEventRecord(e0);
    for (int i = 0; i < reps; i++) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / reps;
}

int main()
{
    float *a, *b; uint32_t *m;
    const size_t bytes = (size_t)N * N * 4;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&m, (size_t)N * N / 8);
    hipMemset(a, 0, bytes); hipMemset(b, 0, bytes); hipMemset(m, 0, (size_t)N * N / 8);
    auto report = [&](const char *name, float us) { printf("%-44s %8.1f us  %6.2f TB/s\n", name, us, 2.0 * bytes / us * 1e-6); };
    for (int blocks : {2048, 8192, 65536})
        { char nm[64]; snprintf(nm, 64, "flat copy, %d blocks", blocks);
          report(nm, time_it([&] { hipLaunchKernelGGL(flat_copy, dim3(blocks), dim3(256), 0, 0, (const float4 *)a, (float4 *)b, bytes / 16); })); }
    {   // ping-pong (a -> b, b -> a, ...), as consecutive sweeps do: every launch reads what the previous one wrote
        int flip = 0;
        for (int blocks : {65536}) {
            report("flat copy ping-pong, 65536 blocks", time_it([&] { const float4 *src = (const float4 *)(flip ? b : a); float4 *dst = (float4 *)(flip ? a : b); flip ^= 1;
                hipLaunchKernelGGL(flat_copy, dim3(blocks), dim3(256), 0, 0, src, dst, bytes / 16); }));
        }
        const int ns = N / 256;
        for (int rpt : {8, 16, 64}) {
            const int ntasks = ns * ((N + rpt - 1) / rpt), nblk = (ntasks + 3) / 4;
            char nm[96];
            snprintf(nm, 96, "strip rpt=%d 3-row + halo + mask nt ping-pong", rpt);
            report(nm, time_it([&] { const float *src = flip ? b : a; float *dst = flip ? a : b; flip ^= 1;
                hipLaunchKernelGGL((strip<4, true>), dim3(nblk), dim3(256), 0, 0, src, dst, m, rpt, ns, ntasks); }));
            snprintf(nm, 96, "strip rpt=%d 3-row stencil nt ping-pong", rpt);
            report(nm, time_it([&] { const float *src = flip ? b : a; float *dst = flip ? a : b; flip ^= 1;
                hipLaunchKernelGGL((strip<2, true>), dim3(nblk), dim3(256), 0, 0, src, dst, m, rpt, ns, ntasks); }));
        }
    }
    const int nstrips = N / 256;
    for (int rpt : {8, 16, 64, 256}) {
        const int ntasks = nstrips * ((N + rpt - 1) / rpt), nblk = (ntasks + 3) / 4;
        char nm[96];
#define RUN(MODE, NT, LABEL) snprintf(nm, 96, "strip rpt=%d %s%s", rpt, LABEL, NT ? " nt" : ""); \
        report(nm, time_it([&] { hipLaunchKernelGGL((strip<MODE, NT>), dim3(nblk), dim3(256), 0, 0, a, b, m, rpt, nstrips, ntasks); }));
        RUN(1, false, "copy") RUN(1, true, "copy")
        RUN(2, false, "3-row stencil") RUN(2, true, "3-row stencil")
        RUN(3, false, "3-row + halo") RUN(4, false, "3-row + halo + mask") RUN(4, true, "3-row + halo + mask")
    }
    return 0;
}
