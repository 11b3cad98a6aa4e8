// ubench_mfma_f64.hip -- can the f64 matrix pipe serve as a second FMA port for the precise-math kernel? (gfx950)
//
// The precise cell update is bound by f64 VALU issue (~58 four-cycle instructions per cell).  About half of them are
// d = CONST * x + y.  v_mfma_f64_4x4x4f64 with A = CONST * I (per 4x4 block) computes exactly that for all 64 lanes:
// D[i][j] = sum_k A[i][k] B[k][j] + C[i][j] = CONST * B[i][j] + C[i][j], the other three products being exact zeros.
// Questions answered here:
//   1. layout / exactness: with A = c on the diagonal lanes, is D bit-identical to fma(c, b, cc) lane by lane?
//   2. issue cost of the MFMA alone, and whether it overlaps with v_fma_f64 issued by the same and by other waves.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_mfma_f64.hip -o tools/ubench_mfma_f64 && tools/ubench_mfma_f64
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

__global__ void probe(const double *b, const double *c, double cst, double *d_mfma, double *d_fma, int diag_only)
{
    const int lane = threadIdx.x;
    const int e = lane & 15;
    const double a = (!diag_only || e == 0 || e == 5 || e == 10 || e == 15) ? cst : 0.0;
    d_mfma[lane] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b[lane], c[lane], 0, 0, 0);
    d_fma[lane] = fma(cst, b[lane], c[lane]);
}

#define BODY_FMA(N)                                                                                  \
    _Pragma("unroll") for (int j = 0; j < N; j++) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x[j]) : "v"(p), "v"(q));
#define BODY_MFMA(N)                                                                                 \
    _Pragma("unroll") for (int j = 0; j < N; j++) asm volatile("v_mfma_f64_4x4x4f64 %0, %1, %2, %0" : "+v"(y[j]) : "v"(am), "v"(p));

template <int NF, int NM>
__global__ __launch_bounds__(256) void mix(float *out, int iters, unsigned long long *clk)
{
    double x[16], y[8];
    const double p = 1.0000001, q = 1e-9;
    const int e = threadIdx.x & 15;
    const double am = (e == 0 || e == 5 || e == 10 || e == 15) ? 1.0000001 : 0.0;
    for (int j = 0; j < 16; j++) x[j] = threadIdx.x * 1e-3 + j;
    for (int j = 0; j < 8; j++) y[j] = threadIdx.x * 1e-3 + j;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; i++) {
        // interleave: NF VALU FMAs and NM MFMAs per iteration, all independent of each other within the iteration
        BODY_MFMA(NM)
        BODY_FMA(NF)
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    double t = 0;
    for (int j = 0; j < 16; j++) t += x[j];
    for (int j = 0; j < 8; j++) t += y[j];
    if (t == 123.456) out[0] = (float)t;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

typedef void (*kern_t)(float *, int, unsigned long long *);

static void run(const char *name, kern_t k, int nf, int nm, float *out, unsigned long long *dclk, int waves_per_simd)
{
    const int iters = 4000;
    const int blocks = 256 * waves_per_simd;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, iters, dclk);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, iters, dclk);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long clk[2];
    hipMemcpy(clk, dclk, 16, hipMemcpyDeviceToHost);
    const double ghz = (double)clk[0] / ((double)clk[1] * 10.0);
    const double cyc_iter = (double)ms * 1e-3 * ghz * 1e9 / ((double)iters * waves_per_simd);
    printf("  %-22s waves/SIMD %d: %7.2f cycles per wave-iteration (%2d v_fma_f64 + %d mfma_4x4x4)  [%.3f ms, %.2f GHz]\n", name,
           waves_per_simd, cyc_iter, nf, nm, ms, ghz);
}

int main()
{
    // 1. layout / exactness
    double hb[64], hc[64], hm[64], hf[64], *b, *c, *dm, *df;
    srand(7);
    for (int i = 0; i < 64; i++) {
        hb[i] = ldexp((double)rand() / RAND_MAX - 0.5, rand() % 40 - 20);
        hc[i] = ldexp((double)rand() / RAND_MAX - 0.5, rand() % 40 - 20);
    }
    hipMalloc(&b, 512); hipMalloc(&c, 512); hipMalloc(&dm, 512); hipMalloc(&df, 512);
    hipMemcpy(b, hb, 512, hipMemcpyHostToDevice); hipMemcpy(c, hc, 512, hipMemcpyHostToDevice);
    for (int diag = 1; diag >= 0; diag--) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, b, c, 0x1.71547652b82fep+5, dm, df, diag);
        hipMemcpy(hm, dm, 512, hipMemcpyDeviceToHost); hipMemcpy(hf, df, 512, hipMemcpyDeviceToHost);
        int same = 0;
        for (int i = 0; i < 64; i++) same += memcmp(&hm[i], &hf[i], 8) == 0;
        printf("probe (A = c %s): %d / 64 lanes bit-identical to fma(c, b, cc)\n", diag ? "on lanes 0,5,10,15 of each 16" : "everywhere", same);
        if (diag && same != 64)
            for (int i = 0; i < 16; i++) printf("   lane %2d  mfma %.17g  fma %.17g\n", i, hm[i], hf[i]);
    }
    // 2. throughput
    float *out; unsigned long long *dclk;
    hipMalloc(&out, 1024); hipMalloc(&dclk, 64);
    for (int w : {1, 2, 4, 8}) {
        run("fma only", mix<16, 0>, 16, 0, out, dclk, w);
        run("mfma only", mix<0, 4>, 0, 4, out, dclk, w);
        run("16 fma + 1 mfma", mix<16, 1>, 16, 1, out, dclk, w);
        run("16 fma + 2 mfma", mix<16, 2>, 16, 2, out, dclk, w);
        run("16 fma + 4 mfma", mix<16, 4>, 16, 4, out, dclk, w);
        run("12 fma + 4 mfma", mix<12, 4>, 12, 4, out, dclk, w);
        run("8 fma + 4 mfma", mix<8, 4>, 8, 4, out, dclk, w);
    }
    return 0;
}
