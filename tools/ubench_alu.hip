// ubench_alu.hip -- throughput of the VALU instructions the precise-math kernel is built from (gfx950).
// Each kernel issues ITER x 16 independent copies of one instruction per wave; WAVES waves per SIMD keep the pipe
// full.  Reported: cycles per wave-instruction per SIMD at the measured clock (s_memtime / s_memrealtime).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_alu.hip -o tools/ubench_alu && tools/ubench_alu
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

#define REP16(X) _Pragma("unroll") for (int j = 0; j < 8; j++) { X } _Pragma("unroll") for (int j = 0; j < 8; j++) { X }

#define KERNEL(NAME, DECL, BODY, SINK)                                                          \
    __global__ __launch_bounds__(256) void NAME(float *out, int iters, unsigned long long *clk) \
    {                                                                                            \
        DECL;                                                                                    \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                                    \
        unsigned long long r0 = __builtin_amdgcn_s_memrealtime();                                \
        for (int i = 0; i < iters; i++) {                                                        \
            REP16(BODY)                                                                          \
        }                                                                                        \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                                    \
        unsigned long long r1 = __builtin_amdgcn_s_memrealtime();                                \
        SINK;                                                                                    \
        if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }         \
    }

#define F32DECL float a[8], b = 1.0001f, c = 0.5f; int ia[8]; for (int j = 0; j < 8; j++) { a[j] = threadIdx.x * 1e-3f + 1.0f + j; ia[j] = threadIdx.x + j; }
#define F64DECL double a[8], b = 1.0001, c = 0.5; float fa[8]; int ia[8]; for (int j = 0; j < 8; j++) { a[j] = threadIdx.x * 1e-3 + 1.0 + j; fa[j] = threadIdx.x * 0.01f + j; ia[j] = threadIdx.x + j; }
#define SINK32 { float t = 0; for (int j = 0; j < 8; j++) t += a[j] + (float)ia[j]; if (t == 123.456f) out[0] = t; }
#define SINK64 { double t = 0; for (int j = 0; j < 8; j++) t += a[j] + fa[j] + ia[j]; if (t == 123.456) out[0] = (float)t; }

KERNEL(k_fma_f32, F32DECL, asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[j]) : "v"(b), "v"(c));, SINK32)
KERNEL(k_mul_f32, F32DECL, asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[j]) : "v"(b));, SINK32)
KERNEL(k_exp_f32, F32DECL, asm volatile("v_exp_f32 %0, %0" : "+v"(a[j]));, SINK32)
KERNEL(k_log_f32, F32DECL, asm volatile("v_log_f32 %0, %0" : "+v"(a[j]));, SINK32)
KERNEL(k_rcp_f32, F32DECL, asm volatile("v_rcp_f32 %0, %0" : "+v"(a[j]));, SINK32)
KERNEL(k_rndne_f32, F32DECL, asm volatile("v_rndne_f32 %0, %0" : "+v"(a[j]));, SINK32)
KERNEL(k_ldexp_f32, F32DECL, asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(a[j]) : "v"(ia[j]));, SINK32)
KERNEL(k_cvt_i32_f32, F32DECL, asm volatile("v_cvt_i32_f32 %0, %1" : "=v"(ia[j]) : "v"(a[j]));, SINK32)
KERNEL(k_cndmask, F32DECL, asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[j]) : "v"(b));, SINK32)
KERNEL(k_max3_f32, F32DECL, asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[j]) : "v"(b), "v"(c));, SINK32)
KERNEL(k_mov_dpp, F32DECL, asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[j]) : "v"(b));, SINK32)
KERNEL(k_fma_f64, F64DECL, asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[j]) : "v"(b), "v"(c));, SINK64)
KERNEL(k_mul_f64, F64DECL, asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[j]) : "v"(b));, SINK64)
KERNEL(k_add_f64, F64DECL, asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[j]) : "v"(b));, SINK64)
KERNEL(k_rndne_f64, F64DECL, asm volatile("v_rndne_f64 %0, %0" : "+v"(a[j]));, SINK64)
KERNEL(k_cvt_f64_f32, F64DECL, asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a[j]) : "v"(fa[j]));, SINK64)
KERNEL(k_cvt_f32_f64, F64DECL, asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(fa[j]) : "v"(a[j]));, SINK64)
KERNEL(k_cvt_i32_f64, F64DECL, asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(ia[j]) : "v"(a[j]));, SINK64)
KERNEL(k_cvt_f64_i32, F64DECL, asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(a[j]) : "v"(ia[j]));, SINK64)
KERNEL(k_ldexp_f64, F64DECL, asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(a[j]) : "v"(ia[j]));, SINK64)
KERNEL(k_pk_fma_f32, F64DECL, asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[j]) : "v"(b), "v"(c));, SINK64)
KERNEL(k_pk_mul_f32, F64DECL, asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[j]) : "v"(b));, SINK64)

typedef void (*kern_t)(float *, int, unsigned long long *);

static void run(const char *name, kern_t k, float *out, unsigned long long *dclk, int waves_per_simd)
{
    const int iters = 4000;
    const int blocks = 256 * waves_per_simd;  // 256 threads = 4 waves = one per SIMD; x waves_per_simd blocks per CU
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, iters, dclk);  // warm
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, iters, dclk);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long clk[2];
    hipMemcpy(clk, dclk, 16, hipMemcpyDeviceToHost);
    double ghz = (double)clk[0] / ((double)clk[1] * 10.0);  // memrealtime ticks at 100 MHz
    double instr_per_simd = (double)iters * 16 * waves_per_simd;
    double cyc_wave0 = (double)ms * 1e-3 * ghz * 1e9 / instr_per_simd;  // wall time x measured clock (wave 0 is favoured by age arbitration, so its own clock reads low)
    printf("  %-14s waves/SIMD %d: %6.2f cyc/instr/SIMD  (kernel %.3f ms, clock %.2f GHz)\n", name, waves_per_simd, cyc_wave0, ms, ghz);
}

int main()
{
    float *out; unsigned long long *dclk;
    hipMalloc(&out, 1024); hipMalloc(&dclk, 64);
    struct { const char *n; kern_t k; } ks[] = {
        {"v_fma_f32", k_fma_f32}, {"v_mul_f32", k_mul_f32}, {"v_pk_fma_f32", k_pk_fma_f32}, {"v_pk_mul_f32", k_pk_mul_f32},
        {"v_exp_f32", k_exp_f32}, {"v_log_f32", k_log_f32}, {"v_rcp_f32", k_rcp_f32}, {"v_rndne_f32", k_rndne_f32},
        {"v_ldexp_f32", k_ldexp_f32}, {"v_cvt_i32_f32", k_cvt_i32_f32}, {"v_cndmask_b32", k_cndmask}, {"v_max3_f32", k_max3_f32},
        {"v_mov_dpp", k_mov_dpp},
        {"v_fma_f64", k_fma_f64}, {"v_mul_f64", k_mul_f64}, {"v_add_f64", k_add_f64}, {"v_rndne_f64", k_rndne_f64},
        {"v_cvt_f64_f32", k_cvt_f64_f32}, {"v_cvt_f32_f64", k_cvt_f32_f64}, {"v_cvt_i32_f64", k_cvt_i32_f64},
        {"v_cvt_f64_i32", k_cvt_f64_i32}, {"v_ldexp_f64", k_ldexp_f64},
    };
    for (auto &e : ks) { run(e.n, e.k, out, dclk, 1); run(e.n, e.k, out, dclk, 2); run(e.n, e.k, out, dclk, 4); run(e.n, e.k, out, dclk, 8); }
    return 0;
}
