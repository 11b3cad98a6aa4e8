// ubench_alu2.hip -- issue cost of the "light" instructions of the precise-math sweep (selects, compares, shifts,
// 2-operand f32 ops, permutes, 64-bit address adds) and of mixes with v_fma_f64, on gfx950.  Companion of ubench_alu.hip:
// there every f64 op / conversion came out at ~4.3 cycles per wave per SIMD; this one answers whether the light half
// of the kernel's 82 VALU instructions per cell is any cheaper, and whether ds_bpermute shares the VALU's issue slots.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_alu2.hip -o tools/ubench_alu2 && tools/ubench_alu2
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

#define REP16(X) _Pragma("unroll") for (int j = 0; j < 8; j++) { X } _Pragma("unroll") for (int j = 0; j < 8; j++) { X }

#define KERNEL(NAME, BODY)                                                                       \
    __global__ __launch_bounds__(256) void NAME(float *out, int iters, unsigned long long *clk)  \
    {                                                                                            \
        float a[8], b = 1.0001f + threadIdx.x * 1e-4f, c = 0.5f;                                 \
        int ia[8], ib = threadIdx.x * 4;                                                         \
        double d[8], db = 1.0001, dc = 0.5;                                                      \
        unsigned long long la[8];                                                                \
        for (int j = 0; j < 8; j++) {                                                            \
            a[j] = threadIdx.x * 1e-3f + 1.0f + j; ia[j] = threadIdx.x * 4 + j;                  \
            d[j] = threadIdx.x * 1e-3 + 1.0 + j; la[j] = (unsigned long long)out + j;            \
        }                                                                                        \
        int sr = 0;                                                                              \
        __shared__ float tab[1024];                                                              \
        for (int t = threadIdx.x; t < 1024; t += 256) tab[t] = t;                                \
        __syncthreads();                                                                         \
        typedef float f4 __attribute__((ext_vector_type(4)));                                    \
        f4 q[2] = {};                                                                            \
        const unsigned h = (threadIdx.x * 2654435761u) >> 16;                                    \
        int lds_a4 = (int)((h & 63u) * 4u), lds_a8 = (int)((h & 31u) * 8u), lds_a16 = (int)((h & 63u) * 16u); \
        asm volatile("" : "+v"(lds_a4), "+v"(lds_a8), "+v"(lds_a16));                            \
        unsigned long long smask = 0x3333333333333333ull ^ (unsigned long long)iters;            \
        asm volatile("s_mov_b32 vcc_lo, 0x55555555\n s_mov_b32 vcc_hi, 0x55555555" ::: "vcc"); \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                                    \
        unsigned long long r0 = __builtin_amdgcn_s_memrealtime();                                \
        for (int i = 0; i < iters; i++) {                                                        \
            REP16(BODY)                                                                          \
        }                                                                                        \
        asm volatile("s_waitcnt lgkmcnt(0)");                                                    \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                                    \
        unsigned long long r1 = __builtin_amdgcn_s_memrealtime();                                \
        {                                                                                        \
            double t = sr;                                                                       \
            for (int j = 0; j < 8; j++) t += a[j] + (float)ia[j] + d[j] + (double)la[j];         \
            t += q[0].x + q[1].y + tab[(int)a[0] & 1023];                                        \
            if (t == 123.456) out[0] = (float)t;                                                 \
        }                                                                                        \
        if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }         \
    }

#define CLOB : "vcc", "s20", "s21", "s22", "s23"
KERNEL(k_add_f32, asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[j]) : "v"(b));)
KERNEL(k_sub_f32, asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[j]) : "v"(b));)
KERNEL(k_max_f32, asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[j]) : "v"(b));)
KERNEL(k_and_b32, asm volatile("v_and_b32 %0, %0, %1" : "+v"(ia[j]) : "v"(ib));)
KERNEL(k_and_imm, asm volatile("v_and_b32 %0, 0xff800000, %0" : "+v"(ia[j]));)
KERNEL(k_lshl_b32, asm volatile("v_lshlrev_b32 %0, 2, %0" : "+v"(ia[j]));)
KERNEL(k_ashr_i32, asm volatile("v_ashrrev_i32 %0, 5, %0" : "+v"(ia[j]));)
KERNEL(k_add_u32, asm volatile("v_add_u32 %0, %0, %1" : "+v"(ia[j]) : "v"(ib));)
KERNEL(k_mov_b32, asm volatile("v_mov_b32 %0, %1" : "=v"(ia[j]) : "v"(ib));)
KERNEL(k_cnd_vcc, asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[j]) : "v"(b) CLOB);)
KERNEL(k_cnd_sgpr, asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[j]) : "v"(b), "s"(smask));)
KERNEL(k_cmp_vcc, asm volatile("v_cmp_eq_f32 vcc, %0, %1" : : "v"(a[j]), "v"(b) CLOB);)
KERNEL(k_cmp_sgpr, asm volatile("v_cmp_eq_f32_e64 s[22:23], %0, %1" : : "v"(a[j]), "v"(b) CLOB);)
KERNEL(k_cmp_cnd, asm volatile("v_cmp_eq_f32_e64 s[22:23], %0, %1\n v_cndmask_b32_e64 %0, %0, %1, s[22:23]" : "+v"(a[j]) : "v"(b) CLOB);)
KERNEL(k_bfe_i32, asm volatile("v_bfe_i32 %0, %0, 3, 1" : "+v"(ia[j]));)
KERNEL(k_bfi_b32, asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(ia[j]) : "v"(ib), "v"(c));)
KERNEL(k_lshl_add_u32, asm volatile("v_lshl_add_u32 %0, %0, 15, %1" : "+v"(ia[j]) : "v"(ib));)
KERNEL(k_and_or_b32, asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(ia[j]) : "v"(ib), "v"(c));)
KERNEL(k_pk_add_f32, asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(d[j]) : "v"(db));)
KERNEL(k_lshl_add_u64, asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(la[j]) : "v"(db));)
KERNEL(k_readfirstlane, asm volatile("v_readfirstlane_b32 s22, %0" : : "v"(ia[j]) CLOB);)
KERNEL(k_bpermute, asm volatile("ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(4)" : "+v"(ia[j]) : "v"(ib));)
KERNEL(k_fma_f64, asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[j]) : "v"(db), "v"(dc));)
// mixes: two instructions per repetition (cycles are per PAIR)
KERNEL(k_fma64_add32, asm volatile("v_fma_f64 %0, %0, %2, %3\n v_add_f32 %1, %1, %4" : "+v"(d[j]), "+v"(a[j]) : "v"(db), "v"(dc), "v"(b));)
KERNEL(k_fma64_cnd, asm volatile("v_fma_f64 %0, %0, %2, %3\n v_cndmask_b32_e64 %1, %1, %4, %5" : "+v"(d[j]), "+v"(a[j]) : "v"(db), "v"(dc), "v"(b), "s"(smask));)
KERNEL(k_fma64_and, asm volatile("v_fma_f64 %0, %0, %2, %3\n v_and_b32 %1, %1, %4" : "+v"(d[j]), "+v"(ia[j]) : "v"(db), "v"(dc), "v"(ib));)
KERNEL(k_fma64_bperm, asm volatile("v_fma_f64 %0, %0, %2, %3\n ds_bpermute_b32 %1, %4, %1\n s_waitcnt lgkmcnt(4)" : "+v"(d[j]), "+v"(ia[j]) : "v"(db), "v"(dc), "v"(ib));)
KERNEL(k_fma64_x3_bperm, asm volatile("v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %0, %0, %2, %3\n ds_bpermute_b32 %1, %4, %1\n s_waitcnt lgkmcnt(4)" : "+v"(d[j]), "+v"(ia[j]) : "v"(db), "v"(dc), "v"(ib));)
KERNEL(k_add32_and, asm volatile("v_add_f32 %0, %0, %2\n v_and_b32 %1, %1, %3" : "+v"(a[j]), "+v"(ia[j]) : "v"(b), "v"(ib));)

KERNEL(k_cmp_cnd_vcc, asm volatile("v_cmp_eq_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[j]) : "v"(b) CLOB);)
KERNEL(k_cmp_x4cnd_vcc, asm volatile("v_cmp_eq_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[j]) : "v"(b) CLOB);)
KERNEL(k_max3_f32, asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[j]) : "v"(b), "v"(c));)
KERNEL(k_min_f32, asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[j]) : "v"(b));)
KERNEL(k_mul_f32, asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[j]) : "v"(b));)
KERNEL(k_lshr_b32, asm volatile("v_lshrrev_b32 %0, 17, %0" : "+v"(ia[j]));)
KERNEL(k_xor_b32, asm volatile("v_xor_b32 %0, %0, %1" : "+v"(ia[j]) : "v"(ib));)
KERNEL(k_or_b32, asm volatile("v_or_b32 %0, %0, %1" : "+v"(ia[j]) : "v"(ib));)
KERNEL(k_sub_u32, asm volatile("v_sub_u32 %0, %0, %1" : "+v"(ia[j]) : "v"(ib));)
KERNEL(k_cvt_f64_f32, asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[j]) : "v"(a[j]));)
KERNEL(k_mul_lo_u32, asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(ia[j]) : "v"(ib));)
KERNEL(k_lds_b32, asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(4)" : "=v"(ia[j]) : "v"(lds_a4));)
KERNEL(k_lds_b64, asm volatile("ds_read_b64 %0, %1\n s_waitcnt lgkmcnt(4)" : "=v"(d[j]) : "v"(lds_a8));)
KERNEL(k_lds_b128, asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(4)" : "=v"(q[j & 1]) : "v"(lds_a16));)
KERNEL(k_fma64_x3_lds64, asm volatile("v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %0, %0, %2, %3\n ds_read_b64 %1, %4\n s_waitcnt lgkmcnt(4)" : "+v"(d[j]), "=v"(la[j]) : "v"(db), "v"(dc), "v"(lds_a8));)

// VCC written once by a VALU compare before the loop (the prologue's s_mov is overwritten in the first repetition only)
KERNEL(k_cnd_vcc_valu, if (i == 0 && j == 0) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(c), "v"(b) : "vcc"); asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[j]) : "v"(b));)
KERNEL(k_cnd_vcc_salu, asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[j]) : "v"(b));)
KERNEL(k_cnd_sgpr_chain4, asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2\n v_cndmask_b32_e64 %0, %0, %1, %2\n v_cndmask_b32_e64 %0, %0, %1, %2\n v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[j]) : "v"(b), "s"(smask));)
KERNEL(k_cnd_vcc_chain4, asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[j]) : "v"(b));)
KERNEL(k_add_chain4, asm volatile("v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1" : "+v"(a[j]) : "v"(b));)
KERNEL(k_fma64_chain4, asm volatile("v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2" : "+v"(d[j]) : "v"(db), "v"(dc));)
KERNEL(k_cmp_sgpr_x4cnd, asm volatile("v_cmp_eq_f32_e64 s[22:23], %0, %1\n v_cndmask_b32_e64 %0, %0, %1, s[22:23]\n v_cndmask_b32_e64 %0, %0, %1, s[22:23]\n v_cndmask_b32_e64 %0, %0, %1, s[22:23]\n v_cndmask_b32_e64 %0, %0, %1, s[22:23]" : "+v"(a[j]) : "v"(b) CLOB);)

typedef void (*kern_t)(float *, int, unsigned long long *);

static void run(const char *name, kern_t k, float *out, unsigned long long *dclk, int waves_per_simd, int per_rep)
{
    const int iters = 2000;
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
    double ghz = (double)clk[0] / ((double)clk[1] * 10.0);
    double reps_per_simd = (double)iters * 16 * waves_per_simd;
    double cyc = (double)ms * 1e-3 * ghz * 1e9 / reps_per_simd;
    printf("  %-36s waves/SIMD %d: %6.2f cyc per repetition (%d instr) per SIMD  (kernel %.3f ms, clock %.2f GHz)\n", name,
           waves_per_simd, cyc, per_rep, ms, ghz);
    hipEventDestroy(e0); hipEventDestroy(e1);
}

int main()
{
    float *out; unsigned long long *dclk;
    hipMalloc(&out, 1024); hipMalloc(&dclk, 64);
    struct { const char *n; kern_t k; int per_rep; } ks[] = {
        {"v_add_f32", k_add_f32, 1}, {"v_sub_f32", k_sub_f32, 1}, {"v_max_f32", k_max_f32, 1}, {"v_and_b32", k_and_b32, 1},
        {"v_and_b32 imm32", k_and_imm, 1}, {"v_lshlrev_b32", k_lshl_b32, 1}, {"v_ashrrev_i32", k_ashr_i32, 1},
        {"v_add_u32", k_add_u32, 1}, {"v_mov_b32", k_mov_b32, 1}, {"v_cndmask vcc", k_cnd_vcc, 1},
        {"v_cndmask sgpr", k_cnd_sgpr, 1}, {"v_cmp->vcc", k_cmp_vcc, 1}, {"v_cmp->sgpr", k_cmp_sgpr, 1},
        {"v_cmp+v_cndmask", k_cmp_cnd, 2}, {"v_bfe_i32", k_bfe_i32, 1}, {"v_bfi_b32", k_bfi_b32, 1},
        {"v_lshl_add_u32", k_lshl_add_u32, 1}, {"v_and_or_b32", k_and_or_b32, 1}, {"v_pk_add_f32", k_pk_add_f32, 1},
        {"v_lshl_add_u64", k_lshl_add_u64, 1}, {"v_readfirstlane", k_readfirstlane, 1}, {"ds_bpermute_b32", k_bpermute, 1},
        {"v_fma_f64", k_fma_f64, 1}, {"fma64+add32", k_fma64_add32, 2}, {"fma64+cndmask", k_fma64_cnd, 2},
        {"fma64+and", k_fma64_and, 2}, {"fma64+bpermute", k_fma64_bperm, 2}, {"3 fma64+bpermute", k_fma64_x3_bperm, 4},
        {"add32+and", k_add32_and, 2},
        {"v_cmp vcc+v_cndmask vcc", k_cmp_cnd_vcc, 2}, {"v_cmp vcc+4 cndmask vcc", k_cmp_x4cnd_vcc, 5},
        {"v_cndmask vcc (VALU-written once)", k_cnd_vcc_valu, 1}, {"v_cndmask vcc (SALU-written once)", k_cnd_vcc_salu, 1},
        {"4 dependent v_cndmask sgpr", k_cnd_sgpr_chain4, 4}, {"4 dependent v_cndmask vcc", k_cnd_vcc_chain4, 4},
        {"4 dependent v_add_f32", k_add_chain4, 4}, {"4 dependent v_fma_f64", k_fma64_chain4, 4},
        {"v_cmp sgpr+4 cndmask sgpr", k_cmp_sgpr_x4cnd, 5},
        {"v_max3_f32", k_max3_f32, 1}, {"v_min_f32", k_min_f32, 1}, {"v_mul_f32", k_mul_f32, 1},
        {"v_lshrrev_b32", k_lshr_b32, 1}, {"v_xor_b32", k_xor_b32, 1}, {"v_or_b32", k_or_b32, 1}, {"v_sub_u32", k_sub_u32, 1},
        {"v_cvt_f64_f32", k_cvt_f64_f32, 1}, {"v_mul_u32_u24", k_mul_lo_u32, 1},
        {"ds_read_b32 (64 random of 64 words)", k_lds_b32, 1}, {"ds_read_b64 (random of 32 entries)", k_lds_b64, 1},
        {"ds_read_b128 (random of 64 entries)", k_lds_b128, 1}, {"3 fma64+ds_read_b64", k_fma64_x3_lds64, 4},
    };
    for (auto &e : ks) { run(e.n, e.k, out, dclk, 2, e.per_rep); run(e.n, e.k, out, dclk, 8, e.per_rep); }
    return 0;
}
