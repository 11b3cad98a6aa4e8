// cell_update.h -- the per-cell log-sum-exp update, device side (gfx950 only).
//
// Mirrors the ROUNDING SEQUENCE of the reference CPU solver
// (libepic/src/harmonic/harmonic_cpu.cpp:60-70 for 2-D, :110-123 for 3-D), because the
// f32 stagnation point the solver converges to is decided by it:
//     mx = max(neighbours)                                   f32
//     s  = ((e(a-mx) + e(b-mx)) + e(c-mx)) + e(d-mx) ...     f32, left-associated
//     t  = mx + ln(s)                                        f32
//     u  = (float)((double)t - ln(2n))                       f64 subtract, one rounding
//
// Implementations of e() and ln(), chosen per kernel instantiation (template MATH):
//
//  kMathPrecise (default)  The reference calls libm's expf/logf (std::exp/std::log on float).  libm is a
//      third-party dependency that is not under /root/reference: GNU libc 2.35 (Ubuntu 22.04 image), whose
//      expf/logf are Szabolcs Nagy's table+polynomial routines evaluated in double precision
//      (glibc sysdeps/ieee754/flt-32/e_expf.c, e_logf.c; tables e_exp2f_data.c, e_logf_data.c; published in
//      ARM optimized-routines).  They are restated here with the hardware's f64 units (tables in LDS).  Result: <= 0.502 ulp and bit-identical to the host libm over the whole input ranges -- every
//      float in [-104, -0] for exp, every float in [1, 6] for log (tests/test_gpu_libm.py, 1.14e9 inputs).
//      Why it is the default: the relaxation amplifies any SYSTEMATIC error of the update by ~R^2 (R = domain
//      radius in cells, 1e4..1e6); v_log_f32 is biased by -0.1..-0.4 ulp and v_exp_f32 by -0.1 ulp near 1
//      (tools/probe_transcendentals.hip, profiles/r01_probe_transcendentals.txt), which moved the converged field
//      of the 256x256 reference map by 2e-5 (relative) -- outside the 1e-5 parity bar.
//
//  kMathFast  v_exp_f32 / v_log_f32 with f32 base changes.  ~2.5x less ALU; for well-conditioned maps only.
//
//  kMathTol   (further down: "tol math") does not evaluate e() per neighbour at all: every cell's potential is split once into
//      e^u = q 2^n and the cells it is a neighbour of share the pair; its logarithm is its own (TolLn: 256 intervals per
//      binade, exact f32 reduction, table in LDS).  Same rounding stages as above; parity by tolerance, not by bits.
//
// Compile with -ffp-contract=off so no step is fused behind our back (explicit fma() where wanted).
#pragma once
#include <utility>
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace epic_hip {

constexpr float kLog2e = 1.44269504088896340736f;
constexpr float kLn2 = 0.69314718055994530942f;
constexpr double kLn4 = 1.38629436111989061883;  // log(2.0 * 2)
constexpr double kLn6 = 1.79175946922805500081;  // log(2.0 * 3)

__device__ __forceinline__ float hw_exp(float x)  // e^x, x <= 0
{
    return __builtin_amdgcn_exp2f(x * kLog2e);
}

__device__ __forceinline__ float hw_ln(float s)  // ln(s), s in [1, 6]
{
    return __builtin_amdgcn_logf(s) * kLn2;
}

__device__ __forceinline__ float max2(float a, float b) { return __builtin_fmaxf(a, b); }

constexpr int kMathPrecise = 0;
constexpr int kMathFast = 1;
constexpr int kMathTraffic = 2;  // diagnostic: neighbours averaged with 3 adds -- same loads/stores, almost no ALU
// (3 was round 1's df32 mode: removed)
constexpr int kMathTol = 4;      // one exp-class evaluation per cell (shared by its neighbours) + one f64 log: the tolerance mode, see below

// ---- libm-equivalent expf / logf in f64 (glibc 2.35 algorithm, see header) --------------------------------

// glibc's tab[i] = bits(2^(i/32)) - (i << 47) (e_exp2f_data.c, EXP2F_TABLE_BITS = 5); the kernel wants the plain
// 2^(i/32), so math_tables_load() adds the i << 47 back.
__constant__ const uint64_t kExpTab[32] = {
    0x3ff0000000000000, 0x3fefd9b0d3158574, 0x3fefb5586cf9890f, 0x3fef9301d0125b51, 0x3fef72b83c7d517b,
    0x3fef54873168b9aa, 0x3fef387a6e756238, 0x3fef1e9df51fdee1, 0x3fef06fe0a31b715, 0x3feef1a7373aa9cb,
    0x3feedea64c123422, 0x3feece086061892d, 0x3feebfdad5362a27, 0x3feeb42b569d4f82, 0x3feeab07dd485429,
    0x3feea47eb03a5585, 0x3feea09e667f3bcd, 0x3fee9f75e8ec5f74, 0x3feea11473eb0187, 0x3feea589994cce13,
    0x3feeace5422aa0db, 0x3feeb737b0cdc5e5, 0x3feec49182a3f090, 0x3feed503b23e255d, 0x3feee89f995ad3ad,
    0x3feeff76f2fb5e47, 0x3fef199bdd85529c, 0x3fef3720dcef9069, 0x3fef5818dcfba487, 0x3fef7c97337b9b5f,
    0x3fefa4afa2a490da, 0x3fefd0765b6e4540,
};
// {invc, logc} per sub-interval of [OFF, 2 OFF)   (glibc e_logf_data.c, LOGF_TABLE_BITS = 4)
__constant__ const double kLogTab[32] = {
    0x1.661ec79f8f3bep+0, -0x1.57bf7808caadep-2, 0x1.571ed4aaf883dp+0, -0x1.2bef0a7c06ddbp-2,
    0x1.49539f0f010bp+0,  -0x1.01eae7f513a67p-2, 0x1.3c995b0b80385p+0, -0x1.b31d8a68224e9p-3,
    0x1.30d190c8864a5p+0, -0x1.6574f0ac07758p-3, 0x1.25e227b0b8eap+0,  -0x1.1aa2bc79c81p-3,
    0x1.1bb4a4a1a343fp+0, -0x1.a4e76ce8c0e5ep-4, 0x1.12358f08ae5bap+0, -0x1.1973c5a611cccp-4,
    0x1.0953f419900a7p+0, -0x1.252f438e10c1ep-5, 0x1p+0,               0x0p+0,
    0x1.e608cfd9a47acp-1, 0x1.aa5aa5df25984p-5,  0x1.ca4b31f026aap-1,  0x1.c5e53aa362eb4p-4,
    0x1.b2036576afce6p-1, 0x1.526e57720db08p-3,  0x1.9c2d163a1aa2dp-1, 0x1.bc2860d22477p-3,
    0x1.886e6037841edp-1, 0x1.1058bc8a07ee1p-2,  0x1.767dcf5534862p-1, 0x1.4043057b6ee09p-2,
};

// Where the tables live: in LDS, both.  log: 64 entries of 16 bytes read with one ds_read_b128.  exp: the 32 entries
// 2^(j/32) replicated 8 times (256 doubles), so that the byte offset of entry k mod 32 is (k & 0xff) << 3 -- ONE
// instruction with sub-dword addressing (v_lshlrev_b32_sdwa ... src1_sel:BYTE_0) -- read with one ds_read_b64.
// Earlier forms kept the tables in registers, one entry per lane, fetched with ds_bpermute_b32: that costs 24 cycles per
// wave per SIMD on the LDS crossbar against 8.4 for a ds_read_b64 and 16 for a ds_read_b128 (tools/ubench_alu2.hip), and
// the crossbar was the co-limiter of the kernel next to the VALU (8192^2 sweep: 155.8 us with the exp table on the
// crossbar, 152.9 with a 32-entry LDS table addressed by and + shift, 149.7 with the replicated table; same box).
// The log table is expanded over the binade index k = 0..3 (arguments in [0.7, 11.2) cover the sums of 4 or 6 terms
// <= 1 with one term == 1): entry 16 k + i = {invc_i 2^-k, logc_i + k ln2}, so the kernel needs neither k, nor a
// multiply, nor the normalised argument z = s 2^-k (s invc_i 2^-k is the same real number as z invc_i, so the fma that
// forms r rounds alike).
struct MathTab {
    const double2 *ln;   // LDS: {invc, logc + k ln2} x 64
    const double *ex;    // LDS: 2^((j & 31)/32), j = 0..255
    // Round 5: the f64 constants of the two routines, and where two of them live.  A VOP3 instruction reads at most one scalar operand, so
    // the first fma of each Horner chain (coefficient AND addend constant) takes its addend from a vector register pair.
    //   kConstsPlain  literals, the compiler's choice: as rounds 1-4.
    //   kConstsKeep   those two addends (e2, c0) are held in register pairs the compiler cannot see through and the fma is issued as the
    //                 three-address v_fma_f64 (fma_keep_addend).  In the kernels with scalar branches in their row loop -- the red-black
    //                 ones -- the compiler otherwise picks the two-address v_fmac_f64 and COPIES the addend first, every time (v_mov_b64 +
    //                 v_fmac_f64: 1.6 moves per cell; 1.2-4 % of their VALU instructions, and with the two SGPR pairs gone the tracked pair
    //                 kernel spills fewer scalars into VGPR lanes: 341 -> 281 v_readlane / v_writelane per 40 cells).  The plain Jacobi
    //                 kernels do not make the copies and the list-driven single sweeps are at the SGPR limit, where the two scalar pairs the
    //                 asm operands pin cost more in spills than the moves do: they stay plain.
    // `consts` is a compile-time constant after inlining.  (All nine constants in vector registers -- 18 VGPRs, 341 -> 203 lane moves,
    // -5 % VALU instructions in the tracked pair kernel, occupancy 7 -> 5 -- measured the same as kConstsKeep on whole relaxations,
    // 2.31 against 2.33 s plain; not kept.  profiles/r05_experiments.txt item 12.)
    double inv_ln2n, shift, e3, e2, e1, a1, c0, a0, ln4;
    int consts;
};
constexpr int kConstsPlain = 0, kConstsKeep = 1;
constexpr int kLnTabEntries = 64;
constexpr int kExTabEntries = 256;
constexpr int kMathLdsDoubles = 2 * kLnTabEntries + kExTabEntries;   // 3 KiB per workgroup

// Staging is split in two so that a kernel can put its first row loads between them: math_tables_fetch() starts the
// loads of this lane's table entries from constant memory, math_tables_commit() writes them to LDS.  EVERY WAVE WRITES
// THE WHOLE TABLE (lane L: exp entries L, L + 64, L + 128, L + 192 -- one value, the table is periodic in 32 -- and
// log entry L): the waves of a workgroup write identical bytes, a wave's own LDS instructions execute in order, so a
// wave may read as soon as it has written and NO workgroup barrier is needed -- which also means a wave that finds no
// work may leave before its siblings get here.  `lds` = kMathLdsDoubles doubles of LDS, 16-byte aligned.
struct MathTabRegs {
    uint64_t ex;
    double invc, y0;
};
__device__ __forceinline__ MathTabRegs math_tables_fetch()
{
    const int lane = threadIdx.x & 63, k = lane >> 4, i = lane & 15;
    MathTabRegs r;
    r.ex = kExpTab[lane & 31] + ((uint64_t)(lane & 31) << 47);
    r.invc = __builtin_ldexp(kLogTab[2 * i], -k);  // exact: a power of two
    r.y0 = kLogTab[2 * i + 1] + (double)k * 0x1.62e42fefa39efp-1;
    return r;
}
// Where the two tables sit inside `lds`: pure address arithmetic on the kernel's __shared__ array, so that the LDS
// offsets of the lookups are compile-time constants (keep it out of loop-carried variables: a table pointer that went
// through the task loop of the list-driven kernels cost one v_add_u32 per lookup).
#ifndef EPIC_PRECISE_VGPR_CONSTS
#define EPIC_PRECISE_VGPR_CONSTS 1   // build knob (A/B): 0 = every kernel kConstsPlain, as rounds 1-4
#endif
__device__ __forceinline__ double fma_keep_addend(int consts, double c, double r, double addend)
{
    if (consts == kConstsKeep) {
        double y;
        asm("v_fma_f64 %0, %1, %2, %3" : "=v"(y) : "s"(c), "v"(r), "v"(addend));
        return y;
    }
    return __builtin_fma(c, r, addend);
}
__device__ __forceinline__ MathTab math_tables_at(double *lds, int consts = kConstsPlain)
{
    if (!EPIC_PRECISE_VGPR_CONSTS) consts = kConstsPlain;
    // glibc e_expf.c: N / ln2, the shift, C[0..2] / N^(3..1);  e_logf.c: A[1], A[2], A[0];  harmonic_cpu.cpp:70: log(2.0 * n)
    double inv_ln2n = 0x1.71547652b82fep+5, shift = 0x1.8p+52, e3 = 0x1.c6af84b912394p-20, e2 = 0x1.ebfce50fac4f3p-13,
           e1 = 0x1.62e42ff0c52d6p-6, a1 = 0x1.5575b0be00b6ap-2, c0 = -0x1.ffffef20a4123p-2, a0 = -0x1.00ea348b88334p-2, ln4 = kLn4;
    if (consts == kConstsKeep) asm volatile("" : "+v"(e2), "+v"(c0));
    return MathTab{reinterpret_cast<double2 *>(lds), lds + 2 * kLnTabEntries, inv_ln2n, shift, e3, e2, e1, a1, c0, a0, ln4, consts};
}
__device__ __forceinline__ void math_tables_commit(const MathTabRegs &r, double *lds)
{
    const int lane = threadIdx.x & 63;
    MathTab t{};   // (addresses only)
    t.ln = reinterpret_cast<double2 *>(lds);
    t.ex = lds + 2 * kLnTabEntries;
    static_assert(kExTabEntries == 256 && kLnTabEntries == 64, "four exp entries and one log entry per lane");
#pragma unroll
    for (int j = 0; j < kExTabEntries; j += 64) reinterpret_cast<uint64_t *>(const_cast<double *>(t.ex))[lane + j] = r.ex;
    const_cast<double2 *>(t.ln)[lane] = double2{r.invc, r.y0};
}
__device__ __forceinline__ MathTab math_tables_load(double *lds, int consts = kConstsPlain)
{
    math_tables_commit(math_tables_fetch(), lds);
    return math_tables_at(lds, consts);
}

// e^x for x <= 0.  glibc e_expf.c: z = x N/ln2, k = round(z), r = z - k, s = 2^(k/N) from the table, cubic in r,
// all in double, one rounding to float at the end.  Differences to the C source that do not change the float result
// for any input the sweeps can produce (the double result carries ~2^-30 of slack; tests/test_gpu_libm.py compares
// every float in [-104, -0] with the host libm): k is taken from the
// low word of fma(x, N/ln2, 1.5 * 2^52) (glibc's own non-intrinsic path does the same with an add), r comes from a
// second fma instead of a rounded product, the cubic is in Horner form.
__device__ __forceinline__ float precise_exp(float x, const MathTab &tab)
{
    const double xd = (double)x;
    const double ks = __builtin_fma(xd, tab.inv_ln2n, tab.shift);
    const int ki = (int)(uint32_t)__builtin_bit_cast(uint64_t, ks);  // low word: k in two's complement
    const double kd = ks - tab.shift;
    const double r = __builtin_fma(xd, tab.inv_ln2n, -kd);
    // glibc forms s = 2^(k/N) by adding k << 47 to the table word; 2^((k mod N)/N) scaled by ldexp is the same number
    // and, unlike the integer add, degrades to 0 for the x = -1e6 terms (neighbours that are obstacles) without a clamp.
    int off;  // (k & 0xff) << 3: byte 0 of k, shifted, in one instruction
    asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(off) : "v"(3), "v"(ki));
    const double s0 = *reinterpret_cast<const double *>(reinterpret_cast<const char *>(tab.ex) + off);
    double y = fma_keep_addend(tab.consts, tab.e3, r, tab.e2);
    y = __builtin_fma(y, r, tab.e1);
    y = __builtin_fma(y, r, 1.0);
    return (float)__builtin_ldexp(y * s0, ki >> 5);
}

// ln(s) for s in [0.7, 11.2).  glibc e_logf.c: s = 2^k z with z in [OFF, 2 OFF) split into 16 sub-intervals,
// r = z invc - 1, ln s = log1p(r) + logc + k ln2 with a cubic for log1p, all in double.
__device__ __forceinline__ double precise_ln_d(float sf, const MathTab &tab)  // the value before glibc's final rounding
{
    const uint32_t tmp = __builtin_bit_cast(uint32_t, sf) - 0x3f330000u;
    // table entry 16 k + i = bits 24..19 of tmp; its byte offset is that times 16
    const double2 ent = *reinterpret_cast<const double2 *>(reinterpret_cast<const char *>(tab.ln) + ((tmp >> 15) & 0x3f0u));
    const double invc = ent.x, y0 = ent.y;
    const double r = __builtin_fma((double)sf, invc, -1.0);  // = z invc_i - 1 with z = s 2^-k, see MathTab
    const double r2 = r * r;
    double y = fma_keep_addend(tab.consts, tab.a1, r, tab.c0);
    y = __builtin_fma(tab.a0, r2, y);
    return __builtin_fma(y, r2, y0 + r);
}
__device__ __forceinline__ float precise_ln(float sf, const MathTab &tab) { return (float)precise_ln_d(sf, tab); }

// ---- packed f32 (v_pk_fma_f32 / v_pk_add_f32: two lanes' worth per instruction at the issue cost of one f64 op) --------
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f pk_fma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
// by-value helpers: __builtin_bit_cast applied directly to a vector ELEMENT expression (v.y) reads element 0
__device__ __forceinline__ uint32_t f2u(float f) { return __builtin_bit_cast(uint32_t, f); }
__device__ __forceinline__ float u2f(uint32_t u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ v2f splat(float x) { return v2f{x, x}; }
// (Round 1's `df32` mode -- packed double-float exp / log per neighbour, <= 0.53 ulp, 99.8 % agreement with libm -- lived
// here; it was no faster than the reworked precise kernel, could not stop by the reference's test under Jacobi on umass,
// and is superseded by the tol mode below.  Removed in round 2; DESIGN.md section 2 keeps what was learned from it.)

// ---- kMathTol: the logarithm of the sum ---------------------------------------------------------------------------
// ln S for the sum S of 2n terms <= 1.4143 of which the largest is >= 0.7071 (S in [0.5, 8) in 2-D, [0.5, 16) in 3-D).
// The tol mode makes no claim on glibc's bits, so its logarithm is built for this chip's instruction prices rather than
// copied from logf (round 2 used glibc's algorithm in f64: 8 f64-class instructions):
//     ln S = lnc + log1p(r),   r = S invc - 1,
// over 256 sub-intervals per binade picked by the exponent's low bits and the top 8 mantissa bits of S.  invc = v / 512
// 2^-k with a 9-bit integer v (the rounded reciprocal of the sub-interval's centre), so S invc - 1 has at most 24
// significant bits and ONE f32 fma gives r EXACTLY (|r| <= 2^-8.45); log1p(r) = r + r^2 (-1/2 + r / 3) needs nothing beyond
// f32 (three cheap instructions), and only lnc -- the table's -ln(invc), correctly rounded double -- is added in f64.
// Measured against 200-bit arithmetic over [0.707, 8.49] (tools/gen_ln_table.py): max error 2^-32.9, rms 2^-35.0, mean
// +1e-12; the glibc algorithm in f64 it replaces: max 2^-28.7, rms 2^-32.1, mean -8e-12.  Cost: 2 integer + 3 f32 + 2
// f64-class instructions against 3 + 0 + 8.  The table is 16 bytes per entry, 16 KiB (2-D: four binades, the entry of
// binade k at ((k & 3) << 8) + j so that the raw bits of S index it) or 20 KiB (3-D: five binades in order); every
// workgroup copies it from constant memory into LDS once (tol_ln_stage: all its waves, one barrier), which is why the
// tol kernels are launched as resident workgroups that walk their tasks (kernels_2d.hip, kernels_3d.hip).
struct alignas(16) TolLnEntry { double lnc; float invc; float pad; };
__constant__ const TolLnEntry kTolLnTab[5 * 256] = {
#include "tol_ln_table.inc"
};
// Layout in LDS.  0: the entries as they are (16 B each; one ds_read_b96 per lookup).  1 (round 5): two arrays of 8-byte slots behind one
// address -- lnc of entry i at byte 8 i, invc at byte 8 (entries + i) -- read by ds_read_b64 + ds_read_b32 offset:8 * entries: the same bytes in
// the same 16 B per entry, but 2 + 2 LDS-array cycles per lookup instead of ds_read_b96's 8 (MI355X_MICROARCH.md, LDS table), random
// entries spread over 32 and 16 bank slots instead of 8, and invc arrives in a register of its own, so the two cells of a pair get theirs
// side by side without a move (6 fewer VALU instructions per step of the fused pass).
#ifndef EPIC_TOL_SPLIT_READS
#define EPIC_TOL_SPLIT_READS 1
#endif
template <int BINADES> struct TolLn {
    static_assert(BINADES == 4 || BINADES == 5, "2-D sums stay below 8, 3-D sums below 16");
    static constexpr int kEntries = BINADES * 256;
    static constexpr int kLdsBytes = kEntries * (int)sizeof(TolLnEntry);
    static constexpr int kInvcAt = kEntries * 8;   // split layout: byte offset of the invc array
    // Every thread of the workgroup copies its share; ends with a workgroup barrier: call it before any wave may leave.
    static __device__ __forceinline__ void stage(TolLnEntry *lds)
    {
        for (int e = threadIdx.x; e < kEntries; e += blockDim.x) {
            const int slot = BINADES == 4 ? (((((e >> 8) + 126) & 3) << 8) | (e & 255)) : e;
#if EPIC_TOL_SPLIT_READS
            const TolLnEntry c = kTolLnTab[e];
            reinterpret_cast<double *>(lds)[slot] = c.lnc;
            reinterpret_cast<float *>(reinterpret_cast<char *>(lds) + kInvcAt)[2 * slot] = c.invc;
#else
            lds[slot] = kTolLnTab[e];
#endif
        }
        __syncthreads();
    }
    // byte offset of the entry of S: ((k & 3) << 8 | j) or ((k - 126) << 8 | j), times the slot size
    static __device__ __forceinline__ uint32_t offset(float s)
    {
        const uint32_t b = f2u(s);
#if EPIC_TOL_SPLIT_READS
        return BINADES == 4 ? ((b >> 12) & 0x1ff8u) : (((b - 0x3f000000u) >> 12) & 0x3ff8u);
#else
        return BINADES == 4 ? ((b >> 11) & 0x3ff0u) : (((b - 0x3f000000u) >> 11) & 0x7ff0u);
#endif
    }
    static __device__ __forceinline__ double ln(float s, const TolLnEntry *lds)
    {
        const char *at = reinterpret_cast<const char *>(lds) + offset(s);
#if EPIC_TOL_SPLIT_READS
        const double lnc = *reinterpret_cast<const double *>(at);
        const float invc = *reinterpret_cast<const float *>(at + kInvcAt);
#else
        const double lnc = reinterpret_cast<const TolLnEntry *>(at)->lnc;
        const float invc = reinterpret_cast<const TolLnEntry *>(at)->invc;
#endif
        const float r = __builtin_fmaf(s, invc, -1.0f);
        const float r2 = r * r;
        const float w = __builtin_fmaf(r2, __builtin_fmaf(r, 0x1.555556p-2f, -0.5f), r);
        return lnc + (double)w;
    }
};

// ---- kMathTol: one exp-class evaluation and one log per CELL instead of per NEIGHBOUR -------------------------------
// The reference evaluates u' = mx + ln(sum_i e^(u_i - mx)) - ln 2n with 2n expf and one logf per cell
// (harmonic_cpu.cpp:60-70), and every u_i goes through expf 2n times per sweep, once for each of its neighbours.  Here
// every cell's potential is split ONCE per sweep into  e^u = q 2^n :
//     n = rint(u log2 e),  f = u log2 e - n in [-1/2, 1/2]  (log2 e as hi + lo, two fma),  q = 2^f in [0.707, 1.414]
//     by a degree-7 f32 polynomial (tools/gen_exp2_poly.py: 0.003 ulp approximation error, mean 4e-4 ulp),
// all of it element-wise on the row a lane has just loaded -- so it runs as PACKED f32 (v_pk_fma_f32: two cells per
// instruction, and the dwordx4 a lane loads is two aligned register pairs already).  The neighbours of a cell reuse
// those pairs:
//     N = max n_i,   S = ((q_a 2^(n_a-N) + q_b 2^(n_b-N)) + q_c 2^(n_c-N)) + ...    f32, the reference's order
//     l = (float)(ln S - f_mx ln2)       = ln of the reference's s (S / q_mx; f_mx = mx log2 e - N), f64 inside, rounded
//                                          to f32 where the reference rounds logf(s)
//     t = mx + l,   u' = (float)((double)t - ln 2n)                                  as the reference
// The rounding stages of the reference -- f32 sum, l to f32, mx + l to f32, the f64 subtraction to f32 -- are all there;
// what differs is the noise inside the sum (the terms carry the rounding of q instead of the rounding of expf, ~4e-8
// relative either way, zero mean).  The terms are never formed by subtracting the maximum first, so Jacobi's two
// interleaved chains see the same term for the same neighbour value; on the reference's maps Jacobi, red-black and any
// tiling end in ONE fixed point with max |du| = 0 exactly, which is what lets the reference's absolute termination test
// fire under Jacobi (round 1's per-neighbour double-float mode ended in two fixed points one ulp apart on umass).
// tools/tol_study.c is the same arithmetic on the CPU, operation for operation, and relaxes the reference's maps with
// it; oracle/harmonic_oracle.c holds the checker's copy (oracle_tol_*), against which the kernels are bit-identical.
// n is carried as the BIT PATTERN of zm = u log2 e + 1.5 * 2^23 (an integer-valued float whose low mantissa bits are n
// in two's complement): bit patterns of such floats order and subtract like the integers, so max and differences need
// no conversion.  |u| <= 1e6 (the seed of obstacles / unreached cells) keeps n inside the 22 bits the trick has.
constexpr float kTolLog2eHi = 0x1.715476p+0f, kTolLog2eLo = 0x1.4ae0bep-26f;  // hi + lo = log2(e) to 49 bits
constexpr float kTolMagic = 12582912.0f;                                      // 1.5 * 2^23
constexpr uint32_t kTolMagicBits = 0x4b400000u;
constexpr double kLn2d = 0x1.62e42fefa39efp-1;

struct Split2 { v2f q; v2f zm; };  // zm's bit patterns carry n
__device__ __forceinline__ Split2 tol_split2(v2f u)
{
    Split2 s;
    s.zm = pk_fma(u, splat(kTolLog2eHi), splat(kTolMagic));
    const v2f nf = s.zm - splat(kTolMagic);
    v2f f = pk_fma(u, splat(kTolLog2eHi), -nf);
    f = pk_fma(u, splat(kTolLog2eLo), f);
    v2f p = pk_fma(f, splat(0x1.e5ba06p-17f), splat(0x1.44227cp-13f));
    p = pk_fma(f, p, splat(0x1.5da0f4p-10f));
    p = pk_fma(f, p, splat(0x1.3b2a4ap-7f));
    p = pk_fma(f, p, splat(0x1.c6b072p-5f));
    p = pk_fma(f, p, splat(0x1.ebfbep-3f));
    p = pk_fma(f, p, splat(0x1.62e43p-1f));
    s.q = pk_fma(f, p, splat(1.0f));
    return s;
}
// one cell, unpacked (v_fma_f32: the same IEEE operations as the packed form, so the same bits)
struct Split1 { float q, zm; };
__device__ __forceinline__ Split1 tol_split1(float u)
{
    Split1 s;
    s.zm = __builtin_fmaf(u, kTolLog2eHi, kTolMagic);
    const float nf = s.zm - kTolMagic;
    float f = __builtin_fmaf(u, kTolLog2eHi, -nf);
    f = __builtin_fmaf(u, kTolLog2eLo, f);
    float p = __builtin_fmaf(f, 0x1.e5ba06p-17f, 0x1.44227cp-13f);
    p = __builtin_fmaf(f, p, 0x1.5da0f4p-10f);
    p = __builtin_fmaf(f, p, 0x1.3b2a4ap-7f);
    p = __builtin_fmaf(f, p, 0x1.c6b072p-5f);
    p = __builtin_fmaf(f, p, 0x1.ebfbep-3f);
    p = __builtin_fmaf(f, p, 0x1.62e43p-1f);
    s.q = __builtin_fmaf(f, p, 1.0f);
    return s;
}
// a row of four cells: q and the bit patterns of zm
struct Split4 { float qx, qy, qz, qw; uint32_t nx, ny, nz, nw; };
__device__ __forceinline__ Split4 tol_split4(const float4 &u)
{
    const Split2 a = tol_split2(v2f{u.x, u.y}), b = tol_split2(v2f{u.z, u.w});
    return Split4{a.q.x, a.q.y, b.q.x, b.q.y, f2u(a.zm.x), f2u(a.zm.y), f2u(b.zm.x), f2u(b.zm.y)};
}
__device__ __forceinline__ uint32_t umax2(uint32_t a, uint32_t b) { return a > b ? a : b; }
// One term: q 2^(n - N).  v_ldexp_f32 takes the full i32 exponent: a neighbour at the seed (-1e6: n - N ~ -1.4e6) gives 0.
__device__ __forceinline__ float tol_term(float q, uint32_t n, uint32_t nmax) { return __builtin_ldexpf(q, (int)(n - nmax)); }
// Everything a cell's update needs from the maximum mx of its neighbours: N (as the bit pattern of zm, the same fma as in
// tol_split2, so it IS the largest of the neighbours' patterns) and f = mx log2 e - N, the fraction the split of that
// neighbour found, recomputed from mx in three f32 instructions -- cheaper than selecting it among the neighbours.
struct TolMax { float mx, f; uint32_t nmax; };
__device__ __forceinline__ TolMax tol_max(float mx)
{
    const float zm = __builtin_fmaf(mx, kTolLog2eHi, kTolMagic);
    const float nf = zm - kTolMagic;
    float f = __builtin_fmaf(mx, kTolLog2eHi, -nf);
    f = __builtin_fmaf(mx, kTolLog2eLo, f);
    return TolMax{mx, f, f2u(zm)};
}
// The reference's last steps on s_ref = S / q_mx = S 2^-f, whose logarithm is ln S - f ln2:
//     l = (float)(ln S - f ln2)          ONE rounding to f32, where the reference rounds logf(s)
//     t = mx + l                         f32, as the reference
//     u' = (float)((double)t - ln 2n)    as the reference (kLn4 / kLn6)
template <int BINADES>
__device__ __forceinline__ float tol_finish(float s, const TolMax &m, double ln2n, const TolLnEntry *tab)
{
    const float l = (float)__builtin_fma(-(double)m.f, kLn2d, TolLn<BINADES>::ln(s, tab));
    const float t = m.mx + l;
    return (float)((double)t - ln2n);
}
// The same update for TWO cells at a time, in three phases, so that a kernel can keep the table reads of one pair of cells
// in flight while it works on the other pair: pre(x, z), issue(x, z), pre(y, w), issue(y, w), wait for the first pair,
// post(x, z), wait for the second, post(y, w).  One cell at a time the compiler issued a lookup and parked the wave at
// s_waitcnt lgkmcnt(0) twice per cell -- sixteen LDS round trips per step of the fused pass (28 % of all wave cycles; with
// the reads merely grouped and waited for at once they still cost 10 % of the pass: 190 vs 171 us per launch with the reads
// compiled out, same box; profiles/r03_experiments.txt).  The element-wise parts (N and f of the maximum, the sum) are
// packed f32 over the pair.  Same operations on the same values as tol_update_2d / _3d: same bits.
struct TolPre2 { v2f s, mx, f; };
struct TolLnRaw { uint32_t lo, hi, invc; };  // a table entry as it leaves the LDS: the two halves of lnc, and invc
// N (bit patterns) and f of the two maxima: tol_max, packed
__device__ __forceinline__ void tol_max2(v2f mx, v2f &f, uint32_t &n0, uint32_t &n1)
{
    const v2f zm = pk_fma(mx, splat(kTolLog2eHi), splat(kTolMagic));
    const v2f nf = zm - splat(kTolMagic);
    f = pk_fma(mx, splat(kTolLog2eHi), -nf);
    f = pk_fma(mx, splat(kTolLog2eLo), f);
    n0 = f2u(zm.x);
    n1 = f2u(zm.y);
}
// A neighbour that lives in the NEXT LANE (the fused passes: lane i - 1 holds the cell to the left of lane i's first cell, lane
// i + 1 the one to the right of its last) costs three lane shifts -- u, q, n -- as moves.  Two of the three consumers are VOP2
// instructions, which take the shift as a DPP modifier on their first operand: the maximum and the exponent difference are
// therefore written out with the shift folded in (the compiler forms v_max3 first and then cannot); only q still moves.
// s_nop 1: a DPP operand must not have been written by the VALU in the two wait states before (the compiler does not look
// into the statement).  Lanes without a neighbour read zero bits (bound_ctrl), as with the plain shifts: halo lanes.
__device__ __forceinline__ float max_from_left(float v, float m)   // lane i: max(v of lane i - 1, m)
{
    float r;
    asm("s_nop 1\n\tv_max_f32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(r) : "v"(v), "v"(m));
    return r;
}
__device__ __forceinline__ float max_from_right(float v, float m)  // lane i: max(v of lane i + 1, m)
{
    float r;
    asm("s_nop 1\n\tv_max_f32_dpp %0, %1, %2 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(r) : "v"(v), "v"(m));
    return r;
}
__device__ __forceinline__ uint32_t sub_from_left(uint32_t n, uint32_t nmax)   // lane i: n of lane i - 1, minus nmax
{
    uint32_t r;
    asm("s_nop 1\n\tv_sub_u32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(r) : "v"(n), "v"(nmax));
    return r;
}
__device__ __forceinline__ uint32_t sub_from_right(uint32_t n, uint32_t nmax)
{
    uint32_t r;
    asm("s_nop 1\n\tv_sub_u32_dpp %0, %1, %2 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(r) : "v"(n), "v"(nmax));
    return r;
}
// The pair (x, z) of a fused pass: x's left neighbour (third in the reference's order) is the LAST cell of the lane to the
// left -- uw, nw are this lane's own last cell (unshifted), qw_left its q already shifted.  Otherwise tol_pre2_2d.
__device__ __forceinline__ TolPre2 tol_pre2_2d_left(float ua0, float ub0, float uw, float ud0, float qa0, uint32_t na0, float qb0, uint32_t nb0,
                                                    float qw_left, uint32_t nw, float qd0, uint32_t nd0, float ua1, float ub1, float uc1,
                                                    float ud1, float qa1, uint32_t na1, float qb1, uint32_t nb1, float qc1, uint32_t nc1,
                                                    float qd1, uint32_t nd1)
{
    TolPre2 p;
    p.mx = v2f{max_from_left(uw, max2(max2(ua0, ub0), ud0)), max2(max2(max2(ua1, ub1), uc1), ud1)};
    uint32_t n0, n1;
    tol_max2(p.mx, p.f, n0, n1);
    v2f s = v2f{tol_term(qa0, na0, n0), tol_term(qa1, na1, n1)} + v2f{tol_term(qb0, nb0, n0), tol_term(qb1, nb1, n1)};
    s = s + v2f{__builtin_ldexpf(qw_left, (int)sub_from_left(nw, n0)), tol_term(qc1, nc1, n1)};
    p.s = s + v2f{tol_term(qd0, nd0, n0), tol_term(qd1, nd1, n1)};
    return p;
}
// The pair (y, w): w's right neighbour (fourth) is the FIRST cell of the lane to the right -- ux, nx this lane's own first cell.
__device__ __forceinline__ TolPre2 tol_pre2_2d_right(float ua0, float ub0, float uc0, float ud0, float qa0, uint32_t na0, float qb0, uint32_t nb0,
                                                     float qc0, uint32_t nc0, float qd0, uint32_t nd0, float ua1, float ub1, float uc1,
                                                     float ux, float qa1, uint32_t na1, float qb1, uint32_t nb1, float qc1, uint32_t nc1,
                                                     float qx_right, uint32_t nx)
{
    TolPre2 p;
    p.mx = v2f{max2(max2(max2(ua0, ub0), uc0), ud0), max_from_right(ux, max2(max2(ua1, ub1), uc1))};
    uint32_t n0, n1;
    tol_max2(p.mx, p.f, n0, n1);
    v2f s = v2f{tol_term(qa0, na0, n0), tol_term(qa1, na1, n1)} + v2f{tol_term(qb0, nb0, n0), tol_term(qb1, nb1, n1)};
    s = s + v2f{tol_term(qc0, nc0, n0), tol_term(qc1, nc1, n1)};
    p.s = s + v2f{tol_term(qd0, nd0, n0), __builtin_ldexpf(qx_right, (int)sub_from_right(nx, n1))};
    return p;
}
// cell 0: neighbours a0 b0 c0 d0 in the reference's order of summation (their u, q, n); cell 1 likewise
__device__ __forceinline__ TolPre2 tol_pre2_2d(float ua0, float ub0, float uc0, float ud0, float qa0, uint32_t na0, float qb0, uint32_t nb0,
                                               float qc0, uint32_t nc0, float qd0, uint32_t nd0, float ua1, float ub1, float uc1, float ud1,
                                               float qa1, uint32_t na1, float qb1, uint32_t nb1, float qc1, uint32_t nc1, float qd1,
                                               uint32_t nd1)
{
    TolPre2 p;
    p.mx = v2f{max2(max2(max2(ua0, ub0), uc0), ud0), max2(max2(max2(ua1, ub1), uc1), ud1)};
    uint32_t n0, n1;
    tol_max2(p.mx, p.f, n0, n1);
    v2f s = v2f{tol_term(qa0, na0, n0), tol_term(qa1, na1, n1)} + v2f{tol_term(qb0, nb0, n0), tol_term(qb1, nb1, n1)};
    s = s + v2f{tol_term(qc0, nc0, n0), tol_term(qc1, nc1, n1)};
    p.s = s + v2f{tol_term(qd0, nd0, n0), tol_term(qd1, nd1, n1)};
    return p;
}
// six neighbours each, x0-1, x0+1, x1-1, x1+1, x2-1, x2+1 (harmonic_cpu.cpp:118-123): u[], q[], n[] of cell 0 and of cell 1
struct TolNb6 { float u0, u1, u2, u3, u4, u5, q0, q1, q2, q3, q4, q5; uint32_t n0, n1, n2, n3, n4, n5; };
__device__ __forceinline__ TolPre2 tol_pre2_3d(const TolNb6 &a, const TolNb6 &b)
{
    TolPre2 p;
    p.mx = v2f{max2(max2(max2(max2(max2(a.u0, a.u1), a.u2), a.u3), a.u4), a.u5),
               max2(max2(max2(max2(max2(b.u0, b.u1), b.u2), b.u3), b.u4), b.u5)};
    uint32_t n0, n1;
    tol_max2(p.mx, p.f, n0, n1);
    v2f s = v2f{tol_term(a.q0, a.n0, n0), tol_term(b.q0, b.n0, n1)} + v2f{tol_term(a.q1, a.n1, n0), tol_term(b.q1, b.n1, n1)};
    s = s + v2f{tol_term(a.q2, a.n2, n0), tol_term(b.q2, b.n2, n1)};
    s = s + v2f{tol_term(a.q3, a.n3, n0), tol_term(b.q3, b.n3, n1)};
    s = s + v2f{tol_term(a.q4, a.n4, n0), tol_term(b.q4, b.n4, n1)};
    p.s = s + v2f{tol_term(a.q5, a.n5, n0), tol_term(b.q5, b.n5, n1)};
    return p;
}
// LDS byte address of the table entry of S (the low half of a generic pointer into LDS is the LDS address)
template <int BINADES>
__device__ __forceinline__ uint32_t tol_ln_addr(float s, const TolLnEntry *lds)
{
    return (uint32_t)(uintptr_t)lds + TolLn<BINADES>::offset(s);
}
// The reads of a pair are ISSUED here and WAITED FOR later (tol_ln_wait).  Written out in assembly: the compiler waits after
// every read of its own, and it does not count LDS operations issued from inline assembly, so the waits are ours.  LDS
// operations of a wave complete in order, so "at most N outstanding" (lgkmcnt(N)) leaves only the N youngest reads in
// flight whatever scalar loads are in flight beside them.  The scheduling barriers keep the compiler from moving the code
// that is meant to cover the round trip to the other side of the statement.
typedef unsigned vu3_t __attribute__((ext_vector_type(3)));
typedef unsigned vu2_t __attribute__((ext_vector_type(2)));
#if EPIC_TOL_SPLIT_READS
constexpr int kTolReadsPerCell = 2;
struct TolLnPair { vu2_t a, b; uint32_t ia, ib; };
#else
constexpr int kTolReadsPerCell = 1;
struct TolLnPair { vu3_t a, b; };
#endif
template <int BINADES>
__device__ __forceinline__ TolLnPair tol_ln_issue(const TolPre2 &p, const TolLnEntry *lds)
{
    TolLnPair r;
    const uint32_t a0 = tol_ln_addr<BINADES>(p.s.x, lds), a1 = tol_ln_addr<BINADES>(p.s.y, lds);
#if EPIC_TOL_SPLIT_READS
    asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %5\n\tds_read_b32 %2, %4 offset:%6\n\tds_read_b32 %3, %5 offset:%6"
                 : "=&v"(r.a), "=&v"(r.b), "=&v"(r.ia), "=&v"(r.ib) : "v"(a0), "v"(a1), "n"(TolLn<BINADES>::kInvcAt));
#else
    asm volatile("ds_read_b96 %0, %2\n\tds_read_b96 %1, %3" : "=&v"(r.a), "=&v"(r.b) : "v"(a0), "v"(a1));
#endif
    __builtin_amdgcn_sched_barrier(0);
    return r;
}
// the pair is in its registers once at most `N` younger table lookups (cells) of this wave are still outstanding
template <int N>
__device__ __forceinline__ void tol_ln_wait(TolLnPair &r, TolLnRaw &e0, TolLnRaw &e1)
{
    static_assert(N * kTolReadsPerCell <= 15, "lgkmcnt is a 4-bit counter");
    __builtin_amdgcn_sched_barrier(0);
#if EPIC_TOL_SPLIT_READS
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(r.a), "+v"(r.b), "+v"(r.ia), "+v"(r.ib) : "n"(N * kTolReadsPerCell));
    e0 = TolLnRaw{r.a.x, r.a.y, r.ia};
    e1 = TolLnRaw{r.b.x, r.b.y, r.ib};
#else
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(r.a), "+v"(r.b) : "n"(N));
    e0 = TolLnRaw{r.a.x, r.a.y, r.a.z};
    e1 = TolLnRaw{r.b.x, r.b.y, r.b.z};
#endif
}
// phase 3 of one cell up to l (TolLn::ln and the first rounding of tol_finish) ...
__device__ __forceinline__ float tol_post_l(float s, float f, const TolLnRaw &e)
{
    const float invc = u2f(e.invc);
    const double lnc = __builtin_bit_cast(double, (uint64_t)e.lo | ((uint64_t)e.hi << 32));
    const float r = __builtin_fmaf(s, invc, -1.0f);
    const float r2 = r * r;
    const float w = __builtin_fmaf(r2, __builtin_fmaf(r, 0x1.555556p-2f, -0.5f), r);
    return (float)__builtin_fma(-(double)f, kLn2d, lnc + (double)w);
}
// ... and of the pair: t = mx + l (packed), u' = (float)((double)t - ln 2n)
__device__ __forceinline__ void tol_post2(const TolPre2 &p, const TolLnRaw &e0, const TolLnRaw &e1, double ln2n, float &o0, float &o1)
{
    const v2f t = p.mx + v2f{tol_post_l(p.s.x, p.f.x, e0), tol_post_l(p.s.y, p.f.y, e1)};
    o0 = (float)((double)t.x - ln2n);
    o1 = (float)((double)t.y - ln2n);
}
// neighbours in the reference's order of summation: up, down, left, right (harmonic_cpu.cpp:65-68); u* = their values
__device__ __forceinline__ float tol_update_2d(float uu, float ud, float ul, float ur, float qu, uint32_t nu, float qd,
                                               uint32_t nd, float ql, uint32_t nl, float qr, uint32_t nr, const TolLnEntry *tab)
{
    const TolMax m = tol_max(max2(max2(max2(uu, ud), ul), ur));
    float s = tol_term(qu, nu, m.nmax) + tol_term(qd, nd, m.nmax);
    s = s + tol_term(ql, nl, m.nmax);
    s = s + tol_term(qr, nr, m.nmax);
    return tol_finish<4>(s, m, kLn4, tab);
}
// x0-1, x0+1, x1-1, x1+1, x2-1, x2+1 (harmonic_cpu.cpp:118-123)
__device__ __forceinline__ float tol_update_3d(float u0, float u1, float u2, float u3, float u4, float u5, float q0, uint32_t n0,
                                               float q1, uint32_t n1, float q2, uint32_t n2, float q3, uint32_t n3, float q4,
                                               uint32_t n4, float q5, uint32_t n5, const TolLnEntry *tab)
{
    const TolMax m = tol_max(max2(max2(max2(max2(max2(u0, u1), u2), u3), u4), u5));
    float s = tol_term(q0, n0, m.nmax) + tol_term(q1, n1, m.nmax);
    s = s + tol_term(q2, n2, m.nmax);
    s = s + tol_term(q3, n3, m.nmax);
    s = s + tol_term(q4, n4, m.nmax);
    s = s + tol_term(q5, n5, m.nmax);
    return tol_finish<5>(s, m, kLn6, tab);
}

// ---- selects on lane masks held in SGPR pairs ----------------------------------------------------------------------
// Measured on gfx950 (tools/ubench_alu2.hip, profiles/r01_ubench_alu2.txt): v_cndmask_b32 in its VOP2 form (mask in
// VCC) issues in ~22 cycles unless it directly follows the compare that wrote VCC; the VOP3 form with the mask in an
// ordinary SGPR pair issues in 4.3 like any other VALU instruction.  The selects of the update are therefore written
// out in the VOP3 form, on masks that come from v_cmp (ballot of a comparison) and are combined with scalar s_or.
typedef uint64_t lmask;
__device__ __forceinline__ lmask lanes_eq(float a, float b) { return __builtin_amdgcn_ballot_w64(a == b); }
// lanes whose two values differ as BIT PATTERNS (activity tracking: "was this cell rewritten with different bits")
__device__ __forceinline__ lmask lanes_ne(float a, float b)
{
    return __builtin_amdgcn_ballot_w64(__builtin_bit_cast(uint32_t, a) != __builtin_bit_cast(uint32_t, b));
}
__device__ __forceinline__ float sel(lmask m, float if_set, float if_clear)
{
#ifndef EPIC_ASM_SELECTS
    // the lane mask handed to the compiler as the i1 it is: one v_cndmask_b32_e64 on the SGPR pair (checked in the ISA:
    // all selects of the sweep kernels come out in the VOP3 form), and the compiler sees the instruction
    return __builtin_amdgcn_inverse_ballot_w64(m) ? if_set : if_clear;
#else  // the same instruction written out (A/B builds: make EXTRA=-DEPIC_ASM_SELECTS)
    float r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(if_clear), "v"(if_set), "s"(m));
    return r;
#endif
}
// ---- selects as writes under a narrowed EXEC mask ------------------------------------------------------------------
// "r = m ? a op b : r" for a cheap op (f32 add / sub) costs less as the instruction itself executed by the lanes of m
// only than as the instruction plus a v_cndmask: 2-operand f32 adds issue in 2.4 cycles, a select in 4.4, and the two
// s_mov that narrow and restore EXEC run on the scalar unit (8192^2 sweep 150.2 -> 147.6 us, same box; the lock select
// of the kernels, a plain move, did not gain and stays a v_cndmask).  EXEC is set back to ALL LANES afterwards, not to a
// saved copy (one s_mov less per select, 147.8 -> 146.5 us): the update must be called with all 64 lanes active -- which
// the sweep kernels guarantee anyway (rows are padded to whole wave strips; the DPP neighbour shifts need it too).
// The statements are `asm volatile`: executed where they are written, never sunk into a lane-masked region of the
// compiler's own (where "all lanes" would be wrong); it also schedules better (145.8 -> 144.1 us).
// EXEC IS ON THE CLOBBER LIST: the compiler then knows that the statement writes EXEC and keeps the wait states the
// hardware wants between an EXEC write and a DPP instruction (5) by itself -- without the clobber it scheduled a DPP
// shift two instructions behind such a restore, which left lanes 12..15 of every 16 with stale neighbours.  (clang
// warns that EXEC is a reserved register that "may not be preserved": it is not -- it is all-lanes afterwards, which is
// what it was before, the precondition stated above.)  tools/isa_hazards.py checks the distance in the generated ISA.
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void sub_where(float &r, lmask m, float a, float b)  // lanes of m: r = a - b
{
    asm volatile("s_mov_b64 exec, %1\n\tv_sub_f32 %0, %2, %3\n\ts_mov_b64 exec, -1"
        : "+v"(r) : "s"(m), "v"(a), "v"(b) : "exec");
}
__device__ __forceinline__ void add_where(float &r, lmask m, float a, float b)  // lanes of m: r = a + b
{
    asm volatile("s_mov_b64 exec, %1\n\tv_add_f32 %0, %2, %3\n\ts_mov_b64 exec, -1"
        : "+v"(r) : "s"(m), "v"(a), "v"(b) : "exec");
}
__device__ __forceinline__ void add_one_where(float &r, lmask m, float a)  // lanes of m: r = a + 1
{
    asm volatile("s_mov_b64 exec, %1\n\tv_add_f32 %0, 1.0, %2\n\ts_mov_b64 exec, -1"
        : "+v"(r) : "s"(m), "v"(a) : "exec");
}
// lanes of m: t = a + 1, then r = t + b (the two additions of the "maximum is in the vertical pair" case, one narrowing)
__device__ __forceinline__ void add_one_add_where(float &r, lmask m, float a, float b)
{
    float t;
    asm volatile("s_mov_b64 exec, %2\n\tv_add_f32 %1, 1.0, %3\n\tv_add_f32 %0, %1, %4\n\ts_mov_b64 exec, -1"
        : "+v"(r), "=&v"(t) : "s"(m), "v"(a), "v"(b) : "exec");
}

#pragma clang diagnostic pop

template <int MATH>
__device__ __forceinline__ float m_exp(float x, const MathTab &lds)
{
    return MATH == kMathFast ? hw_exp(x) : precise_exp(x, lds);
}
template <int MATH>
__device__ __forceinline__ float m_ln(float s, const MathTab &lds)
{
    return MATH == kMathFast ? hw_ln(s) : precise_ln(s, lds);
}

template <int MATH>
__device__ __forceinline__ float cell_update_2d(float up, float down, float left, float right, const MathTab &lds)
{
    if (MATH == kMathTraffic) return ((up + down) + left) + right;
    float mx = max2(max2(max2(up, down), left), right);
    if (MATH == kMathPrecise) {
        // One of the four terms is expf(0) = 1 exactly -- the neighbour that IS the maximum (the first one, if several
        // tie: the others go through the exp and come out as 1 as well).  Evaluating only the other three saves one of
        // the four f64 exps (10 of the 51 four-cycle instructions per cell) for a handful of selects and compares.
        // The reference's sum is ((e_up + e_down) + e_left) + e_right; its first addition commutes, so the vertical
        // pair may be taken as hv = max(up, down), lv = min(up, down): e_lv is always evaluated, and with
        //   P = "hv is the maximum",  Q = "left is the maximum" (else right is)
        // the other two arguments are  b = P ? left : hv,  c = P | Q ? right : left,  and the sum reads
        //   P      : ((1 + E_lv) + E_b) + E_c        (b = left, c = right)
        //   !P,  Q : ((E_b + E_lv) + 1) + E_c        (b = hv,   c = right)
        //   !P, !Q : ((E_b + E_lv) + E_c) + 1        (b = hv,   c = left)
        // i.e. E_lv + (P ? 1 : E_b), then + (P ? E_b : Q ? 1 : E_c), then + (P | Q ? E_c : 1) -- the same f32 additions
        // of the same values in the same order as the four-exp form, hence the same bits.
        const float hv = max2(up, down), lv = __builtin_fminf(up, down);
        mx = max2(max2(hv, left), right);
        const lmask P = lanes_eq(hv, mx), Q = lanes_eq(left, mx), PQ = P | Q;
        // the selects are writes under a narrowed EXEC mask (above): every lane first takes the "right is the maximum"
        // form, the lanes of Q and then of P overwrite it with theirs
        float db = hv - mx, dc = left - mx;
        sub_where(db, P, left, mx);
        sub_where(dc, PQ, right, mx);
        const float ea = precise_exp(lv - mx, lds), eb = precise_exp(db, lds), ec = precise_exp(dc, lds);
        const float t1 = ea + eb;
        float t2 = t1 + ec;
        add_one_where(t2, Q, t1);
        add_one_add_where(t2, P, ea, eb);   // P wins over Q: (ea + 1) + eb
        float s = t2 + 1.0f;
        add_where(s, PQ, t2, ec);
        const float t = mx + precise_ln(s, lds);
        return (float)((double)t - lds.ln4);
    }
    float s = m_exp<MATH>(up - mx, lds) + m_exp<MATH>(down - mx, lds);
    s = s + m_exp<MATH>(left - mx, lds);
    s = s + m_exp<MATH>(right - mx, lds);
    float t = mx + m_ln<MATH>(s, lds);
    return (float)((double)t - kLn4);
}

template <int MATH>
__device__ __forceinline__ float cell_update_3d(float a0, float a1, float b0, float b1, float c0, float c1,
                                                const MathTab &lds)
{
    float mx = max2(max2(max2(max2(max2(a0, a1), b0), b1), c0), c1);
    float s = m_exp<MATH>(a0 - mx, lds) + m_exp<MATH>(a1 - mx, lds);
    s = s + m_exp<MATH>(b0 - mx, lds);
    s = s + m_exp<MATH>(b1 - mx, lds);
    s = s + m_exp<MATH>(c0 - mx, lds);
    s = s + m_exp<MATH>(c1 - mx, lds);
    float t = mx + m_ln<MATH>(s, lds);
    return (float)((double)t - kLn6);
}

// ---- raw buffer descriptors ------------------------------------------------------------------------------------
// The sweeps address rows through V# descriptors (buffer_load / buffer_store ... offen with an SGPR soffset), which keeps
// every address computation on the scalar unit.  Word 3 of a gfx9-family raw buffer descriptor: DATA_FORMAT = 32-bit
// (bits 15..18 = 4), everything else 0 (no swizzle, no typed conversion); num_records is the byte range checked by the
// hardware -- the kernels clamp their row indices themselves and keep offsets below 2 GiB, so it is simply the maximum.
constexpr int kRawBufferWord3 = 0x00020000;
constexpr int kRawBufferRange = 0x7fffffff;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t raw_buffer(const void *base)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, kRawBufferRange, kRawBufferWord3);
}
constexpr int kStoreNonTemporal = 2;  // cache-policy operand of the buffer store builtins: the "nt" bit

// One row of a lane (16 bytes) to memory, non-temporal.
// HAZARD (measured on gfx950, ROCm 7.2): a VALU instruction that writes a data register of a 16-byte buffer store in the
// instruction slots right behind the store can still reach the store -- with v_pk_fma_f32 directly behind it, lanes
// 12..15 of every 16 of the SECOND data register went to memory with the new value (the first Horner step of the next
// row's split instead of u).  The compiler's hazard recognizer knows this hazard but exempts stores whose soffset is an
// SGPR, which is how every store here is addressed, so it pads nothing.  store_row() therefore keeps the four data
// registers alive past the store (they are operands of the statement behind it, which the "memory" clobber orders after
// the store) and pads four wait states before anything else may issue.  tools/isa_hazards.py scans the generated ISA
// for any VALU write to store data within that distance (tests/test_isa_hazards.py).
typedef unsigned vu4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_row(const __amdgpu_buffer_rsrc_t &rsrc, float x, float y, float z, float w, unsigned lane_off,
                                          unsigned row_off)
{
    const vu4_t data = {__builtin_bit_cast(uint32_t, x), __builtin_bit_cast(uint32_t, y), __builtin_bit_cast(uint32_t, z),
                        __builtin_bit_cast(uint32_t, w)};
    __builtin_amdgcn_raw_buffer_store_b128(data, rsrc, lane_off, row_off, kStoreNonTemporal);
    asm volatile("s_nop 3" : : "v"(data) : "memory");
}

// Full-wave (64-lane) shifts by one lane: one v_mov_b32_dpp each on gfx9-family ISAs, through the compiler's builtin so
// that its hazard recognizer places the wait states a DPP instruction needs (5 behind a write to EXEC -- the masked adds
// above declare theirs --, 2 behind a VALU write to its source VGPR).  EPIC_ASM_DPP selects the round-1 form with the
// instruction written out behind an s_nop 4 (A/B builds).
// lane i receives lane i-1's `v`; lane 0 keeps `edge`.
__device__ __forceinline__ float wave_from_left(float v, float edge)
{
#ifndef EPIC_ASM_DPP
    return u2f(__builtin_amdgcn_update_dpp(f2u(edge), f2u(v), 0x138 /* wave_shr:1 */, 0xf, 0xf, false));
#else
    float r = edge;
    asm volatile("s_nop 4\n\tv_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(r) : "v"(v));
    return r;
#endif
}
// lane i receives lane i+1's `v`; lane 63 keeps `edge`.
__device__ __forceinline__ float wave_from_right(float v, float edge)
{
#ifndef EPIC_ASM_DPP
    return u2f(__builtin_amdgcn_update_dpp(f2u(edge), f2u(v), 0x130 /* wave_shl:1 */, 0xf, 0xf, false));
#else
    float r = edge;
    asm volatile("s_nop 4\n\tv_mov_b32_dpp %0, %1 wave_shl:1 row_mask:0xf bank_mask:0xf" : "+v"(r) : "v"(v));
    return r;
#endif
}

// Three registers' values of lane K (left*) and of lane 32 + K (right*), broadcast to all lanes: six ds_bpermute_b32 (the
// LDS crossbar; no LDS memory, no VALU issue), the lane selected by the instruction's OFFSET field on top of an address
// register holding 0 in every lane (`zero`).  Written out because the builtin either becomes v_readlane_b32 + v_mov_b32 (when the
// compiler can see that the address is uniform: two VALU instructions per value) or keeps one address register per lane
// number (when it cannot: twenty registers in the ten-row trip of the tol sweep, which then spills).  The compiler does
// not count LDS operations issued from inline assembly, so the block ends with its own wait.
template <int K>
__device__ __forceinline__ void edge_from_lanes(int zero, float a, float b, float c, float &la, float &ra, float &lb, float &rb,
                                                float &lc, float &rc)
{
    static_assert(K >= 0 && K < 32, "lanes K and 32 + K");
    asm("ds_bpermute_b32 %0, %6, %7 offset:%10\n\t"
        "ds_bpermute_b32 %1, %6, %7 offset:%11\n\t"
        "ds_bpermute_b32 %2, %6, %8 offset:%10\n\t"
        "ds_bpermute_b32 %3, %6, %8 offset:%11\n\t"
        "ds_bpermute_b32 %4, %6, %9 offset:%10\n\t"
        "ds_bpermute_b32 %5, %6, %9 offset:%11\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(la), "=&v"(ra), "=&v"(lb), "=&v"(rb), "=&v"(lc), "=&v"(rc)
        : "v"(zero), "v"(a), "v"(b), "v"(c), "n"(4 * K), "n"(4 * (32 + K)));
}

// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>): a loop whose index is a constant expression in the body
template <class F, int... J>
__device__ __forceinline__ void unrolled_seq(F &f, std::integer_sequence<int, J...>) { (f(std::integral_constant<int, J>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void unrolled(F &f) { unrolled_seq(f, std::make_integer_sequence<int, N>{}); }

// max over the 64 lanes of a non-negative float, result valid in every lane.
__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = max2(v, __shfl_xor(v, off, 64));
    return v;
}

}  // namespace epic_hip
