// cell_update.h -- the per-cell log-sum-exp update, device side (gfx950 only).
//
// Mirrors the ROUNDING SEQUENCE of the reference CPU solver
// (libepic/src/harmonic/harmonic_cpu.cpp:60-70 for 2-D, :110-123 for 3-D), because the
// f32 stagnation point the solver converges to is decided by it:
//     mx = max(neighbours)                                   f32
//     s  = ((e(a-mx) + e(b-mx)) + e(c-mx)) + e(d-mx) ...     f32, left-associated
//     t  = mx + ln(s)                                        f32
//     u  = (float)((double)t - ln(2n))                       f64 subtract, one rounding
// e() and ln() are the CDNA4 hardware transcendentals (v_exp_f32 / v_log_f32, 1 ulp) with
// the base change done in f32; this is NOT the reference CUDA kernel's all-f32
// `- 1.38629436f` form (harmonic_gpu.cu:57-61), which lands 1e-2 away on ill-conditioned
// maps.  Compile with -ffp-contract=off so no step is fused.
#pragma once
#include <hip/hip_runtime.h>

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

__device__ __forceinline__ float cell_update_2d(float up, float down, float left, float right)
{
    float mx = max2(max2(max2(up, down), left), right);
    float s = hw_exp(up - mx) + hw_exp(down - mx);
    s = s + hw_exp(left - mx);
    s = s + hw_exp(right - mx);
    float t = mx + hw_ln(s);
    return (float)((double)t - kLn4);
}

__device__ __forceinline__ float cell_update_3d(float a0, float a1, float b0, float b1, float c0, float c1)
{
    float mx = max2(max2(max2(max2(max2(a0, a1), b0), b1), c0), c1);
    float s = hw_exp(a0 - mx) + hw_exp(a1 - mx);
    s = s + hw_exp(b0 - mx);
    s = s + hw_exp(b1 - mx);
    s = s + hw_exp(c0 - mx);
    s = s + hw_exp(c1 - mx);
    float t = mx + hw_ln(s);
    return (float)((double)t - kLn6);
}

// Full-wave (64-lane) shifts by one lane: one v_mov_b32_dpp each on gfx9-family ISAs.
// lane i receives lane i-1's `v`; lane 0 keeps `edge`.
__device__ __forceinline__ float wave_from_left(float v, float edge)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, edge),
                                                                 __builtin_bit_cast(int, v), 0x138 /*wave_shr:1*/,
                                                                 0xf, 0xf, false));
}
// lane i receives lane i+1's `v`; lane 63 keeps `edge`.
__device__ __forceinline__ float wave_from_right(float v, float edge)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, edge),
                                                                 __builtin_bit_cast(int, v), 0x130 /*wave_shl:1*/,
                                                                 0xf, 0xf, false));
}

// max over the 64 lanes of a non-negative float, result valid in every lane.
__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = max2(v, __shfl_xor(v, off, 64));
    return v;
}

}  // namespace epic_hip
