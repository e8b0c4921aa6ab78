// Device-side scalar math shared by the kernels (gfx950, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/blurry_edges_hip.h"

namespace be {

// eta = 10^(2 erf(p) - 2)                              utils/postprocessing_loss.py:88-89
__device__ __forceinline__ float param2eta(float p) {
#pragma clang fp contract(off)
    return powf(10.0f, erff(p) * 2.0f - 2.0f);
}

// utils/depth_etas.py:23-34 -- same operation order as the reference, no contraction.
__device__ __forceinline__ float etas2depth(const be_depth_consts& c, float e1, float e2, int& branch) {
#pragma clang fp contract(off)
    const float I = c.intercept;
    const float c1 = (-c.sin_w) * e1 + c.cos_w * (e2 - I);
    const float c2 = (-c.sin_m) * (e1 - I) + c.cos_m * e2;
    const float c3 = (-c.sin_w) * (e1 - I) + c.cos_w * e2;
    const float sum_h = (e1 + e2 - I) / 2.0f;
    float e11, e22;
    if (c1 > 0.0f)      { e11 = sum_h;                        e22 = I + sum_h;              branch = 0; }
    else if (c2 > 0.0f) { e11 = I + (e1 - e2 - I) / 2.0f;     e22 = (e2 - e1 + I) / 2.0f;   branch = 1; }
    else if (c3 < 0.0f) { e11 = I + sum_h;                    e22 = sum_h;                  branch = 2; }
    else                { e11 = e1;                           e22 = e2;                     branch = 3; }
    // python_float / tensor is tensor.reciprocal() * python_float in PyTorch (Tensor.__rtruediv__): two roundings
    return (1.0f / (c.k2 * (e11 * e11 - e22 * e22) + c.den_const)) * c.numerator;
}

// utils/depth_etas.py:36-37
__device__ __forceinline__ float depth2sigma(const be_depth_consts& c, float depth, float rho_prime) {
#pragma clang fp contract(off)
    return fabsf((1.0f / depth - rho_prime) * c.s + 1.0f) / c.k;
}

// torch.remainder(a, 2*pi) for fp32 (result in [0, 2pi)): fmod, then shift negatives up.
__device__ __forceinline__ float remainder_2pi(float a) {
#pragma clang fp contract(off)
    const float two_pi = 6.283185307179586f;
    float r = fmodf(a, two_pi);
    if (r != 0.0f && r < 0.0f) r += two_pi;
    return r;
}

// Smish(x) = x * tanh(log(1 + sigmoid(x)))            models/local_stage.py:4-6
// With u = 1 + sigmoid(x): tanh(log u) = (u^2-1)/(u^2+1); in t = exp(-|x|) this is a ratio of two
// quadratics that never overflows:  x>=0: (3+2t)/(5+6t+2t^2)   x<0: (3t^2+2t)/(5t^2+6t+2).
// exp through v_exp_f32 and the quotient through v_rcp_f32 (1 ulp each): the epilogue evaluates this once per output
// element, and with the IEEE expf + correctly-rounded division it was a quarter of the short-K launches.
__device__ __forceinline__ float smish(float x) {
    // branch-free: both cases are num = t (n2 t + 2) + n0, den = t (d2 t + 6) + d0 with (n2, n0, d2, d0) = (0, 3, 2, 5) for
    // x >= 0 and (3, 0, 5, 2) for x < 0 - the same fmaf chains as the two-branch form (bit-identical), four selects instead
    // of a divergent branch per element (the epilogues evaluate this once per output value)
    const float t = __expf(-fabsf(x));
    const bool pos = x >= 0.0f;
    const float n2 = pos ? 0.0f : 3.0f, n0 = pos ? 3.0f : 0.0f, d2 = pos ? 2.0f : 5.0f, d0 = pos ? 5.0f : 2.0f;
    const float num = fmaf(t, fmaf(n2, t, 2.0f), n0);
    const float den = fmaf(t, fmaf(d2, t, 6.0f), d0);
    return x * (num * __frcp_rn(den));
}

// Sum over the 64 lanes of a wave; every lane gets the total.  Fixed order -> bitwise reproducible.
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// The same sum with register-to-register lane exchanges only (no ds_bpermute: a butterfly step through the LDS crossbar costs an
// LDS instruction and its latency): v_permlane32_swap / v_permlane16_swap for the partners 32 and 16 lanes away, DPP for the
// four steps inside a row of 16 (rotate by 8 = xor 8, half-row mirror = xor 7, quad permutes xor 2 and xor 1: the four masks span
// the row).  Fixed order -> bitwise reproducible; NOT the order of wave_sum, so the two differ in the last bits.
__device__ __forceinline__ float wave_sum_dpp(float v) {
    {
        auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
    {
        auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x128, 0xf, 0xf, false));      // row_ror:8
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x141, 0xf, 0xf, false));      // row_half_mirror
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x4e, 0xf, 0xf, false));       // quad_perm [2,3,0,1]
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0xb1, 0xf, 0xf, false));       // quad_perm [1,0,3,2]
    return v;
}

__device__ __forceinline__ double wave_sum_dpp(double v) {
    auto halves = [](double x, uint32_t& lo, uint32_t& hi) { const uint64_t b = __double_as_longlong(x); lo = (uint32_t)b; hi = (uint32_t)(b >> 32); };
    auto join = [](uint32_t lo, uint32_t hi) { return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo)); };
    uint32_t lo, hi;
    {
        halves(v, lo, hi);
        auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
        auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
        v = join(a[0], b[0]) + join(a[1], b[1]);
    }
    {
        halves(v, lo, hi);
        auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
        auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
        v = join(a[0], b[0]) + join(a[1], b[1]);
    }
#define BE_DPP_STEP_D(CTRL)                                                                                          \
    {                                                                                                                \
        halves(v, lo, hi);                                                                                           \
        v += join(__builtin_amdgcn_update_dpp(0u, lo, CTRL, 0xf, 0xf, false), __builtin_amdgcn_update_dpp(0u, hi, CTRL, 0xf, 0xf, false)); \
    }
    BE_DPP_STEP_D(0x128) BE_DPP_STEP_D(0x141) BE_DPP_STEP_D(0x4e) BE_DPP_STEP_D(0xb1)
#undef BE_DPP_STEP_D
    return v;
}

}  // namespace be
