// Winograd F(3x3,3x3) transform arithmetic shared by every kernel that applies it (be_wino.hip: the stand-alone transform
// kernels of small batches; be_wino_fused.hip: the GEMM whose epilogue transforms in registers).  ONE order of floating
// point operations for each transform, written with explicit fmaf and contraction off, so that a patch gets bit-identical
// results whichever kernel family its batch size selects (tests/test_hip_parity.py checks that).
#pragma once
#include <hip/hip_runtime.h>

namespace be {

typedef float wf32x4 __attribute__((ext_vector_type(4)));

// B^T (5x5, interpolation points 0, 1, -1, 2, inf) applied to a 5-vector:
//   o0 = 2 d0 - d1 - 2 d2 + d3      o1 = -2 d1 - d2 + d3      o2 = 2 d1 - 3 d2 + d3      o3 = -d1 + d3      o4 = 2 d1 - d2 - 2 d3 + d4
__device__ __forceinline__ void wino_bt5(float d0, float d1, float d2, float d3, float d4, float o[5]) {
#pragma clang fp contract(off)
    o[0] = __builtin_fmaf(-2.0f, d2, __builtin_fmaf(2.0f, d0, -d1)) + d3;
    o[1] = __builtin_fmaf(-2.0f, d1, d3) - d2;
    o[2] = __builtin_fmaf(2.0f, d1, __builtin_fmaf(-3.0f, d2, d3));
    o[3] = d3 - d1;
    o[4] = __builtin_fmaf(-2.0f, d3, __builtin_fmaf(2.0f, d1, -d2)) + d4;
}

__device__ __forceinline__ void wino_bt5(const wf32x4 d0, const wf32x4 d1, const wf32x4 d2, const wf32x4 d3, const wf32x4 d4,
                                         wf32x4 o[5]) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float t[5];
        wino_bt5(d0[k], d1[k], d2[k], d3[k], d4[k], t);
#pragma unroll
        for (int r = 0; r < 5; ++r) o[r][k] = t[r];
    }
}

// Input transform of one 5x5 window d[row][col] -> v[5 r + c] = (B^T d B)[r][c]: columns first, then rows.
template <class T>
__device__ __forceinline__ void wino_in25(const T d[5][5], T v[25]) {
    T t[5][5];
#pragma unroll
    for (int c = 0; c < 5; ++c) {
        T o[5];
        wino_bt5(d[0][c], d[1][c], d[2][c], d[3][c], d[4][c], o);
#pragma unroll
        for (int r = 0; r < 5; ++r) t[r][c] = o[r];
    }
#pragma unroll
    for (int r = 0; r < 5; ++r) {
        T o[5];
        wino_bt5(t[r][0], t[r][1], t[r][2], t[r][3], t[r][4], o);
#pragma unroll
        for (int c = 0; c < 5; ++c) v[5 * r + c] = o[c];
    }
}

// A^T (3x5):  [1 1 1 1 0; 0 1 -1 2 0; 0 1 1 4 1].  Output transform Y[3 r + c] = sum_z A^T[r][z1] A^T[c][z2] M[z], z = 5 z1 + z2,
// accumulated position by position in ascending z with one fmaf each (every coefficient is +-2^k, so each step is the
// correctly rounded Y + coef * M): the GEMM kernel can fold position z into Y the moment its accumulator is complete.
__host__ __device__ constexpr float wino_at(int o, int z) {
    return o == 0 ? (z < 4 ? 1.0f : 0.0f)
         : o == 1 ? (z == 1 ? 1.0f : z == 2 ? -1.0f : z == 3 ? 2.0f : 0.0f)
                  : (z == 0 ? 0.0f : z == 3 ? 4.0f : 1.0f);
}

// Y[9] += coef(z) * m for a compile-time position z
template <int Z>
__device__ __forceinline__ void wino_out_step(float m, float Y[9]) {
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            constexpr float dummy = 0.0f; (void)dummy;
            const float coef = wino_at(r, Z / 5) * wino_at(c, Z % 5);
            if (coef != 0.0f) Y[3 * r + c] = __builtin_fmaf(coef, m, Y[3 * r + c]);
        }
}

template <int Z = 0>
__device__ __forceinline__ void wino_out_all(const float m[25], float Y[9]) {
    wino_out_step<Z>(m[Z], Y);
    if constexpr (Z < 24) wino_out_all<Z + 1>(m, Y);
}

// the 25 transform-domain values of one tile -> its 3x3 output block (vector form: one channel quad per thread)
__device__ __forceinline__ void wino_out9(const wf32x4 m[25], wf32x4 y[9]) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float mm[25], Y[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int z = 0; z < 25; ++z) mm[z] = m[z][k];
        wino_out_all<0>(mm, Y);
#pragma unroll
        for (int o = 0; o < 9; ++o) y[o][k] = Y[o];
    }
}

}  // namespace be
