// Winograd transform arithmetic shared by every kernel that applies it (be_wino.hip).  ONE order of floating point operations for
// each transform, written with explicit fmaf and contraction off, so that a patch gets bit-identical results whichever kernel
// family its batch size selects (tests/test_hip_parity.py checks that).
//
// Tile shape (round 4).  A 6x6 map is cut into tiles of BE_WINO_TH x 3 outputs:
//   BE_WINO_TH = 3: F(3,3) along both axes - four 5x5 tiles, 25 positions each: 100 transform-domain values (and multiplies) per
//                   map and channel pair (rounds 1-3);
//   BE_WINO_TH = 6: F(6,3) along the ROWS (8 points: 0, +-1, +-2, +-1/2, inf - the classic set), F(3,3) along the columns - two 8x5
//                   tiles, 40 positions each: 80 values and multiplies, i.e. -20 % work for the matrix pipe AND -20 % transform-domain
//                   HBM traffic.  Price: about twice the rounding error of the 5x5 tiles (lab/wino_tile_error.py: 4-5e-6 against
//                   2.3e-6 per layer, L-inf / L-inf; F(6,3) along BOTH axes would cost 1e-5 and does not fit the 1e-5 tolerance).
// Positions are numbered z = 5 zr + zc (zr: transform row 0..NR-1, zc: transform column 0..4).
#pragma once
#include <hip/hip_runtime.h>

#ifndef BE_WINO_TH
#define BE_WINO_TH 6
#endif

namespace be {

typedef float wf32x4 __attribute__((ext_vector_type(4)));

constexpr int WINO_TH = BE_WINO_TH;             // output rows per tile
constexpr int WINO_NR = WINO_TH + 2;            // transform-domain rows (5 or 8)
constexpr int WINO_TY = 6 / WINO_TH;            // tile rows per 6x6 map (2 or 1)
constexpr int WINO_TPI = 2 * WINO_TY;           // tiles per image (4 or 2)
constexpr int WINO_NPOS = WINO_NR * 5;          // positions = GEMMs per layer (25 or 40)
constexpr int WINO_OUT = WINO_TH * 3;           // outputs per tile (9 or 18)
static_assert(WINO_TH == 3 || WINO_TH == 6, "BE_WINO_TH must be 3 or 6");

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

// B^T (8x8, points 0, +-1, +-2, +-1/2, inf) applied to an 8-vector; rows 1..6 come in (even + odd, even - odd) pairs:
//   o0 = d0 - 21/4 d2 + 21/4 d4 - d6                         o7 = -d1 + 21/4 d3 - 21/4 d5 + d7
//   o1,2 = (d2 - 17/4 d4 + d6) +- (d1 - 17/4 d3 + d5)
//   o3,4 = (1/4 d2 - 5/4 d4 + d6) +- (1/2 d1 - 5/2 d3 + 2 d5)
//   o5,6 = (4 d2 - 5 d4 + d6) +- (2 d1 - 5/2 d3 + 1/2 d5)
__device__ __forceinline__ void wino_bt8(const float d[8], float o[8]) {
#pragma clang fp contract(off)
    o[0] = __builtin_fmaf(5.25f, d[4] - d[2], d[0] - d[6]);
    o[7] = __builtin_fmaf(5.25f, d[3] - d[5], d[7] - d[1]);
    const float e1 = __builtin_fmaf(-4.25f, d[4], d[2] + d[6]);
    const float f1 = __builtin_fmaf(-4.25f, d[3], d[1] + d[5]);
    o[1] = e1 + f1;
    o[2] = e1 - f1;
    const float e2 = __builtin_fmaf(0.25f, d[2], __builtin_fmaf(-1.25f, d[4], d[6]));
    const float f2 = __builtin_fmaf(0.5f, d[1], __builtin_fmaf(-2.5f, d[3], 2.0f * d[5]));
    o[3] = e2 + f2;
    o[4] = e2 - f2;
    const float e3 = __builtin_fmaf(4.0f, d[2], __builtin_fmaf(-5.0f, d[4], d[6]));
    const float f3 = __builtin_fmaf(2.0f, d[1], __builtin_fmaf(-2.5f, d[3], 0.5f * d[5]));
    o[5] = e3 + f3;
    o[6] = e3 - f3;
}

// the row-direction (vertical) input transform of one column of the window: NR values in, NR out
__device__ __forceinline__ void wino_bt_rows(const float d[WINO_NR], float o[WINO_NR]) {
    if constexpr (WINO_NR == 5) wino_bt5(d[0], d[1], d[2], d[3], d[4], o);
    else wino_bt8(d, o);
}

// Input transform of one NR x 5 window d[row][col] -> v[5 r + c] = (B_r^T d B_c)[r][c]: columns first (the vertical transform of
// each of the five columns), then rows (the 5-point transform of each transform row).
// V = a float vector type (4 or 2 channels per thread: the arithmetic is per channel, so the width changes no bit)
template <class V>
__device__ __forceinline__ void wino_in(const V d[WINO_NR][5], V v[WINO_NPOS]) {
#pragma unroll
    for (int k = 0; k < (int)(sizeof(V) / sizeof(float)); ++k) {
        float t[WINO_NR][5];
#pragma unroll
        for (int c = 0; c < 5; ++c) {
            float col[WINO_NR], o[WINO_NR];
#pragma unroll
            for (int r = 0; r < WINO_NR; ++r) col[r] = d[r][c][k];
            wino_bt_rows(col, o);
#pragma unroll
            for (int r = 0; r < WINO_NR; ++r) t[r][c] = o[r];
        }
#pragma unroll
        for (int r = 0; r < WINO_NR; ++r) {
            float o[5];
            wino_bt5(t[r][0], t[r][1], t[r][2], t[r][3], t[r][4], o);
#pragma unroll
            for (int c = 0; c < 5; ++c) v[5 * r + c][k] = o[c];
        }
    }
}

// A^T (3x5):  [1 1 1 1 0; 0 1 -1 2 0; 0 1 1 4 1]
__host__ __device__ constexpr float wino_at(int o, int z) {
    return o == 0 ? (z < 4 ? 1.0f : 0.0f)
         : o == 1 ? (z == 1 ? 1.0f : z == 2 ? -1.0f : z == 3 ? 2.0f : 0.0f)
                  : (z == 0 ? 0.0f : z == 3 ? 4.0f : 1.0f);
}

// ---- BE_WINO_TH == 3: Y[3 r + c] = sum_z A^T[r][z1] A^T[c][z2] M[z], z = 5 z1 + z2, accumulated position by position in ascending
// z with one fmaf each (every coefficient is +-2^k, so each step is the correctly rounded Y + coef * M) - rounds 1-3's order, kept
template <int Z>
__device__ __forceinline__ void wino_out_step(float m, float Y[9]) {
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float coef = wino_at(r, Z / 5) * wino_at(c, Z % 5);
            if (coef != 0.0f) Y[3 * r + c] = __builtin_fmaf(coef, m, Y[3 * r + c]);
        }
}
template <int Z = 0>
__device__ __forceinline__ void wino_out_all(const float m[25], float Y[9]) {
    wino_out_step<Z>(m[Z], Y);
    if constexpr (Z < 24) wino_out_all<Z + 1>(m, Y);
}

// ---- BE_WINO_TH == 6: separable.  Columns first - each of the 8 transform rows gives 3 values through A^T (3x5) - then rows:
// each of the 3 columns gives 6 outputs through A^T (6x8):
//   y0 = m0 + (m1 + m2) + (m3 + m4) + (m5 + m6)              y1 = (m1 - m2) + 2 (m3 - m4) + 1/2 (m5 - m6)
//   y2 = (m1 + m2) + 4 (m3 + m4) + 1/4 (m5 + m6)             y3 = (m1 - m2) + 8 (m3 - m4) + 1/8 (m5 - m6)
//   y4 = (m1 + m2) + 16 (m3 + m4) + 1/16 (m5 + m6)           y5 = (m1 - m2) + 32 (m3 - m4) + 1/32 (m5 - m6) + m7
// (all coefficients are powers of two: every fmaf below rounds once)
__device__ __forceinline__ void wino_at5(const float m[5], float o[3]) {
#pragma clang fp contract(off)
    o[0] = ((m[0] + m[1]) + m[2]) + m[3];
    o[1] = __builtin_fmaf(2.0f, m[3], m[1] - m[2]);
    o[2] = __builtin_fmaf(4.0f, m[3], m[1] + m[2]) + m[4];
}
__device__ __forceinline__ void wino_at8(const float m[8], float o[6]) {
#pragma clang fp contract(off)
    const float s12 = m[1] + m[2], d12 = m[1] - m[2], s34 = m[3] + m[4], d34 = m[3] - m[4], s56 = m[5] + m[6], d56 = m[5] - m[6];
    o[0] = ((m[0] + s12) + s34) + s56;
    o[1] = __builtin_fmaf(0.5f, d56, __builtin_fmaf(2.0f, d34, d12));
    o[2] = __builtin_fmaf(0.25f, s56, __builtin_fmaf(4.0f, s34, s12));
    o[3] = __builtin_fmaf(0.125f, d56, __builtin_fmaf(8.0f, d34, d12));
    o[4] = __builtin_fmaf(0.0625f, s56, __builtin_fmaf(16.0f, s34, s12));
    o[5] = __builtin_fmaf(0.03125f, d56, __builtin_fmaf(32.0f, d34, d12)) + m[7];
}

// the NPOS transform-domain values of one tile -> its TH x 3 output block, y[3 r + c] (vector form: one channel quad per thread)
template <class V>
__device__ __forceinline__ void wino_out(const V m[WINO_NPOS], V y[WINO_OUT]) {
#pragma unroll
    for (int k = 0; k < (int)(sizeof(V) / sizeof(float)); ++k) {
        if constexpr (WINO_TH == 3) {
            float mm[25], Y[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int z = 0; z < 25; ++z) mm[z] = m[z][k];
            wino_out_all<0>(mm, Y);
#pragma unroll
            for (int o = 0; o < 9; ++o) y[o][k] = Y[o];
        } else {
            float t[8][3];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const float row[5] = {m[5 * r][k], m[5 * r + 1][k], m[5 * r + 2][k], m[5 * r + 3][k], m[5 * r + 4][k]};
                wino_at5(row, t[r]);
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float col[8] = {t[0][c], t[1][c], t[2][c], t[3][c], t[4][c], t[5][c], t[6][c], t[7][c]};
                float o[6];
                wino_at8(col, o);
#pragma unroll
                for (int r = 0; r < 6; ++r) y[3 * r + c][k] = o[r];
            }
        }
    }
}

// G applied along the kernel ROWS for the weight pack: 3 taps (a, b, c) of one kernel column -> NR values.
//   5x3 (points 0, 1, -1, 2, inf):  1/2 [1 0 0], -1/2 [1 1 1], -1/6 [1 -1 1], 1/6 [1 2 4], [0 0 1]
//   8x3 (points 0, +-1, +-2, +-1/2, inf):  [1 0 0], -2/9 [1 1 1], -2/9 [1 -1 1], 1/90 [1 2 4], 1/90 [1 -2 4], 32/45 [1 1/2 1/4],
//                                         32/45 [1 -1/2 1/4], [0 0 1]
__device__ __forceinline__ void wino_g5(const float a, const float b, const float c, float out[5]) {
    out[0] = 0.5f * a;
    out[1] = -0.5f * (a + b + c);
    out[2] = -(a - b + c) * (1.0f / 6.0f);
    out[3] = (a + 2.0f * b + 4.0f * c) * (1.0f / 6.0f);
    out[4] = c;
}
__device__ __forceinline__ void wino_g_rows(const float a, const float b, const float c, float out[WINO_NR]) {
    if constexpr (WINO_NR == 5) {
        wino_g5(a, b, c, out);
    } else {
        // evaluated in double and rounded once: the pack runs once per weight version, and 1/90 / 32/45 are not float32 numbers
        const double A = a, B = b, C = c;
        out[0] = a;
        out[1] = (float)(-(A + B + C) * (2.0 / 9.0));
        out[2] = (float)(-(A - B + C) * (2.0 / 9.0));
        out[3] = (float)((A + 2.0 * B + 4.0 * C) * (1.0 / 90.0));
        out[4] = (float)((A - 2.0 * B + 4.0 * C) * (1.0 / 90.0));
        out[5] = (float)((A + 0.5 * B + 0.25 * C) * (32.0 / 45.0));
        out[6] = (float)((A - 0.5 * B + 0.25 * C) * (32.0 / 45.0));
        out[7] = c;
    }
}

}  // namespace be
