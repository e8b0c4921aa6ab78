// Wedge geometry and soft-indicator math shared by the render kernels (device side).
// Reference: utils/postprocessing_loss.py:26-30,43-95 ; blurry_edges_test.py:47-61.
// Everything here keeps the reference's operation order with contraction off, so hard decisions
// (inside tests, masks) agree with the PyTorch-CPU path on identical inputs.
#pragma once
#include "be_device_math.h"

namespace be {

constexpr float kRoot2 = 1.41421353816986083984375f;     // float32(sqrt(2)), :92
constexpr float kPi = 3.14159265358979323846f;

// One patch's boundary structure: two vertices, four ray directions, two inside-signs.
struct WedgeGeom {
    float x0, y0, x1, y1;
    float s11, c11, s12, c12, s21, c21, s22, c22;
    float sg1, sg2;
};

// p8 = (x0,y0,x1,y1,theta1,phi1,theta2,phi2); wrap: angles <- remainder(., 2pi) first
__device__ __forceinline__ WedgeGeom make_geom(const float* p8, bool wrap) {
#pragma clang fp contract(off)
    WedgeGeom g;
    g.x0 = p8[0]; g.y0 = p8[1]; g.x1 = p8[2]; g.y1 = p8[3];
    float t1 = p8[4], f1 = p8[5], t2 = p8[6], f2 = p8[7];
    if (wrap) { t1 = remainder_2pi(t1); f1 = remainder_2pi(f1); t2 = remainder_2pi(t2); f2 = remainder_2pi(f2); }
    g.sg1 = remainder_2pi(f1) < kPi ? 1.0f : -1.0f;          // :46
    g.sg2 = remainder_2pi(f2) < kPi ? 1.0f : -1.0f;          // :47
    const float t1p = t1 + f1, t2p = t2 + f2;                 // :49-50
    g.s11 = sinf(t1);  g.c11 = cosf(t1);  g.s12 = sinf(t1p); g.c12 = cosf(t1p);
    g.s21 = sinf(t2);  g.c21 = cosf(t2);  g.s22 = sinf(t2p); g.c22 = cosf(t2p);
    return g;
}

// signed distance to one ray (:26-30, :50-76)
__device__ __forceinline__ float ray_dist(float px, float py, float vx, float vy, float s, float c, float w) {
#pragma clang fp contract(off)
    const float dx = px - vx, dy = py - vy;
    const float edge = (-s) * dx + c * dy;
    const float axial = c * dx + s * dy;
    if (axial < 0.0f) {
        const float aw = axial * w;
        const float r = sqrtf(edge * edge + aw * aw);
        return edge < 0.0f ? -r : r;
    }
    return edge;
}

// signed distances of a pixel to the two wedges (:78-86)
__device__ __forceinline__ void wedge_dists(const WedgeGeom& g, float px, float py, float w, float& d1, float& d2) {
#pragma clang fp contract(off)
    const float d11 = ray_dist(px, py, g.x0, g.y0, g.s11, g.c11, w);
    const float d12 = ray_dist(px, py, g.x0, g.y0, g.s12, g.c12, w);
    const float d21 = ray_dist(px, py, g.x1, g.y1, g.s21, g.c21, w);
    const float d22 = ray_dist(px, py, g.x1, g.y1, g.s22, g.c22, w);
    const float in1 = (g.sg1 * d11 > 0.0f && g.sg1 * d12 < 0.0f) ? g.sg1 : -g.sg1;     // strict :80
    const float in2 = (g.sg2 * d21 >= 0.0f && g.sg2 * d22 <= 0.0f) ? g.sg2 : -g.sg2;   // closed :81
    d1 = fminf(fabsf(d11), fabsf(d12)) * in1;
    d2 = fminf(fabsf(d21), fabsf(d22)) * in2;
}

// (u0,u1,u2) from distances and the two scaled blur radii r_k = sqrt(2)*eta_k  (:91-95)
__device__ __forceinline__ void indicators(float d1, float d2, float r1, float r2, float& u0, float& u1, float& u2) {
#pragma clang fp contract(off)
    const float h1 = 0.5f * (1.0f + erff(d1 / r1));
    const float h2 = 0.5f * (1.0f + erff(d2 / r2));
    u0 = (1.0f - h1) * (1.0f - h2);
    u1 = h1 * (1.0f - h2);
    u2 = h2;
}

// boundary response exp(-d_B^2/delta^2)  (blurry_edges_test.py:59-61 ; local_training.py:42-44)
__device__ __forceinline__ float boundary_value(float d1, float d2, float delta_sq) {
#pragma clang fp contract(off)
    const float a1 = fabsf(d1), a2 = fabsf(d2);
    const float db = d2 >= 0.0f ? d2 : (a1 < a2 ? a1 : a2);
    return expf(-(db * db) / delta_sq);
}

// depth mask in {0,1,2}  (blurry_edges_test.py:47-54); densify_w: the '--densify w' rule
__device__ __forceinline__ int depth_mask(float d1, float d2, float delta_sq, bool densify_w) {
#pragma clang fp contract(off)
    if (densify_w) return d2 > 0.0f ? 2 : (d1 > 0.0f ? 1 : 0);
    const bool m1 = expf(-(d1 * d1) / delta_sq) > 0.5f;
    const bool m2 = expf(-(d2 * d2) / delta_sq) > 0.5f;
    return (m2 || d2 >= 0.0f) ? (m2 ? 2 : 0) : (m1 ? 1 : 0);
}

// 3x3 SPD solve by cofactors in fp64: col[c][k] = sum_j inv(G)[k][j] * b[j][c]
struct Colors9 { float c[9]; };   // [rgb][wedge]
__device__ __forceinline__ Colors9 solve_colors(float g00, float g01, float g02, float g11, float g12, float g22,
                                                const float (&b)[9] /* [wedge][rgb] */) {
    const double A00 = g00, A01 = g01, A02 = g02, A11 = g11, A12 = g12, A22 = g22;
    const double C00 = A11 * A22 - A12 * A12, C01 = A02 * A12 - A01 * A22, C02 = A01 * A12 - A02 * A11;
    const double C11 = A00 * A22 - A02 * A02, C12 = A01 * A02 - A00 * A12, C22 = A00 * A11 - A01 * A01;
    const double idet = 1.0 / (A00 * C00 + A01 * C01 + A02 * C02);
    Colors9 o;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        const double b0 = b[ch], b1 = b[3 + ch], b2 = b[6 + ch];
        o.c[ch * 3 + 0] = (float)((C00 * b0 + C01 * b1 + C02 * b2) * idet);
        o.c[ch * 3 + 1] = (float)((C01 * b0 + C11 * b1 + C12 * b2) * idet);
        o.c[ch * 3 + 2] = (float)((C02 * b0 + C12 * b1 + C22 * b2) * idet);
    }
    return o;
}

}  // namespace be
