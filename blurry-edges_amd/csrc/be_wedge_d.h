// fp64 wedge geometry + its adjoint, shared by the training-loss kernels (be_local_loss.hip, be_global_loss.hip).
#pragma once
#include "be_wedge.h"

namespace be_d {

constexpr float kInvSqrtPi = 0.56418958354775628695f;

// The training loss is evaluated with fp64 geometry.  An edge can be far sharper than the pixel pitch
// (eta down to 1e-4 against a pitch of 0.1), so the few pixels within ~eta of it carry the whole erf gradient with a
// sensitivity of 1/eta to their own distance: in fp32 the gradient of such a patch is only good to ~1e-3 (the
// reference's own fp32-vs-fp64 autograd gradients differ by that much); fp64 here costs nothing (64 x 441 pixels per
// step) and puts the gradient on the fp64 reference to ~1e-6.
typedef double real;

struct GeomD { real x0, y0, x1, y1, s11, c11, s12, c12, s21, c21, s22, c22, sg1, sg2; };

__device__ __forceinline__ real wrap_2pi_d(real a) {
    const real two_pi = 6.283185307179586476925286766559;
    real r = fmod(a, two_pi);
    if (r < 0.0) r += two_pi;
    return r;
}

// geometry from explicit double parameters (x0,y0,x1,y1,theta1,phi1,theta2,phi2); angles are wrapped here
__device__ __forceinline__ GeomD make_geom_dv(const real* v) {
    const real t1 = wrap_2pi_d(v[4]), f1 = wrap_2pi_d(v[5]), t2 = wrap_2pi_d(v[6]), f2 = wrap_2pi_d(v[7]);
    GeomD g;
    g.x0 = v[0]; g.y0 = v[1]; g.x1 = v[2]; g.y1 = v[3];
    g.sg1 = f1 < 3.14159265358979323846 ? 1.0 : -1.0;
    g.sg2 = f2 < 3.14159265358979323846 ? 1.0 : -1.0;
    g.s11 = sin(t1); g.c11 = cos(t1); g.s12 = sin(t1 + f1); g.c12 = cos(t1 + f1);
    g.s21 = sin(t2); g.c21 = cos(t2); g.s22 = sin(t2 + f2); g.c22 = cos(t2 + f2);
    return g;
}

__device__ __forceinline__ GeomD make_geom_d(const float* p8) {
    // local_training.py:33 wraps the angles first.  Done in fp64 here: the fp32 wrap (modulus float32(2*pi), off by
    // 1.7e-7) moves a wrapped angle by ~2e-7 rad, which a razor-sharp edge turns into a 1e-3 gradient change.
    const real t1 = wrap_2pi_d(p8[4]), f1 = wrap_2pi_d(p8[5]), t2 = wrap_2pi_d(p8[6]), f2 = wrap_2pi_d(p8[7]);
    GeomD g;
    g.x0 = p8[0]; g.y0 = p8[1]; g.x1 = p8[2]; g.y1 = p8[3];
    g.sg1 = f1 < 3.14159265358979323846 ? 1.0 : -1.0;
    g.sg2 = f2 < 3.14159265358979323846 ? 1.0 : -1.0;
    const real a1 = t1, a1p = t1 + f1, a2 = t2, a2p = t2 + f2;
    g.s11 = sin(a1); g.c11 = cos(a1); g.s12 = sin(a1p); g.c12 = cos(a1p);
    g.s21 = sin(a2); g.c21 = cos(a2); g.s22 = sin(a2p); g.c22 = cos(a2p);
    return g;
}

// d(ray distance)/d(edge, axial) for the selected branch of utils/postprocessing_loss.py:67-76
__device__ __forceinline__ void ray_dist_grad(real px, real py, real vx, real vy, real s, real c, real w,
                                              real& dist, real& edge, real& axial, real& d_edge, real& d_axial) {
    const real dx = px - vx, dy = py - vy;
    edge = (-s) * dx + c * dy;
    axial = c * dx + s * dy;
    if (axial < 0.0) {
        const real aw = axial * w;
        const real r = sqrt(edge * edge + aw * aw);
        const real sg = edge < 0.0 ? -1.0 : 1.0;
        dist = sg * r;
        const real ir = r > 0.0 ? 1.0 / r : 0.0;
        d_edge = sg * edge * ir;
        d_axial = sg * w * w * axial * ir;
    } else {
        dist = edge; d_edge = 1.0; d_axial = 0.0;
    }
}

__device__ __forceinline__ real wedge_dist_d(real px, real py, real vx, real vy, real sA, real cA, real sB, real cB,
                                             real sg, bool closed, real w) {
    real dA, eA, aA, deA, daA, dB, eB, aB, deB, daB;
    ray_dist_grad(px, py, vx, vy, sA, cA, w, dA, eA, aA, deA, daA);
    ray_dist_grad(px, py, vx, vy, sB, cB, w, dB, eB, aB, deB, daB);
    const bool inside = closed ? (sg * dA >= 0.0 && sg * dB <= 0.0) : (sg * dA > 0.0 && sg * dB < 0.0);
    return fmin(fabs(dA), fabs(dB)) * (inside ? sg : -sg);
}

// Adjoint of one wedge: accumulates d/d(vx, vy, theta, phi) given dL/d(dist_k) at this pixel.
__device__ __forceinline__ void wedge_backward(real px, real py, real vx, real vy, real sA, real cA, real sB,
                                               real cB, real sg, bool closed, real w, real g_dist,
                                               real& g_vx, real& g_vy, real& g_th, real& g_ph) {
    real dA, eA, aA, deA, daA, dB, eB, aB, deB, daB;
    ray_dist_grad(px, py, vx, vy, sA, cA, w, dA, eA, aA, deA, daA);
    ray_dist_grad(px, py, vx, vy, sB, cB, w, dB, eB, aB, deB, daB);
    const bool inside = closed ? (sg * dA >= 0.0 && sg * dB <= 0.0) : (sg * dA > 0.0 && sg * dB < 0.0);
    const real ind = inside ? sg : -sg;
    const real absA = fabs(dA), absB = fabs(dB);
    // dist = min(|dA|,|dB|) * ind ; ties split evenly (torch.min backward)
    real gA = 0., gB = 0.;
    const real sA_ = dA > 0. ? 1. : (dA < 0. ? -1. : 0.), sB_ = dB > 0. ? 1. : (dB < 0. ? -1. : 0.);
    if (absA < absB) gA = g_dist * ind * sA_;
    else if (absB < absA) gB = g_dist * ind * sB_;
    else { gA = 0.5 * g_dist * ind * sA_; gB = 0.5 * g_dist * ind * sB_; }
    // edge = -s dx + c dy : d/dvx = s, d/dvy = -c, d/dang = -axial ; axial = c dx + s dy : d/dvx = -c, d/dvy = -s, d/dang = edge
    const real gAe = gA * deA, gAa = gA * daA, gBe = gB * deB, gBa = gB * daB;
    g_vx += gAe * sA - gAa * cA + gBe * sB - gBa * cB;
    g_vy += -gAe * cA - gAa * sA - gBe * cB - gBa * sB;
    const real gangA = -gAe * aA + gAa * eA, gangB = -gBe * aB + gBa * eB;
    g_th += gangA + gangB;            // theta feeds both rays (theta and theta + phi)
    g_ph += gangB;                    // phi only the second
}

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}


}  // namespace be_d
