// GlobalLoss forward + analytic backward in one launch, one wavefront per patch position (gfx950).
//
// Replaces GlobalLoss.get_patches + get_loss (global_training.py:69-139) and the autograd graph under them.
// The reference detaches the folded global image / boundary before the consistency terms (:94,:100,:106), so once
// the folded maps exist (be_render_full_f32 + be_fold_records_f32 + be_image_derivative_f32) every term is local
// to a patch:
//   color, color_cons  : sum_c (P - gt)^2, sum_c (P - G)^2               per aperture image, per pixel
//   smthns, smthns_cons: sum_c (Sobel(P) - deri_gt)^2, (Sobel(P) - Sobel(G))^2   per aperture, 19x19 interior
//   bndry_cons, bndry_loc: (B - Gb)^2, (log2(bdist+1) B)^2
//   depth              : ((depth_map - bndry_depth) mask)^2, normalised by the GLOBAL mask count: the kernel
//                        returns numerator, count and the un-normalised gradient separately; the host combines.
// Nothing is unfolded: the "patches" of the image-sized tensors are gathered on read at (stride*i + r, stride*j + c).
// Geometry in fp64 (be_wedge_d.h); one colour set shared by the two apertures (882-row ridge system).
#include "be_common.h"
#include "be_wedge_d.h"

namespace {

using namespace be_d;
constexpr int NPIX = BE_NPIX, R = BE_R, PASSES = 7, WAVES = 4, Q = 19, NQ = Q * Q;

struct GLArgs {
    const float* est;        // [B,P,12] raw GlobalStage output
    const float* img_fit;    // [B,2,H,W,3] channels-last
    const float* img_gt;     // [B,2,H,W,3]
    const float* G;          // [B,2,3,H,W]   folded current image (detached)
    const float* Gd;         // [B,2,3,H-2,W-2] its Sobel magnitude
    const float* Gb;         // [B,H,W]       folded current boundary map
    const float* bdist;      // [B,H,W]
    const float* deri;       // [B,2,H-2,W-2,3] channels-last
    const float* bdepth;     // [B,H,W]
    float* partial;          // [B*P,8]: six term sums, depth numerator, mask count
    float* grad;             // [B*P,12] gradient of the six mean terms (scaled)
    float* gdepth;           // [B*P,4]  gradient of the depth NUMERATOR w.r.t. est[8:12] (host scales by gamma/count)
    be_depth_consts dc;
    float wc, wcc, wbc, ws, wsc, wbl;      // gamma_k / N_k
    int B, P, hp, wp, H, W, stride;
};

// d depth / d (eta_a, eta_b) of utils/depth_etas.py:23-34 for the branch taken
__device__ __forceinline__ real depth_and_grad(const be_depth_consts& c, real e1, real e2, real& dz1, real& dz2) {
    const real I = c.intercept;
    const real c1 = -(real)c.sin_w * e1 + (real)c.cos_w * (e2 - I);
    const real c2 = -(real)c.sin_m * (e1 - I) + (real)c.cos_m * e2;
    const real c3 = -(real)c.sin_w * (e1 - I) + (real)c.cos_w * e2;
    real e11, e22, a11, b11, a22, b22;                         // e11 = ..., d e11/d e1 = a11, d e11/d e2 = b11
    if (c1 > 0.0)      { e11 = (e1 + e2 - I) / 2;       e22 = I + (e1 + e2 - I) / 2;  a11 = .5; b11 = .5;  a22 = .5;  b22 = .5; }
    else if (c2 > 0.0) { e11 = I + (e1 - e2 - I) / 2;   e22 = (e2 - e1 + I) / 2;      a11 = .5; b11 = -.5; a22 = -.5; b22 = .5; }
    else if (c3 < 0.0) { e11 = I + (e1 + e2 - I) / 2;   e22 = (e1 + e2 - I) / 2;      a11 = .5; b11 = .5;  a22 = .5;  b22 = .5; }
    else               { e11 = e1; e22 = e2;                                          a11 = 1.; b11 = 0.;  a22 = 0.;  b22 = 1.; }
    const real den = (real)c.k2 * (e11 * e11 - e22 * e22) + (real)c.den_const;
    const real z = (real)c.numerator / den;
    const real k = -(real)c.numerator * (real)c.k2 * 2.0 / (den * den);
    dz1 = k * (e11 * a11 - e22 * a22);
    dz2 = k * (e11 * b11 - e22 * b22);
    return z;
}

__global__ __launch_bounds__(64 * WAVES)
void k_global_loss(be_render_opts o, GLArgs a) {
    __shared__ float lin[R];
    __shared__ float sPatch[WAVES][3][NPIX];
    __shared__ float sDx[WAVES][3][NQ];
    __shared__ float sDy[WAVES][3][NQ];
    if (threadIdx.x < R) lin[threadIdx.x] = o.lin[threadIdx.x];
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t total = (int64_t)a.B * a.P;
    const int64_t gp_raw = (int64_t)blockIdx.x * WAVES + wv;
    const bool active = gp_raw < total;
    const int64_t gp = active ? gp_raw : total - 1;
    const int b = (int)(gp / a.P), p = (int)(gp % a.P);
    const int pi = p / a.wp, pj = p % a.wp;
    const int y0 = a.stride * pi, x0 = a.stride * pj;
    const size_t HW = (size_t)a.H * a.W, HWd = (size_t)(a.H - 2) * (a.W - 2);

    // ---- parameters (global_training.py:141-145) and their chain factors
    const float* e = a.est + gp * 12;
    real v[8];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = 3.0 * (real)e[k];
#pragma unroll
    for (int k = 4; k < 8; ++k) v[k] = ((real)e[k] + 1.0) * 3.14159265358979323846;
    const GeomD g = make_geom_dv(v);
    real eta[4], rad[4], deta[4];                               // d eta / d est
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const real pk = (real)e[8 + k] + 0.5;
        eta[k] = pow(10.0, 2.0 * erf(pk) - 2.0);
        rad[k] = (real)be::kRoot2 * eta[k];
        deta[k] = eta[k] * 2.302585092994046 * 4.0 * (real)kInvSqrtPi * exp(-pk * pk);
    }

    // ---- pass 1: distances, indicators of both blur sets, normal equations over the 882 rows
    real d1s[PASSES], d2s[PASSES];
    float h[2][2][PASSES];                                      // [set][wedge][pass]
    float gs[6] = {0, 0, 0, 0, 0, 0}, bs[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int it = 0; it < PASSES; ++it) {
        const int pix = it * 64 + lane;
        const bool live = pix < NPIX;
        const int pc = live ? pix : 0;
        const int row = pc / R, col = pc - row * R;
        const real d1 = wedge_dist_d(lin[col], lin[row], g.x0, g.y0, g.s11, g.c11, g.s12, g.c12, g.sg1, false, o.w);
        const real d2 = wedge_dist_d(lin[col], lin[row], g.x1, g.y1, g.s21, g.c21, g.s22, g.c22, g.sg2, true, o.w);
        d1s[it] = d1; d2s[it] = d2;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const float h1 = (float)(0.5 * (1.0 + erf(d1 / rad[2 * s]))), h2 = (float)(0.5 * (1.0 + erf(d2 / rad[2 * s + 1])));
            h[s][0][it] = h1; h[s][1][it] = h2;
            const float u0 = live ? (1.0f - h1) * (1.0f - h2) : 0.f, u1 = live ? h1 * (1.0f - h2) : 0.f, u2 = live ? h2 : 0.f;
            const float* src = a.img_fit + (((size_t)(b * 2 + s) * a.H + y0 + row) * a.W + x0 + col) * 3;
            const float yr = live ? src[0] : 0.f, yg = live ? src[1] : 0.f, yb = live ? src[2] : 0.f;
            gs[0] = fmaf(u0, u0, gs[0]); gs[1] = fmaf(u0, u1, gs[1]); gs[2] = fmaf(u0, u2, gs[2]);
            gs[3] = fmaf(u1, u1, gs[3]); gs[4] = fmaf(u1, u2, gs[4]); gs[5] = fmaf(u2, u2, gs[5]);
            bs[0] = fmaf(u0, yr, bs[0]); bs[1] = fmaf(u0, yg, bs[1]); bs[2] = fmaf(u0, yb, bs[2]);
            bs[3] = fmaf(u1, yr, bs[3]); bs[4] = fmaf(u1, yg, bs[4]); bs[5] = fmaf(u1, yb, bs[5]);
            bs[6] = fmaf(u2, yr, bs[6]); bs[7] = fmaf(u2, yg, bs[7]); bs[8] = fmaf(u2, yb, bs[8]);
        }
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) gs[k] = be::wave_sum(gs[k]);
#pragma unroll
    for (int k = 0; k < 9; ++k) bs[k] = be::wave_sum(bs[k]);
    const double A00 = (double)gs[0] + o.lambda_ridge, A01 = gs[1], A02 = gs[2], A11 = (double)gs[3] + o.lambda_ridge,
                 A12 = gs[4], A22 = (double)gs[5] + o.lambda_ridge;
    double inv[3][3];
    {
        const double C00 = A11 * A22 - A12 * A12, C01 = A02 * A12 - A01 * A22, C02 = A01 * A12 - A02 * A11;
        const double C11 = A00 * A22 - A02 * A02, C12 = A01 * A02 - A00 * A12, C22 = A00 * A11 - A01 * A01;
        const double idet = 1.0 / (A00 * C00 + A01 * C01 + A02 * C02);
        inv[0][0] = C00 * idet; inv[0][1] = inv[1][0] = C01 * idet; inv[0][2] = inv[2][0] = C02 * idet;
        inv[1][1] = C11 * idet; inv[1][2] = inv[2][1] = C12 * idet; inv[2][2] = C22 * idet;
    }
    float Cc[3][3];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int k = 0; k < 3; ++k)
            Cc[c][k] = (float)(inv[k][0] * bs[c] + inv[k][1] * bs[3 + c] + inv[k][2] * bs[6 + c]);

    // ---- pass 2 (per blur set): composite -> colour terms; Sobel terms and their adjoint gathered back
    float gP[2][PASSES][3];
    float T1 = 0, T2 = 0, T4 = 0, T5 = 0;
    for (int s = 0; s < 2; ++s) {
        const float* gt = a.img_gt + ((size_t)(b * 2 + s) * HW) * 3;
        const float* Gs = a.G + (size_t)(b * 2 + s) * 3 * HW;
#pragma unroll
        for (int it = 0; it < PASSES; ++it) {
            const int pix = it * 64 + lane;
            const bool live = pix < NPIX;
            const int pc = live ? pix : 0;
            const int row = pc / R, col = pc - row * R;
            const size_t at = (size_t)(y0 + row) * a.W + x0 + col;
            const float h1 = h[s][0][it], h2 = h[s][1][it];
            const float u0 = (1.0f - h1) * (1.0f - h2), u1 = h1 * (1.0f - h2), u2 = h2;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float val = u0 * Cc[c][0] + u1 * Cc[c][1] + u2 * Cc[c][2];
                float r1 = 0.f, r2 = 0.f;
                if (live) {
                    sPatch[wv][c][pix] = val;
                    r1 = val - gt[at * 3 + c];
                    r2 = val - Gs[c * HW + at];
                }
                T1 = fmaf(r1, r1, T1); T2 = fmaf(r2, r2, T2);
                gP[s][it][c] = 2.0f * (r1 * a.wc + r2 * a.wcc);
            }
        }
        __syncthreads();
        const float* dr = a.deri + ((size_t)(b * 2 + s) * HWd) * 3;
        const float* Gds = a.Gd + (size_t)(b * 2 + s) * 3 * HWd;
        for (int q0 = 0; q0 < NQ; q0 += 64) {
            const int q = q0 + lane;
            if (q < NQ) {
                const int qy = q / Q, qx = q - qy * Q;
                const size_t atd = (size_t)(y0 + qy) * (a.W - 2) + x0 + qx;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float* Pp = sPatch[wv][c] + qy * R + qx;
                    const float p00 = Pp[0], p01 = Pp[1], p02 = Pp[2], p10 = Pp[R], p12 = Pp[R + 2], p20 = Pp[2 * R],
                                p21 = Pp[2 * R + 1], p22 = Pp[2 * R + 2];
                    const float gx = (p02 - p00) + 2.0f * (p12 - p10) + (p22 - p20);
                    const float gy = (p00 - p20) + 2.0f * (p01 - p21) + (p02 - p22);
                    const float sm = sqrtf(gx * gx + gy * gy + 1e-8f);
                    const float t4 = sm - dr[atd * 3 + c], t5 = sm - Gds[c * HWd + atd];
                    T4 = fmaf(t4, t4, T4); T5 = fmaf(t5, t5, T5);
                    const float k = 2.0f * (t4 * a.ws + t5 * a.wsc) / sm;
                    sDx[wv][c][q] = k * gx;
                    sDy[wv][c][q] = k * gy;
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < PASSES; ++it) {
            const int pix = it * 64 + lane;
            if (pix < NPIX) {
                const int row = pix / R, col = pix - row * R;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    float acc = 0.f;
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy) {
                        const int qy = row - dy;
                        if (qy < 0 || qy >= Q) continue;
#pragma unroll
                        for (int dx = 0; dx < 3; ++dx) {
                            const int qx = col - dx;
                            if (qx < 0 || qx >= Q) continue;
                            const float kx = (dx == 0 ? -1.f : (dx == 2 ? 1.f : 0.f)) * (dy == 1 ? 2.f : 1.f);
                            const float ky = (dy == 0 ? 1.f : (dy == 2 ? -1.f : 0.f)) * (dx == 1 ? 2.f : 1.f);
                            acc += kx * sDx[wv][c][qy * Q + qx] + ky * sDy[wv][c][qy * Q + qx];
                        }
                    }
                    gP[s][it][c] += acc;
                }
            }
        }
        __syncthreads();                                        // LDS is reused by the second blur set
    }

    // ---- adjoint of the shared colour solve
    float dC[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int it = 0; it < PASSES; ++it) {
            const bool live = it * 64 + lane < NPIX;
            const float h1 = h[s][0][it], h2 = h[s][1][it];
            const float u[3] = {live ? (1.0f - h1) * (1.0f - h2) : 0.f, live ? h1 * (1.0f - h2) : 0.f, live ? h2 : 0.f};
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int k = 0; k < 3; ++k) dC[c * 3 + k] = fmaf(gP[s][it][c], u[k], dC[c * 3 + k]);
        }
#pragma unroll
    for (int k = 0; k < 9; ++k) dC[k] = be::wave_sum(dC[k]);
    float V[3][3], S[3][3];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int k = 0; k < 3; ++k)
            V[c][k] = (float)(inv[k][0] * dC[c * 3] + inv[k][1] * dC[c * 3 + 1] + inv[k][2] * dC[c * 3 + 2]);
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            float dkj = 0.f, djk = 0.f;
#pragma unroll
            for (int c = 0; c < 3; ++c) { dkj -= V[c][k] * Cc[c][j]; djk -= V[c][j] * Cc[c][k]; }
            S[k][j] = dkj + djk;
        }

    // ---- depth of the two wedges and its derivative w.r.t. the four etas
    real dz1a, dz1b, dz2a, dz2b;
    const real z1 = depth_and_grad(a.dc, eta[0], eta[2], dz1a, dz1b);
    const real z2 = depth_and_grad(a.dc, eta[1], eta[3], dz2a, dz2b);

    // ---- pass 3: per-pixel adjoints down to the twelve parameters
    real gx0 = 0, gy0 = 0, gt1 = 0, gf1 = 0, gx1 = 0, gy1 = 0, gt2 = 0, gf2 = 0, gr[4] = {0, 0, 0, 0};
    real gz1 = 0, gz2 = 0;
    float T3 = 0, T6 = 0, T7 = 0, MS = 0;
#pragma unroll
    for (int it = 0; it < PASSES; ++it) {
        const int pix = it * 64 + lane;
        if (pix < NPIX) {
            const int row = pix / R, col = pix - row * R;
            const real px = lin[col], py = lin[row];
            const real d1 = d1s[it], d2 = d2s[it];
            const size_t at = (size_t)(y0 + row) * a.W + x0 + col;
            real gd1 = 0, gd2 = 0;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const float h1 = h[s][0][it], h2 = h[s][1][it];
                const float u[3] = {(1.0f - h1) * (1.0f - h2), h1 * (1.0f - h2), h2};
                const float* src = a.img_fit + ((size_t)(b * 2 + s) * HW + at) * 3;
                const float y[3] = {src[0], src[1], src[2]};
                float du[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    float acc = 0.f;
#pragma unroll
                    for (int c = 0; c < 3; ++c) acc += gP[s][it][c] * Cc[c][k] + V[c][k] * y[c];
#pragma unroll
                    for (int j = 0; j < 3; ++j) acc += S[k][j] * u[j];
                    du[k] = acc;
                }
                const real dh1 = (du[1] - du[0]) * (1.0f - h2);
                const real dh2 = -du[0] * (1.0f - h1) - du[1] * h1 + du[2];
                const real ra = rad[2 * s], rb = rad[2 * s + 1];
                const real za = d1 / ra, zb = d2 / rb;
                const real ea = exp(-za * za) * (real)kInvSqrtPi, eb = exp(-zb * zb) * (real)kInvSqrtPi;
                gd1 += dh1 * ea / ra; gd2 += dh2 * eb / rb;
                gr[2 * s] -= dh1 * ea * d1 / (ra * ra);
                gr[2 * s + 1] -= dh2 * eb * d2 / (rb * rb);
            }
            // boundary terms (:99-103, :121-125)
            const real a1 = fabs(d1), a2 = fabs(d2);
            const real db = d2 >= 0.0 ? d2 : (a1 < a2 ? a1 : a2);
            const real Bv = exp(-(db * db) / (real)o.delta_sq);
            const size_t atb = (size_t)b * HW + at;
            const real r3 = Bv - (real)a.Gb[atb];
            const real lb = log2((real)a.bdist[atb] + 1.0);
            T3 += (float)(r3 * r3); T6 += (float)((lb * Bv) * (lb * Bv));
            const real gB = 2.0 * r3 * a.wbc + 2.0 * lb * lb * Bv * a.wbl;
            const real gdb = gB * Bv * (-2.0 * db / (real)o.delta_sq);
            if (d2 >= 0.0) gd2 += gdb;
            else if (a1 < a2) gd1 += gdb * (d1 > 0. ? 1. : (d1 < 0. ? -1. : 0.));
            else gd2 += gdb * (d2 > 0. ? 1. : (d2 < 0. ? -1. : 0.));
            // depth term (:127-133): mask as blurry_edges / global_training :83-85
            const bool m1 = exp(-(d1 * d1) / (real)o.delta_sq) > 0.5, m2 = exp(-(d2 * d2) / (real)o.delta_sq) > 0.5;
            const int mk = (m2 || d2 >= 0.0) ? (m2 ? 2 : 0) : (m1 ? 1 : 0);
            const real bdp = a.bdepth[atb];
            if (bdp != 0.0 && mk != 0) {
                const real diff = (mk == 1 ? z1 : z2) - bdp;
                T7 += (float)(diff * diff); MS += 1.0f;
                if (mk == 1) gz1 += 2.0 * diff; else gz2 += 2.0 * diff;
            }
            wedge_backward(px, py, g.x0, g.y0, g.s11, g.c11, g.s12, g.c12, g.sg1, false, o.w, gd1, gx0, gy0, gt1, gf1);
            wedge_backward(px, py, g.x1, g.y1, g.s21, g.c21, g.s22, g.c22, g.sg2, true, o.w, gd2, gx1, gy1, gt2, gf2);
        }
    }
    gx0 = wave_sum_d(gx0); gy0 = wave_sum_d(gy0); gx1 = wave_sum_d(gx1); gy1 = wave_sum_d(gy1);
    gt1 = wave_sum_d(gt1); gf1 = wave_sum_d(gf1); gt2 = wave_sum_d(gt2); gf2 = wave_sum_d(gf2);
#pragma unroll
    for (int k = 0; k < 4; ++k) gr[k] = wave_sum_d(gr[k]);
    gz1 = wave_sum_d(gz1); gz2 = wave_sum_d(gz2);
    T1 = be::wave_sum(T1); T2 = be::wave_sum(T2); T3 = be::wave_sum(T3); T4 = be::wave_sum(T4);
    T5 = be::wave_sum(T5); T6 = be::wave_sum(T6); T7 = be::wave_sum(T7); MS = be::wave_sum(MS);
    if (lane == 0 && active) {
        float* pt = a.partial + gp * 8;
        pt[0] = T1; pt[1] = T2; pt[2] = T3; pt[3] = T4; pt[4] = T5; pt[5] = T6; pt[6] = T7; pt[7] = MS;
        float* go = a.grad + gp * 12;
        const real pi_ = 3.14159265358979323846;
        go[0] = (float)(3.0 * gx0); go[1] = (float)(3.0 * gy0); go[2] = (float)(3.0 * gx1); go[3] = (float)(3.0 * gy1);
        go[4] = (float)(pi_ * gt1); go[5] = (float)(pi_ * gf1); go[6] = (float)(pi_ * gt2); go[7] = (float)(pi_ * gf2);
#pragma unroll
        for (int k = 0; k < 4; ++k) go[8 + k] = (float)(gr[k] * (real)be::kRoot2 * deta[k]);
        float* gd = a.gdepth + gp * 4;                          // depth1 <- (eta0, eta2), depth2 <- (eta1, eta3)
        gd[0] = (float)(gz1 * dz1a * deta[0]); gd[2] = (float)(gz1 * dz1b * deta[2]);
        gd[1] = (float)(gz2 * dz2a * deta[1]); gd[3] = (float)(gz2 * dz2b * deta[3]);
    }
}

}  // namespace

extern "C" int be_global_loss_f32(const be_render_opts* o, const be_depth_consts* dc, const float* est, const float* img_fit,
                                  const float* img_gt, const float* G, const float* Gderi, const float* Gbndry,
                                  const float* bdist, const float* deri, const float* bdepth, const float* gamma6,
                                  float* partial, float* grad, float* grad_depth, int B, int hp, int wp, int H, int W,
                                  int stride, void* stream) {
    BE_REQUIRE(o && dc && est && img_fit && img_gt && G && Gderi && Gbndry && bdist && deri && bdepth && gamma6 && partial &&
               grad && grad_depth, "be_global_loss_f32: null pointer");
    BE_REQUIRE(B > 0 && hp > 0 && wp > 0 && stride > 0, "be_global_loss_f32: bad sizes");
    BE_REQUIRE(stride * (hp - 1) + R <= H && stride * (wp - 1) + R <= W, "be_global_loss_f32: patch grid exceeds the image");
    GLArgs a;
    a.est = est; a.img_fit = img_fit; a.img_gt = img_gt; a.G = G; a.Gd = Gderi; a.Gb = Gbndry; a.bdist = bdist; a.deri = deri;
    a.bdepth = bdepth; a.partial = partial; a.grad = grad; a.gdepth = grad_depth; a.dc = *dc;
    const int P = hp * wp;
    const double n1 = (double)B * 2 * NPIX * P, n3 = (double)B * NPIX * P, n4 = (double)B * 2 * NQ * P;
    a.wc = (float)(gamma6[0] / n1); a.wcc = (float)(gamma6[1] / n1); a.wbc = (float)(gamma6[2] / n3);
    a.ws = (float)(gamma6[3] / n4); a.wsc = (float)(gamma6[4] / n4); a.wbl = (float)(gamma6[5] / n3);
    a.B = B; a.P = P; a.hp = hp; a.wp = wp; a.H = H; a.W = W; a.stride = stride;
    const int64_t blocks = ((int64_t)B * P + WAVES - 1) / WAVES;
    hipLaunchKernelGGL(k_global_loss, dim3((unsigned)blocks), dim3(64 * WAVES), 0, be::as_stream(stream), *o, a);
    return be::check_launch("be_global_loss_f32");
}
