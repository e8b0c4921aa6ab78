// GlobalLoss forward + analytic backward in one launch, one workgroup of seven wavefronts per patch position (gfx950).
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
// Geometry in fp64 (be_wedge_d.h): the signed distances and the gradient chain through them, because a distance error is
// amplified by 1 / eta (down to 1e-4).  The transcendental functions of the per-pixel work run in fp32 ON fp64-accurate arguments
// (z = d / (sqrt2 eta) is formed in fp64 and rounded once): erf and exp(-z^2) then carry 1e-7 relative error, which nothing amplifies.
// One colour set shared by the two apertures (882-row ridge system).
#include "be_common.h"
#include "be_wedge_d.h"

namespace {

using namespace be_d;
constexpr int NPIX = BE_NPIX, R = BE_R, Q = 19, NQ = Q * Q;

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

// One WORKGROUP of 448 threads (7 waves) per patch position, one pixel per thread (441 live).  The first version gave a patch to
// one wave, seven pixels per lane with every per-pixel quantity of all seven kept in registers across the phases: 512 VGPRs, 310
// of them spilled, one wave per SIMD waiting on its own scratch traffic (3.5 ms per batch of 8 images).  Here a thread's state is
// two fp64 distances, four blur indicators and six colour adjoints; the sums over the patch go through LDS.
constexpr int NT = 448, NWV = NT / 64;

struct PatchScalars {            // computed once per patch (wave 0 / thread 0), read by everyone from LDS: what is uniform over the
    GeomD g;                     // patch does not occupy registers in 448 threads
    real eta[4], irad[4], deta[4], idelta;
    real z1, z2, dz1a, dz1b, dz2a, dz2b;
};
struct SolveScalars { float Cc[3][3], inv[3][3], V[3][3], S[3][3]; };

// sum over the workgroup of n floats per thread; every thread gets the totals.  red: [NWV][n] floats of LDS.
// (each call site owns its `red`: no barrier is needed to protect an earlier use)
template <int N>
__device__ __forceinline__ void block_sum(float (&v)[N], float* red, int lane, int wv) {
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] = be::wave_sum_dpp(v[k]);
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < N; ++k) red[wv * N + k] = v[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < N; ++k) {
        float t = red[k];
#pragma unroll
        for (int w = 1; w < NWV; ++w) t += red[w * N + k];
        v[k] = t;
    }
}
template <int N>
__device__ __forceinline__ void block_sum_d(real (&v)[N], real* red, int lane, int wv) {
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] = be::wave_sum_dpp(v[k]);
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < N; ++k) red[wv * N + k] = v[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < N; ++k) {
        real t = red[k];
#pragma unroll
        for (int w = 1; w < NWV; ++w) t += red[w * N + k];
        v[k] = t;
    }
}

__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(4, 4)))       // 128 registers: two workgroups of 7 waves per CU
void k_global_loss(be_render_opts o, GLArgs a) {
    __shared__ float lin[R];
    __shared__ float sPatch[2][3][NPIX];                       // both blur sets at once: no barrier between the sets
    __shared__ float sDx[2][3][NQ];
    __shared__ float sDy[2][3][NQ];
    __shared__ PatchScalars sc;
    __shared__ SolveScalars sv[NWV];                           // every wave keeps its own copy of the 3x3 solve: no block barrier for it
    __shared__ float red_nb[NWV * 15], red_dc[NWV * 9];
    __shared__ real red_d[NWV * 22];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid < R) lin[tid] = o.lin[tid];
    const int64_t gp = blockIdx.x;
    const int b = (int)(gp / a.P), p = (int)(gp % a.P);
    const int pi = p / a.wp, pj = p % a.wp;
    const int y0 = a.stride * pi, x0 = a.stride * pj;
    const size_t HW = (size_t)a.H * a.W, HWd = (size_t)(a.H - 2) * (a.W - 2);

    // ---- parameters (global_training.py:141-145) and their chain factors, once per patch.  The fp64 sin / cos / pow / erf / exp
    // are a few hundred instructions each: one lane per value, on two waves (the eight trigonometric values on wave 0, the four
    // eta chains on wave 1), not eight values one after the other on every lane while six waves wait.
    {
        const float* e = a.est + gp * 12;
        if (wv == 0 && lane < 8) {
            const int w2 = (lane >> 1) & 1;                          // wedge 0 / 1
            const real th = wrap_2pi_d(((real)e[4 + 2 * w2] + 1.0) * 3.14159265358979323846);
            const real ph = wrap_2pi_d(((real)e[5 + 2 * w2] + 1.0) * 3.14159265358979323846);
            const real ang = (lane & 1) ? th + ph : th;              // lane: 0 t1, 1 t1+f1, 2 t2, 3 t2+f2; +4: cosine
            const real val = lane < 4 ? sin(ang) : cos(ang);
            real* gsc = lane < 4 ? (lane == 0 ? &sc.g.s11 : lane == 1 ? &sc.g.s12 : lane == 2 ? &sc.g.s21 : &sc.g.s22)
                                 : (lane == 4 ? &sc.g.c11 : lane == 5 ? &sc.g.c12 : lane == 6 ? &sc.g.c21 : &sc.g.c22);
            *gsc = val;
            if (lane == 1) sc.g.sg1 = ph < 3.14159265358979323846 ? 1.0 : -1.0;
            if (lane == 3) sc.g.sg2 = ph < 3.14159265358979323846 ? 1.0 : -1.0;
            if (lane == 0) { sc.g.x0 = 3.0 * (real)e[0]; sc.g.y0 = 3.0 * (real)e[1]; sc.g.x1 = 3.0 * (real)e[2]; sc.g.y1 = 3.0 * (real)e[3]; }
        }
        if (wv == 1 && lane < 4) {
            const real pk = (real)e[8 + lane] + 0.5;
            const real eta = pow(10.0, 2.0 * erf(pk) - 2.0);
            sc.eta[lane] = eta;
            sc.irad[lane] = 1.0 / ((real)be::kRoot2 * eta);
            sc.deta[lane] = eta * 2.302585092994046 * 4.0 * (real)kInvSqrtPi * exp(-pk * pk);
        }
        if (wv == 1 && lane == 4) sc.idelta = 1.0 / (real)o.delta_sq;
    }
    __syncthreads();
    const real irad[4] = {sc.irad[0], sc.irad[1], sc.irad[2], sc.irad[3]};

    const int pix = tid;
    const bool live = pix < NPIX;
    const int pc = live ? pix : 0;
    const int row = pc / R, col = pc - row * R;
    const size_t at = (size_t)(y0 + row) * a.W + x0 + col;

    // ---- pass 1: distances, indicators of both blur sets, normal equations over the 882 rows
    real d1, d2;
    {
        const GeomD g = sc.g;
        d1 = wedge_dist_d(lin[col], lin[row], g.x0, g.y0, g.s11, g.c11, g.s12, g.c12, g.sg1, false, o.w);
        d2 = wedge_dist_d(lin[col], lin[row], g.x1, g.y1, g.s21, g.c21, g.s22, g.c22, g.sg2, true, o.w);
    }
    float h[2][2], y[2][3], u[2][3];
    float nb[15] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};        // gs[6], bs[9]
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const float h1 = 0.5f * (1.0f + erff((float)(d1 * irad[2 * s]))), h2 = 0.5f * (1.0f + erff((float)(d2 * irad[2 * s + 1])));
        h[s][0] = h1; h[s][1] = h2;
        u[s][0] = live ? (1.0f - h1) * (1.0f - h2) : 0.f; u[s][1] = live ? h1 * (1.0f - h2) : 0.f; u[s][2] = live ? h2 : 0.f;
        const float* src = a.img_fit + ((size_t)(b * 2 + s) * HW + at) * 3;
        y[s][0] = live ? src[0] : 0.f; y[s][1] = live ? src[1] : 0.f; y[s][2] = live ? src[2] : 0.f;
        const float u0 = u[s][0], u1 = u[s][1], u2 = u[s][2];
        nb[0] = fmaf(u0, u0, nb[0]); nb[1] = fmaf(u0, u1, nb[1]); nb[2] = fmaf(u0, u2, nb[2]);
        nb[3] = fmaf(u1, u1, nb[3]); nb[4] = fmaf(u1, u2, nb[4]); nb[5] = fmaf(u2, u2, nb[5]);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            nb[6 + c] = fmaf(u0, y[s][c], nb[6 + c]); nb[9 + c] = fmaf(u1, y[s][c], nb[9 + c]); nb[12 + c] = fmaf(u2, y[s][c], nb[12 + c]);
        }
    }
    block_sum(nb, red_nb, lane, wv);
    SolveScalars& my = sv[wv];                                   // this wave's copy (LDS operations of a wave are ordered)
    if (lane == 0) {
        const float* gs = nb;
        const float* bs = nb + 6;
        const double A00 = (double)gs[0] + o.lambda_ridge, A01 = gs[1], A02 = gs[2], A11 = (double)gs[3] + o.lambda_ridge,
                     A12 = gs[4], A22 = (double)gs[5] + o.lambda_ridge;
        const double C00 = A11 * A22 - A12 * A12, C01 = A02 * A12 - A01 * A22, C02 = A01 * A12 - A02 * A11;
        const double C11 = A00 * A22 - A02 * A02, C12 = A01 * A02 - A00 * A12, C22 = A00 * A11 - A01 * A01;
        const double idet = 1.0 / (A00 * C00 + A01 * C01 + A02 * C02);
        const double iv[3][3] = {{C00 * idet, C01 * idet, C02 * idet}, {C01 * idet, C11 * idet, C12 * idet}, {C02 * idet, C12 * idet, C22 * idet}};
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                my.Cc[c][k] = (float)(iv[k][0] * bs[c] + iv[k][1] * bs[3 + c] + iv[k][2] * bs[6 + c]);
                my.inv[c][k] = (float)iv[c][k];
            }
    }
    if (tid == 64) {                                             // depth of the two wedges and its derivative w.r.t. the four etas
        real a_, b_;                                             // (first read in pass 3, two barriers later)
        sc.z1 = depth_and_grad(a.dc, sc.eta[0], sc.eta[2], a_, b_); sc.dz1a = a_; sc.dz1b = b_;
        sc.z2 = depth_and_grad(a.dc, sc.eta[1], sc.eta[3], a_, b_); sc.dz2a = a_; sc.dz2b = b_;
    }
    __builtin_amdgcn_wave_barrier();

    // ---- pass 2 (both blur sets): composite -> colour terms; Sobel terms and their adjoint gathered back
    float gP[2][3];
    float T[7] = {0, 0, 0, 0, 0, 0, 0};                          // T1 T2 T4 T5 | T3 T6 | T7 ; MS separately
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const float* gt = a.img_gt + ((size_t)(b * 2 + s) * HW) * 3;
        const float* Gs = a.G + (size_t)(b * 2 + s) * 3 * HW;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float val = u[s][0] * my.Cc[c][0] + u[s][1] * my.Cc[c][1] + u[s][2] * my.Cc[c][2];
            float r1 = 0.f, r2 = 0.f;
            if (live) {
                sPatch[s][c][pix] = val;
                r1 = val - gt[at * 3 + c];
                r2 = val - Gs[c * HW + at];
            }
            T[0] = fmaf(r1, r1, T[0]); T[1] = fmaf(r2, r2, T[1]);
            gP[s][c] = 2.0f * (r1 * a.wc + r2 * a.wcc);
        }
    }
    __syncthreads();
    if (tid < NQ) {
        const int q = tid;
        const int qy = q / Q, qx = q - qy * Q;
        const size_t atd = (size_t)(y0 + qy) * (a.W - 2) + x0 + qx;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const float* dr = a.deri + ((size_t)(b * 2 + s) * HWd) * 3;
            const float* Gds = a.Gd + (size_t)(b * 2 + s) * 3 * HWd;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float* Pp = sPatch[s][c] + qy * R + qx;
                const float p00 = Pp[0], p01 = Pp[1], p02 = Pp[2], p10 = Pp[R], p12 = Pp[R + 2], p20 = Pp[2 * R],
                            p21 = Pp[2 * R + 1], p22 = Pp[2 * R + 2];
                const float gx = (p02 - p00) + 2.0f * (p12 - p10) + (p22 - p20);
                const float gy = (p00 - p20) + 2.0f * (p01 - p21) + (p02 - p22);
                const float sm = sqrtf(gx * gx + gy * gy + 1e-8f);
                const float t4 = sm - dr[atd * 3 + c], t5 = sm - Gds[c * HWd + atd];
                T[2] = fmaf(t4, t4, T[2]); T[3] = fmaf(t5, t5, T[3]);
                const float k = 2.0f * (t4 * a.ws + t5 * a.wsc) / sm;
                sDx[s][c][q] = k * gx;
                sDy[s][c][q] = k * gy;
            }
        }
    }
    __syncthreads();
    if (live) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
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
                        acc += kx * sDx[s][c][qy * Q + qx] + ky * sDy[s][c][qy * Q + qx];
                    }
                }
                gP[s][c] += acc;
            }
    }

    // ---- adjoint of the shared colour solve
    float dC[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int k = 0; k < 3; ++k) dC[c * 3 + k] = fmaf(gP[s][c], u[s][k], dC[c * 3 + k]);      // u is 0 on dead threads
    block_sum(dC, red_dc, lane, wv);
    if (lane == 0) {
        float V[3][3];
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                V[c][k] = my.inv[k][0] * dC[c * 3] + my.inv[k][1] * dC[c * 3 + 1] + my.inv[k][2] * dC[c * 3 + 2];
                my.V[c][k] = V[c][k];
            }
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                float dkj = 0.f, djk = 0.f;
#pragma unroll
                for (int c = 0; c < 3; ++c) { dkj -= V[c][k] * my.Cc[c][j]; djk -= V[c][j] * my.Cc[c][k]; }
                my.S[k][j] = dkj + djk;
            }
    }
    __builtin_amdgcn_wave_barrier();

    // ---- pass 3: per-pixel adjoints down to the twelve parameters
    real acc[14] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};   // gx0 gy0 gt1 gf1 gx1 gy1 gt2 gf2 | gr[4] | gz1 gz2
    float MS = 0.f;
    if (live) {
        const real px = lin[col], py = lin[row];
        const real idelta = sc.idelta;
        real gd1 = 0, gd2 = 0;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const float h1 = h[s][0], h2 = h[s][1];
            float du[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                float t = 0.f;
#pragma unroll
                for (int c = 0; c < 3; ++c) t += gP[s][c] * my.Cc[c][k] + my.V[c][k] * y[s][c];
#pragma unroll
                for (int j = 0; j < 3; ++j) t += my.S[k][j] * u[s][j];
                du[k] = t;
            }
            const real dh1 = (du[1] - du[0]) * (1.0f - h2);
            const real dh2 = -du[0] * (1.0f - h1) - du[1] * h1 + du[2];
            const real ia = irad[2 * s], ib = irad[2 * s + 1];
            const real za = d1 * ia, zb = d2 * ib;
            const float zaf = (float)za, zbf = (float)zb;
            const real ea = (real)(__expf(-zaf * zaf) * kInvSqrtPi), eb = (real)(__expf(-zbf * zbf) * kInvSqrtPi);
            gd1 += dh1 * ea * ia; gd2 += dh2 * eb * ib;
            acc[8 + 2 * s] -= dh1 * ea * za * ia;
            acc[8 + 2 * s + 1] -= dh2 * eb * zb * ib;
        }
        // boundary terms (:99-103, :121-125)
        const real a1 = fabs(d1), a2 = fabs(d2);
        const real db = d2 >= 0.0 ? d2 : (a1 < a2 ? a1 : a2);
        const real Bv = (real)__expf(-(float)(db * db * idelta));
        const size_t atb = (size_t)b * HW + at;
        const real r3 = Bv - (real)a.Gb[atb];
        const real lb = (real)__log2f(a.bdist[atb] + 1.0f);
        T[4] += (float)(r3 * r3); T[5] += (float)((lb * Bv) * (lb * Bv));
        const real gB = 2.0 * r3 * a.wbc + 2.0 * lb * lb * Bv * a.wbl;
        const real gdb = gB * Bv * (-2.0 * db * idelta);
        if (d2 >= 0.0) gd2 += gdb;
        else if (a1 < a2) gd1 += gdb * (d1 > 0. ? 1. : (d1 < 0. ? -1. : 0.));
        else gd2 += gdb * (d2 > 0. ? 1. : (d2 < 0. ? -1. : 0.));
        // depth term (:127-133): mask as blurry_edges / global_training :83-85
        const bool m1 = d1 * d1 * idelta < 0.69314718055994530942, m2 = d2 * d2 * idelta < 0.69314718055994530942;   // exp(-d^2 / delta^2) > 0.5
        const int mk = (m2 || d2 >= 0.0) ? (m2 ? 2 : 0) : (m1 ? 1 : 0);
        const real bdp = a.bdepth[atb];
        if (bdp != 0.0 && mk != 0) {
            const real diff = (mk == 1 ? sc.z1 : sc.z2) - bdp;
            T[6] += (float)(diff * diff); MS += 1.0f;
            if (mk == 1) acc[12] += 2.0 * diff; else acc[13] += 2.0 * diff;
        }
        const GeomD g = sc.g;
        wedge_backward(px, py, g.x0, g.y0, g.s11, g.c11, g.s12, g.c12, g.sg1, false, o.w, gd1, acc[0], acc[1], acc[2], acc[3]);
        wedge_backward(px, py, g.x1, g.y1, g.s21, g.c21, g.s22, g.c22, g.sg2, true, o.w, gd2, acc[4], acc[5], acc[6], acc[7]);
    }
    real fin[22];                                                // the 14 parameter adjoints and the 8 term sums: one reduction
#pragma unroll
    for (int k = 0; k < 14; ++k) fin[k] = acc[k];
    fin[14] = T[0]; fin[15] = T[1]; fin[16] = T[4]; fin[17] = T[2]; fin[18] = T[3]; fin[19] = T[5]; fin[20] = T[6]; fin[21] = MS;
    block_sum_d(fin, red_d, lane, wv);                           // (order of the partial record: T1 T2 T3 T4 T5 T6 T7 MS)
#pragma unroll
    for (int k = 0; k < 14; ++k) acc[k] = fin[k];
    if (tid == 0) {
        float* pt = a.partial + gp * 8;
#pragma unroll
        for (int k = 0; k < 8; ++k) pt[k] = (float)fin[14 + k];
        float* go = a.grad + gp * 12;
        const real pi_ = 3.14159265358979323846;
        go[0] = (float)(3.0 * acc[0]); go[1] = (float)(3.0 * acc[1]); go[2] = (float)(3.0 * acc[4]); go[3] = (float)(3.0 * acc[5]);
        go[4] = (float)(pi_ * acc[2]); go[5] = (float)(pi_ * acc[3]); go[6] = (float)(pi_ * acc[6]); go[7] = (float)(pi_ * acc[7]);
#pragma unroll
        for (int k = 0; k < 4; ++k) go[8 + k] = (float)(acc[8 + k] * (real)be::kRoot2 * sc.deta[k]);
        float* gd = a.gdepth + gp * 4;                          // depth1 <- (eta0, eta2), depth2 <- (eta1, eta3)
        gd[0] = (float)(acc[12] * sc.dz1a * sc.deta[0]); gd[2] = (float)(acc[12] * sc.dz1b * sc.deta[2]);
        gd[1] = (float)(acc[13] * sc.dz2a * sc.deta[1]); gd[3] = (float)(acc[13] * sc.dz2b * sc.deta[3]);
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
    hipLaunchKernelGGL(k_global_loss, dim3((unsigned)((int64_t)B * P)), dim3(NT), 0, be::as_stream(stream), *o, a);
    return be::check_launch("be_global_loss_f32");
}
