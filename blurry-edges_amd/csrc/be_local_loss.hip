// LocalLoss forward + analytic backward in one launch, one wavefront (large batches) or four (the training batch) per patch (gfx950).
//
// Replaces LocalLoss.get_patches + LocalLoss.forward (local_training.py:32-52) and everything autograd records
// under them (params2dists, params2etas, dists2indicators, the ridge solve through inverse_3by3, the composite,
// the boundary map, get_image_derivative: utils/postprocessing_loss.py:43-117) plus their backward.
//
//   loss = mean_px sum_c (gt - patch)^2  +  beta_b * mean_px (bdist * B)^2  +  beta_s * mean_q sum_c (deri - Sobel(patch))^2
//
// The gradient d loss / d est [B,10] is derived by hand (chain: Sobel^T -> composite -> normal equations
// (dC -> v = G^-1 dC -> db = v, dG = -v C^T) -> indicators -> erf -> wedge distances -> vertices / angles / eta).
// Hard selections (min, where, sign) take the derivative of the selected branch, as autograd does.
// Per patch the kernel reads 441*3*2 + 441 + 361*3 floats and writes 10 + 3 floats; LDS holds the rendered patch
// and the two Sobel adjoint images of the patch (14 KB per patch).
//
// WPP = wavefronts per patch.  A training step has 64 patches: with one wavefront each the launch is 64 waves walking 441 pixels in
// 7 passes of fp64 erf / exp chains, every instruction's latency exposed (59 us, round 3 trace).  WPP = 4 gives a patch a whole
// workgroup (2 passes); the per-patch sums cross the four waves through LDS in a fixed order, so results stay run-to-run identical.
#include "be_common.h"
#include "be_wedge_d.h"

namespace {

constexpr int NPIX = BE_NPIX, R = BE_R, WAVES = 4;
constexpr int Q = 19, NQ = Q * Q;                 // valid-Sobel output 19x19

struct LossArgs {
    const float* est;        // [B,10] raw CNN output
    const float* img_fit;    // [B,21,21,3] channels-last: pixels the colours are regressed on
    const float* gt;         // [B,21,21,3]
    const float* bdist;      // [B,21,21]
    const float* deri;       // [B,19,19,3]
    float* partial;          // [B,3]: sum_px sum_c (gt-patch)^2, sum_px (bdist*B)^2, sum_q sum_c (deri-S)^2
    float* grad;             // [B,10] d loss / d est (already scaled by the three means and betas), or null
    float* patches;          // [B,3,21,21] or null
    float* boundary;         // [B,21,21] or null
    float w1, w2, w3;        // 1/(B*441), beta_b/(B*441), beta_s/(B*361)
    int64_t n;
};

using namespace be_d;

// sums of N per-lane values over the lanes of a patch: DPP wave sums, then (WPP > 1) the waves' totals through LDS, added in wave order
template <int WPP, int N, class T>
__device__ __forceinline__ void patch_sum(T (&v)[N], double* red, int wave, int lane64) {
#pragma unroll
    for (int k = 0; k < N; ++k) {
        if constexpr (sizeof(T) == 8) v[k] = wave_sum_d(v[k]); else v[k] = be::wave_sum(v[k]);
    }
    if constexpr (WPP > 1) {
        T* r = reinterpret_cast<T*>(red);
        __syncthreads();                                       // the previous sum's readers are done with `red`
        if (lane64 == 0) {
#pragma unroll
            for (int k = 0; k < N; ++k) r[wave * N + k] = v[k];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < N; ++k) {
            T t = r[k];
#pragma unroll
            for (int w = 1; w < WPP; ++w) t += r[w * N + k];
            v[k] = t;
        }
    }
}

template <int WPP>
__global__ __launch_bounds__(64 * WAVES)
void k_local_loss(be_render_opts o, LossArgs a) {
    constexpr int SLOTS = WAVES / WPP, T = 64 * WPP, PASSES = (NPIX + T - 1) / T;
    __shared__ float lin[R];
    __shared__ float sPatch[SLOTS][3][NPIX];
    __shared__ float sDx[SLOTS][3][NQ];
    __shared__ float sDy[SLOTS][3][NQ];
    __shared__ double red[WAVES * 16];
    if (threadIdx.x < R) lin[threadIdx.x] = o.lin[threadIdx.x];
    __syncthreads();
    const int lane = threadIdx.x % T, wv = threadIdx.x / T;      // lane within the patch, patch slot of the workgroup
    const int lane64 = threadIdx.x & 63, wave = (threadIdx.x >> 6) % WPP;
    const int64_t patch_raw = (int64_t)blockIdx.x * SLOTS + wv;
    const bool active = patch_raw < a.n;
    const int64_t patch = active ? patch_raw : a.n - 1;        // idle waves shadow the last patch (no stores)

    const float* p = a.est + patch * 10;
    const GeomD g = make_geom_d(p);                             // local_training.py:33 wraps the angles first
    const real eta1 = pow(10.0, 2.0 * erf((real)p[8]) - 2.0), eta2 = pow(10.0, 2.0 * erf((real)p[9]) - 2.0);
    const real r1 = (real)be::kRoot2 * eta1, r2 = (real)be::kRoot2 * eta2;
    const float* fit = a.img_fit + patch * NPIX * 3;
    const float* gt = a.gt + patch * NPIX * 3;

    real d1s[PASSES], d2s[PASSES];
    float h1s[PASSES], h2s[PASSES];
    float gs[6] = {0, 0, 0, 0, 0, 0}, bs[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int it = 0; it < PASSES; ++it) {
        const int pix = it * T + lane;
        const bool live = pix < NPIX;
        const int pc = live ? pix : 0;
        const int row = pc / R, col = pc - row * R;
        const real d1 = wedge_dist_d(lin[col], lin[row], g.x0, g.y0, g.s11, g.c11, g.s12, g.c12, g.sg1, false, o.w);
        const real d2 = wedge_dist_d(lin[col], lin[row], g.x1, g.y1, g.s21, g.c21, g.s22, g.c22, g.sg2, true, o.w);
        const float h1 = (float)(0.5 * (1.0 + erf(d1 / r1))), h2 = (float)(0.5 * (1.0 + erf(d2 / r2)));
        d1s[it] = d1; d2s[it] = d2; h1s[it] = h1; h2s[it] = h2;
        const float u0 = live ? (1.0f - h1) * (1.0f - h2) : 0.f, u1 = live ? h1 * (1.0f - h2) : 0.f, u2 = live ? h2 : 0.f;
        const float yr = live ? fit[pc * 3] : 0.f, yg = live ? fit[pc * 3 + 1] : 0.f, yb = live ? fit[pc * 3 + 2] : 0.f;
        gs[0] = fmaf(u0, u0, gs[0]); gs[1] = fmaf(u0, u1, gs[1]); gs[2] = fmaf(u0, u2, gs[2]);
        gs[3] = fmaf(u1, u1, gs[3]); gs[4] = fmaf(u1, u2, gs[4]); gs[5] = fmaf(u2, u2, gs[5]);
        bs[0] = fmaf(u0, yr, bs[0]); bs[1] = fmaf(u0, yg, bs[1]); bs[2] = fmaf(u0, yb, bs[2]);
        bs[3] = fmaf(u1, yr, bs[3]); bs[4] = fmaf(u1, yg, bs[4]); bs[5] = fmaf(u1, yb, bs[5]);
        bs[6] = fmaf(u2, yr, bs[6]); bs[7] = fmaf(u2, yg, bs[7]); bs[8] = fmaf(u2, yb, bs[8]);
    }
    patch_sum<WPP>(gs, red, wave, lane64);
    patch_sum<WPP>(bs, red, wave, lane64);
    // G (with ridge), its inverse by cofactors (fp64), colours C[c][k]
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
    float Cc[3][3];                                           // [channel][wedge]
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int k = 0; k < 3; ++k)
            Cc[c][k] = (float)(inv[k][0] * bs[c] + inv[k][1] * bs[3 + c] + inv[k][2] * bs[6 + c]);

    // ---- composite -> LDS, colour loss and its adjoint
    float gP[PASSES][3];
    float L1 = 0.f;
#pragma unroll
    for (int it = 0; it < PASSES; ++it) {
        const int pix = it * T + lane;
        const bool live = pix < NPIX;
        const float h1 = h1s[it], h2 = h2s[it];
        const float u0 = (1.0f - h1) * (1.0f - h2), u1 = h1 * (1.0f - h2), u2 = h2;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float v = u0 * Cc[c][0] + u1 * Cc[c][1] + u2 * Cc[c][2];
            float r = 0.f;
            if (live) {
                sPatch[wv][c][pix] = v;
                r = v - gt[pix * 3 + c];
                if (a.patches && active) a.patches[(patch * 3 + c) * NPIX + pix] = v;
            }
            L1 = fmaf(r, r, L1);
            gP[it][c] = 2.0f * r * a.w1;
        }
    }
    __syncthreads();
    // ---- Sobel magnitude on the 19x19 interior, smoothness loss and its adjoint images
    float L3 = 0.f;
    const float* dr = a.deri + patch * NQ * 3;
    for (int q0 = 0; q0 < NQ; q0 += T) {
        const int q = q0 + lane;
        if (q < NQ) {
            const int qy = q / Q, qx = q - qy * Q;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float* P = sPatch[wv][c] + qy * R + qx;
                const float p00 = P[0], p01 = P[1], p02 = P[2], p10 = P[R], p12 = P[R + 2], p20 = P[2 * R], p21 = P[2 * R + 1],
                            p22 = P[2 * R + 2];
                const float gx = (p02 - p00) + 2.0f * (p12 - p10) + (p22 - p20);      // [[-1,0,1],[-2,0,2],[-1,0,1]]
                const float gy = (p00 - p20) + 2.0f * (p01 - p21) + (p02 - p22);      // [[1,2,1],[0,0,0],[-1,-2,-1]]
                const float s = sqrtf(gx * gx + gy * gy + 1e-8f);
                const float t = dr[q * 3 + c] - s;
                L3 = fmaf(t, t, L3);
                const float k = -2.0f * t * a.w3 / s;
                sDx[wv][c][q] = k * gx;
                sDy[wv][c][q] = k * gy;
            }
        }
    }
    __syncthreads();
    // ---- gather the Sobel adjoint back onto the 21x21 grid
#pragma unroll
    for (int it = 0; it < PASSES; ++it) {
        const int pix = it * T + lane;
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
                gP[it][c] += acc;
            }
        }
    }
    // ---- dC[c][k] = sum_px gP_c u_k ; v_c = G^-1 dC[c] ; S = dG + dG^T with dG = -sum_c v_c C[c]^T
    float dC[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int it = 0; it < PASSES; ++it) {
        const bool live = it * T + lane < NPIX;
        const float h1 = h1s[it], h2 = h2s[it];
        const float u[3] = {live ? (1.0f - h1) * (1.0f - h2) : 0.f, live ? h1 * (1.0f - h2) : 0.f, live ? h2 : 0.f};
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int k = 0; k < 3; ++k) dC[c * 3 + k] = fmaf(gP[it][c], u[k], dC[c * 3 + k]);
    }
    patch_sum<WPP>(dC, red, wave, lane64);
    float V[3][3], S[3][3];                                   // V[c][k]
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
    // ---- per-pixel adjoints down to the ten parameters
    real gx0 = 0, gy0 = 0, gt1 = 0, gf1 = 0, gx1 = 0, gy1 = 0, gt2 = 0, gf2 = 0, gr1 = 0, gr2 = 0;
    float L2 = 0;
#pragma unroll
    for (int it = 0; it < PASSES; ++it) {
        const int pix = it * T + lane;
        if (pix < NPIX) {
            const int row = pix / R, col = pix - row * R;
            const real px = lin[col], py = lin[row];
            const real d1 = d1s[it], d2 = d2s[it];
            const float h1 = h1s[it], h2 = h2s[it];
            const float u[3] = {(1.0f - h1) * (1.0f - h2), h1 * (1.0f - h2), h2};
            const float y[3] = {fit[pix * 3], fit[pix * 3 + 1], fit[pix * 3 + 2]};
            float du[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                float acc = 0.f;
#pragma unroll
                for (int c = 0; c < 3; ++c) acc += gP[it][c] * Cc[c][k] + V[c][k] * y[c];
#pragma unroll
                for (int j = 0; j < 3; ++j) acc += S[k][j] * u[j];
                du[k] = acc;
            }
            const float dh1 = (du[1] - du[0]) * (1.0f - h2);
            const float dh2 = -du[0] * (1.0f - h1) - du[1] * h1 + du[2];
            const real z1 = d1 / r1, z2 = d2 / r2;
            const real e1 = exp(-z1 * z1) * (real)kInvSqrtPi, e2 = exp(-z2 * z2) * (real)kInvSqrtPi;
            real gd1 = dh1 * e1 / r1, gd2 = dh2 * e2 / r2;
            gr1 -= dh1 * e1 * d1 / (r1 * r1);
            gr2 -= dh2 * e2 * d2 / (r2 * r2);
            // boundary localisation term (local_training.py:42-44,50)
            const real a1 = fabs(d1), a2 = fabs(d2);
            const real db = d2 >= 0.0 ? d2 : (a1 < a2 ? a1 : a2);
            const real Bv = exp(-(db * db) / (real)o.delta_sq);
            const real bd = a.bdist[patch * NPIX + pix];
            L2 += (float)((bd * Bv) * (bd * Bv));
            const real gdb = 2.0 * bd * bd * Bv * a.w2 * Bv * (-2.0 * db / (real)o.delta_sq);
            if (d2 >= 0.0) gd2 += gdb;
            else if (a1 < a2) gd1 += gdb * (d1 > 0. ? 1. : (d1 < 0. ? -1. : 0.));
            else gd2 += gdb * (d2 > 0. ? 1. : (d2 < 0. ? -1. : 0.));
            if (a.boundary && active) a.boundary[patch * NPIX + pix] = (float)Bv;
            wedge_backward(px, py, g.x0, g.y0, g.s11, g.c11, g.s12, g.c12, g.sg1, false, o.w, gd1, gx0, gy0, gt1, gf1);
            wedge_backward(px, py, g.x1, g.y1, g.s21, g.c21, g.s22, g.c22, g.sg2, true, o.w, gd2, gx1, gy1, gt2, gf2);
        }
    }
    real gsum[10] = {gx0, gy0, gx1, gy1, gt1, gf1, gt2, gf2, gr1, gr2};
    float lsum[3] = {L1, L2, L3};
    patch_sum<WPP>(gsum, red, wave, lane64);
    patch_sum<WPP>(lsum, red, wave, lane64);
    gx0 = gsum[0]; gy0 = gsum[1]; gx1 = gsum[2]; gy1 = gsum[3]; gt1 = gsum[4]; gf1 = gsum[5]; gt2 = gsum[6]; gf2 = gsum[7];
    gr1 = gsum[8]; gr2 = gsum[9];
    if (lane == 0 && active) {
        a.partial[patch * 3] = lsum[0]; a.partial[patch * 3 + 1] = lsum[1]; a.partial[patch * 3 + 2] = lsum[2];
        if (a.grad) {
            float* go = a.grad + patch * 10;
            go[0] = (float)gx0; go[1] = (float)gy0; go[2] = (float)gx1; go[3] = (float)gy1;
            go[4] = (float)gt1; go[5] = (float)gf1; go[6] = (float)gt2; go[7] = (float)gf2;
            // r = sqrt2 * eta, eta = 10^(2 erf(p) - 2): d eta / d p = eta * ln10 * 4/sqrt(pi) * exp(-p^2)
            const real c10 = 2.302585092994046 * 4.0 * (real)kInvSqrtPi;
            go[8] = (float)(gr1 * (real)be::kRoot2 * eta1 * c10 * exp(-(real)p[8] * p[8]));
            go[9] = (float)(gr2 * (real)be::kRoot2 * eta2 * c10 * exp(-(real)p[9] * p[9]));
        }
    }
}

}  // namespace

extern "C" int be_local_loss_f32(const be_render_opts* o, const float* est, const float* img_fit, const float* gt,
                                 const float* bdist, const float* deri, float beta_bndry, float beta_smooth,
                                 float* partial, float* grad_est, float* patches, float* boundary, int64_t n,
                                 void* stream) {
    BE_REQUIRE(n >= 0, "be_local_loss_f32: n < 0");
    if (n == 0) return BE_OK;
    BE_REQUIRE(o && est && img_fit && gt && bdist && deri && partial, "be_local_loss_f32: null pointer");
    LossArgs a{est, img_fit, gt, bdist, deri, partial, grad_est, patches, boundary,
               1.0f / ((float)n * NPIX), beta_bndry / ((float)n * NPIX), beta_smooth / ((float)n * NQ), n};
    if (n <= 1024) {                     // a training batch: a workgroup per patch (the launch would not fill the chip otherwise)
        hipLaunchKernelGGL(k_local_loss<WAVES>, dim3((unsigned)n), dim3(64 * WAVES), 0, be::as_stream(stream), *o, a);
        return be::check_launch("be_local_loss_f32");
    }
    const int64_t blocks = (n + WAVES - 1) / WAVES;
    BE_REQUIRE(blocks <= 0x7fffffff, "be_local_loss_f32: n too large");
    hipLaunchKernelGGL(k_local_loss<1>, dim3((unsigned)blocks), dim3(64 * WAVES), 0, be::as_stream(stream), *o, a);
    return be::check_launch("be_local_loss_f32");
}

namespace {
__global__ void k_local_loss_finish(const float* __restrict__ partial, int B, float beta_b, float beta_s, float* __restrict__ out) {
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;           // one wavefront: lane l takes patches l, l + 64, ...; then the DPP tree (fixed order)
    for (int i = threadIdx.x; i < B; i += 64) { s0 += partial[3 * i]; s1 += partial[3 * i + 1]; s2 += partial[3 * i + 2]; }
    s0 = be_d::wave_sum_d(s0); s1 = be_d::wave_sum_d(s1); s2 = be_d::wave_sum_d(s2);
    if (threadIdx.x != 0) return;
    const double n1 = (double)B * NPIX, n2 = (double)B * NQ;
    out[0] = (float)(s0 / n1 + (double)beta_b * s1 / n1 + (double)beta_s * s2 / n2);
}
}  // namespace

extern "C" int be_local_loss_finish_f32(const float* partial, int B, float beta_bndry, float beta_smooth, float* loss_out, void* stream) {
    BE_REQUIRE(partial && loss_out && B > 0, "be_local_loss_finish_f32: bad arguments");
    hipLaunchKernelGGL(k_local_loss_finish, dim3(1), dim3(64), 0, be::as_stream(stream), partial, B, beta_bndry, beta_smooth, loss_out);
    return be::check_launch("be_local_loss_finish_f32");
}
