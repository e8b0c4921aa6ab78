// Adjoints of the fine-grained PostProcess / DepthEtas operators of be_compat.hip and be_elementwise.hip, so that a caller's
// subclass of PostProcessLocalBase / PostProcessGlobalBase written against the reference (LocalLoss, local_training.py:10-52;
// GlobalLoss, global_training.py:11-157) runs `loss.backward()` through the inherited methods on the GPU library.
// Every kernel takes the cotangent of the forward's output and writes the cotangent of its inputs; derivatives are evaluated in
// fp64 from the fp32 operands (a wedge edge can be far sharper than the pixel pitch - see be_wedge_d.h - and these kernels are
// HBM-bound with a handful of transcendentals per element, so the wider arithmetic is free) and rounded once.
// Per-patch reductions (the 8 geometry parameters, the 2 blur radii) are summed in a fixed order: bitwise reproducible.
#include "be_common.h"
#include "be_wedge_d.h"

namespace {

constexpr int NPIX = BE_NPIX, R = BE_R;
constexpr int BLOCK = 256, WAVES = BLOCK / 64;

// sum of NV doubles per thread over the workgroup; thread 0 .. NV-1 end up holding total[v] in out[v] (LDS), fixed order
template <int NV>
__device__ __forceinline__ void block_sum(double (&v)[NV], double (*red)[WAVES], double* out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const double s = be_d::wave_sum_d(v[k]);
        if (lane == 0) red[k][wave] = s;
    }
    __syncthreads();
    if (threadIdx.x < NV) {
        double s = 0.0;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) s += red[threadIdx.x][w];
        out[threadIdx.x] = s;
    }
    __syncthreads();
}

// d params [N,8] = J^T d dists [N,2,441]                                       utils/postprocessing_loss.py:43-86
__global__ __launch_bounds__(BLOCK) void k_params2dists_bwd(be_render_opts o, const float* __restrict__ params,
                                                            const float* __restrict__ gdists, float* __restrict__ gparams, int64_t n) {
    const int64_t patch = blockIdx.x;
    if (patch >= n) return;
    __shared__ be_d::GeomD g;
    __shared__ double red[8][WAVES], tot[8];
    if (threadIdx.x == 0) g = be_d::make_geom_d(params + patch * 8);
    __syncthreads();
    double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};          // x0 y0 x1 y1 theta1 phi1 theta2 phi2
    const double w = o.w;
    for (int pix = threadIdx.x; pix < NPIX; pix += BLOCK) {
        const int row = pix / R, col = pix - row * R;
        const double px = o.lin[col], py = o.lin[row];
        const double g1 = gdists[patch * 2 * NPIX + pix], g2 = gdists[patch * 2 * NPIX + NPIX + pix];
        be_d::wedge_backward(px, py, g.x0, g.y0, g.s11, g.c11, g.s12, g.c12, g.sg1, false, w, g1, acc[0], acc[1], acc[4], acc[5]);
        be_d::wedge_backward(px, py, g.x1, g.y1, g.s21, g.c21, g.s22, g.c22, g.sg2, true, w, g2, acc[2], acc[3], acc[6], acc[7]);
    }
    block_sum<8>(acc, red, tot);
    if (threadIdx.x < 8) gparams[patch * 8 + threadIdx.x] = (float)tot[threadIdx.x];
}

// d dists [N,2,441], d etas [N,2] from d wedges [N,3,441]                      :91-95
__global__ __launch_bounds__(BLOCK) void k_dists2indicators_bwd(const float* __restrict__ dists, const float* __restrict__ etas,
                                                                const float* __restrict__ gw, float* __restrict__ gdists,
                                                                float* __restrict__ getas, int64_t n) {
    const int64_t patch = blockIdx.x;
    if (patch >= n) return;
    __shared__ double red[2][WAVES], tot[2];
    const double root2 = (double)be::kRoot2;                                      // the forward's float32 sqrt(2)
    const double r1 = root2 * (double)etas[patch * 2], r2 = root2 * (double)etas[patch * 2 + 1];
    double acc[2] = {0, 0};
    for (int pix = threadIdx.x; pix < NPIX; pix += BLOCK) {
        const double d1 = dists[patch * 2 * NPIX + pix], d2 = dists[patch * 2 * NPIX + NPIX + pix];
        const double t1 = d1 / r1, t2 = d2 / r2;
        const double h1 = 0.5 * (1.0 + erf(t1)), h2 = 0.5 * (1.0 + erf(t2));
        const float* G = gw + patch * 3 * NPIX + pix;
        const double gu0 = G[0], gu1 = G[NPIX], gu2 = G[2 * NPIX];
        // u0 = (1-h1)(1-h2), u1 = h1 (1-h2), u2 = h2
        const double gh1 = (gu1 - gu0) * (1.0 - h2);
        const double gh2 = gu2 - gu0 * (1.0 - h1) - gu1 * h1;
        const double dh1 = (double)be_d::kInvSqrtPi * exp(-t1 * t1) / r1, dh2 = (double)be_d::kInvSqrtPi * exp(-t2 * t2) / r2;   // dh/dd
        gdists[patch * 2 * NPIX + pix] = (float)(gh1 * dh1);
        gdists[patch * 2 * NPIX + NPIX + pix] = (float)(gh2 * dh2);
        acc[0] += gh1 * dh1 * (-t1);                                              // dh/dr = -t dh/dd
        acc[1] += gh2 * dh2 * (-t2);
    }
    block_sum<2>(acc, red, tot);
    if (threadIdx.x < 2) getas[patch * 2 + threadIdx.x] = (float)(tot[threadIdx.x] * root2);     // dr/d eta = sqrt(2)
}

// d A = -M^T (d M) M^T with M = A^-1 (the forward's output)                     :104-112
__global__ void k_inverse3x3_bwd(const float* __restrict__ inv, const float* __restrict__ gout, float* __restrict__ ga, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double M[9], G[9], T[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) { M[k] = inv[i * 9 + k]; G[k] = gout[i * 9 + k]; }
#pragma unroll
    for (int r = 0; r < 3; ++r)                                                   // T = M^T G
#pragma unroll
        for (int c = 0; c < 3; ++c) T[r * 3 + c] = M[0 * 3 + r] * G[0 * 3 + c] + M[1 * 3 + r] * G[1 * 3 + c] + M[2 * 3 + r] * G[2 * 3 + c];
#pragma unroll
    for (int r = 0; r < 3; ++r)                                                   // dA = -T M^T
#pragma unroll
        for (int c = 0; c < 3; ++c)
            ga[i * 9 + r * 3 + c] = (float)(-(T[r * 3 + 0] * M[c * 3 + 0] + T[r * 3 + 1] * M[c * 3 + 1] + T[r * 3 + 2] * M[c * 3 + 2]));
}

// adjoint of the Sobel magnitude, gather form: every input pixel collects from the <= 9 output pixels whose window holds it  :114-117
__global__ void k_sobel_mag_bwd(const float* __restrict__ img, const float* __restrict__ gout, float* __restrict__ gimg,
                                int64_t planes, int H, int W) {
    const int oh = H - 2, ow = W - 2;
    const int64_t total = planes * H * W;
    const int64_t gs = (int64_t)gridDim.x * blockDim.x;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gs) {
        const int X = (int)(idx % W);
        const int Y = (int)((idx / W) % H);
        const int64_t p = idx / ((int64_t)W * H);
        const float* P0 = img + p * H * W;
        const float* G0 = gout + p * oh * ow;
        double acc = 0.0;
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                const int y = Y - a, x = X - b;                                   // output pixel whose tap (a,b) is this input pixel
                if (y < 0 || x < 0 || y >= oh || x >= ow) continue;
                const float sx = (b == 0 ? -1.f : (b == 2 ? 1.f : 0.f)) * (a == 1 ? 2.f : 1.f);     // [[-1,0,1],[-2,0,2],[-1,0,1]]
                const float sy = (a == 0 ? 1.f : (a == 2 ? -1.f : 0.f)) * (b == 1 ? 2.f : 1.f);     // [[1,2,1],[0,0,0],[-1,-2,-1]]
                if (sx == 0.f && sy == 0.f) continue;
                const float* P = P0 + (int64_t)y * W + x;
                const double p00 = P[0], p01 = P[1], p02 = P[2], p10 = P[W], p12 = P[W + 2], p20 = P[2 * W], p21 = P[2 * W + 1],
                             p22 = P[2 * W + 2];
                const double gx = (p02 - p00) + 2.0 * (p12 - p10) + (p22 - p20);
                const double gy = (p00 - p20) + 2.0 * (p01 - p21) + (p02 - p22);
                const double mag = sqrt(gx * gx + gy * gy + 1e-8);
                acc += (double)G0[(int64_t)y * ow + x] * (gx * sx + gy * sy) / mag;
            }
        gimg[idx] = (float)acc;
    }
}

// d p = d eta * eta ln(10) 2 (2/sqrt(pi)) exp(-p^2)                             :88-89
__global__ void k_params2etas_bwd(const float* __restrict__ p, const float* __restrict__ geta, float* __restrict__ gp, int64_t n) {
    const int64_t gs = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gs) {
        const double x = p[i];
        const double eta = pow(10.0, 2.0 * erf(x) - 2.0);
        gp[i] = (float)((double)geta[i] * eta * 2.302585092994045684 * 4.0 * (double)be_d::kInvSqrtPi * exp(-x * x));
    }
}

// exp(-x^2 / delta^2) and its adjoint                                          :97-98
__global__ void k_norm_gauss(const float* __restrict__ x, float* __restrict__ y, float delta_sq, int64_t n) {
#pragma clang fp contract(off)
    const int64_t gs = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gs) { const float v = x[i]; y[i] = expf(-(v * v) / delta_sq); }
}
__global__ void k_norm_gauss_bwd(const float* __restrict__ x, const float* __restrict__ gy, float* __restrict__ gx, float delta_sq,
                                 int64_t n) {
    const int64_t gs = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gs) {
        const double v = x[i], d2 = delta_sq;
        gx[i] = (float)((double)gy[i] * exp(-(v * v) / d2) * (-2.0 * v / d2));
    }
}

// adjoint of DepthEtas.etas2depth: the branch is the forward's (same fp32 half-plane tests)     utils/depth_etas.py:23-34
__global__ void k_etas2depth_bwd(be_depth_consts c, const float* __restrict__ e1, const float* __restrict__ e2,
                                 const float* __restrict__ gz, float* __restrict__ g1, float* __restrict__ g2, int64_t n) {
    const int64_t gs = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gs) {
        int br;
        (void)be::etas2depth(c, e1[i], e2[i], br);
        const double a = e1[i], b = e2[i], I = c.intercept;
        double e11, e22, a11, b11, a22, b22;                                      // d e11 / d (a, b), d e22 / d (a, b)
        const double sum_h = (a + b - I) * 0.5;
        if (br == 0)      { e11 = sum_h;                   e22 = I + sum_h;             a11 = .5; b11 = .5;  a22 = .5;  b22 = .5; }
        else if (br == 1) { e11 = I + (a - b - I) * 0.5;   e22 = (b - a + I) * 0.5;     a11 = .5; b11 = -.5; a22 = -.5; b22 = .5; }
        else if (br == 2) { e11 = I + sum_h;               e22 = sum_h;                 a11 = .5; b11 = .5;  a22 = .5;  b22 = .5; }
        else              { e11 = a;                       e22 = b;                     a11 = 1.; b11 = 0.;  a22 = 0.;  b22 = 1.; }
        const double den = (double)c.k2 * (e11 * e11 - e22 * e22) + (double)c.den_const;
        const double dz_dden = -(double)c.numerator / (den * den);
        const double z11 = dz_dden * (double)c.k2 * 2.0 * e11, z22 = -dz_dden * (double)c.k2 * 2.0 * e22;
        const double g = gz[i];
        g1[i] = (float)(g * (z11 * a11 + z22 * a22));
        g2[i] = (float)(g * (z11 * b11 + z22 * b22));
    }
}

// adjoint of DepthEtas.depth2sigma                                              utils/depth_etas.py:36-37
__global__ void k_depth2sigma_bwd(be_depth_consts c, const float* __restrict__ depth, float rho_prime, const float* __restrict__ geta,
                                  float* __restrict__ gdepth, int64_t n) {
    const int64_t gs = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gs) {
        const double z = depth[i];
        const double u = (1.0 / z - (double)rho_prime) * (double)c.s + 1.0;
        const double sg = u > 0.0 ? 1.0 : (u < 0.0 ? -1.0 : 0.0);
        gdepth[i] = (float)((double)geta[i] * sg * (double)c.s * (-1.0 / (z * z)) / (double)c.k);
    }
}

// adjoint of nn.Fold (k_fold_patches of be_compat.hip): every patch entry reads the cotangent of the pixel it was added to
// (mode 0), divided by that pixel's overlap count (mode 1)                                      :151-173
struct FoldBArgs {
    const float* gout;
    float* gsrc;
    int64_t s_b, s_c, s_r, s_col, s_pi, s_pj;
    int B, C, hp, wp, H, W, stride, mode;
};
__global__ void k_fold_patches_bwd(FoldBArgs a) {
    const int64_t total = (int64_t)a.B * a.C * NPIX * a.hp * a.wp;
    const int64_t gs = (int64_t)gridDim.x * blockDim.x;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gs) {
        int64_t t = idx;
        const int j = (int)(t % a.wp); t /= a.wp;
        const int i = (int)(t % a.hp); t /= a.hp;
        const int pix = (int)(t % NPIX); t /= NPIX;
        const int c = (int)(t % a.C);
        const int64_t b = t / a.C;
        const int r = pix / R, col = pix - r * R;
        const int y = a.stride * i + r, x = a.stride * j + col;
        float g = a.gout[((b * a.C + c) * a.H + y) * a.W + x];
        if (a.mode == 1) {
            const int s = a.stride;
            const int i_lo = y - (R - 1) < 0 ? 0 : (y - (R - 1) + s - 1) / s, i_hi = min(y / s, a.hp - 1);
            const int j_lo = x - (R - 1) < 0 ? 0 : (x - (R - 1) + s - 1) / s, j_hi = min(x / s, a.wp - 1);
            g = g / (float)((i_hi - i_lo + 1) * (j_hi - j_lo + 1));
        }
        a.gsrc[b * a.s_b + c * a.s_c + (int64_t)r * a.s_r + (int64_t)col * a.s_col + i * a.s_pi + j * a.s_pj] = g;
    }
}

// est[:, c0:c1] <- remainder(est[:, c0:c1], 2 pi) in place                                     local_training.py:33
__global__ void k_wrap_angles_inplace(float* __restrict__ est, int64_t n, int ld, int c0, int c1) {
    const int nc = c1 - c0;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * nc) return;
    float* p = est + (i / nc) * ld + c0 + (int)(i % nc);
    *p = be::remainder_2pi(*p);
}

inline unsigned grid_for(int64_t n, int cap = 4096) {
    int64_t g = (n + 255) / 256;
    return (unsigned)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace

extern "C" int be_params2dists_bwd_f32(const be_render_opts* o, const float* params8, const float* gdists, float* gparams8, int64_t n,
                                       void* stream) {
    BE_REQUIRE(n >= 0, "be_params2dists_bwd_f32: n < 0");
    if (n == 0) return BE_OK;
    BE_REQUIRE(o && params8 && gdists && gparams8 && n <= 0x7fffffff, "be_params2dists_bwd_f32: bad arguments");
    hipLaunchKernelGGL(k_params2dists_bwd, dim3((unsigned)n), dim3(BLOCK), 0, be::as_stream(stream), *o, params8, gdists, gparams8, n);
    return be::check_launch("be_params2dists_bwd_f32");
}

extern "C" int be_dists2indicators_bwd_f32(const float* dists, const float* etas, const float* gwedges, float* gdists, float* getas,
                                           int64_t n, void* stream) {
    BE_REQUIRE(n >= 0, "be_dists2indicators_bwd_f32: n < 0");
    if (n == 0) return BE_OK;
    BE_REQUIRE(dists && etas && gwedges && gdists && getas && n <= 0x7fffffff, "be_dists2indicators_bwd_f32: bad arguments");
    hipLaunchKernelGGL(k_dists2indicators_bwd, dim3((unsigned)n), dim3(BLOCK), 0, be::as_stream(stream), dists, etas, gwedges, gdists,
                       getas, n);
    return be::check_launch("be_dists2indicators_bwd_f32");
}

extern "C" int be_inverse3x3_bwd_f32(const float* inv, const float* gout, float* ga, int64_t n, void* stream) {
    BE_REQUIRE(n >= 0, "be_inverse3x3_bwd_f32: n < 0");
    if (n == 0) return BE_OK;
    BE_REQUIRE(inv && gout && ga, "be_inverse3x3_bwd_f32: null pointer");
    hipLaunchKernelGGL(k_inverse3x3_bwd, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, be::as_stream(stream), inv, gout, ga, n);
    return be::check_launch("be_inverse3x3_bwd_f32");
}

extern "C" int be_image_derivative_bwd_f32(const float* img, const float* gout, float* gimg, int64_t planes, int H, int W, void* stream) {
    BE_REQUIRE(img && gout && gimg && planes > 0 && H > 2 && W > 2, "be_image_derivative_bwd_f32: bad arguments");
    hipLaunchKernelGGL(k_sobel_mag_bwd, dim3(grid_for(planes * H * W, 8192)), dim3(256), 0, be::as_stream(stream), img, gout, gimg, planes,
                       H, W);
    return be::check_launch("be_image_derivative_bwd_f32");
}

extern "C" int be_params2etas_bwd_f32(const float* p, const float* geta, float* gp, int64_t n, void* stream) {
    BE_REQUIRE(n >= 0, "be_params2etas_bwd_f32: n < 0");
    if (n == 0) return BE_OK;
    BE_REQUIRE(p && geta && gp, "be_params2etas_bwd_f32: null pointer");
    hipLaunchKernelGGL(k_params2etas_bwd, dim3(grid_for(n)), dim3(256), 0, be::as_stream(stream), p, geta, gp, n);
    return be::check_launch("be_params2etas_bwd_f32");
}

extern "C" int be_normalized_gaussian_f32(const float* x, float* y, float delta_sq, int64_t n, void* stream) {
    BE_REQUIRE(n >= 0 && delta_sq > 0.0f, "be_normalized_gaussian_f32: n < 0 or delta_sq <= 0");
    if (n == 0) return BE_OK;
    BE_REQUIRE(x && y, "be_normalized_gaussian_f32: null pointer");
    hipLaunchKernelGGL(k_norm_gauss, dim3(grid_for(n)), dim3(256), 0, be::as_stream(stream), x, y, delta_sq, n);
    return be::check_launch("be_normalized_gaussian_f32");
}

extern "C" int be_normalized_gaussian_bwd_f32(const float* x, const float* gy, float* gx, float delta_sq, int64_t n, void* stream) {
    BE_REQUIRE(n >= 0 && delta_sq > 0.0f, "be_normalized_gaussian_bwd_f32: n < 0 or delta_sq <= 0");
    if (n == 0) return BE_OK;
    BE_REQUIRE(x && gy && gx, "be_normalized_gaussian_bwd_f32: null pointer");
    hipLaunchKernelGGL(k_norm_gauss_bwd, dim3(grid_for(n)), dim3(256), 0, be::as_stream(stream), x, gy, gx, delta_sq, n);
    return be::check_launch("be_normalized_gaussian_bwd_f32");
}

extern "C" int be_etas2depth_bwd_f32(const be_depth_consts* c, const float* eta1, const float* eta2, const float* gdepth, float* geta1,
                                     float* geta2, int64_t n, void* stream) {
    BE_REQUIRE(n >= 0, "be_etas2depth_bwd_f32: n < 0");
    if (n == 0) return BE_OK;
    BE_REQUIRE(c && eta1 && eta2 && gdepth && geta1 && geta2, "be_etas2depth_bwd_f32: null pointer");
    hipLaunchKernelGGL(k_etas2depth_bwd, dim3(grid_for(n, 2048)), dim3(256), 0, be::as_stream(stream), *c, eta1, eta2, gdepth, geta1, geta2, n);
    return be::check_launch("be_etas2depth_bwd_f32");
}

extern "C" int be_depth2sigma_bwd_f32(const be_depth_consts* c, const float* depth, float rho_prime, const float* geta, float* gdepth,
                                      int64_t n, void* stream) {
    BE_REQUIRE(n >= 0, "be_depth2sigma_bwd_f32: n < 0");
    if (n == 0) return BE_OK;
    BE_REQUIRE(c && depth && geta && gdepth, "be_depth2sigma_bwd_f32: null pointer");
    hipLaunchKernelGGL(k_depth2sigma_bwd, dim3(grid_for(n, 2048)), dim3(256), 0, be::as_stream(stream), *c, depth, rho_prime, geta, gdepth, n);
    return be::check_launch("be_depth2sigma_bwd_f32");
}

extern "C" int be_fold_patches_bwd_f32(const float* gout, float* gsrc, int B, int C, int hp, int wp, int H, int W, int stride,
                                       int64_t s_b, int64_t s_c, int64_t s_r, int64_t s_col, int64_t s_pi, int64_t s_pj, int mode,
                                       void* stream) {
    BE_REQUIRE(gout && gsrc, "be_fold_patches_bwd_f32: null pointer");
    BE_REQUIRE(B > 0 && C > 0 && hp > 0 && wp > 0 && stride > 0 && (mode == 0 || mode == 1), "be_fold_patches_bwd_f32: bad sizes / mode");
    BE_REQUIRE(stride * (hp - 1) + R <= H && stride * (wp - 1) + R <= W, "be_fold_patches_bwd_f32: patch grid exceeds the image");
    FoldBArgs a{gout, gsrc, s_b, s_c, s_r, s_col, s_pi, s_pj, B, C, hp, wp, H, W, stride, mode};
    hipLaunchKernelGGL(k_fold_patches_bwd, dim3(grid_for((int64_t)B * C * NPIX * hp * wp, 16384)), dim3(256), 0, be::as_stream(stream), a);
    return be::check_launch("be_fold_patches_bwd_f32");
}

extern "C" int be_wrap_angles_inplace_f32(float* est, int64_t n, int ld, int col0, int col1, void* stream) {
    BE_REQUIRE(n >= 0, "be_wrap_angles_inplace_f32: n < 0");
    if (n == 0) return BE_OK;
    BE_REQUIRE(est && ld > 0 && col0 >= 0 && col1 > col0 && col1 <= ld, "be_wrap_angles_inplace_f32: bad arguments");
    hipLaunchKernelGGL(k_wrap_angles_inplace, dim3((unsigned)((n * (col1 - col0) + 255) / 256)), dim3(256), 0, be::as_stream(stream), est, n,
                       ld, col0, col1);
    return be::check_launch("be_wrap_angles_inplace_f32");
}
