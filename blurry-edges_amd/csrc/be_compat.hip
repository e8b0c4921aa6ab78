// Fine-grained operators of the reference's PostProcess base classes as HIP kernels, for scripts that subclass
// PostProcessLocalBase / PostProcessGlobalBase and call the inherited methods one by one
// (utils/postprocessing_loss.py:43-117,151-173).  The fused passes (be_render*.hip) are the fast path; these
// exist so that such callers still run on the GPU library instead of falling back to eager ATen ops.
#include "be_common.h"
#include "be_wedge.h"

namespace {

constexpr int NPIX = BE_NPIX, R = BE_R;

// params [N,8] -> dists [N,2,441]                       utils/postprocessing_loss.py:43-86
__global__ void k_params2dists(be_render_opts o, const float* __restrict__ params, float* __restrict__ dists, int64_t n) {
    const int64_t patch = blockIdx.x;
    if (patch >= n) return;
    __shared__ be::WedgeGeom g;
    if (threadIdx.x == 0) g = be::make_geom(params + patch * 8, o.wrap_angles != 0);
    __syncthreads();
    for (int pix = threadIdx.x; pix < NPIX; pix += blockDim.x) {
        const int row = pix / R, col = pix - row * R;
        float d1, d2;
        be::wedge_dists(g, o.lin[col], o.lin[row], o.w, d1, d2);
        dists[patch * 2 * NPIX + pix] = d1;
        dists[patch * 2 * NPIX + NPIX + pix] = d2;
    }
}

// dists [N,2,441], etas [N,2] -> wedges [N,3,441]       :91-95
__global__ void k_dists2indicators(const float* __restrict__ dists, const float* __restrict__ etas,
                                   float* __restrict__ wedges, int64_t n) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * NPIX) return;
    const int64_t patch = idx / NPIX;
    const int pix = (int)(idx - patch * NPIX);
    float r1, r2;
    {
#pragma clang fp contract(off)
        r1 = be::kRoot2 * etas[patch * 2];
        r2 = be::kRoot2 * etas[patch * 2 + 1];
    }
    float u0, u1, u2;
    be::indicators(dists[patch * 2 * NPIX + pix], dists[patch * 2 * NPIX + NPIX + pix], r1, r2, u0, u1, u2);
    float* w = wedges + patch * 3 * NPIX + pix;
    w[0] = u0; w[NPIX] = u1; w[2 * NPIX] = u2;
}

// inverse of n 3x3 matrices (cofactors, fp64 inside)    :104-112
__global__ void k_inverse3x3(const float* __restrict__ a, float* __restrict__ out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* m = a + i * 9;
    const double a00 = m[0], a01 = m[1], a02 = m[2], a10 = m[3], a11 = m[4], a12 = m[5], a20 = m[6], a21 = m[7], a22 = m[8];
    const double c00 = a11 * a22 - a12 * a21, c01 = a12 * a20 - a10 * a22, c02 = a10 * a21 - a11 * a20;
    const double idet = 1.0 / (a00 * c00 + a01 * c01 + a02 * c02);
    float* o = out + i * 9;
    o[0] = (float)(c00 * idet); o[1] = (float)((a02 * a21 - a01 * a22) * idet); o[2] = (float)((a01 * a12 - a02 * a11) * idet);
    o[3] = (float)(c01 * idet); o[4] = (float)((a00 * a22 - a02 * a20) * idet); o[5] = (float)((a02 * a10 - a00 * a12) * idet);
    o[6] = (float)(c02 * idet); o[7] = (float)((a01 * a20 - a00 * a21) * idet); o[8] = (float)((a00 * a11 - a01 * a10) * idet);
}

// per-channel Sobel magnitude, valid padding: img [N,C,H,W] -> [N,C,H-2,W-2]      :114-117
__global__ void k_sobel_mag(const float* __restrict__ img, float* __restrict__ out, int64_t planes, int H, int W) {
    const int oh = H - 2, ow = W - 2;
    const int64_t total = planes * oh * ow;
    const int64_t gs = (int64_t)gridDim.x * blockDim.x;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gs) {
#pragma clang fp contract(off)
        const int x = (int)(idx % ow);
        const int y = (int)((idx / ow) % oh);
        const int64_t p = idx / ((int64_t)ow * oh);
        const float* P = img + (p * H + y) * W + x;
        const float p00 = P[0], p01 = P[1], p02 = P[2], p10 = P[W], p12 = P[W + 2], p20 = P[2 * W], p21 = P[2 * W + 1],
                    p22 = P[2 * W + 2];
        const float gx = (p02 - p00) + 2.0f * (p12 - p10) + (p22 - p20);
        const float gy = (p00 - p20) + 2.0f * (p01 - p21) + (p02 - p22);
        out[idx] = sqrtf(gx * gx + gy * gy + 1e-8f);
    }
}

// nn.Fold of a strided patch tensor, owner computes: out[b][c][y][x] = sum over covering patches (i,j) of
// src[b*s_b + c*s_c + (y-stride*i)*s_r + (x-stride*j)*s_col + i*s_pi + j*s_pj]; mode 0 sum, 1 mean (/count),
// 2 count of positive entries (depth mask)                                        :151-173
struct FoldPArgs {
    const float* src;
    const int32_t* src_i;     // int32 source for mode 2 when the mask is an int tensor (else null)
    float* out;
    int64_t s_b, s_c, s_r, s_col, s_pi, s_pj;
    int B, C, hp, wp, H, W, stride, mode;
};
__global__ void k_fold_patches(FoldPArgs a) {
    const int64_t total = (int64_t)a.B * a.C * a.H * a.W;
    const int64_t gs = (int64_t)gridDim.x * blockDim.x;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gs) {
        const int x = (int)(idx % a.W);
        const int y = (int)((idx / a.W) % a.H);
        const int c = (int)((idx / ((int64_t)a.W * a.H)) % a.C);
        const int64_t b = idx / ((int64_t)a.W * a.H * a.C);
        const int s = a.stride;
        int i_lo = y - (R - 1) < 0 ? 0 : (y - (R - 1) + s - 1) / s, i_hi = min(y / s, a.hp - 1);
        int j_lo = x - (R - 1) < 0 ? 0 : (x - (R - 1) + s - 1) / s, j_hi = min(x / s, a.wp - 1);
        float acc = 0.f; int cnt = 0;
        for (int i = i_lo; i <= i_hi; ++i)
            for (int j = j_lo; j <= j_hi; ++j) {
                const int64_t at = b * a.s_b + c * a.s_c + (int64_t)(y - s * i) * a.s_r + (int64_t)(x - s * j) * a.s_col +
                                   i * a.s_pi + j * a.s_pj;
                if (a.mode == 2) acc += (a.src_i ? (a.src_i[at] > 0) : (a.src[at] > 0.0f)) ? 1.0f : 0.0f;
                else acc += a.src[at];
                ++cnt;
            }
        a.out[idx] = a.mode == 1 ? acc / (float)cnt : acc;
    }
}

}  // namespace

extern "C" int be_params2dists_f32(const be_render_opts* o, const float* params8, float* dists, int64_t n, void* stream) {
    BE_REQUIRE(n >= 0, "be_params2dists_f32: n < 0");
    if (n == 0) return BE_OK;
    BE_REQUIRE(o && params8 && dists && n <= 0x7fffffff, "be_params2dists_f32: bad arguments");
    hipLaunchKernelGGL(k_params2dists, dim3((unsigned)n), dim3(128), 0, be::as_stream(stream), *o, params8, dists, n);
    return be::check_launch("be_params2dists_f32");
}

extern "C" int be_dists2indicators_f32(const float* dists, const float* etas, float* wedges, int64_t n, void* stream) {
    BE_REQUIRE(n >= 0, "be_dists2indicators_f32: n < 0");
    if (n == 0) return BE_OK;
    BE_REQUIRE(dists && etas && wedges, "be_dists2indicators_f32: null pointer");
    hipLaunchKernelGGL(k_dists2indicators, dim3((unsigned)((n * NPIX + 255) / 256)), dim3(256), 0, be::as_stream(stream), dists,
                       etas, wedges, n);
    return be::check_launch("be_dists2indicators_f32");
}

extern "C" int be_inverse3x3_f32(const float* a, float* out, int64_t n, void* stream) {
    BE_REQUIRE(n >= 0, "be_inverse3x3_f32: n < 0");
    if (n == 0) return BE_OK;
    BE_REQUIRE(a && out, "be_inverse3x3_f32: null pointer");
    hipLaunchKernelGGL(k_inverse3x3, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, be::as_stream(stream), a, out, n);
    return be::check_launch("be_inverse3x3_f32");
}

extern "C" int be_image_derivative_f32(const float* img, float* out, int64_t planes, int H, int W, void* stream) {
    BE_REQUIRE(img && out && planes > 0 && H > 2 && W > 2, "be_image_derivative_f32: bad arguments");
    const int64_t total = planes * (H - 2) * (W - 2);
    int64_t g = (total + 255) / 256; if (g > 4096) g = 4096;
    hipLaunchKernelGGL(k_sobel_mag, dim3((unsigned)g), dim3(256), 0, be::as_stream(stream), img, out, planes, H, W);
    return be::check_launch("be_image_derivative_f32");
}

extern "C" int be_fold_patches_f32(const float* src, const int32_t* src_int, float* out, int B, int C, int hp, int wp,
                                   int H, int W, int stride, int64_t s_b, int64_t s_c, int64_t s_r, int64_t s_col,
                                   int64_t s_pi, int64_t s_pj, int mode, void* stream) {
    BE_REQUIRE((src || src_int) && out, "be_fold_patches_f32: null pointer");
    BE_REQUIRE(B > 0 && C > 0 && hp > 0 && wp > 0 && stride > 0 && mode >= 0 && mode <= 2, "be_fold_patches_f32: bad sizes");
    BE_REQUIRE(stride * (hp - 1) + R <= H && stride * (wp - 1) + R <= W, "be_fold_patches_f32: patch grid exceeds the image");
    FoldPArgs a{src, src_int, out, s_b, s_c, s_r, s_col, s_pi, s_pj, B, C, hp, wp, H, W, stride, mode};
    const int64_t total = (int64_t)B * C * H * W;
    int64_t g = (total + 255) / 256; if (g > 8192) g = 8192;
    hipLaunchKernelGGL(k_fold_patches, dim3((unsigned)g), dim3(256), 0, be::as_stream(stream), a);
    return be::check_launch("be_fold_patches_f32");
}
