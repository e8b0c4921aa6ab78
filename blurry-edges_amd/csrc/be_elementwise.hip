// DepthEtas solve and eta map as HIP kernels (gfx950).
//   params2etas ........ utils/postprocessing_loss.py:88-89
//   etas2depth ......... utils/depth_etas.py:23-34
//   depth2sigma ........ utils/depth_etas.py:36-37
// These are tiny HBM/launch-bound elementwise kernels (24 B per pair); they exist standalone because the
// reference exposes them as the DepthEtas API.  The arithmetic keeps the reference's operation order with
// floating-point contraction OFF, so the three half-plane tests pick the same branch as PyTorch-CPU does.
#include "be_common.h"
#include "be_device_math.h"

#pragma clang fp contract(off)

namespace {

__global__ void k_params2etas(const float* __restrict__ p, float* __restrict__ eta, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) eta[i] = be::param2eta(p[i]);
}

__global__ void k_etas2depth(be_depth_consts c, const float* __restrict__ e1, const float* __restrict__ e2,
                             float* __restrict__ z, int32_t* __restrict__ branch, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        int br;
        z[i] = be::etas2depth(c, e1[i], e2[i], br);
        if (branch) branch[i] = br;
    }
}

__global__ void k_depth2sigma(be_depth_consts c, const float* __restrict__ depth, float rho_prime,
                              float* __restrict__ eta, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) eta[i] = be::depth2sigma(c, depth[i], rho_prime);
}

__global__ void k_local_depth(be_depth_consts c, const float* __restrict__ params10, float* __restrict__ depth,
                              int64_t n_pairs) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // one thread per (pair, wedge)
    if (i >= 2 * n_pairs) return;
    const int64_t pair = i >> 1;
    const int k = (int)(i & 1);
    const float ea = be::param2eta(params10[pair * 10 + 8 + k]);
    const float eb = be::param2eta(params10[(n_pairs + pair) * 10 + 8 + k]);
    int br;
    depth[i] = be::etas2depth(c, ea, eb, br);
}

inline int grid_for(int64_t n, int block) {
    int64_t g = (n + block - 1) / block;
    return (int)(g < 1 ? 1 : (g > 2048 ? 2048 : g));      // cap + grid-stride (memory-bound sizing)
}

}  // namespace

extern "C" int be_params2etas_f32(const float* p, float* eta, int64_t n, void* stream) {
    BE_REQUIRE(n >= 0, "be_params2etas_f32: n < 0");
    if (n == 0) return BE_OK;
    BE_REQUIRE(p && eta, "be_params2etas_f32: null pointer");
    hipLaunchKernelGGL(k_params2etas, dim3(grid_for(n, 256)), dim3(256), 0, be::as_stream(stream), p, eta, n);
    return be::check_launch("be_params2etas_f32");
}

extern "C" int be_etas2depth_f32(const be_depth_consts* c, const float* eta1, const float* eta2, float* depth,
                                 int32_t* branch, int64_t n, void* stream) {
    BE_REQUIRE(n >= 0, "be_etas2depth_f32: n < 0");
    if (n == 0) return BE_OK;
    BE_REQUIRE(c && eta1 && eta2 && depth, "be_etas2depth_f32: null pointer");
    hipLaunchKernelGGL(k_etas2depth, dim3(grid_for(n, 256)), dim3(256), 0, be::as_stream(stream), *c, eta1, eta2,
                       depth, branch, n);
    return be::check_launch("be_etas2depth_f32");
}

extern "C" int be_depth2sigma_f32(const be_depth_consts* c, const float* depth, float rho_prime, float* eta,
                                  int64_t n, void* stream) {
    BE_REQUIRE(n >= 0, "be_depth2sigma_f32: n < 0");
    if (n == 0) return BE_OK;
    BE_REQUIRE(c && depth && eta, "be_depth2sigma_f32: null pointer");
    hipLaunchKernelGGL(k_depth2sigma, dim3(grid_for(n, 256)), dim3(256), 0, be::as_stream(stream), *c, depth,
                       rho_prime, eta, n);
    return be::check_launch("be_depth2sigma_f32");
}

extern "C" int be_local_depth_f32(const be_depth_consts* c, const float* params10, float* depth, int64_t n_pairs,
                                  void* stream) {
    BE_REQUIRE(n_pairs >= 0, "be_local_depth_f32: n_pairs < 0");
    if (n_pairs == 0) return BE_OK;
    BE_REQUIRE(c && params10 && depth, "be_local_depth_f32: null pointer");
    const int64_t nt = 2 * n_pairs;
    hipLaunchKernelGGL(k_local_depth, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, be::as_stream(stream), *c,
                       params10, depth, n_pairs);
    return be::check_launch("be_local_depth_f32");
}

// ---- eval_depth (utils/metrics.py:3-20) on the device: delta1-3, RMSE (cm), AbsRel (cm) over the masked, cropped pixels
// of a [B,H,W] batch, summed jointly as the reference does.  One workgroup, fp64 accumulation, fixed order.
namespace {
__global__ __launch_bounds__(1024)
void k_eval_depth(const float* __restrict__ pred, const float* __restrict__ gt, const float* __restrict__ thr_src,
                  int B, int H, int W, int crop, float tau, float zmin, float zmax, double* __restrict__ out) {
    __shared__ double red[6][16];
    double acc[6] = {0, 0, 0, 0, 0, 0};               // n(d1), n(d2), n(d3), sum err^2, sum err/gt, count
    const int h = H - 2 * crop, w = W - 2 * crop;
    const int64_t total = (int64_t)B * h * w;
    const float span = zmax - zmin;
    const float t1 = tau, t2 = tau * tau, t3 = tau * tau * tau;
    for (int64_t i = threadIdx.x; i < total; i += blockDim.x) {
        const int x = (int)(i % w) + crop, y = (int)((i / w) % h) + crop;
        const int64_t e = ((i / ((int64_t)w * h)) * H + y) * W + x;
        if (!(thr_src[e] > 0.0f)) continue;            // mask = depth_map > 0 (blurry_edges_test.py:148)
        const float g = gt[e];
        const float p = fminf(fmaxf(pred[e], zmin), zmax);
        const float pn = fminf(fmaxf((p - zmin) / span, 0.f), 1.f), gn = fminf(fmaxf((g - zmin) / span, 0.f), 1.f);
        const float ratio = fmaxf(gn / (pn + 1e-8f), pn / (gn + 1e-8f));
        const float err = fabsf(g - p);
        acc[0] += ratio < t1; acc[1] += ratio < t2; acc[2] += ratio < t3;
        acc[3] += (double)err * err; acc[4] += (double)(err / g); acc[5] += 1.0;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        double v = acc[k];
        for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
        if (lane == 0) red[k][wave] = v;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double s[6];
        for (int k = 0; k < 6; ++k) { s[k] = 0; for (int wv = 0; wv < (int)(blockDim.x >> 6); ++wv) s[k] += red[k][wv]; }
        const double n = s[5];
        out[0] = s[0] / n; out[1] = s[1] / n; out[2] = s[2] / n;
        out[3] = sqrt(s[3] / n) * 100.0; out[4] = s[4] / n * 100.0;
    }
}
}  // namespace

extern "C" int be_eval_depth_f32(const float* pred, const float* gt, const float* mask_src, int B, int H, int W, int crop,
                                 float tau_n, float z_min, float z_max, double* out5, void* stream) {
    BE_REQUIRE(pred && gt && mask_src && out5, "be_eval_depth_f32: null pointer");
    BE_REQUIRE(B > 0 && H > 0 && W > 0 && crop >= 0 && 2 * crop < H && 2 * crop < W, "be_eval_depth_f32: bad shape / crop");
    BE_REQUIRE(z_max > z_min && tau_n > 0.f, "be_eval_depth_f32: bad range");
    hipLaunchKernelGGL(k_eval_depth, dim3(1), dim3(1024), 0, be::as_stream(stream), pred, gt, mask_src, B, H, W, crop, tau_n,
                       z_min, z_max, out5);
    return be::check_launch("be_eval_depth_f32");
}
