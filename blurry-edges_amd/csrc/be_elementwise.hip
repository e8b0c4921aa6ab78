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
