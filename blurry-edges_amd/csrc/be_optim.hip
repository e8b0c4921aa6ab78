// Gradient clipping + AdamW for a model whose gradients live in ONE flat buffer (what the LocalStage backward writes):
// the tail of a training step, local_training.py:107-108 (`clip_grad_norm_(max_norm=1)` + `optimizer.step()` with
// torch.optim.AdamW defaults).  Round 2 used the stock multi-tensor kernels: 13 launches, 0.16 ms of a 2.6 ms step.
// Here: one launch for the squared-norm partials, one for everything else (every workgroup re-derives the total norm from the
// partials in the same fixed order, so all agree bit for bit and no grid-wide barrier is needed).
#include "be_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int CHUNK = 4096;          // elements per workgroup (= per table entry)

__global__ __launch_bounds__(256)
void k_grad_sqsum(const float* __restrict__ g, int64_t n, double* __restrict__ partial, float* __restrict__ step) {
    __shared__ double red[256];
    // the step counter moves here, one launch ahead of its readers (every workgroup of k_clip_adamw reads it: none of them can
    // know when the others have) - a launch of its own in the first version
    if (blockIdx.x == 0 && threadIdx.x == 0) step[0] += 1.0f;
    const int64_t lo = (int64_t)blockIdx.x * CHUNK, hi = lo + CHUNK < n ? lo + CHUNK : n;
    double s = 0.0;
    if (((uintptr_t)g & 15u) == 0) {
        const int64_t hi4 = lo + ((hi - lo) & ~(int64_t)3);
        for (int64_t i = lo + 4 * threadIdx.x; i < hi4; i += 1024) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(g + i);
            s += (double)v[0] * v[0] + (double)v[1] * v[1] + (double)v[2] * v[2] + (double)v[3] * v[3];
        }
        for (int64_t i = hi4 + threadIdx.x; i < hi; i += 256) s += (double)g[i] * g[i];
    } else {
        for (int64_t i = lo + threadIdx.x; i < hi; i += 256) s += (double)g[i] * g[i];
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w >= 1; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

struct AdamArgs {
    const be_adam_entry* table;      // device: one entry per workgroup
    float* g;                        // flat gradient buffer
    const double* partial; int npartial;
    float max_norm, grad_scale;
    double lr, beta1, beta2, eps, weight_decay;   // hyper-parameters stay doubles (Python floats) until the last moment, as in torch: 1 - beta2 formed
                                                  // from the FLOAT 0.999 is off by 1.3e-5 relative
    const float* step;               // device scalar: the number of this step (k_grad_sqsum has already counted it)
    float* grad_norm;                // device scalar out (the norm before clipping), may be null
    int write_back;
};

__global__ __launch_bounds__(256)
void k_clip_adamw(AdamArgs a) {
    __shared__ double red[256];
    __shared__ float s_coef, s_bc1, s_bc2s;
    // total norm: every workgroup sums the same partials in the same order
    double s = 0.0;
    for (int i = threadIdx.x; i < a.npartial; i += 256) s += a.partial[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w >= 1; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float total = (float)(sqrt(red[0]) * (double)a.grad_scale);
        float coef = a.max_norm / (total + 1e-6f);            // torch.nn.utils.clip_grad_norm_
        if (coef > 1.0f) coef = 1.0f;
        if (a.max_norm <= 0.0f) coef = 1.0f;                  // clipping off
        s_coef = coef * a.grad_scale;
        const double t = (double)a.step[0];
        s_bc1 = (float)(1.0 - pow(a.beta1, t));
        s_bc2s = (float)sqrt(1.0 - pow(a.beta2, t));
        if (blockIdx.x == 0 && a.grad_norm) a.grad_norm[0] = total;
    }
    __syncthreads();
    const be_adam_entry e = a.table[blockIdx.x];
    const float coef = s_coef, bc1 = s_bc1, bc2s = s_bc2s;
    const float step_size = (float)(a.lr / (double)bc1), keep = (float)(1.0 - a.lr * a.weight_decay);
    const float omb1 = (float)(1.0 - a.beta1), omb2 = (float)(1.0 - a.beta2), b2 = (float)a.beta2, eps = (float)a.eps;
    float* gp = a.g + e.goff;
    auto upd = [&](float& p, float& m, float& v, float& g) {
        g *= coef;
        p *= keep;                                            // AdamW: decoupled weight decay, param.mul_(1 - lr * weight_decay)
        m = m + omb1 * (g - m);                               // lerp(m, g, 1 - beta1)
        v = b2 * v + omb2 * g * g;
        const float denom = sqrtf(v) / bc2s + eps;
        p -= step_size * (m / denom);                          // addcdiv_(exp_avg, denom, value=-step_size)
    };
    const bool vec = ((e.n & 3) == 0) && (((uintptr_t)e.p | (uintptr_t)e.m | (uintptr_t)e.v | (uintptr_t)gp) & 15u) == 0;
    if (vec) {
        for (int i = 4 * threadIdx.x; i < e.n; i += 1024) {
            f32x4 p = *reinterpret_cast<f32x4*>(e.p + i), m = *reinterpret_cast<f32x4*>(e.m + i);
            f32x4 v = *reinterpret_cast<f32x4*>(e.v + i), g = *reinterpret_cast<f32x4*>(gp + i);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float pk = p[k], mk = m[k], vk = v[k], gk = g[k];
                upd(pk, mk, vk, gk);
                p[k] = pk; m[k] = mk; v[k] = vk; g[k] = gk;
            }
            *reinterpret_cast<f32x4*>(e.p + i) = p; *reinterpret_cast<f32x4*>(e.m + i) = m; *reinterpret_cast<f32x4*>(e.v + i) = v;
            if (a.write_back) *reinterpret_cast<f32x4*>(gp + i) = g;
        }
    } else {
        for (int i = threadIdx.x; i < e.n; i += 256) {
            float p = e.p[i], m = e.m[i], v = e.v[i], g = gp[i];
            upd(p, m, v, g);
            e.p[i] = p; e.m[i] = m; e.v[i] = v;
            if (a.write_back) gp[i] = g;
        }
    }
}

}  // namespace

extern "C" int be_adam_chunk(void) { return CHUNK; }

extern "C" int be_clip_adamw_f32(const be_adam_entry* table_device, int nentries, float* grad_flat, int64_t n_flat, double* partial,
                                 int npartial_cap, float max_norm, float grad_scale, double lr, double beta1, double beta2, double eps,
                                 double weight_decay, float* step_device, float* grad_norm_out, int write_back, void* stream) {
    BE_REQUIRE(table_device && grad_flat && partial && step_device && nentries > 0 && n_flat > 0, "be_clip_adamw_f32: bad arguments");
    const int nblk = (int)((n_flat + CHUNK - 1) / CHUNK);
    BE_REQUIRE(nblk <= npartial_cap, "be_clip_adamw_f32: partial buffer too small (%d blocks)", nblk);
    hipStream_t s = be::as_stream(stream);
    hipLaunchKernelGGL(k_grad_sqsum, dim3(nblk), dim3(256), 0, s, grad_flat, n_flat, partial, step_device);
    AdamArgs a{table_device, grad_flat, partial, nblk, max_norm, grad_scale, lr, beta1, beta2, eps, weight_decay, step_device,
               grad_norm_out, write_back};
    hipLaunchKernelGGL(k_clip_adamw, dim3(nentries), dim3(256), 0, s, a);
    return be::check_launch("be_clip_adamw_f32");
}
