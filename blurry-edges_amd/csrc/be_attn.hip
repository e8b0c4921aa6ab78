// GlobalStage encoder pieces as HIP kernels (inference): fp32 flash-style multi-head attention on the f32 MFMA,
// residual + LayerNorm, positional-encoding add.  The linears (in-proj, out-proj, FFN, generator) run on the
// implicit-GEMM kernel of be_conv.hip as 1x1 "convs".
//
// Replaces nn.TransformerEncoderLayer (post-norm, ReLU FFN, eval mode) of models/global_stage.py:28-32 for
// L = 4096 tokens, d_model 128, 8 heads of 16.  The reference's eager attention materialises a [8,4096,4096]
// score tensor per layer (537 MB); here scores never leave registers.
//
// k_attention: one wavefront = 32 queries of one (batch, head), streaming all keys in blocks of 32.
//   S^T = K Q^T on v_mfma_f32_32x32x2_f32 with the KEY on the accumulator rows and the QUERY on the lane, so the
//   softmax statistics of a query are lane-local (16 scores in registers + one exchange with lane^32);
//   the probabilities stay in the accumulator registers and are fed straight back as the B operand of
//   O^T = V^T P^T: accumulator register r of lane-half h IS key (r&3)+8(r>>2)+4h, so using that key order for the
//   k-steps needs no data movement at all; V is read pre-transposed ([head][d][L]) so the A operand is 4 x 16-B loads.
//   d_head = 16 fills only half of the 32 accumulator rows of the PV product (the other half multiplies zeros).
#include "be_common.h"
#include "be_device_math.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int DH = 16;       // head dim
constexpr int KB = 32;       // keys per block

// qkv [T, 3*D] (token-major rows from the in-projection) -> Q [BH][L][16] (pre-scaled by scale*log2 e),
// K [BH][L][16], Vt [BH][16][L];  T = B*L, D = H*16
__global__ void k_qkv_split(const float* __restrict__ qkv, float* __restrict__ Q, float* __restrict__ K,
                            float* __restrict__ Vt, int B, int L, int H, float qscale) {
    const int64_t total = (int64_t)B * L * H * DH;
    const int64_t gs = (int64_t)gridDim.x * blockDim.x;
    const int D = H * DH;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gs) {
        const int d = (int)(idx % DH);
        const int h = (int)((idx / DH) % H);
        const int64_t t = idx / (DH * H);                 // token index in [0, B*L)
        const int64_t b = t / L, l = t % L;
        const float* row = qkv + t * 3 * D + h * DH + d;
        const int64_t bh = b * H + h;
        Q[(bh * L + l) * DH + d] = row[0] * qscale;
        K[(bh * L + l) * DH + d] = row[D];
        Vt[(bh * DH + d) * L + l] = row[2 * D];
    }
}

__global__ __launch_bounds__(256)
void k_attention(const float* __restrict__ Q, const float* __restrict__ K, const float* __restrict__ Vt,
                 float* __restrict__ out, int L, int H) {
    // grid.x = L/128 query blocks (4 waves x 32 queries), grid.y = B*H
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int j = lane & 31, h = lane >> 5;
    const int bh = blockIdx.y;
    const int q0 = (blockIdx.x * 4 + wave) * 32;
    const float* Qh = Q + (size_t)bh * L * DH;
    const float* Kh = K + (size_t)bh * L * DH;
    const float* Vh = Vt + (size_t)bh * DH * L;

    // B operand of S^T = K Q^T: this lane's query, d = 8h .. 8h+7
    const f32x4 qa = *reinterpret_cast<const f32x4*>(Qh + (size_t)(q0 + j) * DH + 8 * h);
    const f32x4 qb = *reinterpret_cast<const f32x4*>(Qh + (size_t)(q0 + j) * DH + 8 * h + 4);
    const float qf[8] = {qa.x, qa.y, qa.z, qa.w, qb.x, qb.y, qb.z, qb.w};

    f32x16 o;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = 0.f;
    float m = -INFINITY, lsum = 0.f;
    const bool vrow = j < DH;                              // lanes 16..31 of each half feed zero rows of V^T

    // prefetch block 0
    f32x4 ka = *reinterpret_cast<const f32x4*>(Kh + (size_t)j * DH + 8 * h);
    f32x4 kb = *reinterpret_cast<const f32x4*>(Kh + (size_t)j * DH + 8 * h + 4);
    f32x4 vt[4];
#pragma unroll
    for (int g = 0; g < 4; ++g)
        vt[g] = vrow ? *reinterpret_cast<const f32x4*>(Vh + (size_t)j * L + 8 * g + 4 * h) : f32x4{0.f, 0.f, 0.f, 0.f};

    const int nkb = L / KB;
    for (int kblk = 0; kblk < nkb; ++kblk) {
        const float kf[8] = {ka.x, ka.y, ka.z, ka.w, kb.x, kb.y, kb.z, kb.w};
        f32x4 vc[4] = {vt[0], vt[1], vt[2], vt[3]};
        // prefetch the next key/value block (clamped on the last iteration)
        const int kn = kblk + 1 < nkb ? kblk + 1 : kblk;
        ka = *reinterpret_cast<const f32x4*>(Kh + (size_t)(kn * KB + j) * DH + 8 * h);
        kb = *reinterpret_cast<const f32x4*>(Kh + (size_t)(kn * KB + j) * DH + 8 * h + 4);
#pragma unroll
        for (int g = 0; g < 4; ++g)
            vt[g] = vrow ? *reinterpret_cast<const f32x4*>(Vh + (size_t)j * L + kn * KB + 8 * g + 4 * h)
                         : f32x4{0.f, 0.f, 0.f, 0.f};

        // S^T[key][query] (log2 units: Q carries scale*log2 e)
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
        for (int t = 0; t < 8; ++t) s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[t], qf[t], s, 0, 0, 0);

        // online softmax for this lane's query: 16 keys here, 16 on lane^32
        float mloc = s[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mloc = fmaxf(mloc, s[r]);
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
        const float mnew = fmaxf(m, mloc);
        const float alpha = exp2f(m - mnew);
        m = mnew;
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = exp2f(s[r] - mnew); psum += s[r]; }
        lsum = lsum * alpha + psum;
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r] *= alpha;
        // O^T[d][query] += V^T[d][key] P[key][query]; k-step r uses key (r&3)+8(r>>2)+4h: register r of s as it stands
#pragma unroll
        for (int r = 0; r < 16; ++r) o = __builtin_amdgcn_mfma_f32_32x32x2f32(vc[r >> 2][r & 3], s[r], o, 0, 0, 0);
    }
    const float ltot = lsum + __shfl_xor(lsum, 32, 64);
    const float inv = 1.0f / ltot;
    // accumulator row d = (r&3) + 8*(r>>2) + 4h; rows >= 16 (r >= 8) are the zero padding
    const int Dm = H * DH;
    const int b = bh / H, hd = bh % H;
    float* dst = out + ((size_t)b * L + q0 + j) * Dm + hd * DH;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        f32x4 v = {o[4 * g] * inv, o[4 * g + 1] * inv, o[4 * g + 2] * inv, o[4 * g + 3] * inv};
        *reinterpret_cast<f32x4*>(dst + 8 * g + 4 * h) = v;
    }
}

// y = LayerNorm(x (+ res)) over the last dim D (= 128): one wave per row, two elements per lane
__global__ __launch_bounds__(256)
void k_add_layernorm(const float* __restrict__ x, const float* __restrict__ res, const float* __restrict__ gamma,
                     const float* __restrict__ beta, float* __restrict__ y, int64_t rows, int D, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + row * D;
    float v[4];
    float s = 0.f;
    const int per = D / 64;                                 // 2 for D = 128 (<= 4 supported)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (i < per) {
            v[i] = xr[lane + 64 * i] + (res ? res[row * D + lane + 64 * i] : 0.f);
            s += v[i];
        }
    }
    const float mean = be::wave_sum(s) / D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) if (i < per) { const float d = v[i] - mean; q += d * d; }
    const float rstd = 1.0f / sqrtf(be::wave_sum(q) / D + eps);
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (i < per) y[row * D + lane + 64 * i] = (v[i] - mean) * rstd * gamma[lane + 64 * i] + beta[lane + 64 * i];
}

// x[b][l][:] += pe[l][:]
__global__ void k_add_pe(float* __restrict__ x, const float* __restrict__ pe, int64_t total, int64_t per_batch) {
    const int64_t gs = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gs) x[i] += pe[i % per_batch];
}

}  // namespace

extern "C" int be_attention_f32(const float* qkv, float* out, float* workspace, int B, int L, int H, void* stream) {
    BE_REQUIRE(qkv && out && workspace, "be_attention_f32: null pointer");
    BE_REQUIRE(B > 0 && H > 0 && L > 0 && L % 128 == 0, "be_attention_f32: L must be a multiple of 128 (got %d)", L);
    BE_REQUIRE(be::aligned16(qkv) && be::aligned16(out) && be::aligned16(workspace), "be_attention_f32: 16-byte alignment");
    hipStream_t s = be::as_stream(stream);
    const size_t n = (size_t)B * H * L * DH;
    float *Q = workspace, *K = workspace + n, *Vt = workspace + 2 * n;
    const float qscale = 0.25f * 1.44269504088896340736f;            // 1/sqrt(16) * log2(e)
    int64_t g = ((int64_t)n + 255) / 256; if (g > 4096) g = 4096;
    hipLaunchKernelGGL(k_qkv_split, dim3((unsigned)g), dim3(256), 0, s, qkv, Q, K, Vt, B, L, H, qscale);
    hipLaunchKernelGGL(k_attention, dim3(L / 128, B * H), dim3(256), 0, s, Q, K, Vt, out, L, H);
    return be::check_launch("be_attention_f32");
}

extern "C" size_t be_attention_workspace_floats(int B, int L, int H) { return (size_t)3 * B * H * L * DH; }

extern "C" int be_add_layernorm_f32(const float* x, const float* res, const float* gamma, const float* beta, float* y,
                                    int64_t rows, int D, float eps, void* stream) {
    BE_REQUIRE(x && gamma && beta && y && rows > 0, "be_add_layernorm_f32: bad arguments");
    BE_REQUIRE(D % 64 == 0 && D <= 256, "be_add_layernorm_f32: D must be 64, 128, 192 or 256");
    hipLaunchKernelGGL(k_add_layernorm, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, be::as_stream(stream), x, res, gamma,
                       beta, y, rows, D, eps);
    return be::check_launch("be_add_layernorm_f32");
}

extern "C" int be_add_pe_f32(float* x, const float* pe, int64_t batches, int64_t per_batch, void* stream) {
    BE_REQUIRE(x && pe && batches > 0 && per_batch > 0, "be_add_pe_f32: bad arguments");
    const int64_t total = batches * per_batch;
    int64_t g = (total + 255) / 256; if (g > 4096) g = 4096;
    hipLaunchKernelGGL(k_add_pe, dim3((unsigned)g), dim3(256), 0, be::as_stream(stream), x, pe, total, per_batch);
    return be::check_launch("be_add_pe_f32");
}
