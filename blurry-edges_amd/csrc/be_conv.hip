// LocalStage convolutions as fp32-MFMA implicit GEMM on gfx950 (v_mfma_f32_32x32x2_f32: exact fp32,
// bit-for-bit an fmaf chain), with the conv bias + eval-mode BatchNorm folded into the packed weights and
// bias + optional residual add + Smish fused into the epilogue.
//
// Replaces nn.Conv2d + nn.BatchNorm2d (+ Smish, + residual add) of models/local_stage.py:11-17,20-28,34-37,
// 54-56 and nn.Linear + nn.BatchNorm1d + Smish of :44-50 (a Linear is the 1x1 case on a 1x1 image).
//
// GEMM view: D[M = N*H*W output pixels][Cout] = A[M][K] * B[K][Cout].  Activations are NHWC so that a K-chunk of
// input channels of one tap is contiguous per output pixel: the A tile is gathered straight from the activation
// tensor (zero-filled outside the image), never materialised (no im2col buffer).
//   - block tile BM x BN (BM 128 or 64, BN = 32*WN*NT in {32,64,96,128}), 256 threads = 4 waves, wave tile
//     (32*MT) x (32*NT); small-M problems (training batches) take the 64x64 / 128x32 tiles
//   - K-chunk BKT = 16 floats (templated 8/16/32; 16 measured best: 40 KB of LDS, three workgroups per CU); A/B chunks
//     are register-staged (issue the loads of chunk k+1, run the MFMAs of chunk k, then write the staged registers
//     to the other LDS buffer; one barrier per chunk); sched_barriers pin that order
//   - LDS rows padded by 4 floats: the ds_read_b128 fragment reads (16-lane groups) and the ds_write_b128 staging
//     writes are bank-conflict free
//   - fragment order inside a group of 8 k: lane half h holds k = 4h..4h+3, MFMA j consumes element j of both
//     operands, so A and B agree on k without any shuffle
//   - K order = (32-channel chunk outer, tap inner), weights packed in exactly that order so the B stream is linear
//   - pixel-major M tiles for large batches (a tile = one pixel position of BM images): the taps that fall into the
//     zero padding are skipped for the whole tile; an image group's tiles share one XCD (see the kernel)
//   - an optional second input x2 appends a 1x1 conv to the K loop (residual block: downsample branch fused in)
#include <mutex>
#include "be_common.h"
#include "be_device_math.h"
#include "be_igemm_body.h"
#include <cstdlib>

namespace {

using namespace be_igemm;

// __launch_bounds__(256, w): w = workgroups per CU the LDS admits (= waves per SIMD), so the register allocator may
// use 512/w registers (see be_igemm_body.h)
template <int WM, int WN, int MT, int NT, int MODE, int BKT, int PRIO, bool UNI = false>
__global__ __launch_bounds__(256, BKT == 16 ? (PRIO == 2 ? 4 : 3) : (BKT == 8 ? 4 : 2))
void k_conv_igemm(ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    conv_igemm_body<WM, WN, MT, NT, MODE, BKT, PRIO, UNI>(a, smem, blockIdx.x, blockIdx.y, blockIdx.z);
}

// K chunks (of bkt floats) all tiles of a launch walk together: pixel-major tiles visit only the taps inside the image
double executed_chunks(const ConvArgs& a, bool row8, int bkt) {
    double chunks = 0.0;
    const int ntap_all = row8 ? 7 : a.ks * a.ks, ncc = row8 ? 1 : a.nchunk / ntap_all;
    if (a.pixmaj) {
        const int half = a.ks >> 1;
        for (int py = 0; py < a.H; ++py)
            for (int px = 0; px < a.W; ++px) {
                int nt = 0;
                for (int t = 0; t < ntap_all; ++t) {
                    if (row8) nt += (unsigned)(py + t - 3) < (unsigned)a.H;
                    else nt += (unsigned)(py + t / a.ks - half) < (unsigned)a.H && (unsigned)(px + t % a.ks - half) < (unsigned)a.W;
                }
                chunks += (double)nt * ncc;
            }
        chunks *= (double)(a.m_tiles / a.HW) * a.n_tiles * (32 / bkt);
    } else {
        chunks = (double)a.m_tiles * a.n_tiles * a.nchunk * (32 / bkt);
    }
    if (a.x2) chunks += (double)a.m_tiles * a.n_tiles * (a.Cin2 / 32) * (32 / bkt);
    return chunks;
}

template <int WM, int WN, int MT, int NT, int MODE, int BKT = 32, int PRIO = 0, bool UNI = false>
int launch_conv(const ConvArgs& a, hipStream_t s, int kernel_id) {
    constexpr int BN = WN * NT * 32;
    constexpr int BM = WM * MT * 32;
    constexpr size_t lds = (size_t)2 * (BM + BN) * (BKT + 4) * sizeof(float);
    static be::DeviceFlags attr_set{};                      // dynamic-LDS cap raised once per device (thread-safe)
    if (int rc_ = be::ensure_dynamic_lds(reinterpret_cast<const void*>(&k_conv_igemm<WM, WN, MT, NT, MODE, BKT, PRIO, UNI>), lds, attr_set)) return rc_;
    // slots per XCD: flat tiles are dealt round-robin; pixel-major tiles keep a group of images on one XCD
    const int per_xcd = a.pixmaj == 1 ? ((a.m_tiles / a.HW + 7) / 8) * a.HW : (a.m_tiles + 7) / 8;
    const unsigned grid = (unsigned)(8 * per_xcd * a.n_tiles);
    {   // algorithmic work of this launch: 2*M*K*Cout with the REAL K (no padding); bytes = in + weights + out
        const double k_real = MODE == MODE_ROW8 ? 3.0 * 49.0 : (double)a.Cin * a.ks * a.ks;
        const double cin_real = MODE == MODE_ROW8 ? 3.0 : (double)a.Cin;
        // MFMA work actually issued: every tile runs (its number of K chunks) x (2*BM*BN*BKT) flops, padding included;
        // pixel-major tiles visit only the taps inside the image
        const double chunks = executed_chunks(a, MODE == MODE_ROW8, BKT);
        const double k2_real = a.x2 ? (double)a.Cin2 : 0.0;
        const double nb = a.nbatch > 1 ? a.nbatch : 1;
        be::ProfileScope prof(s, kernel_id, nb * 2.0 * a.M * (k_real + k2_real) * a.Cout,
                              nb * 4.0 * (a.M * (cin_real + k2_real) + (k_real + k2_real) * a.Cout + (double)a.M * a.Cout * (a.res ? 2 : 1)),
                              nb * chunks * 2.0 * BM * BN * BKT);
        hipLaunchKernelGGL((k_conv_igemm<WM, WN, MT, NT, MODE, BKT, PRIO, UNI>), dim3(grid, a.ksplit > 1 ? a.ksplit : 1, a.nbatch > 1 ? a.nbatch : 1),
                           dim3(256), lds, s, a);
    }
    return be::check_launch("be_conv_nhwc_f32");
}

// tuning knob for A/B runs: BE_CONV_VARIANT=32 selects the K-chunk-32 instantiations
inline int conv_variant() {
    static const int v = getenv("BE_CONV_VARIANT") ? atoi(getenv("BE_CONV_VARIANT")) : 0;
    return v;
}

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

inline int conv_nchunk(int cin, int ksize) { return ksize == 7 ? 7 : (cin / BK) * ksize * ksize; }

// ------------------------------------------------------------------------------------------- weight packing
struct PackArgs {
    const float *w, *b, *gamma, *beta, *mean, *var;
    float eps;
    int cout, cin, ks, chw_hw, cout_pad, ktot;
    float *pw, *pb;
    int row_stride, col_off, bias_add;   // destination row stride / first column (floats); 1: add to the bias instead of set
    int dgrad;      // 1: pack W^T with mirrored taps: rows = input channels, K = (cout chunk, tap) -> data-gradient conv
};

__device__ __forceinline__ void pack_body(const PackArgs& p) {
    const int64_t total = (int64_t)p.cout_pad * p.ktot;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += stride) {
        const int co = (int)(idx / p.ktot), k = (int)(idx % p.ktot);
        float v = 0.0f;
        if (p.dgrad) {
            // row = channel of the gradient being produced (an INPUT channel of the forward conv, in our NHWC order)
            const int row = co, taps = p.ks * p.ks;
            if (row < p.cin) {
                const int kc = k / BK, e = k % BK, cc = kc / taps, tap = kc % taps;
                const int oc = cc * BK + e;                                   // forward output channel
                const int ci_ref = p.chw_hw > 0 ? (row % (p.cin / p.chw_hw)) * p.chw_hw + row / (p.cin / p.chw_hw) : row;
                if (oc < p.cout) v = p.w[((size_t)oc * p.cin + ci_ref) * taps + (taps - 1 - tap)];
            }
            p.pw[(size_t)co * p.row_stride + p.col_off + k] = v;
            continue;
        }
        if (co < p.cout) {
            const float scale = p.gamma ? p.gamma[co] / sqrtf(p.var[co] + p.eps) : 1.0f;
            const int kc = k / BK, e = k % BK;
            if (p.ks == 7) {                           // conv1: chunk = kernel row, e = pixel*4 + channel
                const int kh = kc, kw = e >> 2, ci = e & 3;
                if (kw < 7 && ci < p.cin) v = p.w[((size_t)(co * p.cin + ci) * 7 + kh) * 7 + kw] * scale;
            } else {
                const int taps = p.ks * p.ks;
                const int cc = kc / taps, tap = kc % taps;
                int ci = cc * BK + e;                  // NHWC channel index of the consumer
                size_t src;
                if (p.chw_hw > 0) {                    // features flattened from (C,H,W): ours are (H,W,C)
                    const int c_real = ci % (p.cin / p.chw_hw), hw = ci / (p.cin / p.chw_hw);
                    src = (size_t)co * p.cin + (size_t)c_real * p.chw_hw + hw;
                } else {
                    src = ((size_t)(co * p.cin + ci)) * taps + tap;
                }
                v = p.w[src] * scale;
            }
        }
        p.pw[(size_t)co * p.row_stride + p.col_off + k] = v;
    }
    // bias: one thread per output channel
    for (int64_t co = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; co < p.cout_pad; co += stride) {
        float v = 0.0f;
        if (!p.dgrad && co < p.cout) {
            const float b = p.b ? p.b[co] : 0.0f;
            if (p.gamma) v = (b - p.mean[co]) * (p.gamma[co] / sqrtf(p.var[co] + p.eps)) + p.beta[co];
            else v = b;
        }
        if (p.bias_add) p.pb[co] += v; else p.pb[co] = v;
    }
}

__global__ void k_pack(PackArgs p) { pack_body(p); }

// The job-table form of the same packs (a training step re-packs 29 MB of weights twice, fc.1's 1024 x 3456 matrix being half of
// it): pack_body's element-per-thread walk spends its time in 64-bit divisions and, for the transposed pack, reads one float per
// cache line.  Here every source element is read once in runs of >= 64 B and takes its new place through LDS (round 3: 54 -> see
// profiles/r03_local_train_step_sequence.txt).  Same values in the same places as pack_body (tests compare the two).
constexpr int PACK_ROW_MAX = 4096;                  // floats of one source row (cin * taps) the forward form stages
constexpr int PACK_LDS_FLOATS = PACK_ROW_MAX > BK * (16 * 9 + 1) ? PACK_ROW_MAX : BK * (16 * 9 + 1);
static_assert(BK == 32, "the pack tiles below are laid out for 32-channel chunks");

template <int TAPS>
__device__ __forceinline__ void pack_fwd_rows(const PackArgs& p, float* lds) {
    // out[row][(cc * TAPS + tap) * BK + e] = w[row][cc * BK + e][tap] * scale: a permutation of the row, which is contiguous in w
    const int rowlen = p.cin * TAPS, cpg = p.chw_hw > 0 ? p.cin / p.chw_hw : 1;
    for (int row = blockIdx.x; row < p.cout_pad; row += gridDim.x) {
        const bool real = row < p.cout;
        __syncthreads();
        if (real)
            for (int i0 = threadIdx.x; i0 < rowlen; i0 += 4 * blockDim.x) {      // four loads in flight per thread
                float t[4];
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (i0 + j * (int)blockDim.x < rowlen) t[j] = p.w[(size_t)row * rowlen + i0 + j * blockDim.x];
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (i0 + j * (int)blockDim.x < rowlen) lds[i0 + j * blockDim.x] = t[j];
            }
        __syncthreads();
        const float scale = real && p.gamma ? p.gamma[row] / sqrtf(p.var[row] + p.eps) : 1.0f;
        float* out = p.pw + (size_t)row * p.row_stride + p.col_off;
        for (int k = threadIdx.x; k < p.ktot; k += blockDim.x) {
            const int kc = k / BK, e = k % BK, cc = kc / TAPS, tap = kc - cc * TAPS, ci = cc * BK + e;
            int src;
            if (p.chw_hw > 0) { const int hw = ci / cpg; src = (ci - hw * cpg) * p.chw_hw + hw; }
            else src = ci * TAPS + tap;
            out[k] = real ? lds[src] * scale : 0.0f;
        }
    }
}

template <int TAPS>
__device__ __forceinline__ void pack_dgrad_tiles(const PackArgs& p, float* lds) {
    // out[row(ci)][(cc * TAPS + tap) * BK + e] = w[cc * BK + e][ci][TAPS - 1 - tap]: tiles of one chunk of 32 output channels x 16
    // input channels x TAPS, read as 32 runs of 16 * TAPS floats, written as runs of 32
    constexpr int CT = 16, LD = CT * TAPS + 1;
    const int ncc = p.cout / BK, nct = p.cout_pad / CT, cpg = p.chw_hw > 0 ? p.cin / p.chw_hw : 1;
    for (int u = blockIdx.x; u < ncc * nct; u += gridDim.x) {
        const int cc = u / nct, ct = u - cc * nct, ci0 = ct * CT;
        __syncthreads();
        {
            const int left = p.cin - ci0, run = (left < CT ? (left > 0 ? left : 0) : CT) * TAPS;
            // (both output-channel halves and all positions of a run as independent loads: up to 2 x 9 in flight per thread)
            constexpr int NI = (CT * TAPS + 15) / 16;
            float t[2][NI];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float* src = p.w + ((size_t)(cc * BK + (threadIdx.x >> 4) + 16 * h) * p.cin + ci0) * TAPS;
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    if ((int)(threadIdx.x & 15) + 16 * j < run) t[h][j] = src[(threadIdx.x & 15) + 16 * j];
            }
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    if ((int)(threadIdx.x & 15) + 16 * j < run) lds[((threadIdx.x >> 4) + 16 * h) * LD + (threadIdx.x & 15) + 16 * j] = t[h][j];
        }
        __syncthreads();
        const int e = threadIdx.x & 31;
        for (int l = threadIdx.x >> 5; l < CT; l += 8) {
            const int ci = ci0 + l;                                // the reference's input channel index
            // our NHWC row of that channel (features flattened from (C,H,W): models/local_stage.py:73)
            int row = ci;
            if (p.chw_hw > 0 && ci < p.cin) { const int c_real = ci / p.chw_hw, hw = ci - c_real * p.chw_hw; row = hw * cpg + c_real; }
            float* out = p.pw + (size_t)row * p.row_stride + p.col_off + (size_t)cc * TAPS * BK + e;
#pragma unroll
            for (int tap = 0; tap < TAPS; ++tap) out[tap * BK] = ci < p.cin ? lds[e * LD + l * TAPS + (TAPS - 1 - tap)] : 0.0f;
        }
    }
}

__device__ __forceinline__ void pack_bias(const PackArgs& p) {
    const int stride = gridDim.x * blockDim.x;
    for (int co = blockIdx.x * blockDim.x + threadIdx.x; co < p.cout_pad; co += stride) {
        float v = 0.0f;
        if (!p.dgrad && co < p.cout) {
            const float b = p.b ? p.b[co] : 0.0f;
            if (p.gamma) v = (b - p.mean[co]) * (p.gamma[co] / sqrtf(p.var[co] + p.eps)) + p.beta[co];
            else v = b;
        }
        if (p.bias_add) p.pb[co] += v; else p.pb[co] = v;
    }
}

// a table of pack jobs in ONE launch (blockIdx.y = job): a training step re-packs every layer's weights for the forward
// and for the data-gradient convolution, 28 launches of a few microseconds otherwise
__global__ void k_pack_jobs(const be_pack_job* __restrict__ jobs) {
    const be_pack_job j = jobs[blockIdx.y];
    PackArgs p;
    p.w = j.weight; p.b = j.bias; p.gamma = j.bn_gamma; p.beta = j.bn_beta; p.mean = j.bn_mean; p.var = j.bn_var;
    p.eps = j.bn_eps; p.cout = j.cout; p.cin = j.cin; p.ks = j.ksize; p.chw_hw = j.layout_chw_hw;
    p.pw = j.packed_w; p.pb = j.packed_bias; p.col_off = 0; p.bias_add = 0; p.dgrad = j.dgrad;
    if (j.dgrad) { p.cout_pad = (j.cin + 31) / 32 * 32; p.ktot = (j.cout / BK) * j.ksize * j.ksize * BK; }
    else { p.cout_pad = (j.cout + 31) / 32 * 32; p.ktot = (j.ksize == 7 ? 7 : (j.cin / BK) * j.ksize * j.ksize) * BK; }
    p.row_stride = p.ktot;
    __shared__ float lds[PACK_LDS_FLOATS];
    const int taps = j.ksize * j.ksize;
    if (j.ksize == 7 || (!j.dgrad && (j.cin * taps > PACK_ROW_MAX || j.cin % BK))) { pack_body(p); return; }
    if (j.dgrad) { if (taps == 9) pack_dgrad_tiles<9>(p, lds); else pack_dgrad_tiles<1>(p, lds); }
    else { if (taps == 9) pack_fwd_rows<9>(p, lds); else pack_fwd_rows<1>(p, lds); }
    pack_bias(p);
}

// ------------------------------------------------------------------------------------------- pooling / layout
__global__ void k_maxpool_nhwc(const float* __restrict__ x, float* __restrict__ y, int n, int h, int w, int c4,
                               int oh, int ow, int k, int stride, int pad, int ldx4) {
    const int64_t total = (int64_t)n * oh * ow * c4;
    const int64_t gs = (int64_t)gridDim.x * blockDim.x;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gs) {
        const int cq = (int)(idx % c4);
        int64_t t = idx / c4;
        const int ox = (int)(t % ow); t /= ow;
        const int oy = (int)(t % oh);
        const int64_t img = t / oh;
        float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        if (k == 3 && 2 * pad <= k) {
            // the nine taps as nine independent loads (a tap outside the image re-reads the nearest one inside - the maximum does
            // not change): with a branch per tap the loads went out one at a time and the kernel waited 85 % of its wave cycles
            float4 v[9];
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                const int yy = min(max(oy * stride - pad + dy, 0), h - 1);
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const int xx = min(max(ox * stride - pad + dx, 0), w - 1);
                    v[3 * dy + dx] = reinterpret_cast<const float4*>(x)[((img * h + yy) * w + xx) * ldx4 + cq];
                }
            }
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                m.x = fmaxf(m.x, v[t].x); m.y = fmaxf(m.y, v[t].y); m.z = fmaxf(m.z, v[t].z); m.w = fmaxf(m.w, v[t].w);
            }
            reinterpret_cast<float4*>(y)[idx] = m;
            continue;
        }
        for (int dy = 0; dy < k; ++dy) {
            const int yy = oy * stride - pad + dy;
            if ((unsigned)yy >= (unsigned)h) continue;
            for (int dx = 0; dx < k; ++dx) {
                const int xx = ox * stride - pad + dx;
                if ((unsigned)xx >= (unsigned)w) continue;
                const float4 v = reinterpret_cast<const float4*>(x)[((img * h + yy) * w + xx) * ldx4 + cq];
                m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
            }
        }
        reinterpret_cast<float4*>(y)[idx] = m;
    }
}

__global__ void k_nchw3_to_nhwc4(const float* __restrict__ x, float* __restrict__ y, int64_t n, int hw) {
    const int64_t total = n * hw;
    const int64_t gs = (int64_t)gridDim.x * blockDim.x;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gs) {
        const int64_t img = idx / hw;
        const int p = (int)(idx - img * hw);
        const float* s = x + img * 3 * hw + p;
        reinterpret_cast<float4*>(y)[idx] = make_float4(s[0], s[hw], s[2 * hw], 0.0f);
    }
}

// the same staging, but gathering the 21x21 window of patch (first + i) straight from the image pair
__global__ void k_view_to_nhwc4(be_patch_view v, int64_t P, int64_t first, float* __restrict__ y, int64_t n) {
    const int64_t total = n * BE_NPIX;
    const int64_t gs = (int64_t)gridDim.x * blockDim.x;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gs) {
        const int64_t patch = first + idx / BE_NPIX;
        const int p = (int)(idx % BE_NPIX);
        const int row = p / BE_R, col = p - row * BE_R;
        const int64_t pg = patch % P;
        const float* s = v.base + (patch / P) * v.s_aperture + (pg / v.wp) * v.s_pi + (pg % v.wp) * v.s_pj +
                         row * v.s_row + col * v.s_col;
        reinterpret_cast<float4*>(y)[idx] = make_float4(s[0], s[v.s_chan], s[2 * v.s_chan], 0.0f);
    }
}

// the padded form of the two stagings: [n,h,wrow,4], image column c at padded column c + 3, zeros elsewhere (the 8-pixel
// kernel rows of conv1 then never leave a row: be_conv_pm.hip)
__global__ void k_nchw3_to_nhwc4p(const float* __restrict__ x, float* __restrict__ y, int64_t n, int h, int w, int wrow) {
    const int64_t total = n * h * wrow;
    const int64_t gs = (int64_t)gridDim.x * blockDim.x;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gs) {
        const int64_t rowi = idx / wrow;
        const int col = (int)(idx - rowi * wrow) - 3;
        const int64_t img = rowi / h;
        const int row = (int)(rowi - img * h);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if ((unsigned)col < (unsigned)w) {
            const float* s = x + img * 3 * h * w + row * w + col;
            v = make_float4(s[0], s[h * w], s[2 * h * w], 0.0f);
        }
        reinterpret_cast<float4*>(y)[idx] = v;
    }
}

__global__ void k_view_to_nhwc4p(be_patch_view v, int64_t P, int64_t first, float* __restrict__ y, int64_t n, int wrow) {
    const int64_t total = n * BE_R * wrow;
    const int64_t gs = (int64_t)gridDim.x * blockDim.x;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gs) {
        const int64_t rowi = idx / wrow;
        const int col = (int)(idx - rowi * wrow) - 3;
        const int64_t patch = first + rowi / BE_R;
        const int row = (int)(rowi % BE_R);
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        if ((unsigned)col < (unsigned)BE_R) {
            const int64_t pg = patch % P;
            const float* s = v.base + (patch / P) * v.s_aperture + (pg / v.wp) * v.s_pi + (pg % v.wp) * v.s_pj +
                             row * v.s_row + col * v.s_col;
            o = make_float4(s[0], s[v.s_chan], s[2 * v.s_chan], 0.0f);
        }
        reinterpret_cast<float4*>(y)[idx] = o;
    }
}

inline unsigned grid_cap(int64_t total, int block) {
    int64_t g = (total + block - 1) / block;
    return (unsigned)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

}  // namespace

extern "C" size_t be_conv_packed_floats(int cout, int cin, int ksize) {
    if (cout <= 0 || cin <= 0) return 0;
    if (ksize == 7) return (size_t)round_up(cout, 32) * 7 * BK;
    if ((ksize != 1 && ksize != 3) || cin % BK) return 0;
    return (size_t)round_up(cout, 32) * conv_nchunk(cin, ksize) * BK;
}

extern "C" int be_conv_pack_f32(const float* w, const float* b, const float* g, const float* beta, const float* mean,
                                const float* var, float eps, int cout, int cin, int ksize, int chw_hw, float* pw,
                                float* pb, void* stream) {
    BE_REQUIRE(w && pw && pb, "be_conv_pack_f32: null pointer");
    BE_REQUIRE(cout > 0 && cin > 0, "be_conv_pack_f32: bad channel counts");
    BE_REQUIRE((g == nullptr) == (beta == nullptr) && (g == nullptr) == (mean == nullptr) &&
               (g == nullptr) == (var == nullptr), "be_conv_pack_f32: BatchNorm tensors must be all set or all null");
    // (the fourth channel of the NHWC4 staging is padding: the large-batch kernel does not even issue its products)
    if (ksize == 7) BE_REQUIRE(cin <= 3, "be_conv_pack_f32: the 7x7 row-gather mode takes cin <= 3 (got %d)", cin);
    else BE_REQUIRE((ksize == 1 || ksize == 3) && cin % BK == 0,
                    "be_conv_pack_f32: ksize %d / cin %d unsupported (ksize 1|3 with cin %% 32 == 0, or 7)", ksize, cin);
    BE_REQUIRE(chw_hw == 0 || (ksize == 1 && cin % chw_hw == 0), "be_conv_pack_f32: bad layout_chw_hw");
    const int ktot = conv_nchunk(cin, ksize) * BK;
    PackArgs p{w, b, g, beta, mean, var, eps, cout, cin, ksize, chw_hw, round_up(cout, 32), ktot, pw, pb, ktot, 0, 0, 0};
    hipLaunchKernelGGL(k_pack, dim3(grid_cap((int64_t)p.cout_pad * p.ktot, 256)), dim3(256), 0, be::as_stream(stream), p);
    return be::check_launch("be_conv_pack_f32");
}

extern "C" size_t be_conv_fused2_packed_floats(int cout, int cin, int ksize, int cin2) {
    const size_t k1 = be_conv_packed_floats(cout, cin, ksize);
    if (!k1 || cin2 <= 0 || cin2 % BK) return 0;
    return k1 + (size_t)round_up(cout, 32) * cin2;
}

extern "C" int be_conv_pack_fused2_f32(const float* w, const float* b, const float* g, const float* beta, const float* mean,
                                       const float* var, const float* w2, const float* b2, const float* g2,
                                       const float* beta2, const float* mean2, const float* var2, float eps, int cout,
                                       int cin, int ksize, int cin2, float* pw, float* pb, void* stream) {
    BE_REQUIRE(w && w2 && pw && pb, "be_conv_pack_fused2_f32: null pointer");
    BE_REQUIRE((ksize == 1 || ksize == 3) && cin % BK == 0 && cin2 > 0 && cin2 % BK == 0 && cout > 0,
               "be_conv_pack_fused2_f32: ksize 1|3, cin and cin2 %% 32 == 0");
    BE_REQUIRE((g == nullptr) == (beta == nullptr) && (g == nullptr) == (mean == nullptr) && (g == nullptr) == (var == nullptr) &&
               (g2 == nullptr) == (beta2 == nullptr) && (g2 == nullptr) == (mean2 == nullptr) && (g2 == nullptr) == (var2 == nullptr),
               "be_conv_pack_fused2_f32: BatchNorm tensors must be all set or all null per branch");
    const int k1 = conv_nchunk(cin, ksize) * BK, ktot = k1 + cin2, cp = round_up(cout, 32);
    hipStream_t s = be::as_stream(stream);
    PackArgs p1{w, b, g, beta, mean, var, eps, cout, cin, ksize, 0, cp, k1, pw, pb, ktot, 0, 0, 0};
    hipLaunchKernelGGL(k_pack, dim3(grid_cap((int64_t)cp * k1, 256)), dim3(256), 0, s, p1);
    PackArgs p2{w2, b2, g2, beta2, mean2, var2, eps, cout, cin2, 1, 0, cp, cin2, pw, pb, ktot, k1, 1, 0};
    hipLaunchKernelGGL(k_pack, dim3(grid_cap((int64_t)cp * cin2, 256)), dim3(256), 0, s, p2);
    return be::check_launch("be_conv_pack_fused2_f32");
}

extern "C" size_t be_conv_dgrad_packed_floats(int cout, int cin, int ksize) {
    if (cout <= 0 || cin <= 0 || cout % BK || (ksize != 1 && ksize != 3)) return 0;
    return (size_t)round_up(cin, 32) * (cout / BK) * ksize * ksize * BK;
}

extern "C" int be_conv_pack_dgrad_f32(const float* w, int cout, int cin, int ksize, int chw_hw, float* pw, float* pb,
                                      void* stream) {
    BE_REQUIRE(w && pw && pb, "be_conv_pack_dgrad_f32: null pointer");
    BE_REQUIRE(cout > 0 && cin > 0 && cout % BK == 0 && (ksize == 1 || ksize == 3),
               "be_conv_pack_dgrad_f32: needs cout %% 32 == 0 and ksize 1|3 (got cout %d, ksize %d)", cout, ksize);
    BE_REQUIRE(chw_hw == 0 || (ksize == 1 && cin % chw_hw == 0), "be_conv_pack_dgrad_f32: bad layout_chw_hw");
    const int kt = (cout / BK) * ksize * ksize * BK;
    PackArgs p{w, nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, cout, cin, ksize, chw_hw, round_up(cin, 32), kt, pw, pb,
               kt, 0, 0, 1};
    hipLaunchKernelGGL(k_pack, dim3(grid_cap((int64_t)p.cout_pad * p.ktot, 256)), dim3(256), 0, be::as_stream(stream), p);
    return be::check_launch("be_conv_pack_dgrad_f32");
}

extern "C" int be_conv_pack_jobs_f32(const be_pack_job* jobs_device, int njobs, void* stream) {
    BE_REQUIRE(jobs_device && njobs > 0 && njobs <= 65535, "be_conv_pack_jobs_f32: bad arguments");
    // 512 workgroups per job: the launch costs ~0.5 ns per workgroup whether or not it has work (4096 per job: 62 us, all dispatch);
    // 256 ... 1536 differ by 3 us of a 1.52 ms step
    hipLaunchKernelGGL(k_pack_jobs, dim3(512, njobs), dim3(256), 0, be::as_stream(stream), jobs_device);
    return be::check_launch("be_conv_pack_jobs_f32");
}

struct BatchParams { int n; int64_t xb, wb, yb; };
static thread_local BatchParams g_batch = {1, 0, 0, 0};   // set around one dispatch by be_conv_nhwc_batched_f32

// y = act(sum_s partial[s] + bias (+ res)): the epilogue of a split-K launch, fixed summation order
__global__ void k_splitk_reduce(const float* __restrict__ partial, int S, int64_t M, int Cout, int ldp,
                                const float* __restrict__ bias, const float* __restrict__ res, int act,
                                float* __restrict__ y, int ldy) {
    const int64_t total = M * Cout;
    const int64_t gs = (int64_t)gridDim.x * blockDim.x;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gs) {
        const int64_t m = idx / Cout;
        const int c = (int)(idx - m * Cout);
        float v = partial[m * ldp + c];
        for (int s = 1; s < S; ++s) v += partial[((int64_t)s * M + m) * ldp + c];
        if (bias) v += bias[c];
        if (res) v += res[m * ldy + c];
        if (act == 1) v = be::smish(v);
        else if (act == 2) v = fmaxf(v, 0.0f);
        y[m * ldy + c] = v;
    }
}

// defer (training units): when the K loop was split the S raw slices stay in scratch [S][M][ldp] and NO reduce kernel is launched -
// the caller's own kernel sums them (with the bias) while it does its other work on the tile; S = 1: y = conv + bias as always
struct SplitOut { int S, ldp; be::ConvPrep* prep; bool pad64; };   // prep != null: do not launch at all - hand the prepared launch back;
                                                                    // pad64: 96 output channels may run as two 64-wide uniform tiles
static int conv_dispatch(const be_conv_desc* d, const float* x, const float* x2, int cin2, const float* pw, const float* pb,
                         const float* res, float* y, int ldy, void* stream, void* scratch = nullptr, size_t scratch_bytes = 0,
                         SplitOut* defer = nullptr);

extern "C" int be_conv_nhwc_f32(const be_conv_desc* d, const float* x, const float* pw, const float* pb,
                                const float* res, float* y, int ldy, void* stream) {
    return conv_dispatch(d, x, nullptr, 0, pw, pb, res, y, ldy, stream);
}

extern "C" int be_conv_nhwc_fused2_f32(const be_conv_desc* d, const float* x, const float* x2, int cin2, const float* pw,
                                       const float* pb, float* y, int ldy, void* stream) {
    BE_REQUIRE(x2 && cin2 > 0 && cin2 % BK == 0, "be_conv_nhwc_fused2_f32: x2 / cin2 (%% 32 == 0) required");
    BE_REQUIRE(d && d->ksize != 7, "be_conv_nhwc_fused2_f32: not available for the 7x7 row-gather mode");
    BE_REQUIRE(be::aligned16(x2), "be_conv_nhwc_fused2_f32: x2 must be 16-byte aligned");
    return conv_dispatch(d, x, x2, cin2, pw, pb, nullptr, y, ldy, stream);
}

extern "C" int be_conv_nhwc_batched_f32(const be_conv_desc* d, const float* x, const float* pw, const float* pb, float* y, int ldy,
                                        int nbatch, int64_t x_stride, int64_t w_stride, int64_t y_stride, void* stream) {
    BE_REQUIRE(nbatch >= 1 && nbatch <= 65535, "be_conv_nhwc_batched_f32: nbatch outside [1, 65535]");
    BE_REQUIRE(x_stride % 4 == 0 && w_stride % 4 == 0, "be_conv_nhwc_batched_f32: strides must keep 16-byte alignment");
    g_batch = {nbatch, x_stride, w_stride, y_stride};
    const int rc = conv_dispatch(d, x, nullptr, 0, pw, pb, nullptr, y, ldy, stream);
    g_batch = {1, 0, 0, 0};
    return rc;
}

extern "C" int be_conv_nhwc_splitk_f32(const be_conv_desc* d, const float* x, const float* pw, const float* pb,
                                       const float* res, float* y, int ldy, void* scratch, size_t scratch_bytes, void* stream) {
    BE_REQUIRE(scratch && be::aligned16(scratch), "be_conv_nhwc_splitk_f32: scratch required (16-byte aligned)");
    return conv_dispatch(d, x, nullptr, 0, pw, pb, res, y, ldy, stream, scratch, scratch_bytes);
}

int be::conv_train(const be_conv_desc* d, const float* x, const float* pw, const float* pb, const float* res, float* y, int ldy,
                   void* scratch, size_t scratch_bytes, int* S_out, int* ldp_out, void* stream) {
    SplitOut so{1, 0, nullptr, false};
    const int rc = conv_dispatch(d, x, nullptr, 0, pw, pb, res, y, ldy, stream, scratch, scratch_bytes, &so);
    *S_out = so.S; *ldp_out = so.ldp;
    return rc;
}

int be::conv_train_prepare(const be_conv_desc* d, const float* x, const float* pw, const float* pb, const float* res, float* y, int ldy,
                           void* scratch, size_t scratch_bytes, be::ConvPrep* prep, bool pad64) {
    prep->variant = -1;
    SplitOut so{1, 0, prep, pad64};
    const int rc = conv_dispatch(d, x, nullptr, 0, pw, pb, res, y, ldy, nullptr, scratch, scratch_bytes, &so);
    return rc;        // prep->variant stays -1 for a shape outside the small-M tiles: the caller launches it on its own instead
}

static int conv_dispatch(const be_conv_desc* d, const float* x, const float* x2, int cin2, const float* pw, const float* pb,
                         const float* res, float* y, int ldy, void* stream, void* scratch, size_t scratch_bytes, SplitOut* defer) {
    BE_REQUIRE(d && x && pw && y, "be_conv_nhwc_f32: null pointer");
    BE_REQUIRE(d->n > 0 && d->h > 0 && d->w > 0 && d->cout > 0, "be_conv_nhwc_f32: empty shape");
    BE_REQUIRE(d->h < 32768 && d->w < 32768, "be_conv_nhwc_f32: image too large");
    const bool row8 = d->ksize == 7;
    if (row8) BE_REQUIRE(d->cin == 4, "be_conv_nhwc_f32: ksize 7 needs the NHWC4 input (cin = 4)");
    else BE_REQUIRE((d->ksize == 1 || d->ksize == 3) && d->cin % BK == 0,
                    "be_conv_nhwc_f32: ksize %d / cin %d unsupported", d->ksize, d->cin);
    BE_REQUIRE(ldy >= d->cout, "be_conv_nhwc_f32: ldy < cout");
    BE_REQUIRE(be::aligned16(x) && be::aligned16(pw), "be_conv_nhwc_f32: x / packed_w must be 16-byte aligned");
    const int64_t M = (int64_t)d->n * d->h * d->w;
    BE_REQUIRE(M * (int64_t)d->cin < (int64_t)1 << 40 && M < (int64_t)1 << 31, "be_conv_nhwc_f32: batch too large");
    ConvArgs a;
    a.x = x; a.w = pw; a.bias = pb; a.res = res; a.y = y;
    a.M = (int)M; a.H = d->h; a.W = d->w; a.HW = d->h * d->w; a.Cin = d->cin; a.Cout = d->cout; a.ldy = ldy;
    a.ks = d->ksize; a.nchunk = conv_nchunk(d->cin, d->ksize); a.Ktot = a.nchunk * BK + (x2 ? cin2 : 0); a.act = d->act;
    a.x2 = x2; a.Cin2 = cin2;
    a.m_tiles = (int)((M + 127) / 128);
    a.pixmaj = 0; a.Nimg = d->n;
    a.ksplit = 1; a.ldp = 0; a.partial = nullptr;
    a.nbatch = g_batch.n; a.xb = g_batch.xb; a.wb = g_batch.wb; a.yb = g_batch.yb;
    const int cp = round_up(d->cout, 32);
    hipStream_t s = be::as_stream(stream);
    // Small-M regime (training at batch 64: M = 2304 rows at 6x6): the 128-row tiles give a few dozen workgroups on
    // 256 CUs and one launch lasts as long as ONE workgroup's serial K loop.  Below ~1.5 workgroups per CU switch to
    // 64x64 (or 128x32) tiles: 4x the workgroups, each with a quarter of the work.
    static const bool no_rows = getenv("BE_NO_GEMM_ROWS") != nullptr;             // A/B knob
    // linears over many rows with a 128-wide output (GlobalStage at 8 x 4096 tokens: out-projection, second feed-forward linear
    // and three of the four data gradients): 256 row tiles x ONE column tile is "few tiles" for the rule below, which sent them to
    // 64x64 tiles (33 us per launch); they are plain row GEMMs for the LDS-DMA kernel (bit-identical, see further down)
    // prepare-only calls (be::conv_train_prepare: `stream` is not the caller's stream and the operands may not be final yet): a
    // shape that is not handed back as a prepared small-M launch must leave WITHOUT launching anything - prep->variant stays -1
    // and the caller runs the unit on its own (layer0 at batch >= 406 and wide linears over >= 16384 rows fell
    // through to a launch on the null stream here, inside a hipGraph capture among others)
    const bool prep_only = defer && defer->prep;
    if (d->ksize == 1 && !x2 && cp == 128 && M >= 16384 && a.nbatch <= 1 && !no_rows && d->cin % 16 == 0) {
        if (prep_only) return BE_OK;
        return be::gemm_rows(x, M, d->cin, pw, d->cout, pb, res, d->act, y, ldy, stream);
    }
    if (prep_only && !(!row8 && (int64_t)a.m_tiles * ((cp + 127) / 128) < 384 && scratch)) return BE_OK;
    if (!row8 && (int64_t)a.m_tiles * ((cp + 127) / 128) < 384) {
        // 96 output channels (the data gradients that flow into layer0 / layer1, whose inputs have 96 channels) as TWO 64-wide uniform
        // tiles, the second half empty (a third more MFMA work, on tiles that are twice as fast as the bordered 128 x 32 ones).  Only
        // where the caller allows it - the backward: the forward keeps its tiles, and with them its bits (pad64)
        const bool pad64 = defer && defer->pad64 && cp % 64 == 32 && cp > 64 && !x2 && a.nbatch <= 1 &&
                           ((d->ksize > 1 && d->n >= 64 && d->n % 64 == 0 && conv_variant() != 99) || (d->ksize == 1 && M % 64 == 0)) &&
                           M * (int64_t)d->cin * 4 < ((int64_t)1 << 32) && (int64_t)cp * a.Ktot * 4 < ((int64_t)1 << 32) &&
                           getenv("BE_NO_UNI_TILES") == nullptr && getenv("BE_NO_PAD64") == nullptr;      // = every condition of `uni` below
        const bool t64 = cp % 64 == 0 || pad64;
        // (round 3 tried 64 x 128 tiles - a wave owning 32 x 64 - for the training convolutions onto 384 channels: 79 -> 66 us and
        // 56 -> 48 us for those two layers, 1 % of the step; the narrower layers lose.  Not kept: the different split of the K loop
        // moves the train-mode logits by 1e-6, and on the teacher-forced test's worst-conditioned batch that is a 7e-6 loss
        // difference against a 3e-6 bound written for the 64 x 64 tiles)
        if (t64) {
            a.m_tiles = (int)((M + 63) / 64); a.n_tiles = (cp + 63) / 64;
            if (d->ksize > 1 && d->n >= 64 && d->n % 64 == 0 && conv_variant() != 99) {
                // full 64-image tiles (the training batch): pixel-major here too, tiles dealt round-robin over the
                // XCDs because there are only a few image groups
                a.pixmaj = 2;
                a.m_tiles = a.HW * (d->n / 64);
            }
        } else {
            a.n_tiles = cp / 32;
        }
        // uniform 64 x 64 tiles (be_igemm_body.h, UNI): the pixel-major tiles of whole 64-image groups, and 1x1 convolutions /
        // linears whose row count is a multiple of 64 - no border test, addresses = per-thread constant + uniform offset
        static const bool no_uni = getenv("BE_NO_UNI_TILES") != nullptr;              // A/B knob
        const bool uni = t64 && !no_uni && !x2 && a.nbatch <= 1 && M * (int64_t)d->cin * 4 < ((int64_t)1 << 32) &&
                         (int64_t)cp * a.Ktot * 4 < ((int64_t)1 << 32) && (a.pixmaj == 2 || (d->ksize == 1 && M % 64 == 0));
        // split-K when the caller lent scratch: a few hundred workgroups each walking the whole K loop leave most of
        // the chip idle (batch-64 training: 144-216 tiles, up to 216 chunks each); S slices make S times the
        // workgroups, each 1/S as long, and a small reduce kernel applies bias / residual / activation
        if (scratch) {
            const int64_t tiles = (int64_t)a.m_tiles * a.n_tiles;
            const int kchunks = a.nchunk * 2;                               // 16-float chunks of the longest tile
            int S = (int)((768 + tiles - 1) / tiles);
            if (S > 8) S = 8;
            while (S > 1 && kchunks / S < 12) --S;
            while (S > 1 && (size_t)S * M * cp * sizeof(float) > scratch_bytes) --S;
            if (defer && defer->prep) {                                     // hand the launch back (be_train.hip: k_unit_gemms)
                if (S > 1) { a.ksplit = S; a.ldp = cp; a.partial = static_cast<float*>(scratch); }
                be::ConvPrep* pr = defer->prep;
                pr->args = a; pr->variant = uni ? 2 : (t64 ? 0 : 1); pr->S = S > 1 ? S : 1; pr->ldp = S > 1 ? cp : 0;
                pr->gx = (unsigned)(8 * ((a.m_tiles + 7) / 8) * a.n_tiles);
                pr->flops = 2.0 * a.M * (double)a.Cin * a.ks * a.ks * a.Cout;
                pr->flops_exec = executed_chunks(a, false, 16) * 2.0 * (t64 ? 64.0 * 64.0 : 128.0 * 32.0) * 16.0;
                return BE_OK;
            }
            if (S > 1) {
                a.ksplit = S; a.ldp = cp; a.partial = static_cast<float*>(scratch);
                const int rc = uni ? launch_conv<2, 2, 1, 1, MODE_TAPS, 16, 0, true>(a, s, BE_KERNEL_CONV_SMALL)
                               : t64 ? launch_conv<2, 2, 1, 1, MODE_TAPS, 16, 0>(a, s, BE_KERNEL_CONV_SMALL)
                                     : launch_conv<4, 1, 1, 1, MODE_TAPS, 16, 0>(a, s, BE_KERNEL_CONV_SMALL);
                if (rc) return rc;
                if (defer) { defer->S = S; defer->ldp = cp; return BE_OK; }
                const int64_t total = M * d->cout;
                hipLaunchKernelGGL(k_splitk_reduce, dim3(grid_cap(total, 256)), dim3(256), 0, s, a.partial, S, M, d->cout, cp,
                                   pb, res, d->act, y, ldy);
                return be::check_launch("be_conv_nhwc_splitk_f32");
            }
        }
        return uni ? launch_conv<2, 2, 1, 1, MODE_TAPS, 16, 0, true>(a, s, BE_KERNEL_CONV_SMALL)
               : t64 ? launch_conv<2, 2, 1, 1, MODE_TAPS, 16, 0>(a, s, BE_KERNEL_CONV_SMALL)
                     : launch_conv<4, 1, 1, 1, MODE_TAPS, 16, 0>(a, s, BE_KERNEL_CONV_SMALL);
    }
    // 1x1 convolutions / linears of large batches are plain row GEMMs: the LDS-DMA kernel of be_wino.hip (bit-identical)
    if (d->ksize == 1 && !x2 && cp % 128 == 0 && M >= 4096 && a.nbatch <= 1 && !no_rows && d->cin % 16 == 0)
        return be::gemm_rows(x, M, d->cin, pw, d->cout, pb, res, d->act, y, ldy, stream);
    // large batches of small images, 3x3: the pixel-major LDS-DMA kernel (be_conv_pm.hip; bit-identical)
    if (d->ksize == 3 && d->n >= 512 && !res && a.nbatch <= 1 && conv_variant() == 0 && be::aligned16(y) && ldy % 4 == 0) {
        const int rc = be::conv_pm(d, x, d->w, x2, cin2, pw, pb, y, ldy, a.Ktot, stream);
        if (rc != 1) return rc;
    }
    // large batches of small images: pixel-major tiles skip the taps that fall into the zero padding
    if (d->ksize > 1 && d->n >= 512 && conv_variant() != 99) {
        a.pixmaj = 1;
        a.m_tiles = a.HW * ((d->n + 127) / 128);
    }
    if (row8) {
        BE_REQUIRE(cp == 64, "be_conv_nhwc_f32: ksize 7 is built for cout 64 (got %d)", d->cout);
        a.n_tiles = 1;
        if (conv_variant() == 32) return launch_conv<4, 1, 1, 2, MODE_ROW8, 32, 0>(a, s, BE_KERNEL_CONV_ROW8_128x64);
        return launch_conv<4, 1, 1, 2, MODE_ROW8, 16, 0>(a, s, BE_KERNEL_CONV_ROW8_128x64);
    }
    if (cp % 128 == 0) {
        a.n_tiles = cp / 128;
        // K-chunk 16 (40 KB of LDS, three workgroups per CU) measured 8 % faster than 32 (two per CU); 8 is slower
        // again (barrier per 16 MFMAs); s_setprio around the MFMA phase and a 4th workgroup per CU change nothing.
        if (conv_variant() == 32) return launch_conv<2, 2, 2, 2, MODE_TAPS, 32, 0>(a, s, BE_KERNEL_CONV_128x128);
        return launch_conv<2, 2, 2, 2, MODE_TAPS, 16, 0>(a, s, BE_KERNEL_CONV_128x128);
    }
    if (cp % 96 == 0) {
        a.n_tiles = cp / 96;
        if (conv_variant() == 32) return launch_conv<4, 1, 1, 3, MODE_TAPS, 32, 0>(a, s, BE_KERNEL_CONV_128x96);
        return launch_conv<4, 1, 1, 3, MODE_TAPS, 16, 0>(a, s, BE_KERNEL_CONV_128x96);
    }
    if (cp % 64 == 0)  { a.n_tiles = cp / 64;  return launch_conv<4, 1, 1, 2, MODE_TAPS>(a, s, BE_KERNEL_CONV_128x64); }
    a.n_tiles = cp / 32;
    return launch_conv<4, 1, 1, 1, MODE_TAPS>(a, s, BE_KERNEL_CONV_128x32);
}

extern "C" int be_maxpool_nhwc_f32(const float* x, float* y, int n, int h, int w, int c, int k, int stride, int pad,
                                   void* stream) {
    return be_maxpool_nhwc_ld_f32(x, c, y, n, h, w, c, k, stride, pad, stream);
}

extern "C" int be_maxpool_nhwc_ld_f32(const float* x, int ldx, float* y, int n, int h, int w, int c, int k, int stride,
                                      int pad, void* stream) {
    BE_REQUIRE(x && y, "be_maxpool_nhwc_f32: null pointer");
    BE_REQUIRE(n > 0 && h > 0 && w > 0 && c > 0 && c % 4 == 0, "be_maxpool_nhwc_f32: bad shape (c %% 4 == 0)");
    BE_REQUIRE(ldx >= c && ldx % 4 == 0 && be::aligned16(x), "be_maxpool_nhwc_f32: ldx must be >= c and a multiple of 4");
    BE_REQUIRE(k > 0 && stride > 0 && pad >= 0 && 2 * pad <= k, "be_maxpool_nhwc_f32: bad window");
    const int oh = (h + 2 * pad - k) / stride + 1, ow = (w + 2 * pad - k) / stride + 1;
    BE_REQUIRE(oh > 0 && ow > 0, "be_maxpool_nhwc_f32: empty output");
    const int64_t total = (int64_t)n * oh * ow * (c / 4);
    {
        be::ProfileScope prof(be::as_stream(stream), BE_KERNEL_MAXPOOL, 0.0, 4.0 * n * c * ((double)h * w + (double)oh * ow), 0.0);
        hipLaunchKernelGGL(k_maxpool_nhwc, dim3(grid_cap(total, 256)), dim3(256), 0, be::as_stream(stream), x, y, n, h, w,
                           c / 4, oh, ow, k, stride, pad, ldx / 4);
    }
    return be::check_launch("be_maxpool_nhwc_f32");
}

extern "C" int be_view_to_nhwc4_f32(const be_patch_view* view, int64_t patches_per_image, int64_t first, float* y,
                                    int64_t n, void* stream) {
    BE_REQUIRE(view && view->base && y && n > 0 && first >= 0 && patches_per_image > 0 && view->wp > 0,
               "be_view_to_nhwc4_f32: bad arguments");
    hipLaunchKernelGGL(k_view_to_nhwc4, dim3(grid_cap(n * BE_NPIX, 256)), dim3(256), 0, be::as_stream(stream), *view,
                       patches_per_image, first, y, n);
    return be::check_launch("be_view_to_nhwc4_f32");
}

extern "C" int be_nchw3_to_nhwc4p_f32(const float* x, float* y, int64_t n, int h, int w, int wrow, void* stream) {
    BE_REQUIRE(x && y && n > 0 && h > 0 && w > 0 && wrow >= w + 7, "be_nchw3_to_nhwc4p_f32: bad arguments (wrow >= w + 7)");
    hipLaunchKernelGGL(k_nchw3_to_nhwc4p, dim3(grid_cap(n * h * wrow, 256)), dim3(256), 0, be::as_stream(stream), x, y, n, h, w, wrow);
    return be::check_launch("be_nchw3_to_nhwc4p_f32");
}

extern "C" int be_view_to_nhwc4p_f32(const be_patch_view* view, int64_t patches_per_image, int64_t first, float* y,
                                     int64_t n, int wrow, void* stream) {
    BE_REQUIRE(view && view->base && y && n > 0 && first >= 0 && patches_per_image > 0 && view->wp > 0 && wrow >= BE_R + 7,
               "be_view_to_nhwc4p_f32: bad arguments (wrow >= 28)");
    hipLaunchKernelGGL(k_view_to_nhwc4p, dim3(grid_cap(n * BE_R * wrow, 256)), dim3(256), 0, be::as_stream(stream), *view,
                       patches_per_image, first, y, n, wrow);
    return be::check_launch("be_view_to_nhwc4p_f32");
}

extern "C" int be_conv7x7_nhwc4p_f32(const be_conv_desc* d, const float* x, int wrow, const float* pw, const float* pb, float* y,
                                     int ldy, void* stream) {
    BE_REQUIRE(d && x && pw && y, "be_conv7x7_nhwc4p_f32: null pointer");
    BE_REQUIRE(d->ksize == 7 && d->cin == 4 && d->n >= 1 && wrow >= d->w + 7 && ldy >= d->cout,
               "be_conv7x7_nhwc4p_f32: ksize 7, cin 4, wrow >= w + 7, ldy >= cout required");
    BE_REQUIRE(be::aligned16(x) && be::aligned16(pw) && be::aligned16(y), "be_conv7x7_nhwc4p_f32: 16-byte alignment");
    const int rc = be::conv_pm(d, x, wrow, nullptr, 0, pw, pb, y, ldy, 7 * BK, stream);
    if (rc == 1) return be::fail(BE_EINVAL, "be_conv7x7_nhwc4p_f32: shape not supported (cout_pad must be 64)");
    return rc;
}

extern "C" int be_nchw3_to_nhwc4_f32(const float* x, float* y, int64_t n, int hw, void* stream) {
    BE_REQUIRE(x && y && n > 0 && hw > 0, "be_nchw3_to_nhwc4_f32: bad arguments");
    hipLaunchKernelGGL(k_nchw3_to_nhwc4, dim3(grid_cap(n * hw, 256)), dim3(256), 0, be::as_stream(stream), x, y, n, hw);
    return be::check_launch("be_nchw3_to_nhwc4_f32");
}
