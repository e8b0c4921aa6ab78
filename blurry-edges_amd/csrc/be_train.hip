// Training-mode kernels of LocalStage (models/local_stage.py:30-73 under autograd, local_training.py:103-108):
// batch-statistics BatchNorm forward/backward, Smish backward, max-pool backward, weight gradients on fp32 MFMA,
// the small last Linear.  Data gradients (dgrad) reuse the implicit-GEMM conv kernel with a transposed / mirrored
// weight pack (be_conv.hip).  All reductions are two-stage with a fixed order (no float atomics): results are
// bitwise reproducible run to run.
//
// Activations are NHWC matrices [M = N*H*W rows][C channels]; a BatchNorm channel is a column.
#include "be_common.h"
#include "be_device_math.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------ column reductions
// partial[(split*C + c)*2 + {0,1}] = sum over the split's rows of {a, b}; a/b are produced by a functor.
constexpr int COLS_PER_BLOCK = 64;

struct ColArgs {
    const float* y;         // [M][C]   pre-BN conv output
    const float* dout;      // [M][C]   incoming gradient (backward only)
    const float* s_in;      // [M][C]   Smish input (backward, act=1) or null
    const float* mean;      // [C]
    const float* invstd;    // [C]
    float* ds;              // [M][C]   out: dout * smish'(s_in)  (backward only; may alias dout)
    double* partial;        // [S][C][2]
    int M, C, rows_per_split;
};

__device__ __forceinline__ float smish_grad(float x) {
    // d/dx [x * g(x)], g = tanh(log(1+sigmoid x)) = (u^2-1)/(u^2+1), u = 1+sigma; g' = 4u sigma(1-sigma)/(u^2+1)^2
    const float e = expf(-fabsf(x));
    const float sig = x >= 0.0f ? 1.0f / (1.0f + e) : e / (1.0f + e);
    const float u = 1.0f + sig, u2 = u * u;
    const float g = (u2 - 1.0f) / (u2 + 1.0f);
    const float gp = 4.0f * u * sig * (1.0f - sig) / ((u2 + 1.0f) * (u2 + 1.0f));
    return g + x * gp;
}

template <int MODE>   // 0: forward stats {y, y^2}; 1: backward {ds, ds*xhat} (also writes ds)
__global__ __launch_bounds__(256)
void k_col_reduce(ColArgs a) {
    __shared__ double red[4][COLS_PER_BLOCK][2];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c = blockIdx.x * COLS_PER_BLOCK + tx;
    const int r0 = blockIdx.y * a.rows_per_split;
    const int r1 = min(a.M, r0 + a.rows_per_split);
    double s0 = 0.0, s1 = 0.0;
    if (c < a.C) {
        float mu = 0.f, is = 0.f;
        if (MODE == 1) { mu = a.mean[c]; is = a.invstd[c]; }
        for (int r = r0 + ty; r < r1; r += 4) {
            const size_t at = (size_t)r * a.C + c;
            if (MODE == 0) {
                const float v = a.y[at];
                s0 += v; s1 += (double)v * v;
            } else {
                float d = a.dout[at];
                if (a.s_in) d *= smish_grad(a.s_in[at]);
                a.ds[at] = d;
                s0 += d; s1 += (double)d * ((a.y[at] - mu) * is);
            }
        }
    }
    red[ty][tx][0] = s0; red[ty][tx][1] = s1;
    __syncthreads();
    if (ty == 0 && c < a.C) {
        double t0 = red[0][tx][0] + red[1][tx][0] + red[2][tx][0] + red[3][tx][0];
        double t1 = red[0][tx][1] + red[1][tx][1] + red[2][tx][1] + red[3][tx][1];
        a.partial[((size_t)blockIdx.y * a.C + c) * 2] = t0;
        a.partial[((size_t)blockIdx.y * a.C + c) * 2 + 1] = t1;
    }
}

// forward finalize: mean / invstd (biased variance) + running-stat update (momentum, unbiased variance)
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// one wave per channel: lanes stride over the split partials, fixed-order butterfly (bitwise reproducible)
__global__ void k_bn_finalize_fwd(const double* partial, int S, int C, int M, float eps, float momentum, float* mean,
                                  float* invstd, float* run_mean, float* run_var) {
    const int c = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (c >= C) return;
    double s0 = 0.0, s1 = 0.0;
    for (int s = lane; s < S; s += 64) { s0 += partial[((size_t)s * C + c) * 2]; s1 += partial[((size_t)s * C + c) * 2 + 1]; }
    s0 = wave_sum_f64(s0); s1 = wave_sum_f64(s1);
    if (lane != 0) return;
    const double mu = s0 / M;
    double var = s1 / M - mu * mu;
    if (var < 0.0) var = 0.0;
    mean[c] = (float)mu;
    invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (run_mean) {
        const double unb = M > 1 ? var * M / (M - 1) : var;
        run_mean[c] = (float)((1.0 - momentum) * run_mean[c] + momentum * mu);
        run_var[c] = (float)((1.0 - momentum) * run_var[c] + momentum * unb);
    }
}

// backward finalize: dgamma = sum ds*xhat, dbeta = sum ds
__global__ void k_bn_finalize_bwd(const double* partial, int S, int C, float* dgamma, float* dbeta) {
    const int c = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (c >= C) return;
    double s0 = 0.0, s1 = 0.0;
    for (int s = lane; s < S; s += 64) { s0 += partial[((size_t)s * C + c) * 2]; s1 += partial[((size_t)s * C + c) * 2 + 1]; }
    s0 = wave_sum_f64(s0); s1 = wave_sum_f64(s1);
    if (lane != 0) return;
    dbeta[c] = (float)s0;
    dgamma[c] = (float)s1;
}

// out = act( (y-mean)*invstd*gamma + beta (+ res) ); s_in (optional) keeps the Smish input for the backward
__global__ void k_bn_apply_fwd(const float* __restrict__ y, const float* __restrict__ mean, const float* __restrict__ invstd,
                               const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ res,
                               float* __restrict__ s_in, float* __restrict__ out, int64_t total, int C, int act) {
    const int64_t gs = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gs) {
        const int c = (int)(i % C);
        float z = (y[i] - mean[c]) * invstd[c] * gamma[c] + beta[c];
        if (res) z += res[i];
        if (s_in) s_in[i] = z;
        out[i] = act ? be::smish(z) : z;
    }
}

// dy = gamma*invstd*(ds - dbeta/M - xhat*dgamma/M)
__global__ void k_bn_apply_bwd(const float* __restrict__ ds, const float* __restrict__ y, const float* __restrict__ mean,
                               const float* __restrict__ invstd, const float* __restrict__ gamma,
                               const float* __restrict__ dgamma, const float* __restrict__ dbeta, float* __restrict__ dy,
                               int64_t total, int C, float inv_m) {
    const int64_t gs = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gs) {
        const int c = (int)(i % C);
        const float xh = (y[i] - mean[c]) * invstd[c];
        dy[i] = gamma[c] * invstd[c] * (ds[i] - dbeta[c] * inv_m - xh * dgamma[c] * inv_m);
    }
}

// column sums of a matrix (bias gradients): out[c] = sum_r a[r][c], two-stage
__global__ __launch_bounds__(256)
void k_col_sum(const float* a, double* partial, int M, int C, int rows_per_split) {
    __shared__ double red[4][COLS_PER_BLOCK];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c = blockIdx.x * COLS_PER_BLOCK + tx;
    const int r0 = blockIdx.y * rows_per_split, r1 = min(M, r0 + rows_per_split);
    double s = 0.0;
    if (c < C) for (int r = r0 + ty; r < r1; r += 4) s += a[(size_t)r * C + c];
    red[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && c < C) partial[(size_t)blockIdx.y * C + c] = red[0][tx] + red[1][tx] + red[2][tx] + red[3][tx];
}
__global__ void k_col_sum_final(const double* partial, int S, int C, float* out) {
    const int c = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (c >= C) return;
    double s = 0.0;
    for (int k = lane; k < S; k += 64) s += partial[(size_t)k * C + c];
    s = wave_sum_f64(s);
    if (lane == 0) out[c] = (float)s;
}

// ------------------------------------------------------------------------------------------ max-pool backward
// dx[n,y,x,c] = sum of dout over the output windows whose FIRST maximum (scan order dy,dx) is this pixel
__global__ void k_maxpool_bwd(const float* __restrict__ x, const float* __restrict__ dout, float* __restrict__ dx, int n,
                              int h, int w, int c, int oh, int ow, int k, int stride, int pad) {
    const int64_t total = (int64_t)n * h * w * c;
    const int64_t gs = (int64_t)gridDim.x * blockDim.x;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gs) {
        const int ch = (int)(idx % c);
        int64_t t = idx / c;
        const int xx = (int)(t % w); t /= w;
        const int yy = (int)(t % h);
        const int64_t img = t / h;
        float acc = 0.f;
        // output windows that contain (yy,xx): oy*stride - pad <= yy <= oy*stride - pad + k - 1
        const int oy_lo = max(0, (yy + pad - k + 1 + stride - 1) / stride), oy_hi = min(oh - 1, (yy + pad) / stride);
        const int ox_lo = max(0, (xx + pad - k + 1 + stride - 1) / stride), ox_hi = min(ow - 1, (xx + pad) / stride);
        for (int oy = oy_lo; oy <= oy_hi; ++oy)
            for (int ox = ox_lo; ox <= ox_hi; ++ox) {
                float best = -INFINITY; int by = -1, bx = -1;
                for (int dy = 0; dy < k; ++dy) {
                    const int y2 = oy * stride - pad + dy;
                    if ((unsigned)y2 >= (unsigned)h) continue;
                    for (int dx2 = 0; dx2 < k; ++dx2) {
                        const int x2 = ox * stride - pad + dx2;
                        if ((unsigned)x2 >= (unsigned)w) continue;
                        const float v = x[((img * h + y2) * w + x2) * c + ch];
                        if (v > best) { best = v; by = y2; bx = x2; }
                    }
                }
                if (by == yy && bx == xx) acc += dout[((img * oh + oy) * ow + ox) * c + ch];
            }
        dx[idx] = acc;
    }
}

// ------------------------------------------------------------------------------------------ weight gradient (MFMA)
// dW[co][ci][tap] = sum_m dy[m][co] * x[m + tap][ci]  -- GEMM with K = output pixels.  Block tile 64 co x 64 ci for
// one tap and one M slice; both operands are channel-contiguous, so an LDS image [k = pixel][channel] feeds the
// 32x32x2 MFMA directly (lane = channel, the two k of a step are two LDS rows).
struct WgradArgs {
    const float* x;        // [N,H,W,Cin]
    const float* dy;       // [N,H,W,Cout]
    float* partial;        // [S][Cout*Cin*taps] in the reference's OIHW order
    int M, H, W, HW, Cin, Cout, ks, rows_per_split, chw_hw, cin_tiles;
};

__global__ __launch_bounds__(256)
void k_wgrad(WgradArgs a) {
    constexpr int LD = 68;                               // 64 + 4 floats per LDS row
    __shared__ __attribute__((aligned(16))) float As[2][32][LD];
    __shared__ __attribute__((aligned(16))) float Bs[2][32][LD];
    const int co0 = (blockIdx.x / a.cin_tiles) * 64, ci0 = (blockIdx.x % a.cin_tiles) * 64;
    const int tap = blockIdx.y, half = a.ks >> 1;
    const int tdy = tap / a.ks - half, tdx = tap % a.ks - half;
    const int m_begin = blockIdx.z * a.rows_per_split, m_end = min(a.M, m_begin + a.rows_per_split);
    const int tid = threadIdx.x;
    const int q = tid & 15, r0 = tid >> 4;               // staging: 16 lanes x 16 B = 64 channels of one pixel row
    const int wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const bool a_col_ok = co0 + 4 * q < a.Cout, b_col_ok = ci0 + 4 * q < a.Cin;

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    f32x4 a_st[2], b_st[2];
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};

    auto load = [&](int m0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = m0 + r0 + 16 * i;
            a_st[i] = zero; b_st[i] = zero;
            if (m < m_end) {
                if (a_col_ok) a_st[i] = *reinterpret_cast<const f32x4*>(a.dy + (size_t)m * a.Cout + co0 + 4 * q);
                const int pp = m % a.HW, yy = pp / a.W + tdy, xx = pp % a.W + tdx;
                if (b_col_ok && (unsigned)yy < (unsigned)a.H && (unsigned)xx < (unsigned)a.W) {
                    const float* xr = a.x + ((int64_t)m + tdy * a.W + tdx) * a.Cin;
                    if (a.chw_hw > 0) {
                        // fc.1: the tile's columns run in the REFERENCE's (C,H,W) feature order so that dW is stored in
                        // contiguous runs; our (H,W,C) activations are gathered instead (64 rows: nothing).  Column j =
                        // c*HW + hw of the reference is our column hw*(Cin/HW) + c.
                        const int cpl = a.Cin / a.chw_hw;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int j = ci0 + 4 * q + e;
                            b_st[i][e] = xr[(j % a.chw_hw) * cpl + j / a.chw_hw];
                        }
                    } else {
                        b_st[i] = *reinterpret_cast<const f32x4*>(xr + ci0 + 4 * q);
                    }
                }
            }
        }
    };
    auto store = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            *reinterpret_cast<f32x4*>(&As[buf][r0 + 16 * i][4 * q]) = a_st[i];
            *reinterpret_cast<f32x4*>(&Bs[buf][r0 + 16 * i][4 * q]) = b_st[i];
        }
    };
    const int nchunk = (m_end - m_begin + 31) / 32;
    if (nchunk > 0) { load(m_begin); store(0); }
    __syncthreads();
    for (int kc = 0; kc < nchunk; ++kc) {
        const int buf = kc & 1;
        if (kc + 1 < nchunk) load(m_begin + 32 * (kc + 1));
#pragma unroll
        for (int s = 0; s < 16; ++s)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[buf][2 * s + lh][wm * 32 + li], Bs[buf][2 * s + lh][wn * 32 + li],
                                                       acc, 0, 0, 0);
        if (kc + 1 < nchunk) store(buf ^ 1);
        __syncthreads();
    }
    // D[i = co][j = ci]: lane holds column j = lane&31, rows (r&3) + 8*(r>>2) + 4*(lane>>5)
    const int taps = a.ks * a.ks;
    float* out = a.partial + (size_t)blockIdx.z * a.Cout * a.Cin * taps;
    const int ci = ci0 + wn * 32 + li;
    if (ci < a.Cin) {
        const int ci_ref = ci;       // (fc.1: the columns already run in the reference's order, see the gather above)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (co < a.Cout) out[((size_t)co * a.Cin + ci_ref) * taps + tap] = acc[r];
        }
    }
}

__global__ void k_sum_splits(const float* partial, float* out, int64_t n, int S) {
    const int64_t gs = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gs) {
        float s = 0.f;
        for (int k = 0; k < S; ++k) s += partial[(size_t)k * n + i];
        out[i] = s;
    }
}

// conv1 weight gradient (Cin = 3, 7x7): dW[co][ci][kh][kw] = sum_m dy[m][co] * x4[pixel m + (kh-3,kw-3)][ci].
// 0.5 GFLOP per step at batch 64: a plain FMA kernel, one thread per (co, ci, kh, kw), one grid slice per group of
// whole images so the pixel loops run over the valid window only (no per-row div/mod, no bounds tests).
__global__ void k_wgrad_conv1(const float* __restrict__ x4, const float* __restrict__ dy, float* __restrict__ partial, int N,
                              int H, int W, int Cout, int imgs_per_split) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;          // co fastest: coalesced dy reads, broadcast x4 reads
    const int total = Cout * 3 * 49;
    if (e >= total) return;
    const int co = e % Cout, rest = e / Cout;
    const int ci = rest % 3, t = rest / 3, kh = t / 7, kw = t % 7;
    const int n0 = blockIdx.y * imgs_per_split, n1 = min(N, n0 + imgs_per_split);
    const int y_lo = max(0, 3 - kh), y_hi = min(H, H + 3 - kh);   // output rows whose tap (kh) falls inside the image
    const int x_lo = max(0, 3 - kw), x_hi = min(W, W + 3 - kw);
    // seven independent partial sums per row: the fourteen loads of a step are in flight together (a single running sum
    // made every step wait for its two loads: 182 us for 0.5 GFLOP); fixed order -> bitwise reproducible
    float acc[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int n = n0; n < n1; ++n)
        for (int y = y_lo; y < y_hi; ++y) {
            const float* dyp = dy + ((size_t)(n * H + y) * W) * Cout + co;
            const float* xp = x4 + ((size_t)(n * H + y + kh - 3) * W + (kw - 3)) * 4 + ci;
            int x = x_lo;
            for (; x + 7 <= x_hi; x += 7) {
#pragma unroll
                for (int j = 0; j < 7; ++j) acc[j] = fmaf(dyp[(size_t)(x + j) * Cout], xp[(x + j) * 4], acc[j]);
            }
#pragma unroll
            for (int j = 0; j < 6; ++j)                                  // tail: static register indices
                if (x + j < x_hi) acc[j] = fmaf(dyp[(size_t)(x + j) * Cout], xp[(x + j) * 4], acc[j]);
        }
    partial[(size_t)blockIdx.y * total + ((co * 3 + ci) * 7 + kh) * 7 + kw] =
        ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + acc[6]);
}

// last Linear (1024 -> 10): dx[m][k] = sum_j dy[m][j] W[j][k]; dW[j][k] = sum_m dy[m][j] x[m][k]; db[j] = sum_m dy[m][j]
__global__ void k_linear_small_bwd(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ dy,
                                   float* __restrict__ dx, float* __restrict__ dw, float* __restrict__ db, int M, int K, int J) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < M * K) {                                    // dx
        const int m = idx / K, k = idx % K;
        float s = 0.f;
        for (int j = 0; j < J; ++j) s = fmaf(dy[m * J + j], w[j * K + k], s);
        dx[idx] = s;
    }
    if (idx < J * K) {                                    // dW
        const int j = idx / K, k = idx % K;
        float s = 0.f;
        for (int m = 0; m < M; ++m) s = fmaf(dy[m * J + j], x[m * K + k], s);
        dw[idx] = s;
    }
    if (idx < J) {
        float s = 0.f;
        for (int m = 0; m < M; ++m) s += dy[m * J + idx];
        db[idx] = s;
    }
}

inline unsigned cap_grid(int64_t total, int block, int64_t cap = 4096) {
    int64_t g = (total + block - 1) / block;
    return (unsigned)(g < 1 ? 1 : (g > cap ? cap : g));
}

inline int pick_splits(int M, int col_blocks, int max_splits, int target_blocks = 1024) {
    // enough blocks to cover the chip (~512) without making slices shorter than 32 rows; the finalize kernels walk
    // the splits with one wave per channel (fixed-order butterfly)
    // (narrow matrices - conv1's 64 channels are ONE column block over 28 224 rows - need many short slices: with 128 the
    // reduction was 128 workgroups walking 55 dependent steps each, 55 us for 7 MB)
    if (max_splits > 512) max_splits = 512;
    int s = (target_blocks + col_blocks - 1) / col_blocks;
    const int by_rows = (M + 31) / 32;
    if (s > by_rows) s = by_rows;
    if (s > max_splits) s = max_splits;
    return s < 1 ? 1 : s;
}

}  // namespace

extern "C" size_t be_train_scratch_bytes(void) { return (size_t)64 * 1024 * 1024; }   // split partials (largest: wgrad)

extern "C" int be_bn_train_fwd_f32(const float* y, const float* gamma, const float* beta, const float* res, float eps,
                                   float momentum, float* run_mean, float* run_var, float* mean, float* invstd,
                                   float* s_in, float* out, int M, int C, int act, void* scratch, size_t scratch_bytes,
                                   void* stream) {
    BE_REQUIRE(y && gamma && beta && mean && invstd && out && scratch, "be_bn_train_fwd_f32: null pointer");
    BE_REQUIRE(M > 0 && C > 0, "be_bn_train_fwd_f32: empty");
    const int cb = (C + COLS_PER_BLOCK - 1) / COLS_PER_BLOCK;
    const int S = pick_splits(M, cb, 512);
    BE_REQUIRE((size_t)S * C * 2 * sizeof(double) <= scratch_bytes, "be_bn_train_fwd_f32: scratch too small");
    hipStream_t s = be::as_stream(stream);
    ColArgs a{y, nullptr, nullptr, nullptr, nullptr, nullptr, static_cast<double*>(scratch), M, C, (M + S - 1) / S};
    hipLaunchKernelGGL(k_col_reduce<0>, dim3(cb, S), dim3(256), 0, s, a);
    hipLaunchKernelGGL(k_bn_finalize_fwd, dim3((C + 3) / 4), dim3(256), 0, s, static_cast<const double*>(scratch), S, C, M,
                       eps, momentum, mean, invstd, run_mean, run_var);
    const int64_t total = (int64_t)M * C;
    hipLaunchKernelGGL(k_bn_apply_fwd, dim3(cap_grid(total, 256)), dim3(256), 0, s, y, mean, invstd, gamma, beta, res, s_in,
                       out, total, C, act);
    return be::check_launch("be_bn_train_fwd_f32");
}

extern "C" int be_bn_train_bwd_f32(const float* dout, const float* s_in, const float* y, const float* mean,
                                   const float* invstd, const float* gamma, float* ds, float* dy, float* dgamma,
                                   float* dbeta, int M, int C, void* scratch, size_t scratch_bytes, void* stream) {
    BE_REQUIRE(dout && y && mean && invstd && gamma && ds && dy && dgamma && dbeta && scratch, "be_bn_train_bwd_f32: null pointer");
    BE_REQUIRE(M > 0 && C > 0, "be_bn_train_bwd_f32: empty");
    const int cb = (C + COLS_PER_BLOCK - 1) / COLS_PER_BLOCK;
    const int S = pick_splits(M, cb, 512);
    BE_REQUIRE((size_t)S * C * 2 * sizeof(double) <= scratch_bytes, "be_bn_train_bwd_f32: scratch too small");
    hipStream_t s = be::as_stream(stream);
    ColArgs a{y, dout, s_in, mean, invstd, ds, static_cast<double*>(scratch), M, C, (M + S - 1) / S};
    hipLaunchKernelGGL(k_col_reduce<1>, dim3(cb, S), dim3(256), 0, s, a);
    hipLaunchKernelGGL(k_bn_finalize_bwd, dim3((C + 3) / 4), dim3(256), 0, s, static_cast<const double*>(scratch), S, C,
                       dgamma, dbeta);
    const int64_t total = (int64_t)M * C;
    hipLaunchKernelGGL(k_bn_apply_bwd, dim3(cap_grid(total, 256)), dim3(256), 0, s, ds, y, mean, invstd, gamma, dgamma, dbeta,
                       dy, total, C, 1.0f / (float)M);
    return be::check_launch("be_bn_train_bwd_f32");
}

extern "C" int be_col_sum_f32(const float* a, float* out, int M, int C, void* scratch, size_t scratch_bytes, void* stream) {
    BE_REQUIRE(a && out && scratch && M > 0 && C > 0, "be_col_sum_f32: bad arguments");
    const int cb = (C + COLS_PER_BLOCK - 1) / COLS_PER_BLOCK;
    const int S = pick_splits(M, cb, 512);
    BE_REQUIRE((size_t)S * C * sizeof(double) <= scratch_bytes, "be_col_sum_f32: scratch too small");
    hipStream_t s = be::as_stream(stream);
    hipLaunchKernelGGL(k_col_sum, dim3(cb, S), dim3(256), 0, s, a, static_cast<double*>(scratch), M, C, (M + S - 1) / S);
    hipLaunchKernelGGL(k_col_sum_final, dim3((C + 3) / 4), dim3(256), 0, s, static_cast<const double*>(scratch), S, C, out);
    return be::check_launch("be_col_sum_f32");
}

extern "C" int be_maxpool_nhwc_bwd_f32(const float* x, const float* dout, float* dx, int n, int h, int w, int c, int k,
                                       int stride, int pad, void* stream) {
    BE_REQUIRE(x && dout && dx && n > 0 && h > 0 && w > 0 && c > 0, "be_maxpool_nhwc_bwd_f32: bad arguments");
    const int oh = (h + 2 * pad - k) / stride + 1, ow = (w + 2 * pad - k) / stride + 1;
    const int64_t total = (int64_t)n * h * w * c;
    hipLaunchKernelGGL(k_maxpool_bwd, dim3(cap_grid(total, 256)), dim3(256), 0, be::as_stream(stream), x, dout, dx, n, h, w, c,
                       oh, ow, k, stride, pad);
    return be::check_launch("be_maxpool_nhwc_bwd_f32");
}

extern "C" int be_conv_wgrad_f32(const float* x, const float* dy, float* dw, int n, int h, int w, int cin, int cout,
                                 int ksize, int layout_chw_hw, void* scratch, size_t scratch_bytes, void* stream) {
    BE_REQUIRE(x && dy && dw && scratch, "be_conv_wgrad_f32: null pointer");
    BE_REQUIRE(n > 0 && h > 0 && w > 0 && cin > 0 && cout > 0, "be_conv_wgrad_f32: empty");
    const int M = n * h * w;
    hipStream_t s = be::as_stream(stream);
    if (ksize == 7) {
        BE_REQUIRE(cin == 4, "be_conv_wgrad_f32: ksize 7 takes the NHWC4 input (cin = 4); dW has 3 input channels");
        const int total = cout * 3 * 49;
        const int per = (n + 63) / 64;                       // images per grid slice (<= 64 slices: k_sum_splits walks them serially)
        const int S = (n + per - 1) / per;
        BE_REQUIRE((size_t)S * total * sizeof(float) <= scratch_bytes, "be_conv_wgrad_f32: scratch too small");
        hipLaunchKernelGGL(k_wgrad_conv1, dim3((total + 255) / 256, S), dim3(256), 0, s, x, dy, static_cast<float*>(scratch), n,
                           h, w, cout, per);
        hipLaunchKernelGGL(k_sum_splits, dim3(cap_grid(total, 256)), dim3(256), 0, s, static_cast<const float*>(scratch), dw,
                           (int64_t)total, S);
        return be::check_launch("be_conv_wgrad_f32(conv1)");
    }
    BE_REQUIRE((ksize == 1 || ksize == 3) && cin % 4 == 0 && cout % 4 == 0, "be_conv_wgrad_f32: ksize 1|3, channels %% 4 == 0");
    BE_REQUIRE(layout_chw_hw == 0 || (ksize == 1 && cin % layout_chw_hw == 0), "be_conv_wgrad_f32: bad layout_chw_hw");
    BE_REQUIRE(be::aligned16(x) && be::aligned16(dy), "be_conv_wgrad_f32: x / dy must be 16-byte aligned");
    const int taps = ksize * ksize;
    const int ct = (cout + 63) / 64, it = (cin + 63) / 64;
    const int64_t wsize = (int64_t)cout * cin * taps;
    int S = pick_splits(M, ct * it * taps, 64, 512);       // every split is another full copy of dW to sum
    while (S > 1 && (size_t)S * wsize * sizeof(float) > scratch_bytes) --S;
    BE_REQUIRE((size_t)S * wsize * sizeof(float) <= scratch_bytes, "be_conv_wgrad_f32: scratch too small");
    int rows = (M + S - 1) / S; rows = (rows + 31) / 32 * 32;
    S = (M + rows - 1) / rows;
    WgradArgs a{x, dy, static_cast<float*>(scratch), M, h, w, h * w, cin, cout, ksize, rows, layout_chw_hw, it};
    hipLaunchKernelGGL(k_wgrad, dim3(ct * it, taps, S), dim3(256), 0, s, a);
    hipLaunchKernelGGL(k_sum_splits, dim3(cap_grid(wsize, 256)), dim3(256), 0, s, static_cast<const float*>(scratch), dw, wsize, S);
    return be::check_launch("be_conv_wgrad_f32");
}

extern "C" int be_linear_small_bwd_f32(const float* x, const float* w, const float* dy, float* dx, float* dw, float* db,
                                       int M, int K, int J, void* stream) {
    BE_REQUIRE(x && w && dy && dx && dw && db && M > 0 && K > 0 && J > 0, "be_linear_small_bwd_f32: bad arguments");
    const int total = (M > J ? M : J) * K;
    hipLaunchKernelGGL(k_linear_small_bwd, dim3((total + 255) / 256), dim3(256), 0, be::as_stream(stream), x, w, dy, dx, dw, db,
                       M, K, J);
    return be::check_launch("be_linear_small_bwd_f32");
}
