// Training-mode kernels of LocalStage (models/local_stage.py:30-73 under autograd, local_training.py:103-108):
// batch-statistics BatchNorm forward/backward, Smish backward, max-pool backward, weight gradients on fp32 MFMA,
// the small last Linear.  Data gradients (dgrad) reuse the implicit-GEMM conv kernel with a transposed / mirrored
// weight pack (be_conv.hip).  All reductions are two-stage with a fixed order (no float atomics): results are
// bitwise reproducible run to run.
//
// Activations are NHWC matrices [M = N*H*W rows][C channels]; a BatchNorm channel is a column.
#include <cstdlib>
#include "be_common.h"
#include "be_device_math.h"
#include "be_igemm_body.h"
#include "be_train_sk.h"

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
    int pm;                // 1: pixel-major K walk (3x3, batch a multiple of 32): a chunk = ONE pixel of 32 consecutive images
};

constexpr int WGRAD64_LDS_FLOATS = 2 * 2 * 32 * 68;
// body as a device function: bx / by / bz = blockIdx of the stand-alone kernel; smem: WGRAD64_LDS_FLOATS floats
__device__ __forceinline__ void wgrad64_body(const WgradArgs& a, float* smem, const int bx, const int by, const int bz) {
    constexpr int LD = 68;                               // 64 + 4 floats per LDS row
    float (*As)[32][LD] = reinterpret_cast<float (*)[32][LD]>(smem);
    float (*Bs)[32][LD] = reinterpret_cast<float (*)[32][LD]>(smem + 2 * 32 * LD);
    const int co0 = (bx / a.cin_tiles) * 64, ci0 = (bx % a.cin_tiles) * 64;
    const int tap = by, half = a.ks >> 1;
    const int tdy = tap / a.ks - half, tdx = tap % a.ks - half;
    const int m_begin = bz * a.rows_per_split, m_end = min(a.M, m_begin + a.rows_per_split);
    const int tid = threadIdx.x;
    const int q = tid & 15, r0 = tid >> 4;               // staging: 16 lanes x 16 B = 64 channels of one pixel row
    const int wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const bool a_col_ok = co0 + 4 * q < a.Cout, b_col_ok = ci0 + 4 * q < a.Cin;
    // a wave whose 32 x 32 sub-tile lies outside the matrix (96 channels = a full tile + half a tile) stages its share of the
    // operands but issues no MFMA: the matrix pipe goes to the workgroups that share the CU
    const bool live = co0 + wm * 32 < a.Cout && ci0 + wn * 32 < a.Cin;

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
    if (a.pm) {
        // pixel-major K walk (see wgrad128_body): a chunk = one pixel of 32 consecutive images; chunks whose tap falls outside the
        // image are skipped, the others need no border test and no zero fill
        const int u_begin = bz * (a.rows_per_split / 32), u_end = min(a.M / 32, u_begin + a.rows_per_split / 32);
        const int64_t tap_off = (int64_t)(tdy * a.W + tdx) * a.Cin;
        auto next_valid = [&](int u) {
            for (; u < u_end; ++u) {
                const int pp = u % a.HW, yy = pp / a.W + tdy, xx = pp % a.W + tdx;
                if ((unsigned)yy < (unsigned)a.H && (unsigned)xx < (unsigned)a.W) break;
            }
            return u;
        };
        auto load_u = [&](int u) {
            const int ic = u / a.HW, pp = u - ic * a.HW;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int64_t m = (int64_t)(ic * 32 + r0 + 16 * i) * a.HW + pp;
                a_st[i] = a_col_ok ? *reinterpret_cast<const f32x4*>(a.dy + m * a.Cout + co0 + 4 * q) : zero;
                b_st[i] = b_col_ok ? *reinterpret_cast<const f32x4*>(a.x + m * a.Cin + tap_off + ci0 + 4 * q) : zero;
            }
        };
        int u = next_valid(u_begin);
        if (u < u_end) { load_u(u); store(0); }
        __syncthreads();
        int buf = 0;
        while (u < u_end) {
            const int un = next_valid(u + 1);
            if (un < u_end) load_u(un);
            if (live) {
#pragma unroll
                for (int s = 0; s < 16; ++s)
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[buf][2 * s + lh][wm * 32 + li], Bs[buf][2 * s + lh][wn * 32 + li],
                                                               acc, 0, 0, 0);
            }
            if (un < u_end) store(buf ^ 1);
            __syncthreads();
            buf ^= 1;
            u = un;
        }
    } else {
    const int nchunk = (m_end - m_begin + 31) / 32;
    if (nchunk > 0) { load(m_begin); store(0); }
    __syncthreads();
    for (int kc = 0; kc < nchunk; ++kc) {
        const int buf = kc & 1;
        if (kc + 1 < nchunk) load(m_begin + 32 * (kc + 1));
        if (live) {
#pragma unroll
            for (int s = 0; s < 16; ++s)
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[buf][2 * s + lh][wm * 32 + li], Bs[buf][2 * s + lh][wn * 32 + li],
                                                           acc, 0, 0, 0);
        }
        if (kc + 1 < nchunk) store(buf ^ 1);
        __syncthreads();
    }
    }
    // D[i = co][j = ci]: lane holds column j = lane&31, rows (r&3) + 8*(r>>2) + 4*(lane>>5)
    const int taps = a.ks * a.ks;
    float* out = a.partial + (size_t)bz * a.Cout * a.Cin * taps;
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

__global__ __launch_bounds__(256)
void k_wgrad(WgradArgs a) {
    __shared__ __attribute__((aligned(16))) float smem[WGRAD64_LDS_FLOATS];
    wgrad64_body(a, smem, blockIdx.x, blockIdx.y, blockIdx.z);
}

// Round 3: the weight-gradient GEMM for channel counts that are multiples of 128 (layers 1-3: 75 % of the weight-gradient
// FLOPs).  k_wgrad's 64x64 tile gives a wave ONE 32x32 MFMA tile: two 4-byte LDS reads per MFMA and a tile's operands staged for
// 16 MFMAs per wave (measured 50-62 TFLOP/s).  Here a workgroup owns 128 cout x 128 cin of one tap and one slice of the pixels,
// a wave 64 x 64 as 2 x 2 MFMA tiles whose rows / columns INTERLEAVE (lane li holds channels 2 li and 2 li + 1 of its 64), so one
// ds_read_b64 per operand feeds four MFMAs; K chunks of 16 pixels, register-staged double buffer (33 KB of LDS).  Slices go to
// scratch as [S][tap][cout][cin] - a lane's two accumulators of a column pair are one 8-byte store, 256 contiguous bytes per half
// wave - and k_bwd_post transposes to the reference's [cout][cin][kh][kw] while it sums the slices.
struct Wgrad128Args {
    const float* x; const float* dy; float* partial;
    int M, H, W, HW, Cin, Cout, ks, rows_per_split, cin_tiles;
    int pm;        // 1: pixel-major K walk (3x3, batch a multiple of 16): a chunk = ONE pixel of 16 consecutive images
};

constexpr int WGRAD128_LDS_FLOATS = 2 * 2 * 16 * 132;
__device__ __forceinline__ void wgrad128_body(const Wgrad128Args& a, float* smem, const int bx, const int by, const int bz) {
    constexpr int BKW = 16, LD = 132;                      // 128 + 4 floats per LDS row
    float (*As)[BKW][LD] = reinterpret_cast<float (*)[BKW][LD]>(smem);
    float (*Bs)[BKW][LD] = reinterpret_cast<float (*)[BKW][LD]>(smem + 2 * BKW * LD);
    const int co0 = (bx / a.cin_tiles) * 128, ci0 = (bx % a.cin_tiles) * 128;
    const int tap = by, half = a.ks >> 1;
    const int tdy = tap / a.ks - half, tdx = tap % a.ks - half;
    const int m_begin = bz * a.rows_per_split, m_end = min(a.M, m_begin + a.rows_per_split);
    const int tid = threadIdx.x;
    const int q = tid & 31, r0 = tid >> 5;                 // staging: 32 lanes x 16 B = the 128 channels of one pixel row; 8 rows per pass
    const int wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    typedef float f32x2 __attribute__((ext_vector_type(2)));

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    f32x4 a_st[2], b_st[2];
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    const int64_t tap_off = (int64_t)(tdy * a.W + tdx) * a.Cin;

    auto load = [&](int m0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = m0 + r0 + 8 * i;
            a_st[i] = zero; b_st[i] = zero;
            if (m < m_end) {
                a_st[i] = *reinterpret_cast<const f32x4*>(a.dy + (size_t)m * a.Cout + co0 + 4 * q);
                const int pp = m % a.HW, yy = pp / a.W + tdy, xx = pp % a.W + tdx;
                if ((unsigned)yy < (unsigned)a.H && (unsigned)xx < (unsigned)a.W)
                    b_st[i] = *reinterpret_cast<const f32x4*>(a.x + (int64_t)m * a.Cin + tap_off + ci0 + 4 * q);
            }
        }
    };
    auto store = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            *reinterpret_cast<f32x4*>(&As[buf][r0 + 8 * i][4 * q]) = a_st[i];
            *reinterpret_cast<f32x4*>(&Bs[buf][r0 + 8 * i][4 * q]) = b_st[i];
        }
    };
#define BE_WGRAD128_MFMA(BUF)                                                                                   \
    _Pragma("unroll") for (int s2 = 0; s2 < BKW / 2; ++s2) {                                                    \
        const f32x2 av = *reinterpret_cast<const f32x2*>(&As[BUF][2 * s2 + lh][wm * 64 + 2 * li]);             \
        const f32x2 bv = *reinterpret_cast<const f32x2*>(&Bs[BUF][2 * s2 + lh][wn * 64 + 2 * li]);             \
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[0], bv[0], acc[0][0], 0, 0, 0);                     \
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[0], bv[1], acc[0][1], 0, 0, 0);                     \
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[1], bv[0], acc[1][0], 0, 0, 0);                     \
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[1], bv[1], acc[1][1], 0, 0, 0);                     \
    }
    if (a.pm) {
        // Pixel-major K walk: a chunk = one output pixel of 16 consecutive images, so the tap's shifted pixel is inside the image for
        // the whole chunk or for none of it - chunks outside are SKIPPED (3x3 on 6x6: 21 % of the products of the flat walk were
        // zeros), nothing is zero-filled, and an address is a per-thread constant plus a wave-uniform offset.  Chunk unit u ->
        // (image group u / HW, pixel u % HW): a slice's units visit every pixel, so the slices stay balanced.
        const int u_begin = bz * (a.rows_per_split / BKW), u_end = min(a.M / BKW, u_begin + a.rows_per_split / BKW);
        const unsigned a_vo[2] = {(unsigned)(r0 * a.HW * a.Cout + co0 + 4 * q) * 4u, (unsigned)((r0 + 8) * a.HW * a.Cout + co0 + 4 * q) * 4u};
        const unsigned b_vo[2] = {(unsigned)(r0 * a.HW * a.Cin + ci0 + 4 * q) * 4u, (unsigned)((r0 + 8) * a.HW * a.Cin + ci0 + 4 * q) * 4u};
        auto next_valid = [&](int u) {                    // first unit >= u whose pixel has this tap inside the image (or u_end)
            for (; u < u_end; ++u) {
                const int pp = u % a.HW, yy = pp / a.W + tdy, xx = pp % a.W + tdx;
                if ((unsigned)yy < (unsigned)a.H && (unsigned)xx < (unsigned)a.W) break;
            }
            return u;
        };
        auto load_u = [&](int u) {
            const int ic = u / a.HW, pp = u - ic * a.HW;
            const int64_t m0 = (int64_t)ic * BKW * a.HW + pp;
            const char* pa = reinterpret_cast<const char*>(a.dy + m0 * a.Cout);
            const char* pb = reinterpret_cast<const char*>(a.x + m0 * a.Cin + tap_off);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a_st[i] = *reinterpret_cast<const f32x4*>(pa + a_vo[i]);
                b_st[i] = *reinterpret_cast<const f32x4*>(pb + b_vo[i]);
            }
        };
        int u = next_valid(u_begin);
        if (u < u_end) { load_u(u); store(0); }
        __syncthreads();
        int buf = 0;
        while (u < u_end) {
            const int un = next_valid(u + 1);
            if (un < u_end) load_u(un);
            __builtin_amdgcn_sched_barrier(0);
            BE_WGRAD128_MFMA(buf)
            __builtin_amdgcn_sched_barrier(0);
            if (un < u_end) store(buf ^ 1);
            __syncthreads();
            buf ^= 1;
            u = un;
        }
    } else {
    const int nchunk = (m_end - m_begin + BKW - 1) / BKW;
    if (nchunk > 0) { load(m_begin); store(0); }
    __syncthreads();
    for (int kc = 0; kc < nchunk; ++kc) {
        const int buf = kc & 1;
        if (kc + 1 < nchunk) load(m_begin + BKW * (kc + 1));
        __builtin_amdgcn_sched_barrier(0);
        BE_WGRAD128_MFMA(buf)
        __builtin_amdgcn_sched_barrier(0);
        if (kc + 1 < nchunk) store(buf ^ 1);
        __syncthreads();
    }
    }
#undef BE_WGRAD128_MFMA
    // D tile (i, j): row (e&3) + 8*(e>>2) + 4*lh = position ii in the wave's interleaved rows -> co = co0 + wm*64 + 2*ii + i;
    // column li -> ci = ci0 + wn*64 + 2*li + j: the pair j = 0, 1 is one 8-byte store
    float* out = a.partial + ((size_t)bz * a.ks * a.ks + tap) * a.Cout * a.Cin;
    const int ci = ci0 + wn * 64 + 2 * li;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int co = co0 + wm * 64 + 2 * ((e & 3) + 8 * (e >> 2) + 4 * lh) + i;
            f32x2 v = {acc[i][0][e], acc[i][1][e]};
            *reinterpret_cast<f32x2*>(out + (size_t)co * a.Cin + ci) = v;
        }
}

__global__ __launch_bounds__(256, 2)
void k_wgrad128(Wgrad128Args a) {
    __shared__ __attribute__((aligned(16))) float smem[WGRAD128_LDS_FLOATS];
    wgrad128_body(a, smem, blockIdx.x, blockIdx.y, blockIdx.z);
}

// A unit's weight-gradient GEMM and its data-gradient convolution in ONE launch (VERDICT r2 #1a).  Both read dy, neither fills the
// chip at batch 64 (a 3x3 layer: ~400 weight-gradient workgroups of 20-30 us, ~860 data-gradient workgroups of ~10 us), and a
// hipGraph replays kernels one after the other: as one grid the two sets of workgroups share the CUs.  The grid holds up to TWO
// units' jobs - the 3x3 convolution and the 1x1 downsample of a residual block read the same input and are independent of each
// other, forward and backward: workgroups [0, w_end[0]) take the first weight gradient, [w_end[0], w_end[1]) the second (the long
// ones first), then the convolutions ([.., c_end0) the first, the rest the second; the forward pair is two convolutions and no
// weight gradient).  Every range is padded to a multiple of 8 so that the convolution's XCD-aware tile map - workgroup id mod
// 8 = XCD - is unchanged.  CV: the convolutions' tile variant (be::ConvPrep), the same for both.
struct WJob {
    Wgrad128Args w128;
    WgradArgs w64;
    int wkind;                // 1: k_wgrad128 tiles, 0: k_wgrad tiles
    int real, wx, wy;         // workgroups of the stand-alone launch and its grid x / y
};
struct UnitGemmsArgs {
    be_igemm::ConvArgs ca[2];
    WJob w[2];
    int w_end[2];             // padded ends of the two weight-gradient ranges (w_end[1] = first convolution workgroup)
    int c_end0, c_real[2];    // padded size of the first convolution's range; real workgroups (gx x S) of both
    int cgx[2];               // convolution: workgroups per K slice
};
constexpr int BWD_GEMMS_LDS_FLOATS = WGRAD64_LDS_FLOATS > WGRAD128_LDS_FLOATS ? WGRAD64_LDS_FLOATS : WGRAD128_LDS_FLOATS;   // > the conv's 6400

template <int CV>
__global__ __launch_bounds__(256, 3)
void k_unit_gemms(UnitGemmsArgs g) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int b = blockIdx.x;
    if (b < g.w_end[1]) {
        const int j = b >= g.w_end[0];
        if (j) b -= g.w_end[0];
        const WJob& w = g.w[j];
        if (b >= w.real) return;
        const int bx = b % w.wx, by = (b / w.wx) % w.wy, bz = b / (w.wx * w.wy);
        if (w.wkind == 1) wgrad128_body(w.w128, smem, bx, by, bz);
        else wgrad64_body(w.w64, smem, bx, by, bz);
    } else {
        int c = b - g.w_end[1];
        const int j = c >= g.c_end0;
        if (j) c -= g.c_end0;
        if (c >= g.c_real[j]) return;
        const be_igemm::ConvArgs& ca = g.ca[j];
        const int cgx = g.cgx[j];
        if (CV == 0) be_igemm::conv_igemm_body<2, 2, 1, 1, be_igemm::MODE_TAPS, 16, 0>(ca, smem, c % cgx, c / cgx, 0);
        else if (CV == 1) be_igemm::conv_igemm_body<4, 1, 1, 1, be_igemm::MODE_TAPS, 16, 0>(ca, smem, c % cgx, c / cgx, 0);
        else be_igemm::conv_igemm_body<2, 2, 1, 1, be_igemm::MODE_TAPS, 16, 0, true>(ca, smem, c % cgx, c / cgx, 0);   // uniform tiles
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
                                   float* __restrict__ dx, float* __restrict__ dw, float* __restrict__ db, int M, int K, int J,
                                   int nb_dx, int nb_dw) {
    // three roles side by side (a thread that did dx, then dW, then db in turn made the launch three dependent chains long: 19 us)
    const int b = blockIdx.x;
    if (b < nb_dx) {                                      // dx
        const int idx = b * blockDim.x + threadIdx.x;
        if (idx < M * K) {
            const int m = idx / K, k = idx % K;
            float s = 0.f;
            for (int j0 = 0; j0 < J; j0 += 8) {           // eight loads in flight, accumulated in order
                float a[8], b[8];
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (j0 + u < J) { a[u] = dy[m * J + j0 + u]; b[u] = w[(j0 + u) * K + k]; }
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (j0 + u < J) s = fmaf(a[u], b[u], s);
            }
            dx[idx] = s;
        }
    } else if (b < nb_dx + nb_dw) {                       // dW
        const int idx = (b - nb_dx) * blockDim.x + threadIdx.x;
        if (idx < J * K) {
            const int j = idx / K, k = idx % K;
            float s = 0.f;
            for (int m0 = 0; m0 < M; m0 += 8) {
                float a[8], b[8];
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (m0 + u < M) { a[u] = dy[(m0 + u) * J + j]; b[u] = x[(m0 + u) * K + k]; }
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (m0 + u < M) s = fmaf(a[u], b[u], s);
            }
            dw[idx] = s;
        }
    } else {
        const int idx = threadIdx.x;
        if (idx < J) {
            float s = 0.f;
            for (int m = 0; m < M; ++m) s += dy[m * J + idx];
            db[idx] = s;
        }
    }
}


// =====================================================================================================================
// Round 3: the training UNIT (conv / linear -> BatchNorm with batch statistics -> [+ residual] -> [Smish]) in few launches.
// Round 2 ran a unit's forward as conv, k_splitk_reduce, k_col_reduce<0>, k_bn_finalize_fwd, k_bn_apply_fwd and its backward
// as k_col_reduce<1>, k_bn_finalize_bwd, k_bn_apply_bwd, k_wgrad, k_sum_splits, k_col_sum, k_col_sum_final, conv (dgrad),
// k_splitk_reduce: at batch 64 every one of these is a 4-9 us launch (232 per step).  Now:
//   forward   conv | k_bn_stats (sums the split-K slices + bias -> y AND the per-row-block column sums of y, y^2)
//             | k_bn_fwd_apply (finishes the statistics for its 32 columns in its prologue - <= 32 row-block partials, fixed
//               order, every workgroup of a column block computes the same bits - then normalises, adds the residual, Smish)
//   backward  k_bn_bwd_reduce (ds = dout * smish'(s_in), column sums of ds, ds * xhat) | k_bn_bwd_apply (finishes dgamma /
//             dbeta in its prologue, writes dy and per-workgroup column sums of dy = the bias-gradient partials)
//             | k_wgrad + data-gradient conv | k_bwd_post (ONE launch: weight-gradient slices -> dW, bias partials -> db, the
//             data-gradient split-K slices (+ the other branch's dx) -> dx)
// All reductions keep a fixed order (fp64 partials, no atomics): bitwise reproducible run to run.
// A workgroup = 32 columns x 32 row lanes; a thread moves float4 (4 columns), a wave touches 8 rows x 128 B.
// =====================================================================================================================
constexpr int UC = 32;              // columns per workgroup
constexpr int UR = 32;              // row lanes per workgroup

// fixed-order sum of the 32 row lanes' fp64 partials for `nv` values per column (nv <= 2), via LDS
template <int NV>
__device__ __forceinline__ void block_col_sums(double (&v)[NV][4], double* lds /* [UR][UC][NV] */, double* out, size_t out_stride_c,
                                               int C, int c_base) {
    const int tx = threadIdx.x & 7, ty = threadIdx.x >> 3;
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int k = 0; k < NV; ++k) lds[(ty * UC + tx * 4 + e) * NV + k] = v[k][e];
    __syncthreads();
    if (threadIdx.x < UC * NV) {
        const int col = threadIdx.x / NV, k = threadIdx.x % NV;
        double t = 0.0;
        for (int r = 0; r < UR; ++r) t += lds[(r * UC + col) * NV + k];
        if (c_base + col < C) out[(size_t)(c_base + col) * out_stride_c + k] = t;
    }
}

// fixed-order sum of `nrb` row-block partials [nrb][C][NV] for this workgroup's 32 columns -> tot[col][k] (LDS, doubles)
template <int NV>
__device__ __forceinline__ void finish_partials(const double* __restrict__ partial, int nrb, int C, int c_base, double* part_lds /* [8][UC][NV] */,
                                                double* tot /* [UC][NV] */) {
    const int col = threadIdx.x & 31, pr = threadIdx.x >> 5;        // 8 partial lanes per column
    double a[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) a[k] = 0.0;
    if (c_base + col < C)
        for (int rb0 = pr; rb0 < nrb; rb0 += 64) {      // this lane's partials rb0, rb0 + 8, ...: eight loads in flight (a load per
            double t[8][NV];                             // add left the prologue waiting on 16 L2 round trips), added in that order
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int k = 0; k < NV; ++k)
                    if (rb0 + 8 * j < nrb) t[j][k] = partial[((size_t)(rb0 + 8 * j) * C + c_base + col) * NV + k];
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int k = 0; k < NV; ++k)
                    if (rb0 + 8 * j < nrb) a[k] += t[j][k];
        }
#pragma unroll
    for (int k = 0; k < NV; ++k) part_lds[(pr * UC + col) * NV + k] = a[k];
    __syncthreads();
    if (threadIdx.x < UC * NV) {
        const int c2 = threadIdx.x / NV, k = threadIdx.x % NV;
        double t = 0.0;
        for (int r = 0; r < 8; ++r) t += part_lds[(r * UC + c2) * NV + k];
        tot[c2 * NV + k] = t;
    }
    __syncthreads();
}

// The four BatchNorm kernels of a unit take up to TWO jobs per launch (blockIdx.z picks one): the two units of a residual block
// that are independent of each other - conv1 and the 1x1 downsample in the forward (same input), and again in the backward (both
// gradients are ready once conv2's backward is done) - have the same shape [M, planes] and run side by side (round 3).
template <class A> struct Two { A j[2]; };

struct StatsArgs {
    const float* partial;   // [S][M][ldp] raw split-K slices, or null (S = 0: y is final already)
    const float* bias;      // [C] added to the slice sum (S > 0)
    float* y;               // [M][C]
    double* stats;          // [nrb][C][2]
    int S, M, C, ldp, rows_per_block;
    int sk;                 // round 6: the slices come from the balanced launch (be_train_sk.h): a tile's slice count follows from `g`
    be_sk::ConvGeom g;
};

__global__ __launch_bounds__(256)
void k_bn_stats(Two<StatsArgs> two) {
    const StatsArgs& a = two.j[blockIdx.z];
    __shared__ double lds[UR * UC * 2];
    const int tx = threadIdx.x & 7, ty = threadIdx.x >> 3;
    const int c_base = blockIdx.x * UC, c = c_base + tx * 4;
    const int r0 = blockIdx.y * a.rows_per_block, r1 = min(a.M, r0 + a.rows_per_block);
    double v[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    if (c < a.C) {
        f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
        if (a.S > 0 && a.bias) b4 = *reinterpret_cast<const f32x4*>(a.bias + c);
        if (a.S > 0) {
            for (int r = r0 + ty; r < r1; r += UR) {
                int nS = a.S;
                if (a.sk) {                                  // the slices of this row's tile (pixel of a 64-image group x 64 columns)
                    const int img = r / a.g.HW, pp = r - img * a.g.HW;
                    int ts, n;
                    be_sk::conv_span(a.g, img >> 6, pp, c >> 6, ts, n);
                    nS = be_sk::slices_of(ts, n, a.g.Q);
                }
                // the (<= 8) slices of a row are independent loads: issue them together, then add in slice order
                f32x4 p[8];
#pragma unroll
                for (int s = 0; s < 8; ++s)
                    if (s < nS) p[s] = *reinterpret_cast<const f32x4*>(a.partial + ((size_t)s * a.M + r) * a.ldp + c);
                f32x4 t = p[0];
#pragma unroll
                for (int s = 1; s < 8; ++s) if (s < nS) t += p[s];
                t += b4;
                *reinterpret_cast<f32x4*>(a.y + (size_t)r * a.C + c) = t;
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[0][e] += t[e]; v[1][e] += (double)t[e] * t[e]; }
            }
        } else {
            // four rows in flight (conv1: 7 rows per thread, each an HBM / L2 round trip when loaded one by one), summed in row order
            for (int rb = r0 + ty; rb < r1; rb += 4 * UR) {
                f32x4 t[4];
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (rb + j * UR < r1) t[j] = *reinterpret_cast<const f32x4*>(a.y + (size_t)(rb + j * UR) * a.C + c);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (rb + j * UR < r1) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) { v[0][e] += t[j][e]; v[1][e] += (double)t[j][e] * t[j][e]; }
                    }
            }
        }
    }
    block_col_sums<2>(v, lds, a.stats + (size_t)blockIdx.y * a.C * 2, 2, a.C, c_base);
}

struct FwdApplyArgs {
    const float* y; const double* stats; const float* gamma; const float* beta; const float* res;
    float* run_mean; float* run_var; float* mean; float* invstd; float* s_in; float* out;
    int nrb, M, C, rows_per_block, act;
    float eps, momentum;
};

__global__ __launch_bounds__(256)
void k_bn_fwd_apply(Two<FwdApplyArgs> two) {
    const FwdApplyArgs a = two.j[blockIdx.z];
    __shared__ double part[8 * UC * 2];
    __shared__ double tot[UC * 2];
    __shared__ float s_mu[UC], s_is[UC];
    const int c_base = blockIdx.x * UC;
    finish_partials<2>(a.stats, a.nrb, a.C, c_base, part, tot);
    if (threadIdx.x < UC && c_base + (int)threadIdx.x < a.C) {
        const int col = threadIdx.x, c = c_base + col;
        const double mu = tot[col * 2] / a.M;
        double var = tot[col * 2 + 1] / a.M - mu * mu;
        if (var < 0.0) var = 0.0;
        const float mf = (float)mu, isf = (float)(1.0 / sqrt(var + (double)a.eps));
        s_mu[col] = mf; s_is[col] = isf;
        if (blockIdx.y == 0) {                                  // one workgroup per column block publishes the statistics
            a.mean[c] = mf; a.invstd[c] = isf;
            if (a.run_mean) {
                const double unb = a.M > 1 ? var * a.M / (a.M - 1) : var;
                a.run_mean[c] = (float)((1.0 - a.momentum) * a.run_mean[c] + a.momentum * mu);
                a.run_var[c] = (float)((1.0 - a.momentum) * a.run_var[c] + a.momentum * unb);
            }
        }
    }
    __syncthreads();
    const int tx = threadIdx.x & 7, ty = threadIdx.x >> 3, c = c_base + tx * 4;
    if (c >= a.C) return;
    const f32x4 g = *reinterpret_cast<const f32x4*>(a.gamma + c), b = *reinterpret_cast<const f32x4*>(a.beta + c);
    f32x4 mu, is;
#pragma unroll
    for (int e = 0; e < 4; ++e) { mu[e] = s_mu[tx * 4 + e]; is[e] = s_is[tx * 4 + e]; }
    const int r0 = blockIdx.y * a.rows_per_block, r1 = min(a.M, r0 + a.rows_per_block);
    for (int rb = r0 + ty; rb < r1; rb += 4 * UR) {         // four rows in flight
        f32x4 yv[4], rv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (rb + j * UR < r1) {
                const size_t at = (size_t)(rb + j * UR) * a.C + c;
                yv[j] = *reinterpret_cast<const f32x4*>(a.y + at);
                if (a.res) rv[j] = *reinterpret_cast<const f32x4*>(a.res + at);
            }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (rb + j * UR < r1) {
                const size_t at = (size_t)(rb + j * UR) * a.C + c;
                f32x4 z;
#pragma unroll
                for (int e = 0; e < 4; ++e) z[e] = (yv[j][e] - mu[e]) * is[e] * g[e] + b[e];
                if (a.res) z += rv[j];
                if (a.s_in) *reinterpret_cast<f32x4*>(a.s_in + at) = z;
                if (a.act) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) z[e] = be::smish(z[e]);
                }
                *reinterpret_cast<f32x4*>(a.out + at) = z;
            }
    }
}

struct BwdReduceArgs {
    const float* dout; const float* s_in; const float* y; const float* mean; const float* invstd;
    float* ds; double* partial;   // [nrb][C][2]: sum ds, sum ds * xhat
    int M, C, rows_per_block;
};

__global__ __launch_bounds__(256)
void k_bn_bwd_reduce(Two<BwdReduceArgs> two) {
    const BwdReduceArgs a = two.j[blockIdx.z];
    __shared__ double lds[UR * UC * 2];
    const int tx = threadIdx.x & 7, ty = threadIdx.x >> 3;
    const int c_base = blockIdx.x * UC, c = c_base + tx * 4;
    const int r0 = blockIdx.y * a.rows_per_block, r1 = min(a.M, r0 + a.rows_per_block);
    double v[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    if (c < a.C) {
        const f32x4 mu = *reinterpret_cast<const f32x4*>(a.mean + c), is = *reinterpret_cast<const f32x4*>(a.invstd + c);
        for (int rb = r0 + ty; rb < r1; rb += 4 * UR) {         // four rows in flight, accumulated in row order
            f32x4 dv[4], sv[4], yv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (rb + j * UR < r1) {
                    const size_t at = (size_t)(rb + j * UR) * a.C + c;
                    dv[j] = *reinterpret_cast<const f32x4*>(a.dout + at);
                    if (a.s_in) sv[j] = *reinterpret_cast<const f32x4*>(a.s_in + at);
                    yv[j] = *reinterpret_cast<const f32x4*>(a.y + at);
                }
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (rb + j * UR < r1) {
                    const size_t at = (size_t)(rb + j * UR) * a.C + c;
                    f32x4 d = dv[j];
                    if (a.s_in) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) d[e] *= smish_grad(sv[j][e]);
                    }
                    *reinterpret_cast<f32x4*>(a.ds + at) = d;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v[0][e] += d[e]; v[1][e] += (double)d[e] * ((yv[j][e] - mu[e]) * is[e]); }
                }
        }
    }
    block_col_sums<2>(v, lds, a.partial + (size_t)blockIdx.y * a.C * 2, 2, a.C, c_base);
}

struct BwdApplyArgs {
    const float* ds; const float* y; const float* mean; const float* invstd; const float* gamma; const double* partial;
    float* dy; float* dgamma; float* dbeta;
    double* dbpart;           // [gridDim.y][C]: column sums of dy per workgroup (bias-gradient partials)
    int nrb, M, C, rows_per_block;
    float inv_m;
};

__global__ __launch_bounds__(256)
void k_bn_bwd_apply(Two<BwdApplyArgs> two) {
    const BwdApplyArgs a = two.j[blockIdx.z];
    __shared__ double part[8 * UC * 2];
    __shared__ double tot[UC * 2];
    __shared__ double lds[UR * UC];
    __shared__ float s_db[UC], s_dg[UC];
    const int c_base = blockIdx.x * UC;
    finish_partials<2>(a.partial, a.nrb, a.C, c_base, part, tot);
    if (threadIdx.x < UC && c_base + (int)threadIdx.x < a.C) {
        const int col = threadIdx.x;
        const float db = (float)tot[col * 2], dg = (float)tot[col * 2 + 1];
        s_db[col] = db; s_dg[col] = dg;
        if (blockIdx.y == 0) { a.dbeta[c_base + col] = db; a.dgamma[c_base + col] = dg; }
    }
    __syncthreads();
    const int tx = threadIdx.x & 7, ty = threadIdx.x >> 3, c = c_base + tx * 4;
    double v[1][4] = {{0, 0, 0, 0}};
    if (c < a.C) {
        const f32x4 g = *reinterpret_cast<const f32x4*>(a.gamma + c);
        const f32x4 mu = *reinterpret_cast<const f32x4*>(a.mean + c), is = *reinterpret_cast<const f32x4*>(a.invstd + c);
        f32x4 db, dg;
#pragma unroll
        for (int e = 0; e < 4; ++e) { db[e] = s_db[tx * 4 + e]; dg[e] = s_dg[tx * 4 + e]; }
        const int r0 = blockIdx.y * a.rows_per_block, r1 = min(a.M, r0 + a.rows_per_block);
        for (int rb = r0 + ty; rb < r1; rb += 4 * UR) {         // four rows in flight, accumulated in row order
            f32x4 dv[4], yv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (rb + j * UR < r1) {
                    const size_t at = (size_t)(rb + j * UR) * a.C + c;
                    dv[j] = *reinterpret_cast<const f32x4*>(a.ds + at);
                    yv[j] = *reinterpret_cast<const f32x4*>(a.y + at);
                }
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (rb + j * UR < r1) {
                    f32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float xh = (yv[j][e] - mu[e]) * is[e];
                        o[e] = g[e] * is[e] * (dv[j][e] - db[e] * a.inv_m - xh * dg[e] * a.inv_m);
                        v[0][e] += o[e];
                    }
                    *reinterpret_cast<f32x4*>(a.dy + (size_t)(rb + j * UR) * a.C + c) = o;
                }
        }
    }
    block_col_sums<1>(v, lds, a.dbpart + (size_t)blockIdx.y * a.C, 1, a.C, c_base);
}

// ONE launch after a unit's weight-gradient GEMM and data-gradient convolution: workgroups [0, nb_w) sum the weight-gradient
// slices into dW, [nb_w, nb_w + nb_b) finish the bias gradient, the rest sum the data-gradient split-K slices (+ the other
// branch's dx of a residual block) into dx.  Fixed order everywhere.
struct PostJob {              // one unit's parameter-gradient roles
    const float* wpart; float* dw; int64_t wsize; int wS; int conv1_map, cout1;   // conv1_map: slices are [cout][7][8 px][4 ch]
    int wtaps;                // > 0: slices are [S][tap][cout][cin] (k_wgrad128), transposed here
    int chw;                  // > 0 (wtaps == 1, balanced launch): the slices' columns run (hw, c) - our activations' order - and dW is
                              //      stored in the reference's (c, hw) column order, hw < chw (fc.1: models/local_stage.py:44-46)
    const double* dbpart; float* db; int nb_rows, C;
    int nb_w, nb_b;
};
struct PostArgs {             // up to two units (a residual block's conv1 and downsample: their input gradients are summed here)
    PostJob j[2];
    int nj;
    const float* xpart; const float* xadd; float* dx; int64_t xM; int xC, xldp, xS;
    const float* xpart2; int xldp2, xS2;         // the second unit's data-gradient slices (xS2 = 0: none)
    // round 6 (be_train_sk.h): the slices come from the balanced launch - how many a tile has follows from its place on the axis
    int sk;
    be_sk::WGeom wsk[2];
    be_sk::ConvGeom xsk[2];
};

__device__ __forceinline__ void post_job(const PostJob& a, const int b, const be_sk::WGeom& wgeo, const bool sk) {
    if (b < a.nb_w) {
        if (a.conv1_map) {                                  // dW[co][ci][kh][kw] <- slice[co][kh][kw][ci] of a 224-column row
            const int64_t total = (int64_t)a.cout1 * 147;
            for (int64_t i = (int64_t)b * 256 + threadIdx.x; i < total; i += (int64_t)a.nb_w * 256) {
                const int co = (int)(i / 147), rr = (int)(i % 147), ci = rr / 49, kh = (rr % 49) / 7, kw = rr % 7;
                const int64_t src = (int64_t)co * 224 + kh * 32 + kw * 4 + ci;
                float s = 0.f;
                for (int k0 = 0; k0 < a.wS; k0 += 8) {      // eight independent loads in flight, added in slice order
                    float p[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) p[k] = k0 + k < a.wS ? a.wpart[(int64_t)(k0 + k) * a.cout1 * 224 + src] : 0.f;
#pragma unroll
                    for (int k = 0; k < 8; ++k) if (k0 + k < a.wS) s += p[k];
                }
                a.dw[i] = s;
            }
            return;
        }
        if (a.wtaps == 1 && a.wS >= 32) {
            // MANY slices of a small matrix (GlobalStage's linears over 8 x 4096 tokens: up to 128 slices of 49 152 - 98 304 floats).
            // One thread per element walked all of them in 16 dependent rounds of 8 loads (101 us per call, 32 calls per step = 12 %
            // of the global training step).  Here a workgroup takes 64 positions x 4 slice groups: group g sums the slices g, g + 4,
            // g + 8, ... (8 loads in flight, added in that order), the four partial sums meet in LDS in a fixed order.
            __shared__ float red4[4][64];
            const int64_t plane = a.wsize;
            const int pos = threadIdx.x & 63, grp = threadIdx.x >> 6;
            for (int64_t i0 = (int64_t)b * 64; i0 < plane; i0 += (int64_t)a.nb_w * 64) {
                const int64_t i = i0 + pos;
                float s = 0.f;
                if (i < plane)
                    for (int k0 = grp; k0 < a.wS; k0 += 32) {
                        float p[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j) p[j] = k0 + 4 * j < a.wS ? a.wpart[(int64_t)(k0 + 4 * j) * plane + i] : 0.f;
#pragma unroll
                        for (int j = 0; j < 8; ++j) if (k0 + 4 * j < a.wS) s += p[j];
                    }
                red4[grp][pos] = s;
                __syncthreads();
                if (grp == 0 && i < plane) a.dw[i] = (red4[0][pos] + red4[1][pos]) + (red4[2][pos] + red4[3][pos]);
                __syncthreads();
            }
            return;
        }
        if (a.chw > 0 && sk) {
            // fc.1 on the balanced launch (one tap): the slices' columns run (hw, c) - our activations' order - and dW is stored in the
            // reference's (c, hw) column order (models/local_stage.py:44-46: Flatten of [C,H,W]).  A branch of its own: with the column
            // map inside the general loop below the compiler spilled that loop (every k_bwd_post of the step 45 us slower)
            const int cin = wgeo.cin, cpl = cin / a.chw;
            const int64_t plane = a.wsize;
            for (int64_t i = (int64_t)b * 256 + threadIdx.x; i < plane; i += (int64_t)a.nb_w * 256) {
                const int co = (int)(i / cin), ci = (int)(i - (int64_t)co * cin);
                int ts, n;
                be_sk::w_span(wgeo, 0, (co / wgeo.tm) * wgeo.cin_tiles + ci / wgeo.tn, ts, n);
                const int nS = be_sk::slices_of(ts, n, wgeo.Q);
                float s = 0.f;
                for (int k = 0; k < nS; ++k) s += a.wpart[(int64_t)k * plane + i];
                a.dw[(int64_t)co * cin + (ci % cpl) * a.chw + ci / cpl] = s;
            }
            return;
        }
        if (a.wtaps) {                                      // k_wgrad128's slices [S][tap][cout*cin] -> dW[cout][cin][tap]
            // a workgroup sums 256 consecutive (cout, cin) positions for every tap (coalesced reads of each slice plane), parks
            // the 256 x taps results in LDS and writes them out as ONE contiguous run (the transposed stores straight from
            // registers were 4-byte pieces 36 bytes apart: 20-30 us per call)
            __shared__ float tbuf[256 * 9];
            const int64_t plane = a.wsize / a.wtaps;        // cout * cin
            for (int64_t i0 = (int64_t)b * 256; i0 < plane; i0 += (int64_t)a.nb_w * 256) {
                const int64_t i = i0 + threadIdx.x;
                if (i < plane) {
                    // balanced launch: the slices of THIS element's 128 x 128 tile, tap by tap (be_train_sk.h)
                    int tile_j = 0;
                    if (sk) { const int co = (int)(i / wgeo.cin), ci = (int)(i - (int64_t)co * wgeo.cin); tile_j = (co / wgeo.tm) * wgeo.cin_tiles + ci / wgeo.tn; }
                    for (int t = 0; t < a.wtaps; ++t) {
                        int nS = a.wS;
                        if (sk) { int ts, n; be_sk::w_span(wgeo, t, tile_j, ts, n); nS = be_sk::slices_of(ts, n, wgeo.Q); }
                        float s = 0.f;
                        for (int k0 = 0; k0 < nS; k0 += 8) {            // eight slices in flight, added in slice order
                            float p[8];
#pragma unroll
                            for (int k = 0; k < 8; ++k) p[k] = k0 + k < nS ? a.wpart[((int64_t)(k0 + k) * a.wtaps + t) * plane + i] : 0.f;
#pragma unroll
                            for (int k = 0; k < 8; ++k) if (k0 + k < nS) s += p[k];
                        }
                        tbuf[threadIdx.x * a.wtaps + t] = s;
                    }
                }
                __syncthreads();
                const int64_t n_here = (plane - i0 < 256 ? plane - i0 : 256) * a.wtaps;
                for (int64_t j = threadIdx.x; j < n_here; j += 256) a.dw[i0 * a.wtaps + j] = tbuf[j];
                __syncthreads();
            }
            return;
        }
        const int64_t n4 = a.wsize >> 2;                    // weight tensors of the convs / linears: multiples of 4 floats
        for (int64_t i = (int64_t)b * 256 + threadIdx.x; i < n4; i += (int64_t)a.nb_w * 256) {
            f32x4 s = reinterpret_cast<const f32x4*>(a.wpart)[i];
            for (int k0 = 1; k0 < a.wS; k0 += 8) {
                f32x4 p[8];
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (k0 + k < a.wS) p[k] = reinterpret_cast<const f32x4*>(a.wpart + (int64_t)(k0 + k) * a.wsize)[i];
#pragma unroll
                for (int k = 0; k < 8; ++k) if (k0 + k < a.wS) s += p[k];
            }
            reinterpret_cast<f32x4*>(a.dw)[i] = s;
        }
    } else if (b < a.nb_w + a.nb_b) {
        // bias gradient: 32 columns per workgroup, 8 lanes per column walk the row-block partials (a single lane walking up
        // to 128 dependent loads was 20-30 us), combined in lane order
        __shared__ double red[8 * 32];
        const int col = threadIdx.x & 31, pr = threadIdx.x >> 5, c = (b - a.nb_w) * 32 + col;
        double s = 0.0;
        if (c < a.C)
            for (int r0 = pr; r0 < a.nb_rows; r0 += 64) {   // eight loads in flight, added in the same order
                double t[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) if (r0 + 8 * j < a.nb_rows) t[j] = a.dbpart[(size_t)(r0 + 8 * j) * a.C + c];
#pragma unroll
                for (int j = 0; j < 8; ++j) if (r0 + 8 * j < a.nb_rows) s += t[j];
            }
        red[pr * 32 + col] = s;
        __syncthreads();
        if (threadIdx.x < 32 && c < a.C) {
            double t = 0.0;
            for (int r = 0; r < 8; ++r) t += red[r * 32 + col];
            a.db[c] = (float)t;
        }
    }
}

__global__ __launch_bounds__(256)
void k_bwd_post(PostArgs g) {
    int b = blockIdx.x;
    const int n_j0 = g.j[0].nb_w + g.j[0].nb_b, n_j1 = g.nj > 1 ? g.j[1].nb_w + g.j[1].nb_b : 0;
    if (b < n_j0) {
        post_job(g.j[0], b, g.wsk[0], g.sk != 0);
    } else if (b < n_j0 + n_j1) {
        post_job(g.j[1], b - n_j0, g.wsk[1], g.sk != 0);
    } else {
        const PostArgs& a = g;
        const int bx = b - n_j0 - n_j1, nb_x = gridDim.x - n_j0 - n_j1;
        const int c4n = a.xC >> 2;
        const int64_t total = a.xM * c4n;
        for (int64_t i = (int64_t)bx * 256 + threadIdx.x; i < total; i += (int64_t)nb_x * 256) {
            const int64_t m = i / c4n;
            const int c = (int)(i - m * c4n) * 4;
            int xS = a.xS, xS2 = a.xS2;
            if (a.sk) {                                         // balanced launch: the slices of this element's tile (be_train_sk.h)
                const int HW = a.xsk[0].HW, img = (int)(m / HW), pp = (int)(m - (int64_t)img * HW);
                int ts, n;
                be_sk::conv_span(a.xsk[0], img >> 6, pp, c >> 6, ts, n);
                xS = be_sk::slices_of(ts, n, a.xsk[0].Q);
                if (a.xS2) { be_sk::conv_span(a.xsk[1], img >> 6, pp, c >> 6, ts, n); xS2 = be_sk::slices_of(ts, n, a.xsk[1].Q); }
            }
            f32x4 p[8];
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (k < xS) p[k] = *reinterpret_cast<const f32x4*>(a.xpart + ((int64_t)k * a.xM + m) * a.xldp + c);
            f32x4 s = p[0];
#pragma unroll
            for (int k = 1; k < 8; ++k) if (k < xS) s += p[k];
            if (xS2) {                                          // the second unit's slices: summed on their own in slice order, then
                f32x4 s2 = {0.f, 0.f, 0.f, 0.f};                // added - the bits of two single-unit calls (dx_add = the second's dx)
                for (int k0 = 0; k0 < xS2; k0 += 4) {           // four loads in flight
                    f32x4 q[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (k0 + k < xS2) q[k] = *reinterpret_cast<const f32x4*>(a.xpart2 + ((int64_t)(k0 + k) * a.xM + m) * a.xldp2 + c);
#pragma unroll
                    for (int k = 0; k < 4; ++k) if (k0 + k < xS2) { if (k0 + k == 0) s2 = q[k]; else s2 += q[k]; }
                }
                s += s2;
            }
            if (a.xadd) s += *reinterpret_cast<const f32x4*>(a.xadd + m * a.xC + c);
            *reinterpret_cast<f32x4*>(a.dx + m * a.xC + c) = s;
        }
    }
}

// conv1's weight gradient on the matrix pipe (round 2: a scalar FMA kernel, 89 + 16 us for 0.53 GFLOP).  Same GEMM as k_wgrad
// with the 7x7x3 taps laid out as conv1's forward lays them out: column j = kh * 32 + px * 4 + ch of a 224-column row (px = kw,
// the eighth pixel and the fourth channel are padding), x4 the NHWC4 staging.  A tile = 64 cout x 2 kernel rows; K = pixels.
struct WgradC1Args { const float* x4; const float* dy; float* partial; int M, H, W, HW, Cout, rows_per_split; };

__global__ __launch_bounds__(256)
void k_wgrad_conv1_mfma(WgradC1Args a) {
    constexpr int LD = 68;
    __shared__ __attribute__((aligned(16))) float As[2][32][LD];
    __shared__ __attribute__((aligned(16))) float Bs[2][32][LD];
    const int kh0 = blockIdx.y * 2;
    const int m_begin = blockIdx.z * a.rows_per_split, m_end = min(a.M, m_begin + a.rows_per_split);
    const int tid = threadIdx.x, q = tid & 15, r0 = tid >> 4;
    const int wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int kh = kh0 + (q >> 3), tdy = kh - 3, tdx = (q & 7) - 3;     // this thread's staged pixel of the B tile
    const bool b_ok = kh < 7 && (q & 7) < 7;
    const bool a_ok = 4 * q < a.Cout;
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
                if (a_ok) a_st[i] = *reinterpret_cast<const f32x4*>(a.dy + (size_t)m * a.Cout + 4 * q);
                const int pp = m % a.HW, yy = pp / a.W + tdy, xx = pp % a.W + tdx;
                if (b_ok && (unsigned)yy < (unsigned)a.H && (unsigned)xx < (unsigned)a.W)
                    b_st[i] = *reinterpret_cast<const f32x4*>(a.x4 + ((int64_t)m + tdy * a.W + tdx) * 4);
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
    float* out = a.partial + (size_t)blockIdx.z * a.Cout * 224;
    const int col = kh0 * 32 + wn * 32 + li;                              // column of the 224-wide row
    if (col < 224) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (co < a.Cout) out[(size_t)co * 224 + col] = acc[r];
        }
    }
}

// last Linear forward (1024 -> 10) of the training step: one wave per output element (round 2 ran it as a 128x32-tile conv:
// 8 workgroups walking 64 K chunks each, 45 us)
__global__ __launch_bounds__(256)
void k_linear_small_fwd(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b, float* __restrict__ y,
                        int M, int K, int J) {
    const int o = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (o >= M * J) return;
    const int m = o / J, j = o % J;
    float s = 0.f;
    for (int k = lane * 4; k < K; k += 256) {
        const f32x4 xv = *reinterpret_cast<const f32x4*>(x + (size_t)m * K + k), wv = *reinterpret_cast<const f32x4*>(w + (size_t)j * K + k);
        s = fmaf(xv[0], wv[0], s); s = fmaf(xv[1], wv[1], s); s = fmaf(xv[2], wv[2], s); s = fmaf(xv[3], wv[3], s);
    }
    s = be::wave_sum(s);
    if (lane == 0) y[o] = s + b[j];
}

// max-pool forward that also records WHICH element of the window won (first maximum in scan order, PyTorch's rule), one byte per
// output value, and the backward that reads those bytes: an input pixel receives the gradient of the (at most four) windows whose
// recorded winner it is.  Round 2's backward re-scanned every window that contains the pixel (36 loads per value).
__global__ void k_maxpool_fwd_idx(const float* __restrict__ x, float* __restrict__ y, unsigned char* __restrict__ idx, int n, int h,
                                  int w, int c4, int oh, int ow, int k, int stride, int pad) {
    const int64_t total = (int64_t)n * oh * ow * c4;
    const int64_t gs = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gs) {
        const int cq = (int)(i % c4);
        int64_t t = i / c4;
        const int ox = (int)(t % ow); t /= ow;
        const int oy = (int)(t % oh);
        const int64_t img = t / oh;
        f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        unsigned wi[4] = {0, 0, 0, 0};
        if (k <= 3) {
            // all taps as independent loads (a tap outside the image loads the nearest pixel inside and is not compared), then the
            // scan in the same order: the first maximum wins, as before
            f32x4 v[9];
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx)
                    if (dy < k && dx < k) {
                        const int yy = min(max(oy * stride - pad + dy, 0), h - 1), xx = min(max(ox * stride - pad + dx, 0), w - 1);
                        v[3 * dy + dx] = reinterpret_cast<const f32x4*>(x)[((img * h + yy) * w + xx) * c4 + cq];
                    }
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx)
                    if (dy < k && dx < k && (unsigned)(oy * stride - pad + dy) < (unsigned)h && (unsigned)(ox * stride - pad + dx) < (unsigned)w) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) if (v[3 * dy + dx][e] > m[e]) { m[e] = v[3 * dy + dx][e]; wi[e] = (unsigned)(dy * k + dx); }
                    }
        } else
        for (int dy = 0; dy < k; ++dy) {
            const int yy = oy * stride - pad + dy;
            if ((unsigned)yy >= (unsigned)h) continue;
            for (int dx = 0; dx < k; ++dx) {
                const int xx = ox * stride - pad + dx;
                if ((unsigned)xx >= (unsigned)w) continue;
                const f32x4 v = reinterpret_cast<const f32x4*>(x)[((img * h + yy) * w + xx) * c4 + cq];
#pragma unroll
                for (int e = 0; e < 4; ++e) if (v[e] > m[e]) { m[e] = v[e]; wi[e] = (unsigned)(dy * k + dx); }
            }
        }
        reinterpret_cast<f32x4*>(y)[i] = m;
        reinterpret_cast<unsigned*>(idx)[i] = wi[0] | (wi[1] << 8) | (wi[2] << 16) | (wi[3] << 24);
    }
}

__global__ void k_maxpool_bwd_idx(const unsigned char* __restrict__ idx, const float* __restrict__ dout, float* __restrict__ dx, int n,
                                  int h, int w, int c4, int oh, int ow, int k, int stride, int pad) {
    const int64_t total = (int64_t)n * h * w * c4;
    const int64_t gs = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gs) {
        const int cq = (int)(i % c4);
        int64_t t = i / c4;
        const int xx = (int)(t % w); t /= w;
        const int yy = (int)(t % h);
        const int64_t img = t / h;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const int oy_lo = max(0, (yy + pad - k + 1 + stride - 1) / stride), oy_hi = min(oh - 1, (yy + pad) / stride);
        const int ox_lo = max(0, (xx + pad - k + 1 + stride - 1) / stride), ox_hi = min(ow - 1, (xx + pad) / stride);
        for (int oy = oy_lo; oy <= oy_hi; ++oy)
            for (int ox = ox_lo; ox <= ox_hi; ++ox) {
                const int64_t o = ((img * oh + oy) * ow + ox) * c4 + cq;
                const unsigned me = (unsigned)((yy - (oy * stride - pad)) * k + (xx - (ox * stride - pad)));
                const unsigned wv = reinterpret_cast<const unsigned*>(idx)[o];
                const f32x4 d = reinterpret_cast<const f32x4*>(dout)[o];
#pragma unroll
                for (int e = 0; e < 4; ++e) if (((wv >> (8 * e)) & 255u) == me) acc[e] += d[e];
            }
        reinterpret_cast<f32x4*>(dx)[i] = acc;
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

// scratch of the training units: [0, 1 MB) BatchNorm row-block partials, [1, 2 MB) bias-gradient partials, [2, 42 MB) split-K
// slices of the forward / data-gradient convolution, [42, 112 MB) weight-gradient slices (the older single-purpose entry
// points use the buffer from its start)
constexpr size_t SCR_STATS = 0, SCR_DBPART = (size_t)1 << 20, SCR_CONV = (size_t)2 << 20, SCR_WGRAD = (size_t)42 << 20,
                 SCR_TOTAL = (size_t)112 << 20;
extern "C" size_t be_train_scratch_bytes(void) { return SCR_TOTAL; }

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

// launches the weight-gradient GEMM; the S slices [S][wsize] land in `partial` (S == 1: `partial` may be dw itself)
static int wgrad_slices(const float* x, const float* dy, float* partial, size_t partial_bytes, int n, int h, int w, int cin, int cout,
                        int ksize, int layout_chw_hw, hipStream_t s, int* S_out, int64_t* wsize_out) {
    const int M = n * h * w;
    BE_REQUIRE((ksize == 1 || ksize == 3) && cin % 4 == 0 && cout % 4 == 0, "be_conv_wgrad_f32: ksize 1|3, channels %% 4 == 0");
    BE_REQUIRE(layout_chw_hw == 0 || (ksize == 1 && cin % layout_chw_hw == 0), "be_conv_wgrad_f32: bad layout_chw_hw");
    BE_REQUIRE(be::aligned16(x) && be::aligned16(dy), "be_conv_wgrad_f32: x / dy must be 16-byte aligned");
    const int taps = ksize * ksize;
    const int ct = (cout + 63) / 64, it = (cin + 63) / 64;
    const int64_t wsize = (int64_t)cout * cin * taps;
    int S = pick_splits(M, ct * it * taps, 64, 512);       // every split is another full copy of dW to sum
    while (S > 1 && (size_t)S * wsize * sizeof(float) > partial_bytes) --S;
    BE_REQUIRE((size_t)S * wsize * sizeof(float) <= partial_bytes, "be_conv_wgrad_f32: scratch too small");
    int rows = (M + S - 1) / S; rows = (rows + 31) / 32 * 32;
    S = (M + rows - 1) / rows;
    WgradArgs a{x, dy, partial, M, h, w, h * w, cin, cout, ksize, rows, layout_chw_hw, it, 0};
    hipLaunchKernelGGL(k_wgrad, dim3(ct * it, taps, S), dim3(256), 0, s, a);
    *S_out = S; *wsize_out = wsize;
    return be::check_launch("be_conv_wgrad_f32");
}

extern "C" int be_conv_wgrad_f32(const float* x, const float* dy, float* dw, int n, int h, int w, int cin, int cout,
                                 int ksize, int layout_chw_hw, void* scratch, size_t scratch_bytes, void* stream) {
    BE_REQUIRE(x && dy && dw && scratch, "be_conv_wgrad_f32: null pointer");
    BE_REQUIRE(n > 0 && h > 0 && w > 0 && cin > 0 && cout > 0, "be_conv_wgrad_f32: empty");
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
    int S = 1; int64_t wsize = 0;
    const int rc = wgrad_slices(x, dy, static_cast<float*>(scratch), scratch_bytes, n, h, w, cin, cout, ksize, layout_chw_hw, s, &S, &wsize);
    if (rc) return rc;
    hipLaunchKernelGGL(k_sum_splits, dim3(cap_grid(wsize, 256)), dim3(256), 0, s, static_cast<const float*>(scratch), dw, wsize, S);
    return be::check_launch("be_conv_wgrad_f32");
}

// ---------------------------------------------------------------------------------------------- training units (round 3)
namespace {
struct RowBlocks { int n, rows; };
inline RowBlocks row_blocks(int M, int want) {             // `want` blocks of a multiple of 32 rows (>= 1 block)
    if (want < 1) want = 1;
    int rows = (M + want - 1) / want;
    rows = (rows + UR - 1) / UR * UR;
    return RowBlocks{(M + rows - 1) / rows, rows};
}
// row blocks of the two reductions (= partials the apply kernels' prologues walk: 512 B each per workgroup): enough workgroups
// to cover the chip for the narrow matrices (conv1: 2 column blocks over 28 224 rows), few partials for the wide ones
inline RowBlocks stat_blocks(int M, int C) {
    const int cb = C / UC, cap = cb >= 8 ? 36 : (cb >= 3 ? 64 : 128);
    int w = (M + 63) / 64;
    return row_blocks(M, w > cap ? cap : w);
}
inline RowBlocks apply_blocks(int M, int C, int cap) {
    const int cb = C / UC;
    int w = (768 + cb - 1) / cb;
    if (w > cap) w = cap;
    return row_blocks(M, w);
}
}  // namespace

// ---- host side of the units.  One or TWO units per call: a pair shares every launch (see k_unit_gemms); each unit of a pair works
// in its own half of every scratch region.
namespace {
struct ScrPart { char *stats, *dbpart, *conv, *wgrad; size_t stats_b, dbpart_b, conv_b, wgrad_b; };
inline ScrPart scr_part(char* sc, int j, int nu) {
    const size_t st = (SCR_DBPART - SCR_STATS) / nu, db = (SCR_CONV - SCR_DBPART) / nu, cv = ((SCR_WGRAD - SCR_CONV) / nu) & ~(size_t)255,
                 wg = ((SCR_TOTAL - SCR_WGRAD) / nu) & ~(size_t)255;
    return ScrPart{sc + SCR_STATS + j * st, sc + SCR_DBPART + j * db, sc + SCR_CONV + j * cv, sc + SCR_WGRAD + j * wg, st, db, cv, wg};
}

template <int CV>
int launch_unit_gemms(const UnitGemmsArgs& g, unsigned grid, hipStream_t s) {
    constexpr size_t lds = (size_t)BWD_GEMMS_LDS_FLOATS * sizeof(float);
    static be::DeviceFlags flags{};
    if (int rc_ = be::ensure_dynamic_lds(reinterpret_cast<const void*>(&k_unit_gemms<CV>), lds, flags)) return rc_;
    hipLaunchKernelGGL(k_unit_gemms<CV>, dim3(grid), dim3(256), lds, s, g);
    return BE_OK;
}
inline int launch_unit_gemms(int variant, const UnitGemmsArgs& g, unsigned grid, hipStream_t s) {
    return variant == 0 ? launch_unit_gemms<0>(g, grid, s) : variant == 1 ? launch_unit_gemms<1>(g, grid, s) : launch_unit_gemms<2>(g, grid, s);
}
inline int pad8(int v) { return (v + 7) / 8 * 8; }

int check_fwd_unit(const be_train_unit_fwd& u, const char* who) {
    BE_REQUIRE(u.x && u.packed_w && u.packed_bias && u.gamma && u.beta && u.y && u.mean && u.invstd && u.out, "%s: null pointer", who);
    const int64_t M64 = (int64_t)u.desc.n * u.desc.h * u.desc.w;
    const int C = u.desc.cout;
    BE_REQUIRE(M64 > 0 && M64 < ((int64_t)1 << 31) && C > 0 && C % UC == 0 && C <= 1024, "%s: cout %% 32 == 0, <= 1024", who);
    BE_REQUIRE(be::aligned16(u.y) && be::aligned16(u.out) && be::aligned16(u.gamma) && be::aligned16(u.beta) && be::aligned16(u.packed_bias) &&
               (!u.res || be::aligned16(u.res)) && (!u.s_in || be::aligned16(u.s_in)), "%s: 16-byte alignment", who);
    return BE_OK;
}

// the two BatchNorm launches of the forward for nu units of the same [M, C]; S / ldp: the convolutions' K slices (S = 1: y is final)
int fwd_bn_launches(const be_train_unit_fwd* const* u, int nu, const int* S, const int* ldp, float eps, float momentum, char* sc,
                    hipStream_t s, const char* who, const be_sk::ConvGeom* sk = nullptr, float* const* sk_part = nullptr) {
    const int M = u[0]->desc.n * u[0]->desc.h * u[0]->desc.w, C = u[0]->desc.cout;
    const RowBlocks sb = stat_blocks(M, C);
    const RowBlocks ab = apply_blocks(M, C, 256);
    Two<StatsArgs> st{};
    Two<FwdApplyArgs> fa{};
    for (int j = 0; j < nu; ++j) {
        const ScrPart sp = scr_part(sc, j, nu);
        BE_REQUIRE((size_t)sb.n * C * 2 * sizeof(double) <= sp.stats_b, "%s: statistics region too small", who);
        st.j[j] = StatsArgs{S[j] > 1 ? reinterpret_cast<const float*>(sp.conv) : nullptr, u[j]->packed_bias, u[j]->y,
                            reinterpret_cast<double*>(sp.stats), S[j] > 1 ? S[j] : 0, M, C, ldp[j], sb.rows, 0, {}};
        if (sk) { st.j[j].partial = sk_part[j]; st.j[j].S = 1; st.j[j].ldp = (C + 63) / 64 * 64; st.j[j].sk = 1; st.j[j].g = sk[j]; }
        fa.j[j] = FwdApplyArgs{u[j]->y, reinterpret_cast<const double*>(sp.stats), u[j]->gamma, u[j]->beta, u[j]->res, u[j]->run_mean,
                               u[j]->run_var, u[j]->mean, u[j]->invstd, u[j]->s_in, u[j]->out, sb.n, M, C, ab.rows, u[j]->act, eps, momentum};
    }
    hipLaunchKernelGGL(k_bn_stats, dim3(C / UC, sb.n, nu), dim3(256), 0, s, st);
    hipLaunchKernelGGL(k_bn_fwd_apply, dim3(C / UC, ab.n, nu), dim3(256), 0, s, fa);
    return BE_OK;
}

// BE_NO_TRAIN_SK: rounds 3-5's launches everywhere; BE_NO_TRAIN_SK_FWD: the forward alone (A/B knobs)
inline bool sk_fwd_enabled() {
    static const bool off = getenv("BE_NO_TRAIN_SK_FWD") != nullptr;
    return !off && be::sk_enabled();
}

int unit_fwd_one(const be_train_unit_fwd& u, float eps, float momentum, char* sc, void* stream, const char* who) {
    if (int rc = check_fwd_unit(u, who)) return rc;
    be_conv_desc dc = u.desc; dc.act = 0;
    int S = 1, ldp = 0;
    const be_train_unit_fwd* one[1] = {&u};
    if (sk_fwd_enabled() && be::sk_fwd_eligible(u.desc)) {
        // round 6: the convolution on the balanced persistent launch (be_train_sk.hip), raw slices summed by k_bn_stats
        be::SkFwdIn in{&u.desc, u.x, u.packed_w, reinterpret_cast<float*>(sc + SCR_CONV), SCR_WGRAD - SCR_CONV};
        be_sk::ConvGeom cg;
        be::SkPlan plan;
        if (be::sk_plan_fwd(&in, 1, &cg, &plan) == BE_OK) {
            if (int rc = be::sk_run(&plan, be::as_stream(stream))) return rc;
            float* part[1] = {in.part};
            return fwd_bn_launches(one, 1, &S, &ldp, eps, momentum, sc, be::as_stream(stream), who, &cg, part);
        }
    }
    if (int rc = be::conv_train(&dc, u.x, u.packed_w, u.packed_bias, nullptr, u.y, dc.cout, sc + SCR_CONV, SCR_WGRAD - SCR_CONV, &S, &ldp, stream))
        return rc;
    return fwd_bn_launches(one, 1, &S, &ldp, eps, momentum, sc, be::as_stream(stream), who);
}
}  // namespace

extern "C" int be_train_unit_fwd_f32(const be_conv_desc* d, const float* x, const float* pw, const float* pb, const float* gamma,
                                     const float* beta, const float* res, float eps, float momentum, float* run_mean,
                                     float* run_var, float* y, float* mean, float* invstd, float* s_in, float* out, int act,
                                     void* scratch, size_t scratch_bytes, void* stream) {
    BE_REQUIRE(d && scratch, "be_train_unit_fwd_f32: null pointer");
    BE_REQUIRE(scratch_bytes >= SCR_TOTAL && be::aligned16(scratch), "be_train_unit_fwd_f32: scratch of be_train_scratch_bytes() bytes required");
    const be_train_unit_fwd u{*d, x, pw, pb, gamma, beta, res, run_mean, run_var, y, mean, invstd, s_in, out, act};
    if (int rc = unit_fwd_one(u, eps, momentum, static_cast<char*>(scratch), stream, "be_train_unit_fwd_f32")) return rc;
    return be::check_launch("be_train_unit_fwd_f32");
}

extern "C" int be_train_unit_pair_fwd_f32(const be_train_unit_fwd* a, const be_train_unit_fwd* b, float eps, float momentum,
                                          void* scratch, size_t scratch_bytes, void* stream) {
    const char* who = "be_train_unit_pair_fwd_f32";
    BE_REQUIRE(a && b && scratch, "%s: null pointer", who);
    BE_REQUIRE(scratch_bytes >= SCR_TOTAL && be::aligned16(scratch), "%s: scratch of be_train_scratch_bytes() bytes required", who);
    if (int rc = check_fwd_unit(*a, who)) return rc;
    if (int rc = check_fwd_unit(*b, who)) return rc;
    BE_REQUIRE(a->desc.n == b->desc.n && a->desc.h == b->desc.h && a->desc.w == b->desc.w && a->desc.cout == b->desc.cout,
               "%s: the two units must produce the same [n,h,w,cout]", who);
    char* sc = static_cast<char*>(scratch);
    hipStream_t s = be::as_stream(stream);
    const be_train_unit_fwd* u[2] = {a, b};
    static const bool no_pair = getenv("BE_NO_UNIT_PAIR") != nullptr;             // A/B knob
    if (!no_pair && sk_fwd_enabled() && be::sk_fwd_eligible(a->desc) && be::sk_fwd_eligible(b->desc)) {
        // round 6: both convolutions as problems of ONE balanced persistent launch; each unit's slices in its half of the region
        be::SkFwdIn in[2];
        for (int j = 0; j < 2; ++j) {
            const ScrPart sp = scr_part(sc, j, 2);
            in[j] = be::SkFwdIn{&u[j]->desc, u[j]->x, u[j]->packed_w, reinterpret_cast<float*>(sp.conv), sp.conv_b};
        }
        be_sk::ConvGeom cg[2];
        be::SkPlan plan;
        if (be::sk_plan_fwd(in, 2, cg, &plan) == BE_OK) {
            if (int rc = be::sk_run(&plan, s)) return rc;
            int S1[2] = {1, 1}, l0[2] = {0, 0};
            float* part[2] = {in[0].part, in[1].part};
            if (int rc = fwd_bn_launches(u, 2, S1, l0, eps, momentum, sc, s, who, cg, part)) return rc;
            return be::check_launch(who);
        }
    }
    be::ConvPrep prep[2];
    bool together = !no_pair;
    for (int j = 0; j < 2 && together; ++j) {
        const ScrPart sp = scr_part(sc, j, 2);
        be_conv_desc dc = u[j]->desc; dc.act = 0;
        if (int rc = be::conv_train_prepare(&dc, u[j]->x, u[j]->packed_w, u[j]->packed_bias, nullptr, u[j]->y, dc.cout, sp.conv, sp.conv_b, &prep[j]))
            return rc;
        together = prep[j].variant >= 0 && (j == 0 || prep[j].variant == prep[0].variant);
    }
    if (!together) {          // shapes the merged launch does not take: one after the other (same results)
        if (int rc = unit_fwd_one(*a, eps, momentum, sc, stream, who)) return rc;
        if (int rc = unit_fwd_one(*b, eps, momentum, sc, stream, who)) return rc;
        return be::check_launch(who);
    }
    UnitGemmsArgs g{};
    int S[2], ldp[2];
    for (int j = 0; j < 2; ++j) {
        g.ca[j] = prep[j].args; g.cgx[j] = (int)prep[j].gx; g.c_real[j] = (int)prep[j].gx * prep[j].S;
        S[j] = prep[j].S; ldp[j] = prep[j].ldp;
    }
    g.w_end[0] = g.w_end[1] = 0;
    g.c_end0 = pad8(g.c_real[0]);
    {
        be::ProfileScope prof(s, BE_KERNEL_TRAIN_BWD_GEMMS, prep[0].flops + prep[1].flops, 0.0, prep[0].flops_exec + prep[1].flops_exec);
        if (int rc = launch_unit_gemms(prep[0].variant, g, (unsigned)(g.c_end0 + g.c_real[1]), s)) return rc;
    }
    if (int rc = fwd_bn_launches(u, 2, S, ldp, eps, momentum, sc, s, who)) return rc;
    return be::check_launch(who);
}

namespace {
int check_bwd_unit(const be_train_unit_bwd& u, const char* who) {
    BE_REQUIRE(u.x && u.dout && u.y && u.mean && u.invstd && u.gamma && u.ds && u.dy && u.dgamma && u.dbeta && u.dw && u.db, "%s: null pointer", who);
    BE_REQUIRE((u.dx != nullptr) == (u.dgrad_packed_w != nullptr), "%s: dx and the data-gradient pack go together", who);
    const int64_t M64 = (int64_t)u.desc.n * u.desc.h * u.desc.w;
    const int C = u.desc.cout;
    BE_REQUIRE(M64 > 0 && M64 < ((int64_t)1 << 31) && C > 0 && C % UC == 0 && C <= 1024, "%s: cout %% 32 == 0, <= 1024", who);
    BE_REQUIRE(be::aligned16(u.dout) && be::aligned16(u.y) && be::aligned16(u.ds) && be::aligned16(u.dy) && be::aligned16(u.mean) &&
               be::aligned16(u.invstd) && be::aligned16(u.gamma) && (!u.s_in || be::aligned16(u.s_in)) && be::aligned16(u.dw) &&
               (!u.dx || be::aligned16(u.dx)) && (!u.dx_add || be::aligned16(u.dx_add)), "%s: 16-byte alignment", who);
    if (u.desc.ksize != 7) {
        BE_REQUIRE((u.desc.ksize == 1 || u.desc.ksize == 3) && u.desc.cin % 4 == 0 && C % 4 == 0, "%s: ksize 1|3, channels %% 4 == 0", who);
        BE_REQUIRE(u.layout_chw_hw == 0 || (u.desc.ksize == 1 && u.desc.cin % u.layout_chw_hw == 0), "%s: bad layout_chw_hw", who);
        BE_REQUIRE(be::aligned16(u.x), "%s: x must be 16-byte aligned", who);
    }
    return BE_OK;
}

// the two BatchNorm launches of the backward for nu units of the same [M, C]: ds and its column sums; dgamma / dbeta, dy and the
// bias-gradient partials.  -> row blocks of the second (the partials k_bwd_post sums)
int bwd_bn_launches(const be_train_unit_bwd* const* u, int nu, char* sc, hipStream_t s, int* nb_rows, const char* who) {
    const int M = u[0]->desc.n * u[0]->desc.h * u[0]->desc.w, C = u[0]->desc.cout;
    const RowBlocks sb = stat_blocks(M, C);
    const RowBlocks ab = apply_blocks(M, C, 128);
    Two<BwdReduceArgs> ra{};
    Two<BwdApplyArgs> ba{};
    for (int j = 0; j < nu; ++j) {
        const ScrPart sp = scr_part(sc, j, nu);
        BE_REQUIRE((size_t)sb.n * C * 2 * sizeof(double) <= sp.stats_b && (size_t)ab.n * C * sizeof(double) <= sp.dbpart_b,
                   "%s: reduction regions too small", who);
        double* part = reinterpret_cast<double*>(sp.stats);
        ra.j[j] = BwdReduceArgs{u[j]->dout, u[j]->s_in, u[j]->y, u[j]->mean, u[j]->invstd, u[j]->ds, part, M, C, sb.rows};
        ba.j[j] = BwdApplyArgs{u[j]->ds, u[j]->y, u[j]->mean, u[j]->invstd, u[j]->gamma, part, u[j]->dy, u[j]->dgamma, u[j]->dbeta,
                               reinterpret_cast<double*>(sp.dbpart), sb.n, M, C, ab.rows, 1.0f / (float)M};
    }
    // 1. ds = dout * smish'(s_in); row-block column sums of ds and ds * xhat
    hipLaunchKernelGGL(k_bn_bwd_reduce, dim3(C / UC, sb.n, nu), dim3(256), 0, s, ra);
    // 2. dgamma / dbeta finished in the prologue; dy; column sums of dy per workgroup
    hipLaunchKernelGGL(k_bn_bwd_apply, dim3(C / UC, ab.n, nu), dim3(256), 0, s, ba);
    *nb_rows = ab.n;
    return BE_OK;
}

// the weight-gradient job of a 1x1 / 3x3 unit: tiles, K slices (partials at wpart, at most wpart_bytes) and the matching post role
int plan_wgrad(const be_train_unit_bwd& u, float* wpart, size_t wpart_bytes, WJob* wj, PostJob* pj, double* flops_exec, const char* who) {
    const be_conv_desc* d = &u.desc;
    const int M = d->n * d->h * d->w, C = d->cout;
    const int taps = d->ksize * d->ksize;
    const int64_t wsize = (int64_t)C * d->cin * taps;
    static const bool no128 = getenv("BE_NO_WGRAD128") != nullptr;            // A/B knob
    if (!no128 && u.layout_chw_hw == 0 && C % 128 == 0 && d->cin % 128 == 0 && M >= 256) {
        const int tiles = (C / 128) * (d->cin / 128) * taps;
        // slices: every one is another copy of dW to write and to sum (a 3x3 layer: 2.4-5.3 MB each), so few of them -
        // ~400 workgroups, at most 8 slices for the 3x3 layers, 24 for the (small) 1x1 ones
        int S = (400 + tiles - 1) / tiles;
        const int s_cap = taps == 9 ? 8 : 24;
        if (S > s_cap) S = s_cap;
        if (S < 1) S = 1;
        if (S > M / 64) S = M / 64;
        while (S > 1 && (size_t)S * wsize * sizeof(float) > wpart_bytes) --S;
        BE_REQUIRE((size_t)S * wsize * sizeof(float) <= wpart_bytes, "%s: scratch too small", who);
        int rows = (M + S - 1) / S; rows = (rows + 15) / 16 * 16;
        S = (M + rows - 1) / rows;
        wj->wkind = 1;
        static const bool no_pm = getenv("BE_NO_WGRAD_PM") != nullptr;          // A/B knob
        const int pm = !no_pm && d->ksize == 3 && d->n % 16 == 0 && (int64_t)M * (C > d->cin ? C : d->cin) * 4 < ((int64_t)1 << 32);
        wj->w128 = Wgrad128Args{u.x, u.dy, wpart, M, d->h, d->w, d->h * d->w, d->cin, C, d->ksize, rows, d->cin / 128, pm};
        wj->wx = (C / 128) * (d->cin / 128); wj->wy = taps; wj->real = wj->wx * taps * S;
        pj->wpart = wpart; pj->dw = u.dw; pj->wsize = wsize; pj->wS = S; pj->conv1_map = 0; pj->cout1 = C; pj->wtaps = taps;
        pj->nb_w = (int)cap_grid(wsize / taps, 256, 1024);
        // executed: the pixel-major walk skips the (pixel, tap) pairs outside the image: (3H - 2)(3W - 2) of 9 HW remain
        *flops_exec = pm ? 2.0 * d->n * (3.0 * d->h - 2.0) * (3.0 * d->w - 2.0) * C * d->cin : 2.0 * M * (double)wsize;
    } else {
        const int ct = (C + 63) / 64, it = (d->cin + 63) / 64;
        int S = pick_splits(M, ct * it * taps, 64, 512);       // every split is another full copy of dW to sum
        while (S > 1 && (size_t)S * wsize * sizeof(float) > wpart_bytes) --S;
        BE_REQUIRE((size_t)S * wsize * sizeof(float) <= wpart_bytes, "%s: scratch too small", who);
        int rows = (M + S - 1) / S; rows = (rows + 31) / 32 * 32;
        S = (M + rows - 1) / rows;
        wj->wkind = 0;
        static const bool no_pm64 = getenv("BE_NO_WGRAD_PM") != nullptr;        // A/B knob
        const int pm64 = !no_pm64 && d->ksize == 3 && d->n % 32 == 0 && u.layout_chw_hw == 0;
        wj->w64 = WgradArgs{u.x, u.dy, wpart, M, d->h, d->w, d->h * d->w, d->cin, C, d->ksize, rows, u.layout_chw_hw, it, pm64};
        wj->wx = ct * it; wj->wy = taps; wj->real = ct * it * taps * S;
        pj->wpart = wpart; pj->dw = u.dw; pj->wsize = wsize; pj->wS = S; pj->conv1_map = 0; pj->cout1 = C; pj->wtaps = 0;
        pj->nb_w = (int)cap_grid(wsize / 4, 256, 1024);
        // executed: 32 x 32 sub-tiles (waves whose sub-tile lies outside the matrix issue nothing)
        *flops_exec = (pm64 ? 2.0 * d->n * (3.0 * d->h - 2.0) * (3.0 * d->w - 2.0) : 2.0 * M * (double)taps) * ((C + 31) / 32 * 32) * ((d->cin + 31) / 32 * 32);
    }
    return BE_OK;
}
inline dim3 wjob_grid(const WJob& w) { return dim3(w.wx, w.wy, w.real / (w.wx * w.wy)); }

// Round 6: the balanced launch (be_train_sk.hip) for the units it takes.  Fills the post kernel's roles for unit j of `pa` (the
// bias role's partials are the caller's).  The weight-gradient slices are [slice][tap][cout][cin] as k_wgrad128's.
void sk_post_roles(const be_train_unit_bwd& u, const be::SkUnitIn& in, const be::SkUnitOut& out, PostArgs* pa, int j) {
    const be_conv_desc* d = &u.desc;
    const int taps = d->ksize * d->ksize;
    PostJob& pj = pa->j[j];
    pj.wpart = in.wpart; pj.dw = u.dw; pj.wsize = (int64_t)d->cout * d->cin * taps; pj.wS = 1; pj.conv1_map = 0; pj.cout1 = d->cout;
    pj.wtaps = taps; pj.chw = u.layout_chw_hw;
    pj.nb_w = (int)cap_grid(pj.wsize / taps, 256, 1024);
    pa->sk = 1;
    pa->wsk[j] = out.wg;
    pa->xsk[j] = out.cg;
}
// one unit: true when the balanced launch ran (or failed: *rc), false = "not mine" (the caller takes rounds 3-5's path)
bool sk_single(const be_train_unit_bwd& u, char* sc, hipStream_t s, PostArgs* pa, const char* who, int* rc) {
    const be_conv_desc* d = &u.desc;
    be::SkUnitIn in{&u, reinterpret_cast<float*>(sc + SCR_CONV), SCR_WGRAD - SCR_CONV, reinterpret_cast<float*>(sc + SCR_WGRAD), SCR_TOTAL - SCR_WGRAD};
    be::SkUnitOut out;
    be::SkPlan plan;
    if (be::sk_plan(&in, 1, &out, &plan)) return false;
    *rc = be::sk_run(&plan, s);
    if (*rc) return true;
    sk_post_roles(u, in, out, pa, 0);
    pa->xpart = in.cpart; pa->xadd = u.dx_add; pa->dx = u.dx; pa->xM = (int64_t)d->n * d->h * d->w; pa->xC = d->cin;
    pa->xldp = out.ldp; pa->xS = 1;
    (void)who;
    return true;
}

int unit_bwd_one(const be_train_unit_bwd& u, char* sc, void* stream, const char* who) {
    if (int rc = check_bwd_unit(u, who)) return rc;
    const be_conv_desc* d = &u.desc;
    const int M = d->n * d->h * d->w, C = d->cout;
    hipStream_t s = be::as_stream(stream);
    const be_train_unit_bwd* one[1] = {&u};
    int nb_rows = 0;
    if (int rc = bwd_bn_launches(one, 1, sc, s, &nb_rows, who)) return rc;
    // 3. + 4. the weight-gradient GEMM and the data-gradient convolution (through the transposed / mirrored pack): slices of both
    //    stay in scratch.  With an input gradient wanted the two run as ONE launch (k_unit_gemms), else the weight gradient alone.
    PostArgs pa{};
    pa.nj = 1;
    PostJob& pj = pa.j[0];
    int rc_sk = BE_OK;
    float* wpart = reinterpret_cast<float*>(sc + SCR_WGRAD);
    pj.dbpart = reinterpret_cast<const double*>(sc + SCR_DBPART); pj.db = u.db; pj.nb_rows = nb_rows; pj.C = C; pj.nb_b = C / 32;
    if (d->ksize == 7) {
        BE_REQUIRE(d->cin == 4 && C == 64 && !u.dx, "%s: ksize 7 is conv1 (NHWC4 staging, 64 outputs, no input gradient)", who);
        int rows = (M + 63) / 64; rows = (rows + 31) / 32 * 32;
        const int S = (M + rows - 1) / rows;
        BE_REQUIRE((size_t)S * C * 224 * sizeof(float) <= SCR_TOTAL - SCR_WGRAD, "%s: scratch too small", who);
        WgradC1Args wa{u.x, u.dy, wpart, M, d->h, d->w, d->h * d->w, C, rows};
        hipLaunchKernelGGL(k_wgrad_conv1_mfma, dim3(1, 4, S), dim3(256), 0, s, wa);
        pj.wpart = wpart; pj.dw = u.dw; pj.wsize = (int64_t)C * 147; pj.wS = S; pj.conv1_map = 1; pj.cout1 = C;
        pj.nb_w = (C * 147 + 255) / 256;
    } else if (be::sk_enabled() && be::sk_eligible(u) && sk_single(u, sc, s, &pa, who, &rc_sk)) {
        if (rc_sk) return rc_sk;
    } else {
        UnitGemmsArgs g{};
        double w_exec = 0.0;
        if (int rc = plan_wgrad(u, wpart, SCR_TOTAL - SCR_WGRAD, &g.w[0], &pj, &w_exec, who)) return rc;
        static const bool no_merge = getenv("BE_NO_BWD_MERGE") != nullptr;        // A/B knob
        be::ConvPrep prep;
        prep.variant = -1;
        be_conv_desc dd{d->n, d->h, d->w, C, d->cin, d->ksize, 0};
        if (u.dx && !no_merge)
            if (int rc = be::conv_train_prepare(&dd, u.dy, u.dgrad_packed_w, u.dgrad_packed_bias, u.dx_add, u.dx, d->cin, sc + SCR_CONV,
                                                SCR_WGRAD - SCR_CONV, &prep, true)) return rc;
        if (prep.variant >= 0) {
            g.ca[0] = prep.args; g.cgx[0] = (int)prep.gx; g.c_real[0] = (int)prep.gx * prep.S; g.c_real[1] = 0;
            g.w_end[0] = g.w_end[1] = pad8(g.w[0].real);
            g.c_end0 = g.c_real[0];
            const double wflops = 2.0 * M * (double)pj.wsize;
            be::ProfileScope prof(s, BE_KERNEL_TRAIN_BWD_GEMMS, prep.flops + wflops, 0.0, prep.flops_exec + w_exec);
            if (int rc = launch_unit_gemms(prep.variant, g, (unsigned)(g.w_end[1] + g.c_real[0]), s)) return rc;
            if (prep.S > 1) {
                pa.xpart = reinterpret_cast<const float*>(sc + SCR_CONV); pa.xadd = u.dx_add; pa.dx = u.dx; pa.xM = M; pa.xC = d->cin;
                pa.xldp = prep.ldp; pa.xS = prep.S;
            }
        } else {
            if (g.w[0].wkind == 1) hipLaunchKernelGGL(k_wgrad128, wjob_grid(g.w[0]), dim3(256), 0, s, g.w[0].w128);
            else hipLaunchKernelGGL(k_wgrad, wjob_grid(g.w[0]), dim3(256), 0, s, g.w[0].w64);
            if (u.dx) {
                int S = 1, ldp = 0;
                if (int rc = be::conv_train(&dd, u.dy, u.dgrad_packed_w, u.dgrad_packed_bias, u.dx_add, u.dx, d->cin, sc + SCR_CONV,
                                            SCR_WGRAD - SCR_CONV, &S, &ldp, stream)) return rc;
                if (S > 1) {
                    pa.xpart = reinterpret_cast<const float*>(sc + SCR_CONV); pa.xadd = u.dx_add; pa.dx = u.dx; pa.xM = M; pa.xC = d->cin;
                    pa.xldp = ldp; pa.xS = S;
                }
            }
        }
    }
    const int nb_x = pa.xS ? (int)cap_grid((int64_t)M * (d->cin / 4), 256, 2048) : 0;
    // 5. slices -> dW, partials -> db, slices (+ the other branch) -> dx: one launch
    hipLaunchKernelGGL(k_bwd_post, dim3(pj.nb_w + pj.nb_b + nb_x), dim3(256), 0, s, pa);
    return BE_OK;
}
}  // namespace

extern "C" int be_train_unit_bwd_f32(const be_conv_desc* d, const float* x, const float* dout, const float* s_in, const float* y,
                                     const float* mean, const float* invstd, const float* gamma, const float* dgrad_pw,
                                     const float* dgrad_pb, const float* dx_add, int layout_chw_hw, float* ds, float* dy,
                                     float* dgamma, float* dbeta, float* dw, float* db, float* dx, void* scratch,
                                     size_t scratch_bytes, void* stream) {
    BE_REQUIRE(d && scratch, "be_train_unit_bwd_f32: null pointer");
    BE_REQUIRE(scratch_bytes >= SCR_TOTAL && be::aligned16(scratch), "be_train_unit_bwd_f32: scratch of be_train_scratch_bytes() bytes required");
    const be_train_unit_bwd u{*d, x, dout, s_in, y, mean, invstd, gamma, dgrad_pw, dgrad_pb, dx_add, layout_chw_hw, ds, dy, dgamma, dbeta, dw, db, dx};
    if (int rc = unit_bwd_one(u, static_cast<char*>(scratch), stream, "be_train_unit_bwd_f32")) return rc;
    return be::check_launch("be_train_unit_bwd_f32");
}

extern "C" int be_train_unit_pair_bwd_f32(const be_train_unit_bwd* a, const be_train_unit_bwd* b, void* scratch, size_t scratch_bytes,
                                          void* stream) {
    const char* who = "be_train_unit_pair_bwd_f32";
    BE_REQUIRE(a && b && scratch, "%s: null pointer", who);
    BE_REQUIRE(scratch_bytes >= SCR_TOTAL && be::aligned16(scratch), "%s: scratch of be_train_scratch_bytes() bytes required", who);
    if (int rc = check_bwd_unit(*a, who)) return rc;
    if (int rc = check_bwd_unit(*b, who)) return rc;
    BE_REQUIRE(a->desc.ksize != 7 && b->desc.ksize != 7, "%s: ksize 1|3", who);
    BE_REQUIRE(a->desc.n == b->desc.n && a->desc.h == b->desc.h && a->desc.w == b->desc.w && a->desc.cout == b->desc.cout &&
               a->desc.cin == b->desc.cin && a->x == b->x, "%s: the two units must share the input and produce the same [n,h,w,cout]", who);
    BE_REQUIRE(a->dx && b->dx && a->dx != b->dx && !a->dx_add && !b->dx_add,
               "%s: both dx buffers required (a->dx receives the sum, b->dx is working space), dx_add must be NULL", who);
    char* sc = static_cast<char*>(scratch);
    hipStream_t s = be::as_stream(stream);
    const be_train_unit_bwd* u[2] = {a, b};
    const be_conv_desc* d = &a->desc;
    const int M = d->n * d->h * d->w, C = d->cout;
    static const bool no_pair = getenv("BE_NO_UNIT_PAIR") != nullptr;             // A/B knob
    if (!no_pair && be::sk_enabled() && be::sk_eligible(*a) && be::sk_eligible(*b)) {
        // Round 6: both units' weight gradients and data gradients as ONE balanced launch (be_train_sk.hip).  The slice regions are
        // split in proportion to what a slice of each unit takes (a 3x3 weight gradient is nine times its block's 1x1 one).
        const int64_t w0 = (int64_t)a->desc.ksize * a->desc.ksize, w1 = (int64_t)b->desc.ksize * b->desc.ksize;
        const size_t wreg = SCR_TOTAL - SCR_WGRAD, creg = SCR_WGRAD - SCR_CONV;
        const size_t wa = (size_t)((double)wreg * w0 / (w0 + w1)) & ~(size_t)255;
        be::SkUnitIn in[2] = {{a, reinterpret_cast<float*>(sc + SCR_CONV), creg / 2, reinterpret_cast<float*>(sc + SCR_WGRAD), wa},
                              {b, reinterpret_cast<float*>(sc + SCR_CONV + creg / 2), creg / 2, reinterpret_cast<float*>(sc + SCR_WGRAD + wa), wreg - wa}};
        be::SkUnitOut out[2];
        be::SkPlan plan;
        if (be::sk_plan(in, 2, out, &plan) == BE_OK) {
            int nb_rows = 0;
            if (int rc = bwd_bn_launches(u, 2, sc, s, &nb_rows, who)) return rc;
            if (int rc = be::sk_run(&plan, s)) return rc;
            PostArgs pa{};
            pa.nj = 2;
            for (int j = 0; j < 2; ++j) {
                const ScrPart sp = scr_part(sc, j, 2);
                PostJob& pj = pa.j[j];
                pj.dbpart = reinterpret_cast<const double*>(sp.dbpart); pj.db = u[j]->db; pj.nb_rows = nb_rows; pj.C = C; pj.nb_b = C / 32;
                sk_post_roles(*u[j], in[j], out[j], &pa, j);
            }
            pa.xpart = in[0].cpart; pa.xldp = out[0].ldp; pa.xS = 1;
            pa.xpart2 = in[1].cpart; pa.xldp2 = out[1].ldp; pa.xS2 = 1;
            pa.dx = a->dx; pa.xM = M; pa.xC = d->cin;
            const int nb_x = (int)cap_grid((int64_t)M * (d->cin / 4), 256, 2048);
            hipLaunchKernelGGL(k_bwd_post, dim3(pa.j[0].nb_w + pa.j[0].nb_b + pa.j[1].nb_w + pa.j[1].nb_b + nb_x), dim3(256), 0, s, pa);
            return be::check_launch(who);
        }
    }
    be::ConvPrep prep[2];
    bool together = !no_pair;
    for (int j = 0; j < 2 && together; ++j) {
        const ScrPart sp = scr_part(sc, j, 2);
        be_conv_desc dd{d->n, d->h, d->w, C, d->cin, u[j]->desc.ksize, 0};
        // b with a single K slice writes its input gradient straight into b->dx, which the last kernel then adds
        if (int rc = be::conv_train_prepare(&dd, u[j]->dy, u[j]->dgrad_packed_w, u[j]->dgrad_packed_bias, nullptr, u[j]->dx, d->cin, sp.conv,
                                            sp.conv_b, &prep[j], true)) return rc;
        together = prep[j].variant >= 0 && (j == 0 ? prep[j].S > 1 && prep[j].S <= 8 : prep[j].variant == prep[0].variant && prep[j].S <= 8);
    }
    if (!together) {          // shapes the merged launches do not take: b, then a with b's input gradient added (same sums)
        if (int rc = unit_bwd_one(*b, sc, stream, who)) return rc;
        be_train_unit_bwd a2 = *a;
        a2.dx_add = b->dx;
        if (int rc = unit_bwd_one(a2, sc, stream, who)) return rc;
        return be::check_launch(who);
    }
    int nb_rows = 0;
    if (int rc = bwd_bn_launches(u, 2, sc, s, &nb_rows, who)) return rc;
    UnitGemmsArgs g{};
    PostArgs pa{};
    pa.nj = 2;
    double w_exec[2] = {0.0, 0.0}, wflops = 0.0;
    for (int j = 0; j < 2; ++j) {
        const ScrPart sp = scr_part(sc, j, 2);
        PostJob& pj = pa.j[j];
        pj.dbpart = reinterpret_cast<const double*>(sp.dbpart); pj.db = u[j]->db; pj.nb_rows = nb_rows; pj.C = C; pj.nb_b = C / 32;
        if (int rc = plan_wgrad(*u[j], reinterpret_cast<float*>(sp.wgrad), sp.wgrad_b, &g.w[j], &pj, &w_exec[j], who)) return rc;
        wflops += 2.0 * M * (double)pj.wsize;
        g.ca[j] = prep[j].args; g.cgx[j] = (int)prep[j].gx; g.c_real[j] = (int)prep[j].gx * prep[j].S;
    }
    g.w_end[0] = pad8(g.w[0].real);
    g.w_end[1] = g.w_end[0] + pad8(g.w[1].real);
    g.c_end0 = pad8(g.c_real[0]);
    {
        be::ProfileScope prof(s, BE_KERNEL_TRAIN_BWD_GEMMS, prep[0].flops + prep[1].flops + wflops, 0.0,
                              prep[0].flops_exec + prep[1].flops_exec + w_exec[0] + w_exec[1]);
        if (int rc = launch_unit_gemms(prep[0].variant, g, (unsigned)(g.w_end[1] + g.c_end0 + g.c_real[1]), s)) return rc;
    }
    pa.xpart = reinterpret_cast<const float*>(scr_part(sc, 0, 2).conv); pa.xldp = prep[0].ldp; pa.xS = prep[0].S;
    pa.dx = a->dx; pa.xM = M; pa.xC = d->cin;
    if (prep[1].S > 1) { pa.xpart2 = reinterpret_cast<const float*>(scr_part(sc, 1, 2).conv); pa.xldp2 = prep[1].ldp; pa.xS2 = prep[1].S; }
    else pa.xadd = b->dx;
    const int nb_x = (int)cap_grid((int64_t)M * (d->cin / 4), 256, 2048);
    hipLaunchKernelGGL(k_bwd_post, dim3(pa.j[0].nb_w + pa.j[0].nb_b + pa.j[1].nb_w + pa.j[1].nb_b + nb_x), dim3(256), 0, s, pa);
    return be::check_launch(who);
}

// Parameter gradients of a Linear over many rows (GlobalStage at 8 x 4096 tokens: y = x W^T + b): dW = dy^T x and db = column
// sums of dy in TWO launches (round 2: k_wgrad, k_sum_splits, k_col_sum, k_col_sum_final).  One grid holds the weight-gradient
// workgroups (k_wgrad128 tiles over S row slices) and the workgroups that form the column sums of dy per row block; k_bwd_post
// then sums the slices into dW and the partials into db.
namespace {
struct LinGradArgs { Wgrad128Args w; int n_w, wx; const float* dy; double* dbpart; int M, C, rows_per_block, cbx; };

__global__ __launch_bounds__(256, 2)
void k_lin_grads(LinGradArgs g) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int b = blockIdx.x;
    if (b < g.n_w) { wgrad128_body(g.w, smem, b % g.wx, 0, b / g.wx); return; }
    const int c = b - g.n_w, bx = c % g.cbx, by = c / g.cbx;
    const int tx = threadIdx.x & 7, ty = threadIdx.x >> 3;
    const int c_base = bx * UC, col = c_base + tx * 4;
    const int r0 = by * g.rows_per_block, r1 = min(g.M, r0 + g.rows_per_block);
    double v[1][4] = {{0, 0, 0, 0}};
    if (col < g.C)
        for (int r = r0 + ty; r < r1; r += UR) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(g.dy + (size_t)r * g.C + col);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[0][e] += t[e];
        }
    block_col_sums<1>(v, reinterpret_cast<double*>(smem), g.dbpart + (size_t)by * g.C, 1, g.C, c_base);
}
}  // namespace

extern "C" int be_linear_param_grads_f32(const float* x, const float* dy, float* dw, float* db, int M, int cin, int cout,
                                         void* scratch, size_t scratch_bytes, void* stream) {
    BE_REQUIRE(x && dy && dw && db && scratch && M > 0, "be_linear_param_grads_f32: bad arguments");
    BE_REQUIRE(cin % 128 == 0 && cout % 128 == 0 && cout <= 1024 && M >= 256, "be_linear_param_grads_f32: cin, cout multiples of 128, M >= 256");
    BE_REQUIRE(scratch_bytes >= SCR_TOTAL && be::aligned16(scratch) && be::aligned16(x) && be::aligned16(dy) && be::aligned16(dw),
               "be_linear_param_grads_f32: scratch of be_train_scratch_bytes() bytes, 16-byte aligned operands");
    char* sc = static_cast<char*>(scratch);
    hipStream_t s = be::as_stream(stream);
    float* wpart = reinterpret_cast<float*>(sc + SCR_WGRAD);
    double* dbpart = reinterpret_cast<double*>(sc + SCR_DBPART);
    const int tiles = (cout / 128) * (cin / 128);
    const int64_t wsize = (int64_t)cout * cin;
    int S = (512 + tiles - 1) / tiles;
    if (S > M / 256) S = M / 256;
    if (S > 128) S = 128;
    if (S < 1) S = 1;
    while (S > 1 && (size_t)S * wsize * sizeof(float) > SCR_TOTAL - SCR_WGRAD) --S;
    int rows = (M + S - 1) / S; rows = (rows + 15) / 16 * 16;
    S = (M + rows - 1) / rows;
    LinGradArgs g{};
    g.w = Wgrad128Args{x, dy, wpart, M, 1, 1, 1, cin, cout, 1, rows, cin / 128, 0};
    g.wx = tiles; g.n_w = tiles * S;
    const RowBlocks rb = row_blocks(M, 32);
    g.dy = dy; g.dbpart = dbpart; g.M = M; g.C = cout; g.rows_per_block = rb.rows; g.cbx = cout / UC;
    constexpr size_t lds = (size_t)WGRAD128_LDS_FLOATS * sizeof(float);
    static be::DeviceFlags f{};
    if (int rc_ = be::ensure_dynamic_lds(reinterpret_cast<const void*>(&k_lin_grads), lds, f)) return rc_;
    hipLaunchKernelGGL(k_lin_grads, dim3(g.n_w + g.cbx * rb.n), dim3(256), lds, s, g);
    PostArgs pa{};
    pa.nj = 1;
    PostJob& pj = pa.j[0];
    pj.wpart = wpart; pj.dw = dw; pj.wsize = wsize; pj.wS = S; pj.wtaps = 1; pj.cout1 = cout;
    pj.nb_w = S >= 32 ? (int)cap_grid(wsize, 64, 2048) : (int)cap_grid(wsize, 256, 1024);      // S >= 32: 64 positions per workgroup (post_job)
    pj.dbpart = dbpart; pj.db = db; pj.nb_rows = rb.n; pj.C = cout; pj.nb_b = cout / 32;
    hipLaunchKernelGGL(k_bwd_post, dim3(pj.nb_w + pj.nb_b), dim3(256), 0, s, pa);
    return be::check_launch("be_linear_param_grads_f32");
}

extern "C" int be_linear_small_fwd_f32(const float* x, const float* w, const float* b, float* y, int M, int K, int J, void* stream) {
    BE_REQUIRE(x && w && b && y && M > 0 && J > 0 && K > 0 && K % 4 == 0, "be_linear_small_fwd_f32: bad arguments (K %% 4 == 0)");
    BE_REQUIRE(be::aligned16(x) && be::aligned16(w), "be_linear_small_fwd_f32: x / w must be 16-byte aligned");
    hipLaunchKernelGGL(k_linear_small_fwd, dim3((M * J + 3) / 4), dim3(256), 0, be::as_stream(stream), x, w, b, y, M, K, J);
    return be::check_launch("be_linear_small_fwd_f32");
}

extern "C" int be_maxpool_nhwc_fwd_idx_f32(const float* x, float* y, unsigned char* idx, int n, int h, int w, int c, int k, int stride,
                                           int pad, void* stream) {
    BE_REQUIRE(x && y && idx && n > 0 && h > 0 && w > 0 && c > 0 && c % 4 == 0 && k >= 1 && k <= 15, "be_maxpool_nhwc_fwd_idx_f32: bad arguments");
    BE_REQUIRE(be::aligned16(x) && be::aligned16(y) && (reinterpret_cast<uintptr_t>(idx) & 3u) == 0, "be_maxpool_nhwc_fwd_idx_f32: alignment");
    const int oh = (h + 2 * pad - k) / stride + 1, ow = (w + 2 * pad - k) / stride + 1;
    const int64_t total = (int64_t)n * oh * ow * (c / 4);
    hipLaunchKernelGGL(k_maxpool_fwd_idx, dim3(cap_grid(total, 256)), dim3(256), 0, be::as_stream(stream), x, y, idx, n, h, w, c / 4,
                       oh, ow, k, stride, pad);
    return be::check_launch("be_maxpool_nhwc_fwd_idx_f32");
}

extern "C" int be_maxpool_nhwc_bwd_idx_f32(const unsigned char* idx, const float* dout, float* dx, int n, int h, int w, int c, int k,
                                           int stride, int pad, void* stream) {
    BE_REQUIRE(idx && dout && dx && n > 0 && h > 0 && w > 0 && c > 0 && c % 4 == 0, "be_maxpool_nhwc_bwd_idx_f32: bad arguments");
    BE_REQUIRE(be::aligned16(dout) && be::aligned16(dx) && (reinterpret_cast<uintptr_t>(idx) & 3u) == 0, "be_maxpool_nhwc_bwd_idx_f32: alignment");
    const int oh = (h + 2 * pad - k) / stride + 1, ow = (w + 2 * pad - k) / stride + 1;
    const int64_t total = (int64_t)n * h * w * (c / 4);
    hipLaunchKernelGGL(k_maxpool_bwd_idx, dim3(cap_grid(total, 256)), dim3(256), 0, be::as_stream(stream), idx, dout, dx, n, h, w,
                       c / 4, oh, ow, k, stride, pad);
    return be::check_launch("be_maxpool_nhwc_bwd_idx_f32");
}

extern "C" int be_linear_small_bwd_f32(const float* x, const float* w, const float* dy, float* dx, float* dw, float* db,
                                       int M, int K, int J, void* stream) {
    BE_REQUIRE(x && w && dy && dx && dw && db && M > 0 && K > 0 && J > 0, "be_linear_small_bwd_f32: bad arguments");
    BE_REQUIRE(J <= 256 && (int64_t)M * K < ((int64_t)1 << 31), "be_linear_small_bwd_f32: J <= 256");
    const int nb_dx = (M * K + 255) / 256, nb_dw = (J * K + 255) / 256;
    hipLaunchKernelGGL(k_linear_small_bwd, dim3(nb_dx + nb_dw + 1), dim3(256), 0, be::as_stream(stream), x, w, dy, dx, dw, db,
                       M, K, J, nb_dx, nb_dw);
    return be::check_launch("be_linear_small_bwd_f32");
}
