// Winograd convolution for the 3x3 layers on the 6x6 maps of LocalStage layers 1-3 (models/local_stage.py:20-28, 39-41).
// Tile shape (be_wino_math.h, BE_WINO_TH = 6, the default since round 4): F(6,3) along the rows x F(3,3) along the columns.  A 6x6
// output is TWO tiles of 6x3 outputs; each needs an 8x5 input window (one pixel of zero padding), and
//     Y = A^T [ (G g G^T) o (B^T d B) ] A      summed over the input channels
// turns the convolution into NPOS = 40 independent GEMMs [2N tiles x Cin] x [Cin x Cout] (one per position of the 8x5 transform
// domain): 80 multiplies per (cin, cout) pair and map instead of the 324 of the direct form - 4.05x less work for the matrix
// pipe, in exact fp32 products.  Interpolation points: rows 0, +-1, +-2, +-1/2, inf; columns 0, +-1, 2, inf.
// -DBE_WINO_TH=3 builds rounds 1-3's F(3x3,3x3): four 5x5 tiles per map, 25 GEMMs, 100 multiplies (an A/B target, `make wino3`).
// Accuracy on the whole network against the fp64 oracle: logits 0.9-4.9e-6 over random, trained and stressed weights (5x5 tiles:
// 1.8-3.0e-6; direct fp32 convolutions 1.1-3.3e-6), tolerance 1e-5 - DESIGN.md 3.1 / 4.
//
//   k_wino_pack    weights [Cout,Cin,3,3] (+ folded BatchNorm) -> U [NPOS][Cout_pad][Cin] in the 1x1 layout of k_conv_igemm
//   k_wino_in      x [N,6,6,C] NHWC -> V: tile-major [TPI N][NPOS][C] for large batches, plane-major [NPOS][TPI N][C] for small ones
//                  (HBM-bound: reads 36, writes TPI * NPOS = 80 values per channel)
//   k_wino_gemm    M[xi] = V[xi] U[xi]^T for the NPOS positions xi: one workgroup per 128x128 tile walks all of them (large
//                  batches; small ones go through be_conv_nhwc_batched_f32 on k_conv_igemm, same arithmetic per output element)
//   k_wino_gemm_ws the same GEMMs weight-stationary (B tile in registers, A streamed through LDS by DMA): batches of >= 4096 maps
//   k_wino_out     M -> y [N,6,6,Cout] + bias (+ residual) (+ Smish)
//   k_wino_out_in  conv1 -> conv2 of a residual block: output transform + Smish + input transform, the map stays in registers
#include <cstdlib>
#include "be_common.h"
#include "be_device_math.h"
#include "be_wino_math.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// tile shape (be_wino_math.h): TH x 3 outputs per tile from an NR x 5 window; TPI tiles per 6x6 map; NPOS positions = GEMMs
constexpr int TH = be::WINO_TH, NR = be::WINO_NR, TY = be::WINO_TY, TPI = be::WINO_TPI, NPOS = be::WINO_NPOS, NOUT = be::WINO_OUT;

inline unsigned grid_cap(int64_t total, int block, int64_t limit = 65535) {
    int64_t g = (total + block - 1) / block;
    return (unsigned)(g < 1 ? 1 : (g > limit ? limit : g));
}

__global__ void k_wino_pack(const float* __restrict__ w, const float* __restrict__ b, const float* __restrict__ gamma,
                            const float* __restrict__ beta, const float* __restrict__ mean, const float* __restrict__ var,
                            float eps, int cout, int cin, int cout_pad, float* __restrict__ U, float* __restrict__ bias) {
    const int64_t total = (int64_t)cout_pad * cin;
    const int64_t gs = (int64_t)gridDim.x * blockDim.x;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gs) {
        const int co = (int)(idx / cin), ci = (int)(idx % cin);
        float u[NR][5];
        if (co < cout) {
            const float scale = gamma ? gamma[co] / sqrtf(var[co] + eps) : 1.0f;
            const float* g = w + ((size_t)co * cin + ci) * 9;
            float t[NR][3];                                 // G_rows g: the kernel's three rows -> NR transform rows, column by column
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float col[NR];
                be::wino_g_rows(g[c] * scale, g[3 + c] * scale, g[6 + c] * scale, col);
#pragma unroll
                for (int r = 0; r < NR; ++r) t[r][c] = col[r];
            }
#pragma unroll
            for (int r = 0; r < NR; ++r) be::wino_g5(t[r][0], t[r][1], t[r][2], u[r]);     // (G_rows g) G^T
        } else {
#pragma unroll
            for (int r = 0; r < NR; ++r)
#pragma unroll
                for (int c = 0; c < 5; ++c) u[r][c] = 0.0f;
        }
#pragma unroll
        for (int r = 0; r < NR; ++r)
#pragma unroll
            for (int c = 0; c < 5; ++c) U[(size_t)(5 * r + c) * total + idx] = u[r][c];
    }
    for (int64_t co = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; co < cout_pad; co += gs) {
        float v = 0.0f;
        if (co < cout) {
            const float bb = b ? b[co] : 0.0f;
            v = gamma ? (bb - mean[co]) * (gamma[co] / sqrtf(var[co] + eps)) + beta[co] : bb;
        }
        bias[co] = v;
    }
}

// streaming accesses of the transform kernels: NT bit 0 = non-temporal loads, bit 1 = non-temporal stores (A/B: BE_WINO_NT)
template <int NT, class V> __device__ __forceinline__ V ld_s(const V* p) {
    if constexpr (NT & 1) return __builtin_nontemporal_load(p); else return *p;
}
template <int NT, class V> __device__ __forceinline__ void st_s(V* p, V v) {
    if constexpr (NT & 2) __builtin_nontemporal_store(v, p); else *p = v;
}
// channels per thread of the output-side transform kernels: with the 8x5 tiles a thread holding a channel QUAD needs 40 + 18 + 18
// float4 values live (304 registers: the compiler parked 64-130 of them in AGPRs and the kernels fell to 4.9 TB/s); a channel
// PAIR per thread needs half and runs two waves per SIMD.  Per-channel arithmetic: the width changes no bit.
constexpr int VW_OUT = TH == 6 ? 2 : 4;
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int W> struct VecOf;
template <> struct VecOf<4> { typedef f32x4 type; };
template <> struct VecOf<2> { typedef f32x2 type; };

// one thread = one tile (patch, ty, tx) x one channel quad; arithmetic: be_wino_math.h
template <int NT>
__global__ __launch_bounds__(256)
void k_wino_in(const float* __restrict__ x, float* __restrict__ V, int64_t n, int c4, int tm) {
    const int64_t total = n * TPI * c4;
    const int64_t gs = (int64_t)gridDim.x * blockDim.x;
    // float4 elements between transform positions / between tiles: plane-major V [NPOS][TPI n][C] (small batches) or
    // tile-major V [TPI n][NPOS][C] (large batches: a tile's NPOS x C block is one contiguous piece of HBM for this kernel)
    const int64_t plane = tm ? c4 : n * TPI * c4, ts = tm ? NPOS * c4 : c4;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gs) {
        const int cq = (int)(idx % c4);
        const int64_t tile = idx / c4;
        const int64_t img = tile / TPI;
        const int tt = (int)(tile - img * TPI), ty = tt >> 1, tx = tt & 1;
        const f32x4* src = reinterpret_cast<const f32x4*>(x) + img * 36 * c4 + cq;
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        f32x4 d[NR][5], v[NPOS];
#pragma unroll
        for (int r = 0; r < NR; ++r)
#pragma unroll
            for (int c = 0; c < 5; ++c) {
                const int yy = TH * ty - 1 + r, xx = 3 * tx - 1 + c;
                d[r][c] = ((unsigned)xx < 6u && (unsigned)yy < 6u) ? ld_s<NT>(src + (size_t)(yy * 6 + xx) * c4) : zero;
            }
        be::wino_in(d, v);
        f32x4* dst = reinterpret_cast<f32x4*>(V) + tile * ts + cq;
#pragma unroll
        for (int z = 0; z < NPOS; ++z) st_s<NT>(dst + (size_t)z * plane, v[z]);
    }
}

template <class V>
__device__ __forceinline__ V wino_act(V v, int act) {
    constexpr int W = (int)(sizeof(V) / sizeof(float));
    if (act == 1) {
#pragma unroll
        for (int k = 0; k < W; ++k) v[k] = be::smish(v[k]);
    } else if (act == 2) {
#pragma unroll
        for (int k = 0; k < W; ++k) v[k] = fmaxf(v[k], 0.f);
    }
    return v;
}

template <int NT, int VW = VW_OUT>
__global__ __launch_bounds__(256)
void k_wino_out(const float* __restrict__ M, const float* __restrict__ bias, const float* __restrict__ res,
                float* __restrict__ y, int64_t n, int c4, int act, int tm) {
    typedef typename VecOf<VW>::type vec;
    const int64_t total = n * TPI * c4;
    const int64_t gs = (int64_t)gridDim.x * blockDim.x;
    const int64_t plane = tm ? c4 : n * TPI * c4, ts = tm ? NPOS * c4 : c4;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gs) {
        const int cq = (int)(idx % c4);
        const int64_t tile = idx / c4;
        const int64_t img = tile / TPI;
        const int tt = (int)(tile - img * TPI), ty = tt >> 1, tx = tt & 1;
        const vec* src = reinterpret_cast<const vec*>(M) + tile * ts + cq;
        vec m[NPOS], o[NOUT], rv[NOUT];
#pragma unroll
        for (int z = 0; z < NPOS; ++z) m[z] = ld_s<NT>(src + (size_t)z * plane);
        // the residual's values are fetched with the transform-domain ones (inside the store loop each was a round trip of
        // its own: the kernel sat 58 % of its wave cycles in s_waitcnt)
        if (res) {
#pragma unroll
            for (int r = 0; r < TH; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    rv[3 * r + c] = ld_s<NT>(reinterpret_cast<const vec*>(res) + ((size_t)img * 36 + (TH * ty + r) * 6 + 3 * tx + c) * c4 + cq);
        }
        be::wino_out(m, o);
        const vec bv = reinterpret_cast<const vec*>(bias)[cq];
#pragma unroll
        for (int r = 0; r < TH; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const size_t e = ((size_t)img * 36 + (TH * ty + r) * 6 + 3 * tx + c) * c4 + cq;
                vec v = o[3 * r + c] + bv;
                if (res) v += rv[3 * r + c];
                st_s<NT>(reinterpret_cast<vec*>(y) + e, wino_act(v, act));
            }
    }
}

// conv1 -> conv2 of a residual block without the intermediate map in HBM: one thread = one image x one channel quad reads the
// TPI x NPOS transform-domain values of conv1's result, forms the 6x6 map (+ bias, Smish) in registers and writes the TPI x NPOS
// transform-domain values conv2's GEMMs read.  Saves the 36 values written and re-read (tile overlap) per channel.
template <int NT, int VW = VW_OUT>
__global__ __launch_bounds__(256, VW_OUT == 2 ? 2 : 1)
void k_wino_out_in(const float* __restrict__ M, const float* __restrict__ bias, float* __restrict__ V, int64_t n, int c4, int act,
                   int tm_in, int tm_out) {
    typedef typename VecOf<VW>::type vec;
    const int64_t total = n * c4;
    const int64_t gs = (int64_t)gridDim.x * blockDim.x;
    const int64_t plane = tm_in ? c4 : n * TPI * c4, ts = tm_in ? NPOS * c4 : c4;          // M as conv1's GEMMs wrote it
    const int64_t plane_o = tm_out ? c4 : n * TPI * c4, ts_o = tm_out ? NPOS * c4 : c4;    // V as conv2's GEMMs read it
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gs) {
        const int cq = (int)(idx % c4);
        const int64_t img = idx / c4;
        const vec bv = reinterpret_cast<const vec*>(bias)[cq];
        vec y[6][6];
#pragma unroll
        for (int ty = 0; ty < TY; ++ty)
#pragma unroll
            for (int tx = 0; tx < 2; ++tx) {
                const vec* src = reinterpret_cast<const vec*>(M) + (img * TPI + ty * 2 + tx) * ts + cq;
                vec m[NPOS], o[NOUT];
#pragma unroll
                for (int z = 0; z < NPOS; ++z) m[z] = ld_s<NT>(src + (size_t)z * plane);
                be::wino_out(m, o);
#pragma unroll
                for (int r = 0; r < TH; ++r)
#pragma unroll
                    for (int c = 0; c < 3; ++c) y[TH * ty + r][3 * tx + c] = wino_act(o[3 * r + c] + bv, act);
            }
        const vec zero = vec(0.f);
#pragma unroll
        for (int ty = 0; ty < TY; ++ty)
#pragma unroll
            for (int tx = 0; tx < 2; ++tx) {
                vec d[NR][5], v[NPOS];
#pragma unroll
                for (int r = 0; r < NR; ++r)
#pragma unroll
                    for (int c = 0; c < 5; ++c) {
                        const int yy = TH * ty - 1 + r, xx = 3 * tx - 1 + c;
                        d[r][c] = (xx >= 0 && xx < 6 && yy >= 0 && yy < 6) ? y[yy < 0 ? 0 : (yy > 5 ? 5 : yy)][xx < 0 ? 0 : (xx > 5 ? 5 : xx)] : zero;
                    }
                be::wino_in(d, v);
                vec* dst = reinterpret_cast<vec*>(V) + (img * TPI + ty * 2 + tx) * ts_o + cq;
#pragma unroll
                for (int z = 0; z < NPOS; ++z) st_s<NT>(dst + (size_t)z * plane_o, v[z]);
            }
    }
}

// Last block of LocalStage: output transform + bias + residual + activation + the 2x2 max-pool that follows it
// (models/local_stage.py:42,67: maxpool after layer3), one thread per image and channel quad: the 6x6 map exists only in
// registers, [N,3,3,C] is written (saves the map's round trip through HBM and the pooling launch).
template <int NT, int VW = VW_OUT>
__global__ __launch_bounds__(256, VW_OUT == 2 ? 2 : 1)
void k_wino_out_pool2(const float* __restrict__ M, const float* __restrict__ bias, const float* __restrict__ res,
                      float* __restrict__ y, int64_t n, int c4, int act, int tm) {
    typedef typename VecOf<VW>::type vec;
    const int64_t total = n * c4;
    const int64_t gs = (int64_t)gridDim.x * blockDim.x;
    const int64_t plane = tm ? c4 : n * TPI * c4, ts = tm ? NPOS * c4 : c4;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gs) {
        const int cq = (int)(idx % c4);
        const int64_t img = idx / c4;
        const vec bv = reinterpret_cast<const vec*>(bias)[cq];
        vec v[6][6];
#pragma unroll
        for (int ty = 0; ty < TY; ++ty)
#pragma unroll
            for (int tx = 0; tx < 2; ++tx) {
                const vec* src = reinterpret_cast<const vec*>(M) + (img * TPI + ty * 2 + tx) * ts + cq;
                vec m[NPOS], o[NOUT], rv[NOUT];
#pragma unroll
                for (int z = 0; z < NPOS; ++z) m[z] = ld_s<NT>(src + (size_t)z * plane);
                if (res) {                                   // with the tile's loads, not one by one behind the transform
#pragma unroll
                    for (int r = 0; r < TH; ++r)
#pragma unroll
                        for (int c = 0; c < 3; ++c)
                            rv[3 * r + c] = ld_s<NT>(reinterpret_cast<const vec*>(res) + ((size_t)img * 36 + (TH * ty + r) * 6 + 3 * tx + c) * c4 + cq);
                }
                be::wino_out(m, o);
#pragma unroll
                for (int r = 0; r < TH; ++r)
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        vec w = o[3 * r + c] + bv;
                        if (res) w += rv[3 * r + c];
                        v[TH * ty + r][3 * tx + c] = wino_act(w, act);
                    }
            }
#pragma unroll
        for (int py = 0; py < 3; ++py)
#pragma unroll
            for (int px = 0; px < 3; ++px) {
                vec m;
#pragma unroll
                for (int k = 0; k < VW; ++k)
                    m[k] = fmaxf(fmaxf(v[2 * py][2 * px][k], v[2 * py][2 * px + 1][k]), fmaxf(v[2 * py + 1][2 * px][k], v[2 * py + 1][2 * px + 1][k]));
                st_s<NT>(reinterpret_cast<vec*>(y) + ((size_t)img * 9 + py * 3 + px) * c4 + cq, m);
            }
    }
}

// ---- the 25 transform-domain GEMMs of one (M tile, N tile), walked by ONE workgroup --------------------------------
// M[z] = V[z] U[z]^T for z = 0..24: each problem has a K loop of only Cin/16 = 6-24 chunks, so as separate tiles every
// 12.6 MFLOP pay a prologue (first loads exposed), an epilogue and a workgroup turnover.  Here the software pipeline runs
// straight through the problem boundaries; at a boundary the accumulators are stored raw (no bias / residual / activation:
// the output transform does those) and cleared.  128x128 tile, 2x2 waves of 2x2 MFMA tiles (32x32x2 f32).
//
// Operand staging is global -> LDS directly (global_load_lds_dwordx4: no VGPR round trip, no ds_write, no per-chunk vector
// address arithmetic).  Knock-out runs of the register-staged version (tools/wino_gemm_lab.hip) showed that
// the MFMA pipe loses about as many cycles as the other instructions of the resident waves move registers: a loop of
// nothing but MFMAs runs at 99 % of the matrix rate, + the fragment ds_reads 94 %, + ds_writes 91 %, + global loads into
// registers 80 %; barriers and load latency cost nothing measurable.  So the loop carries as little else as it can.
// One DMA instruction of a wave fills a 1-KB piece = 16 rows x 64 B, lane-linear (LDS destination = piece base + lane x
// 16); without row padding the fragment reads would be 4-way bank conflicts, so the 16-byte quads of a row are
// XOR-swizzled by (row >> 2) & 3 - on the SOURCE address of the DMA and on the ds_read_b128 of the fragments
// (conflict-free for the four 16-lane groups of that instruction).  Rows past M load a valid row and are never stored.
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

struct GemmArgs {
    const float* x;       // problem z: rows [M][K] at x + z * xb
    const float* w;       //            [Npad][K] at w + z * wb
    float* y;             //            [M][ldy] at y + z * yb
    int M, K, N, ldy, nb, m_tiles, n_tiles;
    int64_t xb, wb, yb;
    int lda;              // floats between rows of x (K for a plain matrix; 25 K for the tile-major Winograd V [4n][25][K])
    // the same kernel as a plain row GEMM (1x1 convolutions, linears: EPI = 1): the "problems" a workgroup walks are nb
    // consecutive 128-row tiles of ONE matrix (short K loops get the same continuous pipeline): mrows = 128 nb rows per
    // workgroup, zrows = 128 rows per problem, xb = 128 K, yb = 128 ldy, wb = 0.  Winograd: mrows = 128, zrows = 0.
    int mrows, zrows;
    const float* bias;    // EPI = 1: y = act(acc + bias[col] (+ res[row][col]))
    const float* res;     //          same row stride as y
    int act;
};

template <int EPI>
__global__ __launch_bounds__(256, 3)
void k_wino_gemm(GemmArgs a) {
    constexpr int BM = 128, BN = 128, BKT = 16;
    constexpr int STAGE = (BM + BN) * BKT;             // floats per stage: A 128 x 16, then B 128 x 16 (16 KB)
    extern __shared__ __attribute__((aligned(16))) float smem_w[];
    const int bid = blockIdx.x;
    const int xcd = bid & 7, slot = bid >> 3;          // all N tiles of an M tile on one XCD
    const int n_tile = slot % a.n_tiles;
    const int m_tile = (slot / a.n_tiles) * 8 + xcd;
    if (m_tile >= a.m_tiles) return;
    const int n0 = n_tile * BN, row_base = m_tile * a.mrows;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lh = lane >> 5;
    // staging: wave w fills pieces 2w, 2w+1 of the A tile and of the B tile; lane -> (row = lane >> 2, slot = lane & 3),
    // and fetches the quad that belongs into that slot after the swizzle
    const int srow = lane >> 2, sq = (lane & 3) ^ ((lane >> 4) & 3);
    unsigned a_off[2], b_off[2];                       // byte offsets from the tile's first row (tiles span < 2^31 bytes)
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int r = (2 * wave + p) * 16 + srow;
        // (walked tiles, zrows > 0, are launched only when every tile is full: M % (128 nb) == 0)
        a_off[p] = (unsigned)((row_base + r < a.M ? r : 0) * a.lda + 4 * sq) * 4u;
        b_off[p] = (unsigned)(r * a.K + 4 * sq) * 4u;  // rows up to Npad exist (zero rows past N)
    }
    const float* xt = a.x + (int64_t)row_base * a.lda; // uniform
    const float* wt = a.w + (int64_t)n0 * a.K;
    const int kchunks = a.K / BKT, total = kchunks * a.nb;
    // fragment reads: row = 64 wm + 32 i + li, quad (lh + 2 g) ^ ((li >> 2) & 3)
    const int fsw = (li >> 2) & 3;
    const int a_fr0 = (wm * 64 + li) * BKT + 4 * (lh ^ fsw);
    const int a_fr1 = (wm * 64 + li) * BKT + 4 * ((lh + 2) ^ fsw);
    const int b_fr0 = BM * BKT + (wn * 64 + li) * BKT + 4 * (lh ^ fsw);
    const int b_fr1 = BM * BKT + (wn * 64 + li) * BKT + 4 * ((lh + 2) ^ fsw);
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    int lz = 0, lk = 0;                                // (problem, chunk) of the next load, advanced incrementally
#define WG_LOAD(BUF)                                                                                            \
    do {                                                                                                        \
        const char* xs_ = reinterpret_cast<const char*>(xt + (int64_t)lz * a.xb + lk * BKT);                     \
        const char* ws_ = reinterpret_cast<const char*>(wt + (int64_t)lz * a.wb + lk * BKT);                     \
        float* st_ = smem_w + (BUF) * STAGE + (2 * wave) * 256;                                                 \
        _Pragma("unroll") for (int p_ = 0; p_ < 2; ++p_) {                                                      \
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(xs_ + a_off[p_]), (lds_ptr_t)(st_ + p_ * 256), 16, 0, 0);            \
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(ws_ + b_off[p_]), (lds_ptr_t)(st_ + BM * BKT + p_ * 256), 16, 0, 0); \
        }                                                                                                       \
        if (lk + 1 < kchunks) ++lk; else if (lz + 1 < a.nb) { lk = 0; ++lz; }   /* past the end: the last chunk again */ \
    } while (0)
    const unsigned y_off = (unsigned)((wm * 64 + 4 * lh) * a.ldy + wn * 64 + li) * 4u;
    float bias_v[2] = {0.f, 0.f};
    if (EPI) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int c = n0 + wn * 64 + j * 32 + li;
            bias_v[j] = (a.bias && c < a.N) ? a.bias[c] : 0.0f;
        }
    }
    WG_LOAD(0);
    __syncthreads();                                   // (hipcc drains the DMA with vmcnt(0) before the barrier)
    int cz = 0, ck = 0;                                // (problem, chunk) being multiplied
    for (int kc = 0; kc < total; ++kc) {
        const int buf = kc & 1;
        WG_LOAD(buf ^ 1);                              // everyone left that buffer at the barrier of the last iteration
        __builtin_amdgcn_sched_barrier(0);
        {
            const float* sb = smem_w + buf * STAGE;
            f32x4 af[2][2], bf[2][2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                af[0][i] = *reinterpret_cast<const f32x4*>(sb + a_fr0 + i * 32 * BKT);
                bf[0][i] = *reinterpret_cast<const f32x4*>(sb + b_fr0 + i * 32 * BKT);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                af[1][i] = *reinterpret_cast<const f32x4*>(sb + a_fr1 + i * 32 * BKT);
                bf[1][i] = *reinterpret_cast<const f32x4*>(sb + b_fr1 + i * 32 * BKT);
            }
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[g][i].x, bf[g][j].x, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[g][i].y, bf[g][j].y, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[g][i].z, bf[g][j].z, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[g][i].w, bf[g][j].w, acc[i][j], 0, 0, 0);
                    }
        }
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        if (++ck == kchunks) {                          // problem cz is complete: store, clear
            const int zrow = row_base + cz * a.zrows;
            const int64_t yo = (int64_t)cz * a.yb + (int64_t)row_base * a.ldy + n0;                           // uniform
            char* yt = reinterpret_cast<char*>(a.y + yo);
            const char* rt = a.res ? reinterpret_cast<const char*>(a.res + yo) : nullptr;
            const bool interior = zrow + BM <= a.M && n0 + BN <= a.N;
#define WG_VALUE(J)                                                                                             \
            float v_ = acc[i][J][r];                                                                            \
            if (EPI) {                                                                                          \
                v_ += bias_v[J];                                                                                \
                if (a.res) v_ += reinterpret_cast<const float*>(rt + (size_t)ro * a.ldy * 4 + y_off)[(J) * 32]; \
                if (a.act == 1) v_ = be::smish(v_); else if (a.act == 2) v_ = fmaxf(v_, 0.0f);                  \
            }
            if (interior) {                             // no per-element bounds checks (they cost 10 instructions a store)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int ro = i * 32 + (r & 3) + 8 * (r >> 2);
                        float* yr = reinterpret_cast<float*>(yt + (size_t)ro * a.ldy * 4 + y_off);
                        { WG_VALUE(0) yr[0] = v_; }
                        { WG_VALUE(1) yr[32] = v_; }
                    }
            } else {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const bool c_ok = n0 + wn * 64 + j * 32 + li < a.N;
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int ro = i * 32 + (r & 3) + 8 * (r >> 2);
                            if (c_ok && zrow + wm * 64 + 4 * lh + ro < a.M) {
                                WG_VALUE(j)
                                reinterpret_cast<float*>(yt + (size_t)ro * a.ldy * 4 + y_off)[j * 32] = v_;
                            }
                        }
                }
            }
#undef WG_VALUE
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
            ck = 0; ++cz;
        }
    }
#undef WG_LOAD
}

// ---- weight-stationary form of the same GEMMs (large batches, K = 96 / 256 / 384) ------------------------------------------
// A workgroup owns ONE (problem z, N tile), keeps that B tile [128 cols][K] in REGISTERS (K fragment registers per lane: 384
// at K = 384, so one wave per SIMD) and walks a range of M tiles streaming only A: half the DMA pieces and half the fragment
// reads per MFMA of k_wino_gemm.  The issue model of DESIGN 3.1d prices the loop at 2048 / (2048 + 64 + 120 + 60) = 89 %; with
// one wave per SIMD every latency has to be hidden inside the wave: three LDS buffers for A, DMA two chunks ahead, the
// fragments of chunk s+1 read while the MFMAs of chunk s run, one raw barrier per chunk, explicit vmcnt waits (the wait for a tile's
// successor DMA sits in FRONT of that tile's stores, so no count depends on the number of store instructions).  Same order of operations per output element as k_wino_gemm:
// bit-identical.  Measured (tools/wino_gemm_lab.hip, weight_stationary_run13.log): +3...5 % over k_wino_gemm.
// KIND (0: the 25 problems of a Winograd layer, 1: a plain row GEMM - the blocks' 1x1 downsamples) changes nothing in the code: it
// gives the two uses different kernel symbols, so that a profiler's per-kernel averages do not mix launches of different sizes
template <int KCH, int KIND>
__global__ __launch_bounds__(256, 1)
void k_wino_gemm_ws(GemmArgs a, int mgroups) {
    constexpr int BM = 128, BKT = 16, ABUF = BM * BKT;     // floats per A stage (8 KB)
    extern __shared__ __attribute__((aligned(16))) float smem_w[];
    // workgroup -> (problem z, M range, N tile): the N tiles of one (z, M range) are consecutive slots of ONE XCD (ids congruent
    // mod 8 share an XCD), so they run side by side and the A rows they all stream come from HBM once and from that L2 after
    const int bid = blockIdx.x;
    const int xcd = bid & 7, slot = bid >> 3;
    const int n_tile = slot % a.n_tiles;
    const int unit = (slot / a.n_tiles) * 8 + xcd;
    const int z = unit / mgroups, mg = unit % mgroups;
    const int tiles_per = (a.m_tiles + mgroups - 1) / mgroups;
    const int t0 = mg * tiles_per, t1 = min(a.m_tiles, t0 + tiles_per);
    if (z >= a.nb || t0 >= t1) return;
    const int n0 = n_tile * 128;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lh = lane >> 5;
    const int srow = lane >> 2, sq = (lane & 3) ^ ((lane >> 4) & 3);
    const int fsw = (li >> 2) & 3;
    const int fr0 = li * BKT + 4 * (lh ^ fsw), fr1 = li * BKT + 4 * ((lh + 2) ^ fsw);
    // ---- B tile -> registers through LDS, eight K chunks (64 KB) per pass: all the DMAs of a pass are in flight together, ONE
    //      wait per pass (chunk by chunk the fill was 24 dependent DMA round trips: ~36 us of a ~530 us workgroup)
    f32x4 breg[KCH][2][2];
    {
        constexpr int PASS = 8;
        const float* wt = a.w + (int64_t)z * a.wb + (int64_t)n0 * a.K;
        unsigned b_off[2];
#pragma unroll
        for (int p = 0; p < 2; ++p) b_off[p] = (unsigned)(((2 * wave + p) * 16 + srow) * a.K + 4 * sq) * 4u;
#pragma unroll
        for (int c0 = 0; c0 < KCH; c0 += PASS) {
#pragma unroll
            for (int cc = 0; cc < PASS; ++cc)
                if (c0 + cc < KCH) {
#pragma unroll
                    for (int p = 0; p < 2; ++p)
                        __builtin_amdgcn_global_load_lds((glb_ptr_t)(reinterpret_cast<const char*>(wt + (c0 + cc) * BKT) + b_off[p]),
                                                         (lds_ptr_t)(smem_w + cc * ABUF + (2 * wave + p) * 256), 16, 0, 0);
                }
            __syncthreads();
#pragma unroll
            for (int cc = 0; cc < PASS; ++cc)
                if (c0 + cc < KCH) {
                    const float* st = smem_w + cc * ABUF;
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        breg[c0 + cc][0][j] = *reinterpret_cast<const f32x4*>(st + (wn * 64 + j * 32) * BKT + fr0);
                        breg[c0 + cc][1][j] = *reinterpret_cast<const f32x4*>(st + (wn * 64 + j * 32) * BKT + fr1);
                    }
                }
            __syncthreads();                               // everyone has its fragments before the next pass / the A ring overwrites
        }
    }
    // ---- stream A over the M tiles [t0, t1): flat step s = (tile, chunk); every tile is full (the launcher checks M % 128 == 0)
    unsigned a_off[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) a_off[p] = (unsigned)(((2 * wave + p) * 16 + srow) * a.lda + 4 * sq) * 4u;
    const float* xz = a.x + (int64_t)z * a.xb;
    int l_tile = t0, l_c = 0, l_buf = 0;                   // next DMA: (tile, chunk) into ring slot l_buf
#define WS_DMA()                                                                                                \
    do {                                                                                                        \
        const int lt_ = l_tile < t1 ? l_tile : t1 - 1;                       /* past the end: a harmless re-read */  \
        const char* xs_ = reinterpret_cast<const char*>(xz + (int64_t)lt_ * BM * a.lda + l_c * BKT);            \
        float* st_ = smem_w + l_buf * ABUF + (2 * wave) * 256;                                                  \
        _Pragma("unroll") for (int p_ = 0; p_ < 2; ++p_)                                                        \
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(xs_ + a_off[p_]), (lds_ptr_t)(st_ + p_ * 256), 16, 0, 0); \
        if (++l_c == KCH) { l_c = 0; ++l_tile; }                                                                \
        l_buf = l_buf == 2 ? 0 : l_buf + 1;                                                                     \
    } while (0)
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    f32x4 fa[2][2][2];                                     // [set][g][i]: the fragments of step s and of step s + 1
    WS_DMA();                                              // step 0
    WS_DMA();                                              // step 1
    __builtin_amdgcn_s_waitcnt(0x0F72);                    // vmcnt(2): step 0 has landed
    __builtin_amdgcn_s_barrier();
    int r_buf = 0;                                         // ring slot whose fragments are read next
#define WS_READ(SET)                                                                                            \
    do {                                                                                                        \
        const float* sb_ = smem_w + r_buf * ABUF + (wm * 64) * BKT;                                             \
        _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {                                                      \
            fa[SET][0][i_] = *reinterpret_cast<const f32x4*>(sb_ + i_ * 32 * BKT + fr0);                        \
            fa[SET][1][i_] = *reinterpret_cast<const f32x4*>(sb_ + i_ * 32 * BKT + fr1);                        \
        }                                                                                                       \
        r_buf = r_buf == 2 ? 0 : r_buf + 1;                                                                     \
    } while (0)
    WS_READ(0);
    const unsigned y_off = (unsigned)((wm * 64 + 4 * lh) * a.ldy + wn * 64 + li) * 4u;
    for (int t = t0; t < t1; ++t) {
#pragma unroll
        for (int c = 0; c < KCH; ++c) {
            // the DMA of the NEXT step (issued one step ago) has to land before its fragments are read below: vmcnt(0) - it is
            // the only vector-memory operation in flight here.  After a finished tile that wait has already happened, in front
            // of the tile's stores (below), so nothing here depends on how many store instructions the compiler emitted (round 1
            // waited "all but the newest 63" behind the 64 stores: right only for exactly that store count)
            if (!(c == 0 && t != t0)) __builtin_amdgcn_s_waitcnt(0x0F70);
            __builtin_amdgcn_s_barrier();                  // every wave's pieces of step + 1 are in LDS; slot (step + 2) % 3 is free
            WS_DMA();                                      // step + 2
            __builtin_amdgcn_sched_barrier(0);
            WS_READ((c + 1) & 1);                          // fragments of step + 1 (KCH is even: the parity is static)
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c & 1][g][i].x, breg[c][g][j].x, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c & 1][g][i].y, breg[c][g][j].y, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c & 1][g][i].z, breg[c][g][j].z, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c & 1][g][i].w, breg[c][g][j].w, acc[i][j], 0, 0, 0);
                    }
            __builtin_amdgcn_sched_barrier(0);
        }
        // the first step of the next tile needs its DMA (issued at the top of the step that just ended, a whole chunk of MFMAs
        // ago) in LDS: wait for it HERE, while it is still the only thing in flight, then issue the tile's stores
        __builtin_amdgcn_s_waitcnt(0x0F70);
        __builtin_amdgcn_sched_barrier(0);
        char* yt = reinterpret_cast<char*>(a.y + (int64_t)z * a.yb + (int64_t)t * BM * a.ldy + n0);   // uniform; full tile, N % 128 == 0
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ro = i * 32 + (r & 3) + 8 * (r >> 2);
                float* yr = reinterpret_cast<float*>(yt + (size_t)ro * a.ldy * 4 + y_off);
                yr[0] = acc[i][0][r];
                yr[32] = acc[i][1][r];
                acc[i][0][r] = 0.0f; acc[i][1][r] = 0.0f;
            }
    }
#undef WS_DMA
#undef WS_READ
}

// kid = BE_KERNEL_WINO_GEMM: the 25 problems of a Winograd layer (n patches -> 4 n rows each);  BE_KERNEL_GEMM_ROWS: one plain GEMM
// of n rows
template <int KCH, int KIND = 0>
int launch_ws(const GemmArgs& g, hipStream_t s, int64_t n, int cin, int cout, int kid = BE_KERNEL_WINO_GEMM) {
    constexpr size_t lds = (size_t)8 * 128 * 16 * sizeof(float);     // 64 KB: eight chunks of the B fill (the A ring uses three)
    static be::DeviceFlags attr_set{};                      // dynamic-LDS cap raised once per device (thread-safe)
    if (int rc_ = be::ensure_dynamic_lds(reinterpret_cast<const void*>(&k_wino_gemm_ws<KCH, KIND>), lds, attr_set)) return rc_;
    const int cus = be::device_cu_count();
    // ONE workgroup per CU at a time (the kernel takes the whole register file of a CU).  Workgroup ids go round-robin over the 8
    // XCDs, so what has to fit is per XCD: ceil(units / 8) * n_tiles workgroups in rounds of (CUs per XCD).  mgroups = M ranges per
    // problem: the launch lasts rounds x (tiles per range + the register fill, ~0.5 of a tile's time).  Searched, not guessed
    // (round 4): with 40 problems x 128 row tiles and N = 256 the old rule (first round count that gives >= 8 ranges) chose 9 ranges =
    // 3 rounds of 15 tiles where 16 ranges are exactly 5 rounds of 8 (ideal: 40 tile times per CU; 45 -> 41).
    const int per_xcd = cus / 8 > 0 ? cus / 8 : 1;
    int mgroups = 1;
    {
        const int mg_max = g.m_tiles / 4 > 0 ? g.m_tiles / 4 : 1;                   // >= 4 tiles per register fill
        double best = 1e30;
        for (int mg = 1; mg <= mg_max; ++mg) {
            const int tiles_per = (g.m_tiles + mg - 1) / mg;
            const int ranges = (g.m_tiles + tiles_per - 1) / tiles_per;             // ranges that really have tiles
            if (ranges != mg) continue;
            const int wg_xcd = ((g.nb * mg + 7) / 8) * g.n_tiles;
            const int rounds = (wg_xcd + per_xcd - 1) / per_xcd;
            const double cost = rounds * (tiles_per + 0.5);
            if (cost < best - 1e-9) { best = cost; mgroups = mg; }
        }
    }
    const int units = g.nb * mgroups;
    const unsigned grid = (unsigned)(8 * ((units + 7) / 8) * g.n_tiles);
    {
        const double rows = kid == BE_KERNEL_WINO_GEMM ? (double)TPI * n : (double)n, probs = g.nb;
        be::ProfileScope prof(s, kid, probs * 2.0 * rows * cin * cout,
                              probs * 4.0 * (rows * cin + (double)cin * cout + rows * cout),
                              probs * 2.0 * g.m_tiles * g.n_tiles * 128.0 * 128.0 * cin);
        hipLaunchKernelGGL((k_wino_gemm_ws<KCH, KIND>), dim3(grid), dim3(256), lds, s, g, mgroups);
    }
    return be::check_launch("be_wino_conv3x3_6x6_f32(gemm, weight-stationary)");
}

}  // namespace

extern "C" int be_wino_tile_rows(void) { return TH; }

extern "C" size_t be_wino_packed_floats(int cout, int cin) {
    if (cout <= 0 || cin <= 0 || cin % 32) return 0;
    return (size_t)NPOS * ((cout + 31) / 32 * 32) * cin;
}

extern "C" int be_wino_pack_f32(const float* w, const float* b, const float* gamma, const float* beta, const float* mean,
                                const float* var, float eps, int cout, int cin, float* packed_w, float* packed_bias,
                                void* stream) {
    BE_REQUIRE(w && packed_w && packed_bias, "be_wino_pack_f32: null pointer");
    BE_REQUIRE(cout > 0 && cin > 0 && cin % 32 == 0, "be_wino_pack_f32: cin must be a multiple of 32 (got %d)", cin);
    BE_REQUIRE((gamma == nullptr) == (beta == nullptr) && (gamma == nullptr) == (mean == nullptr) && (gamma == nullptr) == (var == nullptr),
               "be_wino_pack_f32: BatchNorm tensors must be all set or all null");
    const int cp = (cout + 31) / 32 * 32;
    hipLaunchKernelGGL(k_wino_pack, dim3(grid_cap((int64_t)cp * cin, 256, 4096)), dim3(256), 0, be::as_stream(stream), w, b, gamma,
                       beta, mean, var, eps, cout, cin, cp, packed_w, packed_bias);
    return be::check_launch("be_wino_pack_f32");
}

extern "C" size_t be_wino_workspace_floats(int64_t n, int cin, int cout) {
    if (n <= 0) return 0;
    return (size_t)100 * n * ((size_t)cin + cout);              // room for V + M of either tile shape (NPOS x TPI = 100 or 80 values per channel)
}

namespace {

// large batches take k_wino_gemm and the tile-major buffers, small ones the batched k_conv_igemm launch and plane-major buffers
bool wino_large(int64_t n, int cout) {
    static const bool no_persist = getenv("BE_WINO_NO_PERSIST") != nullptr;        // A/B knob
    return ((cout + 31) / 32 * 32) % 128 == 0 && n >= 1024 && !no_persist && (int64_t)NPOS * TPI * n * (int64_t)cout < ((int64_t)1 << 31);
}

int wino_gemms(const float* V, const float* packed_w, float* M, int64_t n, int cin, int cout, hipStream_t s, void* stream) {
    const int cp = (cout + 31) / 32 * 32;
    if (wino_large(n, cout)) {
        // large batches: one workgroup per (M tile, N tile) walks the 25 problems back to back
        constexpr size_t lds = (size_t)2 * (128 + 128) * 16 * sizeof(float);
        static be::DeviceFlags attr_set{};                      // dynamic-LDS cap raised once per device (thread-safe)
        if (int rc_ = be::ensure_dynamic_lds(reinterpret_cast<const void*>(&k_wino_gemm<0>), lds, attr_set)) return rc_;
        // tile-major V [4n][25][cin] and M [4n][25][cout]: problem z = column block z of a row
        const int64_t rows = (int64_t)TPI * n;
        GemmArgs g{V, packed_w, M, (int)rows, cin, cout, NPOS * cout, NPOS, (int)((rows + 127) / 128), cp / 128,
                   (int64_t)cin, (int64_t)cp * cin, (int64_t)cout, NPOS * cin, 128, 0, nullptr, nullptr, 0};
        // weight-stationary form: full tiles only, enough M tiles to amortise the register fill, cout a multiple of 128
        static const bool no_ws = getenv("BE_WINO_NO_WS") != nullptr;               // A/B knob
        // (>= 64 row tiles: a 4096-patch half of the two-stream schedule has 2 x 4096 rows per position with the 8x5 tiles)
        static const int ws_min_tiles = getenv("BE_WINO_WS_MIN_TILES") ? atoi(getenv("BE_WINO_WS_MIN_TILES")) : (TPI == 2 ? 64 : 128);   // A/B knob
        if (!no_ws && rows % 128 == 0 && g.m_tiles >= ws_min_tiles && cout % 128 == 0) {
            if (cin == 96) return launch_ws<6>(g, s, n, cin, cout);
            if (cin == 256) return launch_ws<16>(g, s, n, cin, cout);
            if (cin == 384) return launch_ws<24>(g, s, n, cin, cout);
        }
        const unsigned grid = (unsigned)(8 * ((g.m_tiles + 7) / 8) * g.n_tiles);
        {
            be::ProfileScope prof(s, BE_KERNEL_WINO_GEMM, (double)NPOS * 2.0 * rows * cin * cout,
                                  (double)NPOS * 4.0 * ((double)rows * cin + (double)cin * cout + (double)rows * cout),
                                  (double)NPOS * 2.0 * g.m_tiles * g.n_tiles * 128.0 * 128.0 * cin);
            hipLaunchKernelGGL(k_wino_gemm<0>, dim3(grid), dim3(256), lds, s, g);
        }
        return be::check_launch("be_wino_conv3x3_6x6_f32(gemm)");
    }
    be_conv_desc d;
    d.n = (int)(TPI * n); d.h = 1; d.w = 1; d.cin = cin; d.cout = cout; d.ksize = 1; d.act = 0;
    return be_conv_nhwc_batched_f32(&d, V, packed_w, nullptr, M, cout, NPOS, (int64_t)TPI * n * cin, (int64_t)cp * cin,
                                    (int64_t)TPI * n * cout, stream);
}

int wino_args_ok(const char* who, int64_t n, int cin, int cout) {
    BE_REQUIRE(n > 0 && 4 * n < ((int64_t)1 << 31) / 128, "%s: batch out of range", who);
    BE_REQUIRE(cin % 32 == 0 && cout % 4 == 0 && cin > 0 && cout > 0, "%s: cin %% 32, cout %% 4 required", who);
    return BE_OK;
}

}  // namespace

// 1x1 convolutions and linears of large batches through the same kernel: y[M][ldy] = act(x[M][K] w[Npad][K]^T + bias (+ res)).
// The caller (conv_dispatch in be_conv.hip) has checked: K % 16 == 0, Npad % 128 == 0, 16-byte aligned x / w, M >= 4096.
// Same order of operations per output element as k_conv_igemm (K ascending in chunks of 16, fp32 MFMA): bit-identical.
int be::gemm_rows(const float* x, int64_t M, int K, const float* packed_w, int N, const float* bias, const float* res, int act,
                  float* y, int ldy, void* stream) {
    hipStream_t s = be::as_stream(stream);
    constexpr size_t lds = (size_t)2 * (128 + 128) * 16 * sizeof(float);
    static be::DeviceFlags attr_set{};                      // dynamic-LDS cap raised once per device (thread-safe)
    if (int rc_ = be::ensure_dynamic_lds(reinterpret_cast<const void*>(&k_wino_gemm<1>), lds, attr_set)) return rc_;
    const int cp = (N + 127) / 128 * 128, n_tiles = cp / 128;
    const int64_t tiles = (M + 127) / 128;
    // short K loops: one workgroup walks nb consecutive row tiles (only when all of them are full), about one round of
    // the 768 resident workgroups
    int nb = 1;
    if (M % 128 == 0)
        for (int c = 2; c <= 16; ++c)
            if (tiles % c == 0 && (tiles / c) * n_tiles >= 768) nb = c;
    GemmArgs g{x, packed_w, y, (int)M, K, N, ldy, nb, (int)(tiles / nb), n_tiles, (int64_t)128 * K, 0, (int64_t)128 * ldy,
               K, 128 * nb, nb > 1 ? 128 : 0, bias, res, act};
    const unsigned grid = (unsigned)(8 * ((g.m_tiles + 7) / 8) * g.n_tiles);
    {
        be::ProfileScope prof(s, BE_KERNEL_GEMM_ROWS, 2.0 * M * K * N, 4.0 * ((double)M * K + (double)K * N + (double)M * N),
                              2.0 * tiles * n_tiles * 128.0 * 128.0 * K);
        hipLaunchKernelGGL(k_wino_gemm<1>, dim3(grid), dim3(256), lds, s, g);
    }
    return be::check_launch("be_conv_nhwc_f32(gemm rows)");
}

// a plain row GEMM on the weight-stationary kernel: ONE "problem" (z = 0) of M / 128 row tiles
int be::gemm_rows_ws(const float* x, int64_t M, int K, const float* packed_w, int N, float* y, int ldy, void* stream) {
    static const bool off = getenv("BE_NO_ROWS_WS") != nullptr || getenv("BE_WINO_NO_WS") != nullptr;      // A/B knobs
    if (off || M % 128 || M / 128 < 128 || N % 128 || ldy % 4 || M * (int64_t)ldy >= ((int64_t)1 << 31)) return 1;
    if (!(K == 96 || K == 256 || K == 384) || !be::aligned16(x) || !be::aligned16(packed_w) || !be::aligned16(y)) return 1;
    hipStream_t s = be::as_stream(stream);
    GemmArgs g{x, packed_w, y, (int)M, K, N, ldy, 1, (int)(M / 128), N / 128, 0, 0, 0, K, 128, 0, nullptr, nullptr, 0};
    // (n, cin, cout of the profile record: 2 M K N FLOPs = 25 x 2 x 4 n' x cin x cout with n' = M / 100)
    if (K == 96) return launch_ws<6, 1>(g, s, M, K, N, BE_KERNEL_GEMM_ROWS);
    if (K == 256) return launch_ws<16, 1>(g, s, M, K, N, BE_KERNEL_GEMM_ROWS);
    return launch_ws<24, 1>(g, s, M, K, N, BE_KERNEL_GEMM_ROWS);
}

// The transform kernels read every byte once: non-temporal loads (default) move 5.42-5.46 TB/s where plain loads move 5.24-5.28
// (2.65 -> 2.56 ms of transforms per step; stores: no effect; the max-pool, whose windows overlap, loses a third with them).
// A/B knob: BE_WINO_NT = 0 plain | 1 non-temporal loads (default) | 2 non-temporal stores | 3 both
static int wino_nt() { static const int v = getenv("BE_WINO_NT") ? atoi(getenv("BE_WINO_NT")) & 3 : 1; return v; }
#define BE_WINO_LAUNCH(K, GRID, BLOCK, LDS, STREAM, ...)                                                        \
    do {                                                                                                        \
        switch (wino_nt()) {                                                                                    \
            case 1: hipLaunchKernelGGL(K<1>, GRID, BLOCK, LDS, STREAM, __VA_ARGS__); break;                     \
            case 2: hipLaunchKernelGGL(K<2>, GRID, BLOCK, LDS, STREAM, __VA_ARGS__); break;                     \
            case 3: hipLaunchKernelGGL(K<3>, GRID, BLOCK, LDS, STREAM, __VA_ARGS__); break;                     \
            default: hipLaunchKernelGGL(K<0>, GRID, BLOCK, LDS, STREAM, __VA_ARGS__); break;                    \
        }                                                                                                       \
    } while (0)

extern "C" int be_wino_conv3x3_6x6_f32(const float* x, const float* packed_w, const float* packed_bias, const float* residual,
                                       float* y, int64_t n, int cin, int cout, int act, float* workspace,
                                       size_t workspace_floats, void* stream) {
    BE_REQUIRE(x && packed_w && packed_bias && y && workspace, "be_wino_conv3x3_6x6_f32: null pointer");
    if (int rc = wino_args_ok("be_wino_conv3x3_6x6_f32", n, cin, cout)) return rc;
    BE_REQUIRE(workspace_floats >= be_wino_workspace_floats(n, cin, cout), "be_wino_conv3x3_6x6_f32: workspace too small");
    BE_REQUIRE(be::aligned16(x) && be::aligned16(y) && be::aligned16(workspace) && be::aligned16(packed_w),
               "be_wino_conv3x3_6x6_f32: 16-byte alignment");
    hipStream_t s = be::as_stream(stream);
    float* V = workspace;
    float* M = workspace + (size_t)100 * n * cin;
    const int tm = wino_large(n, cout);
    BE_WINO_LAUNCH(k_wino_in, dim3(grid_cap(n * TPI * (cin / 4), 256)), dim3(256), 0, s, x, V, n, cin / 4, tm);
    if (int rc = be::check_launch("be_wino_conv3x3_6x6_f32(in)")) return rc;
    if (int rc = wino_gemms(V, packed_w, M, n, cin, cout, s, stream)) return rc;
    BE_WINO_LAUNCH(k_wino_out, dim3(grid_cap(n * TPI * (cout / VW_OUT), 256)), dim3(256), 0, s, M, packed_bias, residual, y, n, cout / VW_OUT,
                       act, tm);
    return be::check_launch("be_wino_conv3x3_6x6_f32(out)");
}

extern "C" size_t be_wino_pair_workspace_floats(int64_t n, int cin, int cmid, int cout) {
    if (n <= 0) return 0;
    const size_t big = (size_t)(cin > cmid ? cin : cmid), out = (size_t)(cmid > cout ? cmid : cout);
    return (size_t)100 * n * (big + out);                       // V (cin, then cmid) + M (cmid, then cout)
}

// pool2 = 1: y is [n,3,3,cout], the 2x2 max-pool of the block's output (k_wino_out_pool2)
int be::wino_pair(const float* x, const float* packed_w1, const float* packed_bias1, int act1, const float* packed_w2,
                  const float* packed_bias2, const float* residual, int act2, float* y, int64_t n, int cin, int cmid, int cout,
                  float* workspace, size_t workspace_floats, void* stream, int pool2) {
    BE_REQUIRE(x && packed_w1 && packed_bias1 && packed_w2 && packed_bias2 && y && workspace,
               "be_wino_conv3x3_pair_6x6_f32: null pointer");
    if (int rc = wino_args_ok("be_wino_conv3x3_pair_6x6_f32", n, cin, cmid)) return rc;
    if (int rc = wino_args_ok("be_wino_conv3x3_pair_6x6_f32", n, cmid, cout)) return rc;
    BE_REQUIRE(workspace_floats >= be_wino_pair_workspace_floats(n, cin, cmid, cout), "be_wino_conv3x3_pair_6x6_f32: workspace too small");
    BE_REQUIRE(be::aligned16(x) && be::aligned16(y) && be::aligned16(workspace) && be::aligned16(packed_w1) && be::aligned16(packed_w2),
               "be_wino_conv3x3_pair_6x6_f32: 16-byte alignment");
    hipStream_t s = be::as_stream(stream);
    const size_t big = (size_t)(cin > cmid ? cin : cmid);
    float* V = workspace;
    float* M = workspace + (size_t)100 * n * big;
    const int tm1 = wino_large(n, cmid), tm2 = wino_large(n, cout);
    {
        be::ProfileScope prof(s, BE_KERNEL_WINO_TRANSFORM, 0.0, 4.0 * n * cin * (36.0 + (double)(NPOS * TPI)), 0.0);
        BE_WINO_LAUNCH(k_wino_in, dim3(grid_cap(n * TPI * (cin / 4), 256)), dim3(256), 0, s, x, V, n, cin / 4, tm1);
    }
    if (int rc = be::check_launch("be_wino_conv3x3_pair_6x6_f32(in)")) return rc;
    if (int rc = wino_gemms(V, packed_w1, M, n, cin, cmid, s, stream)) return rc;
    {
        be::ProfileScope prof(s, BE_KERNEL_WINO_TRANSFORM, 0.0, 4.0 * n * cmid * (2.0 * NPOS * TPI), 0.0);
        BE_WINO_LAUNCH(k_wino_out_in, dim3(grid_cap(n * (cmid / VW_OUT), 256)), dim3(256), 0, s, M, packed_bias1, V, n, cmid / VW_OUT, act1,
                           tm1, tm2);
    }
    if (int rc = be::check_launch("be_wino_conv3x3_pair_6x6_f32(out_in)")) return rc;
    if (int rc = wino_gemms(V, packed_w2, M, n, cmid, cout, s, stream)) return rc;
    {
        be::ProfileScope prof(s, BE_KERNEL_WINO_TRANSFORM, 0.0, 4.0 * n * cout * ((double)(NPOS * TPI) + (residual ? 36.0 : 0.0) + (pool2 ? 9.0 : 36.0)), 0.0);
        if (pool2)
            BE_WINO_LAUNCH(k_wino_out_pool2, dim3(grid_cap(n * (cout / VW_OUT), 256)), dim3(256), 0, s, M, packed_bias2, residual, y, n,
                               cout / VW_OUT, act2, tm2);
        else
            BE_WINO_LAUNCH(k_wino_out, dim3(grid_cap(n * TPI * (cout / VW_OUT), 256)), dim3(256), 0, s, M, packed_bias2, residual, y, n,
                               cout / VW_OUT, act2, tm2);
    }
    return be::check_launch("be_wino_conv3x3_pair_6x6_f32(out)");
}

extern "C" int be_wino_conv3x3_pair_6x6_f32(const float* x, const float* packed_w1, const float* packed_bias1, int act1,
                                            const float* packed_w2, const float* packed_bias2, const float* residual, int act2,
                                            float* y, int64_t n, int cin, int cmid, int cout, float* workspace,
                                            size_t workspace_floats, void* stream) {
    return be::wino_pair(x, packed_w1, packed_bias1, act1, packed_w2, packed_bias2, residual, act2, y, n, cin, cmid, cout, workspace,
                         workspace_floats, stream, 0);
}
