// Pixel-major implicit-GEMM convolution for LARGE batches of small images, operands by LDS-DMA (gfx950).
//
// Same arithmetic as k_conv_igemm (be_conv.hip; models/local_stage.py:11-17,34-37: Conv2d + folded BatchNorm + Smish,
// and the residual block's 1x1 downsample appended to conv2's K loop): D[image][Cout] for ONE output pixel of BM
// consecutive images per workgroup, K walked as (32-channel chunk outer, tap inner, 16-float half inner), fp32 MFMA
// 32x32x2, so every output element sees the same chain of fused multiply-adds: results are bit-identical to
// k_conv_igemm (tests/test_hip_parity.py compares the two paths).
//
// What differs is everything around the MFMAs.  Knock-out runs (tools/wino_gemm_lab.hip) showed that the
// matrix pipe loses about as many cycles as the other instructions of the resident waves spend moving registers, whatever
// the occupancy.  A pixel-major tile makes the K walk wave-uniform (every row = the same pixel of another image: the taps
// outside the image are skipped for the whole tile, nothing is ever zero-filled), so here
//   - operands go global -> LDS directly (global_load_lds_dwordx4, 1-KB pieces of 16 rows x 64 B, 16-byte quads
//     XOR-swizzled by (row >> 2) & 3 on the source address and on the fragment read): no staging registers, no
//     ds_write, no per-row bounds test, no per-row 64-bit address arithmetic;
//   - the K walk (chunk -> tap -> offsets) is scalar arithmetic only;
//   - tiles are BM = 256 images (wave tile 64 rows x 32 NT columns): 2 + NT fragment registers feed 2 NT MFMAs per k-step
//     where the 128-row tiles of k_conv_igemm<4,1,1,NT> need 1 + NT for NT;
//   - interior tiles store without per-element bounds checks.
// MODE_ROW8 is conv1 (7x7, Cin 3): it reads a staging of the patch with 4 channels per pixel AND 28 pixels per row (3
// zero pixels left, 4 right: be_nchw3_to_nhwc4p_f32 / be_view_to_nhwc4p_f32), so that the 8-pixel kernel rows never
// leave the row and need no zero-fill either.
#include "be_common.h"
#include "be_device_math.h"
#include <cstdlib>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

enum { PM_TAPS3 = 0, PM_ROW8 = 1 };

struct PmArgs {
    const float* x;       // TAPS3: NHWC [N,H,W,Cin];  ROW8: [N,H,wrow,4] (padded rows)
    const float* x2;      // optional second input [N,H,W,Cin2]: its 1x1 conv is appended to the K loop
    const float* w;       // packed [Cout_pad][Ktot] (be_conv_pack_f32 / be_conv_pack_fused2_f32)
    const float* bias;
    float* y;             // NHWC [N,H,W,ldy]
    int Nimg, H, W, HW, Cin, Cin2, Cout, ldy, Ktot, act, groups, n_tiles, wrow, ncc;
    int64_t istride, istride2;      // floats per image in x / x2
};

template <int MT, int NT, int MODE>
__global__ __launch_bounds__(256, 3)
void k_conv_pm(PmArgs a) {
    constexpr int BM = 128 * MT, BN = 32 * NT, BKT = 16;
    constexpr int STAGE = (BM + BN) * BKT;             // floats per stage
    constexpr int NPA = 2 * MT;                        // A pieces per wave (BM / 16 pieces over 4 waves)
    constexpr int NPB = (BN / 16 + 3) / 4;             // B pieces per wave (the last pass may be partial)
    extern __shared__ __attribute__((aligned(16))) float smem_pm[];

    // ---- workgroup -> (group of BM images, pixel, N tile): a group's pixel tiles stay on one XCD (shared input lines)
    const int bid = blockIdx.x;
    const int xcd = bid & 7, slot = bid >> 3;
    const int n_tile = slot % a.n_tiles;
    const int t = slot / a.n_tiles;
    const int grp = (t / a.HW) * 8 + xcd;
    if (grp >= a.groups) return;
    const int n0 = n_tile * BN, img0 = grp * BM;
    int py, px;
    {   // interior pixels (all taps) first, the border ring (fewer taps) last: the short tiles fill the tail
        const int idx = t % a.HW, ni = (a.H - 2) * (a.W - 2);
        if (idx < ni) { py = 1 + idx / (a.W - 2); px = 1 + idx % (a.W - 2); }
        else {
            const int e = idx - ni;
            if (e < a.W) { py = 0; px = e; }
            else if (e < 2 * a.W) { py = a.H - 1; px = e - a.W; }
            else if (e < 2 * a.W + a.H - 2) { px = 0; py = 1 + e - 2 * a.W; }
            else { px = a.W - 1; py = 1 + e - 2 * a.W - (a.H - 2); }
        }
    }
    // taps (TAPS3: of the 3x3 kernel; ROW8: kernel rows) this pixel has inside the image, 4 bits each
    unsigned long long tap_list = 0;
    int ntap = 0;
#pragma unroll
    for (int k = 0; k < (MODE == PM_TAPS3 ? 9 : 7); ++k) {
        bool ok;
        if (MODE == PM_TAPS3) ok = (unsigned)(py + k / 3 - 1) < (unsigned)a.H && (unsigned)(px + k % 3 - 1) < (unsigned)a.W;
        else ok = (unsigned)(py + k - 3) < (unsigned)a.H;
        if (ok) { tap_list |= (unsigned long long)k << (4 * ntap); ++ntap; }
    }

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int li = lane & 31, lh = lane >> 5;
    // ---- staging roles: lane -> (row = lane >> 2 of the piece, slot = lane & 3), fetching the quad the swizzle puts there
    const int srow = lane >> 2, sq = (lane & 3) ^ ((lane >> 4) & 3);
    unsigned a_off[NPA], a_off2[NPA], b_off[NPB];      // byte offsets from the tile's first row
#pragma unroll
    for (int i = 0; i < NPA; ++i) {
        const int r = (wave + 4 * i) * 16 + srow;
        const int rr = img0 + r < a.Nimg ? r : 0;      // images past the batch: any valid image (never stored)
        a_off[i] = (unsigned)(rr * a.istride + 4 * sq) * 4u;
        a_off2[i] = (unsigned)(rr * a.istride2 + 4 * sq) * 4u;
    }
#pragma unroll
    for (int i = 0; i < NPB; ++i) b_off[i] = (unsigned)(((wave + 4 * i) * 16 + srow) * a.Ktot + 4 * sq) * 4u;
    // uniform bases: the output pixel of the group's first image; the N tile's first weight row
    const float* xpix = a.x + (int64_t)img0 * a.istride + (MODE == PM_TAPS3 ? (py * a.W + px) * a.Cin : (py * a.wrow + px) * 4);
    const float* x2pix = a.x2 ? a.x2 + (int64_t)img0 * a.istride2 + (py * a.W + px) * a.Cin2 : nullptr;
    const float* wt = a.w + (int64_t)n0 * a.Ktot;

    // ---- K walk: main part (cc, tap j, half), then the 1x1 on x2 (16-float chunks k2)
    const int n_main = (MODE == PM_TAPS3 ? a.ncc : 1) * ntap * 2;
    const int total = n_main + (a.x2 ? a.Cin2 / BKT : 0);
    int w_cc = 0, w_j = 0, w_sub = 0, w_k = 0;         // walker state of the NEXT chunk to load
#define PM_LOAD(BUF)                                                                                            \
    do {                                                                                                        \
        float* st_ = smem_pm + (BUF) * STAGE;                                                                   \
        int boff_;                                                                                              \
        if (w_k < n_main) {                                                                                     \
            const int tap_ = (int)((tap_list >> (4 * w_j)) & 15ull);                                            \
            int aoff_;                                                                                          \
            if (MODE == PM_TAPS3) {                                                                             \
                const int ty_ = (tap_ * 11) >> 5, tx_ = tap_ - 3 * ty_;                /* tap / 3, tap % 3 */     \
                aoff_ = ((ty_ - 1) * a.W + tx_ - 1) * a.Cin + w_cc * 32 + w_sub * BKT;                          \
                boff_ = ((w_cc * 9 + tap_) * 2 + w_sub) * BKT;                                                  \
            } else {                                                                                            \
                aoff_ = (tap_ - 3) * a.wrow * 4 + w_sub * BKT;                                                  \
                boff_ = (tap_ * 2 + w_sub) * BKT;                                                               \
            }                                                                                                   \
            const char* xs_ = reinterpret_cast<const char*>(xpix + aoff_);                                      \
            _Pragma("unroll") for (int i_ = 0; i_ < NPA; ++i_)                                                  \
                __builtin_amdgcn_global_load_lds((glb_ptr_t)(xs_ + a_off[i_]), (lds_ptr_t)(st_ + (wave + 4 * i_) * 256), 16, 0, 0); \
            if (++w_sub == 2) { w_sub = 0; if (++w_j == ntap) { w_j = 0; ++w_cc; } }                            \
        } else {                                                                                                \
            const int k2_ = w_k - n_main;                                                                       \
            boff_ = (MODE == PM_TAPS3 ? a.ncc * 9 * 32 : 0) + k2_ * BKT;                                        \
            const char* xs_ = reinterpret_cast<const char*>(x2pix + k2_ * BKT);                                 \
            _Pragma("unroll") for (int i_ = 0; i_ < NPA; ++i_)                                                  \
                __builtin_amdgcn_global_load_lds((glb_ptr_t)(xs_ + a_off2[i_]), (lds_ptr_t)(st_ + (wave + 4 * i_) * 256), 16, 0, 0); \
        }                                                                                                       \
        const char* ws_ = reinterpret_cast<const char*>(wt + boff_);                                            \
        _Pragma("unroll") for (int i_ = 0; i_ < NPB; ++i_)                                                      \
            if (BN / 16 % 4 == 0 || wave + 4 * i_ < BN / 16)                                                    \
                __builtin_amdgcn_global_load_lds((glb_ptr_t)(ws_ + b_off[i_]), (lds_ptr_t)(st_ + BM * BKT + (wave + 4 * i_) * 256), 16, 0, 0); \
        if (w_k + 1 < total) ++w_k; else { w_cc = 0; w_j = 0; w_sub = 0; w_k = 0; }   /* past the end: chunk 0 again (harmless) */ \
    } while (0)

    // ---- fragment reads: row = 64 wave + 32 i + li (A) / 32 j + li (B), quad (lh + 2 g) ^ ((li >> 2) & 3)
    const int fsw = (li >> 2) & 3;
    const int fq0 = 4 * (lh ^ fsw), fq1 = 4 * ((lh + 2) ^ fsw);
    const int a_fr = (wave * 32 * MT + li) * BKT, b_fr = BM * BKT + li * BKT;
    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    PM_LOAD(0);
    __syncthreads();                                   // (hipcc drains the DMA with vmcnt(0) before the barrier)
    for (int kc = 0; kc < total; ++kc) {
        const int buf = kc & 1;
        PM_LOAD(buf ^ 1);                              // everyone left that buffer at the barrier of the last iteration
        __builtin_amdgcn_sched_barrier(0);
        {
            const float* sb = smem_pm + buf * STAGE;
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const int fq = g ? fq1 : fq0;
                f32x4 af[MT], bf[NT];
#pragma unroll
                for (int i = 0; i < MT; ++i) af[i] = *reinterpret_cast<const f32x4*>(sb + a_fr + i * 32 * BKT + fq);
#pragma unroll
                for (int j = 0; j < NT; ++j) bf[j] = *reinterpret_cast<const f32x4*>(sb + b_fr + j * 32 * BKT + fq);
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].x, bf[j].x, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].y, bf[j].y, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].z, bf[j].z, acc[i][j], 0, 0, 0);
                        // ROW8 (conv1): a quad is one pixel's four channels and the fourth is zero in the staging AND in the
                        // packed weights - that product adds exactly 0, so it is not issued (K 224 -> 168 executed columns)
                        if (MODE != PM_ROW8) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].w, bf[j].w, acc[i][j], 0, 0, 0);
                    }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
    }
#undef PM_LOAD

    // ---- epilogue: D[row][col], col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5); row = image of the group
    const int64_t rs = (int64_t)a.HW * a.ldy;          // floats between the same pixel of consecutive images
    char* yt = reinterpret_cast<char*>(a.y + ((int64_t)img0 * a.HW + py * a.W + px) * a.ldy + n0);            // uniform
    const unsigned y_off = (unsigned)((wave * 32 * MT + 4 * lh) * rs + li) * 4u;
    const bool interior = img0 + BM <= a.Nimg && n0 + BN <= a.Cout;
    float bias_v[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int c = n0 + j * 32 + li;
        bias_v[j] = (c < a.Cout && a.bias) ? a.bias[c] : 0.0f;
    }
#define PM_STORE(J)                                                                                             \
    do {                                                                                                        \
        float v_ = acc[i][J][r] + bias_v[J];                                                                    \
        if (a.act == 1) v_ = be::smish(v_); else if (a.act == 2) v_ = fmaxf(v_, 0.0f);                          \
        reinterpret_cast<float*>(yt + (size_t)ro * rs * 4 + y_off)[(J) * 32] = v_;                              \
    } while (0)
    if (interior) {                                    // no per-element bounds checks
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ro = i * 32 + (r & 3) + 8 * (r >> 2);
#pragma unroll
                for (int j = 0; j < NT; ++j) PM_STORE(j);
            }
    } else {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const bool c_ok = n0 + j * 32 + li < a.Cout;
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ro = i * 32 + (r & 3) + 8 * (r >> 2);
                    if (c_ok && img0 + wave * 32 * MT + 4 * lh + ro < a.Nimg) PM_STORE(j);
                }
        }
    }
#undef PM_STORE
}

template <int MT, int NT, int MODE>
int launch_pm(const PmArgs& a, hipStream_t s, int kernel_id) {
    constexpr int BM = 128 * MT, BN = 32 * NT;
    constexpr size_t lds = (size_t)2 * (BM + BN) * 16 * sizeof(float);
    static be::DeviceFlags attr_set{};                      // dynamic-LDS cap raised once per device (thread-safe)
    if (int rc_ = be::ensure_dynamic_lds(reinterpret_cast<const void*>(&k_conv_pm<MT, NT, MODE>), lds, attr_set)) return rc_;
    const unsigned grid = (unsigned)(8 * ((a.groups + 7) / 8) * a.HW * a.n_tiles);
    {   // algorithmic work: 2*M*K*Cout with the REAL K; executed: the chunks the tiles visit x 2*BM*BN*16
        const double M = (double)a.Nimg * a.HW;
        const double k_real = MODE == PM_ROW8 ? 147.0 : 9.0 * a.Cin, cin_real = MODE == PM_ROW8 ? 3.0 : (double)a.Cin;
        const double k2 = a.x2 ? (double)a.Cin2 : 0.0;
        double chunks = 0.0;
        for (int py = 0; py < a.H; ++py)
            for (int px = 0; px < a.W; ++px) {
                int nt = 0;
                if (MODE == PM_ROW8) for (int k = 0; k < 7; ++k) nt += (unsigned)(py + k - 3) < (unsigned)a.H;
                else for (int k = 0; k < 9; ++k) nt += (unsigned)(py + k / 3 - 1) < (unsigned)a.H && (unsigned)(px + k % 3 - 1) < (unsigned)a.W;
                chunks += 2.0 * nt * (MODE == PM_ROW8 ? 1 : a.ncc) + (a.x2 ? a.Cin2 / 16 : 0);
            }
        chunks *= (double)a.groups * a.n_tiles;
        be::ProfileScope prof(s, kernel_id, 2.0 * M * (k_real + k2) * a.Cout,
                              4.0 * (M * (cin_real + k2) + (k_real + k2) * a.Cout + M * a.Cout),
                              chunks * 2.0 * BM * BN * (MODE == PM_ROW8 ? 12 : 16));
        hipLaunchKernelGGL((k_conv_pm<MT, NT, MODE>), dim3(grid), dim3(256), lds, s, a);
    }
    return be::check_launch("be_conv_nhwc_f32(pixel-major)");
}

}  // namespace

// Called by conv_dispatch (be_conv.hip) for large batches (n >= 512): 3x3 convolutions (optionally with the fused 1x1 on
// x2) with cout_pad % 96 == 0 or % 64 == 0, and the 7x7 conv1 on the padded staging (wrow = 28).  Returns BE_OK, an error,
// or 1 = "not mine" (the caller falls back to k_conv_igemm).
int be::conv_pm(const be_conv_desc* d, const float* x, int wrow, const float* x2, int cin2, const float* pw, const float* pb,
                float* y, int ldy, int ktot, void* stream) {
    static const bool off = getenv("BE_NO_CONV_PM") != nullptr;                   // A/B knob
    if (off) return 1;
    const int cp = (d->cout + 31) / 32 * 32;
    PmArgs a;
    a.x = x; a.x2 = x2; a.w = pw; a.bias = pb; a.y = y;
    a.Nimg = d->n; a.H = d->h; a.W = d->w; a.HW = d->h * d->w; a.Cin = d->cin; a.Cin2 = cin2; a.Cout = d->cout; a.ldy = ldy;
    a.Ktot = ktot; a.act = d->act; a.wrow = wrow; a.ncc = d->cin / 32;
    a.istride = d->ksize == 7 ? (int64_t)d->h * wrow * 4 : (int64_t)a.HW * d->cin;
    a.istride2 = (int64_t)a.HW * cin2;
    hipStream_t s = be::as_stream(stream);
    // 32-bit lane offsets: 256 images of input / output must span < 2^32 bytes
    const int64_t span = 256 * 4 * (a.istride > a.istride2 ? a.istride : a.istride2);
    if (span >= ((int64_t)1 << 32) || 256 * 4 * (int64_t)a.HW * ldy >= ((int64_t)1 << 32) || d->h < 3 || d->w < 3) return 1;
    if (d->ksize == 7) {
        if (cp != 64 || d->cin != 4 || wrow < d->w + 7) return 1;
        a.groups = (d->n + 255) / 256; a.n_tiles = 1;
        return launch_pm<2, 2, PM_ROW8>(a, s, BE_KERNEL_CONV_ROW8_128x64);
    }
    if (d->ksize != 3 || d->cin % 32 || (x2 && cin2 % 16)) return 1;
    if (cp % 96 == 0) {
        a.groups = (d->n + 255) / 256; a.n_tiles = cp / 96;
        return launch_pm<2, 3, PM_TAPS3>(a, s, BE_KERNEL_CONV_128x96);
    }
    return 1;
}
