// The implicit-GEMM convolution kernel of be_conv.hip as a device function, shared with be_train.hip (see conv_igemm_body).
#pragma once
#include "be_common.h"
#include "be_device_math.h"

namespace be_igemm {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));   // native vector: HIP's float4 class kept staging arrays in scratch

constexpr int BK = 32;
enum { MODE_TAPS = 0, MODE_ROW8 = 1 };

struct ConvArgs {
    const float* x;       // NHWC activations [N,H,W,Cin]
    const float* w;       // packed [Cout_pad][Ktot]
    const float* bias;    // [Cout_pad]
    const float* res;     // optional residual, same layout/stride as y
    float* y;
    int M, H, W, HW, Cin, Cout, ldy, ks, nchunk, Ktot, act, m_tiles, n_tiles;
    int pixmaj, Nimg;     // pixel-major M tiles: a tile = ONE pixel position of BM consecutive images (see kernel)
    const float* x2;      // optional second input [N,H,W,Cin2]: its 1x1 conv is appended to the K loop (the residual
    int Cin2;             //   block's downsample branch fused into conv2: out = act(conv3x3(x) + conv1x1(x2) + bias))
    int ksplit;           // > 1: blockIdx.y = K slice; raw partial sums go to `partial` [ksplit][M][ldp], no epilogue math
    int ldp;
    float* partial;
    int nbatch;
    int64_t xb, wb, yb;   // batched launches (blockIdx.z = batch index): element strides of x, w, y between batches
};

// __launch_bounds__(256, w): w = workgroups per CU the LDS admits (= waves per SIMD), so the register allocator may
// use 512/w registers: with the default budget it spilled the staged B chunk to scratch and waited for the global
// loads BEFORE the MFMA phase (v1: 60 % MFMA-busy).
// BKT = K-chunk in floats; PRIO = 1: s_setprio around the MFMA phase (measured: no effect; kept for A/B runs).
// The kernel body as a device function (round 3): the training step runs a unit's data-gradient convolution and its weight-gradient
// GEMM in ONE launch (be_train.hip: k_unit_gemms), a workgroup picking its role from its index; bx / by / bz are what blockIdx.x / y / z
// are in the stand-alone kernel k_conv_igemm (be_conv.hip) and smem its dynamic LDS (2 (BM + BN) (BKT + 4) floats).
// UNI (round 3): every row of every tile is inside the problem and every tap a tile visits is inside the image for all of its rows
// (pixel-major tiles of whole image groups; 1x1 convolutions / linears with M a multiple of the tile): no border test, no zero
// fill, and the operand addresses are a per-thread constant + a wave-uniform offset - the register-staged loop then issues ~0 VALU
// instructions per chunk instead of ~30 (the 64 x 64 training tiles issued 4 VALU per MFMA, profiles/r03_train_pmc_summary.json).
// The host asserts the preconditions (conv_dispatch).  Same arithmetic, same order: bit-identical to UNI = false.
template <int WM, int WN, int MT, int NT, int MODE, int BKT, int PRIO, bool UNI = false>
__device__ __forceinline__ void conv_igemm_body(ConvArgs a, float* smem, const int bx, const int by, const int bz) {
    static_assert(WM * WN == 4, "4 waves");
    constexpr int BM = WM * MT * 32;                   // 128 (inference tiles) or 64 (small-M training tile)
    static_assert(BKT == 32 || BKT == 16 || BKT == 8, "K chunk");
    constexpr int BN = WN * NT * 32;
    constexpr int LROW = BKT + 4;                      // floats per LDS row (pad 4: conflict-free b128 reads/writes)
    constexpr int QL = BKT / 4;                        // lanes per staged row (16 B each)
    constexpr int RP = 256 / QL;                       // rows staged per pass
    constexpr int NA = BM / RP, NB = (BN + RP - 1) / RP;   // staging vectors per thread (last B pass may be partial)
    constexpr bool B_PARTIAL = BN % RP != 0;
    constexpr int SUB = 32 / BKT;                      // chunks per 32-channel packing unit
    if (bz) {                                          // batched launch: this block works on batch bz
        a.x += (int64_t)bz * a.xb; a.w += (int64_t)bz * a.wb; a.y += (int64_t)bz * a.yb;
    }
    float* As = smem;                                  // [2][BM][LROW]
    float* Bs = smem + 2 * BM * LROW;                  // [2][BN][LROW]

    // ---- block -> tile (XCD-aware: blocks b, b+8, b+16.. share an XCD and walk the N tiles of one M tile)
    const int bid = bx;
    const int xcd = bid & 7, slot = bid >> 3;
    const int n_tile = slot % a.n_tiles;
    int m_tile;
    if (a.pixmaj == 1) {
        // pixel-major: ALL pixel tiles of one group of BM images run on the same XCD, one after the other (the taps of
        // neighbouring pixels re-read the same input lines: with the tiles of a group spread over the 8 XCDs the L2
        // hit rate fell from 94 % to 69 % and L2-miss reads rose 6x).  Tried and rejected: pixel fastest / N tile slower
        // (one N tile's weights resident at a time): +11 % L2-miss bytes, 1.3 % slower.
        const int t = slot / a.n_tiles;
        const int grp = (t / a.HW) * 8 + xcd;
        m_tile = grp * a.HW + t % a.HW;
    } else {
        m_tile = (slot / a.n_tiles) * 8 + xcd;
    }
    if (m_tile >= a.m_tiles) return;
    const int n0 = n_tile * BN;
    // Row -> output pixel.  Flat tiles take BM consecutive (image, pixel) rows.  Pixel-major tiles (large batches)
    // take ONE pixel position of BM consecutive images: every row of the tile then sees the same zero padding, so the
    // taps that fall outside the image are skipped for the whole tile instead of being multiplied by zeros
    // (3x3 on 6x6: 256 of 324 pixel-tap pairs are inside: 21 % fewer MFMAs for the same convolution).
    // pixel order inside a group of images: interior pixels (all taps) first, the border ring (fewer taps) last, so
    // the short tiles fill the tail of the launch
    int pix_u = 0;
    if (a.pixmaj) {
        const int idx = m_tile % a.HW, ni = (a.H - 2) * (a.W - 2);
        int py, px;
        if (idx < ni) { py = 1 + idx / (a.W - 2); px = 1 + idx % (a.W - 2); }
        else {
            const int e = idx - ni;
            if (e < a.W) { py = 0; px = e; }
            else if (e < 2 * a.W) { py = a.H - 1; px = e - a.W; }
            else if (e < 2 * a.W + a.H - 2) { px = 0; py = 1 + e - 2 * a.W; }
            else { px = a.W - 1; py = 1 + e - 2 * a.W - (a.H - 2); }
        }
        pix_u = py * a.W + px;
    }
    const int row_base = a.pixmaj ? (m_tile / a.HW) * BM : m_tile * BM;      // first image (pixmaj) or first flat row

    const int tid = threadIdx.x;
    const int q = tid % QL, r0 = tid / QL;             // staging: QL lanes x 16 B = one row chunk

    // ---- per-thread A rows: flat output pixel m -> (y,x) for the border test; base offset m*Cin
    int a_off[NA], a_yx[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int rr = row_base + r0 + RP * i;
        const bool live = a.pixmaj ? rr < a.Nimg : rr < a.M;
        const int m = a.pixmaj ? rr * a.HW + pix_u : rr;
        if (live) {
            const int pp = a.pixmaj ? pix_u : m % a.HW;
            const int yy = pp / a.W, xx = pp - yy * a.W;
            a_yx[i] = (yy << 16) | xx;
            a_off[i] = m;                              // multiplied by Cin at use (fits 32 bit: checked on host)
        } else {
            a_yx[i] = -1;                              // row outside the problem: always zero
            a_off[i] = 0;
        }
    }
    // taps this tile has to visit, 4 bits each (flat tiles: all of them)
    const int ntap_all = MODE == MODE_TAPS ? a.ks * a.ks : 7;
    unsigned long long tap_list = 0;
    int ntap = 0;
    {
        const int py = pix_u / a.W, px = pix_u - py * a.W, half = a.ks >> 1;
        for (int t = 0; t < ntap_all; ++t) {
            bool ok = true;
            if (a.pixmaj) {
                if (MODE == MODE_TAPS) {
                    const int yy = py + t / a.ks - half, xx = px + t % a.ks - half;
                    ok = (unsigned)yy < (unsigned)a.H && (unsigned)xx < (unsigned)a.W;
                } else {
                    ok = (unsigned)(py + t - 3) < (unsigned)a.H;
                }
            }
            if (ok) { tap_list |= (unsigned long long)t << (4 * ntap); ++ntap; }
        }
    }
    const float* wbase = a.w + (size_t)n0 * a.Ktot + 4 * q;   // + row * Ktot + chunk offset

    // Staging registers for the next K chunk.  Straight-line helpers on array references (no lambdas, no
    // conditionals around the loads): anything else made hipcc keep b_st in scratch memory.
    f32x4 a_st[NA], b_st[NB];
    unsigned a_ok = 0;                                 // bit i: a_st[i] is inside the image (else stored as zeros)
    // UNI: byte offsets of this thread's staged rows from the (wave-uniform) chunk bases; < 2^32 (host-checked)
    unsigned a_vo[NA], b_vo[NB];
    if (UNI) {
#pragma unroll
        for (int i = 0; i < NA; ++i) a_vo[i] = (unsigned)(a_off[i] * a.Cin + 4 * q) * 4u;
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            // rows past the packed weights (the empty half of a 96-channel layer's second tile): re-read row 0, never stored
            const int row_ = ((B_PARTIAL && RP * i + r0 >= BN) || n0 + RP * i + r0 >= ((a.Cout + 31) & ~31)) ? 0 : RP * i + r0;
            b_vo[i] = (unsigned)(row_ * a.Ktot + 4 * q) * 4u;
        }
    }
    const char* const wbase_u = reinterpret_cast<const char*>(a.w + (size_t)n0 * a.Ktot);      // uniform
#define BE_LOAD_CHUNK_UNI(KC) BE_LOAD_UNI(KC, a_st, b_st)
#define BE_LOAD_UNI(KC, AST, BST)                                                                               \
    do {                                                                                                        \
        const int k32_ = (KC) / SUB, sub_ = (KC) - k32_ * SUB;                                                  \
        const int cc_ = k32_ / ntap, j_ = k32_ - cc_ * ntap;                                                    \
        const int tap_ = (int)((tap_list >> (4 * j_)) & 15ull);                                                 \
        const int half_ = a.ks >> 1;                                                                            \
        const int dy_ = tap_ / a.ks - half_, dx_ = tap_ % a.ks - half_;                                         \
        const int so_ = __builtin_amdgcn_readfirstlane((dy_ * a.W + dx_) * a.Cin + cc_ * 32 + sub_ * BKT);      \
        const int kw_ = __builtin_amdgcn_readfirstlane(((cc_ * ntap_all + tap_) * SUB + sub_) * BKT);           \
        const char* ab_ = reinterpret_cast<const char*>(a.x) + (int64_t)so_ * 4;                                \
        const char* bb_ = wbase_u + (int64_t)kw_ * 4;                                                           \
        _Pragma("unroll") for (int i_ = 0; i_ < NA; ++i_) AST[i_] = *reinterpret_cast<const f32x4*>(ab_ + a_vo[i_]); \
        _Pragma("unroll") for (int i_ = 0; i_ < NB; ++i_) BST[i_] = *reinterpret_cast<const f32x4*>(bb_ + b_vo[i_]); \
    } while (0)
#define BE_STORE_UNI(BUF, AST, BST)                                                                             \
    do {                                                                                                        \
        float* Ad_ = As + (BUF) * BM * LROW;                                                                    \
        float* Bd_ = Bs + (BUF) * BN * LROW;                                                                    \
        _Pragma("unroll") for (int i_ = 0; i_ < NA; ++i_)                                                       \
            *reinterpret_cast<f32x4*>(Ad_ + (r0 + RP * i_) * LROW + 4 * q) = AST[i_];                           \
        _Pragma("unroll") for (int i_ = 0; i_ < NB; ++i_)                                                       \
            if (!B_PARTIAL || RP * i_ + r0 < BN)                                                                \
                *reinterpret_cast<f32x4*>(Bd_ + (r0 + RP * i_) * LROW + 4 * q) = BST[i_];                       \
    } while (0)
#define BE_LOAD_CHUNK(KC)                                                                                       \
    do {                                                                                                        \
        if (UNI) { BE_LOAD_CHUNK_UNI(KC); break; }                                                              \
        int dy_, dx_, coff_;                                                                                    \
        int kw_;   /* chunk index into the packed weights, in BKT units */                                     \
        const bool second_ = MODE == MODE_TAPS && (KC) >= n1;   /* fused 1x1 branch on x2 */                    \
        if (second_) {                                                                                          \
            const int k2_ = (KC) - n1;                                                                          \
            dy_ = 0; dx_ = 0;                                                                                   \
            coff_ = k2_ * BKT + 4 * q;                                                                          \
            kw_ = a.nchunk * SUB + k2_;                                                                         \
        } else if (MODE == MODE_TAPS) {                                                                         \
            const int k32_ = (KC) / SUB, sub_ = (KC) - k32_ * SUB;                                              \
            const int cc_ = k32_ / ntap, j_ = k32_ - cc_ * ntap;                                                \
            const int tap_ = (int)((tap_list >> (4 * j_)) & 15ull);                                             \
            const int half_ = a.ks >> 1;                                                                        \
            dy_ = tap_ / a.ks - half_; dx_ = tap_ % a.ks - half_;                                               \
            coff_ = (dy_ * a.W + dx_) * a.Cin + cc_ * 32 + sub_ * BKT + 4 * q;                                  \
            kw_ = (cc_ * ntap_all + tap_) * SUB + sub_;                                                         \
        } else { /* conv1: 32 floats = kernel row kh, 8 pixels x 4 channels; a chunk is BKT/4 of those pixels */ \
            const int k32_ = (KC) / SUB, sub_ = (KC) - k32_ * SUB;                                              \
            const int kh_ = (int)((tap_list >> (4 * k32_)) & 15ull);                                            \
            dy_ = kh_ - 3; dx_ = q - 3 + (BKT / 4) * sub_;                                                      \
            coff_ = (dy_ * a.W + dx_) * 4;                                                                      \
            kw_ = kh_ * SUB + sub_;                                                                             \
        }                                                                                                       \
        a_ok = 0;                                                                                               \
        _Pragma("unroll") for (int i_ = 0; i_ < NA; ++i_) {                                                     \
            const int yy_ = (a_yx[i_] >> 16) + dy_, xx_ = (a_yx[i_] & 0xffff) + dx_;                            \
            const bool ok_ = a_yx[i_] >= 0 && (unsigned)yy_ < (unsigned)a.H && (unsigned)xx_ < (unsigned)a.W;   \
            /* branch-free: out-of-image taps read the (valid) first 16 B of the tensor and are zeroed */      \
            int64_t off_ = ok_ ? (int64_t)a_off[i_] * (second_ ? a.Cin2 : a.Cin) + coff_ : 0;                   \
            asm volatile("" : "+v"(off_));  /* opaque: keeps the load unconditional (no exec-mask branch) */    \
            a_st[i_] = *reinterpret_cast<const f32x4*>((second_ ? a.x2 : a.x) + off_);  /* zeroed at store */   \
            a_ok |= (ok_ ? 1u : 0u) << i_;                                                                      \
        }                                                                                                       \
        _Pragma("unroll") for (int i_ = 0; i_ < NB; ++i_) {                                                     \
            /* rows past the tile (partial pass, or BN < RP) read row 0 of the tile instead and are never stored */ \
            const int row_ = (B_PARTIAL && RP * i_ + r0 >= BN) ? 0 : RP * i_ + r0;                              \
            b_st[i_] = *reinterpret_cast<const f32x4*>(wbase + (size_t)row_ * a.Ktot + kw_ * BKT);              \
        }                                                                                                       \
    } while (0)
#define BE_STORE_CHUNK(BUF)                                                                                     \
    do {                                                                                                        \
        float* Ad_ = As + (BUF) * BM * LROW;                                                                    \
        float* Bd_ = Bs + (BUF) * BN * LROW;                                                                    \
        _Pragma("unroll") for (int i_ = 0; i_ < NA; ++i_)                                                       \
            *reinterpret_cast<f32x4*>(Ad_ + (r0 + RP * i_) * LROW + 4 * q) =                                    \
                (UNI || ((a_ok >> i_) & 1u)) ? a_st[i_] : f32x4{0.f, 0.f, 0.f, 0.f};                            \
        _Pragma("unroll") for (int i_ = 0; i_ < NB; ++i_)                                                       \
            if (!B_PARTIAL || RP * i_ + r0 < BN)                                                                \
                *reinterpret_cast<f32x4*>(Bd_ + (r0 + RP * i_) * LROW + 4 * q) = b_st[i_];                      \
    } while (0)

    // ---- wave / lane roles for the MFMA phase
    const int wave = tid >> 6, lane = tid & 63;
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, lh = lane >> 5;
    const int a_frag0 = ((wm * MT) * 32 + li) * LROW + 4 * lh;
    const int b_frag0 = ((wn * NT) * 32 + li) * LROW + 4 * lh;

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const int n1 = (MODE == MODE_TAPS ? (a.nchunk / ntap_all) * ntap : ntap) * SUB;       // only the taps this tile visits
    const int nchunk_all = n1 + (a.x2 ? (a.Cin2 / 32) * SUB : 0);                          // + the fused 1x1 branch
    // split-K (small-M launches): this block walks the chunks [kc0, nchunk) of its K slice
    const int kc0 = a.ksplit > 1 ? (int)((int64_t)nchunk_all * by / a.ksplit) : 0;
    const int nchunk = a.ksplit > 1 ? (int)((int64_t)nchunk_all * (by + 1) / a.ksplit) : nchunk_all;
#define BE_MFMA_PHASE(BUF)                                                                                      \
    do {                                                                                                        \
        const float* Ab = As + (BUF) * BM * LROW + a_frag0;                                                     \
        const float* Bb = Bs + (BUF) * BN * LROW + b_frag0;                                                     \
        if (PRIO == 1) __builtin_amdgcn_s_setprio(1);                                                           \
        _Pragma("unroll") for (int g = 0; g < BKT / 8; ++g) {                                                   \
            f32x4 af[MT], bf[NT];                                                                               \
            _Pragma("unroll") for (int i = 0; i < MT; ++i) af[i] = *reinterpret_cast<const f32x4*>(Ab + i * 32 * LROW + 8 * g); \
            _Pragma("unroll") for (int j = 0; j < NT; ++j) bf[j] = *reinterpret_cast<const f32x4*>(Bb + j * 32 * LROW + 8 * g); \
            _Pragma("unroll") for (int i = 0; i < MT; ++i)                                                      \
                _Pragma("unroll") for (int j = 0; j < NT; ++j) {                                                \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].x, bf[j].x, acc[i][j], 0, 0, 0);     \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].y, bf[j].y, acc[i][j], 0, 0, 0);     \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].z, bf[j].z, acc[i][j], 0, 0, 0);     \
                    /* conv1's row mode: the fourth channel of a pixel is zero in the staging and in the pack (cin <= 3): not issued */ \
                    if (MODE != MODE_ROW8) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].w, bf[j].w, acc[i][j], 0, 0, 0); \
                }                                                                                               \
        }                                                                                                       \
        if (PRIO == 1) __builtin_amdgcn_s_setprio(0);                                                           \
    } while (0)

    if (UNI) {
        // operands TWO chunks ahead in two register sets (the addresses cost nothing here): the loads of chunk k + 2 are issued
        // before the MFMAs of chunk k and are not needed until the end of step k + 1 - a global load's latency is ~3 MFMA phases of
        // a 64 x 64 tile, and the one-ahead pipeline of the general path stalled on it every step (matrix pipe 0.37 busy)
        f32x4 a_s2[NA], b_s2[NB];
        const int last = nchunk - 1;
        BE_LOAD_UNI(kc0, a_st, b_st);
        BE_STORE_UNI(kc0 & 1, a_st, b_st);
        BE_LOAD_UNI(kc0 + 1 < nchunk ? kc0 + 1 : last, a_st, b_st);
        __syncthreads();
        // whole pairs of steps first (no exit between the two MFMA phases of a trip: with one the compiler kept the accumulators
        // in two register ranges and copied them across), then the odd step
        const int buf = kc0 & 1;
        int kc = kc0;
#pragma unroll 1
        for (; kc + 1 < nchunk; kc += 2) {
            BE_LOAD_UNI(kc + 2 < nchunk ? kc + 2 : last, a_s2, b_s2);
            __builtin_amdgcn_sched_barrier(0);
            BE_MFMA_PHASE(buf);
            __builtin_amdgcn_sched_barrier(0);
            BE_STORE_UNI(buf ^ 1, a_st, b_st);             // chunk kc + 1
            __syncthreads();
            BE_LOAD_UNI(kc + 3 < nchunk ? kc + 3 : last, a_st, b_st);
            __builtin_amdgcn_sched_barrier(0);
            BE_MFMA_PHASE(buf ^ 1);
            __builtin_amdgcn_sched_barrier(0);
            BE_STORE_UNI(buf, a_s2, b_s2);                 // chunk kc + 2
            __syncthreads();
        }
        if (kc < nchunk) BE_MFMA_PHASE(buf);               // odd count: the last chunk is in LDS already
    } else {
    BE_LOAD_CHUNK(kc0);
    BE_STORE_CHUNK(kc0 & 1);
    __syncthreads();

    for (int kc = kc0; kc < nchunk; ++kc) {
        const int buf = kc & 1;
        // always prefetch (the last iteration re-reads its own chunk into the idle buffer: keeps the body branch-free)
        const int kn = kc + 1 < nchunk ? kc + 1 : kc;
        BE_LOAD_CHUNK(kn);
        __builtin_amdgcn_sched_barrier(0);            // the loads stay ABOVE the MFMA phase (hipcc sank them below it)
        BE_MFMA_PHASE(buf);
        __builtin_amdgcn_sched_barrier(0);            // ... and the LDS hand-over stays below it
        BE_STORE_CHUNK(buf ^ 1);
        __syncthreads();
    }
    }
#undef BE_LOAD_CHUNK
#undef BE_LOAD_CHUNK_UNI
#undef BE_LOAD_UNI
#undef BE_STORE_UNI
#undef BE_MFMA_PHASE
#undef BE_STORE_CHUNK

    // ---- epilogue: D[row][col], col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    if (a.ksplit > 1) {                                // raw partial sums; k_splitk_reduce does bias / residual / activation
        float* P = a.partial + (size_t)by * a.M * a.ldp;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int c = n0 + (wn * NT + j) * 32 + li;
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rr = row_base + (wm * MT + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    const int m = a.pixmaj ? rr * a.HW + pix_u : rr;
                    if (c < a.ldp && (a.pixmaj ? rr < a.Nimg : rr < a.M)) P[(size_t)m * a.ldp + c] = acc[i][j][r];
                }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int c = n0 + (wn * NT + j) * 32 + li;
        const bool c_ok = c < a.Cout;
        const float bias = (c_ok && a.bias) ? a.bias[c] : 0.0f;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rr = row_base + (wm * MT + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const int m = a.pixmaj ? rr * a.HW + pix_u : rr;
                if (c_ok && (a.pixmaj ? rr < a.Nimg : rr < a.M)) {
                    float v = acc[i][j][r] + bias;
                    if (a.res) v += a.res[(size_t)m * a.ldy + c];
                    if (a.act == 1) v = be::smish(v);
                    else if (a.act == 2) v = fmaxf(v, 0.0f);
                    a.y[(size_t)m * a.ldy + c] = v;
                }
            }
        }
    }
}

}  // namespace be_igemm

namespace be {
// be_conv.hip: a small-M training convolution PREPARED but not launched: the kernel arguments, the tile variant (0: 64x64 tiles =
// conv_igemm_body<2,2,1,1,MODE_TAPS,16,0>, 1: 128x32 tiles = <4,1,1,1,...>, 2: 64x64 uniform tiles = <2,2,1,1,...,UNI>), its grid (gx workgroups x S K-slices) and where the raw
// slices go ([S][M][ldp] in scratch when S > 1; S == 1: the epilogue writes y = conv + bias (+ res)).
struct ConvPrep { be_igemm::ConvArgs args; int variant, S, ldp; unsigned gx; double flops, flops_exec; };
int conv_train_prepare(const be_conv_desc* d, const float* x, const float* pw, const float* pb, const float* res, float* y, int ldy,
                       void* scratch, size_t scratch_bytes, ConvPrep* prep, bool pad64 = false);
}  // namespace be
