// Winograd F(3x3,3x3) convolution of large batches with the transforms INSIDE the GEMM kernel's epilogue (round 2).
//
// The separate transform kernels of be_wino.hip were at the HBM copy rate and still a fifth of the LocalStage step: every
// 3x3 layer wrote its 25 transform-domain products M (100 values per patch and channel), a transform kernel read them back,
// formed the 6x6 map and wrote the next layer's transform-domain input V (another 100 values), which the next GEMM read.
// Here a workgroup walks the 25 positions of ITS (row tile, column tile) back to back - one continuous software pipeline,
// like k_wino_gemm - and folds every finished position straight into the nine output accumulators Y of the 3x3 output
// block (Y_o += A^T[o1][z1] A^T[o2][z2] * acc: at most nine fmaf per accumulator register and position).  After position
// 24 a lane holds, for its output channel, the complete 6x6 maps of four patches: the MFMA accumulator layout puts rows
// 8 j + 4 (lane >> 5) + i, i = 0..3, in consecutive registers, and rows are ordered (patch, tile), so those four registers
// are the four 3x3 tiles of one patch.  Bias, residual, Smish, the 2x2 max-pool of the last block and the NEXT layer's
// input transform (B^T d B on the 5x5 windows of the zero-padded map) all happen in those registers; what goes to HBM is
// the next V (and/or the block's output y).  M never exists; per layer the HBM traffic falls from 400 to 200 values per
// patch and channel and two launches out of three disappear.
//
// Price: Y must live in architected VGPRs (VALU cannot touch AGPRs): 9 x 16 registers per 32x32 MFMA tile, so a wave owns
// ONE 32x32 tile (the stand-alone GEMM gave it 64x64) and moves twice the operand bytes per MFMA.  Issue model of DESIGN
// 3.1d per K chunk of 16 and wave: 8 MFMAs x 64 + 4 fragment reads x 16 + 2 DMA pieces x 65 + ~20 = 726 cycles (70 %).
//
// Arithmetic: fp32 MFMA products accumulated in ascending K exactly as k_conv_igemm / k_wino_gemm; the transforms are the
// shared routines of be_wino_math.h, so the small-batch path (separate kernels) gives bit-identical results.
#include <cstdlib>
#include "be_common.h"
#include "be_device_math.h"
#include "be_wino_math.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

struct FusedArgs {
    const float* V;        // tile-major transform-domain input  [4n][25][K]
    const float* U;        // transformed weights                [25][Npad][K]
    const float* bias;     // [Npad]
    const float* res;      // residual [n][6][6][N] or null
    float* y;              // WY: [n][6][6][N]; POOL: [n][3][3][N]
    float* Vout;           // WV: tile-major [4n][25][N]
    int n, K, N, Npad, act;
    int m_tiles, n_tiles;
};

// 64 x 64 tile, four waves (2 x 2) of one 32x32 MFMA tile each; two workgroups per CU (two waves per SIMD, 256 registers each).
// WY: write the block output y;  WV: write the next layer's transform-domain input;  POOL: write maxpool2x2(y) instead of y.
// Pipeline: a stage = G = 2 K-chunks of 16 (16 MFMAs per wave and barrier), ring of R = 3 stages filled by LDS-DMA two stages
// ahead, ONE raw s_barrier per stage and a counted vmcnt (the loop issues no other vector-memory operation, so "all but the
// newest (R - 2) stages' DMAs have landed" is exact) - with 512 MFMA cycles per chunk a two-stage scheme with vmcnt(0) left
// the DMA latency exposed every chunk (first version: 80 TFLOP/s).
template <bool WY, bool WV, bool POOL>
__global__ __launch_bounds__(256, 2)
void k_wino_fused(FusedArgs a) {
    constexpr int BM = 64, BN = 64, BKT = 16, G = 2, R = 3;               // four waves
    constexpr int CHUNK = (BM + BN) * BKT;                            // floats per K-chunk of a stage (8 KB): A 64 x 16, B 64 x 16
    constexpr int STAGE = CHUNK * G;
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    const int bid = blockIdx.x;
    const int xcd = bid & 7, slot = bid >> 3;          // all column tiles of a row tile on ONE XCD, side by side: its A rows
    const int n_tile = slot % a.n_tiles;               // come from HBM once and from that L2 afterwards
    const int m_tile = (slot / a.n_tiles) * 8 + xcd;
    if (m_tile >= a.m_tiles) return;
    const int n0 = n_tile * BN, row_base = m_tile * BM;
    const int M = 4 * a.n;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lh = lane >> 5;
    const int lda = 25 * a.K;
    // staging: a chunk is eight 1-KB pieces (16 rows x 64 B): A rows 16 w.. and B rows 16 w.. are wave w's; lane -> (row = lane >> 2,
    // slot = lane & 3) fetches the 16-byte quad that belongs into that slot after the XOR swizzle by (row >> 2) & 3
    const int srow = lane >> 2, sq = (lane & 3) ^ ((lane >> 4) & 3);
    const int ar = wave * 16 + srow;
    // rows past M (ragged last tile) load row 0 of the tile: valid memory, results never stored
    const unsigned a_off = (unsigned)((row_base + ar < M ? ar : 0) * lda + 4 * sq) * 4u;
    const unsigned b_off = (unsigned)(ar * a.K + 4 * sq) * 4u;
    const float* xt = a.V + (int64_t)row_base * lda;   // uniform
    const float* wt = a.U + (int64_t)n0 * a.K;
    const int64_t wb = (int64_t)a.Npad * a.K;
    const int kstages = a.K / (BKT * G), total = kstages * 25;        // K % 32 == 0: a stage never straddles two positions
    const int fsw = (li >> 2) & 3;
    const int a_fr0 = (wm * 32 + li) * BKT + 4 * (lh ^ fsw), a_fr1 = (wm * 32 + li) * BKT + 4 * ((lh + 2) ^ fsw);
    const int b_fr0 = BM * BKT + (wn * 32 + li) * BKT + 4 * (lh ^ fsw), b_fr1 = BM * BKT + (wn * 32 + li) * BKT + 4 * ((lh + 2) ^ fsw);
    int lz = 0, lk = 0, l_buf = 0;                     // (position, stage within it, ring slot) of the next load
#define WF_LOAD()                                                                                               \
    do {                                                                                                        \
        const char* xs_ = reinterpret_cast<const char*>(xt + (int64_t)lz * a.K + lk * (BKT * G));                \
        const char* ws_ = reinterpret_cast<const char*>(wt + (int64_t)lz * wb + lk * (BKT * G));                 \
        float* st_ = smem_f + l_buf * STAGE + wave * 256;                                                       \
        _Pragma("unroll") for (int g_ = 0; g_ < G; ++g_) {                                                      \
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(xs_ + g_ * BKT * 4 + a_off), (lds_ptr_t)(st_ + g_ * CHUNK), 16, 0, 0);            \
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(ws_ + g_ * BKT * 4 + b_off), (lds_ptr_t)(st_ + g_ * CHUNK + BM * BKT), 16, 0, 0); \
        }                                                                                                       \
        if (lk + 1 < kstages) ++lk; else if (lz + 1 < 25) { lk = 0; ++lz; }   /* past the end: the last stage again */ \
        l_buf = l_buf == R - 1 ? 0 : l_buf + 1;                                                                 \
    } while (0)
    f32x16 acc, Y[9];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
    for (int o = 0; o < 9; ++o)
#pragma unroll
        for (int r = 0; r < 16; ++r) Y[o][r] = 0.0f;
    WF_LOAD();                                         // stage 0
    WF_LOAD();                                         // stage 1
    int cz = 0, ck = 0, r_buf = 0;                     // (position, stage) being multiplied, its ring slot
    for (int st = 0; st < total; ++st) {
        __builtin_amdgcn_s_waitcnt(0x0F70 | (2 * G));  // vmcnt(2 G): everything but the newest stage's DMAs has landed
        __builtin_amdgcn_s_barrier();                  // ... for every wave; and everyone has left slot (st + 2) % 3
        WF_LOAD();                                     // stage st + 2
        __builtin_amdgcn_sched_barrier(0);
        {
            const float* sb = smem_f + r_buf * STAGE;
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const f32x4 a0 = *reinterpret_cast<const f32x4*>(sb + g * CHUNK + a_fr0), b0 = *reinterpret_cast<const f32x4*>(sb + g * CHUNK + b_fr0);
                const f32x4 a1 = *reinterpret_cast<const f32x4*>(sb + g * CHUNK + a_fr1), b1 = *reinterpret_cast<const f32x4*>(sb + g * CHUNK + b_fr1);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, b0.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, b0.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, b0.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, b0.w, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, b1.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, b1.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, b1.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, b1.w, acc, 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        r_buf = r_buf == R - 1 ? 0 : r_buf + 1;
        if (++ck == kstages) {                          // position cz is complete: fold it into the output block, clear
            const int z1 = cz / 5, z2 = cz - 5 * z1;    // uniform
#pragma unroll
            for (int o1 = 0; o1 < 3; ++o1)
#pragma unroll
                for (int o2 = 0; o2 < 3; ++o2) {
                    const float coef = be::wino_at(o1, z1) * be::wino_at(o2, z2);      // scalar; +-2^k or 0
                    if (coef != 0.0f) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) Y[3 * o1 + o2][r] = __builtin_fmaf(coef, acc[r], Y[3 * o1 + o2][r]);
                    }
                }
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
            ck = 0; ++cz;
        }
    }
#undef WF_LOAD
    __builtin_amdgcn_s_waitcnt(0x0F70);                 // the two over-run stages' DMAs, before LDS is released
    // ---- epilogue: registers 4 j + i of every Y[o] = tile i (ty = i >> 1, tx = i & 1) of patch 2 j + lh of this wave's 8 patches.
    // The patch loop is NOT unrolled (one copy of ~1500 instructions instead of four: the unrolled epilogue was 28 000
    // instructions, 3.5 x the instruction cache); the body always works on registers 0..3 and the Y vectors are rotated by
    // four registers at its end.  Addresses = uniform base (patch pair) + a 32-bit lane offset: scalar-base stores.
    const int col = n0 + wn * 32 + li;                  // N % 64 == 0: always a real channel
    const float bv = a.bias[col];
    const int64_t pu0 = (int64_t)(row_base + wm * 32) / 4;            // uniform: this wave's first patch
    const unsigned yoff = (unsigned)((lh * 36 * a.N + col) * 4);      // bytes from the pair's first patch in y / res
    const unsigned poff = (unsigned)((lh * 9 * a.N + col) * 4);       // ... in the pooled y
    const unsigned voff = (unsigned)((lh * 100 * a.N + col) * 4);     // ... in Vout ([4 tiles][25][N] per patch)
#pragma unroll 1
    for (int j = 0; j < 4; ++j) {
        const int64_t pu = pu0 + 2 * j;                 // uniform; this lane's patch is pu + lh
        const bool live = pu + lh < a.n;                // ragged last tile
        // the ~180 row offsets k * N * 4 below are invariant in j: hoisted out of the loop they would take 360 SGPRs (spilled,
        // through VGPR lanes); an opaque copy of N per iteration keeps them as a few scalar multiplies next to their use
        int Nj = a.N;
        asm volatile("" : "+s"(Nj));
        float map[6][6];
        const char* rb = a.res ? reinterpret_cast<const char*>(a.res + (size_t)pu * 36 * Nj) : nullptr;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const int yy = 3 * (i >> 1) + r, xx = 3 * (i & 1) + c;
                    float v = Y[3 * r + c][i] + bv;
                    if (rb) v += live ? *reinterpret_cast<const float*>(rb + (size_t)((yy * 6 + xx) * Nj * 4) + yoff) : 0.0f;
                    if (a.act == 1) v = be::smish(v); else if (a.act == 2) v = fmaxf(v, 0.0f);
                    map[yy][xx] = v;
                }
        if (WY && live) {
            char* yb = reinterpret_cast<char*>(a.y + (size_t)pu * 36 * Nj);
#pragma unroll
            for (int yy = 0; yy < 6; ++yy)
#pragma unroll
                for (int xx = 0; xx < 6; ++xx) *reinterpret_cast<float*>(yb + (size_t)((yy * 6 + xx) * Nj * 4) + yoff) = map[yy][xx];
        }
        if (POOL && live) {
            char* yb = reinterpret_cast<char*>(a.y + (size_t)pu * 9 * Nj);
#pragma unroll
            for (int py = 0; py < 3; ++py)
#pragma unroll
                for (int px = 0; px < 3; ++px)
                    *reinterpret_cast<float*>(yb + (size_t)((py * 3 + px) * Nj * 4) + poff) =
                        fmaxf(fmaxf(map[2 * py][2 * px], map[2 * py][2 * px + 1]), fmaxf(map[2 * py + 1][2 * px], map[2 * py + 1][2 * px + 1]));
        }
        if (WV) {
            char* vb = reinterpret_cast<char*>(a.Vout + (size_t)pu * 100 * Nj);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                float d[5][5], v[25];
#pragma unroll
                for (int r = 0; r < 5; ++r)
#pragma unroll
                    for (int c = 0; c < 5; ++c) {
                        const int yy = 3 * (t >> 1) - 1 + r, xx = 3 * (t & 1) - 1 + c;
                        d[r][c] = (yy >= 0 && yy < 6 && xx >= 0 && xx < 6) ? map[yy < 0 ? 0 : (yy > 5 ? 5 : yy)][xx < 0 ? 0 : (xx > 5 ? 5 : xx)] : 0.0f;
                    }
                be::wino_in25(d, v);
                if (live) {
#pragma unroll
                    for (int z = 0; z < 25; ++z) *reinterpret_cast<float*>(vb + (size_t)((t * 25 + z) * Nj * 4) + voff) = v[z];
                }
            }
        }
#pragma unroll
        for (int o = 0; o < 9; ++o)
            Y[o] = __builtin_shufflevector(Y[o], Y[o], 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 0, 1, 2, 3);
    }
}

template <bool WY, bool WV, bool POOL>
int launch_fused(const FusedArgs& a, hipStream_t s, int cin) {
    constexpr size_t lds = (size_t)3 * 2 * (64 + 64) * 16 * sizeof(float);          // 48 KB: three stages of two K-chunks
    const unsigned grid = (unsigned)(8 * ((a.m_tiles + 7) / 8) * a.n_tiles);
    {
        const double n = a.n;
        be::ProfileScope prof(s, BE_KERNEL_WINO_GEMM, 25.0 * 2.0 * 4 * n * cin * a.N,
                              4.0 * (100.0 * n * cin + 25.0 * cin * a.N + (WV ? 100.0 : 0.0) * n * a.N + (WY ? 36.0 : POOL ? 9.0 : 0.0) * n * a.N),
                              25.0 * 2.0 * a.m_tiles * a.n_tiles * 64.0 * 64.0 * cin);
        hipLaunchKernelGGL((k_wino_fused<WY, WV, POOL>), dim3(grid), dim3(256), lds, s, a);
    }
    return be::check_launch("be_wino (fused transform GEMM)");
}

}  // namespace

namespace be {

// OPT-IN (BE_WINO_FUSED=1), not the default: measured on MI355X (profiles/r02_fused_experiment.md) the kernel is correct and
// halves the transform-domain HBM traffic, but its main loop reaches 115 TFLOP/s where the weight-stationary GEMM does 128
// (a 32x32 wave tile moves twice the operand bytes per MFMA) and the epilogue's ~1300 VALU instructions per patch compete
// with the MFMAs for the same issue slots instead of hiding behind HBM as they do in the stand-alone transform kernels:
// 14.2-14.5 ms per step against 14.0 ms for GEMM + separate transforms.
bool wino_fused_ok(int64_t n, int cin, int cout) {
    static const bool on = getenv("BE_WINO_FUSED") != nullptr && atoi(getenv("BE_WINO_FUSED")) != 0;
    return on && n >= 512 && cin % 32 == 0 && cout % 64 == 0 && 100 * n * (int64_t)(cin > cout ? cin : cout) < ((int64_t)1 << 31);
}

// One Winograd layer on tile-major V [4n][25][cin]: y (mode & 1), the next V (mode & 2), or the pooled y (mode == 4).
int wino_fused(const float* V, const float* packed_w, const float* packed_bias, const float* residual, int act, float* y,
               float* Vout, int64_t n, int cin, int cout, int mode, void* stream) {
    hipStream_t s = be::as_stream(stream);
    FusedArgs a{V, packed_w, packed_bias, residual, y, Vout, (int)n, cin, cout, (cout + 31) / 32 * 32, act,
                (int)((4 * n + 63) / 64), cout / 64};
    switch (mode) {
        case 1: return launch_fused<true, false, false>(a, s, cin);
        case 2: return launch_fused<false, true, false>(a, s, cin);
        case 3: return launch_fused<true, true, false>(a, s, cin);
        case 4: return launch_fused<false, false, true>(a, s, cin);
        default: return be::fail(BE_EINVAL, "wino_fused: bad mode %d", mode);
    }
}

}  // namespace be
