// k_unit_gemms_sk: the weight-gradient GEMMs and data-gradient convolutions of one or two training units as ONE persistent,
// balanced launch (round 6; the decomposition and why: be_train_sk.h).  local_training.py:103-106.
//
// What differs from k_unit_gemms (be_train.hip) besides the decomposition (each step measured: profiles/r06_sk_timeline.txt):
//   - operands go global -> LDS by DMA (global_load_lds_dwordx4) through a ring of three stages, two chunks in flight, counted
//     vmcnt waits and one raw s_barrier per chunk (the idiom of k_wino_gemm_ws): no staging registers, no ds_write;
//   - every row of an LDS image is a whole 128-byte line of its tensor: the convolution's chunk is 32 floats of K (the first form
//     of this kernel kept k_unit_gemms' 16-float chunks - half lines, 16 lines per 1-KB DMA instruction - and every workgroup of a CU
//     advanced at ~10 B/cycle/CU whatever the ring depth (2 or 5 chunks in flight) and whatever the L2 hit rate (all loads aimed at
//     one chunk: 9 % faster); with whole lines the same launches take 12-25 % less time);
//   - the convolution's image is [row][8 quads] with the quads XOR-swizzled by (row >> 1) & 7 (conflict-free ds_read_b128
//     fragments; the padded rows of the register-staged tiles measured 24 % bank conflicts);
//   - the weight gradient's image is [pixel][128 channels], read by ds_read_b64 (32 lanes x 8 B = all 64 banks once).
// Arithmetic: exact fp32 products on v_mfma_f32_32x32x2_f32, the K order inside a tile is the one of the kernels this replaces
// (convolution: 32-channel chunk outer, tap, 16-float half; weight gradient: image group, pixel row, pixel column); only the
// places where a tile's K loop is cut differ, i.e. the grouping of the fp32 partial sums.
#include "be_common.h"
#include "be_train_sk.h"
#include <cstdlib>
#include <cstring>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

struct ConvProb {
    const float* x;        // [N,H,W,Cin] (the data gradient reads dy)
    const float* w;        // packed [Cout_pad][Ktot] (be_conv_pack_dgrad_f32)
    float* part;           // [slices][M][ldp]
    int H, W, Cin, Ktot, ks, M, ldp, g0, G;
    int rows_w;            // rows of the pack (output channels rounded up to 32): a column tile's rows past them re-read row 0, never used
    be_sk::ConvGeom g;
};
struct WProb {
    const float* x;        // [N,H,W,Cin]
    const float* dy;       // [N,H,W,Cout]
    float* part;           // [slices][tap][Cout][Cin]
    int H, W, Cin, Cout, ks, g0, G;
    be_sk::WGeom g;
};
struct SkArgs {
    WProb w[2];
    ConvProb c[2];
    int nw, nc;
    int prio;              // 1: the convolution workgroups run their MFMA loop at s_setprio 1
    long long* trace;      // diagnostic (BE_SK_TRACE=1): per workgroup {start, end} on the 100 MHz clock, {problem, position, segments, XCC}
};

constexpr int STAGE_C = 4096;          // floats per convolution stage: A 64 x 32 + B 64 x 32
constexpr int STAGE_W = 4096;          // floats per weight-gradient stage: dy 16 x 128 + x 16 x 128
constexpr int SK_LDS_FLOATS = 3 * STAGE_W;   // both rings: three stages, two chunks in flight

// s_waitcnt with only vmcnt counted (expcnt 7, lgkmcnt 15 = no wait): vmcnt(n), n < 16
#define SK_VMCNT(n) __builtin_amdgcn_s_waitcnt(0x0F70 | (n))

// ---- one segment of a convolution tile: chunks [k0, k1) of tile (grp, pp, j) -> slice `slice` ----------------------------------
// A chunk = 32 floats of K = one (32-channel unit, tap) of the pack: every row of the LDS image is ONE 128-byte line of its
// tensor (the 16-float chunks of the first form fetched half lines - 16 lines per 1-KB DMA instruction - and the per-workgroup timeline
// showed every workgroup of a CU advancing at ~10 B/cycle/CU whatever the ring depth or the L2 hit rate).  Image [row][8 quads], quad q
// of row r stored at q ^ ((r >> 1) & 7): the 16 lanes a ds_read_b128 serves together ({0-3,12-15,20-27}, ...) then hit 16 different
// 4-bank groups.  MFMA order per row = two 16-float chunks of the first form back to back: the same chain per output element.
__device__ __forceinline__ void conv_segment(const ConvProb& p, float* smem, const int grp, const int pp, const int j, const int k0,
                                             const int k1, const int slice, const int prio, long long (&ph)[4], const bool stamp) {
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int li = lane & 31, lh = lane >> 5, wm = wave >> 1, wn = wave & 1;
    const int HW = p.g.HW, ks = p.ks, half = ks >> 1, ntap_all = ks * ks;
    const int py = pp / p.W, px = pp - py * p.W;
    unsigned long long tap_list = 0;
    int ntap = 0;
    for (int t = 0; t < ntap_all; ++t) {
        const int yy = py + t / ks - half, xx = px + t % ks - half;
        if ((unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W) { tap_list |= (unsigned long long)t << (4 * ntap); ++ntap; }
    }
    // a DMA piece = 8 rows x 128 B; wave w moves pieces w and w + 4 of both operands.  lane -> (row of the piece, quad slot): the DMA
    // writes lane i's 16 bytes at piece + 16 i; it FETCHES the quad the swizzle puts there
    const int rowp = lane >> 3, slot = lane & 7;
    unsigned a_off[2], b_off[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = 8 * (wave + 4 * i) + rowp;
        const int sq = slot ^ ((row >> 1) & 7);
        a_off[i] = (unsigned)(row * HW * p.Cin + 4 * sq) * 4u;
        const int wrow = j * 64 + row < p.rows_w ? j * 64 + row : 0;            // the empty half of a 96-channel layer's second tile
        b_off[i] = (unsigned)(wrow * p.Ktot + 4 * sq) * 4u;
    }
    const float* xpix = p.x + ((int64_t)grp * 64 * HW + pp) * p.Cin;            // uniform: the tile's pixel of the group's first image
    const float* wt = p.w;
    // walker over the chunks: k = cc * ntap + jj
    int w_cc = k0 / ntap, w_j = k0 - w_cc * ntap, l_buf = 0;
#define SK_CONV_DMA()                                                                                           \
    do {                                                                                                        \
        const int tap_ = (int)((tap_list >> (4 * w_j)) & 15ull);                                                \
        const int ty_ = tap_ / ks, tx_ = tap_ - ty_ * ks;                                                       \
        const char* pa_ = reinterpret_cast<const char*>(xpix + ((ty_ - half) * p.W + tx_ - half) * p.Cin + w_cc * 32); \
        const char* pb_ = reinterpret_cast<const char*>(wt + (w_cc * ntap_all + tap_) * 32);                    \
        float* st_ = smem + l_buf * STAGE_C + wave * 256;                                                       \
        _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {                                                      \
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(pa_ + a_off[i_]), (lds_ptr_t)(st_ + i_ * 1024), 16, 0, 0); \
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(pb_ + b_off[i_]), (lds_ptr_t)(st_ + 2048 + i_ * 1024), 16, 0, 0); \
        }                                                                                                       \
        if (++w_j == ntap) { w_j = 0; ++w_cc; }                                                                 \
        l_buf = l_buf == 2 ? 0 : l_buf + 1;                                                                     \
    } while (0)

    const int fsw = (li >> 1) & 7;
    const int a_fr = (wm * 32 + li) * 32, b_fr = 2048 + (wn * 32 + li) * 32;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;

    const long long c0 = stamp ? (long long)__builtin_amdgcn_s_memtime() : 0;
    SK_VMCNT(0);                                   // the previous segment's stores and DMAs of THIS wave are done ...
    __builtin_amdgcn_s_barrier();                  // ... and every wave has left the previous segment's LDS
    const long long c1 = stamp ? (long long)__builtin_amdgcn_s_memtime() : 0;
    if (prio) __builtin_amdgcn_s_setprio(1);       // short MFMA bursts between waits: take the pipe when ready (the weight gradients fill the rest)
    SK_CONV_DMA();
    if (k0 + 1 < k1) SK_CONV_DMA();
    int r_buf = 0;
    const long long c2 = stamp ? (long long)__builtin_amdgcn_s_memtime() : 0;
    for (int k = k0; k < k1; ++k) {
        // chunk k has landed: all but this wave's newest four DMAs (chunk k + 1, when there is one) are done
        if (k + 1 < k1) SK_VMCNT(4); else SK_VMCNT(0);
        __builtin_amdgcn_s_barrier();              // every wave's pieces of chunk k are in LDS; the stage of chunk k - 1 is free
        if (k + 2 < k1) SK_CONV_DMA();
        __builtin_amdgcn_sched_barrier(0);
        {
            const float* sb = smem + r_buf * STAGE_C;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int fq = 4 * ((lh + 2 * g) ^ fsw);
                const f32x4 av = *reinterpret_cast<const f32x4*>(sb + a_fr + fq), bv = *reinterpret_cast<const f32x4*>(sb + b_fr + fq);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc, 0, 0, 0);
            }
        }
        r_buf = r_buf == 2 ? 0 : r_buf + 1;
        __builtin_amdgcn_sched_barrier(0);
    }
    if (prio) __builtin_amdgcn_s_setprio(0);
    const long long c3 = stamp ? (long long)__builtin_amdgcn_s_memtime() : 0;
#undef SK_CONV_DMA
    // raw partial sums -> slice: D[row][col], col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5); row = image of the group
    const int64_t rs = (int64_t)HW * p.ldp;        // floats between the same pixel of consecutive images
    float* P = p.part + ((int64_t)slice * p.M + (int64_t)grp * 64 * HW + pp) * p.ldp + j * 64 + wn * 32 + li;
    P += (int64_t)(wm * 32 + 4 * lh) * rs;
#pragma unroll
    for (int r = 0; r < 16; ++r) P[(int64_t)((r & 3) + 8 * (r >> 2)) * rs] = acc[r];
    if (stamp) {        // diagnostic: cycles of {entry wait, prologue DMA issue, K loop, stores issued} summed over the workgroup's segments
        const long long c4 = (long long)__builtin_amdgcn_s_memtime();
        ph[0] += c1 - c0; ph[1] += c2 - c1; ph[2] += c3 - c2; ph[3] += c4 - c3;
    }
}

// ---- one segment of a weight-gradient tile: valid pixels [v0, v1) of tile (tap, j) -> slice `slice` ----------------------------
// Tile TM cout x TN cin, TM, TN in {128, 64} (128 wherever the channel count is a multiple of 128; 64 for layer0's 64 / 96 channels
// and layer1's 96 inputs: the second tile of a 96-channel side is half empty - its waves move their share of the operands and
// issue no MFMA).  A wave owns (TM / 2) x (TN / 2): with 64 rows / columns lane li holds channels 2 li and 2 li + 1 (one ds_read_b64
// feeds two MFMA tiles), with 32 it holds channel li.  Lanes whose channels lie past the tensor's last channel re-read its last
// quad (never stored): no access leaves the tensor.
template <int TM, int TN>
__device__ __forceinline__ void wgrad_segment(const WProb& p, float* smem, const int tap, const int j, const int v0, const int v1,
                                              const int slice) {
    constexpr int MI = TM / 64, NI = TN / 64;                  // MFMA tiles per wave along cout / cin
    constexpr int PA = TM / 16, PB = TN / 16, NPW = (PA + PB) / 4;   // 1-KB DMA pieces of dy / x per chunk; pieces per wave
    constexpr int RA = 256 / TM, RB = 256 / TN;                // image rows per piece
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int li = lane & 31, lh = lane >> 5, wm = wave >> 1, wn = wave & 1;
    const int HW = p.H * p.W, half = p.ks >> 1;
    const int tdy = tap / p.ks - half, tdx = tap % p.ks - half;
    int y0, hv, x0, wv;
    be_sk::tap_rect(p.H, p.W, tdy, tdx, y0, hv, x0, wv);
    const int co0 = (j / p.g.cin_tiles) * TM, ci0 = (j % p.g.cin_tiles) * TN;
    const bool live = co0 + wm * (TM / 2) < p.Cout && ci0 + wn * (TN / 2) < p.Cin;
    // piece pi = wave + 4 i: the first PA pieces are dy's (PA is a multiple of 4: a given i is the same operand for every wave)
    unsigned vo[NPW];
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
        const int pi = wave + 4 * i;
        if (pi < PA) {
            const int row = RA * pi + lane / (TM / 4), c = min(co0 + 4 * (lane % (TM / 4)), p.Cout - 4);
            vo[i] = (unsigned)(row * HW * p.Cout + c) * 4u;
        } else {
            const int row = RB * (pi - PA) + lane / (TN / 4), c = min(ci0 + 4 * (lane % (TN / 4)), p.Cin - 4);
            vo[i] = (unsigned)(row * HW * p.Cin + c) * 4u;
        }
    }
    const float* xb = p.x + (int64_t)(tdy * p.W + tdx) * p.Cin;
    // walker: v = (ic * hv + ry) * wv + rx
    int w_ic = v0 / (hv * wv), w_ry, w_rx, l_buf = 0;
    { const int rem = v0 - w_ic * hv * wv; w_ry = rem / wv; w_rx = rem - w_ry * wv; }
#define SK_W_DMA()                                                                                              \
    do {                                                                                                        \
        const int64_t m0_ = (int64_t)w_ic * 16 * HW + (y0 + w_ry) * p.W + x0 + w_rx;                            \
        const char* pa_ = reinterpret_cast<const char*>(p.dy + m0_ * p.Cout);                                   \
        const char* pb_ = reinterpret_cast<const char*>(xb + m0_ * p.Cin);                                      \
        float* st_ = smem + l_buf * STAGE_W + wave * 256;                                                       \
        _Pragma("unroll") for (int i_ = 0; i_ < NPW; ++i_)                                                      \
            __builtin_amdgcn_global_load_lds((glb_ptr_t)((4 * i_ < PA ? pa_ : pb_) + vo[i_]), (lds_ptr_t)(st_ + i_ * 1024), 16, 0, 0); \
        if (++w_rx == wv) { w_rx = 0; if (++w_ry == hv) { w_ry = 0; ++w_ic; } }                                 \
        l_buf = l_buf == 2 ? 0 : l_buf + 1;                                                                     \
    } while (0)

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int jj = 0; jj < NI; ++jj)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][jj][r] = 0.f;

    SK_VMCNT(0);
    __builtin_amdgcn_s_barrier();
    SK_W_DMA();
    if (v0 + 1 < v1) SK_W_DMA();
    int r_buf = 0;
    // stage image: dy [16][TM] at 0, x [16][TN] at 16 TM (pieces in order: the DMA of piece pi lands at 256 pi floats)
    const int a_fr = wm * (TM / 2) + MI * li, b_fr = 16 * TM + wn * (TN / 2) + NI * li;
    for (int v = v0; v < v1; ++v) {
        if (v + 1 < v1) SK_VMCNT(NPW); else SK_VMCNT(0);
        __builtin_amdgcn_s_barrier();
        if (v + 2 < v1) SK_W_DMA();
        __builtin_amdgcn_sched_barrier(0);
        if (live) {
            const float* sb = smem + r_buf * STAGE_W;
#pragma unroll
            for (int s2 = 0; s2 < 8; ++s2) {
                float av[MI], bv[NI];
                if (MI == 2) { const f32x2 t = *reinterpret_cast<const f32x2*>(sb + (2 * s2 + lh) * TM + a_fr); av[0] = t[0]; av[MI - 1] = t[1]; }
                else av[0] = sb[(2 * s2 + lh) * TM + a_fr];
                if (NI == 2) { const f32x2 t = *reinterpret_cast<const f32x2*>(sb + (2 * s2 + lh) * TN + b_fr); bv[0] = t[0]; bv[NI - 1] = t[1]; }
                else bv[0] = sb[(2 * s2 + lh) * TN + b_fr];
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int jj = 0; jj < NI; ++jj) acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[jj], acc[i][jj], 0, 0, 0);
            }
        }
        r_buf = r_buf == 2 ? 0 : r_buf + 1;
        __builtin_amdgcn_sched_barrier(0);
    }
#undef SK_W_DMA
    if (!live) return;
    // D tile (i, jj): row (e & 3) + 8 (e >> 2) + 4 lh = position ii of the wave's rows -> co = co0 + wm TM/2 + MI ii + i; column li ->
    // ci = ci0 + wn TN/2 + NI li + jj (NI = 2: the pair jj = 0, 1 is one 8-byte store)
    const int co_w = co0 + wm * (TM / 2) + MI * 4 * lh, ci_l = ci0 + wn * (TN / 2) + NI * li;
    float* out = p.part + ((int64_t)slice * p.g.ntaps + tap) * p.Cout * p.Cin + (int64_t)co_w * p.Cin + ci_l;
    if (ci_l < p.Cin) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int ro = MI * ((e & 3) + 8 * (e >> 2)) + i;
                if (co_w + ro < p.Cout) {
                    if (NI == 2) { f32x2 v = {acc[i][0][e], acc[i][NI - 1][e]}; *reinterpret_cast<f32x2*>(out + (int64_t)ro * p.Cin) = v; }
                    else out[(int64_t)ro * p.Cin] = acc[i][0][e];
                }
            }
    }
}

// workgroup -> (problem, position g on the problem's axis).  Ids congruent mod 8 share an XCD (speed only): the workgroups of
// one XCD take CONSECUTIVE quotas, so an XCD's L2 sees an eighth of the problem's tiles, not all of them.
__global__ __launch_bounds__(256, 3)
void k_unit_gemms_sk(SkArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int b = blockIdx.x;
    const long long t_start = a.trace ? (long long)__builtin_amdgcn_s_memrealtime() : 0;
    const long long c_start = a.trace ? (long long)__builtin_amdgcn_s_memtime() : 0;     // shader cycles: with t_start / t_end the workgroup's own clock
    int n_seg = 0;
    long long ph[4] = {0, 0, 0, 0};
#define SK_TRACE_END(PROB, G_)                                                                                  \
    if (a.trace && threadIdx.x == 0) {                                                                          \
        a.trace[8 * b] = t_start; a.trace[8 * b + 1] = (long long)__builtin_amdgcn_s_memrealtime();             \
        a.trace[8 * b + 2] = (PROB) | ((long long)(G_) << 8) | ((long long)n_seg << 32);                        \
        unsigned xcc_; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_));                      \
        a.trace[8 * b + 3] = (long long)(xcc_ & 15u) | (((long long)__builtin_amdgcn_s_memtime() - c_start) << 8);   \
        for (int q_ = 0; q_ < 4; ++q_) a.trace[8 * b + 4 + q_] = ph[q_];                                        \
    }
    for (int i = 0; i < a.nw; ++i) {
        const WProb& p = a.w[i];
        if (b < p.g0 || b >= p.g0 + p.G) continue;
        const int bl = b - p.g0, g = (bl & 7) * (p.G >> 3) + (bl >> 3);
        int pos = g * p.g.Q;
        const int end = min(p.g.L, pos + p.g.Q);
        if (pos >= end) return;
        int tap, j;
        be_sk::w_find(p.g, pos, tap, j);
        while (pos < end) {
            int ts, n;
            be_sk::w_span(p.g, tap, j, ts, n);
            const int v0 = pos - ts, v1 = min(n, end - ts);
            const int sl = pos / p.g.Q - ts / p.g.Q;
            if (p.g.tm == 128) { if (p.g.tn == 128) wgrad_segment<128, 128>(p, smem, tap, j, v0, v1, sl); else wgrad_segment<128, 64>(p, smem, tap, j, v0, v1, sl); }
            else { if (p.g.tn == 128) wgrad_segment<64, 128>(p, smem, tap, j, v0, v1, sl); else wgrad_segment<64, 64>(p, smem, tap, j, v0, v1, sl); }
            pos = ts + v1;
            ++n_seg;
            if (++j == p.g.wx) { j = 0; ++tap; }
        }
        SK_TRACE_END(i, g)
        return;
    }
    for (int i = 0; i < a.nc; ++i) {
        const ConvProb& p = a.c[i];
        if (b < p.g0 || b >= p.g0 + p.G) continue;
        const int bl = b - p.g0, g = (bl & 7) * (p.G >> 3) + (bl >> 3);
        int pos = g * p.g.Q;
        const int end = min(p.g.L, pos + p.g.Q);
        if (pos >= end) return;
        int grp, pp, j;
        be_sk::conv_find(p.g, pos, grp, pp, j);
        while (pos < end) {
            int ts, n;
            be_sk::conv_span(p.g, grp, pp, j, ts, n);
            const int k0 = pos - ts, k1 = min(n, end - ts);
            conv_segment(p, smem, grp, pp, j, k0, k1, pos / p.g.Q - ts / p.g.Q, a.prio, ph, a.trace != nullptr);
            pos = ts + k1;
            ++n_seg;
            if (++j == p.g.n_tiles) { j = 0; if (++pp == p.g.HW) { pp = 0; ++grp; } }
        }
        SK_TRACE_END(2 + i, g)
        return;
    }
}

double env_num(const char* name, double dflt) {
    const char* v = getenv(name);
    return v && *v ? atof(v) : dflt;
}

struct Prob {            // planner's view of a problem
    int L, nmax, tiles;  // chunks, longest tile, tiles
    double w, f;         // cost of a chunk, fixed cost of a segment (in convolution chunks)
    int smax_buf;        // slices the scratch region holds
    int smax;            // slices per tile the consumer takes
    int G, Q;
};

// slices of the worst tile under quota Q (scan of the axis: a few hundred tiles)
template <class SpanOfTile>
int worst_slices(int tiles, int Q, SpanOfTile span) {
    int worst = 0;
    for (int t = 0; t < tiles; ++t) {
        int ts, n;
        span(t, ts, n);
        const int s = be_sk::slices_of(ts, n, Q);
        if (s > worst) worst = s;
    }
    return worst;
}

// Workgroups per problem: proportional to its cost (chunks x weight + segments x fixed cost), multiples of 8, capped so that no
// tile is cut into more slices than the consumers take / the scratch holds.  false: a problem's scratch holds fewer than two slices.
bool share_workgroups(Prob* pr, int np) {
    // ---- workgroups per problem: proportional to its cost (chunks x weight + segments x fixed cost), multiples of 8, capped so
    //      that no tile is cut into more slices than the consumers take / the scratch holds
    const int slots_per_cu = (int)env_num("BE_SK_SLOTS", 3.0);
    const int G_total = (be::device_cu_count() * (slots_per_cu < 1 ? 1 : slots_per_cu)) & ~7;
    int cap[4];
    for (int i = 0; i < np; ++i) {
        const int smax = pr[i].smax_buf < pr[i].smax ? pr[i].smax_buf : pr[i].smax;
        if (smax < 2) return false;
        // a tile of n chunks cut by quotas of Q has at most ceil(n / Q) + 1 slices: Q >= ceil(nmax / (smax - 1)) keeps it <= smax
        const int qmin = (pr[i].nmax + smax - 2) / (smax - 1);
        cap[i] = (pr[i].L / (qmin < 4 ? 4 : qmin)) & ~7;
        if (cap[i] < 8) cap[i] = 8;
    }
    double share[4];
    bool fixed[4] = {false, false, false, false};
    int left = G_total;
    for (int round = 0; round < np; ++round) {          // water-filling: capped problems keep their cap, the rest share what is left
        double tot = 0.0;
        for (int i = 0; i < np; ++i)
            if (!fixed[i]) tot += share[i] = pr[i].L * pr[i].w + (pr[i].tiles + (double)G_total / np) * pr[i].f;
        bool again = false;
        for (int i = 0; i < np; ++i) {
            if (fixed[i]) continue;
            int g = (int)(left * share[i] / tot + 4.0) & ~7;
            if (g < 8) g = 8;
            if (g >= cap[i]) { pr[i].G = cap[i]; fixed[i] = true; left -= cap[i]; again = true; }
            else pr[i].G = g;
        }
        if (!again) break;
    }
    {   // the rounding may leave the sum a few workgroups off G_total: give / take them at the largest uncapped problem
        int sum = 0, big = -1;
        for (int i = 0; i < np; ++i) { sum += pr[i].G; if (!fixed[i] && (big < 0 || pr[i].G > pr[big].G)) big = i; }
        if (big >= 0 && sum != G_total && pr[big].G + (G_total - sum) >= 8 && pr[big].G + (G_total - sum) <= cap[big]) pr[big].G += G_total - sum;
    }
    return true;
}

// fills a convolution problem's geometry (tile lengths) for [n,h,w,cin] x ks -> n_tiles column tiles of 64; returns the longest tile's taps
int conv_geometry(be_sk::ConvGeom& g, int n, int h, int w, int cin, int ks, int n_tiles) {
    const int HW = h * w, half = ks >> 1, taps = ks * ks;
    g.HW = HW; g.n_tiles = n_tiles; g.kmul = cin / 32; g.ngrp = n / 64;
    g.PP[0] = 0;
    int tmax = 0;
    for (int pp = 0; pp < HW; ++pp) {
        const int py = pp / w, px = pp % w;
        int nt = 0;
        for (int t = 0; t < taps; ++t)
            nt += (unsigned)(py + t / ks - half) < (unsigned)h && (unsigned)(px + t % ks - half) < (unsigned)w;
        g.PP[pp + 1] = (unsigned short)(g.PP[pp] + nt);
        if (nt > tmax) tmax = nt;
    }
    g.L = g.ngrp * (int)g.PP[HW] * g.n_tiles * g.kmul;
    return tmax;
}

// quota of a convolution problem with pc.G workgroups; lowers G until no tile has more slices than allowed.  false: impossible
bool settle_conv(Prob& pc, be_sk::ConvGeom& g) {
    for (;;) {
        pc.Q = (pc.L + pc.G - 1) / pc.G;
        g.Q = pc.Q;
        const int per_grp = g.HW * g.n_tiles;
        const int worst = worst_slices(pc.tiles, pc.Q, [&](int t, int& ts, int& n) {
            be_sk::conv_span(g, t / per_grp, (t % per_grp) / g.n_tiles, t % g.n_tiles, ts, n); });
        if (worst <= pc.smax && worst <= pc.smax_buf) return true;
        if (pc.G <= 8) return false;
        pc.G -= 8;
    }
}

}  // namespace

namespace be {

bool sk_enabled() {
    static const bool off = getenv("BE_NO_TRAIN_SK") != nullptr;                  // A/B knob: rounds 3-5's k_unit_gemms
    return !off;
}

// Units this launch takes: 1x1 / 3x3 convolutions of whole 64-image groups on maps of at most 11 x 11, channel counts multiples of
// 32 and at least 64 (every unit of LocalStage's layers 0-3 and fc.1; not conv1: 3 input channels, no input gradient), input
// gradient wanted.
bool sk_eligible(const be_train_unit_bwd& u) {
    const be_conv_desc& d = u.desc;
    if (d.ksize != 1 && d.ksize != 3) return false;
    if (d.n < 64 || d.n % 64 || d.h * d.w > be_sk::MAX_HW || (d.ksize == 3 && (d.h < 3 || d.w < 3))) return false;
    if (d.cout % 32 || d.cin % 32 || d.cout < 64 || d.cin < 64 || !u.dx || !u.dgrad_packed_w) return false;
    // fc.1 (a linear on flattened (h, w, c) features whose weight is stored in (c, h, w) column order): rows are images, k_bwd_post
    // permutes the columns while it sums the slices
    if (u.layout_chw_hw && !(d.ksize == 1 && d.h * d.w == 1 && d.cin % u.layout_chw_hw == 0)) return false;
    const int64_t M = (int64_t)d.n * d.h * d.w;
    const int cmax = d.cout > d.cin ? d.cout : d.cin;
    if (M * cmax * 4 >= ((int64_t)1 << 31) || (int64_t)d.cin * d.cout * d.ksize * d.ksize * 4 >= ((int64_t)1 << 31)) return false;
    return true;
}

// Plans k_unit_gemms_sk for nu (1 or 2) eligible units: nothing is launched.  Returns BE_OK, or 1 = "not mine" (scratch too small
// for even two slices per tile): the caller then takes the old path.
struct SkPlanData { SkArgs a; int grid; double flops, flops_exec; };
static_assert(sizeof(SkPlanData) <= sizeof(SkPlan), "SkPlan too small");
int sk_plan(const SkUnitIn* in, int nu, SkUnitOut* out, SkPlan* plan) {
    SkPlanData& pd = *reinterpret_cast<SkPlanData*>(plan->blob);
    SkArgs& a = pd.a;
    memset(&pd, 0, sizeof(pd));
    Prob pr[4];
    int np = 0;
    // Cost model, in convolution chunks (64 x 64 x 32: 16 MFMAs per wave).  A weight-gradient chunk (128 x 128 x 16) is 32 MFMAs per
    // wave, twice the matrix work - but what a workgroup's chunk COSTS on a CU shared by three workgroups is its serial stream (wait,
    // barrier, DMA issue, LDS latency, then the MFMAs), and the short chunk carries relatively more of the rest: the per-workgroup
    // timelines (tools/sk_trace.py, profiles/r06_sk_timeline.txt) put the two kinds' median lifetimes level between 1.5 and 2.
    // Within that range the step time is set by WHERE the quotas cut the tiles (how many tiles end up in two slices), not by the
    // balance: the sweeps of the graph-replayed step (profiles/r06_sk_sweep.txt, two boxes, repeated) are flat within +-12 us from
    // 1 to 2.5 except for one sharp, repeatable optimum at weight 1 with no fixed cost per segment (1.315 ms against 1.335-1.36),
    // which is the default: there a 3x3 unit with equal channel counts splits its workgroups 256 / 512, which makes BOTH quotas 36
    // chunks at 384 channels (16 at 256) - a divisor of the interior pixels' tiles (108, 72) and of the centre tap's (144): 1.03 /
    // 1.2 segments per workgroup, almost no tile cut twice.  (Snapping every problem's quota to a divisor of its longest tile - fewer tiles in two slices - was built
    // and measured: 1.356-1.388 ms against 1.320-1.360 without, because tile STARTS stay unaligned in the natural tile order and the
    // snapped counts leave slots empty; removed.)  Environment overrides for re-tuning.
    const double ww = env_num("BE_SK_WW", 1.0), fw = env_num("BE_SK_FW", 0.0), fc = env_num("BE_SK_FC", 0.0);
    double flops = 0.0, flops_exec = 0.0;
    a.prio = (int)env_num("BE_SK_PRIO", 1.0);
    static const bool trace = getenv("BE_SK_TRACE") != nullptr;       // diagnostic: tools/sk_trace.py reads the stamps back
    size_t w_reserve = 0;
    if (trace) {        // the stamps go to the END of the last unit's weight-gradient region (the planner is told it is shorter)
        w_reserve = 65536;
        a.trace = reinterpret_cast<long long*>(reinterpret_cast<char*>(in[nu - 1].wpart) + in[nu - 1].wpart_bytes - w_reserve);
    }
    for (int i = 0; i < nu; ++i) {
        const be_train_unit_bwd& u = *in[i].u;
        const be_conv_desc& d = u.desc;
        const int HW = d.h * d.w, ks = d.ksize, half = ks >> 1, taps = ks * ks, M = d.n * HW;
        // weight gradient
        WProb& w = a.w[i];
        w.x = u.x; w.dy = u.dy; w.part = in[i].wpart; w.H = d.h; w.W = d.w; w.Cin = d.cin; w.Cout = d.cout; w.ks = ks;
        w.g.tm = d.cout % 128 == 0 ? 128 : 64; w.g.tn = d.cin % 128 == 0 ? 128 : 64; w.g.cin = d.cin;
        w.g.cin_tiles = (d.cin + w.g.tn - 1) / w.g.tn; w.g.wx = ((d.cout + w.g.tm - 1) / w.g.tm) * w.g.cin_tiles; w.g.ntaps = taps;
        w.g.PT[0] = 0;
        int nmax = 0;
        for (int t = 0; t < taps; ++t) {
            int y0, hv, x0, wv;
            be_sk::tap_rect(d.h, d.w, t / ks - half, t % ks - half, y0, hv, x0, wv);
            const int n = (d.n / 16) * hv * wv;
            w.g.PT[t + 1] = w.g.PT[t] + n;
            if (n > nmax) nmax = n;
        }
        w.g.L = w.g.PT[taps] * w.g.wx;
        const size_t wsize = (size_t)d.cout * d.cin * taps * sizeof(float);
        // a chunk of a tm x tn tile is tm tn / 16384 of the 128 x 128 chunk's MFMAs; the small tiles carry relatively more of the rest
        const double wtile = ww * (w.g.tm * w.g.tn == 16384 ? 1.0 : (w.g.tm * w.g.tn == 8192 ? env_num("BE_SK_W8K", 0.6) : env_num("BE_SK_W4K", 0.4)));
        pr[np++] = Prob{w.g.L, nmax, w.g.wx * taps, wtile, fw, (int)((in[i].wpart_bytes - (i == nu - 1 ? w_reserve : 0)) / wsize), be_sk::MAX_SLICES_W, 0, 0};
        // data gradient: a convolution of dy [M, Cout] with the transposed, tap-mirrored pack -> [M, Cin]
        ConvProb& c = a.c[i];
        c.x = u.dy; c.w = u.dgrad_packed_w; c.part = in[i].cpart; c.H = d.h; c.W = d.w; c.Cin = d.cout; c.ks = ks;
        c.Ktot = (d.cout / 32) * taps * 32; c.M = M; c.ldp = (d.cin + 63) / 64 * 64; c.rows_w = (d.cin + 31) / 32 * 32;
        const int tmax = conv_geometry(c.g, d.n, d.h, d.w, d.cout, ks, (d.cin + 63) / 64);
        const size_t csize = (size_t)M * c.ldp * sizeof(float);
        pr[np++] = Prob{c.g.L, tmax * c.g.kmul, c.g.ngrp * HW * c.g.n_tiles, 1.0, fc, (int)(in[i].cpart_bytes / csize), be_sk::MAX_SLICES_C, 0, 0};
        flops += 4.0 * M * (double)d.cin * d.cout * taps;
        flops_exec += (double)w.g.L * 2.0 * w.g.tm * w.g.tn * 16 + (double)c.g.L * 2.0 * 64 * 64 * 32;
        out[i].ldp = c.ldp;
    }
    if (!share_workgroups(pr, np)) return 1;
    int g0 = 0;
    for (int i = 0; i < nu; ++i) {
        Prob& pw = pr[2 * i];
        Prob& pc = pr[2 * i + 1];
        WProb& w = a.w[i];
        ConvProb& c = a.c[i];
        for (;;) {          // quota, then the real slice count of the worst tile (defensive: the cap above already bounds it)
            pw.Q = (pw.L + pw.G - 1) / pw.G;
            w.g.Q = pw.Q;
            const int worst = worst_slices(pw.tiles, pw.Q, [&](int t, int& ts, int& n) { be_sk::w_span(w.g, t / w.g.wx, t % w.g.wx, ts, n); });
            if (worst <= pw.smax && worst <= pw.smax_buf) break;
            if (pw.G <= 8) return 1;
            pw.G -= 8;
        }
        if (!settle_conv(pc, c.g)) return 1;
        out[i].cg = c.g; out[i].wg = w.g;
    }
    for (int i = 0; i < nu; ++i) { a.w[i].g0 = g0; a.w[i].G = pr[2 * i].G; g0 += pr[2 * i].G; }       // the long chunks first
    for (int i = 0; i < nu; ++i) { a.c[i].g0 = g0; a.c[i].G = pr[2 * i + 1].G; g0 += pr[2 * i + 1].G; }
    a.nw = nu; a.nc = nu;
    pd.grid = g0; pd.flops = flops; pd.flops_exec = flops_exec;
    return BE_OK;
}


// Forward convolutions of one or two training units (models/local_stage.py:11-17 in train mode; the forward half of
// local_training.py:103) on the same persistent launch: conv-only problems, raw slices for k_bn_stats.
bool sk_fwd_eligible(const be_conv_desc& d) {
    if (d.ksize != 1 && d.ksize != 3) return false;
    // a 3x3 kernel needs a map of at least 3 x 3 (every tap then has a pixel: no empty tile on the axis); a 1x1 takes any map,
    // h = w = 1 included: fc.1 (64 rows x 2304 -> 1024: sixteen tiles of 72 chunks that no equal-slice grid fills)
    if (d.n < 64 || d.n % 64 || d.h * d.w > be_sk::MAX_HW || (d.ksize == 3 && (d.h < 3 || d.w < 3))) return false;
    if (d.cout % 32 || d.cout < 64 || d.cin % 32) return false;    // 96 outputs: two column tiles, the second half empty
    const int64_t M = (int64_t)d.n * d.h * d.w;
    const int cmax = d.cout > d.cin ? d.cout : d.cin;
    return M * cmax * 4 < ((int64_t)1 << 31) && (int64_t)d.cin * d.cout * d.ksize * d.ksize * 4 < ((int64_t)1 << 31);
}

int sk_plan_fwd(const SkFwdIn* in, int nu, be_sk::ConvGeom* out, SkPlan* plan) {
    SkPlanData& pd = *reinterpret_cast<SkPlanData*>(plan->blob);
    SkArgs& a = pd.a;
    memset(&pd, 0, sizeof(pd));
    Prob pr[4];
    const double fc = env_num("BE_SK_FC", 0.0);
    a.prio = 0;
    for (int i = 0; i < nu; ++i) {
        const be_conv_desc& d = *in[i].d;
        const int ks = d.ksize, taps = ks * ks, M = d.n * d.h * d.w;
        ConvProb& c = a.c[i];
        c.x = in[i].x; c.w = in[i].packed_w; c.part = in[i].part; c.H = d.h; c.W = d.w; c.Cin = d.cin; c.ks = ks;
        c.Ktot = (d.cin / 32) * taps * 32; c.M = M; c.ldp = (d.cout + 63) / 64 * 64; c.rows_w = (d.cout + 31) / 32 * 32;
        const int tmax = conv_geometry(c.g, d.n, d.h, d.w, d.cin, ks, (d.cout + 63) / 64);
        const size_t csize = (size_t)M * c.ldp * sizeof(float);
        pr[i] = Prob{c.g.L, tmax * c.g.kmul, c.g.ngrp * c.g.HW * c.g.n_tiles, 1.0, fc, (int)(in[i].part_bytes / csize), be_sk::MAX_SLICES_C, 0, 0};
        pd.flops += 2.0 * M * (double)d.cin * d.cout * taps;
        pd.flops_exec += (double)c.g.L * 2.0 * 64 * 64 * 32;
    }
    if (!share_workgroups(pr, nu)) return 1;
    int g0 = 0;
    for (int i = 0; i < nu; ++i) {
        if (!settle_conv(pr[i], a.c[i].g)) return 1;
        a.c[i].g0 = g0; a.c[i].G = pr[i].G; g0 += pr[i].G;
        out[i] = a.c[i].g;
    }
    a.nw = 0; a.nc = nu;
    pd.grid = g0;
    return BE_OK;
}

int sk_run(const SkPlan* plan, hipStream_t s) {
    const SkPlanData& pd = *reinterpret_cast<const SkPlanData*>(plan->blob);
    constexpr size_t lds = (size_t)SK_LDS_FLOATS * sizeof(float);
    static be::DeviceFlags flags{};
    if (int rc_ = be::ensure_dynamic_lds(reinterpret_cast<const void*>(&k_unit_gemms_sk), lds, flags)) return rc_;
    {
        be::ProfileScope prof(s, BE_KERNEL_TRAIN_BWD_GEMMS, pd.flops, 0.0, pd.flops_exec);
        hipLaunchKernelGGL(k_unit_gemms_sk, dim3(pd.grid), dim3(256), lds, s, pd.a);
    }
    return be::check_launch("k_unit_gemms_sk");
}

}  // namespace be

// Test / inspection hook (CPU, no launch): the plan k_unit_gemms_sk would run for ONE unit [n,h,w,cin] -> cout, ksize 1 | 3 with
// `workgroups` slots split between its weight gradient and its data gradient in proportion `w_share` (0..1).  Writes, per segment,
// {problem (0 weight gradient, 1 convolution), tile, first chunk, last chunk + 1, slice, slices of the tile} to seg (6 ints each, at
// most cap segments) and returns the segment count (< 0: error).  tests/test_host_cpu.py checks that the segments of every tile
// partition its chunks and that the slice numbers are 0 .. slices - 1 - the arithmetic producer and consumers share.
extern "C" int be_train_sk_plan_debug(int n, int h, int w, int cin, int cout, int ksize, int workgroups, double w_share, int* seg,
                                      int cap) {
    BE_REQUIRE(seg && cap > 0 && n >= 64 && n % 64 == 0 && h >= 3 && w >= 3 && h * w <= be_sk::MAX_HW && (ksize == 1 || ksize == 3) &&
               cin % 128 == 0 && cout % 128 == 0 && workgroups >= 16 && w_share > 0.0 && w_share < 1.0, "be_train_sk_plan_debug: bad arguments");
    const int ks = ksize, half = ks >> 1, taps = ks * ks, HW = h * w;
    be_sk::WGeom wg{};
    wg.tm = wg.tn = 128; wg.cin = cin; wg.cin_tiles = cin / 128; wg.wx = (cout / 128) * (cin / 128); wg.ntaps = taps;
    for (int t = 0; t < taps; ++t) {
        int y0, hv, x0, wv;
        be_sk::tap_rect(h, w, t / ks - half, t % ks - half, y0, hv, x0, wv);
        wg.PT[t + 1] = wg.PT[t] + (n / 16) * hv * wv;
    }
    wg.L = wg.PT[taps] * wg.wx;
    be_sk::ConvGeom cg{};
    cg.HW = HW; cg.n_tiles = cin / 64; cg.kmul = cout / 32; cg.ngrp = n / 64;
    for (int pp = 0; pp < HW; ++pp) {
        int nt = 0;
        for (int t = 0; t < taps; ++t)
            nt += (unsigned)(pp / w + t / ks - half) < (unsigned)h && (unsigned)(pp % w + t % ks - half) < (unsigned)w;
        cg.PP[pp + 1] = (unsigned short)(cg.PP[pp] + nt);
    }
    cg.L = cg.ngrp * (int)cg.PP[HW] * cg.n_tiles * cg.kmul;
    int Gw = (int)(workgroups * w_share) & ~7, Gc = (workgroups - Gw) & ~7;
    if (Gw < 8) Gw = 8;
    if (Gc < 8) Gc = 8;
    wg.Q = (wg.L + Gw - 1) / Gw;
    cg.Q = (cg.L + Gc - 1) / Gc;
    int ns = 0;
    auto emit = [&](int prob, int tile, int k0, int k1, int slice, int cnt) {
        if (ns < cap) { int* e = seg + 6 * ns; e[0] = prob; e[1] = tile; e[2] = k0; e[3] = k1; e[4] = slice; e[5] = cnt; }
        ++ns;
    };
    for (int g = 0; g < Gw; ++g) {                      // the walk of k_unit_gemms_sk, weight gradient
        int pos = g * wg.Q;
        const int end = pos + wg.Q < wg.L ? pos + wg.Q : wg.L;
        if (pos >= end) continue;
        int tap, j;
        be_sk::w_find(wg, pos, tap, j);
        while (pos < end) {
            int ts, nn;
            be_sk::w_span(wg, tap, j, ts, nn);
            const int v0 = pos - ts, v1 = nn < end - ts ? nn : end - ts;
            emit(0, tap * wg.wx + j, v0, v1, pos / wg.Q - ts / wg.Q, be_sk::slices_of(ts, nn, wg.Q));
            pos = ts + v1;
            if (++j == wg.wx) { j = 0; ++tap; }
        }
    }
    for (int g = 0; g < Gc; ++g) {                      // ... convolution
        int pos = g * cg.Q;
        const int end = pos + cg.Q < cg.L ? pos + cg.Q : cg.L;
        if (pos >= end) continue;
        int grp, pp, j;
        be_sk::conv_find(cg, pos, grp, pp, j);
        while (pos < end) {
            int ts, nn;
            be_sk::conv_span(cg, grp, pp, j, ts, nn);
            const int k0 = pos - ts, k1 = nn < end - ts ? nn : end - ts;
            emit(1, (grp * HW + pp) * cg.n_tiles + j, k0, k1, pos / cg.Q - ts / cg.Q, be_sk::slices_of(ts, nn, cg.Q));
            pos = ts + k1;
            if (++j == cg.n_tiles) { j = 0; if (++pp == HW) { pp = 0; ++grp; } }
        }
    }
    return ns;
}
