// Lab bench for the 25 transform-domain GEMMs of the Winograd layers (be_wino.hip: k_wino_gemm).  Stand-alone program:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 lab/wino_gemm_lab.hip -o lab/bin/wino_gemm_lab   (or: make -C lab) && ./wino_gemm_lab
// Variants are timed with hipEvents on M = 32768 rows (8192 patches) and checked against variant 0 (the product kernel's
// text).  Knock-outs (wrong results by construction, timing only) are marked KO.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <cmath>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct GemmArgs {
    const float* x;       // [nb][M][K]
    const float* w;       // [nb][Npad][K]
    float* y;             // [nb][M][ldy]
    int M, K, N, ldy, nb, m_tiles, n_tiles;
    int64_t xb, wb, yb;
    long long* dbg;       // [2]: shader-clock cycles and 100 MHz ticks of workgroup 0
    int mgroups;          // WS variants: M-tile groups per (problem, N tile)
};

#define CK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// ---------------------------------------------------------------------------------------------------------------------
// variant family B: the product kernel (workgroup-shared LDS tiles, one barrier per chunk).
//   VAR 0 = product, 1 = KO no barrier, 2 = epilogue stores after the hand-over barrier
template <int VAR, int PF = 1, int NOLOAD = 0, int NOEPI = 0, int PRIO = 0, int KO = 0, int AGPR = 0>   // KO: 1 no barrier, 2 no ds_write, 4 no ds_read
__global__ __launch_bounds__(256, 3)
void k_base(GemmArgs a) {
    if (AGPR) { float z_ = 0.f; asm volatile("; touch an AGPR so the MFMA accumulators are allocated there %0" : "+a"(z_)); }
    if (PRIO) {
        // static, slot-dependent issue priority: the waves that share a SIMD then run their MFMA phases one after the
        // other instead of interleaved (interleaved, they all reach their non-MFMA work at the same time and the pipe idles)
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID, 0, 4)" : "=s"(hw));
        if (blockIdx.x < 4096 && threadIdx.x == 0) a.dbg[2 + blockIdx.x] = hw;
        const unsigned slot3 = hw % 3u;
        if (PRIO == 1) {
            if (slot3 == 0) __builtin_amdgcn_s_setprio(3);
            else if (slot3 == 1) __builtin_amdgcn_s_setprio(2);
            else __builtin_amdgcn_s_setprio(0);
        } else {
            if (hw & 1u) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(0);
        }
    }
    constexpr int BM = 128, BN = 128, BKT = 16, LROW = BKT + 4, RP = 64;
    extern __shared__ __attribute__((aligned(16))) float smem_w[];
    float* As = smem_w;
    float* Bs = smem_w + 2 * BM * LROW;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, slot = bid >> 3;
    const int n_tile = slot % a.n_tiles;
    const int m_tile = (slot / a.n_tiles) * 8 + xcd;
    if (m_tile >= a.m_tiles) return;
    const int n0 = n_tile * BN, row_base = m_tile * BM;
    const int tid = threadIdx.x;
    long long t0 = 0, w0 = 0;
    if (bid == 0 && tid == 0) { t0 = clock64(); w0 = wall_clock64(); }
    const int q = tid & 3, r0 = tid >> 2;
    const float* ap[2];
    const float* bp[2];
    bool a_live[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int ra = row_base + r0 + RP * i;
        a_live[i] = ra < a.M;
        ap[i] = a.x + (size_t)(a_live[i] ? ra : 0) * a.K + 4 * q;
        bp[i] = a.w + (size_t)(n0 + r0 + RP * i) * a.K + 4 * q;
    }
    const int kchunks = a.K / BKT, total = kchunks * a.nb;
    const int wave = tid >> 6, lane = tid & 63;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lh = lane >> 5;
    const int a_frag0 = (wm * 64 + li) * LROW + 4 * lh;
    const int b_frag0 = (wn * 64 + li) * LROW + 4 * lh;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    f32x4 a_st[2][2], b_st[2][2];
    int lz = 0, lk = 0;
#define WG_LOAD(S)                                                                                              \
    do {                                                                                                        \
        const int64_t xo_ = (int64_t)lz * a.xb + lk * BKT, wo_ = (int64_t)lz * a.wb + lk * BKT;                  \
        _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {                                                      \
            a_st[S][i_] = *reinterpret_cast<const f32x4*>(ap[i_] + xo_);                                        \
            b_st[S][i_] = *reinterpret_cast<const f32x4*>(bp[i_] + wo_);                                        \
        }                                                                                                       \
        if (lk + 1 < kchunks) ++lk; else if (lz + 1 < a.nb) { lk = 0; ++lz; }                                   \
    } while (0)
#define WG_STORE(BUF, S)                                                                                        \
    do {                                                                                                        \
        _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {                                                      \
            *reinterpret_cast<f32x4*>(As + (BUF) * BM * LROW + (r0 + RP * i_) * LROW + 4 * q) =                 \
                a_live[i_] ? a_st[S][i_] : f32x4{0.f, 0.f, 0.f, 0.f};                                           \
            *reinterpret_cast<f32x4*>(Bs + (BUF) * BN * LROW + (r0 + RP * i_) * LROW + 4 * q) = b_st[S][i_];    \
        }                                                                                                       \
    } while (0)
#define EPILOGUE()                                                                                              \
    do {                                                                                                        \
        float* yz = a.y + (int64_t)cz * a.yb + (size_t)(row_base + wm * 64 + 4 * lh) * a.ldy + n0 + wn * 64 + li; \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                         \
            const bool c_ok = n0 + wn * 64 + j * 32 + li < a.N;                                                 \
            _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                       \
            _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                    \
                const int ro = i * 32 + (r & 3) + 8 * (r >> 2);                                                 \
                if (c_ok && row_base + wm * 64 + 4 * lh + ro < a.M) yz[(size_t)ro * a.ldy + j * 32] = acc[i][j][r]; \
                acc[i][j][r] = 0.0f;                                                                            \
            }                                                                                                   \
        }                                                                                                       \
        ck = 0; ++cz;                                                                                           \
    } while (0)
#define MFMA_PHASE(BUF)                                                                                         \
    do {                                                                                                        \
        const float* Ab = As + (BUF) * BM * LROW + a_frag0;                                                     \
        const float* Bb = Bs + (BUF) * BN * LROW + b_frag0;                                                     \
        _Pragma("unroll") for (int g = 0; g < BKT / 8; ++g) {                                                   \
            f32x4 af[2], bf[2];                                                                                 \
            if (KO & 4) { af[0] = kaf; af[1] = kaf; bf[0] = kbf; bf[1] = kbf; } else {                           \
            _Pragma("unroll") for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const f32x4*>(Ab + i * 32 * LROW + 8 * g); \
            _Pragma("unroll") for (int j = 0; j < 2; ++j) bf[j] = *reinterpret_cast<const f32x4*>(Bb + j * 32 * LROW + 8 * g); } \
            _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                       \
            _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                     \
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].x, bf[j].x, acc[i][j], 0, 0, 0);         \
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].y, bf[j].y, acc[i][j], 0, 0, 0);         \
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].z, bf[j].z, acc[i][j], 0, 0, 0);         \
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].w, bf[j].w, acc[i][j], 0, 0, 0);         \
            }                                                                                                   \
        }                                                                                                       \
    } while (0)
    int cz = 0, ck = 0;
    f32x4 kaf = {1.f + tid, 2.f, 3.f, 4.f}, kbf = {0.5f, 0.25f * tid, 2.f, 1.f};
    if (PF == 1) {
        WG_LOAD(0);
        WG_STORE(0, 0);
        __syncthreads();
        for (int kc = 0; kc < total; ++kc) {
            const int buf = kc & 1;
            if (!NOLOAD) WG_LOAD(0);                   // past the end: re-reads the last chunk (harmless)
            __builtin_amdgcn_sched_barrier(0);
            MFMA_PHASE(buf);
            __builtin_amdgcn_sched_barrier(0);
            if (VAR != 2) { if (++ck == kchunks) { if (!NOEPI) EPILOGUE(); else { ck = 0; ++cz; } } }
            if (!(KO & 2)) WG_STORE(buf ^ 1, 0);
            if (VAR != 1 && !(KO & 1)) __syncthreads();
            if (VAR == 2) { if (++ck == kchunks) { if (!NOEPI) EPILOGUE(); else { ck = 0; ++cz; } } }
        }
    } else {
        // loads two chunks ahead: staging set s = chunk parity
        WG_LOAD(0);
        WG_STORE(0, 0);
        WG_LOAD(1);
        __syncthreads();
        for (int kc = 0; kc < total; kc += 2) {        // total is even (K / 16 is even for K % 32 == 0)
            WG_LOAD(0);                                // chunk kc + 2
            __builtin_amdgcn_sched_barrier(0);
            MFMA_PHASE(0);
            __builtin_amdgcn_sched_barrier(0);
            WG_STORE(1, 1);                            // chunk kc + 1
            __syncthreads();
            if (++ck == kchunks) EPILOGUE();
            WG_LOAD(1);                                // chunk kc + 3
            __builtin_amdgcn_sched_barrier(0);
            MFMA_PHASE(1);
            __builtin_amdgcn_sched_barrier(0);
            WG_STORE(0, 0);                            // chunk kc + 2
            __syncthreads();
            if (++ck == kchunks) EPILOGUE();
        }
    }
#undef MFMA_PHASE
    if (NOEPI) { cz = 0; EPILOGUE(); }                 // keeps the accumulators alive in the knock-out
    if (bid == 0 && tid == 0) { a.dbg[0] = clock64() - t0; a.dbg[1] = wall_clock64() - w0; }
#undef WG_LOAD
#undef WG_STORE
#undef EPILOGUE
}

// ---------------------------------------------------------------------------------------------------------------------
// variant family W: every wave owns a 64x64 tile and stages its OWN operands through a private LDS region: no workgroup
// barrier anywhere, the MFMA stream of a wave never waits on a sibling.  Single LDS buffer per wave (10 KB): the fragments of
// chunk k are in registers before chunk k+1 overwrites the buffer (LDS instructions of one wave execute in order).
//   per iteration:  ds_read F1(k) | 8 MFMA on F0(k) | ds_write S(k+1), global loads (k+2) -> S | 8 MFMA on F0(k)
//                   | ds_read F0(k+1) | 16 MFMA on F1(k)
template <int VAR>
__global__ __launch_bounds__(256, 3)
void k_wave(GemmArgs a) {
    constexpr int BKT = 16, LROW = BKT + 4;
    extern __shared__ __attribute__((aligned(16))) float smem_w[];
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    float* As = smem_w + wave * (128 * LROW);          // [64][LROW]
    float* Bs = As + 64 * LROW;                        // [64][LROW]
    const int bid = blockIdx.x;
    const int xcd = bid & 7, slot = bid >> 3;
    const int n_tile = slot % a.n_tiles;
    const int m_tile = (slot / a.n_tiles) * 8 + xcd;
    if (m_tile >= a.m_tiles) return;
    const int wm = wave >> 1, wn = wave & 1;
    const int row0 = m_tile * 128 + wm * 64, col0 = n_tile * 128 + wn * 64;
    const int q = lane & 3, r0 = lane >> 2;            // staging: rows r0 + 16 j
    unsigned a_off[4], b_off[4];
    bool a_live[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        a_live[j] = row0 + r0 + 16 * j < a.M;               // row0 itself always exists
        a_off[j] = (unsigned)((a_live[j] ? r0 + 16 * j : 0) * a.K + 4 * q) * 4u;
        b_off[j] = (unsigned)((r0 + 16 * j) * a.K + 4 * q) * 4u;
    }
    const int kchunks = a.K / BKT, total = kchunks * a.nb;
    const int li = lane & 31, lh = lane >> 5;
    const float* Af = As + li * LROW + 4 * lh;
    const float* Bf = Bs + li * LROW + 4 * lh;
    float* Aw = As + r0 * LROW + 4 * q;
    float* Bw = Bs + r0 * LROW + 4 * q;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    f32x4 sa[4], sb[4];
    int lz = 0, lk = 0;
#define W_LOAD()                                                                                                \
    do {                                                                                                        \
        const char* xs_ = reinterpret_cast<const char*>(a.x + (int64_t)lz * a.xb + (int64_t)row0 * a.K + lk * BKT); \
        const char* ws_ = reinterpret_cast<const char*>(a.w + (int64_t)lz * a.wb + (int64_t)col0 * a.K + lk * BKT); \
        _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) {                                                      \
            sa[j_] = *reinterpret_cast<const f32x4*>(xs_ + a_off[j_]);                                          \
            sb[j_] = *reinterpret_cast<const f32x4*>(ws_ + b_off[j_]);                                          \
        }                                                                                                       \
        if (lk + 1 < kchunks) ++lk; else if (lz + 1 < a.nb) { lk = 0; ++lz; }                                  \
    } while (0)
#define W_WRITE()                                                                                               \
    do {                                                                                                        \
        _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) {                                                      \
            *reinterpret_cast<f32x4*>(Aw + 16 * j_ * LROW) = a_live[j_] ? sa[j_] : f32x4{0.f, 0.f, 0.f, 0.f};   \
            *reinterpret_cast<f32x4*>(Bw + 16 * j_ * LROW) = sb[j_];                                            \
        }                                                                                                       \
    } while (0)
#define W_READ(FA, FB, G)                                                                                       \
    do {                                                                                                        \
        _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {                                                      \
            FA[i_] = *reinterpret_cast<const f32x4*>(Af + i_ * 32 * LROW + 8 * (G));                            \
            FB[i_] = *reinterpret_cast<const f32x4*>(Bf + i_ * 32 * LROW + 8 * (G));                            \
        }                                                                                                       \
    } while (0)
#define W_MFMA(FA, FB, I)                                                                                       \
    do {                                                                                                        \
        _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_) {                                                      \
            acc[I][j_] = __builtin_amdgcn_mfma_f32_32x32x2f32(FA[I].x, FB[j_].x, acc[I][j_], 0, 0, 0);          \
            acc[I][j_] = __builtin_amdgcn_mfma_f32_32x32x2f32(FA[I].y, FB[j_].y, acc[I][j_], 0, 0, 0);          \
            acc[I][j_] = __builtin_amdgcn_mfma_f32_32x32x2f32(FA[I].z, FB[j_].z, acc[I][j_], 0, 0, 0);          \
            acc[I][j_] = __builtin_amdgcn_mfma_f32_32x32x2f32(FA[I].w, FB[j_].w, acc[I][j_], 0, 0, 0);          \
        }                                                                                                       \
    } while (0)
    f32x4 fa0[2], fb0[2], fa1[2], fb1[2];
    W_LOAD();
    W_WRITE();
    W_LOAD();
    W_READ(fa0, fb0, 0);
    int cz = 0, ck = 0;
    for (int kc = 0; kc < total; ++kc) {
        // no conditionals in the body: past the end W_LOAD re-reads the last chunk and W_WRITE / W_READ are harmless
        if (VAR == 1) W_READ(fa1, fb1, 1);
        if (VAR == 0) {
            __builtin_amdgcn_sched_barrier(0);
            W_MFMA(fa0, fb0, 0);
            __builtin_amdgcn_sched_barrier(0);
            W_READ(fa1, fb1, 1);
            W_WRITE();
            W_LOAD();
            __builtin_amdgcn_sched_barrier(0);
            W_MFMA(fa0, fb0, 1);
            __builtin_amdgcn_sched_barrier(0);
        } else {
            // one scheduling region: the hand-over instructions ride in the shadow of the 16 MFMAs on F0
            __builtin_amdgcn_sched_barrier(0);
            W_MFMA(fa0, fb0, 0);
            W_WRITE();
            W_LOAD();
            W_MFMA(fa0, fb0, 1);
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // MFMA
                __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);     // DS write
                __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);     // VALU
            }
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // MFMA
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);     // VMEM read
                __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);     // VALU
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        W_READ(fa0, fb0, 0);
        __builtin_amdgcn_sched_barrier(0);
        W_MFMA(fa1, fb1, 0);
        W_MFMA(fa1, fb1, 1);
        __builtin_amdgcn_sched_barrier(0);
        if (++ck == kchunks) {
            float* yz = a.y + (int64_t)cz * a.yb + (size_t)(row0 + 4 * lh) * a.ldy + col0 + li;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const bool c_ok = col0 + j * 32 + li < a.N;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int ro = i * 32 + (r & 3) + 8 * (r >> 2);
                        if (c_ok && row0 + 4 * lh + ro < a.M) yz[(size_t)ro * a.ldy + j * 32] = acc[i][j][r];
                        acc[i][j][r] = 0.0f;
                    }
            }
            ck = 0; ++cz;
        }
    }
#undef W_LOAD
#undef W_WRITE
#undef W_READ
#undef W_MFMA
}

// ---------------------------------------------------------------------------------------------------------------------
// variant family G: operands go global -> LDS directly (global_load_lds_dwordx4, no VGPR round trip, no ds_write), the LDS
// image is lane-linear per 1-KB piece (16 rows x 64 B) with the 16-byte quads of a row XOR-swizzled by (row >> 2) & 3 on
// the SOURCE address and on the fragment read; interior tiles store without per-element bounds checks; a new problem starts
// with C = 0 in its first MFMAs instead of clearing 64 registers.
//   FLAGS: 1 = keep per-element checks everywhere (A/B of the fast epilogue), 2 = clear registers instead of C = 0,
//          knock-outs (timing only): 64 = the DMA always fetches chunk 0 (L2 hits), 4 = no DMA in the loop, 8 = store once at the end, 16 = no barrier, 32 = no ds_read
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

template <int FLAGS, int WGS, int NBUF = 2>
__global__ __launch_bounds__(256, WGS)
void k_glds(GemmArgs a) {
    constexpr int BM = 128, BN = 128, BKT = 16;
    constexpr int STAGE = (BM + BN) * BKT;             // floats per stage: A 128 x 16, then B 128 x 16
    extern __shared__ __attribute__((aligned(16))) float smem_w[];
    const int bid = blockIdx.x;
    const int xcd = bid & 7, slot = bid >> 3;
    const int n_tile = slot % a.n_tiles;
    const int m_tile = (slot / a.n_tiles) * 8 + xcd;
    if (m_tile >= a.m_tiles) return;
    const int n0 = n_tile * BN, row_base = m_tile * BM;
    const int tid = threadIdx.x;
    long long t0 = 0, w0 = 0;
    if (bid == 0 && tid == 0) { t0 = clock64(); w0 = wall_clock64(); }
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lh = lane >> 5;
    // staging: this wave fills pieces 2w, 2w+1 of the A tile and of the B tile; lane -> (row = lane >> 2, slot = lane & 3)
    const int srow = lane >> 2, sq = (lane & 3) ^ ((lane >> 4) & 3);
    unsigned a_off[2], b_off[2];                       // byte offsets from the tile's first row
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int r = (2 * wave + p) * 16 + srow;
        a_off[p] = (unsigned)((row_base + r < a.M ? r : 0) * a.K + 4 * sq) * 4u;       // rows past M: any valid row (never stored)
        b_off[p] = (unsigned)(r * a.K + 4 * sq) * 4u;
    }
    const float* xt = a.x + (int64_t)row_base * a.K;   // uniform
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
    int lz = 0, lk = 0;
#define G_LOAD(BUF)                                                                                             \
    do {                                                                                                        \
        const char* xs_ = reinterpret_cast<const char*>(xt + (int64_t)lz * a.xb + lk * BKT);                     \
        const char* ws_ = reinterpret_cast<const char*>(wt + (int64_t)lz * a.wb + lk * BKT);                     \
        float* st_ = smem_w + (BUF) * STAGE + (2 * wave) * 256;                                                 \
        _Pragma("unroll") for (int p_ = 0; p_ < 2; ++p_) {                                                      \
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(xs_ + a_off[p_]), (lds_ptr_t)(st_ + p_ * 256), 16, 0, 0);             \
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(ws_ + b_off[p_]), (lds_ptr_t)(st_ + BM * BKT + p_ * 256), 16, 0, 0);  \
        }                                                                                                       \
        if (!(FLAGS & 64)) { if (lk + 1 < kchunks) ++lk; else if (lz + 1 < a.nb) { lk = 0; ++lz; } }            \
    } while (0)
#define G_MFMA(BUF, FIRST)                                                                                      \
    do {                                                                                                        \
        const float* sb_ = smem_w + (BUF) * STAGE;                                                              \
        f32x4 af[2][2], bf[2][2];                                                                               \
        if (FLAGS & 32) { _Pragma("unroll") for (int g = 0; g < 2; ++g) _Pragma("unroll") for (int i = 0; i < 2; ++i) { af[g][i] = kaf; bf[g][i] = kbf; } } else { \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                         \
            af[0][i] = *reinterpret_cast<const f32x4*>(sb_ + a_fr0 + i * 32 * BKT);                             \
            bf[0][i] = *reinterpret_cast<const f32x4*>(sb_ + b_fr0 + i * 32 * BKT);                             \
        }                                                                                                       \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                         \
            af[1][i] = *reinterpret_cast<const f32x4*>(sb_ + a_fr1 + i * 32 * BKT);                             \
            bf[1][i] = *reinterpret_cast<const f32x4*>(sb_ + b_fr1 + i * 32 * BKT);                             \
        } }                                                                                                     \
        _Pragma("unroll") for (int g = 0; g < 2; ++g)                                                           \
        _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                           \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                         \
            if (FIRST && g == 0) {                                                                              \
                const f32x16 z_ = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}; \
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[g][i].x, bf[g][j].x, z_, 0, 0, 0);          \
            } else                                                                                              \
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[g][i].x, bf[g][j].x, acc[i][j], 0, 0, 0);   \
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[g][i].y, bf[g][j].y, acc[i][j], 0, 0, 0);       \
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[g][i].z, bf[g][j].z, acc[i][j], 0, 0, 0);       \
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[g][i].w, bf[g][j].w, acc[i][j], 0, 0, 0);       \
        }                                                                                                       \
    } while (0)
    const bool interior = !(FLAGS & 1) && row_base + BM <= a.M && n0 + BN <= a.N;
    f32x4 kaf = {1.f + tid, 2.f, 3.f, 4.f}, kbf = {0.5f, 0.25f * tid, 2.f, 1.f};
    const unsigned y_off = (unsigned)((wm * 64 + 4 * lh) * a.ldy + wn * 64 + li) * 4u;
    G_LOAD(0);
    if (NBUF == 3) G_LOAD(1);
    if (NBUF == 3) { __builtin_amdgcn_s_waitcnt(0x0F74); __builtin_amdgcn_s_barrier(); }   // vmcnt(4): chunk 0 has landed
    else __syncthreads();
    int cz = 0, ck = 0;
    int buf = 0, lbuf = NBUF == 3 ? 2 : 1;
    for (int kc = 0; kc < total; ++kc) {
        if (!(FLAGS & 4)) G_LOAD(lbuf);                // past the end: re-reads the last chunk (harmless)
        __builtin_amdgcn_sched_barrier(0);
        if (ck == 0 && !(FLAGS & 2)) G_MFMA(buf, 1); else G_MFMA(buf, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (NBUF == 3) {
            // the four DMAs just issued may stay in flight across the barrier; everything older has landed
            __builtin_amdgcn_s_waitcnt(0x0F74);
            __builtin_amdgcn_s_barrier();
            buf = buf == 2 ? 0 : buf + 1; lbuf = lbuf == 2 ? 0 : lbuf + 1;
        } else {
            if (!(FLAGS & 16)) __syncthreads();
            buf ^= 1; lbuf ^= 1;
        }
        ++ck;
        if (ck == kchunks && (FLAGS & 8) && kc + 1 < total) { ck = 0; ++cz; }
        if (ck == kchunks) {
            char* yt = reinterpret_cast<char*>(a.y + (int64_t)cz * a.yb + (int64_t)row_base * a.ldy + n0);   // uniform
            if (interior) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int ro = i * 32 + (r & 3) + 8 * (r >> 2);
                        float* yr = reinterpret_cast<float*>(yt + (size_t)ro * a.ldy * 4 + y_off);
                        yr[0] = acc[i][0][r];
                        yr[32] = acc[i][1][r];
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
                            if (c_ok && row_base + wm * 64 + 4 * lh + ro < a.M)
                                reinterpret_cast<float*>(yt + (size_t)ro * a.ldy * 4 + y_off)[j * 32] = acc[i][j][r];
                        }
                }
            }
            if (FLAGS & 2) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
            }
            ck = 0; ++cz;
        }
    }
#undef G_LOAD
#undef G_MFMA
    if (bid == 0 && tid == 0) { a.dbg[0] = clock64() - t0; a.dbg[1] = wall_clock64() - w0; }
}

// ---------------------------------------------------------------------------------------------------------------------
// variant family Wd: ONE workgroup per 128-row tile covers ALL N columns (N = 128 NW: 4 NW waves, wave tile 64 x 64 as before),
// so the A tile is fetched into LDS once per M tile instead of once per N tile: 8 + 8 NW DMA pieces per chunk for 4 NW waves
// (3 / 2.67 per wave where the 128 x 128 workgroups issue 4) - the DMA issue is what the knock-outs price at 11-13 %.
template <int NW>
__global__ __launch_bounds__(256 * NW, 1)
void k_wide(GemmArgs a) {
    constexpr int BM = 128, BN = 128 * NW, BKT = 16, NWAVE = 4 * NW, NPIECE = 8 + 8 * NW, PPW = (NPIECE + NWAVE - 1) / NWAVE;
    constexpr int STAGE = (BM + BN) * BKT;
    extern __shared__ __attribute__((aligned(16))) float smem_w[];
    const int m_tile = blockIdx.x;
    if (m_tile >= a.m_tiles) return;
    const int row_base = m_tile * BM;
    const int tid = threadIdx.x;
    long long t0 = 0, w0 = 0;
    if (m_tile == 0 && tid == 0) { t0 = clock64(); w0 = wall_clock64(); }
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int wm = wave & 1, wn = wave >> 1;
    const int li = lane & 31, lh = lane >> 5;
    const int srow = lane >> 2, sq = (lane & 3) ^ ((lane >> 4) & 3);
    // piece p = wave + NWAVE i: p < 8 -> rows 16 p.. of the A tile, else rows 16 (p - 8).. of the B tile (LDS: A then B)
    unsigned p_off[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int p = wave + NWAVE * i;
        const int r = (p < 8 ? p : p - 8) * 16 + srow;
        p_off[i] = p < 8 ? (unsigned)((row_base + r < a.M ? r : 0) * a.K + 4 * sq) * 4u : (unsigned)(r * a.K + 4 * sq) * 4u;
    }
    const float* xt = a.x + (int64_t)row_base * a.K;
    const float* wt = a.w;
    const int kchunks = a.K / BKT, total = kchunks * a.nb;
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
    int lz = 0, lk = 0;
#define WD_LOAD(BUF)                                                                                            \
    do {                                                                                                        \
        const char* xs_ = reinterpret_cast<const char*>(xt + (int64_t)lz * a.xb + lk * BKT);                     \
        const char* ws_ = reinterpret_cast<const char*>(wt + (int64_t)lz * a.wb + lk * BKT);                     \
        float* st_ = smem_w + (BUF) * STAGE;                                                                    \
        _Pragma("unroll") for (int i_ = 0; i_ < PPW; ++i_) {                                                    \
            const int p_ = wave + NWAVE * i_;                                                                   \
            if (NPIECE % NWAVE == 0 || p_ < NPIECE)                                                             \
                __builtin_amdgcn_global_load_lds((glb_ptr_t)((p_ < 8 ? xs_ : ws_) + p_off[i_]), (lds_ptr_t)(st_ + p_ * 256), 16, 0, 0); \
        }                                                                                                       \
        if (lk + 1 < kchunks) ++lk; else if (lz + 1 < a.nb) { lk = 0; ++lz; }                                   \
    } while (0)
    const unsigned y_off = (unsigned)((wm * 64 + 4 * lh) * a.ldy + wn * 64 + li) * 4u;
    const bool interior = row_base + BM <= a.M && BN <= a.N;
    WD_LOAD(0);
    __syncthreads();
    int cz = 0, ck = 0;
    for (int kc = 0; kc < total; ++kc) {
        const int buf = kc & 1;
        WD_LOAD(buf ^ 1);
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
        if (++ck == kchunks) {
            char* yt = reinterpret_cast<char*>(a.y + (int64_t)cz * a.yb + (int64_t)row_base * a.ldy);
            if (interior) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int ro = i * 32 + (r & 3) + 8 * (r >> 2);
                        float* yr = reinterpret_cast<float*>(yt + (size_t)ro * a.ldy * 4 + y_off);
                        yr[0] = acc[i][0][r];
                        yr[32] = acc[i][1][r];
                    }
            } else {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const bool c_ok = wn * 64 + j * 32 + li < a.N;
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int ro = i * 32 + (r & 3) + 8 * (r >> 2);
                            if (c_ok && row_base + wm * 64 + 4 * lh + ro < a.M)
                                reinterpret_cast<float*>(yt + (size_t)ro * a.ldy * 4 + y_off)[j * 32] = acc[i][j][r];
                        }
                }
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
            ck = 0; ++cz;
        }
    }
#undef WD_LOAD
    if (m_tile == 0 && tid == 0) { a.dbg[0] = clock64() - t0; a.dbg[1] = wall_clock64() - w0; }
}

// ---------------------------------------------------------------------------------------------------------------------
// variant family WS: weight-stationary.  A workgroup owns ONE (problem z, N tile) and keeps that B tile [128 cols][K] in
// REGISTERS (K fragment registers per lane: 384 at K = 384, one wave per SIMD), then walks a range of M tiles streaming only
// A: half the DMA pieces and half the fragment reads per MFMA.  The issue model prices it at 2048 / (2048 + 64 + 120 + 60) =
// 89 %; with one wave per SIMD every latency has to be hidden inside the wave: three LDS buffers for A, DMA two chunks ahead,
// the fragments of chunk s+1 read while the MFMAs of chunk s run, one raw barrier per chunk, counted vmcnt.
template <int KCH>
__global__ __launch_bounds__(256, 1)
void k_ws(GemmArgs a) {
    const int mgroups = a.mgroups;
    constexpr int BM = 128, BKT = 16, ABUF = BM * BKT;     // floats per A stage (8 KB)
    extern __shared__ __attribute__((aligned(16))) float smem_w[];
    const int bid = blockIdx.x;
    const int mg = bid % mgroups, pair = bid / mgroups;
    const int z = pair / a.n_tiles, n_tile = pair % a.n_tiles;
    if (z >= a.nb) return;
    const int tiles_per = (a.m_tiles + mgroups - 1) / mgroups;
    const int t0 = mg * tiles_per, t1 = min(a.m_tiles, t0 + tiles_per);
    if (t0 >= t1) return;
    const int n0 = n_tile * 128;
    const int tid = threadIdx.x;
    long long tc0 = 0, tw0 = 0;
    if (bid == 0 && tid == 0) { tc0 = clock64(); tw0 = wall_clock64(); }
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lh = lane >> 5;
    const int srow = lane >> 2, sq = (lane & 3) ^ ((lane >> 4) & 3);
    const int fsw = (li >> 2) & 3;
    const int fr0 = li * BKT + 4 * (lh ^ fsw), fr1 = li * BKT + 4 * ((lh + 2) ^ fsw);
    // ---- B tile -> registers, chunk by chunk through LDS (buffer 0/1 alternating, plain barriers: once per workgroup)
    f32x4 breg[KCH][2][2];
    {
        const float* wt = a.w + (int64_t)z * a.wb + (int64_t)n0 * a.K;
        unsigned b_off[2];
#pragma unroll
        for (int p = 0; p < 2; ++p) b_off[p] = (unsigned)(((2 * wave + p) * 16 + srow) * a.K + 4 * sq) * 4u;
#pragma unroll
        for (int c = 0; c < KCH; ++c) {
            float* st = smem_w + (c & 1) * ABUF;
#pragma unroll
            for (int p = 0; p < 2; ++p)
                __builtin_amdgcn_global_load_lds((glb_ptr_t)(reinterpret_cast<const char*>(wt + c * BKT) + b_off[p]),
                                                 (lds_ptr_t)(st + (2 * wave + p) * 256), 16, 0, 0);
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                breg[c][0][j] = *reinterpret_cast<const f32x4*>(st + (wn * 64 + j * 32) * BKT + fr0);
                breg[c][1][j] = *reinterpret_cast<const f32x4*>(st + (wn * 64 + j * 32) * BKT + fr1);
            }
        }
        __syncthreads();
    }
    // ---- stream A over the M tiles [t0, t1): flat step s = (tile, chunk)
    const int nsteps = (t1 - t0) * KCH;
    unsigned a_off[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) a_off[p] = (unsigned)(((2 * wave + p) * 16 + srow) * a.K + 4 * sq) * 4u;
    const float* xz = a.x + (int64_t)z * a.xb;
    int l_tile = t0, l_c = 0, l_buf = 0;                   // next DMA: (tile, chunk) into ring slot l_buf
#define WS_DMA()                                                                                                \
    do {                                                                                                        \
        const int lt_ = l_tile < t1 ? l_tile : t1 - 1;                       /* past the end: harmless re-read */   \
        const char* xs_ = reinterpret_cast<const char*>(xz + (int64_t)lt_ * BM * a.K + l_c * BKT);              \
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
    f32x4 fa[2][2][2];                                     // [set][g][i]
    WS_DMA();                                              // step 0
    WS_DMA();                                              // step 1
    __builtin_amdgcn_s_waitcnt(0x0F72);                    // vmcnt(2): step 0 has landed
    __builtin_amdgcn_s_barrier();
    int r_buf = 0;                                         // ring slot of the step whose fragments are read NEXT
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
            // the DMA of the NEXT step (issued one step ago) must have landed before its fragments are read below;
            // behind a tile's 64 stores that is "all but the newest 63"
            if (c == 0 && t != t0) __builtin_amdgcn_s_waitcnt(0xCF7F); else __builtin_amdgcn_s_waitcnt(0x0F70);
            __builtin_amdgcn_s_barrier();
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
        char* yt = reinterpret_cast<char*>(a.y + (int64_t)z * a.yb + (int64_t)t * BM * a.ldy + n0);
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
    if (bid == 0 && tid == 0) { a.dbg[0] = clock64() - tc0; a.dbg[1] = wall_clock64() - tw0; }
}

// ---------------------------------------------------------------------------------------------------------------------
struct Variant { const char* name; void (*fn)(GemmArgs); bool ko; int nw; };   // nw > 0: one workgroup of 256 nw threads per M tile, N = 128 nw only

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 32768;
    const int reps = 5;
    const Variant vars[] = {
        {"B0 register-staged (old) ", k_base<0>, false, 0},
        {"G2 LDS-DMA (product)     ", k_glds<2, 3>, false, 0},
        {"Wd all N in one workgroup", k_wide<2>, false, 2},
        {"Wd all N in one workgroup", k_wide<3>, false, 3},
        {"G2 KO no DMA in loop     ", k_glds<2 | 4, 3>, true, 0},
        {"WS weight-stationary     ", k_ws<6>, false, -6},
        {"WS weight-stationary     ", k_ws<16>, false, -16},
        {"WS weight-stationary     ", k_ws<24>, false, -24},
    };
    const int nv = sizeof(vars) / sizeof(vars[0]);
    constexpr size_t lds = (size_t)3 * (128 + 128) * 16 * sizeof(float);   // 48 KB: enough for the 256-thread variants
    constexpr size_t lds_wide = 82 * 1024;                                 // > 80 KB: one wide workgroup per CU
    for (int v = 0; v < nv; ++v)
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(vars[v].fn), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)(vars[v].nw > 0 ? lds_wide : lds)));
    const int shapes[][2] = {{96, 256}, {256, 256}, {256, 384}, {384, 384}, {384, 256}};
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (auto& sh : shapes) {
        const int K = sh[0], N = sh[1], nb = 25;
        const size_t nx = (size_t)nb * M * K, nw = (size_t)nb * N * K, ny = (size_t)nb * M * N;
        std::vector<float> hx(nx), hw(nw);
        uint32_t s = 12345u + K * 7 + N;
        auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; };
        for (auto& v : hx) v = rnd();
        for (auto& v : hw) v = rnd();
        float *dx, *dw, *dy, *dref; long long* ddbg; CK(hipMalloc(&ddbg, 8 * 4100)); CK(hipMemset(ddbg, 0, 8 * 4100));
        CK(hipMalloc(&dx, nx * 4)); CK(hipMalloc(&dw, nw * 4)); CK(hipMalloc(&dy, ny * 4)); CK(hipMalloc(&dref, ny * 4));
        CK(hipMemcpy(dx, hx.data(), nx * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dw, hw.data(), nw * 4, hipMemcpyHostToDevice));
        GemmArgs g{dx, dw, dref, M, K, N, N, nb, (M + 127) / 128, N / 128, (int64_t)M * K, (int64_t)N * K, (int64_t)M * N, ddbg, 10};
        const unsigned grid = (unsigned)(8 * ((g.m_tiles + 7) / 8) * g.n_tiles);
        const double flop = 2.0 * nb * M * (double)K * N;
        std::vector<float> href(ny), hy(ny);
        g.mgroups = (N == 384 ? 3 : 2) * 256 / (g.nb * g.n_tiles);
        printf("K %d N %d (grid %u; WS: %d M groups per pair, %d workgroups)\n", K, N, grid, g.mgroups, g.nb * g.n_tiles * g.mgroups);
#define LAUNCH(V) do { if (vars[V].nw < 0) hipLaunchKernelGGL(vars[V].fn, dim3(g.nb * g.n_tiles * g.mgroups), dim3(256), lds, 0, g); \
                       else if (vars[V].nw) hipLaunchKernelGGL(vars[V].fn, dim3(8 * ((g.m_tiles + 7) / 8)), dim3(256 * vars[V].nw), lds_wide, 0, g); \
                       else hipLaunchKernelGGL(vars[V].fn, dim3(grid), dim3(256), lds, 0, g); } while (0)
        // correctness pass
        for (int v = 0; v < nv; ++v) {
            if (vars[v].nw > 0 && vars[v].nw != N / 128) continue;
            if (vars[v].nw < 0 && -vars[v].nw != K / 16) continue;
            g.y = v == 0 ? dref : dy;
            CK(hipMemset(g.y, 0xff, ny * 4));
            LAUNCH(v);
            CK(hipGetLastError());
            CK(hipDeviceSynchronize());
            if (v == 0) CK(hipMemcpy(href.data(), dref, ny * 4, hipMemcpyDeviceToHost));
            else if (!vars[v].ko) {
                CK(hipMemcpy(hy.data(), dy, ny * 4, hipMemcpyDeviceToHost));
                size_t bad = 0; double maxd = 0.0;
                for (size_t i = 0; i < ny; ++i)
                    if (memcmp(&hy[i], &href[i], 4)) { ++bad; const double d = fabs((double)hy[i] - href[i]); if (!(d <= maxd)) maxd = d; }
                if (bad) printf("  %s MISMATCH %zu maxdiff %.3g\n", vars[v].name, bad, maxd);
            }
        }
        {
            std::vector<long long> hd(4100);
            CK(hipMemcpy(hd.data(), ddbg, 8 * 4100, hipMemcpyDeviceToHost));
            int hist[16] = {0};
            for (unsigned b = 0; b < grid && b < 4096; ++b) hist[hd[2 + b] & 15]++;
            printf("  wave slot histogram:");
            for (int i = 0; i < 16; ++i) printf(" %d", hist[i]);
            printf("\n");
        }
        // timing: warm clocks first, then rounds that walk the variants in turn
        g.y = dy;
        for (int r = 0; r < 40; ++r) LAUNCH(0);
        CK(hipDeviceSynchronize());
        const int rounds = 4;
        std::vector<float> best(nv, 1e30f), sum(nv, 0.f); std::vector<double> ghz(nv, 0.0);
        for (int rd = 0; rd < rounds; ++rd)
            for (int v = 0; v < nv; ++v) {
                if (vars[v].nw > 0 && vars[v].nw != N / 128) continue;
                if (vars[v].nw < 0 && -vars[v].nw != K / 16) continue;
                CK(hipEventRecord(e0, 0));
                for (int r = 0; r < reps; ++r) LAUNCH(v);
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
                if (ms < best[v]) best[v] = ms;
                sum[v] += ms;
                long long hd[2]; CK(hipMemcpy(hd, ddbg, 16, hipMemcpyDeviceToHost));
                if (hd[1] > 0) ghz[v] = (double)hd[0] / hd[1] * 0.1;
            }
        for (int v = 0; v < nv; ++v)
            if (vars[v].nw == 0 || vars[v].nw == N / 128 || -vars[v].nw == K / 16)
            printf("  %s best %8.4f ms %6.1f TF   mean %8.4f ms %6.1f TF  clock %.3f GHz -> %.1f%% of the MFMA rate at that clock %s\n",
                   vars[v].name, best[v], flop / best[v] / 1e9, sum[v] / rounds, flop / (sum[v] / rounds) / 1e9, ghz[v],
                   ghz[v] > 0 ? 100.0 * (flop / (sum[v] / rounds) / 1e9) / (65.536 * ghz[v]) : 0.0, vars[v].ko ? "(KO)" : "");
        fflush(stdout);
        CK(hipFree(dx)); CK(hipFree(dw)); CK(hipFree(dy)); CK(hipFree(dref));
    }
    return 0;
}
