// GlobalStage encoder pieces as HIP kernels (inference): fp32 flash-style multi-head attention on the f32 MFMA,
// residual + LayerNorm, positional-encoding add.  The linears (in-proj, out-proj, FFN, generator) run on the
// implicit-GEMM kernel of be_conv.hip as 1x1 "convs".
//
// Replaces nn.TransformerEncoderLayer (post-norm, ReLU FFN, eval mode) of models/global_stage.py:28-32 for
// L = 4096 tokens, d_model 128, 8 heads of 16.  The reference's eager attention materialises a [8,4096,4096]
// score tensor per layer (537 MB); here scores never leave registers.
//
// k_attention: one wavefront = 32 queries of one (batch, head), streaming all keys in blocks of 32, on the 16x16x4 f32
//   MFMA (see the comment above the kernel): scores and probabilities live in accumulator registers and are fed straight
//   back as the B operand of the next product; K / Q rows and the transposed V are read as 16-byte loads.
#include "be_common.h"
#include "be_device_math.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int DH = 16;       // head dim
constexpr int KB = 32;       // keys per block

// Counter-based dropout: element `idx` of dropout site `site` is kept iff mix32(idx ^ site_key) >= p * 2^32.  The elementwise
// sites evaluate the same function again in their backward, so they store no mask; the attention forward leaves one keep BIT per
// probability (L^2 / 8 bytes per head) because hashing costs the backward more than reading it (see k_attention).
__host__ __device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
__host__ __device__ __forceinline__ uint32_t site_key(uint32_t seed, uint32_t site) {
    return mix32(seed + 0x9e3779b9U * (site + 1U));
}
// Attention dropout: ONE hash decides the two neighbouring keys 2j, 2j+1 of a query, 16 bits each (the hash was a third of
// the training forward: two 32-bit multiplies per probability).  Element idx = query * L + key of head-batch `hkey` is
// dropped iff its half of mix32((idx >> 1) ^ hkey) is below p * 2^16; the forward holds four consecutive keys per lane (two
// hashes for four elements); k_attn_keep_bits and the mask kernel of the tests evaluate the same function.
__host__ __device__ __forceinline__ uint32_t pair_hash(uint32_t idx, uint32_t hkey) { return mix32((idx >> 1) ^ hkey); }
__host__ __device__ __forceinline__ bool pair_dropped(uint32_t idx, uint32_t hkey, uint32_t t16) {
    return ((pair_hash(idx, hkey) >> (16U * (idx & 1U))) & 0xffffU) < t16;
}
// raw v_exp_f32: exp2f() wraps it in denormal-range scaling (5 instructions); results below 2^-126 flush to zero here,
// which a softmax weight may
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
inline uint32_t drop_threshold(float p) { return p <= 0.f ? 0U : (uint32_t)((double)p * 4294967296.0); }

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

// training: every operand the forward and the two backward kernels read, row-major [BH][L][16] and transposed
// [BH][16][L] (Q pre-scaled in both).  One thread = one (batch, head, token), token fastest: 64-byte row reads and
// writes, and for each d the transposed stores of consecutive threads are consecutive floats.
__global__ __launch_bounds__(256)
void k_qkv_split_train(const float* __restrict__ qkv, float* __restrict__ Q, float* __restrict__ K,
                       float* __restrict__ V, float* __restrict__ Qt, float* __restrict__ Kt,
                       float* __restrict__ Vt, int B, int L, int H, float qscale) {
    // one workgroup = 64 consecutive tokens of one (batch, head).  Reads: a quarter-wave fetches a token's 64-byte q / k / v row
    // of this head (the first version had every lane walk its own row in four pieces: 64 lines touched per instruction for 16
    // useful bytes each); the row-major copies go straight out (4 KB contiguous per matrix), the transposed ones through LDS so
    // that 64 consecutive tokens of one d are one 256-byte store.
    __shared__ float t[3][64][17];
    const int D = H * DH;
    const int tiles = L / 64;
    const int64_t wg = blockIdx.x;
    const int lt = (int)(wg % tiles);
    const int64_t bh = wg / tiles;
    const int64_t b = bh / H, h = bh % H;
    const int l0 = lt * 64;
    const int tok = threadIdx.x >> 2, q4 = threadIdx.x & 3;
    const float* row = qkv + (b * L + l0 + tok) * 3 * D + h * DH + 4 * q4;
    f32x4 v[3];
#pragma unroll
    for (int m = 0; m < 3; ++m) v[m] = *reinterpret_cast<const f32x4*>(row + m * D);
    v[0] *= qscale;
    const int64_t rm = (bh * L + l0 + tok) * DH + 4 * q4;
    *reinterpret_cast<f32x4*>(Q + rm) = v[0];
    *reinterpret_cast<f32x4*>(K + rm) = v[1];
    *reinterpret_cast<f32x4*>(V + rm) = v[2];
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
        for (int e = 0; e < 4; ++e) t[m][tok][4 * q4 + e] = v[m][e];
    __syncthreads();
    float* const outs[3] = {Qt, Kt, Vt};
    const int tk = threadIdx.x & 63, d0 = threadIdx.x >> 6;
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
        for (int dd = 0; dd < 4; ++dd) {
            const int d = d0 + 4 * dd;
            outs[m][(bh * DH + d) * L + l0 + tk] = t[m][tk][d];
        }
}

// ---- attention on v_mfma_f32_16x16x4_f32 -----------------------------------------------------------------------
// With d_head = 16 every product here has a 16-wide side, so the 16x16x4 shape wastes nothing (the 32x32x2 shape
// multiplied zeros in half of its rows for O^T = V^T P^T and the three backward products: 24 -> 16 MFMA-cycles per
// unit in the forward, 80 -> 56 in the backward).  Operand maps of the instruction: lane l (c = l & 15, g = l >> 4)
// supplies A[row c][k = g] and B[k = g][col c]; it receives D[row 4g + r][col c] in register r = 0..3.
// One wave = 32 queries (two column tiles qc) against key blocks of 32 (two row tiles kt):
//   S^T tile [16 keys x 16 queries] = K Q^T, contraction over d in the order d = 4g + t (t = MFMA step), so both
//   operands are ONE 16-byte load of a row-major [L][16] row;
//   a lane then holds, for its query 16 qc + c, the scores of keys 16 kt + 4g + r: the softmax statistics need the
//   four lanes c, c+16, c+32, c+48 (two xor-shuffles); the probabilities stay in those registers and are the B operand
//   of O^T[d][query] += V^T[d][key] P[key][query] with the k-step (kt, r) <-> key 16 kt + 4g + r, whose A operand is one
//   16-byte load of the transposed V ([16][L]).  The accumulator O^T[d = 4g + r][query] is stored as 16-byte rows.
typedef float f32x4v __attribute__((ext_vector_type(4)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ float quad_max(float v) {          // over the lanes c, c+16, c+32, c+48
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float quad_sum(float v) {
    v += __shfl_xor(v, 16, 64);
    return v + __shfl_xor(v, 32, 64);
}

// TRAIN: dropout on the probabilities (nn.MultiheadAttention's dropout, models/global_stage.py:28); the log2-sum-exp of every query
// row and the keep bits of every probability are saved for the backward
// RAGGED: only the first Lv (< L) tokens are real, the rest pads the sequence to a multiple of 128: keys >= Lv get the score
// -inf (probability exactly 0), fully padded key blocks are never visited.  Padded QUERY rows are computed like any other
// (their output is discarded by the caller; in training their upstream gradient is exactly 0).
// a, b: per-lane partial maxima of two independent quantities; on return every lane holds, for each of them, the maximum over its
// four lanes c, c+16, c+32, c+48.  v_permlane32_swap exchanges rows 2,3 of the first operand with rows 0,1 of the second,
// v_permlane16_swap the odd rows of the first with the even rows of the second: three swaps, two max, two moves - no LDS trip.
// (Plain fmaxf on purpose: a swap that reads a register written by INLINE ASM gets none of the wait states the hazard
// recognizer puts between a VALU write and v_permlane*_swap - seen as run-to-run differences in the softmax reference.)
__device__ __forceinline__ void quad_max2(float& a, float& b) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);   // [a0 a1 b0 b1], [a2 a3 b2 b3]
    const float w = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));                                  // [a02 a13 b02 b13]
    auto q = __builtin_amdgcn_permlane16_swap(__float_as_uint(w), __float_as_uint(w), false, false);   // [w0 w0 w2 w2], [w1 w1 w3 w3]
    const float u = fmaxf(__uint_as_float(q[0]), __uint_as_float(q[1]));                                  // [A A B B]
    auto t = __builtin_amdgcn_permlane32_swap(__float_as_uint(u), __float_as_uint(u), false, false);   // [A A A A], [B B B B]
    a = __uint_as_float(t[0]);
    b = __uint_as_float(t[1]);
}

__device__ __forceinline__ float max8(const f32x4v a, const f32x4v b) {
    return fmaxf(fmaxf(fmaxf(a[0], a[1]), fmaxf(a[2], a[3])), fmaxf(fmaxf(b[0], b[1]), fmaxf(b[2], b[3])));
}

// The softmax reference of a query is not its running maximum but a fixed integer m_ref >= max(block 0) - 1, and it enters the
// score product as the INITIAL VALUE of the MFMA accumulator (C = -m_ref): the probabilities are exp2 of the accumulator, with
// no maximum, no cross-lane exchange, no subtraction and no rescaling of O in the loop.  A guard on the block's probability sum
// (> 2^60: the scores have risen 60 binades above the reference) sends the wave through a slow path that raises the
// reference by an integer (exact power-of-two rescale).  fp32 MFMA and VALU instructions share the SIMD's issue slot on this
// chip, so every VALU instruction removed from the loop is time saved.
// two dropout decisions from one pair hash: p0 / p1 are kept or zeroed, their keep bits are shifted into acc
__device__ __forceinline__ void keep2(uint32_t h, uint32_t t16, float& p0, float& p1, uint32_t& acc) {
    asm("v_cmp_ge_u32_sdwa vcc, %3, %4 src0_sel:WORD_0 src1_sel:DWORD\n\t"
        "v_cndmask_b32_e32 %0, 0, %0, vcc\n\t"
        "v_addc_co_u32_e32 %2, vcc, %2, %2, vcc\n\t"
        "v_cmp_ge_u32_sdwa vcc, %3, %4 src0_sel:WORD_1 src1_sel:DWORD\n\t"
        "v_cndmask_b32_e32 %1, 0, %1, vcc\n\t"
        "v_addc_co_u32_e32 %2, vcc, %2, %2, vcc"
        : "+v"(p0), "+v"(p1), "+v"(acc) : "v"(h), "v"(t16) : "vcc");
}

template <bool TRAIN, bool RAGGED>
__global__ __launch_bounds__(256)
void k_attention(const float* __restrict__ Q, const float* __restrict__ K, const float* __restrict__ Vt,
                   float* __restrict__ out, float* __restrict__ lse, int L, int Lv, int H, uint32_t seed, uint32_t thresh,
                   float inv_keep, float* __restrict__ part, uint16_t* __restrict__ keep) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int c = lane & 15, g = lane >> 4;
    const int bh = blockIdx.y;
    const int q0 = (blockIdx.x * 4 + wave) * 32;
    const float* Qh = Q + (size_t)bh * L * DH;
    const float* Kh = K + (size_t)bh * L * DH;
    const float* Vh = Vt + (size_t)bh * DH * L;
    const uint32_t hkey = site_key(seed, (uint32_t)bh);
    uint32_t t16v = thresh >> 16;                 // p * 2^16: the 16-bit threshold of the pair hash, in a vector register
    asm("" : "+v"(t16v));
    uint16_t* keep_out = TRAIN ? keep + ((size_t)bh * (L / KB) + (q0 >> 5)) * (L / KB) * 64 + lane : nullptr;   // [bh][query tile][key block][lane]

    f32x4v qf[2], o[2], nm[2];                      // nm: -m_ref in all four registers (the accumulator's initial value)
    float mref[2], lsum[2];
#pragma unroll
    for (int qc = 0; qc < 2; ++qc) {
        qf[qc] = *reinterpret_cast<const f32x4v*>(Qh + (size_t)(q0 + 16 * qc + c) * DH + 4 * g);
        o[qc] = f32x4v{0.f, 0.f, 0.f, 0.f};
        lsum[qc] = 0.f;
    }
    const int nblk = RAGGED ? (Lv + KB - 1) / KB : L / KB;
    const int kb0 = nblk * blockIdx.z / gridDim.z, nkb = nblk * (blockIdx.z + 1) / gridDim.z;

    auto loadKV = [&](int blk, f32x4v k[2], f32x4v v[2]) {
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            k[kt] = *reinterpret_cast<const f32x4v*>(Kh + (size_t)(blk * KB + 16 * kt + c) * DH + 4 * g);
            v[kt] = *reinterpret_cast<const f32x4v*>(Vh + (size_t)c * L + blk * KB + 16 * kt + 4 * g);
        }
    };
    auto raw_scores = [&](const f32x4v k[2], f32x4v s[2][2], int kblk) {      // C = 0, padded keys -> -inf
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int qc = 0; qc < 2; ++qc) s[kt][qc] = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int qc = 0; qc < 2; ++qc) s[kt][qc] = MFMA16(k[kt][t], qf[qc][t], s[kt][qc]);
        if (RAGGED && (kblk + 1) * KB > Lv) {
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (kblk * KB + 16 * kt + 4 * g + r >= Lv) { s[kt][0][r] = -INFINITY; s[kt][1][r] = -INFINITY; }
        }
    };
    auto block_max = [&](const f32x4v s[2][2], float mx[2]) {                 // per query, over the 32 keys of the block
        mx[0] = max8(s[0][0], s[1][0]);
        mx[1] = max8(s[0][1], s[1][1]);
        quad_max2(mx[0], mx[1]);
    };
    // dropout: keeps / zeroes the 16 probabilities of this lane and leaves their keep bits (decision d = 8 qc + 4 kt + r in bit
    // 15 - d) in the block's mask halfword, which the two backward kernels read instead of hashing again
    auto drop = [&](f32x4v s[2][2], int kblk) {
        uint32_t acc = 0;
#pragma unroll
        for (int qc = 0; qc < 2; ++qc) {
            const uint32_t base = (uint32_t)(q0 + 16 * qc + c) * (uint32_t)L + (uint32_t)(kblk * KB + 4 * g);
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 4; r += 2) {
                    float p0 = s[kt][qc][r], p1 = s[kt][qc][r + 1];
                    keep2(pair_hash(base + 16 * kt + r, hkey), t16v, p0, p1, acc);
                    s[kt][qc][r] = p0; s[kt][qc][r + 1] = p1;
                }
        }
        keep_out[(size_t)kblk * 64] = (uint16_t)acc;
    };
    auto pv = [&](const f32x4v s[2][2], const f32x4v v[2]) {
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int qc = 0; qc < 2; ++qc) o[qc] = MFMA16(v[kt][r], s[kt][qc][r], o[qc]);
    };
    // fast block; `ok` loses a lane's bit when the block's scores sit 60 binades above the reference (the wave then repeats its
    // whole key range in the slow loop below: nothing of the fast pass is kept)
    uint64_t ok = ~0ull;
    auto fast_block = [&](const f32x4v k[2], const f32x4v v[2], int kblk) {
        f32x4v s[2][2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int qc = 0; qc < 2; ++qc) s[kt][qc] = MFMA16(k[kt][0], qf[qc][0], nm[qc]);
#pragma unroll
        for (int t = 1; t < 4; ++t)
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int qc = 0; qc < 2; ++qc) s[kt][qc] = MFMA16(k[kt][t], qf[qc][t], s[kt][qc]);
        if (RAGGED && (kblk + 1) * KB > Lv) {
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (kblk * KB + 16 * kt + 4 * g + r >= Lv) { s[kt][0][r] = -INFINITY; s[kt][1][r] = -INFINITY; }
        }
        float psum[2] = {0.f, 0.f};
#pragma unroll
        for (int qc = 0; qc < 2; ++qc)
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) { s[kt][qc][r] = fast_exp2(s[kt][qc][r]); psum[qc] += s[kt][qc][r]; }
        ok &= __builtin_amdgcn_ballot_w64(psum[0] + psum[1] < 0x1p60f);
        lsum[0] += psum[0]; lsum[1] += psum[1];
        if (TRAIN && thresh) drop(s, kblk);
        pv(s, v);
    };

    f32x4v kA[2], vA[2], kB[2], vB[2];
    loadKV(kb0, kA, vA);
    {   // reference from the first block of this slice (it holds at least one real key)
        f32x4v sr[2][2];
        float mx[2];
        raw_scores(kA, sr, kb0);
        block_max(sr, mx);
#pragma unroll
        for (int qc = 0; qc < 2; ++qc) { mref[qc] = ceilf(mx[qc]); nm[qc] = f32x4v{-mref[qc], -mref[qc], -mref[qc], -mref[qc]}; }
    }
    int kblk = kb0;
    for (; kblk + 1 < nkb; kblk += 2) {                   // branch-free body: two blocks on ping-pong registers
        loadKV(kblk + 1, kB, vB);
        __builtin_amdgcn_sched_barrier(0);               // the loads of the next block are issued BEFORE this block's work
        fast_block(kA, vA, kblk);
        loadKV(kblk + 2 < nkb ? kblk + 2 : kblk, kA, vA);
        __builtin_amdgcn_sched_barrier(0);
        fast_block(kB, vB, kblk + 1);
    }
    if (kblk < nkb) fast_block(kA, vA, kblk);
    if (__builtin_expect(ok != ~0ull, 0)) {
        // the whole key range again with a running maximum (the textbook online softmax): only reached when scores rise by more
        // than 60 binades along the key axis
#pragma unroll
        for (int qc = 0; qc < 2; ++qc) { o[qc] = f32x4v{0.f, 0.f, 0.f, 0.f}; lsum[qc] = 0.f; mref[qc] = -INFINITY; }
        for (kblk = kb0; kblk < nkb; ++kblk) {
            f32x4v k[2], v[2], s[2][2];
            float mx[2];
            loadKV(kblk, k, v);
            raw_scores(k, s, kblk);
            block_max(s, mx);
#pragma unroll
            for (int qc = 0; qc < 2; ++qc) {
                const float mnew = fmaxf(mref[qc], mx[qc]);
                const float alpha = fast_exp2(mref[qc] - mnew);
                mref[qc] = mnew;
                float psum = 0.f;
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) { s[kt][qc][r] = fast_exp2(s[kt][qc][r] - mnew); psum += s[kt][qc][r]; }
                lsum[qc] = lsum[qc] * alpha + psum;
                o[qc] *= alpha;
            }
            if (TRAIN && thresh) drop(s, kblk);
            pv(s, v);
        }
    }
    if (gridDim.z > 1) {
        const size_t rows = (size_t)gridDim.y * L;
        float* po = part + (size_t)blockIdx.z * rows * (DH + 2);
#pragma unroll
        for (int qc = 0; qc < 2; ++qc) {
            const size_t row = (size_t)bh * L + q0 + 16 * qc + c;
            const float ltot = quad_sum(lsum[qc]);
            *reinterpret_cast<f32x4v*>(po + row * DH + 4 * g) = o[qc];
            if (g == 0) { po[rows * DH + row] = mref[qc]; po[rows * (DH + 1) + row] = ltot; }
        }
        return;
    }
    const int Dm = H * DH;
    const int b = bh / H, hd = bh % H;
#pragma unroll
    for (int qc = 0; qc < 2; ++qc) {
        const float ltot = quad_sum(lsum[qc]);
        const float inv = TRAIN ? inv_keep / ltot : 1.0f / ltot;
        if (TRAIN && g == 0) lse[(size_t)bh * L + q0 + 16 * qc + c] = mref[qc] + log2f(ltot);
        *reinterpret_cast<f32x4v*>(out + ((size_t)b * L + q0 + 16 * qc + c) * Dm + hd * DH + 4 * g) = o[qc] * inv;
    }
}


// merges the key slices: out = sum_z o_z 2^(m_z - M) / sum_z l_z 2^(m_z - M), M = max_z m_z; one thread per (row, d quad)
__global__ void k_attn_combine(const float* __restrict__ part, float* __restrict__ out, int nz, int64_t BH, int L, int H) {
    const int64_t rows = BH * L, total = rows * 4;
    const int64_t gs = (int64_t)gridDim.x * blockDim.x;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gs) {
        const int64_t row = idx >> 2;
        const int dq = (int)(idx & 3);
        float M = -INFINITY;
        for (int z = 0; z < nz; ++z) M = fmaxf(M, part[(size_t)z * rows * (DH + 2) + rows * DH + row]);
        f32x4v acc = {0.f, 0.f, 0.f, 0.f};
        float l = 0.f;
        for (int z = 0; z < nz; ++z) {
            const float* pz = part + (size_t)z * rows * (DH + 2);
            const float wgt = fast_exp2(pz[rows * DH + row] - M);
            acc += *reinterpret_cast<const f32x4v*>(pz + row * DH + 4 * dq) * wgt;
            l += pz[rows * (DH + 1) + row] * wgt;
        }
        const int64_t bh = row / L, q = row % L;
        const int64_t b = bh / H, hd = bh % H;
        *reinterpret_cast<f32x4v*>(out + ((size_t)b * L + q) * H * DH + hd * DH + 4 * dq) = acc * (1.0f / l);
    }
}

// ---- attention backward -------------------------------------------------------------------------------------
// With P = softmax(S), Pd = dropout(P), O = Pd V:   dV = Pd^T dO,  dPd = dO V^T,  dS = P o (dropout'(dPd) - Drow),
// Drow_i = sum_d dO_id O_id,  dQ = scale dS K,  dK = scale dS^T Q.   P is recomputed from the saved log2-sum-exp.
// P is recomputed from the saved log2-sum-exp; no [L,L] float tensor is stored; deterministic (fixed summation orders, no atomics).
// dout, out [T, D], lse [BH][L] -> dOh [BH][L][16], dOt [BH][16][L], nD [BH][L] = -keep * sum_d dO * O, nlse = -lse: the two
// per-query constants enter the backward products as the INITIAL VALUES of their accumulators (S' = S - lse, dP' = dP - keep D)
__global__ void k_dout_prep(const float* __restrict__ dout, const float* __restrict__ out, const float* __restrict__ lse,
                              float* __restrict__ dOh, float* __restrict__ dOt, float* __restrict__ nD, float* __restrict__ nlse,
                              float keep, int B, int L, int H) {
    const int64_t total = (int64_t)B * L * H;
    const int64_t gs = (int64_t)gridDim.x * blockDim.x;
    const int D = H * DH;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gs) {
        const int64_t l = idx % L, bh = idx / L;
        const int64_t b = bh / H, h = bh % H;
        const float* g = dout + (b * L + l) * D + h * DH;
        const float* o = out + (b * L + l) * D + h * DH;
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f32x4 gv = *reinterpret_cast<const f32x4*>(g + 4 * i);
            const f32x4 ov = *reinterpret_cast<const f32x4*>(o + 4 * i);
            *reinterpret_cast<f32x4*>(dOh + (bh * L + l) * DH + 4 * i) = gv;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc = fmaf(gv[e], ov[e], acc);
                dOt[(bh * DH + 4 * i + e) * L + l] = gv[e];
            }
        }
        nD[bh * L + l] = -keep * acc;
        nlse[bh * L + l] = -lse[bh * L + l];
    }
}

// ---- experiment: ONE backward kernel (S and dP computed once: 80 MFMAs per 32 x 32 tile instead of 48 + 64) ---------------------
// dK/dV layout (a wave owns 32 keys, streams the queries); the dS tile is transposed through LDS for the extra product
// dQ^T[d][q] += K^T[d][key] dS^T[key][q]; the per-wave dQ partials of a query block are summed over the 8 waves of the workgroup
// in LDS and written as one partial per workgroup (256 keys) - fixed order, no atomics - for a small finishing kernel.
constexpr int FW = 4;

template <bool RAGGED, bool DROP>
__global__ __launch_bounds__(64 * FW)
void k_attn_bwd_fused(const float* __restrict__ Q, const float* __restrict__ K, const float* __restrict__ V,
                      const float* __restrict__ Qt, const float* __restrict__ Kt, const float* __restrict__ dOh,
                      const float* __restrict__ dOt, const float* __restrict__ nlse, const float* __restrict__ nD,
                      float* __restrict__ dqkv, float* __restrict__ dqpart, int L, int Lv, int H,
                      const uint16_t* __restrict__ keep, float inv_keep) {
    __shared__ __attribute__((aligned(16))) float sT[FW][2][2][320];        // [wave][qt][kc][key c][16 queries + 4 pad]: rows of 20 floats
    //                                                                           keep the transposed 4-byte reads of a half-wave on 32 different banks
    __shared__ __attribute__((aligned(16))) float sQ[2][FW][512];           // [buffer][wave][query 32][d 16]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int c = lane & 15, g = lane >> 4;
    const int bh = blockIdx.y;
    const int k0 = (blockIdx.x * FW + wave) * 32;
    const size_t hb = (size_t)bh * L * DH;
    const float* Qh = Q + hb;
    const float* Gh = dOh + hb;
    const float* Qth = Qt + hb;
    const float* Gth = dOt + hb;
    const float* nl_h = nlse + (size_t)bh * L;
    const float* nd_h = nD + (size_t)bh * L;
    const uint16_t* kp = keep + ((size_t)bh * (L / KB) * (L / KB) + (k0 >> 5)) * 64 + 16 * (c >> 2) + 4 * g;
    const uint32_t ksh = 3 - (c & 3);

    f32x4v kf[2], vf[2], tf[2], dv[2], dk[2];
#pragma unroll
    for (int kc = 0; kc < 2; ++kc) {
        kf[kc] = *reinterpret_cast<const f32x4v*>(K + hb + (size_t)(k0 + 16 * kc + c) * DH + 4 * g);
        vf[kc] = *reinterpret_cast<const f32x4v*>(V + hb + (size_t)(k0 + 16 * kc + c) * DH + 4 * g);
        tf[kc] = *reinterpret_cast<const f32x4v*>(Kt + hb + (size_t)c * L + k0 + 16 * kc + 4 * g);      // K^T[d = c][key 16 kc + 4g + r]
        dv[kc] = f32x4v{0.f, 0.f, 0.f, 0.f};
        dk[kc] = f32x4v{0.f, 0.f, 0.f, 0.f};
    }
    struct QB { f32x4v q[2], gq[2], qt[2], gt[2], nl[2], nd[2]; };
    auto load = [&](int blk, QB& x) {
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            x.q[qt] = *reinterpret_cast<const f32x4v*>(Qh + (size_t)(blk * KB + 16 * qt + c) * DH + 4 * g);
            x.gq[qt] = *reinterpret_cast<const f32x4v*>(Gh + (size_t)(blk * KB + 16 * qt + c) * DH + 4 * g);
            x.qt[qt] = *reinterpret_cast<const f32x4v*>(Qth + (size_t)c * L + blk * KB + 16 * qt + 4 * g);
            x.gt[qt] = *reinterpret_cast<const f32x4v*>(Gth + (size_t)c * L + blk * KB + 16 * qt + 4 * g);
            x.nl[qt] = *reinterpret_cast<const f32x4v*>(nl_h + blk * KB + 16 * qt + 4 * g);
            x.nd[qt] = *reinterpret_cast<const f32x4v*>(nd_h + blk * KB + 16 * qt + 4 * g);
        }
    };
    const bool kdead[2] = {RAGGED && k0 + c >= Lv, RAGGED && k0 + 16 + c >= Lv};
    auto loadw = [&](int blk) -> uint2 { return *reinterpret_cast<const uint2*>(kp + (size_t)blk * (L / KB) * 64); };
    float* part = dqpart + ((size_t)blockIdx.x * gridDim.y + bh) * L * DH;           // this workgroup's partial dQ [L][16]
    auto reduce = [&](int qb) {       // sum of the waves' partials of query block qb, element e = query * 16 + d
#pragma unroll
        for (int e = threadIdx.x; e < 512; e += 64 * FW) {
            float t = sQ[qb & 1][0][e];
#pragma unroll
            for (int w2 = 1; w2 < FW; ++w2) t += sQ[qb & 1][w2][e];
            part[(size_t)qb * 512 + e] = t;
        }
    };
    auto block = [&](const QB& x, const uint2 w, int qblk) {
        f32x4v s[2][2], dp[2][2];
        const uint32_t wsh[2] = {DROP ? w.x >> ksh : 0u, DROP ? w.y >> ksh : 0u};
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
#pragma unroll
            for (int kc = 0; kc < 2; ++kc) {
                s[qt][kc] = MFMA16(x.q[qt][0], kf[kc][0], x.nl[qt]);
                dp[qt][kc] = MFMA16(x.gq[qt][0], vf[kc][0], x.nd[qt]);
            }
#pragma unroll
        for (int t = 1; t < 4; ++t)
#pragma unroll
            for (int qt = 0; qt < 2; ++qt)
#pragma unroll
                for (int kc = 0; kc < 2; ++kc) {
                    s[qt][kc] = MFMA16(x.q[qt][t], kf[kc][t], s[qt][kc]);
                    dp[qt][kc] = MFMA16(x.gq[qt][t], vf[kc][t], dp[qt][kc]);
                }
        // the partial dQ of the PREVIOUS query block: every wave has stored its share by now and has 32 MFMAs in flight behind
        // it, so the barrier costs the skew between the waves, not a drained pipe
        if (qblk > 0) { __syncthreads(); reduce(qblk - 1); }
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
#pragma unroll
            for (int kc = 0; kc < 2; ++kc) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = (RAGGED && kdead[kc]) ? 0.f : fast_exp2(s[qt][kc][r]);
                    float a = dp[qt][kc][r], pd = p;
                    if (DROP) {
                        uint32_t km = (uint32_t)__builtin_amdgcn_sbfe((int)wsh[r >> 1], 16 * (r & 1) + 12 - 8 * qt - 4 * kc, 1);
                        asm("" : "+v"(km));       // opaque: the compiler otherwise turns the two bit-selects into v_cmp + 2 v_cndmask
                        const uint32_t ndb = __float_as_uint(x.nd[qt][r]);
                        a = __uint_as_float(ndb ^ (km & (__float_as_uint(a) ^ ndb)));                                  // v_bfi_b32
                        pd = __uint_as_float(km & __float_as_uint(p));
                    }
                    s[qt][kc][r] = p * a;
                    dp[qt][kc][r] = pd;
                }
                // dS tile [queries 4g + r][key c] -> LDS as [key c][query 4g .. 4g + 3]
                *reinterpret_cast<f32x4v*>(&sT[wave][qt][kc][c * 20 + 4 * g]) = s[qt][kc];
            }
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int kc = 0; kc < 2; ++kc) {
                    dv[kc] = MFMA16(x.gt[qt][r], dp[qt][kc][r], dv[kc]);
                    dk[kc] = MFMA16(x.qt[qt][r], s[qt][kc][r], dk[kc]);
                }
        // dQ^T[d][query 16 qt + c] += K^T[d][key] dS^T[key][query]: the B operand of k-step (kc, r) is dS[query c][key 16 kc + 4g + r]
        float* xq = &sQ[qblk & 1][wave][0];
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            f32x4v dq = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kc = 0; kc < 2; ++kc)
#pragma unroll
                for (int r = 0; r < 4; ++r) dq = MFMA16(tf[kc][r], sT[wave][qt][kc][(4 * g + r) * 20 + c], dq);
            *reinterpret_cast<f32x4v*>(xq + (16 * qt + c) * 16 + 4 * g) = dq;       // [query][d 4g .. 4g + 3]
        }
    };
    const int nqb = RAGGED ? (Lv + KB - 1) / KB : L / KB;
    QB A, Bq;
    load(0, A);
    uint2 wA = {0u, 0u}, wB = {0u, 0u}, wA2 = {0u, 0u}, wB2 = {0u, 0u}, wA3 = {0u, 0u}, wB3 = {0u, 0u};
    auto clampq = [&](int b) { return b < nqb ? b : nqb - 1; };
    if (DROP) { wA = loadw(0); wB = loadw(clampq(1)); wA2 = loadw(clampq(2)); wB2 = loadw(clampq(3)); }
    int qblk = 0;
    for (; qblk + 1 < nqb; qblk += 2) {
        load(qblk + 1, Bq);
        if (DROP) { wA3 = loadw(clampq(qblk + 4)); wB3 = loadw(clampq(qblk + 5)); }
        __builtin_amdgcn_sched_barrier(0);
        block(A, wA, qblk);
        load(qblk + 2 < nqb ? qblk + 2 : qblk, A);
        __builtin_amdgcn_sched_barrier(0);
        block(Bq, wB, qblk + 1);
        wA = wA2; wB = wB2; wA2 = wA3; wB2 = wB3;
    }
    if (qblk < nqb) block(A, wA, qblk);
    __syncthreads();
    reduce(nqb - 1);
    const int Dm = H * DH;
    const int b = bh / H, hd = bh % H;
    const float ln2 = 0.69314718055994530942f;
#pragma unroll
    for (int kc = 0; kc < 2; ++kc) {
        float* dst = dqkv + ((size_t)b * L + k0 + 16 * kc + c) * 3 * Dm + hd * DH + 4 * g;
        *reinterpret_cast<f32x4v*>(dst + Dm) = dk[kc] * (ln2 * inv_keep);
        *reinterpret_cast<f32x4v*>(dst + 2 * Dm) = dv[kc] * inv_keep;
    }
}

// dQ = scale * sum over the key groups of the partials; rows of query blocks no workgroup visited (padding) are zero
__global__ void k_attn_dq_finish(const float* __restrict__ part, float* __restrict__ dqkv, int ngroups, int BH, int L, int H,
                                 int rows_done, float scale) {
    const int64_t total = (int64_t)BH * L * 4;
    const int64_t gs = (int64_t)gridDim.x * blockDim.x;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gs) {
        const int dq4 = (int)(idx & 3);
        const int64_t row = idx >> 2;
        const int64_t bh = row / L, l = row % L;
        f32x4v acc = {0.f, 0.f, 0.f, 0.f};
        if (l < rows_done)
            for (int g0 = 0; g0 < ngroups; g0 += 8) {           // eight partials in flight (one load per add left the sum latency-bound: 103 us
                f32x4v p[8];                                    // for 268 MB), added in group order
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (g0 + j < ngroups) p[j] = *reinterpret_cast<const f32x4v*>(part + (((size_t)(g0 + j) * BH + bh) * L + l) * DH + 4 * dq4);
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (g0 + j < ngroups) acc += p[j];
            }
        const int64_t b = bh / H, hd = bh % H;
        *reinterpret_cast<f32x4v*>(dqkv + ((size_t)b * L + l) * 3 * H * DH + hd * DH + 4 * dq4) = acc * scale;
    }
}

// keep mask of the attention dropout, for the parity tests: mask [BH][L][L] (query-major) in {0,1}
__global__ void k_attn_dropout_mask(float* __restrict__ mask, int64_t BH, int L, uint32_t seed, uint32_t thresh) {
    const int64_t total = BH * L * L;
    const int64_t gs = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gs) {
        const int64_t bh = i / ((int64_t)L * L);
        const uint32_t idx = (uint32_t)(i - bh * (int64_t)L * L);
        mask[i] = pair_dropped(idx, site_key(seed, (uint32_t)bh), thresh >> 16) ? 0.f : 1.f;
    }
}

// The keep bits exactly as k_attention<true> stores them ([bh][query tile of 32][key block of 32][lane] halfwords, decision
// d = 8 qc + 4 kt + r of lane (c, g) - query 16 qc + c, key 16 kt + 4 g + r - in bit 15 - d): for a backward call whose forward
// workspace was reused in between, and for the test that pins the forward's stores to the formula.  One wave per halfword line.
__global__ __launch_bounds__(256)
void k_attn_keep_bits(uint16_t* __restrict__ keep, int64_t lines, int L, uint32_t seed, uint32_t thresh) {
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
    const int nb = L / KB;
    const uint32_t t16 = thresh >> 16;
    for (int64_t line = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); line < lines; line += (int64_t)gridDim.x * 4) {
        const int kb = (int)(line % nb), qw = (int)((line / nb) % nb);
        const uint32_t hkey = site_key(seed, (uint32_t)(line / ((int64_t)nb * nb)));
        uint32_t acc = 0;
#pragma unroll
        for (int qc = 0; qc < 2; ++qc)
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const uint32_t idx = (uint32_t)(qw * 32 + 16 * qc + c) * (uint32_t)L + (uint32_t)(kb * KB + 16 * kt + 4 * g + r);
                    acc = (acc << 1) | (pair_dropped(idx, hkey, t16) ? 0u : 1u);
                }
        keep[line * 64 + lane] = (uint16_t)acc;
    }
}

// y = x * keep / (1-p) (* [gate > 0]); gate = the ReLU output makes this the backward of dropout(relu(.))
__global__ void k_dropout(const float* __restrict__ x, const float* __restrict__ gate, float* __restrict__ y, int64_t n,
                          uint32_t key, uint32_t thresh, float inv_keep) {
    const int64_t gs = (int64_t)gridDim.x * blockDim.x;
    if (((n & 3) | ((uintptr_t)x & 15) | ((uintptr_t)y & 15) | ((uintptr_t)gate & 15)) == 0) {
        // four elements per lane as 16-byte accesses (the scalar form below moved 2.2 TB/s on the 16.8 MB activations)
        typedef float v4 __attribute__((ext_vector_type(4)));
        const int64_t n4 = n >> 2;
        for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n4; q += gs) {
            v4 v = reinterpret_cast<const v4*>(x)[q] * inv_keep;
            v4 g = {1.f, 1.f, 1.f, 1.f};
            if (gate) g = reinterpret_cast<const v4*>(gate)[q];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (thresh && mix32((uint32_t)(4 * q + e) ^ key) < thresh) v[e] = 0.f;
                if (!(g[e] > 0.f)) v[e] = 0.f;
            }
            reinterpret_cast<v4*>(y)[q] = v;
        }
        return;
    }
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gs) {
        float v = x[i] * inv_keep;
        if (thresh && mix32((uint32_t)i ^ key) < thresh) v = 0.f;
        if (gate && !(gate[i] > 0.f)) v = 0.f;
        y[i] = v;
    }
}

// training LayerNorm: v = res + dropout(x) is stored (the backward re-derives mean / rstd from it), y = LN(v)
__global__ __launch_bounds__(256)
void k_add_layernorm_train(const float* __restrict__ x, const float* __restrict__ res, const float* __restrict__ gamma,
                           const float* __restrict__ beta, float* __restrict__ v_out, float* __restrict__ y, int64_t rows,
                           float eps, uint32_t key, uint32_t thresh, float inv_keep) {
    constexpr int D = 128;
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float v[2];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int64_t e = row * D + lane + 64 * i;
        float a = x[e] * inv_keep;
        if (thresh && mix32((uint32_t)e ^ key) < thresh) a = 0.f;
        v[i] = a + (res ? res[e] : 0.f);
        v_out[e] = v[i];
        s += v[i];
    }
    const float mean = be::wave_sum(s) / D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) { const float d = v[i] - mean; q += d * d; }
    const float rstd = 1.0f / sqrtf(be::wave_sum(q) / D + eps);
#pragma unroll
    for (int i = 0; i < 2; ++i) y[row * D + lane + 64 * i] = (v[i] - mean) * rstd * gamma[lane + 64 * i] + beta[lane + 64 * i];
}

// LayerNorm backward over D = 128: dv = rstd (g - mean(g) - xhat mean(g xhat)), g = dy gamma; the same dv is the gradient
// of the residual input, and dx = dropout'(dv) of the dropped branch.  Each block reduces its 32 rows' dgamma / dbeta
// contributions into partial[block][2][128]; be_col_sum finishes them.
constexpr int LN_ROWS = 32;
__global__ __launch_bounds__(256)
void k_layernorm_bwd(const float* __restrict__ dy, const float* __restrict__ v, const float* __restrict__ gamma,
                     float* __restrict__ dv, float* __restrict__ dx, float* __restrict__ partial, int64_t rows, float eps,
                     uint32_t key, uint32_t thresh, float inv_keep) {
    constexpr int D = 128;
    __shared__ float red[4][2][D];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float dgam[2] = {0.f, 0.f}, dbet[2] = {0.f, 0.f};
    const float g0 = gamma[lane], g1 = gamma[lane + 64];
    for (int it = 0; it < LN_ROWS / 4; ++it) {
        const int64_t row = (int64_t)blockIdx.x * LN_ROWS + it * 4 + wave;
        if (row >= rows) break;
        const int64_t e0 = row * D + lane, e1 = e0 + 64;
        const float v0 = v[e0], v1 = v[e1];
        const float mean = be::wave_sum(v0 + v1) / D;
        const float c0 = v0 - mean, c1 = v1 - mean;
        const float rstd = 1.0f / sqrtf(be::wave_sum(c0 * c0 + c1 * c1) / D + eps);
        const float x0 = c0 * rstd, x1 = c1 * rstd;
        const float y0 = dy[e0], y1 = dy[e1];
        dgam[0] += y0 * x0; dgam[1] += y1 * x1;
        dbet[0] += y0; dbet[1] += y1;
        const float a0 = y0 * g0, a1 = y1 * g1;
        const float ma = be::wave_sum(a0 + a1) / D;
        const float mb = be::wave_sum(a0 * x0 + a1 * x1) / D;
        const float r0 = rstd * (a0 - ma - x0 * mb), r1 = rstd * (a1 - ma - x1 * mb);
        if (dv) { dv[e0] = r0; dv[e1] = r1; }
        if (dx) {
            float d0 = r0 * inv_keep, d1 = r1 * inv_keep;
            if (thresh && mix32((uint32_t)e0 ^ key) < thresh) d0 = 0.f;
            if (thresh && mix32((uint32_t)e1 ^ key) < thresh) d1 = 0.f;
            dx[e0] = d0; dx[e1] = d1;
        }
    }
    red[wave][0][lane] = dgam[0]; red[wave][0][lane + 64] = dgam[1];
    red[wave][1][lane] = dbet[0]; red[wave][1][lane + 64] = dbet[1];
    __syncthreads();
    const int t = threadIdx.x;                                   // 256 threads = 2 x 128 outputs
    partial[(size_t)blockIdx.x * 2 * D + t] = red[0][t >> 7][t & 127] + red[1][t >> 7][t & 127] + red[2][t >> 7][t & 127] +
                                              red[3][t >> 7][t & 127];
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

extern "C" int be_attention_f32(const float* qkv, float* out, float* workspace, int B, int L, int l_valid, int H, void* stream) {
    BE_REQUIRE(qkv && out && workspace, "be_attention_f32: null pointer");
    BE_REQUIRE(B > 0 && H > 0 && L > 0 && L % 128 == 0, "be_attention_f32: L must be a multiple of 128 (got %d)", L);
    BE_REQUIRE(l_valid > L - 128 && l_valid <= L, "be_attention_f32: l_valid %d outside (L - 128, L] for L = %d", l_valid, L);
    BE_REQUIRE(be::aligned16(qkv) && be::aligned16(out) && be::aligned16(workspace), "be_attention_f32: 16-byte alignment");
    hipStream_t s = be::as_stream(stream);
    const size_t n = (size_t)B * H * L * DH;
    float *Q = workspace, *K = workspace + n, *Vt = workspace + 2 * n;
    const float qscale = 0.25f * 1.44269504088896340736f;            // 1/sqrt(16) * log2(e)
    int64_t g = ((int64_t)n + 255) / 256; if (g > 4096) g = 4096;
    hipLaunchKernelGGL(k_qkv_split, dim3((unsigned)g), dim3(256), 0, s, qkv, Q, K, Vt, B, L, H, qscale);
    // key slices when the query tiles alone leave the chip mostly empty (one 147x147 pair: 256 workgroups on 256 CUs)
    const int wgs = (L / 128) * B * H;
    const int nz = (L >= 2048 && wgs < 512) ? (wgs <= 256 ? 4 : 2) : 1;
    float* part = workspace + 3 * n;
    if (l_valid == L)
        hipLaunchKernelGGL((k_attention<false, false>), dim3(L / 128, B * H, nz), dim3(256), 0, s, Q, K, Vt, out, (float*)nullptr,
                           L, L, H, 0u, 0u, 1.0f, part, (uint16_t*)nullptr);
    else
        hipLaunchKernelGGL((k_attention<false, true>), dim3(L / 128, B * H, nz), dim3(256), 0, s, Q, K, Vt, out, (float*)nullptr,
                           L, l_valid, H, 0u, 0u, 1.0f, part, (uint16_t*)nullptr);
    if (nz > 1) {
        const int64_t total = (int64_t)B * H * L * 4;
        hipLaunchKernelGGL(k_attn_combine, dim3((unsigned)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256)), dim3(256), 0, s,
                           part, out, nz, (int64_t)B * H, L, H);
    }
    return be::check_launch("be_attention_f32");
}

namespace {
// nD = -keep * rowsum(dO o O) and nlse = -lse (the accumulator start values of the backward), keep = the dropout keep bits
struct TrainWs { float *Q, *K, *V, *Qt, *Kt, *Vt, *dOh, *dOt, *nD, *nlse; uint16_t* keep; };
TrainWs train_ws(float* w, int B, int L, int H) {
    const size_t n = (size_t)B * H * L * DH, r = (size_t)B * H * L;
    return {w, w + n, w + 2 * n, w + 3 * n, w + 4 * n, w + 5 * n, w + 6 * n, w + 7 * n, w + 8 * n, w + 8 * n + r,
            reinterpret_cast<uint16_t*>(w + 8 * n + 2 * r)};
}
size_t keep_lines(int B, int L, int H) { return (size_t)B * H * (L / KB) * (L / KB); }      // 64 halfwords each
int attn_args_ok(const char* who, int B, int L, int H, float p, int l_valid = -1) {
    BE_REQUIRE(B > 0 && H > 0 && L > 0 && L % 128 == 0 && L <= 65536, "%s: L must be a multiple of 128, <= 65536 (got %d)", who, L);
    BE_REQUIRE(l_valid == -1 || (l_valid > L - 128 && l_valid <= L), "%s: l_valid %d outside (L - 128, L] for L = %d", who, l_valid, L);
    BE_REQUIRE(p >= 0.f && p < 1.f, "%s: dropout probability %g outside [0,1)", who, (double)p);
    return BE_OK;
}
}  // namespace

extern "C" size_t be_attention_train_workspace_floats(int B, int L, int H) {
    return (size_t)8 * B * H * L * DH + (size_t)2 * B * H * L + keep_lines(B, L, H) * 32;      // 64 halfwords = 32 floats per line
}

extern "C" int be_attention_train_fwd_f32(const float* qkv, float* out, float* lse, float* workspace, int B, int L, int l_valid,
                                          int H, float dropout_p, uint32_t seed, void* stream) {
    BE_REQUIRE(qkv && out && lse && workspace, "be_attention_train_fwd_f32: null pointer");
    if (int rc = attn_args_ok("be_attention_train_fwd_f32", B, L, H, dropout_p, l_valid)) return rc;
    BE_REQUIRE(be::aligned16(qkv) && be::aligned16(out) && be::aligned16(workspace), "be_attention_train_fwd_f32: 16-byte alignment");
    hipStream_t s = be::as_stream(stream);
    const TrainWs w = train_ws(workspace, B, L, H);
    hipLaunchKernelGGL(k_qkv_split_train, dim3((unsigned)((int64_t)B * H * (L / 64))), dim3(256), 0, s, qkv, w.Q, w.K, w.V, w.Qt, w.Kt, w.Vt,
                       B, L, H, 0.25f * 1.44269504088896340736f);
    if (l_valid == L)
        hipLaunchKernelGGL((k_attention<true, false>), dim3(L / 128, B * H), dim3(256), 0, s, w.Q, w.K, w.Vt, out, lse, L, L, H, seed,
                           drop_threshold(dropout_p), 1.0f / (1.0f - dropout_p), (float*)nullptr, w.keep);
    else
        hipLaunchKernelGGL((k_attention<true, true>), dim3(L / 128, B * H), dim3(256), 0, s, w.Q, w.K, w.Vt, out, lse, L, l_valid, H,
                           seed, drop_threshold(dropout_p), 1.0f / (1.0f - dropout_p), (float*)nullptr, w.keep);
    return be::check_launch("be_attention_train_fwd_f32");
}

extern "C" size_t be_attention_bwd_scratch_floats(int B, int L, int H) {
    return (size_t)(L / (KB * FW)) * B * H * L * DH;              // one partial dQ per workgroup of FW x 32 keys
}

extern "C" int be_attention_bwd_f32(const float* qkv, const float* out, const float* lse, const float* dout, float* dqkv,
                                    float* workspace, float* scratch, int operands_ready, int B, int L, int l_valid, int H,
                                    float dropout_p, uint32_t seed, void* stream) {
    BE_REQUIRE(qkv && out && lse && dout && dqkv && workspace && scratch, "be_attention_bwd_f32: null pointer");
    if (int rc = attn_args_ok("be_attention_bwd_f32", B, L, H, dropout_p, l_valid)) return rc;
    BE_REQUIRE(be::aligned16(qkv) && be::aligned16(dqkv) && be::aligned16(workspace) && be::aligned16(lse) && be::aligned16(scratch),
               "be_attention_bwd_f32: 16-byte alignment");
    hipStream_t s = be::as_stream(stream);
    const TrainWs w = train_ws(workspace, B, L, H);
    int64_t g = ((int64_t)B * H * L + 255) / 256; if (g > 8192) g = 8192;
    const uint32_t th = drop_threshold(dropout_p);
    const float ik = 1.0f / (1.0f - dropout_p);
    if (!operands_ready) {        // the workspace of this layer's forward call was reused in between: split q/k/v again, and
        //                           write the keep bits the forward left there again
        hipLaunchKernelGGL(k_qkv_split_train, dim3((unsigned)((int64_t)B * H * (L / 64))), dim3(256), 0, s, qkv, w.Q, w.K, w.V, w.Qt, w.Kt,
                           w.Vt, B, L, H, 0.25f * 1.44269504088896340736f);
        if (th) {
            const int64_t lines = (int64_t)keep_lines(B, L, H);
            hipLaunchKernelGGL(k_attn_keep_bits, dim3((unsigned)((lines + 3) / 4 > 16384 ? 16384 : (lines + 3) / 4)), dim3(256), 0, s,
                               w.keep, lines, L, seed, th);
        }
    }
    hipLaunchKernelGGL(k_dout_prep, dim3((unsigned)g), dim3(256), 0, s, dout, out, lse, w.dOh, w.dOt, w.nD, w.nlse, 1.0f - dropout_p,
                       B, L, H);
    const dim3 grid(L / (KB * FW), B * H);
#define BE_ATTN_BWD(RAG, DROP, LV)                                                                                             \
    hipLaunchKernelGGL((k_attn_bwd_fused<RAG, DROP>), grid, dim3(64 * FW), 0, s, w.Q, w.K, w.V, w.Qt, w.Kt, w.dOh, w.dOt, w.nlse, \
                       w.nD, dqkv, scratch, L, LV, H, w.keep, ik)
    if (l_valid == L) { if (th) { BE_ATTN_BWD(false, true, L); } else { BE_ATTN_BWD(false, false, L); } }
    else { if (th) { BE_ATTN_BWD(true, true, l_valid); } else { BE_ATTN_BWD(true, false, l_valid); } }
#undef BE_ATTN_BWD
    const int rows_done = ((l_valid + KB - 1) / KB) * KB;            // query blocks the kernel visited; the padding rows get zeros
    hipLaunchKernelGGL(k_attn_dq_finish, dim3(4096), dim3(256), 0, s, scratch, dqkv, L / (KB * FW), B * H, L, H,
                       rows_done < L ? rows_done : L, 0.25f * ik);
    return be::check_launch("be_attention_bwd_f32");
}

extern "C" int be_attention_dropout_mask_f32(float* mask, int B, int L, int H, float dropout_p, uint32_t seed, void* stream) {
    BE_REQUIRE(mask, "be_attention_dropout_mask_f32: null pointer");
    if (int rc = attn_args_ok("be_attention_dropout_mask_f32", B, L, H, dropout_p)) return rc;
    const int64_t total = (int64_t)B * H * L * L;
    int64_t g = (total + 255) / 256; if (g > 8192) g = 8192;
    hipLaunchKernelGGL(k_attn_dropout_mask, dim3((unsigned)g), dim3(256), 0, be::as_stream(stream), mask, (int64_t)B * H, L,
                       seed, drop_threshold(dropout_p));
    return be::check_launch("be_attention_dropout_mask_f32");
}

extern "C" size_t be_attention_train_keep_offset_floats(int B, int L, int H) {
    return (size_t)8 * B * H * L * DH + (size_t)2 * B * H * L;
}

extern "C" int be_attention_keep_bits_u16(uint16_t* keep, int B, int L, int H, float dropout_p, uint32_t seed, void* stream) {
    BE_REQUIRE(keep, "be_attention_keep_bits_u16: null pointer");
    if (int rc = attn_args_ok("be_attention_keep_bits_u16", B, L, H, dropout_p)) return rc;
    const int64_t lines = (int64_t)keep_lines(B, L, H);
    hipLaunchKernelGGL(k_attn_keep_bits, dim3((unsigned)((lines + 3) / 4 > 16384 ? 16384 : (lines + 3) / 4)), dim3(256), 0,
                       be::as_stream(stream), keep, lines, L, seed, drop_threshold(dropout_p));
    return be::check_launch("be_attention_keep_bits_u16");
}

extern "C" int be_dropout_f32(const float* x, const float* gate, float* y, int64_t n, float dropout_p, uint32_t seed,
                              uint32_t site, void* stream) {
    BE_REQUIRE(x && y && n > 0 && n < ((int64_t)1 << 32), "be_dropout_f32: bad arguments");
    BE_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, "be_dropout_f32: dropout probability outside [0,1)");
    int64_t g = (n / 4 + 255) / 256; if (g > 4096) g = 4096; if (g < 1) g = 1;
    hipLaunchKernelGGL(k_dropout, dim3((unsigned)g), dim3(256), 0, be::as_stream(stream), x, gate, y, n, site_key(seed, site),
                       drop_threshold(dropout_p), 1.0f / (1.0f - dropout_p));
    return be::check_launch("be_dropout_f32");
}

extern "C" int be_add_layernorm_train_f32(const float* x, const float* res, const float* gamma, const float* beta, float* v,
                                          float* y, int64_t rows, int D, float eps, float dropout_p, uint32_t seed,
                                          uint32_t site, void* stream) {
    BE_REQUIRE(x && gamma && beta && v && y && rows > 0, "be_add_layernorm_train_f32: bad arguments");
    BE_REQUIRE(D == 128, "be_add_layernorm_train_f32: D must be 128 (got %d)", D);
    BE_REQUIRE(rows * D < ((int64_t)1 << 32) && dropout_p >= 0.f && dropout_p < 1.f, "be_add_layernorm_train_f32: bad size / p");
    hipLaunchKernelGGL(k_add_layernorm_train, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, be::as_stream(stream), x, res,
                       gamma, beta, v, y, rows, eps, site_key(seed, site), drop_threshold(dropout_p), 1.0f / (1.0f - dropout_p));
    return be::check_launch("be_add_layernorm_train_f32");
}

extern "C" size_t be_layernorm_bwd_partial_floats(int64_t rows, int D) {
    return (size_t)((rows + LN_ROWS - 1) / LN_ROWS) * 2 * D;
}

extern "C" int be_layernorm_bwd_f32(const float* dy, const float* v, const float* gamma, float* dv, float* dx, float* partial,
                                    int64_t rows, int D, float eps, float dropout_p, uint32_t seed, uint32_t site,
                                    void* stream) {
    BE_REQUIRE(dy && v && gamma && partial && rows > 0, "be_layernorm_bwd_f32: bad arguments");
    BE_REQUIRE(D == 128, "be_layernorm_bwd_f32: D must be 128 (got %d)", D);
    BE_REQUIRE(rows * D < ((int64_t)1 << 32) && dropout_p >= 0.f && dropout_p < 1.f, "be_layernorm_bwd_f32: bad size / p");
    hipLaunchKernelGGL(k_layernorm_bwd, dim3((unsigned)((rows + LN_ROWS - 1) / LN_ROWS)), dim3(256), 0, be::as_stream(stream),
                       dy, v, gamma, dv, dx, partial, rows, eps, site_key(seed, site), drop_threshold(dropout_p),
                       1.0f / (1.0f - dropout_p));
    return be::check_launch("be_layernorm_bwd_f32");
}

extern "C" size_t be_attention_workspace_floats(int B, int L, int H) {
    return (size_t)3 * B * H * L * DH + (size_t)4 * B * H * L * (DH + 2);      // q / k / v^T + up to 4 key-slice partials
}

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
