// LocalStage's head for LARGE batches in ONE kernel: conv1 (7x7, 3 -> 64, padding 3) + folded BatchNorm + Smish + MaxPool(3, 2, 1),
// models/local_stage.py:34-37,42,64-65 - image-major (round 5).
//
// The pixel-major conv1 (be_conv_pm.hip, ROW8) writes its 21 x 21 x 64 map to HBM (925 MB per 8192 patches) only for
// k_maxpool_nhwc to read it back and keep a quarter: 0.79 + 0.25 ms of a 12 ms step.  A pixel-major tile cannot pool (a 3 x 3
// window spans nine tiles).  Here a workgroup owns WHOLE images, one after the other:
//   - the image's padded staging (21 rows x 28 pixels x 4 channels, be_nchw3_to_nhwc4p_f32) sits in LDS between three zero rows
//     above and below: every tap of every output pixel is an in-bounds 16-byte LDS read, no im2col copy exists anywhere;
//   - the weights are STATIONARY in registers: a wave owns 32 output channels and keeps their 7 x 8 x 4 packed taps (28 quads per
//     lane, 112 VGPRs) for the whole launch; 14 row tiles of 32 pixels x 2 channel tiles per image, two row tiles in flight per wave;
//   - the epilogue (bias, Smish) takes the 3 x 3 / stride 2 maxima on the fly: every value goes to the one, two or four pooled
//     cells whose window holds its pixel with an LDS float maximum (ds_max_f32) - the 21 x 21 x 64 map exists nowhere; after a
//     barrier the workgroup writes the 11 x 11 x 64 pooled map (31 KB per image instead of 113 + 31 of HBM writes) and resets the
//     cells.  71 KB of LDS per workgroup: two workgroups per CU, one's epilogue and barriers under the other's MFMAs.
// Arithmetic: the same fp32 MFMA chain per output element as k_conv_igemm / k_conv_pm in ROW8 mode - kernel rows top to bottom,
// pixel pairs left to right, channels 0, 1, 2, lane half = pixel of the pair - with the rows above / below the image multiplied by
// zeros instead of skipped (x + 0 * w = x), the same bias + Smish, the same maximum: bit-identical to conv1 followed by the pool
// kernel (tests/test_hip_parity.py holds it to that) for FINITE activations.  A NaN activation is dropped here (the maxima start
// from -inf and fmaxf / ds_max_f32 return the other operand), where torch.nn.MaxPool2d propagates it: a network whose conv1 emits NaN
// is broken either way, and the reference's checkpoints do not.
#include "be_common.h"
#include "be_device_math.h"
#include <cstdlib>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int R = 21, HW = R * R, WROW = 28, PADR = R + 6, COUT = 64, KTOT = 224, OR_ = 11;
// pixels per image row IN LDS: 37 = 21 + 16.  The fragment reads are ds_read_b128 of 32 consecutive output pixels (16 B each); with
// the staging's 28 pixels per row a tile that wraps into the next image row puts its later lanes 7 pixels further, onto the banks of
// earlier lanes of the same 16-lane group (round 6 counters: 39 % of the LDS cycles were bank conflicts AFTER the pooled cells had
// been spread - they were these reads).  With 37 the LDS pixel index of pixel p is congruent to p modulo 16: conflict-free as one row.
constexpr int LROW = 37;
constexpr int IMG_FLOATS = PADR * LROW * 4;            // 3996: padded image in LDS
constexpr int IMG_QUADS = R * WROW;                    // 588 float4 of real rows
constexpr int POOL_FLOATS = OR_ * OR_ * COUT;          // 7744: the pooled map
// floats between pooled cells in LDS: the two halves of a wave work on pixels 4 apart = cells 2 apart; with 64 floats per cell those
// are 128 floats = the same banks; with 80 they are 160 floats = 32 banks apart (SQ_LDS_BANK_CONFLICT 34 % of the LDS cycles)
constexpr int CELL = COUT + 16;
constexpr int POOL_LDS = OR_ * OR_ * CELL;             // 9680
constexpr size_t LDS_BYTES = (size_t)(2 * IMG_FLOATS + POOL_LDS) * sizeof(float);          // 70 688: two workgroups per CU

__device__ __forceinline__ void lds_fmax(float* p, float v) {
    __builtin_amdgcn_ds_fmaxf((__attribute__((address_space(3))) float*)p, v, __ATOMIC_RELAXED, __MEMORY_SCOPE_WRKGRP, false);
}

__global__ __launch_bounds__(256, 2)
void k_conv1_pool(const float* __restrict__ x4p, const float* __restrict__ w, const float* __restrict__ bias, float* __restrict__ y,
                  int64_t n) {
    extern __shared__ __attribute__((aligned(16))) float smem_c1[];
    float* img_lds = smem_c1;                          // [2][PADR][WROW][4]
    float* pool = smem_c1 + 2 * IMG_FLOATS;            // [11][11][CELL]: running maxima of the image in flight (COUT of every CELL floats used)
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
    const int ntile = wave & 1, mpar = wave >> 1;
    const int cout = ntile * 32 + li;

    // zero rows above and below both image buffers (never written again)
    for (int i = tid; i < 2 * 2 * 3 * LROW; i += 256) {
        const int buf = i / (2 * 3 * LROW), r = i % (2 * 3 * LROW);
        const int row = r < 3 * LROW ? r / LROW : (R + 3) + (r - 3 * LROW) / LROW, col = r % LROW;
        reinterpret_cast<f32x4*>(img_lds + buf * IMG_FLOATS)[row * LROW + col] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // the wave's weights: quad (kh, g) of lane (li, lh) = taps of pixel 2 g + lh of kernel row kh, channels 0..3, for its channel
    f32x4 bw[7][4];
#pragma unroll
    for (int kh = 0; kh < 7; ++kh)
#pragma unroll
        for (int g = 0; g < 4; ++g) bw[kh][g] = *reinterpret_cast<const f32x4*>(w + (size_t)cout * KTOT + kh * 32 + (2 * g + lh) * 4);
    const float bs = bias ? bias[cout] : 0.0f;
    for (int i = tid; i < POOL_LDS / 4; i += 256) reinterpret_cast<f32x4*>(pool)[i] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};

    // first image of this workgroup into buffer 0
    int64_t img = blockIdx.x;
    f32x4 pre[3];
    auto fetch = [&](int64_t im) {
        const f32x4* src = reinterpret_cast<const f32x4*>(x4p + im * (int64_t)(R * WROW * 4));
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int q = tid + 256 * j;
            pre[j] = q < IMG_QUADS ? src[q] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto stash = [&](int buf) {
        f32x4* dst = reinterpret_cast<f32x4*>(img_lds + buf * IMG_FLOATS) + 3 * LROW;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int q = tid + 256 * j;
            if (q < IMG_QUADS) dst[(q / WROW) * LROW + q % WROW] = pre[j];        // staging row of 28 pixels -> LDS row of 37
        }
    };
    if (img < n) { fetch(img); stash(0); }
    __syncthreads();

    int buf = 0;
    for (; img < n; img += gridDim.x, buf ^= 1) {
        const int64_t nxt = img + gridDim.x;
        if (nxt < n) fetch(nxt);                       // the next image's loads fly under this image's MFMAs
        const float* im = img_lds + buf * IMG_FLOATS;
        // ---- this wave's 7 row tiles, two at a time: t = mpar + 2 s
#pragma unroll 1
        for (int s = 0; s < 7; s += 2) {
            const int t0 = mpar + 2 * s, t1 = t0 + 2;
            const bool two = s + 1 < 7;
            const int p0 = min(t0 * 32 + li, HW - 1), p1 = min((two ? t1 : t0) * 32 + li, HW - 1);
            const float* a0 = im + ((p0 / R) * LROW + p0 % R + lh) * 4;
            const float* a1 = im + ((p1 / R) * LROW + p1 % R + lh) * 4;
            f32x16 acc0, acc1;
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
#pragma unroll
            for (int kh = 0; kh < 7; ++kh)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 f0 = *reinterpret_cast<const f32x4*>(a0 + kh * LROW * 4 + g * 8);
                    const f32x4 f1 = *reinterpret_cast<const f32x4*>(a1 + kh * LROW * 4 + g * 8);
                    const f32x4 b = bw[kh][g];
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(f0.x, b.x, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(f1.x, b.x, acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(f0.y, b.y, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(f1.y, b.y, acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(f0.z, b.z, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(f1.z, b.z, acc1, 0, 0, 0);
                }
            // D[row][col]: col = lane & 31 (the channel), row = (r & 3) + 8 (r >> 2) + 4 lh (the pixel of the tile): a lane's registers
            // 4 c .. 4 c + 3 are FOUR CONSECUTIVE pixels.  Pixel (py, px) lies in the windows of the pooled rows py >> 1 and, for odd py,
            // (py >> 1) + 1 (likewise the columns).  Round 6: the four pixels of a group (same image row: 86 % of the groups) meet THREE
            // pooled columns - take those maxima in registers first (a maximum is exact and order-free) and send 3 (6 for odd py) LDS
            // maxima per group instead of 6 (12): round 5 counted 34 % of this kernel's LDS cycles as bank conflicts and MFMA-busy 0.62.
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int qb = (h ? t1 : t0) * 32 + 8 * c4 + 4 * lh;
                    if ((h && !two) || qb >= HW) continue;
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = be::smish((h ? acc1[4 * c4 + e] : acc0[4 * c4 + e]) + bs);
                    const int py = (qb * 3121) >> 16, px0 = qb - py * R;           // qb / 21 for qb < 441
                    if (px0 + 3 < R) {                          // the group stays in its image row (and inside the image: qb + 3 < 441)
                        const int a_ = px0 >> 1;
                        float ma, mb, mc;
                        if (px0 & 1) { ma = v[0]; mb = fmaxf(v[0], fmaxf(v[1], v[2])); mc = fmaxf(v[2], v[3]); }
                        else { ma = fmaxf(v[0], v[1]); mb = fmaxf(v[1], fmaxf(v[2], v[3])); mc = v[3]; }
                        float* c = pool + ((py >> 1) * OR_ + a_) * CELL + cout;
                        lds_fmax(c, ma); lds_fmax(c + CELL, mb); lds_fmax(c + 2 * CELL, mc);
                        if (py & 1) { c += OR_ * CELL; lds_fmax(c, ma); lds_fmax(c + CELL, mb); lds_fmax(c + 2 * CELL, mc); }
                    } else {                                    // the group wraps into the next image row / ends with the image: pixel by pixel
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int q = qb + e;
                            if (q >= HW) continue;
                            const int qy = (q * 3121) >> 16, qx = q - qy * R;
                            float* c = pool + ((qy >> 1) * OR_ + (qx >> 1)) * CELL + cout;
                            lds_fmax(c, v[e]);
                            if (qx & 1) lds_fmax(c + CELL, v[e]);
                            if (qy & 1) {
                                lds_fmax(c + OR_ * CELL, v[e]);
                                if (qx & 1) lds_fmax(c + (OR_ + 1) * CELL, v[e]);
                            }
                        }
                    }
                }
            }
        }
        if (nxt < n) stash(buf ^ 1);
        __syncthreads();
        // ---- the pooled map -> y [img][11][11][64]; the cells go back to -inf for the next image
        float* yo = y + img * (int64_t)POOL_FLOATS;
        for (int i = tid; i < POOL_FLOATS / 4; i += 256) {
            f32x4* cell = reinterpret_cast<f32x4*>(pool + (i >> 4) * CELL) + (i & 15);       // 16 quads of real channels per cell
            reinterpret_cast<f32x4*>(yo)[i] = *cell;
            *cell = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        }
        __syncthreads();
    }
}

}  // namespace

extern "C" int be_conv7x7_pool_nhwc4p_f32(const float* x4p, int64_t n, const float* packed_w, const float* packed_bias, float* y,
                                          void* stream) {
    BE_REQUIRE(x4p && packed_w && y, "be_conv7x7_pool_nhwc4p_f32: null pointer");
    BE_REQUIRE(n > 0 && n < ((int64_t)1 << 31), "be_conv7x7_pool_nhwc4p_f32: n outside (0, 2^31)");
    BE_REQUIRE(be::aligned16(x4p) && be::aligned16(packed_w) && be::aligned16(y), "be_conv7x7_pool_nhwc4p_f32: 16-byte alignment");
    static be::DeviceFlags attr_set{};
    if (int rc_ = be::ensure_dynamic_lds(reinterpret_cast<const void*>(&k_conv1_pool), LDS_BYTES, attr_set)) return rc_;
    hipStream_t s = be::as_stream(stream);
    const int cus = be::device_cu_count();
    const unsigned grid = (unsigned)(n < 2 * cus ? n : 2 * cus);      // two workgroups per CU (55 KB of LDS each), images dealt round-robin
    {
        // algorithmic: 2 * M * 147 * 64; executed: 14 row tiles x 2 channel tiles x 84 MFMAs of 4096 FLOP per image
        const double M = (double)n * HW;
        be::ProfileScope prof(s, BE_KERNEL_CONV1_POOL, 2.0 * M * 147.0 * COUT, 4.0 * (M * 3.0 + 147.0 * COUT + (double)n * OR_ * OR_ * COUT),
                              (double)n * 14.0 * 2.0 * 84.0 * 4096.0);
        hipLaunchKernelGGL(k_conv1_pool, dim3(grid), dim3(256), LDS_BYTES, s, x4p, packed_w, packed_bias, y, n);
    }
    return be::check_launch("be_conv7x7_pool_nhwc4p_f32");
}
