// LocalStage.forward (eval) as a chain of launches on one stream: models/local_stage.py:63-73.
//   conv7x7+BN+Smish -> maxpool(3,2,1) -> block 64->96 @11^2 -> maxpool(3,2,1) -> blocks 96->256, 256->384,
//   384->256 @6^2 -> maxpool(2,2) -> flatten -> Linear 2304->1024 + BN1d + Smish -> Linear 1024->10
// Activations are NHWC in a caller-provided workspace; the batch is walked in sub-batches (default 8192 patches,
// measured best) so the workspace stays bounded whatever N is.  Per sub-batch: staging, conv1, 2-3 pools, layer0 as two direct
// launches (conv1; conv2 with the downsample fused in), layers 1-3 each as a 1x1 downsample launch + the Winograd pair
// (input transform, 40 GEMMs, output+input transform, 40 GEMMs, output transform - layer3's with the 2x2 max-pool in it;
// be_wino.hip), fc.1, fc.4.  Sub-batches of 512 patches and more take the LDS-DMA kernels (be_conv_pm.hip for conv1 on a
// row-padded staging and for layer0, the row GEMM of be_wino.hip for the 1x1s and fc.1); smaller ones k_conv_igemm - same
// results bit for bit.  No allocation, no synchronisation, no process-wide state: graph-capturable, re-entrant.  opts->winograd = 0
// runs layers 1-3 as direct launches like layer0.
#include "be_common.h"
#include <cstdlib>

namespace {

struct LayerSpec { int cout, cin, ks, chw_hw; };
// state_dict() order; conv1 consumes the NHWC4 staging of the NCHW input
constexpr LayerSpec kLayers[15] = {
    {64, 3, 7, 0},
    {96, 64, 3, 0},   {96, 96, 3, 0},   {96, 64, 1, 0},
    {256, 96, 3, 0},  {256, 256, 3, 0}, {256, 96, 1, 0},
    {384, 256, 3, 0}, {384, 384, 3, 0}, {384, 256, 1, 0},
    {256, 384, 3, 0}, {256, 256, 3, 0}, {256, 384, 1, 0},
    {1024, 2304, 1, 9},
    {10, 1024, 1, 0},
};

inline size_t cout_pad(int c) { return (size_t)((c + 31) / 32 * 32); }

struct PackedLayout {
    size_t w_off[15], b_off[15], total;
    size_t uw_off[15], ub_off[15];     // Winograd form (be_wino_math.h: 8x5 tiles) of the 3x3 convs on the 6x6 maps (layers 4,5,7,8,10,11)
    size_t dw_off[15], db_off[15];     // their blocks' 1x1 downsample convs as stand-alone convs (layers 6, 9, 12)
    PackedLayout() {
        size_t o = 0;
        for (int i = 0; i < 15; ++i) {
            const bool conv2 = i == 2 || i == 5 || i == 8 || i == 11;        // carries its block's downsample too
            const bool ds = i == 3 || i == 6 || i == 9 || i == 12;            // ... which therefore has no slot
            w_off[i] = o;
            if (conv2) o += be_conv_fused2_packed_floats(kLayers[i].cout, kLayers[i].cin, kLayers[i].ks, kLayers[i + 1].cin);
            else if (!ds) o += be_conv_packed_floats(kLayers[i].cout, kLayers[i].cin, kLayers[i].ks);
            b_off[i] = o;
            if (!ds) o += cout_pad(kLayers[i].cout);
        }
        for (int i = 4; i < 13; ++i) {
            const bool ds = i == 6 || i == 9 || i == 12;
            uw_off[i] = ub_off[i] = dw_off[i] = db_off[i] = 0;
            if (ds) {
                dw_off[i] = o; o += be_conv_packed_floats(kLayers[i].cout, kLayers[i].cin, 1);
                db_off[i] = o; o += cout_pad(kLayers[i].cout);
            } else {
                uw_off[i] = o; o += be_wino_packed_floats(kLayers[i].cout, kLayers[i].cin);
                ub_off[i] = o; o += cout_pad(kLayers[i].cout);
            }
        }
        zero_off = o; o += 384;            // a zero "bias" for the Winograd blocks' downsample convolutions (their bias lives in conv2's)
        total = o;
    }
    size_t zero_off;
};
const PackedLayout& layout() { static PackedLayout l; return l; }

// workspace regions, floats per patch (lifetimes: see be_local_stage_forward_f32; RA holds conv1's output, then
// each block's intermediate t)
constexpr size_t RA = 28224, RB = 13824, RC = 13824;
// Winograd path: RW = transform-domain input + output of the widest layer (room for 100 values per channel: 25 positions x 4 tiles; the 8x5 tiles use 80,
// 384 + 384 channels), RR = the block's downsample branch [6,6,384]
constexpr size_t RW = 100 * (384 + 384), RR = 13824;
constexpr size_t WS_FLOATS_PER_PATCH = RA + RB + RC + RW + RR;
constexpr int kDefaultChunk = 8192;   // measured: 8192 > 4096 > 2048 (fewer partial rounds of the 512 resident blocks)
inline int chunk_of(int chunk) { return chunk > 0 ? chunk : kDefaultChunk; }

}  // namespace

extern "C" size_t be_local_stage_packed_floats(void) { return layout().total; }

extern "C" size_t be_local_stage_workspace_bytes(int64_t n, int chunk) {
    if (n <= 0) return 0;
    const int64_t nb = n < chunk_of(chunk) ? n : chunk_of(chunk);
    return (size_t)nb * WS_FLOATS_PER_PATCH * sizeof(float);
}

extern "C" int be_local_stage_pack_f32(const float* const* t, float bn_eps, float* packed, void* stream) {
    BE_REQUIRE(t && packed, "be_local_stage_pack_f32: null pointer");
    for (int i = 0; i < BE_LOCAL_STAGE_NTENSORS; ++i) BE_REQUIRE(t[i], "be_local_stage_pack_f32: tensor %d is null", i);
    const PackedLayout& L = layout();
    for (int i = 0; i < 13; ++i) {                    // conv + BatchNorm2d pairs
        const float* const* e = t + 6 * i;            // weight, bias, gamma, beta, mean, var
        const bool conv2 = i == 2 || i == 5 || i == 8 || i == 11, ds = i == 3 || i == 6 || i == 9 || i == 12;
        if (ds) continue;                              // packed together with the block's conv2
        int rc;
        if (conv2) {
            const float* const* d = t + 6 * (i + 1);  // the block's downsample conv + BN
            rc = be_conv_pack_fused2_f32(e[0], e[1], e[2], e[3], e[4], e[5], d[0], d[1], d[2], d[3], d[4], d[5], bn_eps,
                                         kLayers[i].cout, kLayers[i].cin, kLayers[i].ks, kLayers[i + 1].cin,
                                         packed + L.w_off[i], packed + L.b_off[i], stream);
        } else {
            rc = be_conv_pack_f32(e[0], e[1], e[2], e[3], e[4], e[5], bn_eps, kLayers[i].cout, kLayers[i].cin,
                                  kLayers[i].ks, 0, packed + L.w_off[i], packed + L.b_off[i], stream);
        }
        if (rc) return rc;
    }
    for (int i = 4; i < 13; ++i) {                    // 6x6 blocks again, in the form the Winograd path reads
        const float* const* e = t + 6 * i;
        const bool ds = i == 6 || i == 9 || i == 12;
        const int rc = ds ? be_conv_pack_f32(e[0], e[1], e[2], e[3], e[4], e[5], bn_eps, kLayers[i].cout, kLayers[i].cin, 1, 0,
                                             packed + L.dw_off[i], packed + L.db_off[i], stream)
                          : be_wino_pack_f32(e[0], e[1], e[2], e[3], e[4], e[5], bn_eps, kLayers[i].cout, kLayers[i].cin,
                                             packed + L.uw_off[i], packed + L.ub_off[i], stream);
        if (rc) return rc;
    }
    // Winograd blocks: the downsample's (folded) bias moves into conv2's, so that the 1x1 downsample itself is a bias-free GEMM -
    // raw accumulators, which the weight-stationary GEMM kernel can write (be::gemm_rows_ws) - and joins in conv2's output
    // transform exactly as before: y = act(A^T M A + (b2 + b_ds) + x W_ds)
    for (int l0 = 4; l0 < 13; l0 += 3) {
        if (int rc = be::vec_add_inplace(packed + L.ub_off[l0 + 1], packed + L.db_off[l0 + 2], (int)cout_pad(kLayers[l0].cout), stream)) return rc;
    }
    if (hipMemsetAsync(packed + L.zero_off, 0, 384 * sizeof(float), be::as_stream(stream)) != hipSuccess)
        return be::fail(BE_ELAUNCH, "be_local_stage_pack_f32: hipMemsetAsync failed");
    const float* const* f = t + 78;                   // fc.1.w, fc.1.b, fc.2.{gamma,beta,mean,var}, fc.4.w, fc.4.b
    int rc = be_conv_pack_f32(f[0], f[1], f[2], f[3], f[4], f[5], bn_eps, 1024, 2304, 1, 9,
                              packed + L.w_off[13], packed + L.b_off[13], stream);
    if (rc) return rc;
    return be_conv_pack_f32(f[6], f[7], nullptr, nullptr, nullptr, nullptr, bn_eps, 10, 1024, 1, 0,
                            packed + L.w_off[14], packed + L.b_off[14], stream);
}

namespace {

int conv(const float* packed, int li, const float* x, const float* res, float* y, int n, int hw, int act, int ldy,
         void* stream) {
    const PackedLayout& L = layout();
    be_conv_desc d;
    d.n = n; d.h = hw; d.w = hw;
    d.cin = li == 0 ? 4 : kLayers[li].cin;
    d.cout = kLayers[li].cout; d.ksize = kLayers[li].ks; d.act = act;
    return be_conv_nhwc_f32(&d, x, packed + L.w_off[li], packed + L.b_off[li], res, y, ldy, stream);
}

// ResidualBlock (models/local_stage.py:20-28): Smish(BN(conv3(Smish(BN(conv3(x))))) + BN(conv1x1(x))) in two launches:
// the downsample 1x1 rides in the K loop of conv2 (be_conv_nhwc_fused2_f32), no residual tensor in HBM
int block(const float* packed, int l0, const float* x, float* t, float* o, int n, int hw, void* stream) {
    const int c = kLayers[l0].cout;
    int rc;
    if ((rc = conv(packed, l0, x, nullptr, t, n, hw, 1, c, stream))) return rc;
    const PackedLayout& L = layout();
    be_conv_desc d;
    d.n = n; d.h = hw; d.w = hw; d.cin = kLayers[l0 + 1].cin; d.cout = c; d.ksize = 3; d.act = 1;
    return be_conv_nhwc_fused2_f32(&d, t, x, kLayers[l0 + 2].cin, packed + L.w_off[l0 + 1], packed + L.b_off[l0 + 1], o, c, stream);
}

// The same block on a 6x6 map with both 3x3 convolutions in Winograd F(3x3,3x3) form (be_wino.hip): 2.56x fewer multiplies
// than the direct form; the 1x1 downsample runs as its own convolution into `r` and joins in the output transform of conv2.
int block_wino(const float* packed, int l0, const float* x, float* t, float* o, float* r, float* w, int n, void* stream, int pool2 = 0) {
    const PackedLayout& L = layout();
    const int c = kLayers[l0].cout;
    (void)t;                                          // conv1's 6x6 result only ever exists in registers (k_wino_out_in)
    // the 1x1 downsample, bias-free (its bias sits in conv2's, see be_local_stage_pack_f32): large sub-batches on the
    // weight-stationary GEMM (raw accumulators; 128 TFLOP/s where the row GEMM does 92-119), others on the general kernel with a zero bias
    {
        int rc = be::gemm_rows_ws(x, (int64_t)n * 36, kLayers[l0 + 2].cin, packed + L.dw_off[l0 + 2], c, r, c, stream);
        if (rc < 0) return rc;
        if (rc > 0) {
            be_conv_desc d;
            d.n = n; d.h = 6; d.w = 6; d.cin = kLayers[l0 + 2].cin; d.cout = c; d.ksize = 1; d.act = 0;
            if ((rc = be_conv_nhwc_f32(&d, x, packed + L.dw_off[l0 + 2], packed + L.zero_off, nullptr, r, c, stream))) return rc;
        }
    }
    return be::wino_pair(x, packed + L.uw_off[l0], packed + L.ub_off[l0], 1, packed + L.uw_off[l0 + 1], packed + L.ub_off[l0 + 1], r,
                         1, o, n, kLayers[l0].cin, c, c, w, (size_t)n * RW, stream, pool2);
}

}  // namespace

namespace {

// x != nullptr: flat patches [n,3,21,21]; else the patches are gathered from `view` (image pair, f2)
int forward_impl(const float* packed, const float* x, const be_patch_view* view, int64_t P, float* out, int64_t n,
                 void* workspace, size_t workspace_bytes, const be_local_stage_opts* opts, void* stream, const char* who) {
    BE_REQUIRE(n >= 0, "%s: n < 0", who);
    BE_REQUIRE(!opts || opts->chunk >= 0, "%s: opts->chunk < 0", who);
    const int g_wino = opts ? (opts->winograd != 0) : 1;
    const int g_chunk = chunk_of(opts ? opts->chunk : 0);
    if (n == 0) return BE_OK;
    BE_REQUIRE(packed && (x || view) && out && workspace, "%s: null pointer", who);
    BE_REQUIRE(be::aligned16(workspace) && be::aligned16(packed), "%s: workspace / packed must be 16-byte aligned", who);
    if (workspace_bytes < be_local_stage_workspace_bytes(n, g_chunk))
        return be::fail(BE_EWORKSPACE, "%s: workspace %zu B < %zu B needed", who, workspace_bytes,
                        be_local_stage_workspace_bytes(n, g_chunk));
    float* ws = static_cast<float*>(workspace);
    for (int64_t first = 0; first < n; first += g_chunk) {
        const int nb = (int)((n - first) < g_chunk ? (n - first) : g_chunk);
        float* ra = ws;
        float* rb = ra + (size_t)nb * RA;
        float* rc_ = rb + (size_t)nb * RB;
        float* rw = rc_ + (size_t)nb * RC;                // Winograd transform-domain buffers
        float* rr = rw + (size_t)nb * RW;                 // downsample branch of the current block
        const bool wino = g_wino != 0;
        int rc;
        // x4 -> RB ; conv1 -> RA ; pool -> RB(after x4 is dead: RB is big enough to hold both side by side)
        float* x4 = rb;                                   // nb*1764
        float* p1 = rb + (size_t)nb * 1764 * 2;           // nb*7744, placed behind x4 (1764*2 + 7744 <= 13824)
        // large sub-batches: conv1 on the pixel-major LDS-DMA kernel, which reads a staging with 28 pixels per row (3 zero
        // pixels left, 4 right; 2352 floats per patch, still in front of p1)
        static const bool no_pm = getenv("BE_NO_CONV_PM") != nullptr;
        bool pooled = false;
        if (nb >= 512 && !no_pm) {
            if (x) rc = be_nchw3_to_nhwc4p_f32(x + first * 3 * BE_NPIX, x4, nb, BE_R, BE_R, 28, stream);
            else rc = be_view_to_nhwc4p_f32(view, P, first, x4, nb, 28, stream);
            if (rc) return rc;
            const PackedLayout& L = layout();
            // conv1 + Smish + the first max-pool in one image-major launch (be_conv1_pool.hip): the 21 x 21 x 64 map never reaches
            // HBM; bit-identical to the pixel-major conv1 followed by the pool kernel (BE_NO_CONV1_POOL=1: that pair, for A/B runs)
            static const bool no_c1p = getenv("BE_NO_CONV1_POOL") != nullptr;
            if (!no_c1p) {
                if ((rc = be_conv7x7_pool_nhwc4p_f32(x4, nb, packed + L.w_off[0], packed + L.b_off[0], p1, stream))) return rc;
                pooled = true;
            } else {
                be_conv_desc d;
                d.n = nb; d.h = 21; d.w = 21; d.cin = 4; d.cout = 64; d.ksize = 7; d.act = 1;
                if ((rc = be_conv7x7_nhwc4p_f32(&d, x4, 28, packed + L.w_off[0], packed + L.b_off[0], ra, 64, stream))) return rc;
            }
        } else {
            if (x) rc = be_nchw3_to_nhwc4_f32(x + first * 3 * BE_NPIX, x4, nb, BE_NPIX, stream);
            else rc = be_view_to_nhwc4_f32(view, P, first, x4, nb, stream);
            if (rc) return rc;
            if ((rc = conv(packed, 0, x4, nullptr, ra, nb, 21, 1, 64, stream))) return rc;
        }
        if (!pooled && (rc = be_maxpool_nhwc_f32(ra, p1, nb, 21, 21, 64, 3, 2, 1, stream))) return rc;
        // layer0 @11x11: t in RA, out in RC
        if ((rc = block(packed, 1, p1, ra, rc_, nb, 11, stream))) return rc;
        float* p2 = rb;                                   // nb*3456
        if ((rc = be_maxpool_nhwc_f32(rc_, p2, nb, 11, 11, 96, 3, 2, 1, stream))) return rc;
        // layer1: in RB, t RA, out RC
        if ((rc = wino ? block_wino(packed, 4, p2, ra, rc_, rr, rw, nb, stream) : block(packed, 4, p2, ra, rc_, nb, 6, stream))) return rc;
        // layer2: in RC, t RA, out RB
        if ((rc = wino ? block_wino(packed, 7, rc_, ra, rb, rr, rw, nb, stream) : block(packed, 7, rc_, ra, rb, nb, 6, stream))) return rc;
        // layer3: in RB, t RA, out RC; then maxpool(2,2) -> p3 [nb,3,3,256] = the (H,W,C) flatten.  Winograd path: the
        // output transform pools in registers and writes p3 directly (into RC: RB is still the block's input)
        float* p3;
        if (wino) {
            p3 = rc_;
            if ((rc = block_wino(packed, 10, rb, ra, p3, rr, rw, nb, stream, 1))) return rc;
        } else {
            if ((rc = block(packed, 10, rb, ra, rc_, nb, 6, stream))) return rc;
            p3 = rb;                                      // nb*2304
            if ((rc = be_maxpool_nhwc_f32(rc_, p3, nb, 6, 6, 256, 2, 2, 0, stream))) return rc;
        }
        float* f1 = ra;                                   // nb*1024
        if ((rc = conv(packed, 13, p3, nullptr, f1, nb, 1, 1, 1024, stream))) return rc;
        if ((rc = conv(packed, 14, f1, nullptr, out + first * BE_LOCAL_OUT, nb, 1, 0, BE_LOCAL_OUT, stream))) return rc;
    }
    return BE_OK;
}

}  // namespace

extern "C" int be_local_stage_forward_f32(const float* packed, const float* x, float* out, int64_t n,
                                          void* workspace, size_t workspace_bytes, const be_local_stage_opts* opts, void* stream) {
    BE_REQUIRE(x || n == 0, "be_local_stage_forward_f32: null pointer");
    return forward_impl(packed, x, nullptr, 1, out, n, workspace, workspace_bytes, opts, stream, "be_local_stage_forward_f32");
}

extern "C" int be_local_stage_forward_view_f32(const float* packed, const be_patch_view* view, int64_t patches_per_image,
                                               float* out, int64_t n, void* workspace, size_t workspace_bytes,
                                               const be_local_stage_opts* opts, void* stream) {
    BE_REQUIRE((view && view->base && view->wp > 0 && patches_per_image > 0) || n == 0,
               "be_local_stage_forward_view_f32: bad view");
    return forward_impl(packed, nullptr, view, patches_per_image, out, n, workspace, workspace_bytes, opts, stream,
                        "be_local_stage_forward_view_f32");
}
