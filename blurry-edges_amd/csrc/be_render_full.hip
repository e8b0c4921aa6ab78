// Full render pass ("pass B") + owner-computes fold, gfx950.
//
// Replaces PostProcess.get_patches(colors_only=False) and the six nn.Fold aggregations of
// blurry_edges_test.py:36-79,93-99 / utils/postprocessing_loss.py:151-173 (and the per-block patch-tensor
// stitching of blurry_edges_test_big.py:166-189) with two kernels and NO per-patch tensors in HBM:
//
//   k_render_records   one wavefront per patch position: both aperture images are read once (gather-on-read
//                      through a strided patch view: an image pair, an unfolded tensor or flat patches),
//                      the 882-row ridge system is reduced with wave shuffles, solved in fp64, the two
//                      wedge depths come from etas2depth, the refocus radii from depth2sigma and a wave
//                      ballot of the depth mask; result = one 128-byte record per patch.
//   k_fold_records     one thread per output pixel ("owner computes"): walks the <= 11x11 patches that
//                      cover the pixel in a fixed order, re-evaluates the wedges from the record for the four
//                      blur sets (aperture 1, aperture 2, sharpened, refocused) and accumulates the six maps.
//                      Deterministic (no atomics), divides by the analytic overlap count.
//
// The reference materialises 26.5 KB per patch pair and folds it six times; here the per-patch state is
// 128 B and every output byte is written once.
#include "be_common.h"
#include "be_wedge.h"

namespace {

constexpr int NPIX = BE_NPIX;
constexpr int R = BE_R;
constexpr int PASSES = (NPIX + 63) / 64;
constexpr int WAVES_PER_BLOCK = 4;
constexpr int REC = BE_RECORD_FLOATS;   // 32

// record layout (floats)
enum { R_GEOM = 0 /* x0,y0,x1,y1,s11,c11,s12,c12,s21,c21,s22,c22,sg1,sg2 */, R_RAD1 = 14, R_RAD2 = 16, R_RADF = 18,
       R_COL = 20, R_DEPTH = 29, R_FLAGS = 31 };

struct FullArgs {
    const float* params12;   // [P,12]
    be_patch_view v;
    float* records;          // [P,32]
    float* patches;          // [P,2,3,441] or null
    float* shpd;             // [P,3,441] or null
    float* refoc;            // [P,3,441] or null
    float* boundary;         // [P,441] or null
    float* depth_map;        // [P,441] or null
    int32_t* depth_mask;     // [P,441] or null
    float rho_prime;
    int densify_w;
    int64_t n;
};

__device__ __forceinline__ void composite3(const float* col, float u0, float u1, float u2, float* out, int pix) {
#pragma clang fp contract(off)
    out[pix]            = u0 * col[0] + u1 * col[1] + u2 * col[2];
    out[NPIX + pix]     = u0 * col[3] + u1 * col[4] + u2 * col[5];
    out[2 * NPIX + pix] = u0 * col[6] + u1 * col[7] + u2 * col[8];
}

__global__ __launch_bounds__(64 * WAVES_PER_BLOCK)
void k_render_records(be_render_opts o, be_depth_consts dc, FullArgs a) {
    __shared__ float lin[R];
    if (threadIdx.x < R) lin[threadIdx.x] = o.lin[threadIdx.x];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int64_t patch = (int64_t)blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if (patch >= a.n) return;

    const float* p = a.params12 + patch * 12;
    const be::WedgeGeom g = be::make_geom(p, o.wrap_angles != 0);
    float eta[4], rad[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma clang fp contract(off)
        eta[k] = be::param2eta(p[8 + k]);              // (w1,img1) (w2,img1) (w1,img2) (w2,img2)
        rad[k] = be::kRoot2 * eta[k];
    }
    const int pi = (int)(patch / a.v.wp), pj = (int)(patch % a.v.wp);
    const float* img1 = a.v.base + pi * a.v.s_pi + pj * a.v.s_pj;
    const float* img2 = img1 + a.v.s_aperture;

    float d1s[PASSES], d2s[PASSES];
    float gs[6] = {0, 0, 0, 0, 0, 0};
    float bs[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    bool any1 = false, any2 = false;
#pragma unroll
    for (int it = 0; it < PASSES; ++it) {
        const int pix = it * 64 + lane;
        const bool live = pix < NPIX;
        const int pc = live ? pix : 0;
        const int row = pc / R, col = pc - row * R;
        float d1, d2;
        be::wedge_dists(g, lin[col], lin[row], o.w, d1, d2);
        d1s[it] = d1; d2s[it] = d2;
        const int64_t off = row * a.v.s_row + col * a.v.s_col;
#pragma unroll
        for (int im = 0; im < 2; ++im) {
            float u0, u1, u2;
            be::indicators(d1, d2, rad[2 * im], rad[2 * im + 1], u0, u1, u2);
            if (!live) { u0 = 0.f; u1 = 0.f; u2 = 0.f; }
            const float* src = (im ? img2 : img1) + off;
            const float yr = live ? src[0] : 0.f, yg = live ? src[a.v.s_chan] : 0.f, yb = live ? src[2 * a.v.s_chan] : 0.f;
            gs[0] = fmaf(u0, u0, gs[0]); gs[1] = fmaf(u0, u1, gs[1]); gs[2] = fmaf(u0, u2, gs[2]);
            gs[3] = fmaf(u1, u1, gs[3]); gs[4] = fmaf(u1, u2, gs[4]); gs[5] = fmaf(u2, u2, gs[5]);
            bs[0] = fmaf(u0, yr, bs[0]); bs[1] = fmaf(u0, yg, bs[1]); bs[2] = fmaf(u0, yb, bs[2]);
            bs[3] = fmaf(u1, yr, bs[3]); bs[4] = fmaf(u1, yg, bs[4]); bs[5] = fmaf(u1, yb, bs[5]);
            bs[6] = fmaf(u2, yr, bs[6]); bs[7] = fmaf(u2, yg, bs[7]); bs[8] = fmaf(u2, yb, bs[8]);
        }
        if (live) {
            const int m = be::depth_mask(d1, d2, o.delta_sq, a.densify_w != 0);
            any1 |= (m == 1); any2 |= (m == 2);
        }
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) gs[k] = be::wave_sum(gs[k]);
#pragma unroll
    for (int k = 0; k < 9; ++k) bs[k] = be::wave_sum(bs[k]);
    const be::Colors9 col = be::solve_colors(gs[0] + o.lambda_ridge, gs[1], gs[2], gs[3] + o.lambda_ridge, gs[4],
                                             gs[5] + o.lambda_ridge, bs);
    int br;
    const float z1 = be::etas2depth(dc, eta[0], eta[2], br);      // blurry_edges_test.py:44
    const float z2 = be::etas2depth(dc, eta[1], eta[3], br);      // :45
    const bool has1 = __ballot(any1) != 0ull, has2 = __ballot(any2) != 0ull;
    float radf[2];
    {
#pragma clang fp contract(off)
        const float s1 = has1 ? be::depth2sigma(dc, z1, a.rho_prime) : 1e-4f;     // :66-71
        const float s2 = has2 ? be::depth2sigma(dc, z2, a.rho_prime) : 1e-4f;
        radf[0] = be::kRoot2 * s1; radf[1] = be::kRoot2 * s2;
    }
    if (lane == 0) {
        float* r = a.records + patch * REC;
        r[0] = g.x0; r[1] = g.y0; r[2] = g.x1; r[3] = g.y1;
        r[4] = g.s11; r[5] = g.c11; r[6] = g.s12; r[7] = g.c12; r[8] = g.s21; r[9] = g.c21; r[10] = g.s22; r[11] = g.c22;
        r[12] = g.sg1; r[13] = g.sg2;
        r[R_RAD1] = rad[0]; r[R_RAD1 + 1] = rad[1]; r[R_RAD2] = rad[2]; r[R_RAD2 + 1] = rad[3];
        r[R_RADF] = radf[0]; r[R_RADF + 1] = radf[1];
#pragma unroll
        for (int k = 0; k < 9; ++k) r[R_COL + k] = col.c[k];
        r[R_DEPTH] = z1; r[R_DEPTH + 1] = z2;
        r[R_FLAGS] = (float)((has1 ? 1 : 0) | (has2 ? 2 : 0));
    }
    // ---- optional materialised per-patch outputs (parity tests / callers that want the reference's tensors)
    const bool want = a.patches || a.shpd || a.refoc || a.boundary || a.depth_map || a.depth_mask;
    if (!want) return;
    const float rs = be::kRoot2 * 1e-4f;                              // sharpened: eta = 1e-4 for both wedges (:63)
#pragma unroll
    for (int it = 0; it < PASSES; ++it) {
        const int pix = it * 64 + lane;
        if (pix >= NPIX) continue;
        const float d1 = d1s[it], d2 = d2s[it];
        float u0, u1, u2;
        if (a.patches) {
            be::indicators(d1, d2, rad[0], rad[1], u0, u1, u2);
            composite3(col.c, u0, u1, u2, a.patches + patch * 6 * NPIX, pix);
            be::indicators(d1, d2, rad[2], rad[3], u0, u1, u2);
            composite3(col.c, u0, u1, u2, a.patches + patch * 6 * NPIX + 3 * NPIX, pix);
        }
        if (a.shpd) { be::indicators(d1, d2, rs, rs, u0, u1, u2); composite3(col.c, u0, u1, u2, a.shpd + patch * 3 * NPIX, pix); }
        if (a.refoc) { be::indicators(d1, d2, radf[0], radf[1], u0, u1, u2);
                       composite3(col.c, u0, u1, u2, a.refoc + patch * 3 * NPIX, pix); }
        if (a.boundary) a.boundary[patch * NPIX + pix] = be::boundary_value(d1, d2, o.delta_sq);
        const int m = be::depth_mask(d1, d2, o.delta_sq, a.densify_w != 0);
        if (a.depth_mask) a.depth_mask[patch * NPIX + pix] = m;
        if (a.depth_map) a.depth_map[patch * NPIX + pix] = m == 1 ? z1 : (m == 2 ? z2 : 0.0f);
    }
}

// ------------------------------------------------------------------------------------------------ fold
struct FoldArgs {
    const float* records;    // [Hp*Wp,32]
    int hp, wp, H, W, stride;
    int densify_w;
    float* image;            // [2,3,H,W] or null
    float* shpd;             // [3,H,W] or null
    float* refoc;            // [3,H,W] or null
    float* bndry;            // [H,W] or null
    float* depth;            // [H,W] or null
    float* conf;             // [H,W] or null
    int64_t rec_stride;      // batched form (grid.z = image): floats between the records / maps of consecutive images
};

__global__ __launch_bounds__(256)
void k_fold_records(be_render_opts o, FoldArgs a) {
    __shared__ float lin[R];
    if (threadIdx.x < R) lin[threadIdx.x] = o.lin[threadIdx.x];
    __syncthreads();
    const int x = blockIdx.x * 16 + (threadIdx.x & 15);
    const int y = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (x >= a.W || y >= a.H) return;
    if (blockIdx.z) {                                           // image blockIdx.z of a batch
        const size_t b = blockIdx.z, hw = (size_t)a.H * a.W;
        a.records += b * a.rec_stride;
        if (a.image) a.image += b * 6 * hw;
        if (a.shpd) a.shpd += b * 3 * hw;
        if (a.refoc) a.refoc += b * 3 * hw;
        if (a.bndry) a.bndry += b * hw;
        if (a.depth) a.depth += b * hw;
        if (a.conf) a.conf += b * hw;
    }
    // patches covering (y,x): stride*i <= y <= stride*i + 20
    const int s = a.stride;
    int i_lo = (y - (R - 1) + s - 1) / s; if (y - (R - 1) < 0) i_lo = 0;
    int j_lo = (x - (R - 1) + s - 1) / s; if (x - (R - 1) < 0) j_lo = 0;
    int i_hi = y / s; if (i_hi > a.hp - 1) i_hi = a.hp - 1;
    int j_hi = x / s; if (j_hi > a.wp - 1) j_hi = a.wp - 1;
    float acc1[3] = {0, 0, 0}, acc2[3] = {0, 0, 0}, accs[3] = {0, 0, 0}, accf[3] = {0, 0, 0};
    float accb = 0.f, accz = 0.f;
    int cnt = 0, cntz = 0;
    const float rs = be::kRoot2 * 1e-4f;
    for (int i = i_lo; i <= i_hi; ++i) {
        const float py = lin[y - s * i];
        for (int j = j_lo; j <= j_hi; ++j) {
            const float px = lin[x - s * j];
            const float4* rp = reinterpret_cast<const float4*>(a.records + (size_t)(i * a.wp + j) * REC);
            float r[REC];
#pragma unroll
            for (int k = 0; k < REC / 4; ++k) { const float4 t = rp[k]; r[4 * k] = t.x; r[4 * k + 1] = t.y; r[4 * k + 2] = t.z; r[4 * k + 3] = t.w; }
            be::WedgeGeom g;
            g.x0 = r[0]; g.y0 = r[1]; g.x1 = r[2]; g.y1 = r[3];
            g.s11 = r[4]; g.c11 = r[5]; g.s12 = r[6]; g.c12 = r[7]; g.s21 = r[8]; g.c21 = r[9]; g.s22 = r[10]; g.c22 = r[11];
            g.sg1 = r[12]; g.sg2 = r[13];
            float d1, d2;
            be::wedge_dists(g, px, py, o.w, d1, d2);
            const float* col = r + R_COL;
            float u0, u1, u2;
            {
#pragma clang fp contract(off)
                if (a.image) {
                    be::indicators(d1, d2, r[R_RAD1], r[R_RAD1 + 1], u0, u1, u2);
#pragma unroll
                    for (int c = 0; c < 3; ++c) acc1[c] += u0 * col[3 * c] + u1 * col[3 * c + 1] + u2 * col[3 * c + 2];
                    be::indicators(d1, d2, r[R_RAD2], r[R_RAD2 + 1], u0, u1, u2);
#pragma unroll
                    for (int c = 0; c < 3; ++c) acc2[c] += u0 * col[3 * c] + u1 * col[3 * c + 1] + u2 * col[3 * c + 2];
                }
                if (a.shpd) {
                    be::indicators(d1, d2, rs, rs, u0, u1, u2);
#pragma unroll
                    for (int c = 0; c < 3; ++c) accs[c] += u0 * col[3 * c] + u1 * col[3 * c + 1] + u2 * col[3 * c + 2];
                }
                if (a.refoc) {
                    be::indicators(d1, d2, r[R_RADF], r[R_RADF + 1], u0, u1, u2);
#pragma unroll
                    for (int c = 0; c < 3; ++c) accf[c] += u0 * col[3 * c] + u1 * col[3 * c + 1] + u2 * col[3 * c + 2];
                }
                if (a.bndry) accb += be::boundary_value(d1, d2, o.delta_sq);
                if (a.depth || a.conf) {
                    const int m = be::depth_mask(d1, d2, o.delta_sq, a.densify_w != 0);
                    if (m == 1) { accz += r[R_DEPTH]; ++cntz; }
                    else if (m == 2) { accz += r[R_DEPTH + 1]; ++cntz; }
                }
            }
            ++cnt;
        }
    }
    const size_t hw = (size_t)a.H * a.W, at = (size_t)y * a.W + x;
    const float n = (float)cnt;                         // = nn.Fold(ones) at this pixel (>= 1)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        if (a.image) { a.image[c * hw + at] = acc1[c] / n; a.image[(3 + c) * hw + at] = acc2[c] / n; }
        if (a.shpd) a.shpd[c * hw + at] = accs[c] / n;
        if (a.refoc) a.refoc[c * hw + at] = accf[c] / n;
    }
    if (a.bndry) a.bndry[at] = accb / n;
    if (a.depth) a.depth[at] = accz / (cntz > 0 ? (float)cntz : 1.0f);      // postprocessing_loss.py:170-172
    if (a.conf) a.conf[at] = (float)cntz / n;
}

// ------------------------------------------------------------------------------------------------ glue
__global__ void k_unfold(const float* __restrict__ img, float* __restrict__ out, int B, int C, int H, int W, int hp,
                         int wp, int stride) {
    const int64_t total = (int64_t)B * hp * wp * C * NPIX;
    const int64_t gs = (int64_t)gridDim.x * blockDim.x;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gs) {
        int64_t t = idx;
        const int pix = (int)(t % NPIX); t /= NPIX;
        const int c = (int)(t % C); t /= C;
        const int j = (int)(t % wp); t /= wp;
        const int i = (int)(t % hp);
        const int64_t b = t / hp;
        const int r = pix / R, cc = pix - r * R;
        out[idx] = img[((b * C + c) * H + (stride * i + r)) * W + stride * j + cc];
    }
}

// blurry_edges_test.py:123-132: [2,P,10] CNN outputs + [2,P,9] colours -> normalised 38-feature rows
__global__ void k_local_features(const float* __restrict__ params10, const float* __restrict__ colors, float* __restrict__ pm,
                                 int64_t P) {
#pragma clang fp contract(off)
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= P * 38) return;
    const int64_t p = idx / 38;
    const int f = (int)(idx % 38);
    const int im = f / 19, k = f % 19;
    float v;
    if (k < 10) {
        const float q = params10[(im * P + p) * 10 + k];
        if (k < 4) v = q / 3.0f;
        else if (k < 8) v = (be::remainder_2pi(q) - be::kPi) / be::kPi;
        else v = q - 0.5f;
    } else {
        v = (colors[(im * P + p) * 9 + (k - 10)] - 0.5f) * 2.0f;
    }
    pm[idx] = v;
}

// blurry_edges_test.py:134-138: transformer output [P,12] -> wedge parameters
__global__ void k_global_denorm(const float* __restrict__ y, float* __restrict__ est, int64_t P) {
#pragma clang fp contract(off)
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= P * 12) return;
    const int k = (int)(idx % 12);
    const float v = y[idx];
    est[idx] = k < 4 ? v * 3.0f : (k < 8 ? be::remainder_2pi((v + 1.0f) * be::kPi) : v + 0.5f);
}

}  // namespace

extern "C" int be_render_full_f32(const be_render_opts* o, const be_depth_consts* dc, float rho_prime, int densify_w,
                                  const float* params12, const be_patch_view* view, float* records, float* patches,
                                  float* shpd, float* refoc, float* boundary, float* depth_map, int32_t* depth_mask,
                                  int64_t n, void* stream) {
    BE_REQUIRE(n >= 0, "be_render_full_f32: n < 0");
    if (n == 0) return BE_OK;
    BE_REQUIRE(o && dc && params12 && view && view->base && records, "be_render_full_f32: null pointer");
    BE_REQUIRE(view->wp > 0, "be_render_full_f32: view.wp must be > 0");
    BE_REQUIRE(be::aligned16(records), "be_render_full_f32: records must be 16-byte aligned");
    FullArgs a{params12, *view, records, patches, shpd, refoc, boundary, depth_map, depth_mask, rho_prime, densify_w, n};
    const int64_t blocks = (n + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK;
    BE_REQUIRE(blocks <= 0x7fffffff, "be_render_full_f32: n too large");
    hipLaunchKernelGGL(k_render_records, dim3((unsigned)blocks), dim3(64 * WAVES_PER_BLOCK), 0, be::as_stream(stream),
                       *o, *dc, a);
    return be::check_launch("be_render_full_f32");
}

extern "C" int be_fold_records_f32(const be_render_opts* o, const float* records, int hp, int wp, int H, int W,
                                   int stride, int densify_w, float* image, float* shpd, float* refoc, float* bndry,
                                   float* depth, float* conf, void* stream) {
    BE_REQUIRE(o && records, "be_fold_records_f32: null pointer");
    BE_REQUIRE(hp > 0 && wp > 0 && H > 0 && W > 0 && stride > 0, "be_fold_records_f32: bad sizes");
    BE_REQUIRE(stride * (hp - 1) + R <= H && stride * (wp - 1) + R <= W, "be_fold_records_f32: patch grid exceeds the image");
    BE_REQUIRE(be::aligned16(records), "be_fold_records_f32: records must be 16-byte aligned");
    FoldArgs a{records, hp, wp, H, W, stride, densify_w, image, shpd, refoc, bndry, depth, conf, 0};
    hipLaunchKernelGGL(k_fold_records, dim3((W + 15) / 16, (H + 15) / 16), dim3(256), 0, be::as_stream(stream), *o, a);
    return be::check_launch("be_fold_records_f32");
}

extern "C" int be_fold_records_batch_f32(const be_render_opts* o, const float* records, int B, int hp, int wp, int H, int W,
                                         int stride, int densify_w, float* image, float* shpd, float* refoc, float* bndry,
                                         float* depth, float* conf, void* stream) {
    BE_REQUIRE(o && records, "be_fold_records_batch_f32: null pointer");
    BE_REQUIRE(B > 0 && B <= 65535 && hp > 0 && wp > 0 && H > 0 && W > 0 && stride > 0, "be_fold_records_batch_f32: bad sizes");
    BE_REQUIRE(stride * (hp - 1) + R <= H && stride * (wp - 1) + R <= W, "be_fold_records_batch_f32: patch grid exceeds the image");
    BE_REQUIRE(be::aligned16(records), "be_fold_records_batch_f32: records must be 16-byte aligned");
    FoldArgs a{records, hp, wp, H, W, stride, densify_w, image, shpd, refoc, bndry, depth, conf, (int64_t)hp * wp * REC};
    hipLaunchKernelGGL(k_fold_records, dim3((W + 15) / 16, (H + 15) / 16, B), dim3(256), 0, be::as_stream(stream), *o, a);
    return be::check_launch("be_fold_records_batch_f32");
}

extern "C" int be_unfold_patches_f32(const float* img, float* out, int B, int C, int H, int W, int stride, void* stream) {
    BE_REQUIRE(img && out, "be_unfold_patches_f32: null pointer");
    BE_REQUIRE(B > 0 && C > 0 && H >= R && W >= R && stride > 0, "be_unfold_patches_f32: bad sizes");
    const int hp = (H - R) / stride + 1, wp = (W - R) / stride + 1;
    const int64_t total = (int64_t)B * hp * wp * C * NPIX;
    int64_t g = (total + 255) / 256; if (g > 8192) g = 8192;
    hipLaunchKernelGGL(k_unfold, dim3((unsigned)g), dim3(256), 0, be::as_stream(stream), img, out, B, C, H, W, hp, wp, stride);
    return be::check_launch("be_unfold_patches_f32");
}

extern "C" int be_local_features_f32(const float* params10, const float* colors, float* pm, int64_t P, void* stream) {
    BE_REQUIRE(P >= 0, "be_local_features_f32: P < 0");
    if (P == 0) return BE_OK;
    BE_REQUIRE(params10 && colors && pm, "be_local_features_f32: null pointer");
    hipLaunchKernelGGL(k_local_features, dim3((unsigned)((P * 38 + 255) / 256)), dim3(256), 0, be::as_stream(stream),
                       params10, colors, pm, P);
    return be::check_launch("be_local_features_f32");
}

extern "C" int be_global_denorm_f32(const float* y, float* est, int64_t P, void* stream) {
    BE_REQUIRE(P >= 0, "be_global_denorm_f32: P < 0");
    if (P == 0) return BE_OK;
    BE_REQUIRE(y && est, "be_global_denorm_f32: null pointer");
    hipLaunchKernelGGL(k_global_denorm, dim3((unsigned)((P * 12 + 255) / 256)), dim3(256), 0, be::as_stream(stream), y, est, P);
    return be::check_launch("be_global_denorm_f32");
}
