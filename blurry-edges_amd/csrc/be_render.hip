// Blurred-wedge renderer, colours-only pass, one wavefront per 21x21 patch (gfx950, wave64).
//
// Replaces (reference file:line): params2dists utils/postprocessing_loss.py:43-86, params2etas :88-89,
// dists2indicators :91-95, the ridge regression (A^T A + lambda I)^-1 A^T y of blurry_edges_test.py:19-28 /
// local_training.py:37-41 (inverse_3by3 :104-112), the composite local_training.py:41 and the boundary
// map local_training.py:42-44.
//
// Mapping: 441 pixels over 64 lanes = 7 passes; the ten patch parameters are wave-uniform; the pixel
// data [3,21,21] is read once, coalesced (lane = consecutive pixel); the 6+9 normal-equation sums are
// reduced with wave shuffles in a fixed order (bitwise reproducible); the 3x3 system is solved in fp64
// by cofactors on every lane (60 flops); nothing but the requested outputs is written.
// HBM bytes per patch: 5292 (pixels) + 40 (params) in, 36 out (+5292 if the composite is requested).
#include "be_common.h"
#include "be_device_math.h"

namespace {

constexpr int NPIX = BE_NPIX;
constexpr int R = BE_R;
constexpr int PASSES = (NPIX + 63) / 64;   // 7
constexpr int WAVES_PER_BLOCK = 4;

struct RenderArgs {
    const float* params;    // [N,10]
    be_patch_view v;        // pixel data, gathered on read; patch n = (aperture n / P, grid position n % P)
    int64_t P;              // patches per aperture image
    float* colors;          // [N,3,3]
    float* recon;           // [N,3,21,21] or null
    float* boundary;        // [N,21,21] or null
    float* dists;           // [N,2,21,21] or null
    float* wedges;          // [N,3,21,21] or null
    float* gram;            // [N,3,3] or null
    float* aty;             // [N,3,3] or null
    int64_t n;
};

// signed distance to one ray of a wedge (utils/postprocessing_loss.py:26-30,50-76)
__device__ __forceinline__ float ray_distance(float px, float py, float vx, float vy, float s, float c, float w) {
#pragma clang fp contract(off)
    const float dx = px - vx, dy = py - vy;
    const float edge = (-s) * dx + c * dy;
    const float axial = c * dx + s * dy;
    if (axial < 0.0f) {
        const float aw = axial * w;
        const float r = sqrtf(edge * edge + aw * aw);
        return edge < 0.0f ? -r : r;
    }
    return edge;
}

__global__ __launch_bounds__(64 * WAVES_PER_BLOCK)
void k_render_colors(be_render_opts o, RenderArgs a) {
    __shared__ float lin[R];
    if (threadIdx.x < R) lin[threadIdx.x] = o.lin[threadIdx.x];
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int64_t patch = (int64_t)blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if (patch >= a.n) return;                       // whole wave exits together

    // ---- wave-uniform patch parameters
    const float* p = a.params + patch * 10;
    const float x0 = p[0], y0 = p[1], x1 = p[2], y1 = p[3];
    float t1 = p[4], f1 = p[5], t2 = p[6], f2 = p[7];
    if (o.wrap_angles) {                            // blurry_edges_test.py:124 / local_training.py:33
        t1 = be::remainder_2pi(t1); f1 = be::remainder_2pi(f1);
        t2 = be::remainder_2pi(t2); f2 = be::remainder_2pi(f2);
    }
    const float pi_f = 3.14159265358979323846f;
    const float sg1 = be::remainder_2pi(f1) < pi_f ? 1.0f : -1.0f;       // :46
    const float sg2 = be::remainder_2pi(f2) < pi_f ? 1.0f : -1.0f;       // :47
    float s11, c11, s12, c12, s21, c21, s22, c22;
    {
#pragma clang fp contract(off)
        const float t1p = t1 + f1, t2p = t2 + f2;                         // :49-50
        s11 = sinf(t1);  c11 = cosf(t1);  s12 = sinf(t1p); c12 = cosf(t1p);
        s21 = sinf(t2);  c21 = cosf(t2);  s22 = sinf(t2p); c22 = cosf(t2p);
    }
    const float eta1 = be::param2eta(p[8]), eta2 = be::param2eta(p[9]);
    const float root2 = 1.41421353816986083984375f;   // float32(sqrt(2)), :92
    float inv1, inv2;
    {
#pragma clang fp contract(off)
        inv1 = root2 * eta1;
        inv2 = root2 * eta2;
    }

    const int64_t pg = patch % a.P;
    const float* img = a.v.base + (patch / a.P) * a.v.s_aperture + (pg / a.v.wp) * a.v.s_pi + (pg % a.v.wp) * a.v.s_pj;
    float u0[PASSES], u1[PASSES], u2[PASSES];
    float g00 = 0, g01 = 0, g02 = 0, g11 = 0, g12 = 0, g22 = 0;
    float b0r = 0, b0g = 0, b0b = 0, b1r = 0, b1g = 0, b1b = 0, b2r = 0, b2g = 0, b2b = 0;

#pragma unroll
    for (int it = 0; it < PASSES; ++it) {
        const int pix = it * 64 + lane;
        const bool live = pix < NPIX;
        const int pc = live ? pix : 0;
        const int row = pc / R, col = pc - row * R;
        const float px = lin[col], py = lin[row];
        float d1, d2, h1, h2;
        {
#pragma clang fp contract(off)
            const float d11 = ray_distance(px, py, x0, y0, s11, c11, o.w);
            const float d12 = ray_distance(px, py, x0, y0, s12, c12, o.w);
            const float d21 = ray_distance(px, py, x1, y1, s21, c21, o.w);
            const float d22 = ray_distance(px, py, x1, y1, s22, c22, o.w);
            const float in1 = (sg1 * d11 > 0.0f && sg1 * d12 < 0.0f) ? sg1 : -sg1;     // strict :80
            const float in2 = (sg2 * d21 >= 0.0f && sg2 * d22 <= 0.0f) ? sg2 : -sg2;   // closed :81
            d1 = fminf(fabsf(d11), fabsf(d12)) * in1;
            d2 = fminf(fabsf(d21), fabsf(d22)) * in2;
            h1 = 0.5f * (1.0f + erff(d1 / inv1));                                       // :92
            h2 = 0.5f * (1.0f + erff(d2 / inv2));
            u0[it] = live ? (1.0f - h1) * (1.0f - h2) : 0.0f;                           // :93-95
            u1[it] = live ? h1 * (1.0f - h2) : 0.0f;
            u2[it] = live ? h2 : 0.0f;
        }
        const float* src = img + row * a.v.s_row + col * a.v.s_col;
        const float yr = live ? src[0] : 0.0f;
        const float yg = live ? src[a.v.s_chan] : 0.0f;
        const float yb = live ? src[2 * a.v.s_chan] : 0.0f;
        g00 = fmaf(u0[it], u0[it], g00); g01 = fmaf(u0[it], u1[it], g01); g02 = fmaf(u0[it], u2[it], g02);
        g11 = fmaf(u1[it], u1[it], g11); g12 = fmaf(u1[it], u2[it], g12); g22 = fmaf(u2[it], u2[it], g22);
        b0r = fmaf(u0[it], yr, b0r); b0g = fmaf(u0[it], yg, b0g); b0b = fmaf(u0[it], yb, b0b);
        b1r = fmaf(u1[it], yr, b1r); b1g = fmaf(u1[it], yg, b1g); b1b = fmaf(u1[it], yb, b1b);
        b2r = fmaf(u2[it], yr, b2r); b2g = fmaf(u2[it], yg, b2g); b2b = fmaf(u2[it], yb, b2b);
        if (live) {
            if (a.dists)  { a.dists[patch * 2 * NPIX + pix] = d1; a.dists[patch * 2 * NPIX + NPIX + pix] = d2; }
            if (a.wedges) { float* wq = a.wedges + patch * 3 * NPIX + pix;
                            wq[0] = u0[it]; wq[NPIX] = u1[it]; wq[2 * NPIX] = u2[it]; }
            if (a.boundary) {                                   // local_training.py:42-44
#pragma clang fp contract(off)
                const float a1 = fabsf(d1), a2 = fabsf(d2);
                const float db = d2 >= 0.0f ? d2 : (a1 < a2 ? a1 : a2);
                a.boundary[patch * NPIX + pix] = expf(-(db * db) / o.delta_sq);
            }
        }
    }

    // ---- normal equations: wave reduction (fixed order), ridge, fp64 cofactor solve
    g00 = be::wave_sum(g00); g01 = be::wave_sum(g01); g02 = be::wave_sum(g02);
    g11 = be::wave_sum(g11); g12 = be::wave_sum(g12); g22 = be::wave_sum(g22);
    b0r = be::wave_sum(b0r); b0g = be::wave_sum(b0g); b0b = be::wave_sum(b0b);
    b1r = be::wave_sum(b1r); b1g = be::wave_sum(b1g); b1b = be::wave_sum(b1b);
    b2r = be::wave_sum(b2r); b2g = be::wave_sum(b2g); b2b = be::wave_sum(b2b);
    g00 += o.lambda_ridge; g11 += o.lambda_ridge; g22 += o.lambda_ridge;

    const double A00 = g00, A01 = g01, A02 = g02, A11 = g11, A12 = g12, A22 = g22;
    const double C00 = A11 * A22 - A12 * A12, C01 = A02 * A12 - A01 * A22, C02 = A01 * A12 - A02 * A11;
    const double C11 = A00 * A22 - A02 * A02, C12 = A01 * A02 - A00 * A12, C22 = A00 * A11 - A01 * A01;
    const double idet = 1.0 / (A00 * C00 + A01 * C01 + A02 * C02);
    // colour of wedge k, channel c:  col[c][k] = sum_j inv[k][j] * b[j][c]
    float cr0, cr1, cr2, cg0, cg1, cg2, cb0, cb1, cb2;
    cr0 = (float)((C00 * b0r + C01 * b1r + C02 * b2r) * idet);
    cr1 = (float)((C01 * b0r + C11 * b1r + C12 * b2r) * idet);
    cr2 = (float)((C02 * b0r + C12 * b1r + C22 * b2r) * idet);
    cg0 = (float)((C00 * b0g + C01 * b1g + C02 * b2g) * idet);
    cg1 = (float)((C01 * b0g + C11 * b1g + C12 * b2g) * idet);
    cg2 = (float)((C02 * b0g + C12 * b1g + C22 * b2g) * idet);
    cb0 = (float)((C00 * b0b + C01 * b1b + C02 * b2b) * idet);
    cb1 = (float)((C01 * b0b + C11 * b1b + C12 * b2b) * idet);
    cb2 = (float)((C02 * b0b + C12 * b1b + C22 * b2b) * idet);

    if (lane == 0) {
        float* c = a.colors + patch * 9;           // [rgb][wedge]
        c[0] = cr0; c[1] = cr1; c[2] = cr2; c[3] = cg0; c[4] = cg1; c[5] = cg2; c[6] = cb0; c[7] = cb1; c[8] = cb2;
        if (a.gram) { float* g = a.gram + patch * 9;
                      g[0] = g00; g[1] = g01; g[2] = g02; g[3] = g01; g[4] = g11; g[5] = g12; g[6] = g02; g[7] = g12; g[8] = g22; }
        if (a.aty)  { float* b = a.aty + patch * 9;  // [wedge][rgb]
                      b[0] = b0r; b[1] = b0g; b[2] = b0b; b[3] = b1r; b[4] = b1g; b[5] = b1b; b[6] = b2r; b[7] = b2g; b[8] = b2b; }
    }
    if (a.recon) {                                  // local_training.py:41
        float* out = a.recon + patch * 3 * NPIX;
#pragma unroll
        for (int it = 0; it < PASSES; ++it) {
            const int pix = it * 64 + lane;
            if (pix < NPIX) {
#pragma clang fp contract(off)
                out[pix]            = u0[it] * cr0 + u1[it] * cr1 + u2[it] * cr2;
                out[NPIX + pix]     = u0[it] * cg0 + u1[it] * cg1 + u2[it] * cg2;
                out[2 * NPIX + pix] = u0[it] * cb0 + u1[it] * cb1 + u2[it] * cb2;
            }
        }
    }
}

}  // namespace

static int render_colors_impl(const be_render_opts* o, const float* params10, const be_patch_view& v, int64_t P,
                              float* colors, float* recon, float* boundary, float* dists, float* wedges, float* gram,
                              float* aty, int64_t n, void* stream, const char* who) {
    BE_REQUIRE(n >= 0, "%s: n < 0", who);
    if (n == 0) return BE_OK;
    BE_REQUIRE(o && params10 && v.base && colors, "%s: null pointer", who);
    BE_REQUIRE(o->lambda_ridge >= 0.0f && o->delta_sq > 0.0f, "%s: bad options", who);
    BE_REQUIRE(P > 0 && v.wp > 0, "%s: bad patch grid", who);
    RenderArgs a{params10, v, P, colors, recon, boundary, dists, wedges, gram, aty, n};
    const int64_t blocks = (n + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK;
    BE_REQUIRE(blocks <= 0x7fffffff, "%s: n too large", who);
    {
        // algorithmic bytes of pass A: 441 x 3 pixels + 10 parameters in, 9 colours out per patch (the optional outputs are test hooks)
        be::ProfileScope prof(be::as_stream(stream), BE_KERNEL_RENDER_COLORS, 0.0, 4.0 * n * (1323.0 + 10.0 + 9.0), 0.0);
        hipLaunchKernelGGL(k_render_colors, dim3((unsigned)blocks), dim3(64 * WAVES_PER_BLOCK), 0, be::as_stream(stream),
                           *o, a);
    }
    return be::check_launch(who);
}

extern "C" int be_render_colors_f32(const be_render_opts* o, const float* params10, const float* patches,
                                    float* colors, float* recon, float* boundary, float* dists, float* wedges,
                                    float* gram, float* aty, int64_t n, void* stream) {
    BE_REQUIRE(n < ((int64_t)1 << 31), "be_render_colors_f32: n too large");
    // flat patches [N,3,21,21] as a one-row grid of N positions
    be_patch_view v{patches, 0, NPIX, R, 1, 0, 3 * NPIX, (int)(n > 0 ? n : 1)};
    return render_colors_impl(o, params10, v, n > 0 ? n : 1, colors, recon, boundary, dists, wedges, gram, aty, n, stream,
                              "be_render_colors_f32");
}

extern "C" int be_render_colors_view_f32(const be_render_opts* o, const float* params10, const be_patch_view* view,
                                         int64_t patches_per_image, float* colors, float* recon, float* boundary,
                                         float* dists, float* wedges, float* gram, float* aty, int64_t n, void* stream) {
    BE_REQUIRE(view, "be_render_colors_view_f32: null view");
    return render_colors_impl(o, params10, *view, patches_per_image, colors, recon, boundary, dists, wedges, gram, aty, n,
                              stream, "be_render_colors_view_f32");
}
