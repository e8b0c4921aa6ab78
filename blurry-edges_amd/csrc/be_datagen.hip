// Synthetic-shape training data on the GPU: per-pixel work of train_val_data_generator.py:31-275.
// All arithmetic the reference does in numpy float64 is float64 here; rasterisation follows OpenCV's integer algorithms.
// HBM-bound streaming kernels over [N,H,W] images, one thread per pixel (or per element); nothing here is GEMM-shaped.
//
//   k_raster           per object: the FILL and RING bit planes under OpenCV's scan-conversion rules (cv2.circle / cv2.drawContours, :58-76)
//   k_scene            objects far -> near: all-in-focus colour, boundary locations, image depth, boundary depth
//                      (3x3 dilations of the two planes, :77-85,100-103)
//   k_mask / k_blur_h / k_blur_v_composite   per object: binary mask, separable Gaussian PSF of its depth per aperture
//                      (scipy.ndimage.convolve(mode='reflect') of the (2k+1)^2 kernel, k = ceil(3 sigma), :87-94),
//                      alpha-composite onto the two aperture images
//   k_round / k_row_dist / k_col_dist / k_sobel   imgs.round(), city-block distance to the nearest boundary pixel
//                      (= the breadth-first dilation of :105-116), Sobel magnitude with reflected borders (:118-123)
//   k_noise            Poisson(img/255 alpha) + sigma N(0,1), clip, round (:165-182); counter-based splitmix64 streams
//   k_candidates / k_crop   patch centres near boundaries and the 21x21 crops with their in-patch distance (:214-252)
#include "be_common.h"

namespace {

constexpr int SHAPE_INTS = 10;     // kind, nv, x0,y0 .. x3,y3   (circle: x0,y0 = centre, x1 = radius)
constexpr int PROP_F64 = 4;        // z, c0, c1, c2
constexpr int MAXO_LDS = 32;

// ---- rasterisation: OpenCV's scan-conversion rules (round 5) --------------------------------------------------------------
// The reference draws every object with cv2.circle / cv2.drawContours (train_val_data_generator.py:58-76; default LINE_8, shift 0).
// k_raster follows the algorithms those calls run in OpenCV 4.x modules/imgproc/src/drawing.cpp - Circle() (midpoint walk, filled
// rows / eight symmetric points), Line() = clipLine() + the 8-connected LineIterator started from the left end point,
// CollectPolyEdges() + FillEdgeCollection() (16.16 fixed-point edges, active for y0 <= y < y1, runs from ceil(x_left) to
// floor(x_right)) - and leaves two bit planes per object, FILL and RING (the thickness-1 outline), [N][maxo][2][H][ceil(W/32)]
// words; every later kernel reads bits.  One workgroup per (object, image).  oracle/datagen.py restates the same rules in numpy.
constexpr int XY_SHIFT = 16;
__host__ __device__ __forceinline__ int raster_row_words(int W) { return (W + 31) >> 5; }

__device__ __forceinline__ bool plane_bit(const uint32_t* __restrict__ plane, int rw, int x, int y) {
    return (plane[(size_t)y * rw + (x >> 5)] >> (x & 31)) & 1u;
}
__device__ __forceinline__ void set_px(uint32_t* plane, int rw, int H, int W, int x, int y) {
    if ((unsigned)x < (unsigned)W && (unsigned)y < (unsigned)H) atomicOr(&plane[(size_t)y * rw + (x >> 5)], 1u << (x & 31));
}
// pixels x1..x2 (inclusive) of row y, clipped to the image
__device__ __forceinline__ void set_run(uint32_t* plane, int rw, int H, int W, int y, int x1, int x2) {
    if ((unsigned)y >= (unsigned)H) return;
    x1 = max(x1, 0); x2 = min(x2, W - 1);
    if (x1 > x2) return;
    for (int w = x1 >> 5; w <= (x2 >> 5); ++w) {
        const int lo = max(x1 - 32 * w, 0), hi = min(x2 - 32 * w, 31);
        const uint32_t m = (hi == 31 ? 0xffffffffu : ((1u << (hi + 1)) - 1u)) & ~((1u << lo) - 1u);
        atomicOr(&plane[(size_t)y * rw + w], m);
    }
}

// clipLine(Size(W, H), p1, p2): false when nothing of the segment is inside
__device__ __forceinline__ bool clip_line(int W, int H, long long& x1, long long& y1, long long& x2, long long& y2) {
    const long long right = W - 1, bottom = H - 1;
    int c1 = (x1 < 0) + (x1 > right) * 2 + (y1 < 0) * 4 + (y1 > bottom) * 8;
    int c2 = (x2 < 0) + (x2 > right) * 2 + (y2 < 0) * 4 + (y2 > bottom) * 8;
    if ((c1 & c2) == 0 && (c1 | c2) != 0) {
        long long a;
        if (c1 & 12) {
            a = c1 < 8 ? 0 : bottom;
            x1 += (long long)((double)(a - y1) * (double)(x2 - x1) / (double)(y2 - y1));
            y1 = a;
            c1 = (x1 < 0) + (x1 > right) * 2;
        }
        if (c2 & 12) {
            a = c2 < 8 ? 0 : bottom;
            x2 += (long long)((double)(a - y2) * (double)(x2 - x1) / (double)(y2 - y1));
            y2 = a;
            c2 = (x2 < 0) + (x2 > right) * 2;
        }
        if ((c1 & c2) == 0 && (c1 | c2) != 0) {
            if (c1) {
                a = c1 == 1 ? 0 : right;
                y1 += (long long)((double)(a - x1) * (double)(y2 - y1) / (double)(x2 - x1));
                x1 = a;
                c1 = 0;
            }
            if (c2) {
                a = c2 == 1 ? 0 : right;
                y2 += (long long)((double)(a - x2) * (double)(y2 - y1) / (double)(x2 - x1));
                x2 = a;
                c2 = 0;
            }
        }
    }
    return (c1 | c2) == 0;
}

// Line(img, p1, p2, color, 8): LineIterator(leftToRight = true) on the clipped segment
__device__ void draw_line(uint32_t* plane, int rw, int H, int W, int ax, int ay, int bx, int by) {
    long long x1 = ax, y1 = ay, x2 = bx, y2 = by;
    if ((unsigned long long)x1 >= (unsigned long long)W || (unsigned long long)x2 >= (unsigned long long)W ||
        (unsigned long long)y1 >= (unsigned long long)H || (unsigned long long)y2 >= (unsigned long long)H) {
        if (!clip_line(W, H, x1, y1, x2, y2)) return;
    }
    int dx = (int)(x2 - x1), dy = (int)(y2 - y1), sy = 1;
    if (dx < 0) { dx = -dx; dy = -dy; x1 = x2; y1 = y2; }
    if (dy < 0) { dy = -dy; sy = -1; }
    const bool vert = dy > dx;
    if (vert) { const int t = dx; dx = dy; dy = t; }
    int err = dx - 2 * dy;
    const int plus = 2 * dx, minus = -2 * dy;
    int x = (int)x1, y = (int)y1;
    for (int i = 0; i <= dx; ++i) {
        set_px(plane, rw, H, W, x, y);
        const bool m = err < 0;
        err += minus + (m ? plus : 0);
        if (vert) { y += sy; x += m ? 1 : 0; }
        else { x += 1; y += m ? sy : 0; }
    }
}

constexpr int RASTER_MAX_PAIRS = 2048;     // octant steps of the midpoint walk kept in LDS: radius < ~2800

__global__ __launch_bounds__(256)
void k_raster(const int* __restrict__ shape, const int* __restrict__ nobj, int H, int W, int maxo, uint32_t* __restrict__ masks) {
    const int img = blockIdx.y, o = blockIdx.x;
    const int rw = raster_row_words(W);
    uint32_t* fill = masks + ((size_t)img * maxo + o) * 2 * H * rw;
    uint32_t* ring = fill + (size_t)H * rw;
    for (int i = threadIdx.x; i < 2 * H * rw; i += blockDim.x) fill[i] = 0u;
    if (o >= nobj[img]) return;
    __shared__ int s[SHAPE_INTS];
    __shared__ short pair_dx[RASTER_MAX_PAIRS], pair_dy[RASTER_MAX_PAIRS];
    __shared__ int npairs;
    __shared__ long long ex[4], edx[4];
    __shared__ int ey0[4], ey1[4], nedge;
    if (threadIdx.x < SHAPE_INTS) s[threadIdx.x] = shape[((size_t)img * maxo + o) * SHAPE_INTS + threadIdx.x];
    __syncthreads();
    if (s[0] == 0) {
        // Circle(): the octant walk is sequential and short; thread 0 records it, then every (step, row pair) is painted in parallel
        const int cx = s[2], cy = s[3], r = s[4];
        if (threadIdx.x == 0) {
            int err = 0, dx = r, dy = 0, plus = 1, minus = (r << 1) - 1, n = 0;
            while (dx >= dy && n < RASTER_MAX_PAIRS) {
                pair_dx[n] = (short)dx; pair_dy[n] = (short)dy; ++n;
                ++dy; err += plus; plus += 2;
                if (err > 0) { err -= minus; --dx; minus -= 2; }        // mask = (err <= 0) - 1
            }
            npairs = n;
        }
        __syncthreads();
        for (int it = threadIdx.x; it < 4 * npairs; it += blockDim.x) {
            const int k = it >> 2, which = it & 3;
            const int dx = pair_dx[k], dy = pair_dy[k];
            const int yy = which == 0 ? cy - dy : which == 1 ? cy + dy : which == 2 ? cy - dx : cy + dx;
            const int hw = which < 2 ? dx : dy;
            set_run(fill, rw, H, W, yy, cx - hw, cx + hw);
            set_px(ring, rw, H, W, cx - hw, yy);
            set_px(ring, rw, H, W, cx + hw, yy);
        }
        return;
    }
    const int nv = s[1];
    if (threadIdx.x == 0) {
        // CollectPolyEdges(): the non-horizontal edges, x in 16.16 at the upper end point, dx per scan line (C++ integer division)
        int n = 0;
        int px = s[2 + 2 * (nv - 1)], py = s[3 + 2 * (nv - 1)];
        for (int i = 0; i < nv; ++i) {
            const int qx = s[2 + 2 * i], qy = s[3 + 2 * i];
            if (py != qy) {
                const bool down = py < qy;
                ey0[n] = down ? py : qy; ey1[n] = down ? qy : py;
                ex[n] = (long long)(down ? px : qx) << XY_SHIFT;
                edx[n] = (((long long)qx - px) << XY_SHIFT) / ((long long)qy - py);
                ++n;
            }
            px = qx; py = qy;
        }
        nedge = n;
    }
    if ((int)threadIdx.x < nv) {                  // the outline: one Line() per edge, on both planes (the fill draws it too)
        const int i = threadIdx.x, j = i == 0 ? nv - 1 : i - 1;
        draw_line(ring, rw, H, W, s[2 + 2 * j], s[3 + 2 * j], s[2 + 2 * i], s[3 + 2 * i]);
        draw_line(fill, rw, H, W, s[2 + 2 * j], s[3 + 2 * j], s[2 + 2 * i], s[3 + 2 * i]);
    }
    __syncthreads();
    // (the pre-4.5.2 form of the rule - ceil left, floor right, no half-pixel offset; oracle/datagen.py:cv_poly_masks records the version doubt)
    // FillEdgeCollection(): per scan line the active edges in ascending x, consecutive pairs bound a run
    for (int y = threadIdx.x; y < H; y += blockDim.x) {
        long long ax[4], adx[4];
        int na = 0;
        for (int e = 0; e < nedge; ++e)
            if (ey0[e] <= y && y < ey1[e]) { ax[na] = ex[e] + (long long)(y - ey0[e]) * edx[e]; adx[na] = edx[e]; ++na; }
        for (int i = 1; i < na; ++i)              // insertion sort by (x, dx)
            for (int j = i; j > 0 && (ax[j] < ax[j - 1] || (ax[j] == ax[j - 1] && adx[j] < adx[j - 1])); --j) {
                const long long t = ax[j]; ax[j] = ax[j - 1]; ax[j - 1] = t;
                const long long u = adx[j]; adx[j] = adx[j - 1]; adx[j - 1] = u;
            }
        for (int k = 0; k + 1 < na; k += 2) {
            const long long x1 = (ax[k] + (1LL << XY_SHIFT) - 1) >> XY_SHIFT, x2 = ax[k + 1] >> XY_SHIFT;
            if (x1 < W && x2 >= 0) set_run(fill, rw, H, W, y, (int)max(x1, 0LL), (int)min(x2, (long long)W - 1));
        }
    }
}

// the two planes of object o of image img
__device__ __forceinline__ const uint32_t* fill_plane(const uint32_t* masks, int img, int o, int maxo, int H, int rw) {
    return masks + ((size_t)img * maxo + o) * 2 * H * rw;
}

__device__ __forceinline__ int reflect(int i, int n) {            // scipy 'reflect': (d c b a | a b c d | d c b a)
    if (i < 0) i = -i - 1;
    if (i >= n) i = 2 * n - 1 - i;
    return i;
}

__global__ __launch_bounds__(256)
void k_scene(const uint32_t* __restrict__ masks, const double* __restrict__ prop, const int* __restrict__ nobj,
             const double* __restrict__ bg, int H, int W, int maxo, double z_far, double* __restrict__ aif,
             double* __restrict__ bloc, double* __restrict__ idep, double* __restrict__ bdep) {
    __shared__ double pr[MAXO_LDS * PROP_F64];
    const int img = blockIdx.y;
    const int no = nobj[img];
    for (int i = threadIdx.x; i < no * PROP_F64; i += blockDim.x) pr[i] = prop[(size_t)img * maxo * PROP_F64 + i];
    __syncthreads();
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= H * W) return;
    const int y = p / W, x = p - y * W;
    const int rw = raster_row_words(W);
    double c0 = bg[img * 3], c1 = bg[img * 3 + 1], c2 = bg[img * 3 + 2];
    double bl = 0.0, dep = z_far, bd = 0.0;
    for (int o = 0; o < no; ++o) {
        const uint32_t* fill = fill_plane(masks, img, o, maxo, H, rw);
        const uint32_t* ring = fill + (size_t)H * rw;
        const double z = pr[o * PROP_F64];
        // 3x3 dilations of the filled mask and of the outline (:77-78); boundary depth where the dilated fill is (:84-85)
        bool fill_d = false, ol_d = false;
        for (int dy = -1; dy <= 1; ++dy)
            for (int dx = -1; dx <= 1; ++dx) {
                const int xx = x + dx, yy = y + dy;
                if ((unsigned)xx >= (unsigned)W || (unsigned)yy >= (unsigned)H) continue;
                fill_d = fill_d || plane_bit(fill, rw, xx, yy);
                ol_d = ol_d || plane_bit(ring, rw, xx, yy);
            }
        if (fill_d) bd = ol_d ? z : 0.0;
        if (plane_bit(fill, rw, x, y)) {
            dep = z;
            bl = plane_bit(ring, rw, x, y) ? 255.0 : 0.0;
            c0 = pr[o * PROP_F64 + 1]; c1 = pr[o * PROP_F64 + 2]; c2 = pr[o * PROP_F64 + 3];
        }
    }
    const size_t e = (size_t)img * H * W + p;
    aif[e * 3] = c0 / 255.0; aif[e * 3 + 1] = c1 / 255.0; aif[e * 3 + 2] = c2 / 255.0;       // stored /255 (:137)
    bloc[e] = bl; idep[e] = dep; bdep[e] = bd;
}

__global__ void k_fill_bg(const double* __restrict__ bg, double* __restrict__ imgs, int64_t per_img, int64_t total) {
    const int64_t gs = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gs) imgs[i] = bg[(i / per_img) * 3 + i % 3];
}

__global__ void k_mask(const uint32_t* __restrict__ masks, const int* __restrict__ nobj, int o, int H, int W, int maxo,
                       unsigned char* __restrict__ mask) {
    const int img = blockIdx.y;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= H * W || o >= nobj[img]) return;
    const int y = p / W, x = p - y * W;
    const int rw = raster_row_words(W);
    mask[(size_t)img * H * W + p] = plane_bit(fill_plane(masks, img, o, maxo, H, rw), rw, x, y) ? 1 : 0;
}

// tmp[img][a][y][x] = sum_d g(d) mask(y, reflect(x + d)),  g(d) = exp(-d^2 / (2 sigma^2))
__global__ void k_blur_h(const unsigned char* __restrict__ mask, const double* __restrict__ sig, const int* __restrict__ nobj,
                         int o, int H, int W, int maxo, double* __restrict__ tmp) {
    const int img = blockIdx.y;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= H * W || o >= nobj[img]) return;
    const int y = p / W, x = p - y * W;
    const unsigned char* m = mask + ((size_t)img * H + y) * W;
    for (int a = 0; a < 2; ++a) {
        const double s = fmax(sig[((size_t)img * maxo + o) * 2 + a], 1e-6);
        const int k = (int)ceil(fabs(s) * 3.0);
        const double inv = 1.0 / (2.0 * s * s);
        double acc = 0.0;
        for (int d = -k; d <= k; ++d)
            if (m[reflect(x + d, W)]) acc += exp(-(double)(d * d) * inv);
        tmp[(((size_t)img * 2 + a) * H + y) * W + x] = acc;
    }
}

__global__ void k_blur_v_composite(const double* __restrict__ tmp, const double* __restrict__ sig, const double* __restrict__ prop,
                                   const int* __restrict__ nobj, int o, int H, int W, int maxo, double* __restrict__ imgs) {
    const int img = blockIdx.y;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= H * W || o >= nobj[img]) return;
    const int y = p / W, x = p - y * W;
    const double* col = prop + ((size_t)img * maxo + o) * PROP_F64 + 1;
    for (int a = 0; a < 2; ++a) {
        const double s = fmax(sig[((size_t)img * maxo + o) * 2 + a], 1e-6);
        const int k = (int)ceil(fabs(s) * 3.0);
        const double inv = 1.0 / (2.0 * s * s);
        const double* t = tmp + ((size_t)img * 2 + a) * H * W;
        double acc = 0.0, norm = 0.0;
        for (int d = -k; d <= k; ++d) {
            const double g = exp(-(double)(d * d) * inv);
            norm += g;
            acc += g * t[(size_t)reflect(y + d, H) * W + x];
        }
        const double mb = 255.0 * (acc / (norm * norm));             // the blurred 0/255 mask
        if (mb > 0.0) {
            const double w = mb / 255.0;
            double* px = imgs + ((((size_t)img * 2 + a) * H + y) * W + x) * 3;
#pragma unroll
            for (int j = 0; j < 3; ++j) px[j] = w * col[j] + (1.0 - w) * px[j];
        }
    }
}

__global__ void k_round(double* __restrict__ x, int64_t n) {
    const int64_t gs = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gs) x[i] = rint(x[i]);
}

constexpr int FAR = 1 << 28;

__global__ void k_row_dist(const double* __restrict__ bloc, int H, int W, int* __restrict__ d1) {
    const int img = blockIdx.y;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= H * W) return;
    const int y = p / W, x = p - y * W;
    const double* row = bloc + ((size_t)img * H + y) * W;
    int best = FAR;
    for (int xx = 0; xx < W; ++xx)
        if (row[xx] > 0.0) best = min(best, abs(xx - x));
    d1[(size_t)img * H * W + p] = best;
}

__global__ void k_col_dist(const int* __restrict__ d1, int H, int W, double* __restrict__ dist) {
    const int img = blockIdx.y;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= H * W) return;
    const int y = p / W, x = p - y * W;
    const int* col = d1 + (size_t)img * H * W + x;
    int best = FAR;
    for (int yy = 0; yy < H; ++yy) best = min(best, col[(size_t)yy * W] + abs(yy - y));
    dist[(size_t)img * H * W + p] = best >= FAR ? 1.0 : (double)best;     // no boundary at all: the reference yields ones
}

// imgs [NI, H, W, 3] (NI = 2N aperture images), deri same shape: sqrt(gx^2 + gy^2) / 255, reflected borders
__global__ void k_sobel(const double* __restrict__ imgs, int H, int W, double* __restrict__ deri) {
    const int im = blockIdx.y;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= H * W) return;
    const int y = p / W, x = p - y * W;
    const double* I = imgs + (size_t)im * H * W * 3;
    const int ym = reflect(y - 1, H), yp = reflect(y + 1, H), xm = reflect(x - 1, W), xp = reflect(x + 1, W);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        auto at = [&](int yy, int xx) { return I[((size_t)yy * W + xx) * 3 + j]; };
        const double gx = (at(ym, xp) + 2.0 * at(y, xp) + at(yp, xp)) - (at(ym, xm) + 2.0 * at(y, xm) + at(yp, xm));
        const double gy = (at(ym, xm) + 2.0 * at(ym, x) + at(ym, xp)) - (at(yp, xm) + 2.0 * at(yp, x) + at(yp, xp));
        deri[((size_t)im * H * W + p) * 3 + j] = sqrt(gx * gx + gy * gy) / 255.0;
    }
}

// ---- noise ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t splitmix(uint64_t x) {
    x += 0x9E3779B97F4A7C15ULL;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}

struct Stream {                                       // one independent stream per element: hash(key, element) + counter
    uint64_t state;
    __device__ double next() {                        // uniform in (0,1)
        state = splitmix(state);
        return ((double)(state >> 11) + 0.5) * (1.0 / 9007199254740992.0);
    }
};

__device__ double poisson(Stream& r, double lam) {
    if (!(lam > 0.0)) return 0.0;
    if (lam < 10.0) {                                 // multiplication method
        const double lim = exp(-lam);
        double prod = r.next();
        int k = 0;
        while (prod > lim) { prod *= r.next(); ++k; }
        return (double)k;
    }
    // transformed rejection with squeeze (Hoermann 1993), the method numpy uses for lam >= 10
    const double slam = sqrt(lam), loglam = log(lam);
    const double b = 0.931 + 2.53 * slam, a = -0.059 + 0.02483 * b;
    const double invalpha = 1.1239 + 1.1328 / (b - 3.4), vr = 0.9277 - 3.6224 / (b - 2.0);
    for (int it = 0; it < 1000; ++it) {
        const double U = r.next() - 0.5, V = r.next();
        const double us = 0.5 - fabs(U);
        const double k = floor((2.0 * a / us + b) * U + lam + 0.43);
        if (us >= 0.07 && V <= vr) return k;
        if (k < 0.0 || (us < 0.013 && V > us)) continue;
        if (log(V) + log(invalpha) - log(a / (us * us) + b) <= -lam + k * loglam - lgamma(k + 1.0)) return k;
    }
    return floor(lam);                                // unreachable in practice (acceptance > 0.9 per round)
}

__global__ void k_noise(const double* __restrict__ imgs, const double* __restrict__ alpha, double sigma, uint64_t key,
                        int64_t per_sample, int64_t total, double* __restrict__ gt, double* __restrict__ ny) {
    const int64_t gs = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gs) {
        const double a = alpha[i / per_sample];
        const double lam = imgs[i] / 255.0 * a;
        Stream r{splitmix(key ^ ((uint64_t)i * 0xD1B54A32D192ED03ULL))};
        const double k = poisson(r, lam);
        const double n = sqrt(-2.0 * log(r.next())) * cos(6.283185307179586 * r.next());
        double v = k + sigma * n;
        v = fmin(fmax(v, 0.0), a);
        gt[i] = lam;
        ny[i] = rint(v);
    }
}

// ---- patches -------------------------------------------------------------------------------------------------
__global__ void k_candidates(const double* __restrict__ bloc, int H, int W, int reach, int margin, unsigned char* __restrict__ cand) {
    const int img = blockIdx.y;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= H * W) return;
    const int y = p / W, x = p - y * W;
    bool ok = false;
    if (y >= margin && y < H - margin && x >= margin && x < W - margin) {
        const double* B = bloc + (size_t)img * H * W;
        for (int yy = max(0, y - reach); yy <= min(H - 1, y + reach) && !ok; ++yy)
            for (int xx = max(0, x - reach); xx <= min(W - 1, x + reach); ++xx)
                if (B[(size_t)yy * W + xx] > 0.0) { ok = true; break; }
    }
    cand[(size_t)img * H * W + p] = ok ? 1 : 0;
}

struct CropArgs {
    const double *aif, *gt, *ny, *deri, *idep, *bdep, *bloc, *alpha;
    const int64_t* pick;
    const int* aper;
    double *o_aif, *o_gt, *o_ny, *o_deri, *o_idep, *o_bdep, *o_bloc, *o_bdist, *o_alpha;
    int H, W, R;
};

__global__ void k_crop(CropArgs a) {
    const int patch = blockIdx.x;
    const int R = a.R, half = R / 2, RR = R * R;
    const int64_t flat = a.pick[patch];
    const int64_t img = flat / ((int64_t)a.H * a.W);
    const int cy = (int)((flat / a.W) % a.H), cx = (int)(flat % a.W);
    const int ap = a.aper[patch];
    const int y0 = cy - half, x0 = cx - half;
    extern __shared__ unsigned char bnd[];                      // R*R boundary flags of this patch
    for (int q = threadIdx.x; q < RR; q += blockDim.x) {
        const int r = q / R, c = q - r * R;
        const size_t src = ((size_t)img * a.H + y0 + r) * a.W + x0 + c;
        const size_t src2 = (((size_t)img * 2 + ap) * a.H + y0 + r) * a.W + x0 + c;
        const size_t dst = (size_t)patch * RR + q;
        const double b = a.bloc[src];
        bnd[q] = b > 0.0;
        a.o_bloc[dst] = b; a.o_idep[dst] = a.idep[src]; a.o_bdep[dst] = a.bdep[src];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            a.o_aif[dst * 3 + j] = a.aif[src * 3 + j];
            a.o_gt[dst * 3 + j] = a.gt[src2 * 3 + j];
            a.o_ny[dst * 3 + j] = a.ny[src2 * 3 + j];
            a.o_deri[dst * 3 + j] = a.deri[src2 * 3 + j];
        }
    }
    if (threadIdx.x == 0) a.o_alpha[patch] = a.alpha[img];
    __syncthreads();
    for (int q = threadIdx.x; q < RR; q += blockDim.x) {       // city-block distance to the nearest boundary pixel IN the patch
        const int r = q / R, c = q - r * R;
        int best = FAR;
        for (int t = 0; t < RR; ++t)
            if (bnd[t]) best = min(best, abs(t / R - r) + abs(t % R - c));
        a.o_bdist[(size_t)patch * RR + q] = best >= FAR ? 1.0 : (double)best;
    }
}

inline unsigned cap(int64_t total, int block, int64_t limit = 8192) {
    int64_t g = (total + block - 1) / block;
    return (unsigned)(g < 1 ? 1 : (g > limit ? limit : g));
}

int dims_ok(const char* who, int64_t n, int H, int W, int maxo) {
    BE_REQUIRE(n > 0 && n < 65536 && H >= 3 && W >= 3 && (int64_t)H * W < (1 << 24), "%s: bad image batch (n < 65536)", who);
    BE_REQUIRE(maxo > 0 && maxo <= MAXO_LDS, "%s: at most %d objects per image", who, MAXO_LDS);
    return BE_OK;
}

}  // namespace

extern "C" size_t be_datagen_raster_words(int n, int H, int W, int maxo) {
    return (size_t)n * maxo * 2 * H * raster_row_words(W);
}

extern "C" int be_datagen_raster_u32(const int* shape, const int* nobj, int n, int H, int W, int maxo, uint32_t* masks, void* stream) {
    BE_REQUIRE(shape && nobj && masks, "be_datagen_raster_u32: null pointer");
    if (int rc = dims_ok("be_datagen_raster_u32", n, H, W, maxo)) return rc;
    hipLaunchKernelGGL(k_raster, dim3(maxo, n), dim3(256), 0, be::as_stream(stream), shape, nobj, H, W, maxo, masks);
    return be::check_launch("be_datagen_raster_u32");
}

extern "C" int be_datagen_scene_f64(const uint32_t* masks, const double* prop, const int* nobj, const double* bg, int n, int H,
                                    int W, int maxo, double z_far, double* aif, double* bloc, double* idep, double* bdep,
                                    void* stream) {
    BE_REQUIRE(masks && prop && nobj && bg && aif && bloc && idep && bdep, "be_datagen_scene_f64: null pointer");
    if (int rc = dims_ok("be_datagen_scene_f64", n, H, W, maxo)) return rc;
    hipLaunchKernelGGL(k_scene, dim3((H * W + 255) / 256, n), dim3(256), 0, be::as_stream(stream), masks, prop, nobj, bg, H, W,
                       maxo, z_far, aif, bloc, idep, bdep);
    return be::check_launch("be_datagen_scene_f64");
}

extern "C" size_t be_datagen_blur_scratch_bytes(int n, int H, int W) {
    const size_t px = (size_t)n * H * W;
    return ((px + 255) / 256) * 256 + px * 2 * sizeof(double);      // mask bytes | tmp f64 [n,2,H,W] (also reused as int32 [n,H,W])
}

extern "C" int be_datagen_blur_composite_f64(const uint32_t* masks, const double* prop, const int* nobj, const double* bg,
                                             const double* sig, int n, int H, int W, int maxo, int max_nobj, double* imgs,
                                             void* scratch, size_t scratch_bytes, void* stream) {
    BE_REQUIRE(masks && prop && nobj && bg && sig && imgs && scratch, "be_datagen_blur_composite_f64: null pointer");
    if (int rc = dims_ok("be_datagen_blur_composite_f64", n, H, W, maxo)) return rc;
    BE_REQUIRE(max_nobj >= 0 && max_nobj <= maxo, "be_datagen_blur_composite_f64: max_nobj outside [0, maxo]");
    BE_REQUIRE(scratch_bytes >= be_datagen_blur_scratch_bytes(n, H, W), "be_datagen_blur_composite_f64: scratch too small");
    hipStream_t s = be::as_stream(stream);
    const size_t px = (size_t)n * H * W;
    unsigned char* mask = static_cast<unsigned char*>(scratch);
    double* tmp = reinterpret_cast<double*>(mask + ((px + 255) / 256) * 256);
    const int64_t total = (int64_t)px * 2 * 3;
    hipLaunchKernelGGL(k_fill_bg, dim3(cap(total, 256)), dim3(256), 0, s, bg, imgs, (int64_t)2 * H * W * 3, total);
    const dim3 grid((H * W + 255) / 256, n);
    for (int o = 0; o < max_nobj; ++o) {
        hipLaunchKernelGGL(k_mask, grid, dim3(256), 0, s, masks, nobj, o, H, W, maxo, mask);
        hipLaunchKernelGGL(k_blur_h, grid, dim3(256), 0, s, mask, sig, nobj, o, H, W, maxo, tmp);
        hipLaunchKernelGGL(k_blur_v_composite, grid, dim3(256), 0, s, tmp, sig, prop, nobj, o, H, W, maxo, imgs);
    }
    return be::check_launch("be_datagen_blur_composite_f64");
}

extern "C" int be_datagen_finish_f64(double* imgs, const double* bloc, double* bdist, double* deri, int n, int H, int W,
                                     void* scratch, size_t scratch_bytes, void* stream) {
    BE_REQUIRE(imgs && bloc && bdist && deri && scratch, "be_datagen_finish_f64: null pointer");
    if (int rc = dims_ok("be_datagen_finish_f64", n, H, W, 1)) return rc;
    BE_REQUIRE(scratch_bytes >= (size_t)n * H * W * sizeof(int), "be_datagen_finish_f64: scratch too small");
    hipStream_t s = be::as_stream(stream);
    const int64_t total = (int64_t)n * 2 * H * W * 3;
    hipLaunchKernelGGL(k_round, dim3(cap(total, 256)), dim3(256), 0, s, imgs, total);
    int* d1 = static_cast<int*>(scratch);
    const dim3 grid((H * W + 255) / 256, n);
    hipLaunchKernelGGL(k_row_dist, grid, dim3(256), 0, s, bloc, H, W, d1);
    hipLaunchKernelGGL(k_col_dist, grid, dim3(256), 0, s, d1, H, W, bdist);
    hipLaunchKernelGGL(k_sobel, dim3((H * W + 255) / 256, 2 * n), dim3(256), 0, s, imgs, H, W, deri);
    return be::check_launch("be_datagen_finish_f64");
}

extern "C" int be_datagen_noise_f64(const double* imgs, const double* alpha, double sigma, uint32_t seed, int64_t n,
                                    int64_t per_sample, double* gt, double* ny, void* stream) {
    BE_REQUIRE(imgs && alpha && gt && ny && n > 0 && per_sample > 0, "be_datagen_noise_f64: bad arguments");
    const int64_t total = n * per_sample;
    const uint64_t key = 0x6e6f697365ULL ^ ((uint64_t)seed << 32);
    hipLaunchKernelGGL(k_noise, dim3(cap(total, 256, 65536)), dim3(256), 0, be::as_stream(stream), imgs, alpha, sigma, key,
                       per_sample, total, gt, ny);
    return be::check_launch("be_datagen_noise_f64");
}

extern "C" int be_datagen_candidates_f64(const double* bloc, unsigned char* cand, int n, int H, int W, int reach, int margin,
                                         void* stream) {
    BE_REQUIRE(bloc && cand && reach >= 0 && margin >= 0, "be_datagen_candidates_f64: bad arguments");
    if (int rc = dims_ok("be_datagen_candidates_f64", n, H, W, 1)) return rc;
    hipLaunchKernelGGL(k_candidates, dim3((H * W + 255) / 256, n), dim3(256), 0, be::as_stream(stream), bloc, H, W, reach, margin,
                       cand);
    return be::check_launch("be_datagen_candidates_f64");
}

extern "C" int be_datagen_crop_f64(const double* const* in6, const double* bloc, const double* alpha, const int64_t* pick,
                                   const int* aper, int64_t n_patch, int n, int H, int W, int R, double* const* out9,
                                   void* stream) {
    BE_REQUIRE(in6 && out9 && bloc && alpha && pick && aper && n_patch > 0 && n_patch < ((int64_t)1 << 31),
               "be_datagen_crop_f64: bad arguments");
    BE_REQUIRE(R > 0 && R % 2 == 1 && R <= 63 && R <= H && R <= W, "be_datagen_crop_f64: R must be odd, <= 63 and fit the image");
    for (int i = 0; i < 6; ++i) BE_REQUIRE(in6[i], "be_datagen_crop_f64: input %d is null", i);
    for (int i = 0; i < 9; ++i) BE_REQUIRE(out9[i], "be_datagen_crop_f64: output %d is null", i);
    CropArgs a{in6[0], in6[1], in6[2], in6[3], in6[4], in6[5], bloc, alpha, pick, aper,
               out9[0], out9[1], out9[2], out9[3], out9[4], out9[5], out9[6], out9[7], out9[8], H, W, R};
    hipLaunchKernelGGL(k_crop, dim3((unsigned)n_patch), dim3(256), (size_t)R * R, be::as_stream(stream), a);
    return be::check_launch("be_datagen_crop_f64");
}
