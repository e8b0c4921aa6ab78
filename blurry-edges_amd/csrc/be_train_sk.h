// Balanced ("stream-K") decomposition of a training unit's backward GEMMs (round 6): geometry shared by the producer
// (k_unit_gemms_sk, be_train_sk.hip) and the consumers of its slices (k_bwd_post, be_train.hip).
//
// local_training.py:103-106 (loss.backward() through Conv2d + BatchNorm at batch 64): every unit's weight-gradient GEMM and
// data-gradient convolution are a few hundred tiles whose K loops differ in length (border pixels see 4 or 6 taps of a 3x3
// kernel, border taps see 25 or 30 of 36 pixels).  Rounds 3-5 cut each tile's K loop into S equal slices and launched
// tiles x S workgroups: 864 workgroups of equal length on 768 slots is 1.125 waves.  Here every PROBLEM (a weight gradient
// or a convolution) lays its tiles end to end on one axis of K chunks, a launch has exactly as many workgroups as the chip
// has slots, and workgroup g of a problem takes the chunks [g Q, (g + 1) Q): a tile is cut where the quota ends, not where
// S says.  A tile's segments write slices 0, 1, ... of that tile; how many there are follows from the axis alone:
//     slices(tile) = floor((start + n - 1) / Q) - floor(start / Q) + 1,   slice of the segment at p = floor(p / Q) - floor(start / Q)
// so a consumer needs Q and the tile's start - no table in memory, no atomics, a fixed order of every sum (bitwise reproducible).
#pragma once
#include <hip/hip_runtime.h>

namespace be_sk {

constexpr int MAX_HW = 121;         // 11 x 11 maps (layer0); the 6 x 6 layers have 36
constexpr int MAX_SLICES_C = 8;     // slices per convolution tile: what k_bwd_post keeps in flight per element (the planner caps the workgroups accordingly)
constexpr int MAX_SLICES_W = 16;    // slices per weight-gradient tile (k_bwd_post walks them eight at a time)

// A convolution's tiles: ONE output pixel of 64 consecutive images x 64 output channels.  Tile (grp, pp, j) = image group,
// pixel (natural order), column tile; K chunks of 32 floats (one tap of a 32-channel unit): taps(pp) x kmul of them, kmul = Cin / 32.
struct ConvGeom {
    int HW, n_tiles, kmul, ngrp;
    int Q, L;                                   // quota per workgroup, total chunks
    unsigned short PP[MAX_HW + 1];              // PP[pp] = taps inside the image summed over the pixels < pp
};
// A weight gradient's tiles: tm cout x tn cin of ONE tap (tm, tn = 128, or 64 for channel counts that are not multiples of 128);
// K chunks = one valid output pixel of 16 consecutive images.  Tile (tap, j): j = cout tile * cin_tiles + cin tile.
struct WGeom {
    int wx, cin_tiles, ntaps;
    int tm, tn, cin;
    int Q, L;
    int PT[10];                                 // PT[t] = chunks of one tile of the taps < t
};

__host__ __device__ inline int slices_of(int ts, int n, int Q) { return (ts + n - 1) / Q - ts / Q + 1; }

__host__ __device__ inline void conv_span(const ConvGeom& g, int grp, int pp, int j, int& ts, int& n) {
    const int taps = g.PP[pp + 1] - g.PP[pp];
    ts = ((grp * (int)g.PP[g.HW] + (int)g.PP[pp]) * g.n_tiles + j * taps) * g.kmul;
    n = taps * g.kmul;
}
// the tile that holds position p of the axis (0 <= p < L)
__host__ __device__ inline void conv_find(const ConvGeom& g, int p, int& grp, int& pp, int& j) {
    const int q = p / g.kmul, gs = (int)g.PP[g.HW] * g.n_tiles;
    grp = q / gs;
    const int r = q - grp * gs;
    int lo = 0, hi = g.HW;                      // largest pp with PP[pp] * n_tiles <= r
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if ((int)g.PP[mid] * g.n_tiles <= r) lo = mid; else hi = mid;
    }
    pp = lo;
    j = (r - (int)g.PP[pp] * g.n_tiles) / ((int)g.PP[pp + 1] - (int)g.PP[pp]);
}

__host__ __device__ inline void w_span(const WGeom& g, int tap, int j, int& ts, int& n) {
    n = g.PT[tap + 1] - g.PT[tap];
    ts = g.PT[tap] * g.wx + j * n;
}
__host__ __device__ inline void w_find(const WGeom& g, int p, int& tap, int& j) {
    int t = 0;
    while (t + 1 < g.ntaps && g.PT[t + 1] * g.wx <= p) ++t;
    tap = t;
    j = (p - g.PT[t] * g.wx) / (g.PT[t + 1] - g.PT[t]);
}

// valid output pixels of a tap (tdy, tdx): the rectangle [y0, y0 + hv) x [x0, x0 + wv) whose shifted pixel is inside the image
__host__ __device__ inline void tap_rect(int H, int W, int tdy, int tdx, int& y0, int& hv, int& x0, int& wv) {
    y0 = tdy < 0 ? -tdy : 0; hv = H - (tdy < 0 ? -tdy : tdy);
    x0 = tdx < 0 ? -tdx : 0; wv = W - (tdx < 0 ? -tdx : tdx);
}

}  // namespace be_sk

// ---- host side (be_train_sk.hip), called by the unit entry points of be_train.hip
struct be_train_unit_bwd;
struct be_conv_desc;
namespace be {
struct SkUnitIn { const ::be_train_unit_bwd* u; float* cpart; size_t cpart_bytes; float* wpart; size_t wpart_bytes; };
struct SkUnitOut { be_sk::ConvGeom cg; be_sk::WGeom wg; int ldp; };
bool sk_enabled();                                     // BE_NO_TRAIN_SK unset
bool sk_eligible(const ::be_train_unit_bwd& u);        // shapes k_unit_gemms_sk takes
// the plan of the balanced backward GEMMs of nu (1 | 2) eligible units (kernel arguments + grid; nothing is launched):
// BE_OK, or 1 = "not mine" (scratch too small) - decided BEFORE the caller launches anything of the unit
struct SkPlan { alignas(16) char blob[1536]; };
int sk_plan(const SkUnitIn* in, int nu, SkUnitOut* out, SkPlan* plan);
int sk_run(const SkPlan* plan, hipStream_t s);
// the forward convolutions of nu (1 | 2) units on the same launch (conv-only problems): raw slices [slices][M][cout] at `part`
struct SkFwdIn { const ::be_conv_desc* d; const float* x; const float* packed_w; float* part; size_t part_bytes; };
bool sk_fwd_eligible(const ::be_conv_desc& d);
int sk_plan_fwd(const SkFwdIn* in, int nu, be_sk::ConvGeom* out, SkPlan* plan);
}  // namespace be
