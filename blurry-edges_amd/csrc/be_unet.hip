// Glue kernels of the depth-completion U-Net (models/depth_completion_unet.py:79-113 of the reference).
// The convolutions (3x3 + folded BatchNorm + ReLU, the 1x1 head, and the 2x2 stride-2 transposed convolution written
// as a 1x1 convolution with 4*cout outputs) run on the implicit-GEMM kernel of be_conv.hip; what is left is layout:
// NCHW -> zero-padded NHWC staging of the one-channel input, and the pixel shuffle that scatters the 4 sub-pixel
// outputs of the transposed convolution into the channel-concatenated decoder input (torch.cat([skip, up]) with
// F.pad centring, depth_completion_unet.py:57-68), next to the skip tensor the encoder wrote there.
#include "be_common.h"

namespace {

inline unsigned grid_cap(int64_t total, int block) {
    int64_t g = (total + block - 1) / block;
    return (unsigned)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

__global__ void k_nchw_to_nhwc_pad(const float* __restrict__ x, float* __restrict__ y, int64_t n, int c, int64_t hw, int cpad) {
    const int64_t total = n * hw * cpad;
    const int64_t gs = (int64_t)gridDim.x * blockDim.x;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gs) {
        const int ch = (int)(idx % cpad);
        const int64_t p = idx / cpad;                     // img * hw + pixel
        const int64_t img = p / hw, px = p % hw;
        y[idx] = ch < c ? x[(img * c + ch) * hw + px] : 0.0f;
    }
}

// t [n,h,w,4*cout] (sub-pixel (dy,dx) major, channel minor) -> y[n, 2i+dy+top, 2j+dx+left, ch_off + co], row stride ldy
__global__ void k_upconv_scatter(const float* __restrict__ t, float* __restrict__ y, int64_t n, int h, int w, int cout4,
                                 int oh, int ow, int top, int left, int ldy4, int ch_off4) {
    const int64_t total = n * h * w * 4 * cout4;
    const int64_t gs = (int64_t)gridDim.x * blockDim.x;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gs) {
        const int cq = (int)(idx % cout4);
        int64_t r = idx / cout4;
        const int sub = (int)(r % 4); r /= 4;
        const int j = (int)(r % w); r /= w;
        const int i = (int)(r % h);
        const int64_t img = r / h;
        const int oy = 2 * i + (sub >> 1) + top, ox = 2 * j + (sub & 1) + left;
        if ((unsigned)oy >= (unsigned)oh || (unsigned)ox >= (unsigned)ow) continue;      // negative F.pad crops
        reinterpret_cast<float4*>(y)[((img * oh + oy) * ow + ox) * ldy4 + ch_off4 + cq] = reinterpret_cast<const float4*>(t)[idx];
    }
}

}  // namespace

extern "C" int be_nchw_to_nhwc_pad_f32(const float* x, float* y, int64_t n, int c, int64_t hw, int cpad, void* stream) {
    BE_REQUIRE(x && y && n > 0 && c > 0 && hw > 0 && cpad >= c, "be_nchw_to_nhwc_pad_f32: bad arguments");
    hipLaunchKernelGGL(k_nchw_to_nhwc_pad, dim3(grid_cap(n * hw * cpad, 256)), dim3(256), 0, be::as_stream(stream), x, y, n, c,
                       hw, cpad);
    return be::check_launch("be_nchw_to_nhwc_pad_f32");
}

extern "C" int be_upconv2x2_scatter_f32(const float* t, float* y, int64_t n, int h, int w, int cout, int oh, int ow, int top,
                                        int left, int ldy, int ch_off, void* stream) {
    BE_REQUIRE(t && y && n > 0 && h > 0 && w > 0 && cout > 0 && oh > 0 && ow > 0, "be_upconv2x2_scatter_f32: bad arguments");
    BE_REQUIRE(cout % 4 == 0 && ldy % 4 == 0 && ch_off % 4 == 0 && ch_off + cout <= ldy && be::aligned16(t) && be::aligned16(y),
               "be_upconv2x2_scatter_f32: channels must be multiples of 4 and fit the row stride");
    hipLaunchKernelGGL(k_upconv_scatter, dim3(grid_cap(n * h * w * cout, 256)), dim3(256), 0, be::as_stream(stream), t, y, n, h,
                       w, cout / 4, oh, ow, top, left, ldy / 4, ch_off / 4);
    return be::check_launch("be_upconv2x2_scatter_f32");
}
