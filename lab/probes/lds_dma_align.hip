// Probe: does global_load_lds_dwordx4 (16 bytes per lane, global -> LDS without a VGPR round trip) accept a GLOBAL address that
// is only 4-byte aligned?  (A 3-channel, 12-byte-pixel staging of conv1's input would need it: DESIGN 8.2.)
//   hipcc --offload-arch=gfx950 -O2 tools/probes/lds_dma_align.hip -o tools/bin/lds_dma_align && tools/bin/lds_dma_align
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef const __attribute__((address_space(1))) void* glb_ptr_t;
typedef __attribute__((address_space(3))) void* lds_ptr_t;

__global__ void k(const float* src, float* dst, int shift_floats) {
    __shared__ __attribute__((aligned(16))) float st[64 * 4];
    const int lane = threadIdx.x;
    // lane i fetches 16 bytes from src + shift + 4 i floats into LDS slot i (the hardware adds lane * 16 to the LDS base)
    __builtin_amdgcn_global_load_lds((glb_ptr_t)(src + shift_floats + 4 * lane), (lds_ptr_t)st, 16, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    for (int j = 0; j < 4; ++j) dst[4 * lane + j] = st[4 * lane + j];
}

int main() {
    std::vector<float> h(1024);
    for (int i = 0; i < 1024; ++i) h[i] = (float)i;
    float *s, *d;
    hipMalloc(&s, 4096); hipMalloc(&d, 1024);
    hipMemcpy(s, h.data(), 4096, hipMemcpyHostToDevice);
    for (int shift = 0; shift < 4; ++shift) {
        hipMemset(d, 0, 1024);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, s, d, shift);
        hipError_t e = hipDeviceSynchronize();
        std::vector<float> o(256);
        hipMemcpy(o.data(), d, 1024, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < 256; ++i) bad += o[i] != (float)(i + shift);
        printf("global address offset %2d bytes: %s, %d of 256 floats wrong (first four: %g %g %g %g)\n", 4 * shift,
               hipGetErrorString(e), bad, o[0], o[1], o[2], o[3]);
    }
    return 0;
}
