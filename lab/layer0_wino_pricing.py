#!/usr/bin/env python3
"""Layer0 (11 x 11 maps, 64 -> 96 -> 96 channels, 8192 patches) on F(6,3) x F(3,3) Winograd tiles: the A/B priced with MEASURED parts
 instead of built.  An 11 x 11 map needs 2 x 4 tiles of 8 x 5 transform positions: 40 positions x (8 tiles x 8192
patches = 65 536 rows) per convolution.  Measured here on the GPU, with real buffers of the real sizes:
  * the transform-domain GEMMs as ONE row GEMM of 40 x 65 536 rows (the same FLOPs, the same bytes; the weights are 40 small matrices
    instead of one) on the product's kernels - K = 64 and K = 96, N = 96 (the generic 128 x 96 tiles) and N = 128 (the LDS-DMA row GEMM);
  * the traffic of the four transform kernels (V and M written once, read once) as a device copy of the same bytes.
The direct launches it would replace: k_conv_pm<2,3,TAPS3> x 2 = 2.19-2.21 ms (profiles/r06_v31_kernel_stats.csv)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "blurry-edges_amd")]
from be_hip import native  # noqa: E402

dev = "cuda:0"
ROWS = 40 * 8 * 8192


def timed(f, reps=5):
    f(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e3


tot = {}
for cin, cout in ((64, 96), (96, 96), (64, 128), (96, 128)):
    x = torch.randn(ROWS // 64, 8, 8, cin, device=dev)
    w = torch.randn(cout, cin, 1, 1, device=dev) / cin ** 0.5
    pw, pb = native.conv_pack(w, torch.zeros(cout, device=dev))
    y = torch.empty(ROWS // 64, 8, 8, cout, device=dev)
    t = timed(lambda: native.conv_nhwc(x, pw, pb, cout, 1, 0, out=y))
    fl = 2.0 * ROWS * cin * cout
    gb = 4.0 * ROWS * (cin + cout) / 1e9
    print(f"GEMM {ROWS} rows x K {cin} -> N {cout}: {t:.3f} ms = {fl / t / 1e9:.0f} TF, operand + result traffic {gb:.2f} GB = {gb / t:.2f} TB/s")
    tot[(cin, cout)] = t
    del x, y
nbytes = 8192 * 320 * (64 + 96 + 96 + 96) * 4          # V + M of both convolutions
a = torch.empty(nbytes // 8, device=dev); b = torch.empty_like(a)      # half the bytes: a copy reads and writes them once each
tc = timed(lambda: b.copy_(a))
print(f"transform-domain buffers {nbytes / 1e9:.2f} GB, written once + read once as a device copy: {tc:.3f} ms = {nbytes / tc / 1e9:.2f} TB/s")
print(f"sum, N = 96 tiles:  {tot[(64, 96)] + tot[(96, 96)] + tc:.3f} ms   N = 128 (row GEMM): {tot[(64, 128)] + tot[(96, 128)] + tc:.3f} ms   "
      f"(+ ~0.1 ms for the 1x1 downsample that rides in conv2's K loop today)   direct today: 2.19-2.21 ms")
