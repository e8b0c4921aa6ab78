#!/usr/bin/env python3
"""Winograd F(3x3,3x3) vs the direct implicit-GEMM convolution on the 6x6 layers of LocalStage at 8192 patches."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "blurry-edges_amd")]
import numpy as np, torch
from be_hip import native, synth
dev = "cuda:0"
N = 8192
def timed(f, reps=10):
    f(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / reps * 1e3
ws = None
for cin, cout in ((96, 256), (256, 256), (256, 384), (384, 384), (384, 256)):
    x = torch.randn(N, 6, 6, cin, device=dev)
    w = torch.randn(cout, cin, 3, 3, device=dev) / (3 * cin ** 0.5)
    b = torch.zeros(cout, device=dev)
    pw, pb = native.conv_pack(w, b)
    uw, ub = native.wino_pack(w, b)
    t_d = timed(lambda: native.conv_nhwc(x, pw, pb, cout, 3, 1))
    _, ws = native.wino_conv3x3(x, uw, ub, cout, act=1, workspace=ws)
    t_w = timed(lambda: native.wino_conv3x3(x, uw, ub, cout, act=1, workspace=ws))
    fl = 2.0 * N * 36 * 9 * cin * cout
    print(f"{cin:4d} -> {cout:4d}: direct {t_d:.3f} ms ({fl / t_d / 1e9:.0f} TF alg)   winograd {t_w:.3f} ms ({fl / t_w / 1e9:.0f} TF alg)   x{t_d / t_w:.2f}")
