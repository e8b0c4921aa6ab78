#!/usr/bin/env python3
"""Training batch (64 patches): Winograd F(3x3,3x3) (input transform + 25 batched GEMMs + output transform) against the split-K
direct convolution on the 6x6 layers, timed as replayed hipGraphs (launch cost excluded, as in the training step)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "blurry-edges_amd")]
import numpy as np, torch
from be_hip import native, synth, train
dev = "cuda:0"
N = int(sys.argv[1]) if len(sys.argv) > 1 else 64

def graph_time(f, reps=50):
    f(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(10):
            f()
    g.replay(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): g.replay()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / reps / 10 * 1e6

ws = None
sc = train._Scratch.get(torch.device(dev))
tot_d = tot_w = 0.0
for cin, cout in ((96, 256), (256, 256), (256, 384), (384, 384), (384, 256), (256, 256)):
    x = torch.randn(N, 6, 6, cin, device=dev)
    w = torch.randn(cout, cin, 3, 3, device=dev) / (3 * cin ** 0.5)
    b = torch.randn(cout, device=dev) * 0.1
    pw, pb = native.conv_pack(w, b)
    uw, ub = native.wino_pack(w, b)
    y_d = native.conv_nhwc(x, pw, pb, cout, 3, 0, scratch=sc)
    y_w, ws = native.wino_conv3x3(x, uw, ub, cout, act=0, workspace=ws)
    err = float((y_d - y_w).abs().max() / y_d.abs().max())
    t_d = graph_time(lambda: native.conv_nhwc(x, pw, pb, cout, 3, 0, scratch=sc))
    t_w = graph_time(lambda: native.wino_conv3x3(x, uw, ub, cout, act=0, workspace=ws))
    tot_d += t_d; tot_w += t_w
    fl = 2.0 * N * 36 * 9 * cin * cout
    print(f"{cin:4d} -> {cout:4d}: split-K direct {t_d:7.1f} us ({fl / t_d / 1e6:5.0f} TF alg)   winograd {t_w:7.1f} us ({fl / t_w / 1e6:5.0f} TF alg)   x{t_d / t_w:.2f}   rel diff {err:.1e}")
print(f"sum: direct {tot_d:.0f} us, winograd {tot_w:.0f} us")
