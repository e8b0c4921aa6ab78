#!/usr/bin/env python3
"""Experiment: do two half-batches on two streams overlap the matrix-bound and the HBM-bound kernels of LocalStage?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "blurry-edges_amd")]
import numpy as np, torch
from be_hip import synth
import models
dev = "cuda:0"
x_np, _ = synth.synthetic_patch_pairs(4096, seed=synth.SEED_DEFAULT)
sd = {k: torch.from_numpy(np.asarray(v)) for k, v in synth.local_stage_state_dict().items()}
def mk():
    m = models.LocalStage(); m.load_state_dict(sd); return m.to(dev).eval()
m0, m1, m2 = mk(), mk(), mk()
x = torch.from_numpy(x_np).to(dev)
xa, xb = x[:4096].contiguous(), x[4096:].contiguous()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def one():
    with torch.no_grad(): return m0(x)
LAG = int(os.environ.get("LAG_KCYCLES", "0")) * 1000       # stream 2 starts this many GPU cycles later (staggers MFMA / HBM phases)
def two():
    s1.wait_stream(torch.cuda.current_stream()); s2.wait_stream(torch.cuda.current_stream())
    with torch.no_grad():
        with torch.cuda.stream(s1): a = m1(xa)
        with torch.cuda.stream(s2):
            if LAG: torch.cuda._sleep(LAG)
            b = m2(xb)
    torch.cuda.current_stream().wait_stream(s1); torch.cuda.current_stream().wait_stream(s2)
    return a, b
def timed(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
r1 = one(); ra, rb = two()
print("identical:", torch.equal(r1[:4096], ra) and torch.equal(r1[4096:], rb))
print(f"one stream, 8192 patches: {timed(one):.3f} ms   two streams x 4096: {timed(two):.3f} ms")

# ---- N-way split (BE_WINO_WS_MIN_TILES lowers the weight-stationary kernel's minimum so that 2048-patch parts keep it)
NW = int(os.environ.get("NWAY", "0"))
if NW:
    m0.streams = 1
    ms = [mk() for _ in range(NW)]
    for m in ms: m.streams = 1
    ss = [torch.cuda.Stream() for _ in range(NW)]
    per = 8192 // NW
    xs = [x[i * per:(i + 1) * per].contiguous() for i in range(NW)]
    def nway():
        cur = torch.cuda.current_stream()
        outs = []
        with torch.no_grad():
            for m, s, xx in zip(ms, ss, xs):
                s.wait_stream(cur)
                with torch.cuda.stream(s): outs.append(m(xx))
        for s in ss: cur.wait_stream(s)
        return outs
    o = nway()
    print("identical:", torch.equal(torch.cat(o), r1), f"  {NW} streams x {per}: {timed(nway):.3f} ms")

# ---- asymmetric two-way split (desynchronises the two streams' phases): SPLIT = patches of the first part
SP = int(os.environ.get("SPLIT", "0"))
if SP:
    m1.streams = m2.streams = 1
    xa2, xb2 = x[:SP].contiguous(), x[SP:].contiguous()
    def asym():
        s1.wait_stream(torch.cuda.current_stream()); s2.wait_stream(torch.cuda.current_stream())
        with torch.no_grad():
            with torch.cuda.stream(s1): a = m1(xa2)
            with torch.cuda.stream(s2): b = m2(xb2)
        torch.cuda.current_stream().wait_stream(s1); torch.cuda.current_stream().wait_stream(s2)
        return a, b
    a, b = asym()
    print("identical:", torch.equal(torch.cat([a, b]), r1), f"  split {SP}/{8192 - SP}: {timed(asym):.3f} ms")
