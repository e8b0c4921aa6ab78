#!/usr/bin/env python3
"""Micro-benchmark of the GlobalStage attention kernels (forward inference / training forward / backward)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "blurry-edges_amd")]
import numpy as np, torch
from be_hip import native, synth, train_global_stage as tg

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
L, H, p = 4096, 8, (float(sys.argv[3]) if len(sys.argv) > 3 else 0.1)
dev = "cuda:0"
qkv = torch.from_numpy(synth.hash_normal(3, "attn_qkv", (B * L, 384)).astype(np.float32)).to(dev)
dout = torch.from_numpy(synth.hash_normal(4, "attn_dout", (B * L, 128)).astype(np.float32)).to(dev)


def timed(f):
    f(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): r = f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / reps * 1e3, r


t_inf, _ = timed(lambda: native.attention(qkv, B, L, H))
t_fwd, (out, lse, ws) = timed(lambda: tg.attention_train_fwd(qkv, B, L, H, p, 7))
t_bwd, _ = timed(lambda: tg.attention_bwd(qkv, out, lse, dout, B, L, H, p, 7, ws, operands_ready=True))   # as the training step calls it
fl = 4.0 * L * L * 16 * B * H
print(f"B={B}: inference {t_inf:.3f} ms ({fl / t_inf / 1e9:.1f} TF alg)  train fwd {t_fwd:.3f} ms ({fl / t_fwd / 1e9:.1f} TF)  "
      f"bwd {t_bwd:.3f} ms ({2.5 * fl / t_bwd / 1e9:.1f} TF alg)")
