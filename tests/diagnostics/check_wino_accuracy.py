#!/usr/bin/env python3
"""Logits of the Winograd path and of the direct path against the float64 oracle for several weight sets (GPU)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # tests/diagnostics/ -> repo root
sys.path[:0] = [ROOT, os.path.join(ROOT, "blurry-edges_amd")]
import numpy as np, torch
from be_hip import synth
from oracle import local_stage as ols
import models
dev = "cuda:0"
x = torch.from_numpy(synth.uniform_patches(512, name="acc"))
xs = torch.from_numpy(synth.synthetic_patch_pairs(256)[0])
for seed in (1869, 7, 11, 13):
    sd = synth.local_stage_state_dict(seed=seed)
    m = models.LocalStage(); m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}); m = m.to(dev).eval()
    sd64 = ols.to_torch_sd(sd, torch.float64)
    for name, inp in (("uniform", x), ("synthetic pairs", xs)):
        ref = ols.local_stage_forward(sd64, inp.double())
        out = {}
        for w in (True, False):
            m.winograd = w
            with torch.no_grad():
                out[w] = m(inp.to(dev)).cpu().double()
        e = {w: float((out[w] - ref).abs().max() / ref.abs().max()) for w in out}
        print(f"seed {seed:5d} {name:16s}: winograd {e[True]:.2e}   direct {e[False]:.2e}")
