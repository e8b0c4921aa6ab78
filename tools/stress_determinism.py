#!/usr/bin/env python3
"""Stress: the LocalStage forward of one 8192-patch batch, many times, must give bit-identical logits every time (any race in
the hand-rolled LDS-DMA pipelines - counted vmcnt, raw barriers - would show as a run-to-run difference)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "blurry-edges_amd")]
import numpy as np, torch
from be_hip import synth
import models
dev = "cuda:0"
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
x_np, _ = synth.synthetic_patch_pairs(4096, seed=synth.SEED_DEFAULT)
m = models.LocalStage()
m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synth.local_stage_state_dict().items()})
m = m.to(dev).eval()
x = torch.from_numpy(x_np).to(dev)
with torch.no_grad():
    ref = m(x).clone()
    bad = 0
    for i in range(reps):
        y = m(x)
        if not torch.equal(y, ref):
            bad += 1
            print("run", i, "differs in", int((y != ref).sum()), "values")
print(f"{reps} runs, {bad} differing")
sys.exit(1 if bad else 0)
