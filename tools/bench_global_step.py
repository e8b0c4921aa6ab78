#!/usr/bin/env python3
"""GlobalStage training step at batch 8 (bench.py's leg_global_training on its own): ms per step."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "blurry-edges_amd")]
import torch
import bench
from be_hip import native
native.lib()
r = bench.leg_global_training(torch.device("cuda", 0), steps=int(sys.argv[1]) if len(sys.argv) > 1 else 12)
print({k: r[k] for k in ("ms_per_step", "images_per_s", "first_loss", "last_loss")})
