#!/usr/bin/env python3
"""Times utils.global_loss (records + fold + Sobel + the fused loss / gradient kernel) at batch 8, 147 x 147 pairs."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "blurry-edges_amd")]
import numpy as np, torch
import utils
from be_hip import synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda:0")
a = utils.get_args("global_train", argv=[])
a.batch_size = B
helper, dcal = utils.PostProcessGlobalBase(a, dev), utils.DepthEtas(a, dev)
one = {k: torch.from_numpy(v).to(dev) for k, v in synth.synthetic_global_sample(147, 147).items()}
smp = {k: torch.stack([v] * B) for k, v in one.items()}
est = torch.from_numpy(synth.plausible_global_output(4096))[None].repeat(B, 1, 1).to(dev).requires_grad_(True)
gam = {k: getattr(a, "gamma_" + k)[-1] for k in ("color", "color_cons", "bndry_cons", "smthns", "smthns_cons", "bndry_loc", "depth")}


def step():
    est.grad = None
    loss = utils.global_loss(helper, dcal, est, smp["img_gt"], smp["img_gt"], smp["bndry_dist"], smp["deri"], smp["bndry_depth"], gam)
    loss.backward()
    return loss


step(); torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(reps):
    l = step()
torch.cuda.synchronize()
print(f"B={B}: global_loss forward + backward {(time.perf_counter() - t) / reps * 1e3:.3f} ms  loss {float(l):.9f}")
