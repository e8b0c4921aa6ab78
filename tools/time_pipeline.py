#!/usr/bin/env python3
"""Stage timings of DepthPipeline on one 147x147 pair and one 587x587 pair, and of the training step."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "blurry-edges_amd")]
import numpy as np, torch
from be_hip import synth, native
from be_hip.pipeline import DepthPipeline
import models, utils
dev = "cuda:0"
a = utils.get_args("eval", argv=[])
lm = models.LocalStage(); lm.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synth.local_stage_state_dict().items()})
gm = models.GlobalStage(device=dev); gm.load_state_dict({k: torch.from_numpy(v) for k, v in synth.global_stage_state_dict().items()})
pipe = DepthPipeline(lm.to(dev).eval(), gm.to(dev).eval(), utils.PostProcessGlobalBase(a, dev), utils.DepthEtas(a, dev))
img = torch.from_numpy(synth.synthetic_image_pair(147, 147)[0]).to(dev)

def timed(f, n=10):
    f(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): r = f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3, r

with torch.no_grad():
    t_all, out = timed(lambda: pipe(img))
    t_loc, (pat, est10, col, pm) = timed(lambda: pipe.local_pass(img))
    t_cnn, _ = timed(lambda: pipe.local.forward_image_pair(img))
    t_glb, est12 = timed(lambda: pipe.global_pass(pm))
    t_rec, (rec, _) = timed(lambda: pipe.records(est12, img))
    t_fold, _ = timed(lambda: native.fold_records(pipe.helper.render_opts(False), rec, 64, 64, 147, 147))
print(f"147x147 pair: total {t_all:.2f} ms  = local pass {t_loc:.2f} (CNN {t_cnn:.2f}) + global stage {t_glb:.2f} + records {t_rec:.3f} + fold {t_fold:.3f}"
      f"  -> {4096 / t_all * 1e3:.0f} pairs/s")
big = torch.from_numpy(synth.synthetic_image_pair(587, 587, nshape=14)[0]).to(dev)
with torch.no_grad():
    t_big, _ = timed(lambda: pipe.run_big(big), n=2)
print(f"587x587 pair (36 blocks): {t_big:.1f} ms -> {36 * 4096 / t_big * 1e3:.0f} pairs/s")

import models
unet = models.DepthCompletion()
unet.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synth.unet_state_dict().items()})
unet = unet.to(dev).eval()
dm = torch.from_numpy(synth.sparse_depth_map()).to(dev)
t_u, _ = timed(lambda: unet(dm))
dm8 = dm.expand(8, 1, 147, 147).contiguous()
t_u8, _ = timed(lambda: unet(dm8))
print(f"DepthCompletion U-Net 147x147: {t_u:.2f} ms (batch 1), {t_u8:.2f} ms (batch 8)  [30.3 GFLOP per map]")
