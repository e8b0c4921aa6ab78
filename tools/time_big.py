#!/usr/bin/env python3
"""587x587 pair: the reference's block schedule (36 local passes) against the de-duplicated local pass, stage by stage."""
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

def timed(f, n=3):
    f(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): r = f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3, r

big = torch.from_numpy(synth.synthetic_image_pair(587, 587, nshape=24)[0]).to(dev)
with torch.no_grad():
    for streams in (2, 1):
        pipe.local.streams = streams
        t_blk, _ = timed(lambda: pipe.run_big(big, dedup=False))
        t_dd, _ = timed(lambda: pipe.run_big(big))
        t_loc, (_, _, _, pm) = timed(lambda: pipe.local_pass(big))
        print(f"streams {streams}: block schedule {t_blk:.1f} ms, de-duplicated {t_dd:.1f} ms (local pass over the whole grid {t_loc:.1f} ms)")
    feats = torch.stack([pm.view(284, 284, 38)[:64, :64].reshape(4096, 38)] * 12)
    t_g, y = timed(lambda: pipe.globl(feats))
    est12 = native.global_denorm(y[0])
    t_r, _ = timed(lambda: pipe.records(est12, big, window=(0, 0, 147, 147)), n=10)
    print(f"GlobalStage on 12 blocks {t_g:.2f} ms (x3 per image), pass-B records per block {t_r:.3f} ms (x36)")
