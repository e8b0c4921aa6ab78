#!/usr/bin/env python3
"""configs[0]-sized runs on the GPU: LocalStage + pass-A colours + depth for 1 / 32 / 256 pairs (launch-latency regime)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "blurry-edges_amd")]
import numpy as np, torch
from be_hip import native, synth
import models, utils
dev = "cuda:0"
m = models.LocalStage(); m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synth.local_stage_state_dict().items()}); m = m.to(dev).eval()
helper = utils.PostProcessLocalBase(utils.get_args("local_train", argv=[]), dev)
dcal = utils.DepthEtas(utils.get_args("eval", argv=[]), dev)
opts = native.RenderOpts.from_buffer_copy(helper._opts); opts.wrap_angles = 1
for pairs in (1, 32, 256):
    x_np, _ = synth.synthetic_patch_pairs(pairs, seed=synth.SEED_DEFAULT)
    x = torch.from_numpy(x_np).to(dev)
    col = torch.empty(2 * pairs, 3, 3, device=dev); dep = torch.empty(pairs, 2, device=dev)
    def step():
        with torch.no_grad():
            est = m(x); native.render_colors(opts, est, x, colors=col); native.local_depth(dcal.consts, est, out=dep)
    for _ in range(5): step()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(50): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 50
    print(f"{pairs:4d} pair(s): {dt * 1e3:.3f} ms per call  ({pairs / dt:.0f} pairs/s)")
