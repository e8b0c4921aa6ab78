"""Diagnostic (GPU box): the train-mode forward / backward of golden batch g2 against the reference's gradients, tensor by tensor, and
the max-pool windows whose HIP winner differs from the float64 oracle's (tests/pool_flips.py explains the near-ties).  Run it with
BE_NO_TRAIN_SK=1 / BE_NO_TRAIN_SK_FWD=1 to see which forward flips which window.  usage: python tools/g2_probe.py"""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [ROOT, os.path.join(ROOT, "blurry-edges_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
from be_hip import synth
from conftest import load_golden, relmax
import models
g = load_golden("g2_local_stage_train")
m = models.LocalStage()
m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synth.local_stage_state_dict().items()})
m = m.to("cuda:0").train()
T = lambda a: torch.from_numpy(np.asarray(a)).float()
x = T(synth.uniform_patches(64, name="train_patches")).to("cuda:0")
ct = T(synth.f32(synth.hash_normal(synth.SEED_DEFAULT, "train_cotangent", (64, 10)))).to("cuda:0")
y = m(x)
print("logits", relmax(y.detach().cpu(), g["logits"]))
(y * ct).sum().backward()
params = dict(m.named_parameters())
errs = []
for k in g:
    if k.startswith("grad_") and k != "grad_x_sub":
        name = k[len("grad_"):]
        if name.endswith(".0.bias") or name == "fc.1.bias": continue
        errs.append((relmax(params[name].grad.cpu(), g[k]), name))
errs.sort(reverse=True)
print(" ".join("%s:%.1e" % (n, e) for e, n in errs[:8]))
for name in ("layer2.0.conv2.0.weight", "fc.1.weight"):
    gr = params[name].grad.flatten()
    print(name, relmax(gr[::997].cpu(), g["gradsub_" + name]), abs(float(gr.double().norm()) - float(g["gradnorm_" + name])) / float(g["gradnorm_" + name]))
# ---- pool windows whose HIP winner differs from the float64 oracle's, and the float64 gap between the two candidates
import torch.nn.functional as F
from be_hip import train
from oracle import local_stage as ols
state = {k: v.detach().clone() for k, v in m.state_dict().items()}
probe = models.LocalStage().to("cuda:0"); probe.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synth.local_stage_state_dict().items()})
_, S = train.forward_train(x.contiguous(), [v.detach() for v in probe._tensor_list()])
taps = {}
sdd = {k: (torch.from_numpy(np.asarray(v)).double() if np.asarray(v).dtype.kind == "f" else torch.from_numpy(np.asarray(v))) for k, v in synth.local_stage_state_dict().items()}
with torch.no_grad():
    ols.local_stage_forward(sdd, x.cpu().double(), training=True, taps=taps)
for pool, src, k, st, pd in (("pool1", "conv1", 3, 2, 1), ("pool2", "layer0", 3, 2, 1), ("pool3", "layer3", 2, 2, 0)):
    v = taps[src]
    _, oi = F.max_pool2d(v, k, st, pd, return_indices=True)
    idx = S[pool][0].permute(0, 3, 1, 2).cpu().long()
    oh, ow = idx.shape[2], idx.shape[3]
    w_ = v.shape[3]
    oy = torch.arange(oh).view(1, 1, oh, 1) * st - pd
    ox = torch.arange(ow).view(1, 1, 1, ow) * st - pd
    hi = (oy + idx // k) * w_ + (ox + idx % k)
    diff = hi != oi
    gap = 0.0
    if diff.any():
        flat = v.flatten(2)
        a_ = torch.gather(flat, 2, oi.flatten(2)).view_as(oi)[diff]
        b_ = torch.gather(flat, 2, hi.flatten(2)).view_as(hi)[diff]
        gap = float((a_ - b_).abs().max() / v.abs().max())
    print(pool, "windows with another winner than float64:", int(diff.sum()), "largest float64 gap / map scale: %.2e" % gap)
