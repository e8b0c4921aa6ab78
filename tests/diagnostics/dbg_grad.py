import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")   # diagnostics live under tests/: they may import oracle/
sys.path[:0] = [ROOT, os.path.join(ROOT, "blurry-edges_amd")]
import numpy as np, torch
from be_hip import synth
import models, utils
from oracle import local_stage as ols, render as orr
DEV = "cuda:0"
B = 64
data = synth.synthetic_training_patches(B * 6, seed=5)
args = utils.get_args("local_train", argv=[])
model = models.LocalStage().to(DEV)
model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synth.local_stage_state_dict().items()})
helper = utils.PostProcessLocalBase(args, DEV)
model.train()
b = {k: torch.from_numpy(v[:B]).to(DEV) for k, v in data.items()}
est = model(b["img_ny"].permute(0, 3, 1, 2))
est.retain_grad()
loss = utils.local_loss(helper, est, b["img_gt"], b["img_gt"], b["bndry_dist"], b["deri"], args.beta_bndry_loc, args.beta_smthns)
loss.backward()
sd = ols.to_torch_sd(synth.local_stage_state_dict())
names = [k for k, v in sd.items() if v.is_floating_point() and "running_" not in k]
for k in names: sd[k].requires_grad_(True)
c = {k: torch.from_numpy(v[:B]) for k, v in data.items()}
for dt, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
    sdd = {k: (v.detach().to(dt).requires_grad_(v.requires_grad) if v.is_floating_point() else v) for k, v in sd.items()}
    esto = ols.local_stage_forward(sdd, c["img_ny"].permute(0, 3, 1, 2).to(dt), training=True)
    esto.retain_grad()
    lo, _, _ = orr.local_loss(esto, c["img_gt"].to(dt), c["img_gt"].to(dt), c["bndry_dist"].to(dt), c["deri"].to(dt),
                              args.beta_bndry_loc, args.beta_smthns, inverse="solve")
    lo.backward()
    print(tag, "loss", float(loss), float(lo))
    # the loss kernel alone: oracle gradient AT THE HIP logits (same input point)
    e_same = est.detach().cpu().to(dt).requires_grad_(True)
    l2, _, _ = orr.local_loss(e_same, c["img_gt"].to(dt), c["img_gt"].to(dt), c["bndry_dist"].to(dt), c["deri"].to(dt),
                              args.beta_bndry_loc, args.beta_smthns, inverse="solve")
    l2.backward()
    print(tag, "SAME-INPUT d loss/d est rel err", float((est.grad.cpu() - e_same.grad.float()).abs().max() / e_same.grad.abs().max()),
          " logits hip-vs-oracle", float((est.detach().cpu() - esto.detach().float()).abs().max()))
    print(tag, "d loss/d est rel err", float((est.grad.cpu() - esto.grad.float()).abs().max() / esto.grad.abs().max()))
    hp = dict(model.named_parameters())
    worst = []
    for k in names:
        go = sdd[k].grad.float(); gh = hp[k].grad.cpu()
        e = float((gh - go).abs().max() / max(float(go.abs().max()), 1e-30))
        worst.append((e, k, float(go.abs().max())))
    worst = [w for w in worst if not (w[1].endswith(".0.bias") or w[1] == "fc.1.bias")]
    worst.sort(reverse=True)
    for e, k, m in worst[:5]: print(tag, f"{k:32s} rel err {e:.2e}  |g|max {m:.2e}")
    d = (est.grad.cpu() - esto.grad.float()).abs()
    print(tag, "per-column max abs err", [f"{v:.1e}" for v in d.max(dim=0).values.tolist()])
    print(tag, "per-column max |g|    ", [f"{v:.1e}" for v in esto.grad.float().abs().max(dim=0).values.tolist()])
    i = int(d.max(dim=1).values.argmax()); print(tag, "worst patch", i, "est", [f"{v:.3f}" for v in esto[i].tolist()])
    print(tag, " hip grad", [f"{v:.3e}" for v in est.grad[i].tolist()]); print(tag, " ora grad", [f"{v:.3e}" for v in esto.grad[i].tolist()])
