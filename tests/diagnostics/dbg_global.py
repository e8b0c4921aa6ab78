import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")   # diagnostics live under tests/: they may import oracle/
sys.path[:0] = [ROOT, os.path.join(ROOT, "blurry-edges_amd")]
import numpy as np, torch
from be_hip import synth, train_global
import models, utils
from oracle import global_loss as ogl
DEV="cuda:0"
args = utils.get_args("global_train", argv=[]); args.batch_size = 1
local = models.LocalStage().to(DEV)
local.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synth.local_stage_state_dict().items()}); local.eval()
helper, dcal = utils.PostProcessGlobalBase(args, DEV), utils.DepthEtas(args, DEV)
data = train_global.make_dataset(1, DEV, local, helper)
torch.manual_seed(0)
model = models.GlobalStage(device=DEV).to(DEV)
for lyr in model.encoder.layers:
    lyr.dropout.p = lyr.dropout1.p = lyr.dropout2.p = 0.0; lyr.self_attn.dropout = 0.0
model.train()
batch = {k: data[0][k][None] for k in ("pm", "img_gt", "bndry_dist", "deri", "bndry_depth")}
print({k: (tuple(v.shape), bool(torch.isfinite(v).all())) for k, v in batch.items()})
est = model(batch["pm"]); est.retain_grad()
print("est finite", bool(torch.isfinite(est).all()), float(est.abs().max()))
loss = utils.global_loss(helper, dcal, est, batch["img_gt"], batch["img_gt"], batch["bndry_dist"], batch["deri"], batch["bndry_depth"], ogl.GAMMA_FINAL)
print("loss", float(loss))
loss.backward()
print("dest finite", bool(torch.isfinite(est.grad).all()), float(est.grad.abs().max()))
for k, p in model.named_parameters():
    if not torch.isfinite(p.grad).all(): print("NaN grad", k)
opt = torch.optim.AdamW(model.parameters(), lr=1e-4)
gn = torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
print("grad norm", float(gn))
opt.step()
bad = [k for k, p in model.named_parameters() if not torch.isfinite(p).all()]
print("non-finite params after step:", bad[:5], len(bad))
est = model(batch["pm"])
print("est2 finite", bool(torch.isfinite(est).all()))
if not torch.isfinite(est).all():
    # where does it start
    from be_hip import train_global_stage as tg
    t = [v.detach() for v in tg.parameter_list(model)]
    out, S = tg.forward_train(batch["pm"], model.positional_encoding.pe[0], 1, 0.0, 8, 1e-5, t)
    for i, (h, qkv, a, lse, v1, h1, f, v2, ws) in enumerate(S["layers"]):
        print(i, [bool(torch.isfinite(x).all()) for x in (h, qkv, a, lse, v1, h1, f, v2)])
loss2 = utils.global_loss(helper, dcal, est, batch["img_gt"], batch["img_gt"], batch["bndry_dist"], batch["deri"], batch["bndry_depth"], ogl.GAMMA_FINAL)
print("loss2", float(loss2), "terms", [float(v) for v in loss2.grad_fn.terms] if hasattr(loss2.grad_fn, "terms") else None)
print("est2 range", float(est.min()), float(est.max()))
