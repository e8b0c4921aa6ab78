"""Can the GlobalStage training step drive the DEPTH term down?  Overfit one batch of 4 synthetic images with the final gammas and
print the seven un-weighted terms (be_global_loss partial sums) every 25 steps."""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path[:0] = [ROOT, os.path.join(ROOT, "blurry-edges_amd")]
import numpy as np, torch
from be_hip import synth, train_global
import models, utils
DEV = "cuda:0"
args = utils.get_args("global_train", argv=[]); args.batch_size = 4
local = models.LocalStage().to(DEV)
local.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synth.local_stage_state_dict().items()}); local.eval()
helper, dcal = utils.PostProcessGlobalBase(args, DEV), utils.DepthEtas(args, DEV)
data = train_global.make_dataset(4, DEV, local, helper)
batch = {k: torch.stack([d[k] for d in data]) for k in ("pm", "img_gt", "bndry_dist", "deri", "bndry_depth")}
torch.manual_seed(0)
model = models.GlobalStage(device=DEV).to(DEV)
for p in model.parameters():
    if p.dim() > 1: torch.nn.init.xavier_normal_(p)
opt = torch.optim.AdamW(model.parameters(), lr=args.learning_rate)
gam = train_global.GammaSchedule(args).final()
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
B, P = 4, 4096
for it in range(steps + 1):
    model.train()
    est = model(batch["pm"])
    opt.zero_grad(set_to_none=True)
    loss = utils.global_loss(helper, dcal, est, batch["img_gt"], batch["img_gt"], batch["bndry_dist"], batch["deri"], batch["bndry_depth"], gam, empty_mask="zero")
    t = loss.grad_fn.terms
    if it % 25 == 0:
        n1, n3, n4 = B * 2 * 441 * P, B * 441 * P, B * 2 * 361 * P
        terms = [float(t[0]) / n1, float(t[1]) / n1, float(t[2]) / n3, float(t[3]) / n4, float(t[4]) / n4, float(t[5]) / n3, float(t[6]) / max(float(t[7]), 1)]
        print(f"step {it:4d} loss {float(loss):.6f}  color {terms[0]:.5f} ccons {terms[1]:.5f} bcons {terms[2]:.5f} smth {terms[3]:.5f} scons {terms[4]:.5f} bloc {terms[5]:.5f} DEPTH {terms[6]:.5f} (mask {int(t[7])})", flush=True)
    loss.backward()
    torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
    opt.step()
