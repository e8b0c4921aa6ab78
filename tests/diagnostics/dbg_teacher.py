"""Teacher-forced per-step diagnostic of the LocalStage training step (HIP) against the fp64 and fp32 oracle.
Splits the comparison at the logits: (a) d loss / d est at the HIP logits, (b) CNN backward at EQUAL cotangent."""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")   # diagnostics live under tests/: they may import oracle/
sys.path[:0] = [ROOT, os.path.join(ROOT, "blurry-edges_amd")]
import numpy as np, torch
from be_hip import synth, train_local
import models, utils
from oracle import local_stage as ols, render as orr
DEV = "cuda:0"
B, steps = 64, int(sys.argv[1]) if len(sys.argv) > 1 else 20
data = synth.synthetic_training_patches(B * steps, seed=5)
args = utils.get_args("local_train", argv=[])
model = models.LocalStage().to(DEV)
model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synth.local_stage_state_dict().items()})
helper = utils.PostProcessLocalBase(args, DEV)
opt = torch.optim.AdamW(model.parameters(), lr=6e-5)
model.train()
names = [k for k, _ in model.named_parameters()]
zero_grad_names = {k for k in names if k.endswith(".0.bias") or k == "fc.1.bias"}    # a bias in front of a BatchNorm: exact gradient 0
gdata = {k: torch.from_numpy(v).to(DEV) for k, v in data.items()}
for it in range(steps):
    sd_cpu = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    b = {k: v[it * B:(it + 1) * B] for k, v in gdata.items()}
    # HIP forward / backward by hand (no clip, no step) to read est.grad
    est = model(b["img_ny"].permute(0, 3, 1, 2))
    est.retain_grad()
    opt.zero_grad(set_to_none=True)
    loss = utils.local_loss(helper, est, b["img_gt"], b["img_gt"], b["bndry_dist"], b["deri"], args.beta_bndry_loc, args.beta_smthns)
    loss.backward()
    gh = {k: p.grad.detach().cpu().double() for k, p in model.named_parameters()}
    dest_h = est.grad.detach().cpu().double()
    norm_h = float(torch.sqrt(sum((g ** 2).sum() for g in gh.values())))
    line = [f"step {it:2d} loss {float(loss):.6f} |g| hip {norm_h:.5f}"]
    for dt, tag in ((torch.float64, "f64"), (torch.float32, "f32")):
        sdd = {k: (v.to(dt) if v.is_floating_point() else v) for k, v in sd_cpu.items()}
        P = [sdd[k].requires_grad_(True) for k in names]
        cb = {k: torch.from_numpy(v[it * B:(it + 1) * B]).to(dt) for k, v in data.items()}
        esto = ols.local_stage_forward(sdd, cb["img_ny"].permute(0, 3, 1, 2), training=True)
        lo, _, _ = orr.local_loss(esto, cb["img_gt"], cb["img_gt"], cb["bndry_dist"], cb["deri"], args.beta_bndry_loc, args.beta_smthns, inverse="solve")
        dest_o, = torch.autograd.grad(lo, esto, retain_graph=True)
        go = torch.autograd.grad(lo, P, retain_graph=True)
        norm_o = float(torch.sqrt(sum((g.double() ** 2).sum() for g in go)))
        # (a) loss gradient at the HIP logits
        e_same = est.detach().cpu().to(dt).requires_grad_(True)
        l2, _, _ = orr.local_loss(e_same, cb["img_gt"], cb["img_gt"], cb["bndry_dist"], cb["deri"], args.beta_bndry_loc, args.beta_smthns, inverse="solve")
        dest_same, = torch.autograd.grad(l2, e_same)
        # (b) CNN backward at EQUAL cotangent (the HIP d loss / d est pushed through the oracle CNN)
        gb = torch.autograd.grad(esto, P, grad_outputs=dest_h.to(dt))
        eb = max((float((gh[k] - g.double()).norm() / g.double().norm()), k) for k, g in zip(names, gb) if k not in zero_grad_names)
        ee = max((float((gh[k] - g.double()).norm() / g.double().norm()), k) for k, g in zip(names, go) if k not in zero_grad_names)
        zb = max(float(gh[k].norm()) for k in zero_grad_names)
        line.append(f"\n   {tag}: |g| {norm_o:.5f} (hip/ora-1 {norm_h / norm_o - 1:+.2e})  dest: same-input {float((dest_h - dest_same.double()).norm() / dest_same.double().norm()):.2e} "
                    f"end-to-end {float((dest_h - dest_o.double()).norm() / dest_o.double().norm()):.2e}  logits {float((est.detach().cpu().double() - esto.detach().double()).abs().max()):.1e}"
                    f"\n        params: equal-cotangent worst {eb[0]:.2e} ({eb[1]})  end-to-end worst {ee[0]:.2e} ({ee[1]})  |g| of the zero-gradient biases (hip) {zb:.1e}")
        if tag == "f64":
            i = int((dest_h - dest_o.double()).norm(dim=1).argmax())
            line.append(f"\n        worst patch {i}: est {[round(v, 3) for v in esto[i].tolist()]}\n          hip {[f'{v:.2e}' for v in dest_h[i].tolist()]}\n          ora {[f'{v:.2e}' for v in dest_o[i].tolist()]}")
    print("".join(line), flush=True)
    torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
    opt.step()
