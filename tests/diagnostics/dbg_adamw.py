"""Where does be_hip.optim.ClipAdamW differ from clip_grad_norm_ + torch.optim.AdamW?  Prints the worst element per step."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "blurry-edges_amd")]
import numpy as np, torch
import models
from be_hip.optim import ClipAdamW
DEV = "cuda:0"
torch.manual_seed(3)
ma, mb = models.LocalStage().to(DEV), models.LocalStage().to(DEV)
mb.load_state_dict(ma.state_dict())
pa, pb = list(ma.parameters()), list(mb.parameters())
oa, ob = ClipAdamW(pa, lr=1e-3), torch.optim.AdamW(pb, lr=1e-3)
n = sum(p.numel() for p in pa)
for it, scale in enumerate((3e-3, 1e-5, 5e-4)):
    flat = torch.randn(n, device=DEV) * scale
    off = 0
    for a, b in zip(pa, pb):
        a.grad = flat[off:off + a.numel()].view_as(a); b.grad = a.grad.clone(); off += a.numel()
    before = [b.detach().clone() for b in pb]
    m0 = [ob.state[b]["exp_avg"].clone() if b in ob.state and "exp_avg" in ob.state[b] else torch.zeros_like(b) for b in pb]
    v0 = [ob.state[b]["exp_avg_sq"].clone() if b in ob.state and "exp_avg_sq" in ob.state[b] else torch.zeros_like(b) for b in pb]
    # teacher forcing: the HIP optimizer starts every step from the stock optimizer's state
    with torch.no_grad():
        for a, b, m_, v_ in zip(pa, pb, m0, v0):
            a.copy_(b); oa.state[a]["exp_avg"].copy_(m_); oa.state[a]["exp_avg_sq"].copy_(v_)
        oa._step.fill_(float(it))
    nb = torch.nn.utils.clip_grad_norm_(pb, 1.0); ob.step()
    na = oa.clip_and_step(1.0)
    worst = (0, None)
    for k, (a, b, b0) in enumerate(zip(pa, pb, before)):
        d = (a.detach().double() - b.detach().double()).abs()
        tol = torch.from_numpy(np.spacing(np.abs(b.detach().cpu().numpy()).astype(np.float32))).double().to(DEV) + 5e-7 * (b.detach().double() - b0.double()).abs() + 1e-12
        r = d / tol
        j = int(r.argmax())
        if float(r.flatten()[j]) > worst[0]:
            worst = (float(r.flatten()[j]), (k, j))
    k, j = worst[1]
    f = lambda t: float(t.detach().flatten()[j])
    sa, sb = oa.state[pa[k]], ob.state[pb[k]]
    print(f"step {it}: norm {float(na):.9g} / {float(nb):.9g}; worst {worst[0]:.2f} at tensor {k} elem {j}: p0 {f(before[k]):.9g} g {f(pb[k].grad):.9g} m0 {f(m0[k]):.9g} v0 {f(v0[k]):.9g}\n"
          f"    hip p {f(pa[k]):.9g} m {f(sa['exp_avg']):.9g} v {f(sa['exp_avg_sq']):.9g}   torch p {f(pb[k]):.9g} m {f(sb['exp_avg']):.9g} v {f(sb['exp_avg_sq']):.9g}")
