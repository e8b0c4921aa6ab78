"""Max-pool windows whose HIP winner differs from the float64 oracle's (test helper, shared by the training parity tests).

nn.MaxPool2d routes a window's whole gradient to ONE element.  Where the two best candidates of a window are a near-tie in float64
(1e-7 of the map's scale happens about once per 64-patch batch), which one wins is decided by the LAST BIT of the float32 forward: any
rounding-level change in any forward kernel - a different K split, another tile shape - can hand the window to the other element, and
the gradients of everything upstream of that pool move by 1e-3 ... 1e-2 of their scale.  The reference's own float32 run shows the same
events against its float64 run.  The tests therefore (a) demonstrate the tie - a window whose winner differs from float64's AND whose
two candidates are within 1e-5 of the map's scale in float64 - and (b) hold the tensors upstream of such a pool to the event bound,
everything else to the plain bound."""
import torch
import torch.nn.functional as F

POOLS = (("pool1", "conv1", 3, 2, 1), ("pool2", "layer0", 3, 2, 1), ("pool3", "layer3", 2, 2, 0))
# parameter-name prefixes whose gradients pass through each pool's backward
UPSTREAM = {"pool1": ("conv1.",), "pool2": ("conv1.", "layer0."), "pool3": ("conv1.", "layer0.", "layer1.", "layer2.", "layer3.")}


def pool_winner_flips(state, x_gpu, dev="cuda:0"):
    """HIP train-mode forward of `state` on x_gpu against the float64 oracle: {pool: (windows whose winner differs, largest float64 gap
    between the two candidates relative to the map's largest value)}."""
    import models
    from be_hip import train
    from oracle import local_stage as ols
    probe = models.LocalStage().to(dev)
    probe.load_state_dict(state)
    _, S = train.forward_train(x_gpu.to(torch.float32).contiguous(), [v.detach() for v in probe._tensor_list()])
    taps = {}
    sdd = {k: (v.detach().cpu().double() if v.is_floating_point() else v.detach().cpu()) for k, v in state.items()}
    with torch.no_grad():
        ols.local_stage_forward(sdd, x_gpu.cpu().double(), training=True, taps=taps)
    out = {}
    for pool, src, k, st, pd in POOLS:
        v = taps[src]                                                    # [n,c,h,w] float64
        w_ = v.shape[3]
        _, oi = F.max_pool2d(v, k, st, pd, return_indices=True)          # flat y*w + x per [n,c,oh,ow]
        idx = S[pool][0].permute(0, 3, 1, 2).cpu().long()               # HIP: dy*k + dx of the winner, -> [n,c,oh,ow]
        oh, ow = idx.shape[2], idx.shape[3]
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
        out[pool] = (int(diff.sum()), gap)
    return out


def tied_upstream(flips, tie=1e-5):
    """parameter-name prefixes upstream of a pool with a demonstrated near-tie flip"""
    pre = set()
    for pool, (cnt, gap) in flips.items():
        if cnt > 0 and gap <= tie:
            pre.update(UPSTREAM[pool])
    return tuple(sorted(pre))
