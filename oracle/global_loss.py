"""Oracle: GlobalLoss of the global-stage training (TEST INFRASTRUCTURE).

Restates global_training.py:62-157 in the flat one-row-per-patch layout with torch autograd.  Note the
reference DETACHES the folded global image / boundary map before the consistency terms (:94,:100,:106), so no
gradient flows through the fold: every term is local to a patch once the folded maps are known.
"""
from __future__ import annotations

import torch

from . import depth as odepth, render as orr, tiling as ot

GAMMA_FINAL = dict(color=0.1, color_cons=0.05, bndry_cons=0.02, smthns=0.002, smthns_cons=0.002, bndry_loc=0.0001,
                   depth=0.5)                                           # utils/args.py:53-59, last entries


def restore_params(est):
    """global_training.py:141-145: raw GlobalStage output [P,12] -> (params8 [P,8], etas [P,4])."""
    xy = est[:, :4] * 3
    ang = torch.remainder((est[:, 4:8] + 1) * torch.pi, 2 * torch.pi)
    return torch.cat([xy, ang], dim=1), orr.params2etas(est[:, 8:] + 0.5)


def global_loss(c, est, img_fit, img_gt, bndry_dist, deri, bndry_depth, gamma=None, stride=2, inverse="cayley"):
    """est [B,P,12]; img_fit / img_gt [B,2,H,W,3]; bndry_dist [B,H,W]; deri [B,2,H-2,W-2,3]; bndry_depth [B,H,W].
    Returns (loss, dict of the seven un-weighted terms)."""
    g = dict(GAMMA_FINAL if gamma is None else gamma)
    B, P = est.shape[:2]
    H, W = img_gt.shape[2:4]
    t = dict(color=0., color_cons=0., bndry_cons=0., smthns=0., smthns_cons=0., bndry_loc=0.)
    dnum, dden = 0., 0.
    for b in range(B):
        p8, etas = restore_params(est[b])
        dists = orr.params2dists(p8)
        w1 = orr.dists2indicators(dists, etas[:, 0:2])
        w2 = orr.dists2indicators(dists, etas[:, 2:4])
        fit = ot.unfold_patches(img_fit[b].permute(0, 3, 1, 2), stride)            # [2,P,3,21,21]
        gtp = ot.unfold_patches(img_gt[b].permute(0, 3, 1, 2), stride)
        colors, _, _ = orr.ridge_colors([w1, w2], [fit[0], fit[1]], inverse=inverse)
        pat = torch.stack([orr.composite(w1, colors), orr.composite(w2, colors)])  # [2,P,3,21,21]
        bnd = orr.boundary_map(dists)                                              # [P,21,21]
        G = ot.fold_mean(pat, H, W, stride).detach()                               # [2,3,H,W]       :154
        Gb = ot.fold_mean(bnd[None, :, None], H, W, stride).detach()               # [1,1,H,W]       :155
        t["color"] = t["color"] + ((gtp - pat) ** 2).sum(2).sum()
        t["color_cons"] = t["color_cons"] + ((pat - ot.unfold_patches(G, stride)) ** 2).sum(2).sum()
        t["bndry_cons"] = t["bndry_cons"] + ((bnd - ot.unfold_patches(Gb, stride)[0, :, 0]) ** 2).sum()
        pd = orr.image_derivative(pat.reshape(2 * P, 3, 21, 21)).reshape(2, P, 3, 19, 19)
        dgt = ot.unfold_patches(deri[b].permute(0, 3, 1, 2), stride, r=19)
        dG = ot.unfold_patches(orr.image_derivative(G), stride, r=19)
        t["smthns"] = t["smthns"] + ((pd - dgt) ** 2).sum(2).sum()
        t["smthns_cons"] = t["smthns_cons"] + ((pd - dG) ** 2).sum(2).sum()
        lb = ot.unfold_patches(torch.log2(bndry_dist[b] + 1)[None, None], stride)[0, :, 0]
        t["bndry_loc"] = t["bndry_loc"] + ((lb * bnd) ** 2).sum()
        mk = orr.depth_mask(dists)
        z1 = odepth.etas2depth(c, etas[:, 0], etas[:, 2])
        z2 = odepth.etas2depth(c, etas[:, 1], etas[:, 3])
        zmap = torch.where(mk == 1, z1[:, None, None], torch.where(mk == 2, z2[:, None, None], mk.to(z1.dtype)))
        bdp = ot.unfold_patches(bndry_depth[b][None, None], stride)[0, :, 0]
        m = ((bdp != 0) & (mk != 0)).to(z1.dtype)
        dnum = dnum + (((zmap - bdp) * m) ** 2).sum()
        dden = dden + m.sum()
    n1, n3, n4 = B * 2 * 441 * P, B * 441 * P, B * 2 * 361 * P
    terms = dict(color=t["color"] / n1, color_cons=t["color_cons"] / n1, bndry_cons=t["bndry_cons"] / n3,
                 smthns=t["smthns"] / n4, smthns_cons=t["smthns_cons"] / n4, bndry_loc=t["bndry_loc"] / n3,
                 depth=dnum / dden)
    loss = sum(g[k] * terms[k] for k in terms)
    return loss, terms
