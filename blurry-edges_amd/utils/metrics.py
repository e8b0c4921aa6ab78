"""Depth metrics (utils/metrics.py:3-20 of the reference): delta1-3, RMSE (cm), AbsRel (cm)."""
import numpy as np


def eval_depth(pred, gt, msk, crop=0, tau_n=1.25, z_min=0.75, z_max=1.18):
    """numpy arrays [B,H,W] as in the reference; GPU tensors are reduced on the device (be_eval_depth_f32) and only the
    five numbers come back."""
    try:
        import torch
        if isinstance(pred, torch.Tensor) and pred.is_cuda:
            from be_hip import native
            f = lambda t: t.to(torch.float32).contiguous()
            r = native.eval_depth(f(pred), f(gt), f(msk), crop, tau_n, z_min, z_max).cpu().numpy()
            return tuple(float(v) for v in r)
    except ImportError:
        pass
    pred = np.clip(pred, z_min, z_max)
    if crop > 0:
        sl = (slice(None), slice(crop, -crop), slice(crop, -crop))
        pred, gt, msk = pred[sl], gt[sl], msk[sl]
    span = z_max - z_min
    pn = np.clip((pred - z_min) / span, 0, 1)
    gn = np.clip((gt - z_min) / span, 0, 1)
    ratio = np.maximum(gn / (pn + 1e-8), pn / (gn + 1e-8))
    cnt = np.sum(msk)
    deltas = [np.sum((ratio < tau_n ** e) * msk) / cnt for e in (1, 2, 3)]
    err = np.abs(gt - pred)
    rmse = np.sqrt(np.sum(err ** 2 * msk) / cnt)
    absrel = np.sum(err * msk / gt * msk) / cnt
    return deltas[0], deltas[1], deltas[2], rmse * 100, absrel * 100
