"""Oracle: analytic depth-from-defocus solve (TEST INFRASTRUCTURE).

Restates utils/depth_etas.py:3-37 of the reference.  Constants are produced exactly the way the
reference produces them (python float64 for the rational terms, float32 tensor ops for the
intercept and the two angles) so that float32 branch decisions agree bit for bit.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import torch

CAM_DEFAULT = dict(s=0.1104, rho_1=10.0, rho_2=10.2, sigma_cam=0.003, pixel_pitch=5.86e-6)  # utils/args.py:14


@dataclass
class DepthConsts:
    s: float
    numerator: float
    den_const: float
    k: float            # denominator_factor_root
    k2: float           # denominator_factor
    intercept: float    # float32-rounded value, held as python float
    sin_w: float
    cos_w: float
    sin_m: float
    cos_m: float


def depth_consts(cam=None, R=21, mag=4) -> DepthConsts:
    cam = cam or CAM_DEFAULT
    s, r1, r2 = cam["s"], cam["rho_1"], cam["rho_2"]
    nf = R // 2                                                    # depth_etas.py:12
    num = 2 * s ** 2 * (r2 - r1)                                   # :13
    den_c = -s * (r1 - r2) * (r1 * s + r2 * s - 2)                 # :14
    k = nf * cam["pixel_pitch"] * mag / cam["sigma_cam"]           # :15
    # :18 -- float32 tensor arithmetic, op by op
    icpt = torch.abs(torch.tensor(s * (r2 - r1))) * cam["sigma_cam"] / cam["pixel_pitch"] / mag / nf
    th_m = torch.tensor(3 / 4 * math.pi)                           # :20
    th_w = torch.tensor(1 / 4 * math.pi)                           # :21
    return DepthConsts(s=s, numerator=num, den_const=den_c, k=k, k2=k ** 2, intercept=float(icpt),
                       sin_w=float(torch.sin(th_w)), cos_w=float(torch.cos(th_w)),
                       sin_m=float(torch.sin(th_m)), cos_m=float(torch.cos(th_m)))


def etas2depth(c: DepthConsts, eta1: torch.Tensor, eta2: torch.Tensor, return_branch=False):
    """depth_etas.py:23-34.  Works in the dtype of eta1 (float32 follows the reference's op order)."""
    dt = eta1.dtype
    I = torch.tensor(c.intercept, dtype=dt)
    sw, cw = torch.tensor(c.sin_w, dtype=dt), torch.tensor(c.cos_w, dtype=dt)
    sm, cm = torch.tensor(c.sin_m, dtype=dt), torch.tensor(c.cos_m, dtype=dt)
    c1 = -sw * eta1 + cw * (eta2 - I)
    c2 = -sm * (eta1 - I) + cm * eta2
    c3 = -sw * (eta1 - I) + cw * eta2
    b1, b2, b3 = c1 > 0, c2 > 0, c3 < 0
    e11 = torch.where(b1, (eta1 + eta2 - I) / 2,
                      torch.where(b2, I + (eta1 - eta2 - I) / 2,
                                  torch.where(b3, I + (eta1 + eta2 - I) / 2, eta1)))
    e22 = torch.where(b1, I + (eta1 + eta2 - I) / 2,
                      torch.where(b2, (eta2 - eta1 + I) / 2,
                                  torch.where(b3, (eta1 + eta2 - I) / 2, eta2)))
    z = c.numerator / (c.k2 * (e11 ** 2 - e22 ** 2) + c.den_const)
    if return_branch:
        br = torch.where(b1, 0, torch.where(b2, 1, torch.where(b3, 2, 3))).to(torch.int32)
        return z, br
    return z


def depth2sigma(c: DepthConsts, depth: torch.Tensor, rho_prime: float):
    """depth_etas.py:36-37."""
    return torch.abs((1 / depth - rho_prime) * c.s + 1) / c.k
