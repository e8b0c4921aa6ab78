"""DepthEtas behind the reference's interface (utils/depth_etas.py:3-37), computed by HIP kernels.

Constructor arguments, attribute names and method signatures are the reference's.  The constants are
derived the same way (python float64 for the rational terms; float32 tensor arithmetic, operation by
operation, for the intercept and the two angles) and handed to the kernels as one struct, so the three
half-plane tests of etas2depth pick the same branch as the reference does on identical inputs.
"""
import math

import torch

from be_hip import autograd_ops as ag
from be_hip import native


class DepthEtas:
    def __init__(self, args, device):
        cam = args.cam_params
        self.s = cam['s']
        self.device = device
        rho_1, rho_2 = cam['rho_1'], cam['rho_2']
        norm_factor = args.R // 2
        self.numerator = 2 * self.s ** 2 * (rho_2 - rho_1)
        self.denominator_constant = -self.s * (rho_1 - rho_2) * (rho_1 * self.s + rho_2 * self.s - 2)
        self.denominator_factor_root = norm_factor * cam['pixel_pitch'] * args.mag / cam['sigma_cam']
        self.denominator_factor = self.denominator_factor_root ** 2
        # fp32, one rounding per operation, as the reference's tensor expression evaluates it
        icpt = torch.abs(torch.tensor(self.s * (rho_2 - rho_1))) * cam['sigma_cam'] / cam['pixel_pitch'] / args.mag / norm_factor
        th_mid = torch.tensor(3 / 4 * math.pi)
        th_wng = torch.tensor(1 / 4 * math.pi)
        self._consts = native.DepthConsts(
            s=self.s, numerator=self.numerator, den_const=self.denominator_constant,
            k=self.denominator_factor_root, k2=self.denominator_factor, intercept=float(icpt),
            sin_w=float(torch.sin(th_wng)), cos_w=float(torch.cos(th_wng)),
            sin_m=float(torch.sin(th_mid)), cos_m=float(torch.cos(th_mid)))
        self.intercept = icpt.to(device)
        self.theta_mid = th_mid.to(device)
        self.theta_wng = th_wng.to(device)

    @property
    def consts(self):
        return self._consts

    def etas2depth(self, eta1, eta2):
        return ag.Etas2Depth.apply(eta1, eta2, self._consts)            # differentiable (global_training.py:91-92)

    def depth2sigma(self, depth, rho_prime):
        return ag.Depth2Sigma.apply(depth, self._consts, rho_prime)
