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
        if not (eta1.is_cuda or eta2.is_cuda):
            return self._etas2depth_cpu(eta1, eta2)
        return ag.Etas2Depth.apply(eta1, eta2, self._consts)            # differentiable (global_training.py:91-92)

    def depth2sigma(self, depth, rho_prime):
        if not depth.is_cuda:
            return torch.abs((1 / depth - rho_prime) * self.s + 1) / self.denominator_factor_root
        return ag.Depth2Sigma.apply(depth, self._consts, rho_prime)

    def _etas2depth_cpu(self, e1, e2):
        """CPU tensors (BASELINE configs[0], "PyTorch-CPU, plumbing"): the same solve as a torch expression, with the reference's
        operation order (utils/depth_etas.py:23-34) so that fp32 results are the reference's bit for bit (golden g5).  The point
        (e1, e2) is projected orthogonally onto the segment of the valid locus {e2 - e1 = I, e1 + e2 = I, e1 - e2 = I} that the three
        half-plane tests select; a point beyond all three keeps its raw values."""
        c = self._consts
        I = self.intercept.to(e1.device)
        above = -c.sin_w * e1 + c.cos_w * (e2 - I) > 0                  # beyond the upper wing  e2 - e1 = I
        middle = -c.sin_m * (e1 - I) + c.cos_m * e2 > 0                 # below the middle piece e1 + e2 = I
        below = -c.sin_w * (e1 - I) + c.cos_w * e2 < 0                  # beyond the lower wing  e1 - e2 = I
        half_sum = (e1 + e2 - I) / 2
        p1 = torch.where(above, half_sum, torch.where(middle, I + (e1 - e2 - I) / 2, torch.where(below, I + half_sum, e1)))
        p2 = torch.where(above, I + half_sum, torch.where(middle, (e2 - e1 + I) / 2, torch.where(below, half_sum, e2)))
        return self.numerator / (self.denominator_factor * (p1 ** 2 - p2 ** 2) + self.denominator_constant)
