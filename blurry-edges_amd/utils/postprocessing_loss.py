"""Blurred-wedge post-processing behind the reference's class names (utils/postprocessing_loss.py:7-173).

PostProcessBase / PostProcessLocalBase / PostProcessGlobalBase keep the reference's constructor arguments
and attributes (R, batch_size, w, lambda_ridge, x, y, ridge, sobel_x, sobel_y, stride, H, W, H_patches,
W_patches, num_patches).  The math runs in libblurry_edges_hip: one fused kernel per pass instead of the
reference's ~40 elementwise ATen ops, in a flat one-row-per-patch layout.
"""
from abc import ABC

import numpy as np
import torch
import torch.nn as nn

from be_hip import native


class PostProcessBase(nn.Module, ABC):
    def __init__(self, args, device):
        super().__init__()
        self.device = device
        self.R = args.R
        if self.R != native.BE_R:
            raise NotImplementedError(f"the HIP renderer is built for R = {native.BE_R}")
        self.batch_size = args.batch_size
        self.w = args.w
        self.lambda_ridge = (args.alpha_lambda * self.R ** 2) ** 2
        lin = torch.linspace(-1.0, 1.0, self.R)
        yy, xx = torch.meshgrid([lin, lin], indexing='ij')
        self.x, self.y = self.get_xy_mat(xx, yy)
        sx = torch.tensor([[-1, 0, 1], [-2, 0, 2], [-1, 0, 1]], dtype=torch.float32, device=device)
        self.sobel_x = sx[None, None].repeat(3, 1, 1, 1)
        self.sobel_y = (-sx.t())[None, None].repeat(3, 1, 1, 1).contiguous()
        self._opts = native.RenderOpts(lambda_ridge=float(torch.tensor(self.lambda_ridge, dtype=torch.float32)),
                                       w=float(self.w), delta_sq=float(torch.tensor(0.07 ** 2, dtype=torch.float32)),
                                       wrap_angles=0)
        for i, v in enumerate(lin.tolist()):
            self._opts.lin[i] = v

    def get_xy_mat(self, xx, yy):
        raise NotImplementedError

    # ---- elementwise pieces of the reference API that map 1:1 onto a kernel
    def params2etas(self, params):
        return native.params2etas(params)

    def normalized_gaussian(self, x, delta=0.07):
        return torch.exp(-x ** 2 / delta ** 2)

    # ---- fused passes (flat layout)
    def render_opts(self, wrap_angles=False):
        opts = native.RenderOpts.from_buffer_copy(self._opts)
        opts.wrap_angles = int(bool(wrap_angles))
        return opts

    def render_colors(self, params10, patches, wrap_angles=False, want=()):
        """Colours-only pass: params10 [N,10], patches [N,3,21,21] -> (colors [N,3(rgb),3(wedge)], extras).
        One launch replaces params2dists + params2etas + dists2indicators + the ridge solve
        (utils/postprocessing_loss.py:43-112; blurry_edges_test.py:19-34)."""
        return native.render_colors(self.render_opts(wrap_angles), params10.contiguous(), patches.contiguous(), want=want)


class PostProcessLocalBase(PostProcessBase):
    def __init__(self, args, device):
        super().__init__(args, device)
        self.ridge = self.lambda_ridge * torch.eye(3, device=self.device).unsqueeze(0)

    def get_xy_mat(self, xx, yy):
        return xx.view(1, self.R, self.R).to(self.device), yy.view(1, self.R, self.R).to(self.device)


class PostProcessGlobalBase(PostProcessBase):
    def __init__(self, args, device):
        super().__init__(args, device)
        self.ridge = self.lambda_ridge * torch.eye(3, device=self.device)[None, None, None]
        self.stride = args.stride
        self.H, self.W = args.img_size
        self.H_patches = int(np.floor((self.H - self.R) / self.stride) + 1)
        self.W_patches = int(np.floor((self.W - self.R) / self.stride) + 1)

    def get_xy_mat(self, xx, yy):
        return (xx.view(1, self.R, self.R, 1, 1).to(self.device), yy.view(1, self.R, self.R, 1, 1).to(self.device))


class _LocalLossFn(torch.autograd.Function):
    """loss(est) with the analytic gradient computed in the same HIP launch (be_local_loss_f32)."""

    @staticmethod
    def forward(ctx, est, helper, img_fit, gt, bdist, deri, beta_b, beta_s):
        b = est.shape[0]
        partial, grad, _ = native.local_loss(helper.render_opts(False), est.detach().contiguous(), img_fit.contiguous(),
                                             gt.contiguous(), bdist.contiguous(), deri.contiguous(), beta_b, beta_s,
                                             want_grad=True)
        ctx.save_for_backward(grad)
        s = partial.sum(dim=0)                                  # three scalars; deterministic reduction
        return s[0] / (b * 441) + beta_b * s[1] / (b * 441) + beta_s * s[2] / (b * 361)

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return g * grad, None, None, None, None, None, None, None


def local_loss(helper, est, img_fit, gt_img, bndry_dist, deri, beta_bndry_loc, beta_smthns):
    """LocalLoss.forward (local_training.py:47-52) as one fused HIP forward+backward; differentiable w.r.t. est.
    Unlike the reference it does NOT write the wrapped angles back into est."""
    return _LocalLossFn.apply(est, helper, img_fit, gt_img, bndry_dist, deri, float(beta_bndry_loc), float(beta_smthns))
