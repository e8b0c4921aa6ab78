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

from be_hip import autograd_ops as ag
from be_hip import cpu_forms as cf
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

    # ---- the reference's fine-grained methods (utils/postprocessing_loss.py:43-117), each one HIP kernel.
    #      Both layouts of the reference are accepted: local [N,K,...] and global [B,K,...,Hp,Wp] (patch index fastest);
    #      the global one is permuted to the kernels' flat one-row-per-patch layout and back (views + one copy).
    @staticmethod
    def _flat_params(t):
        """[N,K] stays; [B,K,Hp,Wp] -> ([B*Hp*Wp,K], (B,Hp,Wp))."""
        if t.dim() == 2:
            return t.contiguous(), None
        b, k, hp, wp = t.shape
        return t.permute(0, 2, 3, 1).reshape(-1, k).contiguous(), (b, hp, wp)

    @staticmethod
    def _unflat_pixels(t, grid):
        """[N,K,21,21] -> as is, or back to [B,K,21,21,Hp,Wp]."""
        if grid is None:
            return t
        b, hp, wp = grid
        return t.view(b, hp, wp, t.shape[1], t.shape[2], t.shape[3]).permute(0, 3, 4, 5, 1, 2)

    def params2etas(self, params):
        if not params.is_cuda:                                  # BASELINE configs[0] (PyTorch-CPU plumbing): eta = 10^(2 erf(p) - 2)
            return torch.pow(10, torch.erf(params) * 2 - 2)
        return ag.Params2Etas.apply(params)

    # A CPU tensor (BASELINE configs[0]: PyTorch-CPU plumbing; a reference-style subclass left on the CPU) takes the torch expression of
    # be_hip/cpu_forms.py - the reference's own operation order, differentiable by autograd; a GPU tensor takes the HIP kernel or raises.
    def params2dists(self, params):
        if not params.is_cuda:
            return cf.params2dists(self.x.to(params.device), self.y.to(params.device), self.w, params)
        p, grid = self._flat_params(params)
        return self._unflat_pixels(ag.Params2Dists.apply(p, self.render_opts(False)), grid)

    def dists2indicators(self, dists, etas):
        if not dists.is_cuda:
            return cf.dists2indicators(dists, etas)
        if dists.dim() == 6:                                   # global layout [B,2,21,21,Hp,Wp]
            b, _, _, _, hp, wp = dists.shape
            d = dists.permute(0, 4, 5, 1, 2, 3).reshape(-1, 2, self.R, self.R)
            grid = (b, hp, wp)
        else:
            d, grid = dists, None
        e, _ = self._flat_params(etas)
        return self._unflat_pixels(ag.Dists2Indicators.apply(d, e), grid)

    def normalized_gaussian(self, x, delta=0.07):
        if not x.is_cuda:
            return cf.normalized_gaussian(x, delta)
        return ag.NormalizedGaussian.apply(x, float(torch.tensor(delta ** 2, dtype=torch.float32)))

    def inverse_3by3(self, A):
        if not A.is_cuda:
            return cf.inverse_3by3(A)
        return ag.Inverse3x3.apply(A)

    def get_image_derivative(self, img):
        if not img.is_cuda:
            return cf.image_derivative(img, self.sobel_x.to(img.device, img.dtype), self.sobel_y.to(img.device, img.dtype))
        return ag.ImageDerivative.apply(img)

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

        cy = torch.tensor([sum(1 for i in range(self.H_patches) if self.stride * i <= y <= self.stride * i + self.R - 1)
                           for y in range(self.H)], dtype=torch.float32)
        cx = torch.tensor([sum(1 for j in range(self.W_patches) if self.stride * j <= x <= self.stride * j + self.R - 1)
                           for x in range(self.W)], dtype=torch.float32)
        self.num_patches = (cy[:, None] * cx[None, :]).to(self.device)          # = nn.Fold(ones), :139-143

    def get_xy_mat(self, xx, yy):
        return (xx.view(1, self.R, self.R, 1, 1).to(self.device), yy.view(1, self.R, self.R, 1, 1).to(self.device))

    # ---- nn.Fold aggregations of the reference (:151-173) on materialised patch tensors [.., 21,21,Hp,Wp]
    def _fold(self, t, lead, mode):
        """t viewed as [lead, C, 21, 21, Hp, Wp] contiguous -> [lead, C, H, W]; differentiable for modes 0 / 1."""
        p = self.H_patches * self.W_patches
        c = t.numel() // (lead * self.R * self.R * p)
        return ag.FoldPatches.apply(t, lead, c, self.H_patches, self.W_patches, self.H, self.W, self.stride, mode)

    def _fold_cpu(self, t, lead, c):
        """nn.Fold of a CPU patch stack divided by the patch count (utils/postprocessing_loss.py:151-162)"""
        n = self.num_patches.to(t.device)
        return cf.fold(t.reshape(lead, c * self.R ** 2, -1), lead, self.R, self.H, self.W, self.stride) / n

    def local2global_color(self, patches, pair=True):
        if not patches.is_cuda:
            lead = self.batch_size * (2 if pair else 1)
            out = self._fold_cpu(patches, lead, 3)
            return out.view(self.batch_size, 2, 3, self.H, self.W) if pair else out.view(self.batch_size, 3, self.H, self.W)
        if pair:
            return self._fold(patches, self.batch_size * 2, 1).view(self.batch_size, 2, 3, self.H, self.W)
        return self._fold(patches, self.batch_size, 1).view(self.batch_size, 3, self.H, self.W)

    def local2global_bndry(self, bndry_patches):
        if not bndry_patches.is_cuda:
            return self._fold_cpu(bndry_patches, self.batch_size, 1).view(self.batch_size, 1, self.H, self.W)
        return self._fold(bndry_patches, self.batch_size, 1).view(self.batch_size, 1, self.H, self.W)

    def local2global_depth(self, depth_map, depth_mask):
        if not depth_map.is_cuda:
            return cf.local2global_depth(depth_map, depth_mask, self.batch_size, self.R, self.H, self.W, self.H_patches, self.W_patches,
                                         self.stride, self.num_patches.to(depth_map.device))
        is_int = depth_mask.dtype == torch.int32
        with torch.no_grad():                                   # a count: piecewise constant in the mask
            cnt = self._fold(depth_mask if is_int else depth_mask.to(torch.float32), self.batch_size, 2)
        cnt = cnt.view(self.batch_size, self.H, self.W)
        tot = self._fold(depth_map.to(torch.float32), self.batch_size, 0).view(self.batch_size, self.H, self.W)
        conf = cnt / self.num_patches.unsqueeze(0)
        return tot / torch.where(cnt > 0, cnt, torch.ones_like(cnt)), conf


class _LocalLossFn(torch.autograd.Function):
    """loss(est) with the analytic gradient computed in the same HIP launch (be_local_loss_f32)."""

    @staticmethod
    def forward(ctx, est, helper, img_fit, gt, bdist, deri, beta_b, beta_s):
        b = est.shape[0]
        partial, grad, _ = native.local_loss(helper.render_opts(False), est.detach().contiguous(), img_fit.contiguous(),
                                             gt.contiguous(), bdist.contiguous(), deri.contiguous(), beta_b, beta_s,
                                             want_grad=True)
        ctx.save_for_backward(grad)
        return native.local_loss_finish(partial, beta_b, beta_s)     # the three sums (fp64, patch order) and the weights: one launch

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return g * grad, None, None, None, None, None, None, None


def local_loss(helper, est, img_fit, gt_img, bndry_dist, deri, beta_bndry_loc, beta_smthns, write_back=True):
    """LocalLoss.forward (local_training.py:47-52) as one fused HIP forward+backward; differentiable w.r.t. est.
    write_back=True (default, the reference's behaviour): `est[:, 4:8] = remainder(est[:, 4:8], 2 pi)` is written into the
    caller's tensor first (local_training.py:33; float32 remainder, one kernel, the cotangent passes through) - like the
    reference this refuses a leaf tensor that requires grad.  write_back=False leaves est untouched; the loss kernel then wraps
    the raw angles itself in float64 (what the build's own training loops use: 2e-7 rad closer to the float64 reference)."""
    if write_back:
        est = ag.WrapAnglesInplace.apply(est)
    return _LocalLossFn.apply(est, helper, img_fit, gt_img, bndry_dist, deri, float(beta_bndry_loc), float(beta_smthns))


class _GlobalLossFn(torch.autograd.Function):
    """GlobalLoss(est) with the analytic gradient from the same HIP launch (be_global_loss_f32)."""

    @staticmethod
    def forward(ctx, est, helper, dcal, img_fit, img_gt, bndry_dist, deri, bndry_depth, gam, empty_mask="nan"):
        e = est.detach().to(torch.float32).contiguous()
        B, P = e.shape[:2]
        H, W = img_gt.shape[2], img_gt.shape[3]
        hp, wp, st = helper.H_patches, helper.W_patches, helper.stride
        opts = helper.render_opts(False)
        img_fit, img_gt = img_fit.contiguous(), img_gt.contiguous()
        # the CURRENT folded image / boundary (the reference detaches them before the consistency terms)
        recs = torch.stack([native.render_full(opts, dcal.consts, 0.0, False, native.global_denorm(e[b]),
                                               native.view_image_pair_nhwc(img_fit[b], st), pixels=img_fit)[0] for b in range(B)])
        m = native.fold_records_batch(opts, recs, hp, wp, H, W, st, False, want=("image", "bndry"))     # one launch for the batch
        G, Gb = m["image"], m["bndry"]
        Gd = helper.get_image_derivative(G.view(B * 2, 3, H, W)).view(B, 2, 3, H - 2, W - 2)
        g6 = [gam[k] for k in ("color", "color_cons", "bndry_cons", "smthns", "smthns_cons", "bndry_loc")]
        partial, grad, gdep = native.global_loss(opts, dcal.consts, e, img_fit, img_gt, G, Gd, Gb, bndry_dist.contiguous(),
                                                 deri.contiguous(), bndry_depth.contiguous(), g6, hp, wp, st)
        t = partial.double().sum(dim=0)
        n1, n3, n4 = B * 2 * 441 * P, B * 441 * P, B * 2 * 361 * P
        # an empty depth mask: the reference computes `.sum() / mask.sum()` = 0 / 0 = NaN (global_training.py:127) and so does
        # this operator by default; empty_mask="zero" drops the term instead (the build's own training loops ask for that:
        # a NaN there ends a run, and a small synthetic set can produce an image whose mask is empty)
        msum = t[7] if empty_mask == "nan" else torch.clamp(t[7], min=1.0)
        loss = (g6[0] * t[0] + g6[1] * t[1]) / n1 + (g6[2] * t[2] + g6[5] * t[5]) / n3 + (g6[3] * t[3] + g6[4] * t[4]) / n4 \
            + gam["depth"] * t[6] / msum
        grad[:, 8:12] += (gam["depth"] / msum).to(torch.float32) * gdep
        ctx.save_for_backward(grad.view(B, P, 12))
        ctx.terms = t
        return loss.to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return (g * grad,) + (None,) * 9


def global_loss(helper, depth_cal, est, img_ny, img_gt, bndry_dist, deri, bndry_depth, gamma, empty_mask="nan"):
    """GlobalLoss.forward (global_training.py:147-157) as fused HIP launches; differentiable w.r.t. est [B,P,12].
    helper: a PostProcessGlobalBase; gamma: dict with the seven weights color, color_cons, bndry_cons,
    smthns, smthns_cons, bndry_loc, depth (the --gamma_* arguments of utils/args.py:53-59).
    empty_mask: "nan" (default, the reference's arithmetic: depth term = 0 / 0 when no pixel is in the depth mask) or "zero"
    (the term and its gradient are dropped for that batch)."""
    if empty_mask not in ("nan", "zero"):
        raise ValueError("global_loss: empty_mask must be 'nan' or 'zero'")
    return _GlobalLossFn.apply(est, helper, depth_cal, img_ny, img_gt, bndry_dist, deri, bndry_depth, dict(gamma), empty_mask)
