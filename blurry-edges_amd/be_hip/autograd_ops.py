"""torch.autograd.Function wrappers of the fine-grained PostProcess / DepthEtas operators.

A caller that subclasses PostProcessLocalBase / PostProcessGlobalBase the way the reference's training scripts do
(LocalLoss, local_training.py:10-52; GlobalLoss, global_training.py:11-157) chains the inherited methods itself and then calls
`loss.backward()` (local_training.py:106, global_training.py:212).  Every Function here is one HIP forward kernel and one HIP
adjoint kernel (csrc/be_compat.hip, be_compat_bwd.hip, be_elementwise.hip) behind `native.*`, i.e. behind torch.ops.be.* or the
ctypes binding; layouts are the kernels' flat ones, the reference's [B,K,...,Hp,Wp] layouts are permuted by the callers in
utils/postprocessing_loss.py with ordinary (differentiable) view operations.
"""
import torch
from torch.autograd import Function

from be_hip import native


def _c(t):
    return t.contiguous()


class Params2Etas(Function):
    """eta = 10^(2 erf(p) - 2)   (utils/postprocessing_loss.py:88-89)"""

    @staticmethod
    def forward(ctx, p):
        ctx.save_for_backward(p)
        return native.params2etas(p)

    @staticmethod
    def backward(ctx, g):
        (p,) = ctx.saved_tensors
        return native.params2etas_bwd(p, _c(g)).view_as(p)


class Params2Dists(Function):
    """params8 [N,8] -> dists [N,2,21,21]   (utils/postprocessing_loss.py:43-86)"""

    @staticmethod
    def forward(ctx, params8, opts):
        p = _c(params8)
        ctx.save_for_backward(p)
        ctx.opts = opts
        return native.params2dists(opts, p)

    @staticmethod
    def backward(ctx, g):
        (p,) = ctx.saved_tensors
        return native.params2dists_bwd(ctx.opts, p, _c(g)), None


class Dists2Indicators(Function):
    """dists [N,2,21,21], etas [N,2] -> wedges [N,3,21,21]   (:91-95)"""

    @staticmethod
    def forward(ctx, dists, etas):
        d, e = _c(dists), _c(etas)
        ctx.save_for_backward(d, e)
        return native.dists2indicators(d, e)

    @staticmethod
    def backward(ctx, g):
        d, e = ctx.saved_tensors
        gd, ge = native.dists2indicators_bwd(d, e, _c(g))
        return gd, ge


class Inverse3x3(Function):
    """A [...,3,3] -> A^-1 (cofactors, fp64 inside)   (:104-112)"""

    @staticmethod
    def forward(ctx, a):
        inv = native.inverse3x3(_c(a))
        ctx.save_for_backward(inv)
        return inv

    @staticmethod
    def backward(ctx, g):
        (inv,) = ctx.saved_tensors
        return native.inverse3x3_bwd(inv, _c(g))


class ImageDerivative(Function):
    """img [N,C,H,W] -> Sobel magnitude [N,C,H-2,W-2]   (:114-117)"""

    @staticmethod
    def forward(ctx, img):
        x = _c(img)
        ctx.save_for_backward(x)
        return native.image_derivative(x)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return native.image_derivative_bwd(x, _c(g))


class NormalizedGaussian(Function):
    """exp(-x^2 / delta^2)   (:97-98)"""

    @staticmethod
    def forward(ctx, x, delta_sq):
        xc = _c(x)
        ctx.save_for_backward(xc)
        ctx.delta_sq = delta_sq
        return native.normalized_gaussian(xc, delta_sq)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return native.normalized_gaussian_bwd(x, _c(g), ctx.delta_sq), None


class Etas2Depth(Function):
    """DepthEtas.etas2depth (utils/depth_etas.py:23-34); eta1 / eta2 broadcast against each other."""

    @staticmethod
    def forward(ctx, eta1, eta2, consts):
        ctx.shapes = (eta1.shape, eta2.shape)
        a, b = torch.broadcast_tensors(eta1, eta2)
        a, b = _c(a), _c(b)
        ctx.save_for_backward(a, b)
        ctx.consts = consts
        return native.etas2depth(consts, a, b)

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        g1, g2 = native.etas2depth_bwd(ctx.consts, a, b, _c(g))
        s1, s2 = ctx.shapes
        return g1.sum_to_size(s1) if g1.shape != s1 else g1, g2.sum_to_size(s2) if g2.shape != s2 else g2, None


class Depth2Sigma(Function):
    """DepthEtas.depth2sigma (utils/depth_etas.py:36-37)"""

    @staticmethod
    def forward(ctx, depth, consts, rho_prime):
        d = _c(depth)
        ctx.save_for_backward(d)
        ctx.consts, ctx.rho = consts, float(rho_prime)
        return native.depth2sigma(consts, d, rho_prime)

    @staticmethod
    def backward(ctx, g):
        (d,) = ctx.saved_tensors
        return native.depth2sigma_bwd(ctx.consts, d, ctx.rho, _c(g)), None, None


class FoldPatches(Function):
    """nn.Fold of [B,C,21,21,Hp,Wp] -> [B,C,H,W] (sum: mode 0; sum / overlap count: mode 1)   (:151-173)"""

    @staticmethod
    def forward(ctx, src, B, C, hp, wp, H, W, stride, mode):
        ctx.geom = (hp, wp, stride, mode)
        ctx.src_shape = src.shape
        return native.fold_patches(_c(src), B, C, hp, wp, H, W, stride, mode)

    @staticmethod
    def backward(ctx, g):
        hp, wp, stride, mode = ctx.geom
        return (native.fold_patches_bwd(_c(g), hp, wp, stride, mode).view(ctx.src_shape),) + (None,) * 8


class WrapAnglesInplace(Function):
    """est[:, 4:8] <- remainder(est[:, 4:8], 2 pi) written back into est, as LocalLoss.get_patches does to the CNN output
    (local_training.py:33); the slope of remainder is 1, so the cotangent passes through."""

    @staticmethod
    def forward(ctx, est):
        native.wrap_angles_(est, 4, 8)
        ctx.mark_dirty(est)
        return est

    @staticmethod
    def backward(ctx, g):
        return g
