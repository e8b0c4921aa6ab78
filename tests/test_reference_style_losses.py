"""The reference's training-side subclasses keep working on this package (`local_training.py` and
`global_training.py` subclass PostProcessLocalBase / PostProcessGlobalBase, chain the inherited methods and call
`loss.backward()`).  The LocalLoss / GlobalLoss below are written here against the build's base classes in the reference's style
(same method chain, same tensor layouts, the caller's own torch glue between the inherited methods) and are compared with the
goldens the REAL reference produced under autograd (g4: local_training.LocalLoss, g11: global_training.GlobalLoss, float64 runs)
and with the build's fused operators.  Every adjoint kernel is also checked on its own against float64 autograd of the oracle.
"""
import math

import os

import numpy as np
import pytest
import torch

from conftest import load_golden, relmax

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def env():
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test run without a GPU")
    from be_hip import native, synth
    import utils
    native.lib()
    return dict(native=native, synth=synth, utils=utils)


def T(a, dt=torch.float32):
    return torch.from_numpy(np.asarray(a)).to(dt)


# ----------------------------------------------------------------------------------------------- caller-side subclasses
def make_local_loss(utils):
    class LocalLoss(utils.PostProcessLocalBase):
        """a caller's loss module composed of the inherited methods only (the role of local_training.py:10-52)"""

        def __init__(self, args, device):
            super().__init__(args, device)
            self.beta_bndry_loc, self.beta_smthns = args.beta_bndry_loc, args.beta_smthns

        def render(self, est, img_ny):
            est[:, 4:8] = torch.remainder(est[:, 4:8], 2 * torch.pi)          # the in-place write-back of the reference
            dists = self.params2dists(est[:, :8])
            wedges = self.dists2indicators(dists, self.params2etas(est[:, 8:]))
            A = wedges.permute(0, 2, 3, 1).reshape(self.batch_size, -1, 3)
            y = img_ny.reshape(self.batch_size, -1, 3)
            At = A.transpose(1, 2)
            colors = (self.inverse_3by3(At @ A + self.ridge) @ (At @ y)).transpose(1, 2)
            patches = (wedges[:, None] * colors[..., None, None]).sum(dim=2)
            d1, d2 = dists[:, 0], dists[:, 1]
            near = torch.where(d1.abs() < d2.abs(), d1.abs(), d2.abs())
            return patches, self.normalized_gaussian(torch.where(d2 >= 0, d2, near))

        def forward(self, est, img_ny, gt_img, bndry_dist, deri):
            patches, bnd = self.render(est, img_ny)
            fit = ((gt_img - patches.permute(0, 2, 3, 1)) ** 2).sum(-1).mean()
            loc = ((bndry_dist * bnd) ** 2).mean()
            smooth = ((deri.permute(0, 3, 1, 2) - self.get_image_derivative(patches)) ** 2).sum(1).mean()
            return fit + self.beta_bndry_loc * loc + self.beta_smthns * smooth

    return LocalLoss


def make_global_loss(utils):
    import torch.nn as nn

    class GlobalLoss(utils.PostProcessGlobalBase):
        """the role of global_training.py:11-157: pass-B render, folds and the seven weighted terms, from inherited methods"""

        def __init__(self, args, depth_cal, gamma, device):
            super().__init__(args, device)
            self.depthCal, self.g = depth_cal, gamma

        def unfold(self, x, k):
            return nn.Unfold(k, stride=self.stride)(x)

        def grid(self, t, *lead):
            return t.view(*lead, self.H_patches, self.W_patches)

        def colors(self, wedges, img_patches):                      # both apertures share one colour set: 882 rows
            B, Hp, Wp = self.batch_size, self.H_patches, self.W_patches
            A = wedges.permute(0, 5, 6, 1, 3, 4, 2).reshape(B, Hp, Wp, -1, 3)
            y = img_patches.permute(0, 5, 6, 1, 3, 4, 2).reshape(B, Hp, Wp, -1, 3)
            At = A.transpose(-1, -2)
            return (self.inverse_3by3(At @ A + self.ridge) @ (At @ y)).permute(0, 4, 3, 1, 2)

        def forward(self, est, img_ny, img_gt, bndry_dist, deri, bndry_depth):
            B, R, H, W = self.batch_size, self.R, self.H, self.W
            est = est.permute(0, 2, 1).reshape(B, 12, self.H_patches, self.W_patches)
            xy_angles = torch.cat([est[:, :4] * 3, torch.remainder((est[:, 4:8] + 1) * torch.pi, 2 * torch.pi)], dim=1)
            etas = self.params2etas(est[:, 8:] + 0.5)
            as_nchw = lambda im: im.reshape(B * 2, H, W, 3).permute(0, 3, 1, 2)
            img_p = self.grid(self.unfold(as_nchw(img_ny), R), B, 2, 3, R, R)
            gt_p = self.grid(self.unfold(as_nchw(img_gt), R), B, 2, 3, R, R)

            dists = self.params2dists(xy_angles)
            w1, w2 = self.dists2indicators(dists, etas[:, :2]), self.dists2indicators(dists, etas[:, 2:])
            col = self.colors(torch.stack([w1, w2], dim=1), img_p)
            comp = lambda w: (w.unsqueeze(1) * col.unsqueeze(-3).unsqueeze(-3)).sum(dim=2)
            patches = torch.stack([comp(w1), comp(w2)], dim=1)
            d1, d2 = dists[:, 0], dists[:, 1]
            bnd = self.normalized_gaussian(torch.where(d2 >= 0, d2, torch.where(d1.abs() < d2.abs(), d1.abs(), d2.abs()))).unsqueeze(1)
            m1 = (self.normalized_gaussian(d1) > 0.5).to(torch.int32)
            m2 = (self.normalized_gaussian(d2) > 0.5).to(torch.int32) * 2
            mask = torch.where((m2 == 2) | (d2 >= 0), m2, m1)
            z1 = self.depthCal.etas2depth(etas[:, 0], etas[:, 2])[:, None, None]
            z2 = self.depthCal.etas2depth(etas[:, 1], etas[:, 3])[:, None, None]
            zmap = torch.where(mask == 1, z1, torch.where(mask == 2, z2, torch.zeros_like(d1)))

            g_img = self.local2global_color(patches).detach()
            g_bnd = self.local2global_bndry(bnd).detach()
            cons_img = self.grid(self.unfold(g_img.view(B * 2, 3, H, W), R), B, 2, 3, R, R)
            cons_bnd = self.grid(self.unfold(g_bnd, R), B, 1, R, R)
            g_deri = self.get_image_derivative(g_img.view(B * 2, 3, H, W))
            cons_deri = self.grid(self.unfold(g_deri, R - 2), B, 2, 3, R - 2, R - 2)
            gt_deri = self.grid(self.unfold(deri.permute(0, 1, 4, 2, 3).reshape(B * 2, 3, H - 2, W - 2), R - 2), B, 2, 3, R - 2, R - 2)
            p_deri = self.get_image_derivative(patches.permute(0, 1, 5, 6, 2, 3, 4).flatten(0, 3))
            p_deri = p_deri.view(B, 2, self.H_patches, self.W_patches, 3, R - 2, R - 2).permute(0, 1, 4, 5, 6, 2, 3)
            dist_p = self.grid(self.unfold(torch.log2(bndry_dist.unsqueeze(1) + 1), R), B, 1, R, R)
            depth_p = self.grid(self.unfold(bndry_depth.unsqueeze(1), R), B, R, R)
            live = ((depth_p != 0) & (mask != 0)).to(zmap.dtype)
            g = self.g
            return g["color"] * ((gt_p - patches) ** 2).sum(2).mean() \
                + g["color_cons"] * ((patches - cons_img) ** 2).sum(2).mean() \
                + g["bndry_cons"] * ((bnd - cons_bnd) ** 2).mean() \
                + g["smthns"] * ((p_deri - gt_deri) ** 2).sum(2).mean() \
                + g["smthns_cons"] * ((p_deri - cons_deri) ** 2).sum(2).mean() \
                + g["bndry_loc"] * ((dist_p * bnd) ** 2).mean() \
                + g["depth"] * (((zmap - depth_p) * live) ** 2).sum() / live.sum()

    return GlobalLoss


# ----------------------------------------------------------------------------------------------- the two headline tests
def test_reference_style_local_loss_subclass_trains_and_matches_golden_g4(env):
    """loss and d loss / d est of a caller-written LocalLoss(PostProcessLocalBase) against the reference's own LocalLoss under
    autograd (golden g4, float64 run) and against the fused operator utils.local_loss."""
    synth, utils = env["synth"], env["utils"]
    g = load_golden("g4_local_loss")
    B, S = 64, synth.SEED_DEFAULT
    leaf = T(synth.plausible_params10(B, name="loss_params")).to(DEV).requires_grad_(True)
    img = T(synth.f32(synth.hash_uniform(S, "loss_img", (B, 21, 21, 3)))).to(DEV)
    gt = T(synth.f32(synth.hash_uniform(S, "loss_gt", (B, 21, 21, 3)))).to(DEV)
    bd = T(synth.f32(5.0 * synth.hash_uniform(S, "loss_bd", (B, 21, 21)))).to(DEV)
    de = T(synth.f32(synth.hash_uniform(S, "loss_deri", (B, 19, 19, 3)))).to(DEV)
    a = utils.get_args("local_train", argv=[])
    crit = make_local_loss(utils)(a, DEV)
    loss = crit(leaf * 1.0, img, gt, bd, de)                      # the reference mutates its input: hand it a non-leaf
    loss.backward()
    ref = float(g["f64_loss"])
    e_loss = abs(float(loss.detach()) - ref) / abs(ref)
    e_grad = relmax(leaf.grad.cpu(), g["f64_grad"])
    print("subclass LocalLoss: loss %.3e  grad %.3e   (reference fp32 vs its fp64: loss %.3e  grad %.3e)"
          % (e_loss, e_grad, abs(float(g["f32_loss"]) - ref) / abs(ref), relmax(g["f32_grad"], g["f64_grad"])))
    assert e_loss <= 1e-5                                         # the tolerances of test_local_loss_value_and_gradient_vs_fp64_golden
    assert e_grad <= 2e-4
    # ... and the fused operator on the same inputs (write-back on: the reference's behaviour; a leaf is refused like there)
    leaf2 = leaf.detach().clone().requires_grad_(True)
    with pytest.raises(RuntimeError):
        utils.local_loss(crit, leaf2, img, gt, bd, de, a.beta_bndry_loc, a.beta_smthns)
    fused = utils.local_loss(crit, leaf2 * 1.0, img, gt, bd, de, a.beta_bndry_loc, a.beta_smthns)
    fused.backward()
    assert abs(float(fused.detach()) - float(loss.detach())) <= 1e-5 * abs(ref)
    assert relmax(leaf.grad.cpu(), leaf2.grad.cpu()) <= 2e-4


def test_local_loss_write_back_wraps_the_callers_tensor_in_place_like_the_reference(env):
    synth, utils = env["synth"], env["utils"]
    B = 64
    a = utils.get_args("local_train", argv=[])
    h = utils.PostProcessLocalBase(a, DEV)
    p = synth.plausible_params10(B, name="wb").astype(np.float64)
    p[:, 4:8] += 2 * math.pi * (np.arange(B * 4).reshape(B, 4) % 7 - 3)             # whole turns off
    raw = T(synth.f32(p)).to(DEV)
    img = T(synth.f32(synth.hash_uniform(1, "wb_img", (B, 21, 21, 3)))).to(DEV)
    bd = T(synth.f32(synth.hash_uniform(1, "wb_bd", (B, 21, 21)))).to(DEV)
    de = T(synth.f32(synth.hash_uniform(1, "wb_de", (B, 19, 19, 3)))).to(DEV)
    leaf = raw.clone().requires_grad_(True)
    est = leaf * 1.0
    loss = utils.local_loss(h, est, img, img, bd, de, 1e-3, 5e-4)
    loss.backward()
    expect = raw.cpu().clone()
    expect[:, 4:8] = torch.remainder(expect[:, 4:8], 2 * torch.pi)                   # float32 remainder on the CPU = the reference's op
    assert torch.equal(est.detach().cpu(), expect)                                   # bit for bit, written into the caller's tensor
    assert torch.isfinite(leaf.grad).all() and float(leaf.grad.abs().max()) > 0
    # write_back=False leaves the tensor alone and gives the same loss (the kernel wraps for itself)
    est2 = raw.clone()
    loss2 = utils.local_loss(h, est2, img, img, bd, de, 1e-3, 5e-4, write_back=False)
    assert torch.equal(est2, raw)
    assert abs(float(loss2) - float(loss.detach())) <= 2e-5 * abs(float(loss2))


def test_reference_style_global_loss_subclass_trains_and_matches_golden_g11(env):
    """the analogous GlobalLoss(PostProcessGlobalBase) on one 147 x 147 sample against the reference's GlobalLoss under autograd
    (golden g11, float64 run, final gammas) and against utils.global_loss."""
    synth, utils = env["synth"], env["utils"]
    from oracle import global_loss as ogl
    g = load_golden("g11_global_loss")
    a = utils.get_args("global_train", argv=[])
    a.batch_size = 1
    dcal = utils.DepthEtas(a, DEV)
    crit = make_global_loss(utils)(a, dcal, ogl.GAMMA_FINAL, DEV)
    smp = {k: torch.from_numpy(v)[None].to(DEV) for k, v in synth.synthetic_global_sample(147, 147).items()}
    est = torch.from_numpy(synth.plausible_global_output(4096))[None].to(DEV).requires_grad_(True)
    loss = crit(est, smp["img_gt"], smp["img_gt"], smp["bndry_dist"], smp["deri"], smp["bndry_depth"])
    loss.backward()
    ref = float(g["f64_loss"])
    e_loss = abs(float(loss.detach()) - ref) / abs(ref)
    e_grad = relmax(est.grad[0].cpu(), g["f64_grad"])
    print("subclass GlobalLoss: loss %.3e  grad %.3e   (reference fp32 vs its fp64: loss %.3e  grad %.3e)"
          % (e_loss, e_grad, abs(float(g["f32_loss"]) - ref) / abs(ref), relmax(g["f32_grad"], g["f64_grad"])))
    assert e_loss <= 2e-5                                         # the tolerances of test_global_loss_value_and_gradient_vs_fp64_golden
    assert e_grad <= 2e-4
    est2 = est.detach().clone().requires_grad_(True)
    helper = utils.PostProcessGlobalBase(a, DEV)
    fused = utils.global_loss(helper, dcal, est2, smp["img_gt"], smp["img_gt"], smp["bndry_dist"], smp["deri"], smp["bndry_depth"],
                              ogl.GAMMA_FINAL)
    fused.backward()
    assert abs(float(fused.detach()) - float(loss.detach())) <= 4e-5 * abs(ref)
    assert relmax(est.grad.cpu(), est2.grad.cpu()) <= 4e-4


# ----------------------------------------------------------------------------------------------- every adjoint on its own
def _vjp64(fn, inputs, cot):
    xs = [x.detach().double().cpu().requires_grad_(True) for x in inputs]
    out = fn(*xs)
    grads = torch.autograd.grad(out, xs, cot.double().cpu())
    return out.detach(), grads


def test_every_inherited_method_has_the_adjoint_of_its_float64_oracle(env):
    synth, utils = env["synth"], env["utils"]
    from oracle import render as orr, depth as od, tiling as ot
    a = utils.get_args("local_train", argv=[])
    h = utils.PostProcessLocalBase(a, DEV)
    n = 96
    p10 = T(synth.plausible_params10(n, name="adj"))
    cot = lambda shape, name: T(synth.f32(synth.hash_uniform(7, name, shape) - 0.5))

    def run(fn_hip, fn_ref, inputs, name, tol_out, tol_grad):
        xs = [x.clone().to(DEV).requires_grad_(True) for x in inputs]
        out = fn_hip(*xs)
        c = cot(tuple(out.shape), name)
        out.backward(c.to(DEV))
        o_ref, g_ref = _vjp64(fn_ref, inputs, c)
        assert relmax(out.detach().cpu(), o_ref) <= tol_out, name
        for x, gr in zip(xs, g_ref):
            e = relmax(x.grad.cpu(), gr)
            print(f"adjoint {name}: {e:.2e}")
            assert e <= tol_grad, name

    run(h.params2etas, orr.params2etas, [p10[:, 8:]], "params2etas", 2e-6, 1e-6)
    # params2dists: the derivative is piecewise smooth - away from the measure-zero kinks float64 autograd and the kernel agree
    run(h.params2dists, orr.params2dists, [p10[:, :8]], "params2dists", 1e-5, 1e-5)
    d = orr.params2dists(p10[:, :8])
    e = orr.params2etas(p10[:, 8:])
    e = torch.clamp(e, min=3e-3)        # a float32 distance has an absolute error of ~1e-7: keep 1/eta from amplifying it in THIS check
    run(h.dists2indicators, orr.dists2indicators, [d, e], "dists2indicators", 3e-5, 2e-5)
    G = orr.render_pass_a(p10, T(synth.uniform_patches(n, name="adj_img")))["G"]
    run(h.inverse_3by3, lambda A: torch.linalg.inv(A), [G], "inverse_3by3", 1e-5, 2e-5)
    img = T(synth.uniform_patches(6, name="adj_deriv"))
    run(h.get_image_derivative, orr.image_derivative, [img], "image_derivative", 1e-5, 1e-5)
    run(h.normalized_gaussian, orr.normalized_gaussian, [0.3 * (T(synth.uniform_patches(2, name="adj_ng")) - 0.5)], "normalized_gaussian",
        1e-5, 1e-5)
    # DepthEtas on a grid that covers the four branches; points within 1e-4 of a branch line are left out (the derivative jumps there)
    ev = utils.get_args("eval", argv=[])
    dc = utils.DepthEtas(ev, DEV)
    c = od.depth_consts()
    lin = torch.linspace(1e-3, 1.0, 97)
    e1, e2 = [t.reshape(-1).contiguous() for t in torch.meshgrid(lin, lin, indexing="ij")]
    _, br = od.etas2depth(c, e1, e2, return_branch=True)
    _, brp = od.etas2depth(c, e1 + 1e-4, e2 - 1e-4, return_branch=True)
    _, brm = od.etas2depth(c, e1 - 1e-4, e2 + 1e-4, return_branch=True)
    keep = (br == brp) & (br == brm)
    assert set(br[keep].tolist()) == {0, 1, 2, 3}
    e1, e2 = e1[keep], e2[keep]
    run(dc.etas2depth, lambda x, y: od.etas2depth(c, x, y), [e1, e2], "etas2depth", 5e-6, 1e-5)
    z = 0.7 + 0.6 * T(synth.f32(synth.hash_uniform(7, "adj_z", (500,))))
    run(lambda t: dc.depth2sigma(t, 10.39), lambda t: od.depth2sigma(c, t, 10.39), [z], "depth2sigma", 5e-6, 1e-5)
    # broadcasting keeps working (global_training.py:91-92 calls it on [B,Hp,Wp] slices; scalars must reduce their cotangent)
    s1 = torch.tensor(0.4, device=DEV, requires_grad=True)
    zz = dc.etas2depth(s1, e2[:50].to(DEV))
    zz.sum().backward()
    assert s1.grad.shape == s1.shape and torch.isfinite(s1.grad)

    # folds in the reference's global layout (batch 2, 33 x 35 image -> 7 x 8 patches)
    ag = utils.get_args("global_train", argv=[])
    ag.batch_size, ag.img_size = 2, (33, 35)
    hg = utils.PostProcessGlobalBase(ag, DEV)
    hp, wp = hg.H_patches, hg.W_patches
    pat = T(synth.f32(synth.hash_uniform(7, "adj_fold", (2, 2, 3, 21, 21, hp, wp))))

    def ref_color(t):                                              # nn.Fold semantics of utils/postprocessing_loss.py:151-155
        f = torch.nn.Fold(output_size=[33, 35], kernel_size=21, stride=2)(t.reshape(4, 3 * 441, -1))
        return f.view(2, 2, 3, 33, 35) / ot.num_patches(33, 35, dtype=t.dtype)
    run(hg.local2global_color, ref_color, [pat], "local2global_color", 1e-6, 1e-6)
    bp = T(synth.f32(synth.hash_uniform(7, "adj_foldb", (2, 1, 21, 21, hp, wp))))
    run(hg.local2global_bndry,
        lambda t: torch.nn.Fold(output_size=[33, 35], kernel_size=21, stride=2)(t.reshape(2, 441, -1)).view(2, 1, 33, 35)
        / ot.num_patches(33, 35, dtype=t.dtype), [bp], "local2global_bndry", 1e-6, 1e-6)
    mask = (T(synth.f32(synth.hash_uniform(7, "adj_mask", (2, 21, 21, hp, wp)))) * 3).to(torch.int32)
    dm = T(synth.f32(synth.hash_uniform(7, "adj_dm", (2, 21, 21, hp, wp))))

    def ref_depth(t):
        fold = torch.nn.Fold(output_size=[33, 35], kernel_size=21, stride=2)
        cnt = fold((mask.reshape(2, 441, -1) > 0).to(t.dtype)).view(2, 33, 35)
        return fold(t.reshape(2, 441, -1)).view(2, 33, 35) / torch.where(cnt > 0, cnt, torch.ones_like(cnt))
    run(lambda t: hg.local2global_depth(t, mask.to(DEV))[0], ref_depth, [dm], "local2global_depth", 1e-6, 1e-6)


def test_adjoint_kernels_same_results_through_both_bindings(env, monkeypatch):
    """torch.ops.be.* and the ctypes binding call the same C symbols: identical bits for every new forward / adjoint operator."""
    native, synth, utils = env["native"], env["synth"], env["utils"]
    a = utils.get_args("local_train", argv=[])
    h = utils.PostProcessLocalBase(a, DEV)
    dc = utils.DepthEtas(utils.get_args("eval", argv=[]), DEV)
    n = 32
    p10 = T(synth.plausible_params10(n, name="bind")).to(DEV)
    u = lambda shape, name: T(synth.f32(synth.hash_uniform(3, name, shape))).to(DEV)

    def both():
        o = h.render_opts(False)
        p8 = p10[:, :8].contiguous()
        d = native.params2dists(o, p8)
        e = native.params2etas(p10[:, 8:].contiguous())
        w = native.dists2indicators(d, e)
        out = [d, e, w, native.params2dists_bwd(o, p8, u(tuple(d.shape), "gd")), native.params2etas_bwd(p10[:, 8:].contiguous(), u((n, 2), "ge"))]
        out += list(native.dists2indicators_bwd(d, e, u(tuple(w.shape), "gw")))
        A = u((n, 3, 3), "A") + 2 * torch.eye(3, device=DEV)
        inv = native.inverse3x3(A)
        out += [inv, native.inverse3x3_bwd(inv, u((n, 3, 3), "gA"))]
        img = u((4, 3, 21, 21), "img")
        out += [native.image_derivative(img), native.image_derivative_bwd(img, u((4, 3, 19, 19), "gimg"))]
        x = u((1000,), "x") - 0.5
        out += [native.normalized_gaussian(x, 0.0049), native.normalized_gaussian_bwd(x, u((1000,), "gx"), 0.0049)]
        e1, e2 = u((1000,), "e1"), u((1000,), "e2")
        out += [native.etas2depth(dc.consts, e1, e2)] + list(native.etas2depth_bwd(dc.consts, e1, e2, u((1000,), "gz")))
        z = 0.7 + u((1000,), "z")
        out += [native.depth2sigma(dc.consts, z, 10.39), native.depth2sigma_bwd(dc.consts, z, 10.39, u((1000,), "gs"))]
        src = u((2, 3, 21, 21, 4, 5), "src")
        f = native.fold_patches(src, 2, 3, 4, 5, 27, 29, 2, 1)
        out += [f, native.fold_patches_bwd(u((2, 3, 27, 29), "gf"), 4, 5, 2, 1), native.fold_patches_bwd(u((2, 3, 27, 29), "gf"), 4, 5, 2, 0)]
        est = (p10 + 20.0).contiguous()
        out.append(native.wrap_angles_(est))
        return [t.clone() for t in out]

    if os.environ.get("BE_TORCH_OPS", "1") == "0" or os.environ.get("BE_LIB_DIR"):
        pytest.skip("compares the two bindings: the torch-operator binding is switched off in this environment")
    assert native.ops() is not None
    via_ops = both()
    monkeypatch.setattr(native, "_ops", False)                      # BE_TORCH_OPS=0: ctypes alone
    assert native.ops() is None
    via_ctypes = both()
    assert len(via_ops) == len(via_ctypes) == 22
    for i, (x, y) in enumerate(zip(via_ops, via_ctypes)):
        assert torch.equal(x, y), i


def test_pass_b_folds_losses_and_attention_same_results_through_both_bindings(env, monkeypatch):
    """render_full, fold_records (single and batched), local_loss (+ finish), global_loss and the attention forward /
    training forward / backward are torch.ops.be.* operators as well; the product classes go through them (native.ops()), and
    BE_TORCH_OPS=0 (ctypes over the same C symbols) gives the same bits."""
    if os.environ.get("BE_TORCH_OPS", "1") == "0" or os.environ.get("BE_LIB_DIR"):
        pytest.skip("compares the two bindings: the torch-operator binding is switched off in this environment")
    native, synth, utils = env["native"], env["synth"], env["utils"]
    from be_hip import train_global_stage as tgs
    from oracle import global_loss as ogl
    for name in ("render_full", "fold_records", "local_loss", "local_loss_finish", "global_loss", "attention", "attention_train_fwd", "attention_bwd",
                 "etas2depth", "depth2sigma"):
        assert hasattr(native.ops(), name), name
    a = utils.get_args("global_train", argv=[])
    a.batch_size = 1
    hg = utils.PostProcessGlobalBase(a, DEV)
    hl = utils.PostProcessLocalBase(utils.get_args("local_train", argv=[]), DEV)
    dc = utils.DepthEtas(a, DEV)
    img = T(synth.synthetic_image_pair(147, 147)[0]).to(DEV)
    p12 = T(synth.plausible_params12(4096, name="bind12")).to(DEV)
    smp = {k: torch.from_numpy(v)[None].to(DEV) for k, v in synth.synthetic_global_sample(147, 147).items()}
    est = torch.from_numpy(synth.plausible_global_output(4096))[None].to(DEV)
    B_, S = 64, 5
    lest = T(synth.plausible_params10(B_, name="bindl")).to(DEV)
    limg = T(synth.f32(synth.hash_uniform(S, "b_img", (B_, 21, 21, 3)))).to(DEV)
    lbd = T(synth.f32(synth.hash_uniform(S, "b_bd", (B_, 21, 21)))).to(DEV)
    lde = T(synth.f32(synth.hash_uniform(S, "b_de", (B_, 19, 19, 3)))).to(DEV)
    qkv = (T(synth.f32(synth.hash_normal(S, "b_qkv", (2 * 256, 3 * 8 * 16)))) * 0.5).to(DEV)
    dout = T(synth.f32(synth.hash_normal(S, "b_dout", (2 * 256, 8 * 16)))).to(DEV)

    def both():
        opts = hg.render_opts(False)
        rec, ex = native.render_full(opts, dc.consts, 10.39, False, p12, native.view_image_pair(img), want=("patches", "depth_mask", "boundary"),
                                     pixels=img)
        out = [rec, ex["patches"], ex["depth_mask"], ex["boundary"]]
        out += list(native.fold_records(opts, rec, 64, 64, 147, 147).values())
        out += list(native.fold_records_batch(opts, torch.stack([rec, rec]), 64, 64, 147, 147, want=("image", "conf")).values())
        partial, grad, _ = native.local_loss(hl.render_opts(False), lest, limg, limg, lbd, lde, 1e-3, 5e-4)
        out += [partial, grad, native.local_loss_finish(partial, 1e-3, 5e-4).reshape(1)]
        e = est.clone().requires_grad_(True)
        loss = utils.global_loss(hg, dc, e, smp["img_gt"], smp["img_gt"], smp["bndry_dist"], smp["deri"], smp["bndry_depth"], ogl.GAMMA_FINAL)
        loss.backward()
        out += [loss.detach().reshape(1), e.grad]
        o, _ = native.attention(qkv, 2, 256, 8)
        of, lse, ws = tgs.attention_train_fwd(qkv, 2, 256, 8, 0.1, 1234)
        dq, _ = tgs.attention_bwd(qkv, of, lse, dout, 2, 256, 8, 0.1, 1234, ws=ws, operands_ready=True)
        out += [o, of, lse, dq]
        return [t.clone() for t in out]

    if os.environ.get("BE_TORCH_OPS", "1") == "0" or os.environ.get("BE_LIB_DIR"):
        pytest.skip("compares the two bindings: the torch-operator binding is switched off in this environment")
    assert native.ops() is not None
    via_ops = both()
    monkeypatch.setattr(native, "_ops", False)
    assert native.ops() is None
    via_ctypes = both()
    assert len(via_ops) == len(via_ctypes)
    for i, (x, y) in enumerate(zip(via_ops, via_ctypes)):
        assert x.shape == y.shape and torch.equal(x, y), i
