#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REAL reference on PyTorch-CPU.

Runs only in the build container (needs /root/reference, read-only).  Nothing here travels to the
GPU box except the .npz outputs: inputs and weights are regenerated there from
blurry-edges_amd/be_hip/synth.py (portable integer-hash generator), so the fixtures hold expected
OUTPUTS only (plus small inputs where that is cheaper than regenerating).

Import recipe: `utils/__init__.py` star-imports visualization.py which needs cv2
(absent, drawing only) -> register an empty stub module before importing.

usage: python tests/golden/make_golden.py [G1 G2 ...]      (default: all groups)

Lives next to the fixtures it writes (tests/golden/*.npz): it is test infrastructure - it imports the reference and, for G14, the
oracle's rasteriser stand-in - and nothing under blurry-edges_amd/ or bench.py uses it.
"""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(ROOT, "blurry-edges_amd"))
from be_hip import synth  # noqa: E402

# the reference's top-level packages are called models / utils / data, same as the build's:
# import the reference ones under their own names from REF, never mixing the two trees.
sys.path.insert(0, REF)
sys.modules.setdefault("cv2", types.ModuleType("cv2"))
for _m in ("models", "utils", "data"):
    assert _m not in sys.modules
import models as ref_models      # noqa: E402
import utils as ref_utils        # noqa: E402
assert ref_models.__file__.startswith(REF) and ref_utils.__file__.startswith(REF)
import blurry_edges_test as ref_test     # noqa: E402
import local_training as ref_local       # noqa: E402
import global_training as ref_global     # noqa: E402

torch.set_num_threads(8)


def ref_args(mode):
    argv, sys.argv = sys.argv, ["x"]
    try:
        return ref_utils.get_args(mode)
    finally:
        sys.argv = argv


def load_local_stage(seed=synth.SEED_DEFAULT):
    m = ref_models.LocalStage()
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in synth.local_stage_state_dict(seed).items()}
    m.load_state_dict(sd, strict=True)
    return m


def save(name, **arrs):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrs.items()})
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.1f} KiB)")


def n(t):
    return t.detach().cpu().numpy()


# ------------------------------------------------------------------------------------------------
def G1():
    """LocalStage eval: logits for 16 uniform patches + intermediate activations for 2."""
    m = load_local_stage().eval()
    x = torch.from_numpy(synth.uniform_patches(16))
    taps = {}
    pool_calls = []
    hooks = [m.conv1.register_forward_hook(lambda mod, i, o: taps.__setitem__("conv1", o)),
             m.maxpool1.register_forward_hook(lambda mod, i, o: pool_calls.append(o)),
             m.layer0.register_forward_hook(lambda mod, i, o: taps.__setitem__("layer0", o)),
             m.layer1.register_forward_hook(lambda mod, i, o: taps.__setitem__("layer1", o)),
             m.layer2.register_forward_hook(lambda mod, i, o: taps.__setitem__("layer2", o)),
             m.layer3.register_forward_hook(lambda mod, i, o: taps.__setitem__("layer3", o)),
             m.maxpool2.register_forward_hook(lambda mod, i, o: taps.__setitem__("pool3", o)),
             m.fc[3].register_forward_hook(lambda mod, i, o: taps.__setitem__("fc1", o))]
    with torch.no_grad():
        y = m(x)
    for h in hooks:
        h.remove()
    taps["pool1"], taps["pool2"] = pool_calls
    # a second input class: realistic blurred-edge pairs (8 pairs -> 16 patches)
    xs, _ = synth.synthetic_patch_pairs(8)
    with torch.no_grad():
        ys = m(torch.from_numpy(xs))
    m64 = load_local_stage().double().eval()
    with torch.no_grad():
        y64 = m64(x.double())
    save("g1_local_stage_eval", logits=n(y), logits_fp64=n(y64), logits_synth_pairs=n(ys),
         **{"tap_" + k: n(v[:2]) for k, v in taps.items() if k != "conv1"},
         tap_conv1_patch0=n(taps["conv1"][0]))


def G2():
    """LocalStage train mode (batch statistics), batch 64: logits, running-stat updates and gradients of
    representative parameters under loss = sum(logits * cotangent)."""
    m = load_local_stage().train()
    x = torch.from_numpy(synth.uniform_patches(64, name="train_patches"))
    ct = torch.from_numpy(synth.f32(synth.hash_normal(synth.SEED_DEFAULT, "train_cotangent", (64, 10))))
    x.requires_grad_(True)
    y = m(x)
    (y * ct).sum().backward()
    names = ["conv1.0.weight", "conv1.1.weight", "conv1.1.bias", "layer0.0.downsample.0.weight",
             "layer1.0.conv1.1.weight", "layer3.0.conv2.0.bias", "fc.1.bias", "fc.2.weight", "fc.4.weight"]
    params = dict(m.named_parameters())
    out = {"grad_" + k: n(params[k].grad) for k in names}
    # large grads: keep a strided subsample + the exact L2 norm
    for k in ["layer2.0.conv2.0.weight", "fc.1.weight"]:
        g = params[k].grad.flatten()
        out["gradsub_" + k] = n(g[::997])
        out["gradnorm_" + k] = n(g.double().norm())
    out["grad_x_sub"] = n(x.grad.flatten()[::101])
    out["total_grad_norm"] = n(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in m.parameters())))
    sd = m.state_dict()
    save("g2_local_stage_train", logits=n(y),
         run_mean_conv1=n(sd["conv1.1.running_mean"]), run_var_conv1=n(sd["conv1.1.running_var"]),
         run_mean_fc2=n(sd["fc.2.running_mean"]), run_var_fc2=n(sd["fc.2.running_var"]),
         nbt=n(sd["conv1.1.num_batches_tracked"]), **out)


def _local_helper(dtype, batch):
    a = ref_args("local_train")
    a.batch_size = batch
    h = ref_local.LocalLoss(a, torch.device("cpu"))
    if dtype == torch.float64:
        h.x, h.y, h.ridge = h.x.double(), h.y.double(), h.ridge.double()
        h.sobel_x, h.sobel_y = h.sobel_x.double(), h.sobel_y.double()
    return h


def G3():
    """Local-layout render stages for 8 patches, float32 and float64 runs of the reference code."""
    p10 = synth.plausible_params10(8)
    img = synth.uniform_patches(8, name="render_patches")          # [8,3,21,21]
    out = {}
    for tag, dt in (("f32", torch.float32), ("f64", torch.float64)):
        h = _local_helper(dt, 8)
        est = torch.from_numpy(p10).to(dt)
        y = torch.from_numpy(img).to(dt).permute(0, 2, 3, 1).contiguous()   # channels-last, as the dataset gives
        dists = h.params2dists(est[:, :8])
        etas = h.params2etas(est[:, 8:])
        wedges = h.dists2indicators(dists, etas)
        A = wedges.permute(0, 2, 3, 1).reshape(8, -1, 3)
        G = A.permute(0, 2, 1) @ A + h.ridge
        b = A.permute(0, 2, 1) @ y.view(8, -1, 3)
        inv = h.inverse_3by3(G)
        patches, bnd = h.get_patches(est.clone(), y)
        colors = (inv @ b).permute(0, 2, 1)
        out.update({f"{tag}_dists": n(dists), f"{tag}_etas": n(etas), f"{tag}_wedges": n(wedges),
                    f"{tag}_G": n(G), f"{tag}_b": n(b), f"{tag}_inv": n(inv), f"{tag}_colors": n(colors),
                    f"{tag}_patches": n(patches), f"{tag}_boundary": n(bnd)})
    save("g3_render_local", **out)


def G4():
    """LocalLoss value and d loss / d est for batch 64 at the final beta, float32 and float64."""
    B = 64
    S = synth.SEED_DEFAULT
    p10 = synth.plausible_params10(B, name="loss_params")
    img = synth.f32(synth.hash_uniform(S, "loss_img", (B, 21, 21, 3)))
    gt = synth.f32(synth.hash_uniform(S, "loss_gt", (B, 21, 21, 3)))
    bd = synth.f32(5.0 * synth.hash_uniform(S, "loss_bd", (B, 21, 21)))
    de = synth.f32(synth.hash_uniform(S, "loss_deri", (B, 19, 19, 3)))
    out = {}
    for tag, dt in (("f32", torch.float32), ("f64", torch.float64)):
        h = _local_helper(dt, B)
        h.final_beta()
        est = torch.from_numpy(p10).to(dt).requires_grad_(True)
        e2 = est * 1.0      # the reference mutates its input in place; keep the leaf intact
        loss = h(e2, torch.from_numpy(img).to(dt), torch.from_numpy(gt).to(dt),
                 torch.from_numpy(bd).to(dt), torch.from_numpy(de).to(dt))
        loss.backward()
        out[f"{tag}_loss"] = n(loss)
        out[f"{tag}_grad"] = n(est.grad)
    save("g4_local_loss", **out)


def G5():
    """etas2depth / depth2sigma on a dense eta grid over [1e-4,1]^2 (+ a log-spaced one)."""
    a = ref_args("eval")
    d = ref_utils.DepthEtas(a, torch.device("cpu"))
    lin = torch.linspace(1e-4, 1.0, 64)
    e1, e2 = torch.meshgrid(lin, lin, indexing="ij")
    z = d.etas2depth(e1, e2)
    lg = torch.logspace(-4, 0, 48)
    l1, l2 = torch.meshgrid(lg, lg, indexing="ij")
    zl = d.etas2depth(l1, l2)
    depth = torch.linspace(0.6, 2.3, 257)
    sg = d.depth2sigma(depth, a.rho_prime)
    sg1 = d.depth2sigma(depth, a.cam_params["rho_1"])
    save("g5_depth", z_lin=n(z), z_log=n(zl), sigma_rho_prime=n(sg), sigma_rho_1=n(sg1),
         consts=np.array([d.numerator, d.denominator_constant, d.denominator_factor_root,
                          d.denominator_factor, float(d.intercept)], dtype=np.float64),
         z_lin_f64=n(_depth64(d, e1.double(), e2.double())))


def _depth64(d, e1, e2):
    import copy
    d2 = copy.copy(d)
    d2.intercept = d.intercept.double()
    d2.theta_mid, d2.theta_wng = d.theta_mid.double(), d.theta_wng.double()
    return d2.etas2depth(e1, e2)


def G6():
    """Eval-time PostProcess (blurry_edges_test.py:12-100) on one synthetic 147x147 pair with a plausible
    12-parameter field: pass-A colours, pass-B per-patch outputs on a 4x4 sub-grid, six folded maps."""
    a = ref_args("eval")
    dev = torch.device("cpu")
    imgs, _ = synth.synthetic_image_pair(147, 147)
    t_img = torch.from_numpy(imgs)                                                # [2,3,147,147]
    out = {}
    for densify in (None, "w"):
        a.densify = densify
        dcal = ref_utils.DepthEtas(a, dev)
        helper = ref_test.PostProcess(a, dcal, dev)
        img_patches = torch.nn.Unfold(a.R, stride=a.stride)(t_img).view(2, 3, a.R, a.R, 64, 64)
        p10 = torch.from_numpy(np.stack([synth.plausible_params10(4096, name="g6_img1"),
                                         synth.plausible_params10(4096, name="g6_img2")]))   # [2,4096,10]
        colors = helper(p10, img_patches, colors_only=True)                       # [2,3,3,64,64]
        p12 = torch.from_numpy(synth.plausible_params12(4096, name="g6_est"))[None]   # [1,4096,12]
        etas = helper.params2etas(p12.permute(0, 2, 1).view(1, 12, 64, 64)[:, 8:])
        helper.img_patches = img_patches.unsqueeze(0)
        patches, shpd, refoc, bnd, dmap, dmask = helper.get_patches(
            p12.permute(0, 2, 1).view(1, 12, 64, 64)[:, :8], etas, False)
        folded = helper(p12, img_patches, colors_only=False)
        tag = "w_" if densify == "w" else ""
        sub = (slice(20, 24), slice(30, 34))
        out.update({
            tag + "colors_a": n(colors) if densify is None else np.zeros(0),
            tag + "sub_patches": n(patches[0][..., sub[0], sub[1]]),
            tag + "sub_shpd": n(shpd[0][..., sub[0], sub[1]]),
            tag + "sub_refoc": n(refoc[0][..., sub[0], sub[1]]),
            tag + "sub_bnd": n(bnd[0, 0][..., sub[0], sub[1]]),
            tag + "sub_dmap": n(dmap[0][..., sub[0], sub[1]]),
            tag + "sub_dmask": n(dmask[0][..., sub[0], sub[1]]),
            tag + "mask_hist": np.bincount(n(dmask).ravel(), minlength=3),
            tag + "fold_image": folded[0], tag + "fold_shpd": folded[1], tag + "fold_refoc": folded[2],
            tag + "fold_bndry": folded[3], tag + "fold_depth": folded[4], tag + "fold_conf": folded[5],
        })
    save("g6_postprocess_147", **out)


def G7():
    """Unfold ordering on an index image + Fold normaliser."""
    a = ref_args("eval")
    idx = torch.arange(3 * 147 * 147, dtype=torch.float32).view(1, 3, 147, 147)    # exact in fp32 (< 2^24)
    ip = torch.nn.Unfold(a.R, stride=a.stride)(idx).view(1, 3, 21, 21, 64, 64)
    vec = ip.permute(0, 4, 5, 1, 2, 3).reshape(4096, 3, 21, 21)
    sel = [0, 1, 63, 64, 2047, 4095]
    helper = ref_test.PostProcess(ref_args("eval"), None, torch.device("cpu"))
    save("g7_tiling", sel=np.array(sel), vec_sel=n(vec[sel]).astype(np.int32),
         vec_rowsum=n(vec.double().sum(dim=(1, 2, 3))), num_patches=n(helper.num_patches))


def G8():
    """Big-image tiler: which (block, local patch) lands on each of the 284x284 big-grid positions.
    The reference's depth_estimator (blurry_edges_test_big.py:116-189) is run as is on a 587x587 input with stub
    modules: the stub helper marks pixel (0,0) of every local patch with the code block*4096 + i*64 + j + 1, so the
    folded boundary map at image pixel (2I,2J), times the overlap count, is the code of big-grid patch (I,J)."""
    import shutil
    import tempfile
    sys.modules["cv2"].imwrite = lambda *a, **k: True
    import blurry_edges_test_big as ref_big
    argv, sys.argv = sys.argv, ["x"]
    try:
        a = ref_utils.get_args("eval", big=True)
    finally:
        sys.argv = argv
    a.cuda = "cpu"
    tmp = tempfile.mkdtemp(dir=os.path.join(ROOT, "gpurun_out") if os.path.isdir(os.path.join(ROOT, "gpurun_out")) else ROOT)
    a.log_path = tmp

    class Helper:
        device = torch.device("cpu")
        H_patches = W_patches = 64
        count = 0

        def __call__(self, params, img_patches, colors_only):
            if colors_only:
                return torch.zeros(2, 3, 3, 64, 64)
            code = (self.count * 4096 + torch.arange(4096, dtype=torch.float32).view(64, 64) + 1)
            self.count += 1
            bnd = torch.zeros(1, 1, 21, 21, 64, 64)
            bnd[0, 0, 0, 0] = code
            one = torch.zeros(1, 21, 21, 64, 64)
            one[0, 0, 0] = 1
            return (torch.zeros(1, 2, 3, 21, 21, 64, 64), torch.zeros(1, 3, 21, 21, 64, 64),
                    torch.zeros(1, 3, 21, 21, 64, 64), bnd, one.clone(), one)

    seen = {}

    class Vis:
        def visualize(self, i1, i2, c1, c2, shpd, refoc, conf, bndry, gt, depth):
            seen.update(conf=conf, bndry=bndry)
            return np.zeros((2, 2, 3), np.uint8)

    H, W = a.big_img_size
    loader = [(torch.zeros(1, 2, H, W, 3), torch.ones(1, H, W))]
    try:
        with np.errstate(all="ignore"):
            ref_big.depth_estimator(a, lambda v: torch.zeros(v.shape[0], 10), lambda pm: torch.zeros(1, 4096, 12),
                                    Helper(), Vis(), loader)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    hp = (H - a.R) // a.stride + 1
    val = seen["bndry"][0:2 * hp:2, 0:2 * hp:2].astype(np.float64) / seen["conf"][0:2 * hp:2, 0:2 * hp:2]
    code = np.rint(val).astype(np.int32)
    assert np.abs(val - code).max() < 0.1 and code.min() >= 1 and code.max() <= 36 * 4096
    save("g8_big_tiler", code=code, big_img_size=np.array([H, W]), n_margin_patch=np.array(a.n_margin_patch))


def G9():
    """GlobalStage eval on one [1,4096,38] feature tensor: strided subsample + checksums + state-dict key order."""
    m = ref_models.GlobalStage(in_parameter_size=38, out_parameter_size=12, device=torch.device("cpu"))
    keys = list(m.state_dict().keys())
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in synth.global_stage_state_dict().items()}
    assert sorted(keys) == sorted(sd.keys()), set(keys) ^ set(sd.keys())
    m.load_state_dict(sd, strict=True)
    m.eval()
    x = torch.from_numpy(synth.global_features())
    with torch.no_grad():
        y = m(x.clone())
    save("g9_global_stage", keys=np.array(keys), shapes=np.array([str(tuple(v.shape)) for v in m.state_dict().values()]),
         out_sub=n(y[0, ::37]), out_rowsum=n(y[0].double().sum(dim=1)), out_abs_mean=n(y.abs().mean()),
         pe_sub=n(m.positional_encoding.pe[0, ::61]))


def G11():
    """GlobalLoss (global_training.py:11-157) value, its seven terms and d loss / d est on one synthetic 147x147
    sample (batch 1) at the final gammas, float32 and float64 runs of the reference code."""
    a = ref_args("global_train")
    a.batch_size = 1
    smp = synth.synthetic_global_sample(147, 147)
    est_np = synth.plausible_global_output(4096)
    out = {}
    for tag, dt in (("f32", torch.float32), ("f64", torch.float64)):
        dcal = ref_utils.DepthEtas(a, torch.device("cpu"))
        crit = ref_global.GlobalLoss(a, dcal, torch.device("cpu"))
        crit.final_gamma()
        if dt == torch.float64:
            crit.x, crit.y, crit.ridge = crit.x.double(), crit.y.double(), crit.ridge.double()
            crit.sobel_x, crit.sobel_y, crit.num_patches = crit.sobel_x.double(), crit.sobel_y.double(), crit.num_patches.double()
            dcal.intercept, dcal.theta_mid, dcal.theta_wng = dcal.intercept.double(), dcal.theta_mid.double(), dcal.theta_wng.double()
        est = torch.from_numpy(est_np).to(dt)[None].requires_grad_(True)
        T = lambda k: torch.from_numpy(smp[k]).to(dt)[None]
        loss = crit(est, T("img_gt"), T("img_gt"), T("bndry_dist"), T("deri"), T("bndry_depth"))
        loss.backward()
        out[tag + "_loss"] = n(loss)
        out[tag + "_grad"] = n(est.grad[0])
    save("g11_global_loss", **out)


def G10():
    """eval_depth (utils/metrics.py:3-20) on a fixed pair of maps."""
    S = synth.SEED_DEFAULT
    pred = 0.7 + 0.6 * synth.hash_uniform(S, "m_pred", (1, 147, 147))
    gt = 0.75 + 0.43 * synth.hash_uniform(S, "m_gt", (1, 147, 147))
    msk = synth.hash_uniform(S, "m_msk", (1, 147, 147)) > 0.3
    pred = np.where(msk, pred, 0.0)
    r = ref_utils.eval_depth(pred.astype(np.float32), gt.astype(np.float32), pred > 0, crop=10)
    save("g10_metrics", metrics=np.array(r, dtype=np.float64))


def G12():
    """GlobalStage in TRAIN mode with every dropout probability set to 0: output and parameter gradients.
    (The reference's train-mode graph - nn.TransformerEncoder slow path + autograd - without the Philox masks no other
    generator can reproduce.)  B=2, L=256, loss = sum(out * R) with a fixed R; float64 and float32."""
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in synth.global_stage_state_dict().items()}
    B, L = 2, 256
    src = torch.from_numpy(synth.global_features(B * L, name="g12_src").reshape(B, L, 38))
    R = torch.from_numpy(synth.hash_normal(12, "g12_R", (B, L, 12)))
    out = {}
    for tag, dt in (("f64", torch.float64), ("f32", torch.float32)):
        m = ref_models.GlobalStage(in_parameter_size=38, out_parameter_size=12, device=torch.device("cpu"))
        m.load_state_dict(sd, strict=True)
        m = m.to(dt)
        m.positional_encoding.pe = m.positional_encoding.pe.to(dt)
        for mod in m.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0
            if isinstance(mod, torch.nn.MultiheadAttention):
                mod.dropout = 0.0
        m.train()
        y = m(src.to(dt))
        (y * R.to(dt)).sum().backward()
        out[tag + "_out"] = n(y)
        for k, prm in m.named_parameters():             # per parameter: L2 norm + a strided sample of <= 512 entries
            g = prm.grad.reshape(-1)
            out[f"{tag}_gnorm.{k}"] = n(g.double().norm())
            if tag == "f64":
                out[f"{tag}_gsample.{k}"] = n(g[::max(1, g.numel() // 512)])
    save("g12_global_stage_train", **out)


def G13():
    """DepthCompletion U-Net (eval) on a sparse synthetic depth map: output, bottleneck and first-level activations."""
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in synth.unet_state_dict().items()}
    x = torch.from_numpy(synth.sparse_depth_map())
    out = {}
    for tag, dt in (("f64", torch.float64), ("f32", torch.float32)):
        m = ref_models.DepthCompletion()
        m.load_state_dict(sd, strict=True)
        m = m.to(dt).eval()
        with torch.no_grad():
            xin = x.to(dt)
            x1 = m.inc(xin); x2 = m.down1(x1); x3 = m.down2(x2); x4 = m.down3(x3); x5 = m.down4(x4)
            u1 = m.up1(x5, x4)
            y = m(xin)
        out[tag + "_out"] = n(y).astype(np.float32) if tag == "f32" else n(y)
        if tag == "f64":
            out["f64_x1_sample"] = n(x1[0, ::8, ::6, ::6])
            out["f64_x5"] = n(x5[0, ::16]).astype(np.float32)
            out["f64_up1_sample"] = n(u1[0, ::16])
    out["keys"] = np.array(list(m.state_dict().keys()))
    save("g13_unet", **out)


def G14():
    """Synthetic-shape generator: the reference's generate_synthetic_image run on 3 replayed scenes.
    np.random is replayed from be_hip.datagen.draw_scenes' `raw` record (same draws, same order) and the three cv2 drawing
    calls are served by oracle.datagen.Cv2Stub (the build's rasterisation rule); dilation, PSF blur, compositing,
    distance transform and Sobel maps are the reference's own code.  Also DataGenerator's kernel helpers."""
    import importlib
    from be_hip import datagen as dg
    od = importlib.machinery.SourceFileLoader("oracle_datagen", os.path.join(ROOT, "oracle", "datagen.py")).load_module()
    stub = sys.modules["cv2"]
    for k in ("circle", "boxPoints", "drawContours", "imwrite"):
        setattr(stub, k, getattr(od.Cv2Stub, k))
    sys.modules.setdefault("tqdm", types.ModuleType("tqdm"))
    if not hasattr(sys.modules["tqdm"], "tqdm"):
        sys.modules["tqdm"].tqdm = lambda it, **k: it
    ref_gen = importlib.import_module("train_val_data_generator")
    argv, sys.argv = sys.argv, ["x"]
    try:
        a = ref_utils.get_args("data_gen_train_val")
    finally:
        sys.argv = argv
    gen = ref_gen.SyntheticShapeDataGenerator(a)
    scenes = dg.draw_scenes(3, seed=1869, img_size=tuple(a.img_size), num_shape=tuple(a.num_shape), z_range=tuple(a.Z_range))

    class Replay:
        def __init__(self, raw):
            self.q = [("randint", raw["bg"]), ("randint", raw["kind_col"]), ("uniform", raw["z"]), ("uniform", raw["ctr"])]
            # the reference sorts only the depths (far first); kind / colour / centre / per-object draws keep their
            # row, which is how draw_scenes pairs them
            self.per = list(raw["per_obj"])

        def randint(self, low, high=None, size=None):
            if self.q and self.q[0][0] == "randint":
                return self.q.pop(0)[1]
            return self.per.pop(0)

        def uniform(self, low, high=None, size=None):
            if self.q and self.q[0][0] == "uniform":
                return self.q.pop(0)[1]
            return self.per.pop(0)

    out = {}
    real_randint, real_uniform = np.random.randint, np.random.uniform
    for i in range(3):
        rp = Replay(scenes["raw"][i])
        np.random.randint, np.random.uniform = rp.randint, rp.uniform
        try:
            imgs, aif, bloc, idep, bdep, bdist, deri = gen.generate_synthetic_image(int(scenes["nobj"][i]))
        finally:
            np.random.randint, np.random.uniform = real_randint, real_uniform
        assert not rp.q and not rp.per
        out[f"imgs{i}"] = imgs.astype(np.uint8)
        out[f"aif{i}"] = aif.astype(np.uint8)
        out[f"bloc{i}"] = bloc.astype(np.uint8)
        out[f"idep{i}"], out[f"bdep{i}"] = idep.astype(np.float32), bdep.astype(np.float32)
        out[f"bdist{i}"] = bdist.astype(np.int16)
        out[f"deri{i}"] = deri.astype(np.float32)
    zs = np.array([0.75, 0.9, 1.0, 1.18])
    out["sigmas"] = np.stack([gen.get_kernel_sigma(z) for z in zs])
    out["kernel_s2"] = gen.get_blur_kernel(2.0)
    out["kernel_tiny"] = gen.get_blur_kernel(0.0)
    save("g14_datagen", **out)


def _as_double_global(helper, dcal):
    """the reference's eval-time helper with every constant tensor cast to float64 (as G11 does for GlobalLoss)"""
    helper.x, helper.y, helper.ridge = helper.x.double(), helper.y.double(), helper.ridge.double()
    helper.sobel_x, helper.sobel_y = helper.sobel_x.double(), helper.sobel_y.double()
    helper.num_patches = helper.num_patches.double()
    dcal.intercept, dcal.theta_mid, dcal.theta_wng = dcal.intercept.double(), dcal.theta_mid.double(), dcal.theta_wng.double()


def G15():
    """LocalStage -> GlobalStage feature glue, read out of the reference's OWN drivers run with stub modules:
    (a) depth_estimator (blurry_edges_test.py:102-145): stub local module (fixed raw params incl. unwrapped angles), stub
        helper (fixed pass-A colours [2,3,3,Hp,Wp] with distinct rgb/wedge values; records the `est` it is handed for pass B),
        stub global module (records `pm`, returns a fixed [1,P,12]); a 13 x 21 patch grid (45 x 61 image) so that a
        row/column transposition cannot pass;
    (b) ref_data_gen (global_data_pre_cal.py:10-33) with the same stub local module and the reference's real
        local-layout PostProcess cast to float64 -> params_src [1,2,P,19] (colours flattened from the LOCAL layout)."""
    import shutil
    import tempfile
    sys.modules["cv2"].imwrite = lambda *a, **k: True
    sys.modules.setdefault("tqdm", types.ModuleType("tqdm"))
    if not hasattr(sys.modules["tqdm"], "tqdm"):
        sys.modules["tqdm"].tqdm = lambda it, **k: it
    import global_data_pre_cal as ref_pre
    H, W = 45, 61
    hp, wp = (H - 21) // 2 + 1, (W - 21) // 2 + 1           # 13 x 21
    P = hp * wp
    S = synth.SEED_DEFAULT
    p10 = synth.glue_params10(P)                              # [2P,10], angles unwrapped by whole turns, both signs
    col = synth.f32(synth.hash_uniform(S, "g15_colors", (2, 3, 3, hp, wp)))
    y12 = synth.f32(-1.5 + 3.0 * synth.hash_uniform(S, "g15_global_out", (1, P, 12)))
    imgs, _ = synth.synthetic_image_pair(H, W)                # [2,3,H,W]
    rec = {}

    class Helper:
        device = torch.device("cpu")
        H_patches, W_patches = hp, wp

        def __call__(self, params, img_patches, colors_only):
            if colors_only:
                rec["params_a"] = params.clone()
                rec["img_patches_shape"] = np.array(img_patches.shape)
                return torch.from_numpy(col)
            rec["est"] = params.clone()
            z = lambda *s: np.full(s, 0.9, np.float32)
            return z(1, 2, 3, H, W), z(1, 3, H, W), z(1, 3, H, W), z(1, 1, H, W), z(1, H, W), z(1, H, W)

    class Vis:
        def visualize(self, *a):
            return np.zeros((2, 2, 3), np.uint8)

    def local_stub(vec):
        rec["vec_shape"] = np.array(vec.shape)
        return torch.from_numpy(p10).to(vec.dtype)

    def global_stub(pm):
        rec["pm"] = pm.clone()
        return torch.from_numpy(y12)

    a = ref_args("eval")
    tmp = tempfile.mkdtemp(dir=os.path.join(ROOT, "gpurun_out") if os.path.isdir(os.path.join(ROOT, "gpurun_out")) else ROOT)
    a.log_path, a.crop = tmp, 2
    loader = [(torch.from_numpy(imgs).permute(0, 2, 3, 1)[None].contiguous(), torch.ones(1, H, W))]
    try:
        with np.errstate(all="ignore"):
            ref_test.depth_estimator(a, local_stub, global_stub, None, Helper(), Vis(), loader)
        # (b) the second copy of the glue, with the real local-layout colour solve in float64
        b = ref_args("global_pre")
        b.img_size, b.data_path = [H, W], tmp
        hl = ref_pre.PostProcess(b, torch.device("cpu"))
        hl.x, hl.y, hl.ridge = hl.x.double(), hl.y.double(), hl.ridge.double()
        ref_pre.ref_data_gen(b, lambda vec: torch.from_numpy(p10).to(vec.dtype), hl,
                             [torch.from_numpy(imgs).double().permute(0, 2, 3, 1)[None].contiguous()], "g15")
        params_src = np.load(os.path.join(tmp, "params_src_g15.npy"))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    assert tuple(rec["vec_shape"]) == (2 * P, 3, 21, 21) and rec["pm"].shape == (1, P, 38) and rec["est"].shape == (1, P, 12)
    save("g15_glue", grid=np.array([H, W, hp, wp]), params_a=n(rec["params_a"]), pm=n(rec["pm"]), est=n(rec["est"]),
         params_src_f64=params_src)


def G16():
    """float64 run of the eval-time PostProcess of G6 (same inputs; helper tensors cast to double as in G11): pass-A
    colours, pass-B colours, the 4x4 sub-grid outputs and the six folded maps.  Stored as float32 (6e-8 relative - far
    below the 1e-4 they are compared at) except where noted."""
    a = ref_args("eval")
    dev = torch.device("cpu")
    imgs, _ = synth.synthetic_image_pair(147, 147)
    t_img = torch.from_numpy(imgs).double()
    img_patches = torch.nn.Unfold(a.R, stride=a.stride)(t_img).view(2, 3, a.R, a.R, 64, 64)
    p10 = torch.from_numpy(np.stack([synth.plausible_params10(4096, name="g6_img1"),
                                     synth.plausible_params10(4096, name="g6_img2")])).double()
    p12 = torch.from_numpy(synth.plausible_params12(4096, name="g6_est"))[None].double()
    sub = (slice(20, 24), slice(30, 34))
    f32 = lambda t: n(t).astype(np.float32) if torch.is_tensor(t) else np.asarray(t).astype(np.float32)
    out = {}
    for densify in (None, "w"):
        a.densify = densify
        dcal = ref_utils.DepthEtas(a, dev)
        helper = ref_test.PostProcess(a, dcal, dev)
        _as_double_global(helper, dcal)
        tag = "w_" if densify == "w" else ""
        helper.img_patches = img_patches.unsqueeze(0)
        est = p12.permute(0, 2, 1).view(1, 12, 64, 64)
        etas = helper.params2etas(est[:, 8:])
        patches, shpd, refoc, bnd, dmap, dmask = helper.get_patches(est[:, :8], etas, False)
        dists = helper.params2dists(est[:, :8])
        w1 = helper.dists2indicators(dists, etas[:, :2])
        w2 = helper.dists2indicators(dists, etas[:, 2:])
        colors_b = helper.get_colors(torch.cat([w1.unsqueeze(1), w2.unsqueeze(1)], dim=1), helper.img_patches, False)
        folded = helper(p12, img_patches, colors_only=False)
        if densify is None:
            out["colors_a"] = f32(helper(p10, img_patches, colors_only=True))          # [2,3,3,64,64]
            out["colors_b"] = f32(colors_b)                                            # [1,3,3,64,64]
            out.update(sub_patches=f32(patches[0][..., sub[0], sub[1]]), sub_shpd=f32(shpd[0][..., sub[0], sub[1]]),
                       sub_bnd=f32(bnd[0, 0][..., sub[0], sub[1]]),
                       fold_image=f32(folded[0]), fold_shpd=f32(folded[1]), fold_bndry=f32(folded[3]))
        out.update({tag + "sub_refoc": f32(refoc[0][..., sub[0], sub[1]]), tag + "sub_dmap": f32(dmap[0][..., sub[0], sub[1]]),
                    tag + "sub_dmask": n(dmask[0][..., sub[0], sub[1]]),
                    tag + "mask_hist": np.bincount(n(dmask).ravel(), minlength=3),
                    tag + "fold_refoc": f32(folded[2]), tag + "fold_depth": f32(folded[4]), tag + "fold_conf": f32(folded[5])})
    save("g16_postprocess_147_f64", **out)


def G17():
    """Big-image path end to end (blurry_edges_test_big.py:113-215): the reference's own depth_estimator on one 587 x 587 pair
    with its REAL big-image PostProcess (float64 inside the helper; the script stores patches in float32), a stub local module
    (fixed raw parameters per block) and a stub global module (fixed output per block): window extraction, the margin-dropping
    stitch into the 284 x 284 patch grid, the six folds and the confidence threshold, at full size.  Stored: every map
    subsampled by 7 in both directions + the row sums of the full maps."""
    import shutil
    import tempfile
    sys.modules["cv2"].imwrite = lambda *a, **k: True
    import blurry_edges_test_big as ref_big
    argv, sys.argv = sys.argv, ["x"]
    try:
        a = ref_utils.get_args("eval", big=True)
    finally:
        sys.argv = argv
    a.cuda = "cpu"
    tmp = tempfile.mkdtemp(dir=os.path.join(ROOT, "gpurun_out") if os.path.isdir(os.path.join(ROOT, "gpurun_out")) else ROOT)
    a.log_path = tmp
    dev = torch.device("cpu")
    dcal = ref_utils.DepthEtas(a, dev)
    helper = ref_big.PostProcess(a, dcal, dev)
    _as_double_global(helper, dcal)
    H, W = a.big_img_size
    imgs, _ = synth.synthetic_image_pair(H, W, nshape=14)
    count = dict(local=0, glob=0)

    def local_stub(vec):
        k = count["local"]
        count["local"] += 1
        return torch.from_numpy(synth.big_block_params10(k))

    def global_stub(pm):
        k = count["glob"]
        count["glob"] += 1
        return torch.from_numpy(synth.big_block_global_out(k))[None]

    seen = {}

    class Vis:
        def visualize(self, i1, i2, c1, c2, shpd, refoc, conf, bndry, gt, depth):
            seen.update(image=np.stack([c1.transpose(2, 0, 1), c2.transpose(2, 0, 1)]), shpd=shpd.transpose(2, 0, 1),
                        refoc=refoc.transpose(2, 0, 1), conf=conf, bndry=bndry, depth_map=depth)
            return np.zeros((2, 2, 3), np.uint8)

    loader = [(torch.from_numpy(imgs).double().permute(0, 2, 3, 1)[None].contiguous(), torch.ones(1, H, W))]
    try:
        with np.errstate(all="ignore"):
            ref_big.depth_estimator(a, local_stub, global_stub, helper, Vis(), loader)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    assert count["local"] == 36 and count["glob"] == 36
    out = {}
    for k, v in seen.items():
        v = np.asarray(v, dtype=np.float32)
        out[k + "_sub"] = v[..., ::7, ::7]
        out[k + "_rowsum"] = v.astype(np.float64).sum(axis=-1)
    save("g17_big_image", **out)


def G18():
    """The reference's local training loop, free-running: local_training.py:99-108 for 20 steps at batch 64 (LocalStage in train mode with
    batch statistics, LocalLoss at the final betas, clip_grad_norm_(1), AdamW(lr 6e-5)) on the portable synthetic training patches,
    float32 and float64 runs: loss and total gradient norm of every step, final parameters (subsampled) and running statistics."""
    B, STEPS = 64, 20
    data = synth.synthetic_training_patches(B * STEPS, seed=1871)
    a = ref_args("local_train")
    out = {}
    for tag, dt in (("f32", torch.float32), ("f64", torch.float64)):
        torch.manual_seed(0)
        m = load_local_stage().to(dt).train()
        crit = _local_helper(dt, B)
        crit.final_beta()
        opt = torch.optim.AdamW(m.parameters(), lr=a.learning_rate)
        losses, norms = [], []
        for it in range(STEPS):
            b = {k: torch.from_numpy(v[it * B:(it + 1) * B]).to(dt) for k, v in data.items()}
            est = m(b["img_ny"].permute(0, 3, 1, 2))
            opt.zero_grad()
            loss = crit(est, b["img_gt"], b["img_gt"], b["bndry_dist"], b["deri"])       # local_training.py:105 passes the clean image twice
            loss.backward()
            norms.append(float(torch.nn.utils.clip_grad_norm_(m.parameters(), max_norm=1, norm_type=2)))
            opt.step()
            losses.append(float(loss))
            print(tag, it, losses[-1], norms[-1], flush=True)
        sd = m.state_dict()
        out[tag + "_loss"] = np.asarray(losses)
        out[tag + "_grad_norm"] = np.asarray(norms)
        for k in ("conv1.0.weight", "fc.4.weight", "fc.4.bias", "conv1.1.running_mean", "conv1.1.running_var", "fc.2.running_mean",
                  "fc.2.running_var", "layer0.0.conv2.1.weight"):
            out[f"{tag}_final_{k}"] = n(sd[k])
        for k in ("layer2.0.conv2.0.weight", "fc.1.weight"):
            out[f"{tag}_finalsub_{k}"] = n(sd[k].flatten()[::997])
    save("g18_local_training_trajectory", **out)


GROUPS = dict(G1=G1, G2=G2, G3=G3, G4=G4, G5=G5, G6=G6, G7=G7, G8=G8, G9=G9, G10=G10, G11=G11, G12=G12, G13=G13, G14=G14, G15=G15, G16=G16, G17=G17, G18=G18)

if __name__ == "__main__":
    todo = sys.argv[1:] or list(GROUPS)
    for g in todo:
        print("==", g, GROUPS[g].__doc__.strip().splitlines()[0])
        GROUPS[g]()
