"""GPU parity tests for the full render pass, the fold, the tiling glue and the end-to-end pipeline
(BASELINE.json configs[3]: 147x147 image tiled into overlapping patches + GlobalStage aggregation)."""
import numpy as np
import pytest
import torch

from conftest import load_golden, relmax
from be_hip import synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def T(a, dt=torch.float32):
    return torch.from_numpy(np.asarray(a)).to(dt)


@pytest.fixture(scope="module")
def env():
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test run without a GPU")
    import utils
    from be_hip import native
    native.lib()
    a = utils.get_args("eval", argv=[])
    return dict(native=native, args=a, helper=utils.PostProcessGlobalBase(a, DEV), dcal=utils.DepthEtas(a, DEV))


def test_unfold_order_bit_exact(env):
    g = load_golden("g7_tiling")
    idx = torch.arange(3 * 147 * 147, dtype=torch.float32).view(1, 3, 147, 147)
    vec = env["native"].unfold_patches(idx.to(DEV))[0].cpu()
    assert np.array_equal(vec[g["sel"]].numpy().astype(np.int32), g["vec_sel"])
    assert np.array_equal(vec.double().sum(dim=(1, 2, 3)).numpy(), g["vec_rowsum"])


def test_feature_glue_vs_reference_golden(env):
    """Row a18: the HIP glue kernels against what the reference's own depth_estimator / ref_data_gen produced when run
    with stub modules (golden g15, tests/golden/make_golden.py:G15) - not against the oracle."""
    from be_hip.pipeline import DepthPipeline, params_src_layout
    n = env["native"]
    g = load_golden("g15_glue")
    H, W, hp, wp = (int(v) for v in g["grid"])
    P, S = hp * wp, synth.SEED_DEFAULT
    p10 = T(synth.glue_params10(P)).to(DEV)                                              # [2P,10], unwrapped angles
    col = T(synth.f32(synth.hash_uniform(S, "g15_colors", (2, 3, 3, hp, wp))))           # reference layout
    colp = col.permute(0, 3, 4, 1, 2).reshape(2 * P, 3, 3).contiguous().to(DEV)          # the layout be_render_colors emits
    pm = n.local_features(p10, colp)
    # fp32 on both sides; the kernel multiplies by 1/3 and 1/pi where torch divides: 1 ulp
    assert pm.shape == (P, 38) and relmax(pm.cpu(), g["pm"][0]) <= 2e-7
    y12 = T(synth.f32(-1.5 + 3.0 * synth.hash_uniform(S, "g15_global_out", (1, P, 12))))[0].to(DEV)
    est = n.global_denorm(y12).cpu().numpy()
    ref = g["est"][0]
    # remainder(., 2 pi) may land on the other side of the wrap for an angle within an ulp of a multiple of 2 pi
    d = np.abs(est - ref)
    d[:, 4:8] = np.minimum(d[:, 4:8], np.abs(d[:, 4:8] - 2 * np.pi))
    assert float(d.max()) <= 2e-6
    # the whole local pass as be_hip.workflow.global_pre runs it (gather-on-read colours on the 13 x 21 grid), with the
    # golden's fixed "CNN" output in place of LocalStage -> params_src [2,P,19]; colours are a ridge solve: 1e-4 vs fp64
    imgs, _ = synth.synthetic_image_pair(H, W)

    class FixedLocal:
        def forward_image_pair(self, img, stride, window=None):
            return p10
    pipe = DepthPipeline(FixedLocal(), None, env["helper"], None)
    src = params_src_layout(pipe.local_pass(T(imgs).to(DEV))[3]).cpu().numpy()
    ref = g["params_src_f64"][0]
    assert src.shape == ref.shape == (2, P, 19)
    assert relmax(src[..., :10], ref[..., :10]) <= 2e-7
    assert relmax(src[..., 10:], ref[..., 10:]) <= 1e-4


def _pass_b(env, densify, want):
    n = env["native"]
    imgs, _ = synth.synthetic_image_pair(147, 147)
    img = T(imgs).to(DEV)
    p12 = T(synth.plausible_params12(4096, name="g6_est")).to(DEV)
    opts = env["helper"].render_opts(False)
    rec, ex = n.render_full(opts, env["dcal"].consts, 10.39, densify == "w", p12, n.view_image_pair(img), want=want)
    return img, p12, opts, rec, ex


@pytest.mark.parametrize("densify,tag", [(None, ""), ("w", "w_")])
def test_pass_b_per_patch_outputs_vs_golden_and_fp64_oracle(env, densify, tag):
    from oracle import render as orr, depth as od, tiling as ot
    g = load_golden("g6_postprocess_147")
    img, p12, opts, rec, ex = _pass_b(env, densify, ("patches", "shpd", "refoc", "boundary", "depth_map", "depth_mask"))
    ii, jj = np.meshgrid(np.arange(20, 24), np.arange(30, 34), indexing="ij")
    sel = (ii * 64 + jj).ravel()

    def sub(key):
        a = g[tag + key]
        return np.moveaxis(a.reshape(a.shape[:-2] + (16,)), -1, 0)
    c = lambda k: ex[k].cpu().numpy()
    # hard decisions: identical to the reference on identical inputs
    assert np.array_equal(c("depth_mask")[sel], sub("sub_dmask"))
    assert np.array_equal(np.bincount(c("depth_mask").ravel(), minlength=3), g[tag + "mask_hist"])
    assert relmax(c("depth_map")[sel], sub("sub_dmap")) <= 1e-6
    assert relmax(c("boundary")[sel], sub("sub_bnd")) <= 1e-5
    # colour-dependent outputs against the float64 oracle
    pat = ot.unfold_patches(img.cpu())
    r64 = orr.render_pass_b(od.depth_consts(), p12.cpu().double(), pat[0].double(), pat[1].double(), densify=densify)
    assert relmax(rec[:, 20:29].cpu(), r64["colors"].reshape(-1, 9)) <= 1e-4
    assert relmax(c("patches"), torch.stack([r64["patches1"], r64["patches2"]], dim=1)) <= 1e-4
    # sharpened render: eta = 1e-4 turns erf into a step, so a pixel within ~1e-4 of an edge is sensitive to the
    # fp32 rounding of its own distance (App. C: wedges differ by up to 3e-4 between fp32 and fp64 reference runs)
    es = np.abs(c("shpd") - r64["shpd"].numpy()) / float(r64["shpd"].abs().max())
    assert float((es > 1e-4).mean()) <= 1e-4 and float(es.max()) <= 5e-3
    # refocus: eta_refoc = depth2sigma(fp32 depth) can be as small as ~1e-3 near the refocus plane: same sensitivity
    er = np.abs(c("refoc") - r64["refoc"].numpy()) / float(r64["refoc"].abs().max())
    assert float((er > 1e-4).mean()) <= 1e-5 and float(er.max()) <= 1e-3
    assert relmax(rec[:, 29:31].cpu(), torch.stack([r64["depth1"], r64["depth2"]], dim=1)) <= 1e-6
    # the reference's own fp32 patches are within its fp32-vs-fp64 noise of ours
    assert relmax(c("patches")[sel], sub("sub_patches")) <= 2e-2
    # ... and the float64 run of the REFERENCE code itself (golden g16) pins every colour-dependent output at 1e-4
    g64 = load_golden("g16_postprocess_147_f64")

    def sub64(key):
        a = g64[key]
        return np.moveaxis(a.reshape(a.shape[:-2] + (16,)), -1, 0)
    assert relmax(rec[:, 20:29].cpu(), g64["colors_b"][0].reshape(3, 3, 4096).transpose(2, 0, 1).reshape(4096, 9)) <= 1e-4
    assert relmax(c("patches")[sel], sub64("sub_patches")) <= 1e-4
    assert relmax(c("boundary")[sel], sub64("sub_bnd")) <= 1e-5
    assert relmax(c("depth_map")[sel], sub64(tag + "sub_dmap")) <= 1e-6
    assert np.array_equal(c("depth_mask")[sel], sub64(tag + "sub_dmask"))
    assert np.array_equal(np.bincount(c("depth_mask").ravel(), minlength=3), g64[tag + "mask_hist"])
    e = np.abs(c("shpd")[sel] - sub64("sub_shpd")) / float(np.abs(sub64("sub_shpd")).max())
    assert float((e > 1e-4).mean()) <= 1e-3 and float(e.max()) <= 5e-3
    e = np.abs(c("refoc")[sel] - sub64(tag + "sub_refoc")) / float(np.abs(sub64(tag + "sub_refoc")).max())
    assert float((e > 1e-4).mean()) <= 1e-3 and float(e.max()) <= 1e-3


@pytest.mark.parametrize("densify,tag", [(None, ""), ("w", "w_")])
def test_fold_maps_vs_golden(env, densify, tag):
    g = load_golden("g6_postprocess_147")
    n = env["native"]
    img, p12, opts, rec, _ = _pass_b(env, densify, ())
    m = n.fold_records(opts, rec, 64, 64, 147, 147, 2, densify == "w")
    c = lambda k: m[k].cpu().numpy()
    assert relmax(c("conf"), g[tag + "fold_conf"][0]) <= 1e-6
    assert relmax(c("depth"), g[tag + "fold_depth"][0]) <= 1e-5
    assert relmax(c("bndry"), g[tag + "fold_bndry"][0, 0]) <= 1e-5
    # colour maps: the reference's fp32 Cayley-Hamilton noise averages down in the fold (App. C: 3.3e-4)
    assert relmax(c("image"), g[tag + "fold_image"][0]) <= 5e-3
    assert relmax(c("shpd"), g[tag + "fold_shpd"][0]) <= 1e-2
    assert relmax(c("refoc"), g[tag + "fold_refoc"][0]) <= 5e-3
    # tighter: the float64 run of the REFERENCE (golden g16): every folded map at 1e-4 or better
    g64 = load_golden("g16_postprocess_147_f64")
    assert relmax(c("image"), g64["fold_image"][0]) <= 1e-4
    assert relmax(c("shpd"), g64["fold_shpd"][0]) <= 1e-4
    assert relmax(c("refoc"), g64[tag + "fold_refoc"][0]) <= 1e-4
    assert relmax(c("bndry"), g64["fold_bndry"][0, 0]) <= 1e-5
    assert relmax(c("depth"), g64[tag + "fold_depth"][0]) <= 1e-5
    assert relmax(c("conf"), g64[tag + "fold_conf"][0]) <= 1e-6
    # and the float64 oracle fold of the float64 oracle render (itself pinned by g16 to 2e-7)
    from oracle import render as orr, depth as od, tiling as ot
    pat = ot.unfold_patches(img.cpu())
    r64 = orr.render_pass_b(od.depth_consts(), p12.cpu().double(), pat[0].double(), pat[1].double(), densify=densify)
    fi = ot.fold_mean(torch.stack([r64["patches1"], r64["patches2"]]), 147, 147)
    assert relmax(c("image"), fi) <= 1e-4
    assert relmax(c("refoc"), ot.fold_mean(r64["refoc"][None], 147, 147)[0]) <= 1e-4
    # a view over materialised patches gives the same records as the image view (gather-on-read is layout-blind)
    pats = n.unfold_patches(img)
    rec2, _ = n.render_full(opts, env["dcal"].consts, 10.39, densify == "w", p12, n.view_flat_patches(pats, 64))
    assert torch.equal(rec, rec2)
    unf = pats.view(2, 64, 64, 3, 21, 21).permute(0, 3, 4, 5, 1, 2).contiguous()     # reference layout [2,3,21,21,Hp,Wp]
    rec3, _ = n.render_full(opts, env["dcal"].consts, 10.39, densify == "w", p12, n.view_unfolded(unf))
    assert torch.equal(rec, rec3)


def test_global_stage_on_gpu_vs_golden(env):
    import models
    g = load_golden("g9_global_stage")
    m = models.GlobalStage(device=DEV).to(DEV).eval()
    assert list(m.state_dict().keys()) == list(g["keys"])
    m.load_state_dict({k: T(v) for k, v in synth.global_stage_state_dict().items()}, strict=True)
    x = T(synth.global_features()).to(DEV)
    with torch.no_grad():
        y = m(x.clone())
    assert relmax(y[0, ::37].cpu(), g["out_sub"]) <= 1e-4
    assert relmax(y[0].double().sum(dim=1).cpu(), g["out_rowsum"]) <= 1e-4


def _pipeline(env, densify=None):
    import models, utils
    from be_hip.pipeline import DepthPipeline
    lm = models.LocalStage()
    lm.load_state_dict({k: T(v) for k, v in synth.local_stage_state_dict().items()})
    gm = models.GlobalStage(device=DEV)
    gm.load_state_dict({k: T(v) for k, v in synth.global_stage_state_dict().items()})
    return DepthPipeline(lm.to(DEV).eval(), gm.to(DEV).eval(), env["helper"], env["dcal"], densify=densify)


def test_gather_on_read_local_pass_is_bit_identical_to_the_unfolded_one(env):
    """LocalStage + pass A reading their windows from the image pair (whole image, and a block window of
    a larger image) must give exactly what they give on the materialised nn.Unfold tensor."""
    n = env["native"]
    pipe = _pipeline(env)
    big = T(synth.synthetic_image_pair(235, 235, nshape=8)[0]).to(DEV)
    for win in (None, (88, 0, 147, 147), (88, 88, 147, 147)):
        img = big[:, :, :147, :147].contiguous() if win is None else big
        crop = img if win is None else big[:, :, win[0]:win[0] + 147, win[1]:win[1] + 147].contiguous()
        flat = n.unfold_patches(crop).view(-1, 3, 21, 21)
        est_ref = pipe.local(flat)
        col_ref, ex_ref = pipe.helper.render_colors(est_ref, flat, wrap_angles=True, want=("boundary", "aty"))
        view, est10, colors, _ = pipe.local_pass(img, win)
        assert torch.equal(est10, est_ref) and torch.equal(colors, col_ref)
        _, ex = n.render_colors_view(pipe.helper.render_opts(True), est10, view, 4096, want=("boundary", "aty"))
        assert torch.equal(ex["boundary"], ex_ref["boundary"]) and torch.equal(ex["aty"], ex_ref["aty"])
        x4 = torch.empty(100, 21, 21, 4, device=DEV)
        n.check(n.lib().be_view_to_nhwc4_f32(n.C.byref(view), 4096, 4090, n.dptr(x4), 100, n.stream_ptr(x4.device)), "stage")
        assert torch.equal(x4[..., :3], flat[4090:4190].permute(0, 2, 3, 1)) and not x4[..., 3].any()
    with pytest.raises(RuntimeError):
        n.view_image_pair(big, 2, (100, 100, 147, 147))                  # window leaves the image


def test_pipeline_147_stage_by_stage_vs_oracle(env):
    """configs[3]: every stage is checked against the oracle fed with the HIP output of the stage before."""
    import models
    from oracle import render as orr, depth as od, tiling as ot, glue, local_stage as ols
    pipe = _pipeline(env)
    imgs, _ = synth.synthetic_image_pair(147, 147)
    img = T(imgs).to(DEV)
    out = pipe(img)
    pat = ot.unfold_patches(img.cpu())                               # [2,4096,3,21,21]
    est10 = out["est10"].cpu()
    # CNN on a sample of the 8192 patches
    idx = torch.arange(0, 8192, 131)
    ref = ols.local_stage_forward(ols.to_torch_sd(synth.local_stage_state_dict()), pat.reshape(-1, 3, 21, 21)[idx])
    assert relmax(est10[idx], ref) <= 1e-5
    # pass A colours from the HIP logits
    col64 = orr.render_pass_a(orr.wrap_angles10(est10).double(), pat.reshape(-1, 3, 21, 21).double())["colors"]
    assert relmax(out["colors_a"].cpu(), col64) <= 1e-4
    # features + transformer (same module on the CPU) + de-normalisation
    pm = glue.local_features(est10.view(2, 4096, 10), out["colors_a"].cpu().view(2, 4096, 3, 3))
    gm_cpu = models.GlobalStage(device="cpu")
    gm_cpu.load_state_dict({k: T(v) for k, v in synth.global_stage_state_dict().items()})
    with torch.no_grad():
        y = gm_cpu.eval()(pm[None].clone())[0]
    est12 = glue.global_denorm(y)
    d = (out["est12"].cpu() - est12).abs()
    d[:, 4:8] = torch.minimum(d[:, 4:8], 2 * torch.pi - d[:, 4:8])   # angles live on a circle
    assert float(d.max()) <= 2e-3 and float(d.median()) <= 2e-5
    # pass B + fold from the HIP est12
    r64 = orr.render_pass_b(od.depth_consts(), out["est12"].cpu().double(), pat[0].double(), pat[1].double())
    fi = ot.fold_mean(torch.stack([r64["patches1"], r64["patches2"]]), 147, 147)
    assert relmax(out["image"].cpu(), fi) <= 1e-4
    fd, conf = ot.fold_depth(r64["depth_map"][None], r64["depth_mask"][None], 147, 147)
    flips = (out["conf"].cpu() - conf[0]).abs() > 1e-6
    assert float(flips.float().mean()) <= 1e-3                        # branch-flipped pixels are counted
    ok = ~flips
    rmse = float(torch.sqrt(((out["depth"].cpu() - fd[0])[ok] ** 2).mean()))
    assert rmse <= 1e-4, rmse
    thr = torch.where(conf[0] > 0.05, fd[0], torch.zeros_like(fd[0]))
    assert float(((out["depth_map"].cpu() - thr).abs() > 1e-4).float().mean()) <= 1e-3


def test_big_image_tiler_and_fold_587(env):
    """587x587: 36 blocks through the pipeline; the stitched record grid folded on the GPU must equal the oracle
    fold of the oracle render of the same stitched parameters (tiler windows from oracle.tiling.big_tiler)."""
    from oracle import render as orr, depth as od, tiling as ot
    pipe = _pipeline(env)
    imgs, _ = synth.synthetic_image_pair(587, 587, nshape=14)
    img = T(imgs).to(DEV)
    maps = pipe.run_big(img)
    assert tuple(maps["image"].shape) == (2, 3, 587, 587) and torch.isfinite(maps["image"]).all()
    # rebuild the stitched parameter grid block by block with the oracle's window table
    t = ot.big_tiler()
    big12 = torch.zeros(284, 284, 12)
    for bi, bj, top, left, (vs, ve, hs, he), (Vs, Ve, Hs, He) in t["blocks"]:
        b = img[:, :, top:top + 147, left:left + 147].contiguous()
        _, _, _, pm = pipe.local_pass(b)
        est12 = pipe.global_pass(pm).cpu().view(64, 64, 12)
        big12[Vs:Ve, Hs:He] = est12[vs:ve, hs:he]
    pat = ot.unfold_patches(img.cpu())
    # oracle render on a horizontal band of the big grid (rows 100..139) keeps the CPU cost bounded
    rows = slice(100, 140)
    sel = (torch.arange(284 * 284).view(284, 284)[rows]).reshape(-1)
    r = orr.render_pass_b(od.depth_consts(), big12.view(-1, 12)[sel].double(), pat[0][sel].double(), pat[1][sel].double())
    # pixels covered ONLY by patches of that band: image rows 2*100+20 .. 2*139  -> [220, 278]
    full = torch.zeros(2, 284 * 284, 3, 21, 21, dtype=torch.float64)
    full[0, sel], full[1, sel] = r["patches1"], r["patches2"]
    fi = ot.fold_mean(full, 587, 587)
    assert relmax(maps["image"].cpu()[:, :, 220:279], fi[:, :, 220:279]) <= 1e-4


def test_big_image_deduplicated_local_pass_is_bit_identical_to_the_block_schedule(env):
    """run_big(dedup=True) runs LocalStage + pass A + the feature glue once per DISTINCT 21x21 window of the 587x587 pair
    (161 312 patches) and every block gathers its rows; the reference's schedule (blurry_edges_test_big.py:142-165, dedup=False)
    runs them once per block (294 912 patches).  Per-patch kernels, position-independent bit for bit: all maps must be EQUAL,
    on one stream and on the two-stream schedule."""
    pipe = _pipeline(env)
    imgs, _ = synth.synthetic_image_pair(587, 587, nshape=14)
    img = T(imgs).to(DEV)
    ref = pipe.run_big(img, dedup=False)
    for streams in (1, 2):
        pipe.local.streams = streams
        got = pipe.run_big(img)
        for k in ("image", "shpd", "refoc", "bndry", "depth", "conf", "depth_map"):
            assert torch.equal(got[k], ref[k]), (k, streams)
    pipe.local.streams = 2


def test_big_image_path_matches_the_reference_run_g17(env):
    """Golden g17: the reference's own big-image depth_estimator (blurry_edges_test_big.py:113-215; its PostProcess in float64,
    stub networks with fixed outputs per block) on one 587 x 587 pair.  run_big with the same stub outputs must reproduce its
    six folded maps: window extraction, the margin-dropping stitch, pass B, the folds and the 0.05 confidence threshold, full size."""
    from be_hip.pipeline import DepthPipeline
    g = load_golden("g17_big_image")

    class FixedLocal:
        k = 0

        def forward_image_pair(self, img, stride, window):
            est = T(synth.big_block_params10(self.k)).to(DEV)
            self.k += 1
            return est

    class FixedGlobal:
        k = 0

        def __call__(self, pm):
            ys = [T(synth.big_block_global_out(self.k + i)) for i in range(pm.shape[0])]
            self.k += pm.shape[0]
            return torch.stack(ys).to(DEV)

    pipe = DepthPipeline(FixedLocal(), FixedGlobal(), env["helper"], env["dcal"])
    imgs, _ = synth.synthetic_image_pair(587, 587, nshape=14)
    # the stub local module returns fixed outputs PER BLOCK (as the golden's stub did), so this run takes the reference's own
    # schedule, one local pass per block; the de-duplicated default is pinned to it bit for bit by the test below
    maps = pipe.run_big(T(imgs).to(DEV), dedup=False)
    assert pipe.local.k == 36 and pipe.globl.k == 36
    got = {k: maps[k].double().cpu() for k in ("image", "shpd", "refoc", "bndry", "conf", "depth_map")}
    for k in ("image", "shpd", "refoc", "bndry"):                       # fp32 kernels against the fp64 reference helper
        assert relmax(got[k][..., ::7, ::7], g[k + "_sub"]) <= 1e-4, k
        assert relmax(got[k].sum(-1), g[k + "_rowsum"]) <= 1e-4, k
    # confidence: counts of depth-mask elements / 121; an element on its threshold may flip in fp32 (counted)
    flips = (got["conf"][::7, ::7] - T(g["conf_sub"], torch.float64)).abs() > 1e-6
    assert float(flips.float().mean()) <= 2e-3, float(flips.float().mean())
    dm = (got["depth_map"][::7, ::7] - T(g["depth_map_sub"], torch.float64)).abs()
    assert float((dm[~flips] ** 2).mean().sqrt()) <= 1e-4
    assert float((dm > 1e-3).float().mean()) <= 2e-3
    assert relmax(got["conf"].sum(-1), g["conf_rowsum"]) <= 2e-3


def test_base_class_methods_both_layouts_vs_oracle(env):
    """The reference's fine-grained PostProcess methods (what a subclass written against the reference calls)."""
    import utils
    from oracle import render as orr, tiling as ot, depth as od
    h = env["helper"]                                             # PostProcessGlobalBase, batch_size 1
    p12 = T(synth.plausible_params12(4096, name="g6_est"))
    est = p12.t().reshape(1, 12, 64, 64).contiguous().to(DEV)     # reference global layout [B,12,Hp,Wp]
    dists = h.params2dists(est[:, :8])
    assert tuple(dists.shape) == (1, 2, 21, 21, 64, 64)
    d_or = orr.params2dists(p12[:, :8])                           # [P,2,21,21]
    assert relmax(dists[0].permute(3, 4, 0, 1, 2).reshape(4096, 2, 21, 21).cpu(), d_or) <= 1e-5
    etas = h.params2etas(est[:, 8:10].contiguous())
    wed = h.dists2indicators(dists, etas)
    w_or = orr.dists2indicators(d_or.double(), orr.params2etas(p12[:, 8:10].double()))
    assert relmax(wed[0].permute(3, 4, 0, 1, 2).reshape(4096, 3, 21, 21).cpu(), w_or) <= 3e-4
    # local layout
    a = utils.get_args("local_train", argv=[])
    hl = utils.PostProcessLocalBase(a, DEV)
    dl = hl.params2dists(p12[:64, :8].to(DEV))
    assert relmax(dl.cpu(), d_or[:64]) <= 1e-5
    # inverse_3by3 on ridge-regularised Gram matrices, derivative of an image
    r = orr.render_pass_a(T(synth.plausible_params10(64)), T(synth.uniform_patches(64)))
    inv = hl.inverse_3by3(r["G"].to(DEV))
    assert relmax(inv.cpu(), torch.linalg.inv(r["G"].double())) <= 1e-5
    img = T(synth.uniform_patches(8, name="deriv"))
    assert relmax(hl.get_image_derivative(img.to(DEV)).cpu(), orr.image_derivative(img)) <= 1e-5
    # folds of materialised tensors in the reference layout
    imgs, _ = synth.synthetic_image_pair(147, 147)
    pat = ot.unfold_patches(T(imgs))                              # [2,P,3,21,21]
    ref_layout = pat.view(2, 64, 64, 3, 21, 21).permute(0, 3, 4, 5, 1, 2).contiguous().to(DEV)   # [2,3,21,21,Hp,Wp]
    back = h.local2global_color(ref_layout.unsqueeze(0))          # fold(unfold(x)) / count == x
    assert relmax(back[0].cpu(), T(imgs)) <= 1e-5
    assert torch.equal(h.num_patches.cpu(), ot.num_patches(147, 147))
    rb = orr.render_pass_b(od.depth_consts(), p12, pat[0], pat[1])
    dm = rb["depth_map"].view(64, 64, 21, 21).permute(2, 3, 0, 1)[None].contiguous().to(DEV)
    mk = rb["depth_mask"].view(64, 64, 21, 21).permute(2, 3, 0, 1)[None].contiguous().to(DEV)
    depth, conf = h.local2global_depth(dm, mk)
    fd, fc = ot.fold_depth(rb["depth_map"][None], rb["depth_mask"][None], 147, 147)
    assert relmax(conf.cpu(), fc) <= 1e-6 and relmax(depth.cpu(), fd) <= 1e-5
    bp = rb["boundary"].view(64, 64, 21, 21).permute(2, 3, 0, 1)[None, None].contiguous().to(DEV)
    assert relmax(h.local2global_bndry(bp).cpu()[0, 0], ot.fold_mean(rb["boundary"][None, :, None], 147, 147)[0, 0]) <= 1e-5


def test_attention_and_layernorm_kernels_vs_torch(env):
    """be_attention_f32 / be_add_layernorm_f32 against plain PyTorch fp32 (and fp64) on the CPU."""
    n = env["native"]
    B, L, H = 2, 256, 8
    g = torch.Generator().manual_seed(3)
    qkv = torch.randn(B * L, 3 * H * 16, generator=g) * 1.5
    out, _ = n.attention(qkv.to(DEV), B, L, H)
    q, k, v = (t.view(B, L, H, 16).permute(0, 2, 1, 3).double() for t in qkv.split(H * 16, dim=1))
    ref = torch.softmax(q @ k.transpose(-1, -2) / 4.0, dim=-1) @ v                  # [B,H,L,16]
    ref = ref.permute(0, 2, 1, 3).reshape(B * L, H * 16)
    assert relmax(out.cpu(), ref) <= 2e-6
    # a spiked key (one score far above the rest) exercises the running-max rescale
    qkv2 = qkv.clone()
    qkv2[5, H * 16:H * 16 + 16] *= 40.0
    out2, _ = n.attention(qkv2.to(DEV), B, L, H)
    q, k, v = (t.view(B, L, H, 16).permute(0, 2, 1, 3).double() for t in qkv2.split(H * 16, dim=1))
    ref2 = (torch.softmax(q @ k.transpose(-1, -2) / 4.0, dim=-1) @ v).permute(0, 2, 1, 3).reshape(B * L, H * 16)
    assert relmax(out2.cpu(), ref2) <= 2e-6
    x, r = torch.randn(300, 128, generator=g), torch.randn(300, 128, generator=g)
    ga, be_ = torch.rand(128, generator=g) + 0.5, torch.randn(128, generator=g) * 0.1
    y = n.add_layernorm(x.to(DEV), r.to(DEV), ga.to(DEV), be_.to(DEV))
    yr = torch.nn.functional.layer_norm((x + r).double(), (128,), ga.double(), be_.double(), 1e-5)
    assert relmax(y.cpu(), yr) <= 2e-6


def test_depth_completion_unet_on_gpu_vs_golden_and_oracle(env):
    """the '--densify pp' U-Net on the HIP conv kernels against the reference module (g13, float64) on the
    147x147 case, and against the oracle on a batch of two odd-sized maps (F.pad offsets on both axes)."""
    import models
    from oracle import unet as ou
    g = load_golden("g13_unet")
    m = models.DepthCompletion()
    m.load_state_dict({k: T(v) if v.dtype != np.int64 else torch.from_numpy(np.asarray(v)) for k, v in synth.unet_state_dict().items()})
    m = m.to(DEV).eval()
    y = m(T(synth.sparse_depth_map()).to(DEV))
    assert tuple(y.shape) == (1, 1, 147, 147)
    assert relmax(y.cpu(), g["f64_out"]) <= 2e-5
    x = torch.cat([T(synth.sparse_depth_map(75, 54, name="sd_a")), T(synth.sparse_depth_map(75, 54, name="sd_b"))])
    sd = {k: T(v, torch.float64) for k, v in synth.unet_state_dict().items() if not k.endswith("num_batches_tracked")}
    assert relmax(m(x.to(DEV)).cpu(), ou.forward(sd, x.double())) <= 2e-5
    with pytest.raises(NotImplementedError):
        models.DepthCompletion(bilinear=True).to(DEV).eval()(x.to(DEV))
    # --densify pp in the pipeline (blurry_edges_test.py:141-142): the U-Net sees the un-thresholded folded depth map
    from be_hip.pipeline import DepthPipeline
    base = _pipeline(env)
    pp = DepthPipeline(base.local, base.globl, env["helper"], env["dcal"], densify="pp", densify_pp_module=m)
    img = T(synth.synthetic_image_pair(147, 147)[0]).to(DEV)
    out = pp(img)
    assert torch.equal(out["depth_map"], m(out["depth"][None, None])[0, 0]) and torch.equal(out["depth"], base(img)["depth"])
    with pytest.raises(ValueError):
        DepthPipeline(base.local, base.globl, env["helper"], env["dcal"], densify="pp")


def test_datagen_on_gpu_vs_oracle_and_reference_golden(env):
    """the GPU generator against golden g14 (the reference's generate_synthetic_image on the same three
    scenes) and, on six more scenes, against the oracle restatement; noise statistics; patch crops."""
    from be_hip import datagen as dg
    from oracle import datagen as odg
    g = load_golden("g14_datagen")
    sc = dg.draw_scenes(3, seed=1869)
    d = dg.generate(sc, DEV, seed=11)
    H = W = 147
    for i in range(3):
        assert np.array_equal(d["images_aif"][i].cpu().numpy(), g[f"aif{i}"].astype(np.float64) / 255)
        assert np.array_equal(d["boundary_locations"][i].cpu().numpy().astype(np.uint8), g[f"bloc{i}"])
        assert np.array_equal(d["boundary_distances"][i].cpu().numpy().astype(np.int16), g[f"bdist{i}"])
        assert np.array_equal(d["image_depths"][i].cpu().numpy().astype(np.float32), g[f"idep{i}"])
        assert np.array_equal(d["boundary_depths"][i].cpu().numpy().astype(np.float32), g[f"bdep{i}"])
        diff = np.abs(d["images"][i].cpu().numpy() - g[f"imgs{i}"].astype(np.float64))
        assert diff.max() <= 1 and (diff > 0).mean() <= 1e-4, (diff.max(), (diff > 0).mean())     # x.5 rounding ties only
        same = (diff == 0).all(axis=-1)
        same = same & np.roll(same, 1, 1) & np.roll(same, -1, 1) & np.roll(same, 1, 2) & np.roll(same, -1, 2)
        dd = np.abs(d["derivative_maps"][i].cpu().numpy() - g[f"deri{i}"])[same]
        assert dd.max() <= 1e-6
    sc2 = dg.draw_scenes(6, seed=77, name="other")
    d2 = dg.generate(sc2, DEV, seed=12)
    sig = dg.kernel_sigmas(sc2["prop"], sc2["nobj"])
    for i in range(6):
        r = odg.generate_image(sc2["shape"][i], sc2["prop"][i], sc2["nobj"][i], sc2["bg"][i], sig[i], H, W, 1.18)
        assert np.array_equal(d2["boundary_locations"][i].cpu().numpy(), r["boundary_loc"])
        assert np.array_equal(d2["boundary_distances"][i].cpu().numpy(), r["boundary_dist"])
        assert np.array_equal(d2["boundary_depths"][i].cpu().numpy(), r["boundary_depth"])
        diff = np.abs(d2["images"][i].cpu().numpy() - r["imgs"])
        assert diff.max() <= 1 and (diff > 0).mean() <= 1e-4
        assert odg.candidates(r["boundary_loc"]).sum() > 0
    # noise model: gt exact; ny integer-valued in [0, alpha]; mean / variance of (ny - gt) as Poisson + read noise predict
    al = d2["alphas"].view(-1, 1, 1, 1, 1)
    assert np.array_equal(d2["images_gt"].cpu().numpy(), d2["images"].cpu().numpy() / 255 * al.cpu().numpy())   # numpy divides
    ny, gt = d2["images_ny"], d2["images_gt"]
    assert torch.equal(ny, ny.round()) and float(ny.min()) >= 0 and bool((ny <= torch.ceil(al)).all())     # clip, THEN round
    mid = (gt > 20) & (gt < al - 40)                      # away from the clipping at 0 and alpha
    e = (ny - gt)[mid]
    assert abs(float(e.mean())) < 0.05 and abs(float((e ** 2).mean()) / float((gt[mid] + 4.0 + 1 / 12).mean()) - 1) < 0.02
    low = gt < 5                                          # the small-lambda branch of the Poisson sampler
    if int(low.sum()) > 1000:
        k = torch.clamp(ny[low], min=0)
        assert abs(float(k.mean()) - float(torch.clamp(gt[low], min=0).mean())) < 0.6      # clipping at 0 biases upward a little
    # patches: candidates as the oracle's, windows copied verbatim, in-patch distance transform as the oracle's
    pt = dg.crop_patches(d2, 12, seed=5)
    cand = np.stack([odg.candidates(d2["boundary_locations"][i].cpu().numpy()) for i in range(6)])
    flat = pt["index"].cpu().numpy()
    assert cand.reshape(-1)[flat].all() and len(set(flat.tolist())) == 12
    for p_ in range(12):
        im, cy, cx = flat[p_] // (H * W), (flat[p_] // W) % H, flat[p_] % W
        ap = int(pt["aperture"][p_])
        win = (slice(cy - 10, cy + 11), slice(cx - 10, cx + 11))
        assert torch.equal(pt["patches_ny"][p_], d2["images_ny"][im, ap][win])
        assert torch.equal(pt["patches_aif"][p_], d2["images_aif"][im][win])
        bl = d2["boundary_locations"][im][win].cpu().numpy()
        assert np.array_equal(pt["boundary_locations"][p_].cpu().numpy(), bl)
        assert np.array_equal(pt["boundary_distances"][p_].cpu().numpy(), odg.l1_distance(bl > 0))
        assert float(pt["alphas"][p_]) == float(d2["alphas"][im])


def test_rasteriser_planes_vs_the_opencv_rules_of_the_oracle(env):
    """be_datagen_raster_u32 (one workgroup per object: Circle / Line + clipLine / CollectPolyEdges + FillEdgeCollection of OpenCV's
    drawing.cpp, cv2.circle / cv2.drawContours of train_val_data_generator.py:58-76) against oracle.datagen.cv_masks, bit for bit:
    the generator's own scenes, and hand-built edge cases - radius 0 and 1, a circle around a corner, shapes entirely outside,
    zero-area and collinear polygons, horizontal and vertical edges, vertices far outside the image, a non-square image."""
    from be_hip import datagen as dg
    from oracle import datagen as odg
    sc = dg.draw_scenes(4, seed=4242, name="raster")
    got = dg.rasterize(sc["shape"], sc["nobj"], DEV).cpu().numpy()
    seen = set()
    for i in range(4):
        for o in range(dg.MAXO):
            if o >= sc["nobj"][i]:
                assert not got[i, o].any()
                continue
            kind, pts = odg.shape_points(sc["shape"][i, o])
            fill, ring = odg.cv_masks(kind, pts, 147, 147)
            assert np.array_equal(got[i, o, 0], fill) and np.array_equal(got[i, o, 1], ring), (i, o, kind)
            seen.add(kind)
    assert seen == {0, 1, 2}
    H, W = 40, 70
    cases = [(0, 0, 10, 10, 0), (0, 0, 10, 10, 1), (0, 0, 0, 0, 12), (0, 0, 69, 39, 30), (0, 0, -50, -50, 8), (0, 0, 35, 20, 300),
             (1, 4, 5, 5, 5, 5, 5, 5, 5, 5), (1, 4, 3, 3, 30, 3, 30, 20, 3, 20), (1, 4, -20, 10, 35, -15, 90, 10, 35, 35),
             (1, 4, 10, 5, 40, 5, 40, 5, 10, 5), (2, 3, 2, 2, 8, 5, 14, 8), (2, 3, -100, -100, 200, 10, -30, 150),
             (2, 3, 0, 0, 69, 0, 0, 39), (2, 3, 69, 39, 69, 0, 0, 39), (2, 3, 5, 30, 60, 30, 33, 31), (2, 3, 80, 5, 90, 20, 75, 30),
             (1, 4, 10, 10, 50, 30, 50, 10, 10, 30)]                      # the last one is a bow-tie: four active edges per scan line
    rng = np.random.RandomState(9)
    for _ in range(15):
        nv = int(rng.choice([3, 4]))
        cases.append((1 if nv == 4 else 2, nv, *rng.randint(-40, 110, size=2 * nv).tolist()))
    shape = np.zeros((1, dg.MAXO, 10), dtype=np.int32)
    for k, c in enumerate(cases):
        shape[0, k, :len(c)] = c
    got = dg.rasterize(shape, np.array([len(cases)], dtype=np.int32), DEV, H, W).cpu().numpy()
    for k in range(len(cases)):
        kind, pts = odg.shape_points(shape[0, k])
        fill, ring = odg.cv_masks(kind, pts, H, W)
        assert np.array_equal(got[0, k, 0], fill), (k, cases[k])
        assert np.array_equal(got[0, k, 1], ring), (k, cases[k])
    assert not got[0, len(cases):].any()


def test_datagen_files_feed_the_local_training_dataset(env, tmp_path):
    """generate -> crop -> save -> data.ShapeDataset: the .npy wire format either side of training."""
    import data
    from be_hip import datagen as dg
    sc = dg.draw_scenes(8, seed=3, name="files")
    d = dg.generate(sc, DEV, seed=3)
    p = dg.crop_patches(d, 16, seed=3)
    dg.save(d, p, str(tmp_path), "train")
    ds = data.ShapeDataset(DEV, data_path=str(tmp_path / "patches"), train=True, mode="local")
    assert len(ds) == 16
    ny, gt, bd, de = ds[5]
    assert tuple(ny.shape) == (21, 21, 3) and tuple(de.shape) == (19, 19, 3) and ny.is_cuda
    assert torch.allclose(gt, (p["patches_gt"][5] / p["alphas"][5]).float()) and float(gt.max()) <= 1.0 + 1e-6
    assert torch.equal(bd, p["boundary_distances"][5].float())
    g = data.ShapeDataset(DEV, data_path=str(tmp_path), train=True, mode="global_pre")
    assert len(g) == 8 and tuple(g[0].shape) == (2, 147, 147, 3)
    assert np.load(tmp_path / "images_ny_train.npy").dtype == np.float64


def test_eval_depth_on_device_matches_reference_golden(env):
    """utils.eval_depth on GPU tensors (be_eval_depth_f32) against g10 = the reference's numpy function."""
    import utils
    S = synth.SEED_DEFAULT
    pred = 0.7 + 0.6 * synth.hash_uniform(S, "m_pred", (1, 147, 147))
    gt = 0.75 + 0.43 * synth.hash_uniform(S, "m_gt", (1, 147, 147))
    msk = synth.hash_uniform(S, "m_msk", (1, 147, 147)) > 0.3
    pred = np.where(msk, pred, 0.0).astype(np.float32)
    p, g = T(pred).to(DEV), T(gt.astype(np.float32)).to(DEV)
    r = utils.eval_depth(p, g, (p > 0), crop=10)
    assert np.allclose(np.array(r), load_golden("g10_metrics")["metrics"], rtol=2e-6)
    two = utils.eval_depth(torch.cat([p, p]), torch.cat([g, g]), torch.cat([p, p]) > 0, crop=10)
    assert np.allclose(np.array(two), np.array(r), rtol=1e-9)                  # batch is pooled, as in the reference


def test_workflow_datagen_train_precalc_train_eval_end_to_end(env, tmp_path):
    """The five driver scripts of the reference, end to end on this package, through the reference's FILE formats:
    data generator -> ShapeDataset 'local' -> local training (2 epochs) -> checkpoint -> global_pre -> params_src files ->
    ShapeDataset 'global' -> global training (1 epoch) -> checkpoint -> TestDataset -> evaluation metrics."""
    import utils
    from be_hip import datagen as dg, workflow as wf
    root = tmp_path / "data"
    for part, n, seed in (("train", 12, 21), ("val", 6, 22)):
        d = dg.generate(dg.draw_scenes(n, seed=seed, name="wf"), DEV, seed=seed)
        dg.save(d, dg.crop_patches(d, 8 * n, seed=seed), str(root), part)
    models_dir, logs = tmp_path / "weights", tmp_path / "logs"
    common = ["--model_path", str(models_dir), "--cuda", DEV]
    a = utils.get_args("local_train", argv=common + ["--data_path", str(root / "patches"), "--log_path", str(logs), "--epoch_num", "2",
                                                     "--batch_size", "32"])
    curve = wf.local_train(a, quiet=True)
    assert curve.shape == (2,) and np.isfinite(curve).all() and (models_dir / "best_run_exp_local_stage.pth").exists()
    assert (logs / "loss_curve_exp_local_stage.png").exists() and "Best epoch" in (logs / "exp_local_stage_training.txt").read_text()
    a = utils.get_args("global_pre", argv=common + ["--data_path", str(root)])
    wf.global_pre(a, local_weights=str(models_dir / "best_run_exp_local_stage.pth"), quiet=True)
    ps = np.load(root / "params_src_train.npy")
    assert ps.shape == (12, 2, 4096, 19) and ps.dtype == np.float64 and np.isfinite(ps).all()
    a = utils.get_args("global_train", argv=common + ["--data_path", str(root), "--log_path", str(logs), "--epoch_num", "1",
                                                      "--batch_size", "2"])
    gcurve = wf.global_train(a, quiet=True)
    assert np.isfinite(gcurve).all() and (models_dir / "best_run_exp_global_stage.pth").exists()
    # --resume / --time_budget (not in the reference: a 350-epoch schedule in bounded jobs): three epochs in one go against the same
    # three epochs as three jobs that each stop after one epoch and continue from global_resume.ckpt - identical validation curves,
    # identical best checkpoint (weights, AdamW moments, both schedules, the shuffle generator and the dropout seeds all carry over)
    runs = {}
    for tag, extra, calls in (("straight", [], 1), ("resumed", ["--resume", "--time_budget", "1e-9"], 3)):
        md, lg = tmp_path / f"w_{tag}", tmp_path / f"logs_{tag}"
        a = utils.get_args("global_train", argv=["--model_path", str(md), "--cuda", DEV, "--data_path", str(root), "--log_path", str(lg),
                                                 "--epoch_num", "3", "--batch_size", "2", "--dynamic_epoch", "2", "2", "5"] + extra)
        for _ in range(calls):
            c = wf.global_train(a, quiet=True)
        runs[tag] = (c, torch.load(md / "best_run_exp_global_stage.pth", map_location="cpu"), (lg / "exp_global_stage_training.txt").read_text())
    assert np.isfinite(runs["straight"][0]).all() and np.array_equal(runs["straight"][0], runs["resumed"][0])
    for k, v in runs["straight"][1].items():
        assert torch.equal(v, runs["resumed"][1][k]), k
    assert runs["resumed"][2].count("Best epoch") == 1 and runs["resumed"][2].count("Training:") == 1
    assert [ln.split()[0] for ln in runs["resumed"][2].splitlines() if ln[:1].isdigit()] == ["1", "2", "3"]
    # a test set in TestDataset's format made of the validation images (depth_maps = the generator's image depths)
    test_dir = tmp_path / "test"
    test_dir.mkdir()
    np.save(test_dir / "images_ny.npy", np.load(root / "images_ny_val.npy"))
    np.save(test_dir / "depth_maps.npy", np.load(root / "image_depths_val.npy"))
    np.save(test_dir / "alphas.npy", np.load(root / "alphas_val.npy"))
    a = utils.get_args("eval", argv=common + ["--data_path", str(test_dir), "--densify", "w"])
    res = wf.evaluate(a, local_weights=str(models_dir / "best_run_exp_local_stage.pth"),
                      global_weights=str(models_dir / "best_run_exp_global_stage.pth"), quiet=True)
    assert set(res) == {"delta1", "delta2", "delta3", "RMSE", "AbsRel", "seconds_per_pair"}
    assert all(np.isfinite(v) for v in res.values()) and 0 <= res["delta1"] <= res["delta2"] <= res["delta3"] <= 1


def test_evaluate_loads_the_w_checkpoint_for_densify_w_like_the_reference(env, tmp_path):
    """blurry_edges_test.py:187-190 loads pretrained_global_stage_w.pth for `--densify w`; only the big-image
    script always takes the plain name.  The plain file here is NOT a GlobalStage checkpoint, so any path that opens it fails."""
    import models, utils
    from be_hip import datagen as dg, workflow as wf
    d = dg.generate(dg.draw_scenes(2, seed=41, name="evw"), DEV, seed=41)
    dg.save(d, dg.crop_patches(d, 8, seed=41), str(tmp_path / "data"), "val")
    test_dir, wdir = tmp_path / "test", tmp_path / "w"
    test_dir.mkdir(); wdir.mkdir()
    for src, dst in (("images_ny_val", "images_ny"), ("image_depths_val", "depth_maps"), ("alphas_val", "alphas")):
        np.save(test_dir / f"{dst}.npy", np.load(tmp_path / "data" / f"{src}.npy"))
    lm = models.LocalStage()
    lm.load_state_dict({k: T(v) for k, v in synth.local_stage_state_dict().items()})
    torch.save(lm.state_dict(), wdir / "pretrained_local_stage.pth")
    gm = models.GlobalStage(device="cpu")
    gm.load_state_dict({k: T(v) for k, v in synth.global_stage_state_dict().items()})
    torch.save(gm.state_dict(), wdir / "pretrained_global_stage_w.pth")
    torch.save({"not": torch.zeros(1)}, wdir / "pretrained_global_stage.pth")
    common = ["--model_path", str(wdir), "--data_path", str(test_dir), "--cuda", DEV]
    res = wf.evaluate(utils.get_args("eval", argv=common + ["--densify", "w"]), quiet=True)       # takes the _w file
    assert all(np.isfinite(v) for v in res.values())
    with pytest.raises(RuntimeError):                                                              # default densify: the plain file
        wf.evaluate(utils.get_args("eval", argv=common), quiet=True)


def test_reference_style_postprocess_subclass_agrees_with_the_fused_pipeline(env):
    """A caller-written PostProcess(PostProcessGlobalBase) in the style of blurry_edges_test.py:12-100 - composed ONLY
    of the inherited methods, tensors in the reference's [B,.,21,21,Hp,Wp] layout - must give the maps of the fused
    records + fold path (DepthPipeline), i.e. the reference's own evaluation script can keep its subclass."""
    import utils
    n = env["native"]

    class PostProcess(utils.PostProcessGlobalBase):
        def __init__(self, args, depth_cal, device):
            super().__init__(args, device)
            self.depthCal, self.rho_prime = depth_cal, 10.39

        def solve_colors(self, wedges_pair, patches_pair):          # both apertures stacked: one colour set (882 rows)
            B, Hp, Wp = self.batch_size, self.H_patches, self.W_patches
            A = wedges_pair.permute(0, 5, 6, 1, 3, 4, 2).reshape(B, Hp, Wp, -1, 3)
            y = patches_pair.permute(0, 5, 6, 1, 3, 4, 2).reshape(B, Hp, Wp, -1, 3)
            At = A.transpose(-1, -2)
            return torch.matmul(self.inverse_3by3(torch.matmul(At, A) + self.ridge), torch.matmul(At, y)).permute(0, 4, 3, 1, 2)

        def composite(self, wedges, colors):
            return (wedges.unsqueeze(1) * colors.unsqueeze(-3).unsqueeze(-3)).sum(dim=2)

        def forward(self, est12, img_patches):
            est = est12.permute(0, 2, 1).reshape(self.batch_size, 12, self.H_patches, self.W_patches)
            dists = self.params2dists(est[:, :8].contiguous())
            etas = self.params2etas(est[:, 8:].contiguous())
            w1, w2 = self.dists2indicators(dists, etas[:, :2].contiguous()), self.dists2indicators(dists, etas[:, 2:].contiguous())
            colors = self.solve_colors(torch.stack([w1, w2], dim=1), img_patches.unsqueeze(0))
            patches = torch.stack([self.composite(w1, colors), self.composite(w2, colors)], dim=1)
            z1 = self.depthCal.etas2depth(etas[:, 0].contiguous(), etas[:, 2].contiguous())
            z2 = self.depthCal.etas2depth(etas[:, 1].contiguous(), etas[:, 3].contiguous())
            m1 = (self.normalized_gaussian(dists[:, 0]) > 0.5).to(torch.int32)
            m2 = (self.normalized_gaussian(dists[:, 1]) > 0.5).to(torch.int32) * 2
            mask = torch.where((m2 == 2) | (dists[:, 1] >= 0), m2, m1)
            zmap = torch.where(mask == 1, z1[:, None, None], torch.where(mask == 2, z2[:, None, None], torch.zeros_like(dists[:, 0])))
            d1, d2 = dists[:, 0].abs(), dists[:, 1].abs()
            bnd = self.normalized_gaussian(torch.where(dists[:, 1] >= 0, dists[:, 1], torch.minimum(d1, d2)))
            sharp = self.composite(self.dists2indicators(dists, torch.full_like(etas[:, :2], 1e-4)), colors)
            depth, conf = self.local2global_depth(zmap.contiguous(), mask.contiguous())
            return dict(image=self.local2global_color(patches.contiguous())[0], shpd=self.local2global_color(sharp.contiguous(), pair=False)[0],
                        bndry=self.local2global_bndry(bnd[:, None].contiguous())[0, 0], depth=depth[0], conf=conf[0])

    pipe = _pipeline(env)
    img = T(synth.synthetic_image_pair(147, 147)[0]).to(DEV)
    fused = pipe(img)
    helper = PostProcess(env["args"], env["dcal"], DEV)
    unfolded = n.unfold_patches(img).view(2, 64, 64, 3, 21, 21).permute(0, 3, 4, 5, 1, 2).contiguous()      # nn.Unfold layout
    mine = helper(fused["est12"][None], unfolded)
    assert relmax(mine["image"].cpu(), fused["image"].cpu()) <= 2e-3          # the subclass solves colours with the fp32 closed-form inverse
    assert relmax(mine["shpd"].cpu(), fused["shpd"].cpu()) <= 2e-3
    assert relmax(mine["bndry"].cpu(), fused["bndry"].cpu()) <= 1e-5
    assert relmax(mine["conf"].cpu(), fused["conf"].cpu()) <= 1e-6
    ok = (fused["conf"] > 0.05).cpu()
    assert relmax(mine["depth"].cpu()[ok], fused["depth"].cpu()[ok]) <= 1e-5
