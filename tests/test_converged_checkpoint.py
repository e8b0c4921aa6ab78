"""The checkpoints trained by this build (checkpoints/, tools/converge.sh: LocalStage at the reference's full schedule, GlobalStage for
80 epochs) as parity inputs: `blurry_edges_test.py:183-195` evaluates TRAINED weights, and trained weights are where BatchNorm
statistics, weight scales and attention logits stop looking like a random initialisation.  CPU: the files load strict into the
reference-shaped classes and the CPU module tree agrees with the oracle.  GPU: the HIP path against the fp64 oracle stage by stage,
and the whole 147x147 pipeline against the free-running oracle pipeline on one image pair."""
import os

import numpy as np
import pytest
import torch

from conftest import ROOT, relmax
from be_hip import synth

CK = os.path.join(ROOT, "checkpoints")
DEV = "cuda:0"


def _sd(name):
    path = os.path.join(CK, name)
    if not os.path.exists(path):
        pytest.skip(f"{path} is not in this tree")
    return torch.load(path, map_location="cpu")


def _patches(n=96):
    """patches of a generated-looking scene: windows of a synthetic image pair (edges, flat colours, noise)"""
    from oracle import tiling as ot
    imgs, _ = synth.synthetic_image_pair(147, 147)
    flat = ot.unfold_patches(torch.from_numpy(imgs)).reshape(-1, 3, 21, 21)
    return flat[torch.arange(0, flat.shape[0], flat.shape[0] // n)[:n]].contiguous()


def test_checkpoints_load_strict_and_the_cpu_module_tree_agrees_with_the_oracle():
    import models
    from oracle import local_stage as ols, global_stage as ogs
    sl, sg = _sd("pretrained_local_stage.pth"), _sd("pretrained_global_stage.pth")
    lm, gm = models.LocalStage(), models.GlobalStage(device="cpu")
    lm.load_state_dict(sl, strict=True)
    gm.load_state_dict(sg, strict=True)
    assert len(sl) == 100 and len(sg) == 102
    # trained, not initialised: the running statistics have moved away from (0, 1) and every counter has counted the steps
    assert int(sl["conv1.1.num_batches_tracked"]) > 10000
    rv = torch.cat([v.flatten() for k, v in sl.items() if k.endswith("running_var")])
    assert float(rv.max() / rv.min()) > 100
    x = _patches(32)
    lm.eval()
    with torch.no_grad():
        y = lm(x)
        ref = ols.local_stage_forward({k: (v.double() if v.is_floating_point() else v) for k, v in sl.items()}, x.double())
    assert relmax(y.numpy(), ref.numpy()) <= 2e-6                     # stock PyTorch fp32 vs the fp64 oracle at trained weights
    feats = torch.from_numpy(synth.global_features())[:, :512]
    gm.eval()
    with torch.no_grad():
        z = gm(feats.clone())
        zr = ogs.forward({k: v.double() for k, v in sg.items()}, feats.double(), ogs.position_table().double())
    assert relmax(z.numpy(), zr.numpy()) <= 2e-5


@pytest.mark.gpu
def test_hip_path_at_the_converged_checkpoints_vs_the_fp64_oracle():
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test run without a GPU")
    import models, utils
    from oracle import local_stage as ols, global_stage as ogs, render as orr, depth as od
    sl, sg = _sd("pretrained_local_stage.pth"), _sd("pretrained_global_stage.pth")
    lm = models.LocalStage()
    lm.load_state_dict(sl)
    lm = lm.to(DEV).eval()
    x = _patches(96)
    ref = ols.local_stage_forward({k: (v.double() if v.is_floating_point() else v) for k, v in sl.items()}, x.double())
    errs = {}
    with torch.no_grad():
        for wino in (True, False):
            lm.winograd = wino
            errs[wino] = relmax(lm(x.to(DEV)).cpu().numpy(), ref.numpy())
    print(f"converged LocalStage logits vs fp64 oracle: Winograd {errs[True]:.2e}  direct {errs[False]:.2e}")
    assert errs[True] <= 1e-5 and errs[False] <= 1e-5
    lm.winograd = True
    # pass A + depth from the HIP logits at trained weights against the oracle fed the same logits
    a = utils.get_args("eval", argv=[])
    helper = utils.PostProcessLocalBase(utils.get_args("local_train", argv=[]), DEV)
    dcal = utils.DepthEtas(a, DEV)
    from be_hip import native
    with torch.no_grad():
        est = lm(x.to(DEV))
        colors, _ = helper.render_colors(est, x.to(DEV), wrap_angles=True)
        depth = native.local_depth(dcal.consts, est)
    col_o = orr.render_pass_a(orr.wrap_angles10(est.cpu()).double(), x.double())["colors"]
    assert relmax(colors.cpu().numpy(), col_o.numpy()) <= 1e-4
    z_o = orr.local_depth(od.depth_consts(), est.cpu()[:48], est.cpu()[48:])
    # eta = 10^(2 erf(p) - 2) goes through the device's erff / powf here and through torch's on the oracle side (an ulp apart); the
    # solve behind it is bit-exact for equal eta (test_etas2depth_bit_exact_vs_oracle_and_golden)
    assert relmax(depth.cpu().numpy(), z_o.numpy()) <= 2e-6
    # GlobalStage at its trained weights: eval forward of one 4096-token sequence against the fp64 oracle
    gm = models.GlobalStage(device=DEV)
    gm.load_state_dict(sg)
    gm = gm.to(DEV).eval()
    feats = torch.from_numpy(synth.global_features())
    with torch.no_grad():
        y = gm(feats.to(DEV).clone())[0].cpu()
        idx = torch.arange(0, 4096, 61)
        yr = ogs.forward({k: v.double() for k, v in sg.items()}, feats.double(), ogs.position_table().double())[0]
    e = relmax(y[idx].numpy(), yr[idx].numpy())
    print(f"converged GlobalStage output vs fp64 oracle: {e:.2e}")
    assert e <= 1e-4


@pytest.mark.gpu
def test_pipeline_147_at_the_converged_checkpoints_vs_the_free_running_oracle_pipeline():
    """One held-out generated image pair through DepthPipeline and through the oracle pipeline (fp64, stable solve), both free-running
    from the image: the folded depth agrees to 1e-4 m RMSE over the pixels where no patch mask flipped, flips are rare, and the two
    depth maps give the same evaluation metrics."""
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test run without a GPU")
    import models, utils
    from be_hip import datagen as dg
    from be_hip.pipeline import DepthPipeline
    from oracle import global_stage as ogs
    from converged_eval import oracle_depth_map
    sl, sg = _sd("pretrained_local_stage.pth"), _sd("pretrained_global_stage.pth")
    a = utils.get_args("eval", argv=[])
    lm, gm = models.LocalStage(), models.GlobalStage(device=DEV)
    lm.load_state_dict(sl)
    gm.load_state_dict(sg)
    pipe = DepthPipeline(lm.to(DEV).eval(), gm.to(DEV).eval(), utils.PostProcessGlobalBase(a, DEV), utils.DepthEtas(a, DEV),
                         rho_prime=a.rho_prime, densify=None, stride=a.stride)
    d = dg.generate(dg.draw_scenes(1, seed=424243, name="scenes.conv_test"), DEV, seed=424243)
    img = (d["images_ny"][0] / d["alphas"][0]).float().permute(0, 3, 1, 2).contiguous()           # [2,3,147,147], as TestDataset
    with torch.no_grad():
        maps = pipe(img)
        dt = torch.float64
        dm, depth, conf = oracle_depth_map({k: (v.to(dt) if v.is_floating_point() else v) for k, v in sl.items()},
                                           {k: v.to(dt) for k, v in sg.items()}, ogs.position_table().to(dt), img.cpu(), dt, "solve")
    flip = (maps["conf"].cpu() - conf.float()).abs() > 1e-6
    ok = ~flip & (maps["depth_map"].cpu() > 0) & (dm > 0)
    rmse = float(torch.sqrt(((maps["depth"].cpu().double() - depth)[ok] ** 2).mean()))
    print(f"converged pipeline: depth RMSE build - oracle {rmse:.2e} m over {int(ok.sum())} pixels, confidence flips {float(flip.float().mean()):.2e}")
    assert int(ok.sum()) > 500 and rmse <= 1e-4 and float(flip.float().mean()) <= 2e-3
    gt = d["image_depths"][0].cpu().numpy()[None]
    mh = utils.eval_depth(maps["depth_map"].cpu().numpy()[None].astype(np.float64), gt, maps["depth_map"].cpu().numpy()[None] > 0, crop=a.crop)
    mo = utils.eval_depth(dm.numpy()[None], gt, dm.numpy()[None] > 0, crop=a.crop)
    assert np.allclose(np.array(mh), np.array(mo), rtol=2e-3, atol=2e-3)
