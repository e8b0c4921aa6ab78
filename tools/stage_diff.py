"""Stage-by-stage difference between the HIP pipeline and the float64 oracle at given checkpoints on ONE held-out 147 x 147 pair
(diagnostic; run on the GPU box): LocalStage logits -> pass-A colours -> features -> GlobalStage output -> wedge parameters -> depth map.
usage: python tools/stage_diff.py <weights dir> [old weights dir]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "blurry-edges_amd")]
import models, utils  # noqa: E402
from be_hip import datagen as dg  # noqa: E402
from be_hip.pipeline import DepthPipeline  # noqa: E402
from oracle import local_stage as ols, render as orr, depth as od, tiling as ot, glue, global_stage as ogs  # noqa: E402

dev = torch.device("cuda:0")
torch.set_num_threads(16)
ga = utils.get_args("data_gen_train_val", argv=[])
sc = dg.draw_scenes(2, seed=990001, img_size=(147, 147), num_shape=tuple(ga.num_shape), z_range=tuple(ga.Z_range), name="scenes.heldout")
d = dg.generate(sc, dev, alpha_range=tuple(ga.alpha), sigma_read=ga.sigma, seed=990001, z_far=ga.Z_range[1],
                cam=dict(s=ga.cam_params['s'], rho=(ga.cam_params['rho_1'], ga.cam_params['rho_2']), sigma_cam=ga.cam_params['sigma_cam'],
                         pixel_pitch=ga.cam_params['pixel_pitch'], mag=ga.mag))
img = (d["images_ny"][0] / d["alphas"][0]).float().permute(0, 3, 1, 2).contiguous().cpu()
rel = lambda a, b: float((a.double() - b.double()).abs().max() / b.double().abs().max())
for wdir in sys.argv[1:]:
    ea = utils.get_args("eval", argv=["--model_path", wdir])
    load = lambda m, f: (m.load_state_dict(torch.load(os.path.join(wdir, f), map_location=dev)), m.eval())[1]
    local = load(models.LocalStage().to(dev), "pretrained_local_stage.pth")
    globl = load(models.GlobalStage(in_parameter_size=38, out_parameter_size=12, device=dev).to(dev), "pretrained_global_stage.pth")
    pipe = DepthPipeline(local, globl, utils.PostProcessGlobalBase(ea, dev), utils.DepthEtas(ea, dev), rho_prime=ea.rho_prime, densify=None, stride=ea.stride)
    with torch.no_grad():
        _, est10, colors, pm = pipe.local_pass(img.to(dev))
        y_hip = globl(pm.unsqueeze(0))[0]
        est12 = pipe.global_pass(pm)
        maps = pipe(img.to(dev))
        dt = torch.float64
        sl = {k: (v.detach().cpu().to(dt) if v.is_floating_point() else v.detach().cpu()) for k, v in local.state_dict().items()}
        sg = {k: v.detach().cpu().to(dt) for k, v in globl.state_dict().items()}
        pat = ot.unfold_patches(img.to(dt))
        P = pat.shape[1]
        flat = pat.reshape(-1, 3, 21, 21)
        e10 = torch.cat([ols.local_stage_forward(sl, flat[i:i + 1024]) for i in range(0, flat.shape[0], 1024)])
        col = orr.render_pass_a(orr.wrap_angles10(e10), flat, inverse="solve")["colors"]
        pmo = glue.local_features(e10.view(2, P, 10), col.view(2, P, 3, 3))
        pe = ogs.position_table().to(dt)
        yo = ogs.forward(sg, pmo[None], pe)[0]
        # the oracle GlobalStage fed with the HIP features: isolates the transformer from what precedes it
        yo_same = ogs.forward(sg, pm.cpu().to(dt)[None], pe)[0]
        e12o = glue.global_denorm(yo)
        r = orr.render_pass_b(od.depth_consts(), e12o, pat[0], pat[1], inverse="solve")
        fd, conf = ot.fold_depth(r["depth_map"][None], r["depth_mask"][None], 147, 147)
    print(f"== {wdir}")
    print("  logits          hip vs f64: %.2e   |logits| max %.1f" % (rel(est10.cpu(), e10), float(e10.abs().max())))
    print("  pass-A colours  hip vs f64: %.2e" % rel(colors.cpu().view(-1, 3, 3), col))
    print("  features pm     hip vs f64: %.2e   |pm| max %.2f" % (rel(pm.cpu(), pmo), float(pmo.abs().max())))
    print("  GlobalStage out hip vs f64: %.2e (same input: %.2e)   |y| max %.2f" % (rel(y_hip.cpu(), yo), rel(y_hip.cpu(), yo_same), float(yo.abs().max())))
    print("  est12           hip vs f64: %.2e" % rel(est12.cpu(), e12o))
    flip = (maps["conf"].cpu() - conf[0].float()).abs() > 1e-6
    both = (maps["depth"].cpu() > 0) & (fd[0] > 0) & ~flip
    dd = (maps["depth"].cpu().double() - fd[0])[both]
    print("  depth           rmse %.2e m, max %.2e m over %d pixels; confidence flips %.2e of the pixels" % (float(dd.pow(2).mean().sqrt()), float(dd.abs().max()), int(both.sum()), float(flip.float().mean())))
    # sensitivity: the oracle's own float32 run against its float64 run through the transformer
    yo32 = ogs.forward({k: v.float() for k, v in sg.items()}, pmo.float()[None], pe.float())[0]
    print("  (oracle f32 vs f64 through the transformer: %.2e)" % rel(yo32, yo))
