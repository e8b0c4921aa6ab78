#!/usr/bin/env python3
"""Accuracy of a converged model, HIP pipeline against the oracle pipeline (VERDICT r4 "next" #2).  NOT a pytest module: a script
run on the GPU box by tools/converge.sh (STAGE=eval); it lives under tests/ because it calls the oracle, as a checker.

On a held-out synthetic basic-shapes set (the data generator with its own seed and stream name: no image shared with training or
validation), with the checkpoints under --weights:
  1. the HIP pipeline (be_hip.workflow.evaluate = blurry_edges_test.py:102-176) over all --n image pairs: delta1-3, RMSE (cm), AbsRel
     against the per-pixel image depth (the script's protocol) and against the boundary depth the method is trained on (see below);
  2. the ORACLE pipeline - the CPU restatement of the reference, free-running in the reference's own arithmetic (fp32, Cayley-Hamilton
     inverse) from the image to the depth map - over the first --oracle-n pairs, and the HIP pipeline's metrics on the same pairs;
  3. depth RMSE (build - oracle) over the pixels both pipelines give a depth for, with the fraction of pixels where the two
     confidence maps disagree (branch / mask flips), and the same against the fp64 stable-solve oracle;
  4. LocalStage logits at this checkpoint, Winograd and direct, against the fp64 oracle.
Writes <out>/converged_eval.json and prints it.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "blurry-edges_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def oracle_depth_map(sd_local, sd_global, pe, img, dtype=torch.float32, inverse="cayley"):
    """img [2,3,147,147] (CPU) -> (depth_map, depth, conf), all [147,147]: blurry_edges_test.py:117-145 on the oracle."""
    from oracle import local_stage as ols, render as orr, depth as od, tiling as ot, glue, global_stage as ogs
    pat = ot.unfold_patches(img.to(dtype))                                      # [2,P,3,21,21]
    P = pat.shape[1]
    flat = pat.reshape(-1, 3, 21, 21)
    est10 = torch.cat([ols.local_stage_forward(sd_local, flat[i:i + 1024]) for i in range(0, flat.shape[0], 1024)])
    col = orr.render_pass_a(orr.wrap_angles10(est10), flat, inverse=inverse)["colors"]
    pm = glue.local_features(est10.view(2, P, 10), col.view(2, P, 3, 3))
    y = ogs.forward(sd_global, pm[None], pe)[0]
    est12 = glue.global_denorm(y)
    r = orr.render_pass_b(od.depth_consts(), est12, pat[0], pat[1], inverse=inverse)
    fd, conf = ot.fold_depth(r["depth_map"][None], r["depth_mask"][None], img.shape[2], img.shape[3])
    dm = torch.where(conf[0] > 0.05, fd[0], torch.zeros_like(fd[0]))
    return dm, fd[0], conf[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--data", required=True, help="scratch directory (the held-out set is written to <data>/../test_heldout)")
    ap.add_argument("--weights", required=True)
    ap.add_argument("--out", required=True)
    ap.add_argument("--n", type=int, default=200)
    ap.add_argument("--oracle-n", type=int, default=12)
    ap.add_argument("--big", type=int, default=0, help="also score this many held-out 587 x 587 pairs through DepthPipeline.run_big")
    a = ap.parse_args()
    import models, utils
    from be_hip import datagen as dg, workflow as wf
    from oracle import local_stage as ols, global_stage as ogs
    dev = torch.device("cuda:0")
    torch.set_num_threads(min(16, os.cpu_count() or 1))

    # ---- held-out set in TestDataset's format (data/dataset.py:60-73): own seed, own stream name
    test_dir = os.path.join(os.path.dirname(os.path.abspath(a.data)), "test_heldout")
    os.makedirs(test_dir, exist_ok=True)
    ga = utils.get_args("data_gen_train_val", argv=[])
    chunks = []
    for first in range(0, a.n, 512):
        m = min(512, a.n - first)
        sc = dg.draw_scenes(m, seed=990001 + first, img_size=(147, 147), num_shape=tuple(ga.num_shape), z_range=tuple(ga.Z_range),
                            name="scenes.heldout")
        d = dg.generate(sc, dev, alpha_range=tuple(ga.alpha), sigma_read=ga.sigma, seed=990001 + first, z_far=ga.Z_range[1],
                        cam=dict(s=ga.cam_params['s'], rho=(ga.cam_params['rho_1'], ga.cam_params['rho_2']),
                                 sigma_cam=ga.cam_params['sigma_cam'], pixel_pitch=ga.cam_params['pixel_pitch'], mag=ga.mag))
        chunks.append({k: d[k].cpu().numpy() for k in ("images_ny", "image_depths", "boundary_depths", "alphas")})
    for src, dst in (("images_ny", "images_ny"), ("image_depths", "depth_maps"), ("alphas", "alphas")):
        np.save(os.path.join(test_dir, dst + ".npy"), np.concatenate([c[src] for c in chunks]))
    bdep_all = np.concatenate([c["boundary_depths"] for c in chunks])              # [n,H,W]: the occluder's depth on the dilated outlines

    res = dict(n_pairs=a.n, oracle_pairs=a.oracle_n, weights=sorted(os.listdir(a.weights)))
    ea = utils.get_args("eval", argv=["--model_path", a.weights, "--data_path", test_dir])
    res["hip_pipeline_vs_image_depth"] = wf.evaluate(ea, quiet=True)

    # ---- the same pairs through both pipelines
    import data
    from be_hip.pipeline import DepthPipeline
    load = lambda m, f: (m.load_state_dict(torch.load(os.path.join(a.weights, f), map_location=dev)), m.eval())[1]
    local = load(models.LocalStage().to(dev), "pretrained_local_stage.pth")
    globl = load(models.GlobalStage(in_parameter_size=38, out_parameter_size=12, device=dev).to(dev), "pretrained_global_stage.pth")
    pipe = DepthPipeline(local, globl, utils.PostProcessGlobalBase(ea, dev), utils.DepthEtas(ea, dev), rho_prime=ea.rho_prime,
                         densify=None, stride=ea.stride)
    sd_l = {k: v.detach().cpu() for k, v in local.state_dict().items()}
    sd_g = {k: v.detach().cpu() for k, v in globl.state_dict().items()}
    pe = ogs.position_table()
    ds = data.TestDataset("cpu", data_path=test_dir)
    names = ("delta1", "delta2", "delta3", "RMSE", "AbsRel")
    # Which ground truth?  blurry_edges_test.py:148-149 scores the depth map against the per-pixel depth of the test images, which for
    # the reference's TEXTURED test set (test_data_generator.py: every edge lies on the surface whose depth it carries) is the depth
    # the blur encodes.  On flat-coloured basic shapes the only edges are occlusion boundaries: their blur is the OCCLUDER's, which is
    # what the method predicts on both sides of the edge, while the per-pixel depth on the far side is the occluded object's.  So two
    # scores: "vs_image_depth" = the script's protocol with depth_maps = image_depths (as be_hip.workflow evaluate), and
    # "vs_boundary_depth" = against boundary_depths, the target the GlobalLoss depth term trains on (global_training.py:129-137), over
    # the pixels where both exist.
    def score_boundary(dm, bd):
        m = (dm > 0) & (bd > 0)
        c = ea.crop
        if int(m[c:-c, c:-c].sum()) == 0:
            return np.zeros(5), 0
        return np.array(utils.eval_depth(dm[None], np.where(bd > 0, bd, 1.0)[None], m[None].astype(np.float64), crop=ea.crop)), int(m.sum())
    with torch.no_grad():
        tb, nb_px, n_ok = np.zeros(5), 0, 0
        for j in range(a.n):
            img_ny, gt = ds[j]
            dm = pipe(img_ny.permute(0, 3, 1, 2).contiguous().to(dev))["depth_map"].cpu().numpy().astype(np.float64)
            sc, px = score_boundary(dm, bdep_all[j])
            if px > 0:
                tb += sc; nb_px += px; n_ok += 1
    res["hip_pipeline_vs_boundary_depth"] = dict(zip(names, (tb / max(n_ok, 1)).tolist()), pixels_per_pair=nb_px / max(n_ok, 1), pairs=n_ok)
    tot = {k: np.zeros(5) for k in ("hip", "oracle_fp32_reference_arithmetic", "oracle_fp64_stable_solve")}
    totb = {k: np.zeros(5) for k in tot}
    cmp_ = {k: dict(sq=0.0, cnt=0, flips=0, pix=0, maxabs=0.0) for k in ("oracle_fp32_reference_arithmetic", "oracle_fp64_stable_solve")}
    t_or = 0.0
    with torch.no_grad():
        for j in range(a.oracle_n):
            img_ny, gt = ds[j]
            img = img_ny.permute(0, 3, 1, 2).contiguous()
            maps = pipe(img.to(dev))
            hip_dm, hip_d, hip_c = maps["depth_map"].cpu(), maps["depth"].cpu(), maps["conf"].cpu()
            g = gt[None].numpy()
            tot["hip"] += np.array(utils.eval_depth(hip_dm[None].numpy(), g, hip_dm[None].numpy() > 0, crop=ea.crop))
            totb["hip"] += score_boundary(hip_dm.numpy().astype(np.float64), bdep_all[j])[0]
            for key, dt, inv in (("oracle_fp32_reference_arithmetic", torch.float32, "cayley"), ("oracle_fp64_stable_solve", torch.float64, "solve")):
                t0 = time.perf_counter()
                sl = {k: (v.to(dt) if v.is_floating_point() else v) for k, v in sd_l.items()}
                sg = {k: v.to(dt) for k, v in sd_g.items()}
                dm, d, c = oracle_depth_map(sl, sg, pe.to(dt), img, dt, inv)
                if dt == torch.float32:
                    t_or += time.perf_counter() - t0
                tot[key] += np.array(utils.eval_depth(dm[None].float().numpy(), g, dm[None].numpy() > 0, crop=ea.crop))
                totb[key] += score_boundary(dm.double().numpy(), bdep_all[j])[0]
                both = (hip_dm > 0) & (dm > 0)
                s = cmp_[key]
                # a pixel whose set of contributing patches differs between the two (a mask / branch flip somewhere) shows as a
                # different confidence: counted, and excluded from the RMSE
                flip = (hip_c - c.float()).abs() > 1e-6
                ok = both & ~flip
                diff = (hip_d.double() - d.double())[ok]
                s["sq"] += float((diff ** 2).sum()); s["cnt"] += int(ok.sum()); s["flips"] += int(flip.sum()); s["pix"] += flip.numel()
                s["maxabs"] = max(s["maxabs"], float(diff.abs().max()) if diff.numel() else 0.0)
            print(f"pair {j}: done", flush=True)
    n = max(a.oracle_n, 1)
    res["same_pairs"] = {k: dict(vs_image_depth=dict(zip(names, (v / n).tolist())), vs_boundary_depth=dict(zip(names, (totb[k] / n).tolist())))
                         for k, v in tot.items()}
    res["depth_build_minus_oracle"] = {k: dict(rmse_m=float(np.sqrt(s["sq"] / max(s["cnt"], 1))), max_abs_m=s["maxabs"], pixels=s["cnt"],
                                               conf_flip_frac=s["flips"] / max(s["pix"], 1)) for k, s in cmp_.items()}
    res["oracle_seconds_per_pair_fp32"] = t_or / n

    # ---- LocalStage logits at this checkpoint: Winograd and direct against the fp64 oracle
    img_ny, _ = ds[0]
    from oracle import tiling as ot
    flat = ot.unfold_patches(img_ny.permute(0, 3, 1, 2).contiguous()).reshape(-1, 3, 21, 21)
    x = flat[torch.arange(0, flat.shape[0], 16)]                                   # 512 patches of a real test image
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd_l.items()}
    with torch.no_grad():
        ref = ols.local_stage_forward(sd64, x.double())
        errs = {}
        for wino in (True, False):
            local.winograd = wino
            out = local(x.to(dev)).cpu().double()
            errs["winograd" if wino else "direct"] = float((out - ref).abs().max() / ref.abs().max())
        local.winograd = True
        o32 = ols.local_stage_forward(sd_l, x)
        errs["oracle_fp32"] = float((o32.double() - ref).abs().max() / ref.abs().max())
    res["logits_relmax_vs_fp64_oracle"] = errs
    # ---- the big-image path (blurry_edges_test_big.py: 587 x 587, 36 blocks of 147 x 147 with margin patches dropped) with the trained
    #      checkpoints, on held-out generated 587 x 587 scenes (the same object count on 16x the area: large flat shapes)
    if a.big > 0:
        sc = dg.draw_scenes(a.big, seed=880001, img_size=(587, 587), num_shape=tuple(ga.num_shape), z_range=tuple(ga.Z_range), name="scenes.heldout_big")
        d = dg.generate(sc, dev, alpha_range=tuple(ga.alpha), sigma_read=ga.sigma, seed=880001, z_far=ga.Z_range[1],
                        cam=dict(s=ga.cam_params['s'], rho=(ga.cam_params['rho_1'], ga.cam_params['rho_2']),
                                 sigma_cam=ga.cam_params['sigma_cam'], pixel_pitch=ga.cam_params['pixel_pitch'], mag=ga.mag))
        tb, ti, secs = np.zeros(5), np.zeros(5), 0.0
        with torch.no_grad():
            for j in range(a.big):
                img = (d["images_ny"][j] / d["alphas"][j]).float().permute(0, 3, 1, 2).contiguous()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                maps = pipe.run_big(img, n_margin=10)
                torch.cuda.synchronize()
                secs += time.perf_counter() - t0
                dm = maps["depth_map"].cpu().numpy().astype(np.float64)
                tb += score_boundary(dm, d["boundary_depths"][j].cpu().numpy())[0]
                ti += np.array(utils.eval_depth(dm[None], d["image_depths"][j].cpu().numpy()[None], dm[None] > 0, crop=ea.crop))
        res["big_587"] = dict(pairs=a.big, seconds_per_pair=secs / a.big, vs_boundary_depth=dict(zip(names, (tb / a.big).tolist())),
                              vs_image_depth=dict(zip(names, (ti / a.big).tolist())))
    os.makedirs(a.out, exist_ok=True)
    with open(os.path.join(a.out, "converged_eval.json"), "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
