"""The reference's driver scripts as functions + a small CLI, running on this package's HIP path and the reference's file
formats (argument names and defaults = utils/args.py of the reference):

    python -m be_hip.datagen      [...]        train_val_data_generator.py  (data set -> .npy files)
    python -m be_hip.workflow local_train      local_training.py:68-121     (ShapeDataset 'local' -> best_run_exp_local_stage.pth)
    python -m be_hip.workflow global_pre       global_data_pre_cal.py:52-69 (local stage over images -> params_src_*.npy)
    python -m be_hip.workflow global_train     global_training.py:168-224   (ShapeDataset 'global' -> best_run_exp_global_stage.pth)
    python -m be_hip.workflow eval [--big]     blurry_edges_test.py:102-176 / blurry_edges_test_big.py (TestDataset -> metrics)

Data parallel (BASELINE configs[4]): the two training commands run under torchrun, one process per GPU -
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m be_hip.workflow local_train ...
every rank keeps a replica, takes every world-th batch of the epoch, gradients are averaged over RCCL (be_hip.dp), rank 0 logs and
writes the checkpoints (the reference has no distributed code at all: this part is the build's own).

Checkpoints are `torch.save(model.state_dict())` files in the reference's key layout, so they are interchangeable with
the reference's.  Epoch loops, schedules (beta / gamma ramps, ReduceLROnPlateau with the growing patience), xavier
initialisation, clipping and AdamW settings follow the scripts cited above; the compute inside a step is the HIP path.
"""
from __future__ import annotations

import argparse
import os
import sys
import time

import numpy as np
import torch


def _device(args):
    return torch.device(args.cuda if torch.cuda.is_available() else "cpu")


def _dist():
    """(rank, world, dist module or None) for a run started by torchrun (one process per GPU, configs[4]); a plain `python -m
    be_hip.workflow ...` is (0, 1, None).  Backend nccl (= RCCL); BE_DIST_BACKEND=gloo + BE_LOCAL_DEVICE=0 rehearse it on one GPU."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1:
        return 0, 1, None
    import torch.distributed as dist
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("BE_DIST_BACKEND", "nccl")
        lr = int(os.environ.get("BE_LOCAL_DEVICE", os.environ.get("LOCAL_RANK", "0")))
        if torch.cuda.is_available():
            torch.cuda.set_device(lr)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", lr))
        else:
            dist.init_process_group(backend)
    return dist.get_rank(), dist.get_world_size(), dist


def _xavier_(model):
    for p in model.parameters():
        if p.dim() > 1:
            torch.nn.init.xavier_normal_(p)


def _log_header(f, args):
    print('Arguments:', file=f, flush=True)
    for k in vars(args):
        print(f'{k:<20}: {getattr(args, k)}', file=f, flush=True)
    print('\nTraining:', file=f, flush=True)
    print(f'{"Epoch":<10} {"Loss":<20} {"Scheduler patience":<20} {"Learning rate"}', file=f, flush=True)


# ---------------------------------------------------------------------------------------------- local_training.py
def local_train(args, quiet=False, graph=True):
    """graph: replay the training step as a hipGraph (train_local.GraphedStep); batches are gathered on the device
    (data.ShapeDataset.batches) instead of sample by sample."""
    import data, models, utils
    from . import dp
    from .train_local import BetaSchedule, GraphedStep, train_step
    rank, world, dist = _dist()
    utils.set_seed(1869)
    if rank == 0:
        utils.create_directory(args.log_path, overwrite=False)
        os.makedirs(args.model_path, exist_ok=True)
    if dist is not None:
        dist.barrier()
    dev = _device(args) if world == 1 else torch.device("cuda", torch.cuda.current_device())
    tr = data.ShapeDataset(dev, data_path=args.data_path, train=True)
    va = data.ShapeDataset(dev, data_path=args.data_path, train=False)
    model = models.LocalStage().to(dev)
    _xavier_(model)
    # data parallel (configs[4]): every rank keeps a replica, takes every world-th batch of the epoch's (identically shuffled)
    # batch sequence - a global batch of batch_size x world - and the gradients are averaged by the bucketed all-reduce (four buckets) that
    # overlaps the backward (be_hip.dp.GradSync).  graph=True: the step is six hipGraph segments with the RCCL calls between
    # them (train_local.SegmentedGraphStep); one GPU: one hipGraph (GraphedStep)
    sync = dp.GradSync(world) if world > 1 else None
    if world > 1:
        dp.broadcast_parameters(model, src=0)
    from .optim import ClipAdamW
    opt = ClipAdamW(model.parameters(), lr=args.learning_rate)   # AdamW defaults of local_training.py:86, fused with the clipping
    helper = utils.PostProcessLocalBase(args, dev)
    if not graph:
        gstep = None
    elif world > 1:
        from .train_local import SegmentedGraphStep
        gstep = SegmentedGraphStep(model, helper, opt, sync, world=world)
    else:
        gstep = GraphedStep(model, helper, opt)
    sampler = torch.Generator().manual_seed(1869)
    beta = BetaSchedule(args.beta_bndry_loc, args.beta_smthns, args.dynamic_epoch)
    sched = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, 'min', factor=0.9, patience=2, min_lr=args.learning_rate * 0.1)
    curve = np.zeros((args.epoch_num,), dtype=float)
    best, best_epoch = np.inf, 0
    with open(f'{args.log_path}/exp_local_stage_training.txt' if rank == 0 else os.devnull, 'wt') as f:
        _log_header(f, args)
        for epoch in range(args.epoch_num):
            beta.step()
            model.train()
            # every rank takes the same number of steps (collectives pair up) and gathers only its own batches
            for img_ny, img_gt, bndry_dist, deri in tr.batches(args.batch_size, shuffle=True, drop_last=True, generator=sampler,
                                                               rank=rank, world=world):
                b = dict(img_ny=img_ny, img_gt=img_gt, bndry_dist=bndry_dist, deri=deri)
                if gstep is not None:
                    gstep(b, beta.beta_b, beta.beta_s)
                else:
                    train_step(model, helper, opt, b, beta.beta_b, beta.beta_s, world=world, sync=sync)
            if world > 1:
                # BatchNorm statistics are per replica during the epoch; rank 0's go to everyone before validation, so that every
                # rank sees the same validation loss (same lr schedule everywhere) and the checkpoint is what was validated
                dp.broadcast_bn_stats(model, src=0)
            # validation with the final betas (local_training.py:54-66)
            model.eval()
            total = 0.0
            with torch.no_grad():
                for img_ny, img_gt, bndry_dist, deri in va.batches(args.batch_size, shuffle=False, drop_last=True):
                    est = model(img_ny.permute(0, 3, 1, 2).contiguous())
                    total += float(utils.local_loss(helper, est, img_ny, img_gt, bndry_dist, deri, args.beta_bndry_loc,
                                                    args.beta_smthns))
            curve[epoch] = total / max(len(va) // args.batch_size, 1)
            sched.step(curve[epoch])
            sched.patience = 2 + int(np.log2(epoch + 1)) * 3
            print(f'{epoch + 1:<10} {curve[epoch]:<20.10f} {sched.patience:<20} {opt.param_groups[0]["lr"]:.4e}', file=f, flush=True)
            if curve[epoch] < best:
                best, best_epoch = curve[epoch], epoch
                if rank == 0:
                    torch.save(model.state_dict(), f'{args.model_path}/best_run_exp_local_stage.pth')
            if not quiet and rank == 0:
                print(f'epoch {epoch + 1}: validation loss {curve[epoch]:.6f}')
        print(f'\n-- Best epoch is the {best_epoch + 1:d}th, with average loss of {best:.10f}', file=f, flush=True)
    if rank == 0:
        np.save(f'{args.log_path}/loss_curve_exp_local_stage.npy', curve)
        utils.showCurve(args, curve, 'loss_curve_exp_local_stage')
    if dist is not None:
        dist.barrier()
    return curve


# ------------------------------------------------------------------------------------------- global_data_pre_cal.py
@torch.no_grad()
def global_pre(args, local_weights=None, quiet=False):
    """params_src_{train,val}.npy [n, 2, P, 19] float64: the normalised local-stage output + colours of every image."""
    import data, models, utils
    from . import native
    from .pipeline import DepthPipeline, params_src_layout
    dev = _device(args)
    model = models.LocalStage().to(dev)
    model.load_state_dict(torch.load(local_weights or f'{args.model_path}/pretrained_local_stage.pth', map_location=dev))
    model.eval()
    pipe = DepthPipeline(model, None, utils.PostProcessGlobalBase(args, dev), None, stride=args.stride)
    for part, train in (("train", True), ("val", False)):
        ds = data.ShapeDataset(dev, data_path=args.data_path, train=train, mode='global_pre')
        out = None
        for j in range(len(ds)):
            img = ds[j].permute(0, 3, 1, 2).contiguous()                     # [2,3,H,W]
            pm = pipe.local_pass(img)[3]                                      # [P,38] = [P, (aperture, 19)]
            if out is None:
                out = np.zeros((len(ds), 2, pm.shape[0], 19), dtype=np.float64)
            out[j] = params_src_layout(pm).cpu().numpy()
        np.save(f'{args.data_path}/params_src_{part}.npy', out)
        if not quiet:
            print(f'{part}: {len(ds)} images -> params_src_{part}.npy')


# ------------------------------------------------------------------------------------------------ global_training.py
def global_train(args, quiet=False):
    import data, models, utils
    from . import dp
    from .train_global import GammaSchedule, train_step
    rank, world, dist = _dist()
    utils.set_seed(1898)
    if rank == 0:
        utils.create_directory(args.log_path, overwrite=False)
        os.makedirs(args.model_path, exist_ok=True)
    if dist is not None:
        dist.barrier()
    dev = _device(args) if world == 1 else torch.device("cuda", torch.cuda.current_device())
    tr = data.ShapeDataset(dev, data_path=args.data_path, train=True, mode='global')
    va = data.ShapeDataset(dev, data_path=args.data_path, train=False, mode='global')
    sampler = torch.Generator().manual_seed(1898)
    model = models.GlobalStage(in_parameter_size=args.input_size, out_parameter_size=args.output_size, device=dev).to(dev)
    _xavier_(model)
    # data parallel: replicas aligned with rank 0, every world-th batch per rank, ONE all-reduce of the 4.27 MB gradient per step
    flat = dp.flat_grad_buffer(model.parameters()) if world > 1 else None
    if world > 1:
        dp.broadcast_parameters(model, src=0)
        for p_ in model.parameters():
            p_.grad = None
        # every rank seeds torch identically (same shuffle, same initial weights); the dropout masks must NOT be identical across
        # the replicas of one global batch, so each rank salts the seed GlobalStage draws per step
        model.dropout_seed_salt = (rank * 0x9E3779B1) & 0x7FFFFFFF
    from .optim import ClipAdamW
    opt = ClipAdamW(model.parameters(), lr=args.learning_rate, gather=True)      # AdamW defaults of global_training.py:196, fused with the clipping
    helper = utils.PostProcessGlobalBase(args, dev)
    dcal = utils.DepthEtas(args, dev)
    gamma = GammaSchedule(args)
    sched = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, 'min', factor=0.975, patience=5, min_lr=args.learning_rate * 0.5)
    curve = np.zeros((args.epoch_num,), dtype=float)
    best, best_epoch = np.inf, 0
    feats = lambda p: p.permute(0, 2, 1, 3).flatten(2, 3).contiguous()          # [B,2,P,19] -> [B,P,38]
    # --resume (not in the reference): everything the loop carries from one epoch to the next is in {model_path}/global_resume.ckpt -
    # weights, AdamW moments, both schedules, the shuffling generator and torch's RNG (the dropout seeds come from it), the curve and
    # the best checkpoint - so a resumed run takes exactly the steps the uninterrupted one would have taken
    resume_path = f'{args.model_path}/global_resume.ckpt'
    first_epoch = 0
    resumed = bool(getattr(args, "resume", False)) and os.path.exists(resume_path)
    if resumed:
        ck = torch.load(resume_path, map_location=dev, weights_only=False)
        model.load_state_dict(ck["model"])
        opt.load_state_dict(ck["opt"])
        sched.load_state_dict(ck["sched"])
        gamma.idx = ck["gamma_idx"]
        sampler.set_state(ck["sampler"].cpu())                       # generator states are CPU byte tensors (map_location moved them)
        torch.set_rng_state(ck["torch_rng"].cpu())
        first_epoch, best, best_epoch = ck["epoch"], ck["best"], ck["best_epoch"]
        curve[:first_epoch] = ck["curve"][:first_epoch]
        if rank == 0 and ck.get("best_model") is not None:
            torch.save(ck["best_model"], f'{args.model_path}/best_run_exp_global_stage.pth')
    t_start = time.perf_counter()
    budget = float(getattr(args, "time_budget", 0.0) or 0.0)
    stopped_early = False
    with open(f'{args.log_path}/exp_global_stage_training.txt' if rank == 0 else os.devnull, 'at' if resumed else 'wt') as f:
        if not resumed:
            _log_header(f, args)
        for epoch in range(first_epoch, args.epoch_num):
            g = gamma.step()
            model.train()
            for param, _, img_gt, bndry_dist, deri, bndry_depth in tr.batches(args.batch_size, shuffle=True, drop_last=True,
                                                                             generator=sampler, rank=rank, world=world):
                train_step(model, helper, dcal, opt, dict(pm=feats(param), img_gt=img_gt, bndry_dist=bndry_dist, deri=deri,
                                                          bndry_depth=bndry_depth), g, flat=flat, world=world)
            model.eval()
            total, gf = 0.0, gamma.final()
            with torch.no_grad():
                for param, img_ny, img_gt, bndry_dist, deri, bndry_depth in va.batches(args.batch_size, shuffle=False, drop_last=True):
                    est = model(feats(param))
                    total += float(utils.global_loss(helper, dcal, est, img_ny, img_gt, bndry_dist, deri, bndry_depth, gf, empty_mask="zero"))
            gamma.step(idx_update=False)
            curve[epoch] = total / max(len(va) // args.batch_size, 1)
            print(f'{epoch + 1:<10} {curve[epoch]:<20.10f} {sched.patience:<20} {opt.param_groups[0]["lr"]:.4e}', file=f, flush=True)
            if curve[epoch] < best:
                best, best_epoch = curve[epoch], epoch
                if rank == 0:
                    torch.save(model.state_dict(), f'{args.model_path}/best_run_exp_global_stage.pth')
            if epoch >= args.dynamic_epoch[1]:
                sched.step(curve[epoch])
            if not quiet and rank == 0:
                print(f'epoch {epoch + 1}: validation loss {curve[epoch]:.6f}', flush=True)
            if rank == 0 and (getattr(args, "resume", False) or budget > 0):
                bm = f'{args.model_path}/best_run_exp_global_stage.pth'
                torch.save(dict(model=model.state_dict(), opt=opt.state_dict(), sched=sched.state_dict(), gamma_idx=gamma.idx,
                                sampler=sampler.get_state(), torch_rng=torch.get_rng_state(), epoch=epoch + 1, best=best,
                                best_epoch=best_epoch, curve=curve.copy(),
                                best_model=torch.load(bm, map_location="cpu") if os.path.exists(bm) else None), resume_path + ".tmp")
                os.replace(resume_path + ".tmp", resume_path)
            if budget > 0 and epoch + 1 < args.epoch_num:
                stopped_early = time.perf_counter() - t_start > budget
                if dist is not None:                                   # rank 0's clock decides for everyone (collectives pair up)
                    flag = torch.tensor([int(stopped_early)], device=dev if dist.get_backend() == "nccl" else "cpu")
                    dist.broadcast(flag, 0)
                    stopped_early = bool(flag.item())
                if stopped_early:
                    if rank == 0:
                        print(f'-- time budget reached after epoch {epoch + 1} of {args.epoch_num}: continue with --resume', flush=True)
                    break
        if not stopped_early:
            print(f'\n-- Best epoch is the {best_epoch + 1:d}th, with average loss of {best:.10f}.', file=f, flush=True)
    if rank == 0:
        np.save(f'{args.log_path}/loss_curve_exp_global_stage.npy', curve)
        utils.showCurve(args, curve, 'loss_curve_exp_global_stage')
    if dist is not None:
        dist.barrier()
    return curve


# ------------------------------------------------------------------------- blurry_edges_test.py / blurry_edges_test_big.py
@torch.no_grad()
def evaluate(args, big=False, local_weights=None, global_weights=None, pp_weights=None, quiet=False):
    """-> dict(delta1, delta2, delta3, RMSE, AbsRel, seconds_per_pair), averaged over the test set as the scripts do."""
    import data, models, utils
    from .pipeline import DepthPipeline
    dev = _device(args)
    load = lambda m, path: (m.load_state_dict(torch.load(path, map_location=dev)), m.eval())[1]
    local = load(models.LocalStage().to(dev), local_weights or f'{args.model_path}/pretrained_local_stage.pth')
    # blurry_edges_test.py:187-190 loads pretrained_global_stage_w.pth for `--densify w`; the big-image script
    # (blurry_edges_test_big.py) always loads the plain name
    gname = 'pretrained_global_stage_w.pth' if (args.densify == 'w' and not big) else 'pretrained_global_stage.pth'
    globl = load(models.GlobalStage(in_parameter_size=38, out_parameter_size=12, device=dev).to(dev),
                 global_weights or f'{args.model_path}/{gname}')
    pp = None
    if args.densify == 'pp':
        pp = load(models.DepthCompletion().to(dev), pp_weights or f'{args.model_path}/pretrained_depth_completion_pp.pth')
    pipe = DepthPipeline(local, globl, utils.PostProcessGlobalBase(args, dev), utils.DepthEtas(args, dev),
                         rho_prime=args.rho_prime, densify=args.densify, stride=args.stride, densify_pp_module=pp)
    ds = data.TestDataset(dev, data_path=args.data_path)
    tot = np.zeros(5)
    secs = 0.0
    for j in range(len(ds)):
        img_ny, gt = ds[j]
        img = img_ny.permute(0, 3, 1, 2).contiguous()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        maps = pipe.run_big(img, n_margin=args.n_margin_patch) if big else pipe(img)
        torch.cuda.synchronize()
        secs += time.perf_counter() - t0
        depth = maps["depth_map"][None]
        m = np.array(utils.eval_depth(depth, gt[None].to(depth.dtype), depth, crop=args.crop))
        tot += m
        if not quiet:
            print(f'Image pair #{j}: delta1 ={m[0]: .3f}, delta2 ={m[1]: .3f}, delta3 ={m[2]: .3f}, RMSE ={m[3]: .3f} cm, '
                  f'AbsRel ={m[4]: .3f} cm')
    n = max(len(ds), 1)
    res = dict(zip(("delta1", "delta2", "delta3", "RMSE", "AbsRel"), (tot / n).tolist()), seconds_per_pair=secs / n)
    if not quiet:
        print(f'\nAverage running time:{secs / n: .3f} s')
        print('Average metrics for whole dataset: ' + ', '.join(f'{k} ={v: .3f}' for k, v in res.items() if k != "seconds_per_pair"))
    return res


def main(argv=None):
    import utils
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv or argv[0] not in ("local_train", "global_pre", "global_train", "eval"):
        raise SystemExit("usage: python -m be_hip.workflow {local_train|global_pre|global_train|eval} [--big] [reference arguments]")
    cmd, rest = argv[0], argv[1:]
    big = "--big" in rest
    rest = [a for a in rest if a != "--big"]
    if cmd == "eval":
        evaluate(utils.get_args('eval', big=big, argv=rest), big=big)
    elif cmd == "local_train":
        local_train(utils.get_args('local_train', argv=rest))
    elif cmd == "global_pre":
        global_pre(utils.get_args('global_pre', argv=rest))
    else:
        global_train(utils.get_args('global_train', argv=rest))


if __name__ == "__main__":
    main()
