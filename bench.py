#!/usr/bin/env python3
"""bench.py -- patch-pairs/s through the Blurry-Edges hot path on MI355X.

One "step" = one pass of the hot path over one batch of synthetic input, BASELINE.json configs[1]:
a batch of 4096 synthetic 21x21 two-aperture patch pairs (8192 CNN patches) per GPU through
  LocalStage CNN (fp32-MFMA implicit GEMM)  ->  wedge renderer pass A (colours, ridge solve)  ->  DfD depth solve
with the input already resident in HBM.  Prints ONE JSON line (rank 0).

  python bench.py [--gpus N] [--steps K] [--warmup W]          N > 1: this process only spawns the N ranks (children of the same
                                                               script, started before any GPU call) and relays their exit codes
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W      (the same ranks under a launcher)
The line's `launch` object says who ran: launcher, backend, rccl_ranks (world size under backend nccl), RCCL version, every rank's
device / pid / own clock, and value / world.

`value` is that configuration and nothing else.  The same line also carries (outside the timed region of `value`):
  roofline       the dominant kernel of the step, measured live with hipEvents on the launch stream;
                 `frac` = the FLOPs of the algorithm that runs (Winograd tiles on layers 1-3: what the matrix pipe issues)
                 / duration / peak, a utilisation <= 1; `direct_conv_equivalent_frac` = the same time priced in the reference's
                 direct-convolution FLOPs (324 multiplies where the Winograd tiles do 80), which exceeds 1
  extra_configs  configs[2] (local training step as a hipGraph), configs[3] (147x147 and 587x587 image pairs end to end), the global-stage
                 training step at batch 8,
                 each with its own clock and dominant kernel
  dp             the data-parallel training step (configs[4]): at N > 1 K steps of be_hip.train_local.train_step(world=N)
                 with the bucketed RCCL gradient all-reduce (four buckets) overlapped with the backward; at N = 1 the same code, world 1
  cpu_baseline   the oracle on the host cores, the whole 4096-pair workload x 3 runs, median (rank 0, N = 1 only)

Multi-GPU: patch pairs are independent, so every rank runs its own shard of 4096 pairs with NO data-path
collective (weak scaling); RCCL carries the barrier, the MAX of the elapsed time - and, in the dp leg, the gradients.
"""
import argparse
import json
import os
import signal
import subprocess
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")    # dmabuf IPC: what RCCL needs on this driver (before any HIP call)
ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "blurry-edges_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

PAIRS = 4096                          # configs[1]
FLOP_PER_PATCH = 387.716e6            # SURVEY.md A.2 (2*MAC, convs + linears)
FLOP_PER_PAIR = 2 * FLOP_PER_PATCH    # 775.43 MFLOP, SURVEY.md 8d
PEAK_FP32_MFMA_TFLOPS = 157.3         # MI355X_MICROARCH.md: dense fp32 matrix peak
PEAK_HBM_TBPS = 8.0                   # MI355X_MICROARCH.md: HBM3E spec peak (6.29 TB/s is what a float4 copy reaches)
CPU_SAMPLE_PAIRS = 4096               # the WHOLE configs[1] workload (not a quarter of it): ~8 s per run on 16 threads,
CPU_REPEATS = 3                       # 3 repeats, median: ~25 s of CPU work (the task's bound for this leg: 10-30 s)
# kernel id 6 = the transform-domain GEMMs of one Winograd layer (one per position): the hooks count the positions x 2 x tiles x cin x
# cout FLOPs they execute; the same layer as a direct 3x3 convolution on a 6x6 map is 2 x 36 x 9 x n x cin x cout.
# positions x tiles per map = 40 x 2 (8x5 tiles, be_wino_tile_rows() = 6) or 25 x 4 (5x5 tiles): filled in main()
ALGO_OVER_HOOK = {}
TRAFFIC_FILE = next((f for f in (os.path.join(ROOT, "profiles", n) for n in ("r06_pmc_traffic.json", "r05_pmc_traffic.json")) if os.path.exists(f)),
                    os.path.join(ROOT, "profiles", "r06_pmc_traffic.json"))       # the newest counter record of the dominant kernel


def cpu_baseline(x_np, sd_np):
    """The oracle (CPU port of the reference path) timed on this box's host cores on a bounded sample."""
    from oracle import local_stage as ols, render as orr, depth as od
    # the box's CPU share, not the host's core count: affinity mask if set, capped at the 16 cores a 1-GPU box gets
    try:
        ncores = len(os.sched_getaffinity(0))
    except AttributeError:
        ncores = os.cpu_count() or 1
    ncores = max(1, min(ncores, int(os.environ.get("BE_CPU_THREADS", "16"))))
    torch.set_num_threads(ncores)
    sd = ols.to_torch_sd(sd_np)
    p = CPU_SAMPLE_PAIRS
    x = torch.from_numpy(np.concatenate([x_np[:p], x_np[PAIRS:PAIRS + p]]))
    c = od.depth_consts()

    def run():
        with torch.no_grad():
            est = ols.local_stage_forward(sd, x)
            q = orr.wrap_angles10(est)
            col = orr.render_pass_a(q, x)["colors"]
            z = orr.local_depth(c, est[:p], est[p:])
        return est, col, z
    with torch.no_grad():
        ols.local_stage_forward(sd, x[:64])                       # warm the thread pool / allocator
    times = []
    for _ in range(CPU_REPEATS):
        t0 = time.perf_counter()
        est, col, z = run()
        times.append(time.perf_counter() - t0)
    dt = float(np.median(times))
    return dict(value=p / dt, unit="patch-pairs/s", cores=ncores, kind="port",
                sample=f"{p} pairs ({2 * p} CNN patches) of the same synthetic workload, median of {CPU_REPEATS} runs "
                       f"({', '.join('%.2f' % t for t in times)} s), torch {torch.__version__} CPU, {ncores} threads"), (est, col, z), (ols, orr, od)


CHECKPOINT_DIR = os.path.join(ROOT, "checkpoints")
CONVERGED_RECORD = os.path.join(ROOT, "profiles", "r05_converged", "converged_eval.json")


def converged_leg(dev, native, x, x_np, oracle_mods, pairs=512):
    """Part of the cpu_baseline leg (the oracle as CHECKER).  Accuracy keys from a CONVERGED model: checkpoints/pretrained_local_stage.pth is the LocalStage trained here at
    the reference's full schedule (utils/args.py:29-36; tools/converge.sh, profiles/r05_converged/).  Live: the hot path with those
    weights on the bench's own pairs against the oracle with the same weights (logits, depth RMSE over non-flipped pairs, branch
    flips) - parity at trained weights.  Recorded: the end-to-end depth metrics of the HIP and the oracle pipelines on a held-out
    set (tests/converged_eval.py on the GPU box; the oracle pipeline takes 11 s per image pair, too long for this line)."""
    import models, utils
    ols, orr, od = oracle_mods                          # handed over by cpu_baseline(): the one leg of this file that imports oracle/
    path = os.path.join(CHECKPOINT_DIR, "pretrained_local_stage.pth")
    if not os.path.exists(path):
        return None
    sd = torch.load(path, map_location="cpu")
    m = models.LocalStage()
    m.load_state_dict(sd)
    m = m.to(dev).eval()
    dcal = utils.DepthEtas(utils.get_args("eval", argv=[]), dev)
    with torch.no_grad():
        est = m(x)
        depth = native.local_depth(dcal.consts, est)
    torch.cuda.synchronize()
    p = pairs
    xs = torch.from_numpy(np.concatenate([x_np[:p], x_np[PAIRS:PAIRS + p]]))
    with torch.no_grad():
        est_o = ols.local_stage_forward({k: v for k, v in sd.items()}, xs)
        z_o = orr.local_depth(od.depth_consts(), est_o[:p], est_o[p:])
    eh = torch.cat([est[:p], est[PAIRS:PAIRS + p]]).cpu()
    d = depth[:p].cpu() - z_o
    keep = d.abs() / z_o.abs() <= 1e-3
    out = dict(checkpoint="checkpoints/pretrained_local_stage.pth (LocalStage, 1000 epochs x 250 steps of batch 64 on 16 000 synthetic patches)",
               # what "converged" does and does not mean here: LocalStage ran the reference's FULL schedule (its best validation epoch is
               # the 97th); GlobalStage ran 80 of the reference's 350 epochs with the gamma phases compressed in proportion; both were trained on the
               # round-5 generator BEFORE its rasteriser followed the OpenCV rules.  Parity inputs with realistic statistics, not a reproduction of
               # the authors' accuracy.
               qualifier="trained weights: LocalStage full schedule (best epoch 97 of 1000), GlobalStage SHORT schedule (80 of 350 epochs, gamma phases "
                         "compressed), data from the round-5 generator before the OpenCV-rule rasteriser",
               sample_pairs=p, logits_relmax_vs_oracle=float((eh - est_o).abs().max() / est_o.abs().max()),
               depth_rmse_vs_oracle_m=float(torch.sqrt((d[keep] ** 2).mean())), depth_branch_flip_frac=float((~keep).float().mean()))
    try:
        rec = json.load(open(CONVERGED_RECORD))
        out["recorded_end_to_end"] = dict(source=os.path.relpath(CONVERGED_RECORD, ROOT),
                                          **{k: rec[k] for k in ("n_pairs", "oracle_pairs", "hip_pipeline_vs_boundary_depth", "hip_pipeline_vs_image_depth",
                                                                 "same_pairs", "depth_build_minus_oracle") if k in rec})
    except Exception:
        out["recorded_end_to_end"] = None
    return out


def conv_profile(native, fn, iters, peak, per_iter=320):
    """Run fn() `iters` times with a hipEvent pair around every matrix-kernel launch (on the launch stream); returns the
    dominant kernel (most time) with its algorithmic and executed rates, and the list of records.  per_iter: upper bound of
    the matrix-kernel launches of one fn() (two events are created for each)."""
    cap = per_iter * max(1, iters)
    native.profile_enable(cap)
    torch.cuda.synchronize()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    recs = native.profile_read(cap)
    native.profile_enable(0)
    if not recs:
        return None, recs
    # HBM-bound kernels (transforms, pools, pass A) carry their algorithmic bytes: GB/s against the 8 TB/s HBM peak
    hbm = []
    for kid in native.HBM_KERNEL_IDS:
        rr = [r for r in recs if r[0] == kid]
        if rr:
            ms_k, by_k = sum(r[3] for r in rr), sum(r[2] for r in rr)
            hbm.append(dict(kernel=native.KERNEL_NAMES[kid], launches_per_iter=len(rr) // iters, ms_per_iter=round(ms_k / iters, 4),
                            algo_GB_per_iter=round(by_k / iters / 1e9, 3), achieved_TBps=round(by_k / ms_k / 1e9, 3),
                            frac_of_hbm_peak=round(by_k / ms_k / 1e9 / PEAK_HBM_TBPS, 4)))
    recs_all, recs = recs, [r for r in recs if r[0] not in native.HBM_KERNEL_IDS]
    if not recs:
        return None, recs_all
    by_id = {}
    for r in recs:
        by_id[r[0]] = by_id.get(r[0], 0.0) + r[3]
    dom_id = max(by_id, key=by_id.get)
    dom = [r for r in recs if r[0] == dom_id]
    ms = sum(r[3] for r in dom)
    algo = sum(r[1] for r in dom) * ALGO_OVER_HOOK.get(dom_id, 1.0)
    execd = sum(r[4] for r in dom)
    ach = algo / ms / 1e9
    return dict(kernel=native.KERNEL_NAMES[dom_id], kernel_id=dom_id, launches_per_iter=len(dom) // iters,
                avg_launch_ms=round(ms / len(dom), 4), ms_per_iter=round(ms / iters, 4),
                achieved=round(ach, 2), frac=round(ach / peak, 4),
                executed_tflops=round(execd / ms / 1e9, 2), executed_frac=round(execd / ms / 1e9 / peak, 4),
                flop_per_launch=algo / len(dom), executed_flop_per_launch=execd / len(dom),
                algo_bytes_per_launch=sum(r[2] for r in dom) / len(dom),
                all_matrix_kernels_ms_per_iter=round(sum(r[3] for r in recs) / iters, 3), hbm_bound_kernels=hbm), recs


def timed(fn, iters, warmup=2):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


def git_head():
    """the source revision: `git rev-parse` where there is a .git, else what build() baked next to the libraries (the GPU box
    receives a snapshot without .git; `+dirty` = the tree had uncommitted changes when it was built)"""
    try:
        head = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True,
                              timeout=10).stdout.strip()
        if head:
            return head
    except Exception:
        pass
    try:
        info = json.load(open(os.path.join(ROOT, "blurry-edges_amd", "lib", "BUILD_INFO.json")))
        return info["git_head"] + ("+dirty" if info.get("dirty") else "")
    except Exception:
        return None


def build_info():
    """lib/BUILD_INFO.json: what build() recorded about the sources the loaded libraries were compiled from ({} if absent)"""
    try:
        return json.load(open(os.path.join(ROOT, "blurry-edges_amd", "lib", "BUILD_INFO.json")))
    except Exception:
        return {}


def leg_local_training(dev, native, peak, steps=60):
    """configs[2]: local_training.py:99-108 at batch 64 (CNN fwd with batch statistics + LocalLoss + bwd + clip + AdamW) as one
    replayed hipGraph; synthetic basic-shapes patches, portable-generator weights."""
    import models, utils
    from be_hip import dp, synth, train_local
    B = 64
    args = utils.get_args("local_train", argv=[])
    model = models.LocalStage().to(dev)
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)).to(dev) for k, v in synth.local_stage_state_dict().items()})
    helper = utils.PostProcessLocalBase(args, dev)
    from be_hip.optim import ClipAdamW
    opt = ClipAdamW(model.parameters(), lr=args.learning_rate)
    data = {k: torch.from_numpy(v).to(dev) for k, v in synth.synthetic_training_patches(B * 8, seed=1869).items()}
    model.train()
    gstep = train_local.GraphedStep(model, helper, opt, keep_graph=True)
    it = [0]

    def step():
        lo = (it[0] % 8) * B
        it[0] += 1
        return gstep({k: v[lo:lo + B] for k, v in data.items()}, args.beta_bndry_loc, args.beta_smthns)
    first = float(step())                       # eager (creates the optimizer state)
    step()                                      # capture
    ms = timed(step, steps, warmup=3)
    last = float(step())
    # dominant matrix kernel of the step: the same step eagerly (a captured graph has no per-launch events)
    eager = lambda: train_local.train_step(model, helper, opt, {k: v[:B] for k, v in data.items()}, args.beta_bndry_loc, args.beta_smthns)
    eager_ms = timed(eager, 40, warmup=5)                      # the same step as eager launches (4.2 ms in round 2)
    prof, recs = conv_profile(native, eager, 5, peak, per_iter=512)
    try:
        census = native.graph_node_counts(gstep.graph)           # every launch of the replayed step, torch's included
    except Exception as e:
        census = dict(error=f"{type(e).__name__}: {e}")
    return dict(config="configs[2]: local_training.py step, batch 64 (fwd with batch statistics + LocalLoss + bwd + clip 1.0 + AdamW), "
                       "one replayed hipGraph", ms_per_step=round(ms, 4), patches_per_s=round(B / ms * 1e3, 1), steps=steps,
                first_loss=first, last_loss=last, graph_nodes=census, launches_per_step=census.get("kernels"),
                eager_ms_per_step=round(eager_ms, 4),
                matrix_launches_per_step=len(recs) // 5 if recs else None,
                matrix_kernel_ms_per_step=round(sum(r[3] for r in recs) / 5, 4) if recs else None, dominant_kernel=prof)


def leg_image_pairs(dev, native, peak):
    """configs[3]: one synthetic 147x147 pair (4096 patch positions) and one 587x587 pair (36 blocks) through LocalStage ->
    pass A -> GlobalStage -> pass B -> fold (blurry_edges_test.py:117-145, blurry_edges_test_big.py:116-189)."""
    import models, utils
    from be_hip import synth
    from be_hip.pipeline import DepthPipeline
    a = utils.get_args("eval", argv=[])
    lm = models.LocalStage()
    lm.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synth.local_stage_state_dict().items()})
    gm = models.GlobalStage(device=dev)
    gm.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synth.global_stage_state_dict().items()})
    pipe = DepthPipeline(lm.to(dev).eval(), gm.to(dev).eval(), utils.PostProcessGlobalBase(a, dev), utils.DepthEtas(a, dev))
    out = []
    for size, iters, fn_name in ((147, 10, "__call__"), (587, 3, "run_big")):
        img = torch.from_numpy(synth.synthetic_image_pair(size, size, nshape=6 if size == 147 else 24)[0]).to(dev)
        fn = (lambda: pipe(img)) if fn_name == "__call__" else (lambda: pipe.run_big(img))
        ms = timed(fn, iters, warmup=1)
        # per-kernel figures: one stream, so that every launch has the chip to itself (the clock above ran the default schedule)
        timed_streams, lm.streams = lm.streams, 1
        prof, _ = conv_profile(native, fn, 1, peak, per_iter=256 if size == 147 else 4096)
        lm.streams = timed_streams
        if prof:
            prof["schedule"] = f"one stream (the ms_per_pair_of_images clock: {timed_streams} streams)"
        npos = ((size - 21) // 2 + 1) ** 2
        out.append(dict(config=f"configs[3]: {size}x{size} image pair end to end ({npos} patch positions"
                               f"{', 36 blocks of 147x147' if size == 587 else ''}): LocalStage + pass A + GlobalStage + pass B + fold",
                        ms_per_pair_of_images=round(ms, 3), patch_pairs_per_s=round(npos / ms * 1e3, 1), iters=iters,
                        # the reference's 36 blocks overlap by their margin patches (36 x 4096 pairs for 80 656 positions); run_big
                        # runs the local pass once per DISTINCT position, so the CNN sees exactly npos pairs
                        cnn_patch_pairs_per_s=round(npos / ms * 1e3, 1),
                        cnn_patch_pairs_of_the_reference_schedule=npos if size == 147 else 36 * 4096,
                        dominant_kernel=prof))
    return out


def leg_global_training(dev, steps=12):
    """configs[4], global half on one GPU: global_training.py:207-213 at batch 8 (147 x 147 pairs): GlobalStage in train mode
    (dropout 0.1) on the HIP attention / LayerNorm / linear kernels, fused GlobalLoss forward + backward, clip 1.0, AdamW.  Synthetic
    scenes, features from the HIP local pass, portable-generator LocalStage weights, Xavier-initialised GlobalStage."""
    import models, utils
    from be_hip import dp, synth, train_global
    B = 8
    args = utils.get_args("global_train", argv=[])
    args.batch_size = B
    local = models.LocalStage().to(dev)
    local.load_state_dict({k: torch.from_numpy(np.asarray(v)).to(dev) for k, v in synth.local_stage_state_dict().items()})
    local.eval()
    helper, dcal = utils.PostProcessGlobalBase(args, dev), utils.DepthEtas(args, dev)
    data = train_global.make_dataset(B, dev, local, helper)
    torch.manual_seed(1898)
    model = models.GlobalStage(in_parameter_size=args.input_size, out_parameter_size=args.output_size, device=dev).to(dev)
    for p in model.parameters():
        if p.dim() > 1:
            torch.nn.init.xavier_normal_(p)
    from be_hip.optim import ClipAdamW
    opt = ClipAdamW(model.parameters(), lr=1e-4, gather=True)
    gamma = train_global.GammaSchedule(args).final()
    batch = {k: torch.stack([d[k] for d in data]) for k in ("pm", "img_gt", "bndry_dist", "deri", "bndry_depth")}
    model.train()
    losses = []

    def step():
        losses.append(train_global.train_step(model, helper, dcal, opt, batch, gamma))
    ms = timed(step, steps, warmup=2)                      # the warm-up steps grow the allocator by the per-layer workspaces
    return dict(config="configs[4], global half on one GPU: global_training.py step, batch 8 (GlobalStage train mode with dropout 0.1 "
                       "+ fused GlobalLoss + bwd + clip 1.0 + AdamW), eager launches", ms_per_step=round(ms, 3),
                images_per_s=round(B / ms * 1e3, 1), steps=steps, first_loss=float(losses[0]), last_loss=float(losses[-1]))


def leg_dp(dev, native, dist, rank, world, steps, algorithm="allreduce", buckets=4):
    """configs[4], local half: `steps` data-parallel training steps, per-GPU batch 64, gradients averaged by the bucketed
    all-reduce that overlaps the backward (be_hip.dp.GradSync over RCCL); plus the same step without the exchange and the
    exchange alone, so that the exposed communication can be read off.  world = 1 runs the identical code without a group."""
    import models, utils
    from be_hip import dp, synth, train_local
    B = 64
    args = utils.get_args("local_train", argv=[])
    model = models.LocalStage().to(dev)
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)).to(dev) for k, v in synth.local_stage_state_dict().items()})
    helper = utils.PostProcessLocalBase(args, dev)
    from be_hip.optim import ClipAdamW
    opt = ClipAdamW(model.parameters(), lr=args.learning_rate)
    data = {k: torch.from_numpy(v).to(dev) for k, v in synth.synthetic_training_patches(B * 8, seed=1869 + rank).items()}
    model.train()
    own_group = False
    if world == 1 and dist is None and os.environ.get("BE_BENCH_NO_RCCL") is None:
        # N = 1: a one-rank RCCL group, so that the exchange code the N > 1 runs depend on (side stream, events, async handles,
        # broadcasts) runs on every box; the sum over one rank is the identity
        import socket
        import torch.distributed as dist1
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        dist1.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
        own_group = True
    sync = dp.GradSync(world, always=own_group, algorithm=algorithm, groups=dp.GROUPS_BY_COUNT[buckets]) if (world > 1 or own_group) else None
    if world > 1 or own_group:
        dp.broadcast_parameters(model, src=0)
    it = [0]

    def step(use_sync=True):
        lo = (it[0] % 8) * B
        it[0] += 1
        return train_local.train_step(model, helper, opt, {k: v[lo:lo + B] for k, v in data.items()}, args.beta_bndry_loc,
                                      args.beta_smthns, world=world if use_sync else 1, sync=sync if use_sync else None)

    def clock(fn, n):
        for _ in range(3):
            fn()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t = time.perf_counter() - t0
        local.append(round(t / n * 1e3, 4))               # this rank's own clock (the barrier above bounds it by the slowest rank's)
        if dist is not None:
            tt = torch.tensor([t], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            t = float(tt.item())
        return t / n * 1e3
    local = []
    ms = clock(step, steps)
    res = dict(config="configs[4] (LocalStage half): data-parallel local training, per-GPU batch 64, eager launches, "
                      f"{buckets} gradient buckets (be_hip.dp.GROUPS_BY_COUNT) all-reduced on a side stream while the backward runs",
               world=world, global_batch=B * world, steps=steps, dp_step_ms=round(ms, 4),
               patches_per_s=round(B * world / ms * 1e3, 1), allreduce_bytes=4 * sum(p.numel() for p in model.parameters()),
               allreduce_buckets=len(sync.groups) if sync is not None else 0, algorithm=algorithm)
    mine = dict(rank=rank, dp_step_ms=local[-1])
    if world > 1:
        res["compute_only_step_ms"] = round(clock(lambda: step(False), steps), 4)
        flat = torch.zeros(res["allreduce_bytes"] // 4, dtype=torch.float32, device=dev)
        ranges = dp.bucket_ranges([v.numel() for v in model._tensor_list()], sync.groups)

        def exchange():
            for lo, hi in ranges:
                sync.bucket_ready(flat, lo, hi)
            sync.finish()
        mine["compute_only_step_ms"] = local[-1]
        res["allreduce_ms"] = round(clock(exchange, 20), 4)
        mine["allreduce_ms"] = local[-1]
        res["exposed_comm_ms"] = round(res["dp_step_ms"] - res["compute_only_step_ms"], 4)
        res["allreduce_busbw_GBps"] = round(2 * (world - 1) / world * res["allreduce_bytes"] / (res["allreduce_ms"] * 1e-3) / 1e9, 1)
        dp.broadcast_bn_stats(model, src=0)
    if sync is not None:
        # per bucket: bytes and the time from "slice final on the compute stream" to "its collective complete" (events), over a few
        # steps with GradSync's timing switched on - what attributes exposed_comm_ms (the last bucket's time is the exposed part)
        sync.timing = True
        per = None
        for _ in range(5):
            step()
            bt = sync.bucket_times()
            per = bt if per is None else [(b, m0 + m1) for (b, m0), (_, m1) in zip(per, bt)]
        sync.timing = False
        if per:
            res["buckets"] = [dict(bytes=b, issue_to_complete_ms=round(m / 5, 4)) for b, m in per]
            mine["buckets_issue_to_complete_ms"] = [round(m / 5, 4) for _, m in per]
    if own_group:
        res["rccl_one_rank_group"] = True
        res["compute_only_step_ms"] = round(clock(lambda: step(False), steps), 4)
        dp.broadcast_bn_stats(model, src=0)
    if sync is not None:
        # the same step as hipGraph segments (buckets + 1) with the RCCL calls between them (train_local.SegmentedGraphStep): what
        # `be_hip.workflow local_train` runs under torchrun.  Last, because it is the newest code on the RCCL path.
        seg = train_local.SegmentedGraphStep(model, helper, opt, sync, world=world)

        def seg_step():
            lo = (it[0] % 8) * B
            it[0] += 1
            return seg({k: v[lo:lo + B] for k, v in data.items()}, args.beta_bndry_loc, args.beta_smthns)
        res["segmented_graph_step_ms"] = round(clock(seg_step, steps), 4)
        mine["segmented_graph_step_ms"] = local[-1]
        res["segmented_graph_patches_per_s"] = round(B * world / res["segmented_graph_step_ms"] * 1e3, 1)
        # the same step as ONE hipGraph with the bucket collectives captured inside it (SegmentedGraphStep(capture_collectives=True));
        # a failure here (a stack that cannot capture RCCL calls) is recorded, not fatal
        # (N > 1: opt-in with BE_BENCH_CAPTURED_DP=1 - unproven between real ranks, and a capture that fails on one rank only
        #  would leave the others waiting inside a collective)
        if os.environ.get("BE_BENCH_NO_CAPTURED_DP") is None and (world == 1 or os.environ.get("BE_BENCH_CAPTURED_DP") == "1"):
            try:
                cap = train_local.SegmentedGraphStep(model, helper, opt, sync, world=world, capture_collectives=True)

                def cap_step():
                    lo = (it[0] % 8) * B
                    it[0] += 1
                    return cap({k: v[lo:lo + B] for k, v in data.items()}, args.beta_bndry_loc, args.beta_smthns)
                res["captured_graph_step_ms"] = round(clock(cap_step, steps), 4)
                mine["captured_graph_step_ms"] = local[-1]
            except Exception as e:
                res["captured_graph_step_error"] = f"{type(e).__name__}: {e}"[:300]
    # The one multi-GPU run the driver makes has to carry the whole A/B: with more than one rank (or BE_BENCH_DP_AB=1,
    # how the rehearsals exercise this code) the same step is timed, in this process and in this order, with every exchange variant
    # that is bit-tested - all-reduce and reduce-scatter + all-gather, four and two buckets - eager and as hipGraph segments, with the
    # per-bucket issue -> complete times and every rank's own clocks.  The captured-collective form stays opt-in (BE_BENCH_CAPTURED_DP=1:
    # a collective that hangs inside a graph replay cannot be caught in-process).
    if sync is not None and (world > 1 or os.environ.get("BE_BENCH_DP_AB") == "1"):
        ab = []
        for alg, nb in (("allreduce", 4), ("rs_ag", 4), ("allreduce", 2), ("rs_ag", 2)):
            entry = dict(algorithm=alg, buckets=nb)
            mine_v = dict(rank=rank)
            try:
                s2 = dp.GradSync(world, always=own_group, algorithm=alg, groups=dp.GROUPS_BY_COUNT[nb])

                def ab_step():
                    lo = (it[0] % 8) * B
                    it[0] += 1
                    return train_local.train_step(model, helper, opt, {k: v[lo:lo + B] for k, v in data.items()}, args.beta_bndry_loc,
                                                  args.beta_smthns, world=world, sync=s2)
                entry["dp_step_ms"] = round(clock(ab_step, steps), 4)
                mine_v["dp_step_ms"] = local[-1]
                s2.timing = True
                per = None
                for _ in range(5):
                    ab_step()
                    bt = s2.bucket_times()
                    per = bt if per is None else [(b, m0 + m1) for (b, m0), (_, m1) in zip(per, bt)]
                s2.timing = False
                entry["buckets_bytes"] = [4 * (hi - lo) for lo, hi in dp.bucket_ranges([v.numel() for v in model._tensor_list()], s2.groups)]
                # issue -> complete per bucket: events on the RCCL side stream (empty in a gloo rehearsal: that path copies through the host)
                mine_v["buckets_issue_to_complete_ms"] = [round(m / 5, 4) for _, m in per] if per else []
                seg2 = train_local.SegmentedGraphStep(model, helper, opt, s2, world=world)

                def ab_seg():
                    lo = (it[0] % 8) * B
                    it[0] += 1
                    return seg2({k: v[lo:lo + B] for k, v in data.items()}, args.beta_bndry_loc, args.beta_smthns)
                entry["segmented_graph_step_ms"] = round(clock(ab_seg, steps), 4)
                mine_v["segmented_graph_step_ms"] = local[-1]
            except Exception as e:           # a variant that fails on this stack is recorded; the run (and the other variants) go on
                entry["error"] = f"{type(e).__name__}: {e}"[:300]
            if dist is not None and world > 1:
                every = [None] * world
                dist.all_gather_object(every, mine_v)
                entry["per_rank"] = every
            else:
                entry["per_rank"] = [mine_v]
            ab.append(entry)
        ok = [e for e in ab if "segmented_graph_step_ms" in e]
        res["ab"] = ab
        if ok:
            best = min(ok, key=lambda e: e["segmented_graph_step_ms"])
            res["ab_best"] = dict(algorithm=best["algorithm"], buckets=best["buckets"], segmented_graph_step_ms=best["segmented_graph_step_ms"],
                                  patches_per_s=round(B * world / best["segmented_graph_step_ms"] * 1e3, 1))
        res["ab_note"] = ("same process, same order on every rank; *_ms = MAX over ranks, per_rank = every rank's own clock; the one-graph form with "
                          "captured collectives is opt-in: BE_BENCH_CAPTURED_DP=1 python bench.py --gpus N")
    # every rank reports its OWN clocks (rank 0 alone hides a straggler): one line per rank on stderr, and the
    # list of all ranks in rank 0's JSON line.  The `*_ms` figures above are the MAX over ranks of the same clocks.
    print(f"[bench rank {rank}] dp: {json.dumps(mine)}", file=sys.stderr, flush=True)
    if dist is not None and world > 1:
        every = [None] * world
        dist.all_gather_object(every, mine)
        res["per_rank"] = every
    else:
        res["per_rank"] = [mine]
    if own_group:
        dist1.destroy_process_group()
    return res


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks as child processes of this script - RANK /
    LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in their environment, exactly what torch.distributed.run would set - from
    a parent that never touches the GPU, relay their output (rank 0 prints the one JSON line), and return the worst exit code.  A
    rank that dies takes the others with it (by PID) after a grace period, so that a failed run ends instead of hanging in a collective."""
    import socket
    if "BE_LOCAL_DEVICE" not in os.environ and os.environ.get("BE_BENCH_DRYRUN") != "1" and torch.cuda.device_count() < n:      # device_count() does not initialise the GPU
        print(f"bench.py: --gpus {n} but {torch.cuda.device_count()} device(s) visible", file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    # Children must never outlive this process: each one starts in its OWN session (one process group per rank, so
    # its helpers go with it), asks the kernel for SIGKILL when the parent dies (PR_SET_PDEATHSIG - covers a parent that is
    # SIGKILLed itself, which no handler sees), and SIGTERM / SIGINT / SIGHUP to the parent are relayed: terminate, a short grace, kill.
    def _child_setup():
        os.setsid()
        try:
            import ctypes
            ctypes.CDLL("libc.so.6", use_errno=True).prctl(1, signal.SIGKILL, 0, 0, 0)      # PR_SET_PDEATHSIG = 1
        except Exception:
            pass

    procs = []

    def _stop_all(sig):
        for p in procs:
            if p.poll() is None:
                try:
                    os.killpg(p.pid, sig)
                except (ProcessLookupError, PermissionError):
                    pass

    got = {"sig": None}

    def _on_signal(signum, _frame):
        got["sig"] = signum

    old_handlers = {sg: signal.signal(sg, _on_signal) for sg in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP)}
    codes = [None] * n
    try:
        for r in range(n):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                       MASTER_PORT=os.environ.get("MASTER_PORT", str(port)), HSA_ENABLE_IPC_MODE_LEGACY="0", BE_BENCH_LAUNCHER="self")
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, preexec_fn=_child_setup))
        deadline = None
        while any(c is None for c in codes):
            for i, p in enumerate(procs):
                if codes[i] is None:
                    codes[i] = p.poll()
            if got["sig"] is not None and deadline is None:
                print(f"bench.py: signal {got['sig']} received - stopping {sum(c is None for c in codes)} rank(s)", file=sys.stderr, flush=True)
                _stop_all(signal.SIGTERM)
                deadline = time.monotonic() + float(os.environ.get("BE_BENCH_TERM_GRACE", "5"))
            if deadline is None and any(c not in (None, 0) for c in codes):
                deadline = time.monotonic() + float(os.environ.get("BE_BENCH_GRACE", "30"))   # a rank failed: the rest get 30 s to print what they have and leave
            if deadline is not None and time.monotonic() > deadline:
                _stop_all(signal.SIGKILL)
                for i, p in enumerate(procs):
                    if codes[i] is None:
                        codes[i] = p.wait()
            time.sleep(0.2)
    finally:
        # whatever ended the loop (an exception included): no rank process group survives this function
        _stop_all(signal.SIGKILL)
        for p in procs:
            try:
                p.wait(timeout=10)
            except Exception:
                pass
        for sg, h in old_handlers.items():
            signal.signal(sg, h)
    if got["sig"] is not None:
        return 128 + got["sig"]
    return max(abs(c) for c in codes if c is not None) if any(c is not None for c in codes) else 1


def dryrun(args, rank, world):
    """BE_BENCH_DRYRUN=1: the launch path WITHOUT a GPU (a CPU test of what the driver's plain `python bench.py --gpus N` sets in motion):
    rendezvous over gloo, barrier, K trivial timed steps with the MAX over ranks, the `launch` record gathered from every rank, ONE
    JSON line from rank 0, clean teardown.  BE_BENCH_DRYRUN_FAIL_RANK=r makes rank r die before the rendezvous (exit code 5);
    BE_BENCH_DRYRUN_HANG_DIR=d makes every rank write d/rank<r>.pid, ignore SIGTERM and sleep (the dying-parent test)."""
    if os.environ.get("BE_BENCH_DRYRUN_FAIL_RANK") == str(rank):
        os._exit(5)
    if os.environ.get("BE_BENCH_DRYRUN_HANG_DIR"):              # a rank that never finishes (what a stuck collective looks like to the parent)
        with open(os.path.join(os.environ["BE_BENCH_DRYRUN_HANG_DIR"], f"rank{rank}.pid"), "w") as f:
            f.write(str(os.getpid()))
        signal.signal(signal.SIGTERM, signal.SIG_IGN)             # ... and ignores the polite request: the parent has to kill it
        time.sleep(600)
        os._exit(6)
    dist = None
    if world > 1:
        import datetime
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=120))
        dist.barrier()
    t0 = time.perf_counter()
    acc = torch.zeros(1)
    for _ in range(args.steps):
        acc += 1
    if dist is not None:
        dist.barrier()
    elapsed = own = time.perf_counter() - t0
    me = dict(rank=rank, local_rank=int(os.environ.get("LOCAL_RANK", "0")), pid=os.getpid(), ms_per_step=round(own / max(args.steps, 1) * 1e3, 6))
    ranks = [me]
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        ranks = [None] * world
        dist.all_gather_object(ranks, me)
    if rank == 0:
        emit(json.dumps({"dryrun": True, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                         "ms_per_step": round(elapsed / max(args.steps, 1) * 1e3, 6),
                         "launch": dict(launcher=os.environ.get("BE_BENCH_LAUNCHER", "torch.distributed.run" if "TORCHELASTIC_RUN_ID" in os.environ else "none"),
                                        backend="gloo" if dist is not None else None, rccl_ranks=0, ranks=ranks)}))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return 0


_JSON_FD = None


def emit(line):
    """the ONE JSON line, on the process's real stdout"""
    sys.stdout.flush()
    if _JSON_FD is None:
        print(line, flush=True)
    else:
        os.write(_JSON_FD, (line + "\n").encode())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)      # 1.4 s of timed region: long enough for an outside observer to see the load
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--soak-seconds", type=float, default=float(os.environ.get("BE_BENCH_SOAK", "3")),
                    help="after the timed region, keep running the step for this long (untimed; 0 = off)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra_configs / dp legs (profiling runs)")
    ap.add_argument("--chunk", type=int, default=0, help="LocalStage sub-batch (patches); 0 = library default")
    ap.add_argument("--streams", type=int, default=0, help="LocalStage eval schedule: 2 = two half-batches on two side streams "
                                                           "(default), 1 = one stream (what profiles/ are taken with)")
    ap.add_argument("--dp-algorithm", default=os.environ.get("BE_DP_ALGORITHM", "allreduce"), choices=("allreduce", "rs_ag"),
                    help="gradient exchange of the dp leg: one all_reduce per bucket, or reduce_scatter + all_gather (be_hip.dp.GradSync)")
    ap.add_argument("--dp-buckets", type=int, default=int(os.environ.get("BE_DP_BUCKETS", "4")), choices=(2, 4, 5),
                    help="gradient buckets of the dp leg (be_hip.dp.GROUPS_BY_COUNT): 4 (default), 2 or 5")
    ap.add_argument("--only-dp", action="store_true", help="run the dp leg alone (A/B runs of the exchange): no extra_configs, no cpu baseline")
    ap.add_argument("--layers", action="store_true", help="print the per-launch conv timing table to stderr")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "RANK" not in os.environ:
        # the plain command `python bench.py --gpus N`: this process becomes the launcher.  It has made no GPU call (importing torch
        # makes none) and makes none: the N ranks are CHILD processes of the same script, one per GPU, over RCCL
        raise SystemExit(spawn_ranks(args.gpus))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    # stdout carries the JSON line and nothing else: native libraries write there too (RCCL prints a five-line version banner
    # to stdout when its first communicator comes up, gloo a line per rank), so file descriptor 1 is pointed at stderr for the rest of the run and
    # the line goes to a private copy of the real stdout
    global _JSON_FD
    sys.stdout.flush()
    _JSON_FD = os.dup(1)
    os.dup2(2, 1)
    if os.environ.get("BE_BENCH_DRYRUN") == "1":
        return dryrun(args, rank, world)
    if local_rank >= torch.cuda.device_count() >= 1:
        local_rank = 0                             # the launcher masked the devices: each rank sees only its own GPU
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # rehearsal knobs (a 1-GPU box): BE_DIST_BACKEND=gloo BE_LOCAL_DEVICE=0 runs every rank on one device
        backend = os.environ.get("BE_DIST_BACKEND", "nccl")
        if "BE_LOCAL_DEVICE" in os.environ:
            local_rank = int(os.environ["BE_LOCAL_DEVICE"])
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from be_hip import native, synth
    import models, utils
    native.lib()                                   # fails loudly if the HIP library is missing
    wino_rows = native.lib().be_wino_tile_rows()
    wino_mults = 5 * (wino_rows + 2) * (12 // wino_rows)          # multiplies per 6x6 map and channel pair: 80 or 100
    ALGO_OVER_HOOK[6] = (2.0 * 36 * 9) / (2.0 * wino_mults)

    # ---- synthetic workload, resident in HBM before the timed region (each rank: its own shard)
    x_np, z_gt = synth.synthetic_patch_pairs(PAIRS, seed=synth.SEED_DEFAULT + rank)
    sd_np = synth.local_stage_state_dict()
    model = models.LocalStage()
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd_np.items()})
    model = model.to(dev).eval()
    model.chunk = args.chunk                       # per-call option of the C ABI (be_local_stage_opts)
    if args.streams:
        model.streams = args.streams
    helper = utils.PostProcessLocalBase(utils.get_args("local_train", argv=[]), dev)
    dcal = utils.DepthEtas(utils.get_args("eval", argv=[]), dev)
    x = torch.from_numpy(x_np).to(dev)
    colors = torch.empty(2 * PAIRS, 3, 3, device=dev)
    depth = torch.empty(PAIRS, 2, device=dev)
    opts = native.RenderOpts.from_buffer_copy(helper._opts)
    opts.wrap_angles = 1

    def step():
        with torch.no_grad():
            est = model(x)                                                   # [8192,10]
            native.render_colors(opts, est, x, colors=colors)                # [8192,3,3]
            native.local_depth(dcal.consts, est, out=depth)                  # [4096,2]
        return est

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        est = step()
    barrier()
    elapsed = own_elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    pairs_per_s = world * PAIRS * args.steps / elapsed
    # ---- untimed soak (NOT part of `value`): the same step for a few seconds, so that an outside observer sampling the GPU sees
    #      the load (K = 20 steps are 0.24 s of a 40 s run) and so that the line carries a second, longer-window rate to hold
    #      `value` against
    soak = None
    if args.soak_seconds > 0:
        barrier()
        t1 = time.perf_counter()
        k = 0
        while True:
            for _ in range(10):
                step()
            k += 10
            torch.cuda.synchronize()
            if time.perf_counter() - t1 >= args.soak_seconds:
                break
        dt = time.perf_counter() - t1
        soak = dict(steps=k, seconds=round(dt, 3), pairs_per_s_this_rank=round(PAIRS * k / dt, 1))

    # ---- roofline leg: the same K steps again with a hipEvent pair around every matrix-kernel launch (on the launch
    #      stream = torch's current stream); dominant kernel = whichever kernel id takes the most time in the step
    peak = PEAK_FP32_MFMA_TFLOPS
    roof = None
    if rank == 0:
        # per-kernel durations mean something only when nothing else shares the chip: this pass runs the ONE-stream schedule
        # (the timed region above used model.streams, two half-batches on two side streams by default)
        timed_streams, model.streams = model.streams, 1
        prof, recs = conv_profile(native, step, args.steps, peak)
        model.streams = timed_streams
        if args.layers and recs:
            nl = len(recs) // args.steps
            print("launch  kernel  GFLOP(hook)  ms     TFLOP/s(hook)", file=sys.stderr)
            for i in range(nl):
                rr = recs[i::nl]
                ms_i = sum(r[3] for r in rr) / len(rr)
                print(f"{i:4d}  {rr[0][0]:5d}  {rr[0][1] / 1e9:8.2f}  {ms_i:7.4f}  {rr[0][1] / ms_i / 1e9:8.2f}", file=sys.stderr)
        if prof:
            executed_per_pair = sum(r[4] for r in recs) / args.steps / PAIRS
            # achieved / frac = the FLOPs of the algorithm that RUNS (the transform-domain GEMMs of the Winograd tiles) over the
            # launch duration: a utilisation, <= 1.  The same time priced in the reference's direct-convolution FLOPs
            # (SURVEY 8d's per-unit figure) is kept next to it under an explicit name; it exceeds 1 by the arithmetic saving.
            roof = dict(bound="mfma", kernel=prof["kernel"], achieved=prof["executed_tflops"], peak=round(peak, 1), unit="TFLOP/s",
                        frac=prof["executed_frac"], traffic=None,
                        executed_tflops=prof["executed_tflops"], executed_frac=prof["executed_frac"],
                        direct_conv_equivalent_tflops=prof["achieved"], direct_conv_equivalent_frac=prof["frac"],
                        launches_per_step=prof["launches_per_iter"], avg_launch_ms=prof["avg_launch_ms"],
                        dominant_ms_per_step=prof["ms_per_iter"], flop_per_launch=prof["flop_per_launch"],
                        executed_flop_per_launch=prof["executed_flop_per_launch"],
                        algo_bytes_per_launch=prof["algo_bytes_per_launch"],
                        all_matrix_kernels_ms_per_step=prof["all_matrix_kernels_ms_per_iter"],
                        # the HBM-bound kernels of the same pass: algorithmic bytes / hipEvent duration against the 8 TB/s peak
                        hbm_bound_kernels=prof["hbm_bound_kernels"],
                        end_to_end_frac=round(pairs_per_s / world * executed_per_pair / (peak * 1e12), 4),
                        end_to_end_executed_frac=round(pairs_per_s / world * executed_per_pair / (peak * 1e12), 4),
                        end_to_end_direct_conv_equivalent_frac=round(pairs_per_s / world * FLOP_PER_PAIR / (peak * 1e12), 4),
                        executed_mflop_per_pair=round(executed_per_pair / 1e6, 2),
                        schedule="this pass: one stream (model.streams = 1), so that every launch has the chip to itself; the timed region "
                                 f"of `value`: {timed_streams} stream(s) (two half-batches of 4096 patches on two side streams fill each "
                                 "other's tails: DESIGN 3.1f)",
                        definitions="achieved / frac (= executed_*): the FLOPs of the algorithm this kernel runs - the transform-domain "
                                    f"GEMMs of a Winograd layer ({wino_mults} multiplies per 6x6 map and channel pair: "
                                    f"{2 * wino_mults}*n*cin*cout FLOPs), tile padding included; equal to what the MFMA "
                                    "pipe issued (PMC SQ_INSTS_VALU_MFMA_MOPS_F32 x 512) - / its average launch duration (hipEvents on the "
                                    "launch stream, this run) / the dense fp32 matrix peak: the utilisation of the matrix pipe, <= 1; "
                                    "end_to_end_frac prices the whole step that way.  direct_conv_equivalent_*: the same durations priced in "
                                    "the reference's direct-convolution FLOPs (SURVEY 8d: 2*MAC incl. zero-padding taps, 324 multiplies per "
                                    f"map and channel pair where the Winograd tiles do {wino_mults}): exceeds 1 by the arithmetic saving, not by skipped work")
            # HBM bytes per launch come from separate rocprofv3 --pmc passes (profiles/), never from this run.  The record names
            # the kernel sources it was measured on (sha256): while those are byte-identical to the sources of the library that
            # is running, the counters describe the running kernel and `traffic` is filled; otherwise only `traffic_recorded`
            try:
                tr = json.load(open(TRAFFIC_FILE))
                roof["traffic_recorded"] = dict(bytes_per_launch=tr["bytes_per_launch"], source=os.path.relpath(TRAFFIC_FILE, ROOT),
                                                measured_at_git_head=tr.get("git_head"), kernel=tr.get("kernel"),
                                                algo_bytes_per_launch=tr.get("algo_bytes_per_launch"),
                                                kernel_sources=tr.get("kernel_sources"), kernel_source_sha=tr.get("kernel_source_sha"))
                if tr.get("kernel_id") != prof["kernel_id"]:
                    roof["traffic_note"] = f"{os.path.relpath(TRAFFIC_FILE, ROOT)} describes kernel id {tr.get('kernel_id')}, not the dominant one"
                elif (tr.get("kernel_source_sha") and tr["kernel_source_sha"] == build_info().get("kernel_source_sha")
                      and tr.get("wino_tile_rows") == wino_rows):
                    # the record and the RUNNING library were built from the same kernel sources (lib/BUILD_INFO.json, written by
                    # build() at compile time) with the same tile shape: the counters describe the running kernel
                    roof["traffic"] = tr["bytes_per_launch"]
                    roof["traffic_over_algorithmic"] = round(tr["bytes_per_launch"] / tr["algo_bytes_per_launch"], 3)
                else:
                    roof["traffic_note"] = "the dominant kernel's sources changed since the counters were collected: traffic left null"
            except Exception:
                roof["traffic_recorded"] = None

    # ---- what lets a reader verify the run: who ran, on which device, over which library
    prop = torch.cuda.get_device_properties(dev)
    me = dict(rank=rank, local_rank=local_rank, device=f"cuda:{local_rank}", name=prop.name, pid=os.getpid(),
              pci_bus_id=getattr(prop, "pci_bus_id", None), visible=os.environ.get("HIP_VISIBLE_DEVICES", os.environ.get("CUDA_VISIBLE_DEVICES")),
              ms_per_step=round(own_elapsed / args.steps * 1e3, 3))
    ranks_info = [me]
    if dist is not None:
        ranks_info = [None] * world
        dist.all_gather_object(ranks_info, me)
    backend = dist.get_backend() if dist is not None else None
    try:
        rccl_version = ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception:
        rccl_version = None
    launch = dict(launcher=os.environ.get("BE_BENCH_LAUNCHER", "torch.distributed.run" if "TORCHELASTIC_RUN_ID" in os.environ else "none"),
                  backend=backend, rccl_ranks=(dist.get_world_size() if backend == "nccl" else 0), rccl_version=rccl_version,
                  ranks=ranks_info, value_per_gpu=round(pairs_per_s / world, 1))

    # ---- the other configurations, each on its own clock (never part of `value`)
    extra, dp_leg = None, None
    head = git_head()

    def headline(dp_result, cpu=None, more=None):
        """the one JSON line, from whatever has been measured so far"""
        out = {
            "metric": "patch-pairs/s (local CNN + render + depth) on 21x21 synth",
            "value": round(pairs_per_s, 1), "unit": "patch-pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "configs[1]: batch of 4096 synthetic 21x21 two-aperture patch pairs per GPU "
                                   "(8192 CNN patches): LocalStage inference + pass-A colour solve + depth solve",
                       "pairs_per_gpu": PAIRS, "streams": model.streams, "weights": "portable-generator random init (no checkpoint offline)",
                       "sharding": "independent pairs per rank, no data-path collective"},
            "roofline": roof, "cpu_baseline": cpu, "soak": soak, "launch": launch, "extra_configs": extra, "dp": dp_result, "git_head": head,
        }
        out.update(more or {})
        return out

    if not args.no_extra:
        if rank == 0 and not args.only_dp:
            extra = []
            for leg in (lambda: [leg_local_training(dev, native, peak)], lambda: leg_image_pairs(dev, native, peak),
                        lambda: [leg_global_training(dev)]):
                try:
                    extra.extend(leg())
                except Exception as e:                    # a broken side leg must not take the headline number with it
                    extra.append(dict(error=f"{type(e).__name__}: {e}"))
        barrier()
        # N > 1: the dp leg is the first code of this build that moves real data over RCCL between GPUs and it runs AFTER the
        # measurement of `value`: a rank that fails or hangs in it must not cost the run its headline line.  A watchdog on every
        # rank: past the deadline rank 0 prints the line it has (dp: timeout) and every rank leaves without waiting for the others,
        # with exit code 3 - the launcher and the driver see a failed run, the log holds the reason.
        watchdog = None
        if world > 1:
            import threading

            def bail():
                msg = "dp leg did not finish within 180 s (watchdog)"
                print(f"[bench rank {rank}] {msg}", file=sys.stderr, flush=True)
                if rank == 0:
                    emit(json.dumps(headline(dict(error=msg))))
                os._exit(3)                               # the headline is out; a hung collective is NOT a clean run
            watchdog = threading.Timer(180.0, bail)
            watchdog.daemon = True
            watchdog.start()
        try:
            dp_leg = leg_dp(dev, native, dist, rank, world, max(10, args.steps), algorithm=args.dp_algorithm, buckets=args.dp_buckets)
        except Exception as e:
            dp_leg = dict(error=f"{type(e).__name__}: {e}")
            print(f"[bench rank {rank}] dp leg failed: {dp_leg['error']}", file=sys.stderr, flush=True)
            if world > 1:                                 # the other ranks may be waiting inside a collective: do not join them
                if rank == 0:
                    emit(json.dumps(headline(dp_leg)))
                os._exit(3)                               # headline printed, exit code says the dp leg failed
        if watchdog is not None:
            watchdog.cancel()

    cpu = None
    more = {}
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.only_dp:
        cpu, (est_o, col_o, z_o), oracle_mods = cpu_baseline(x_np, sd_np)
        p = CPU_SAMPLE_PAIRS
        with torch.no_grad():
            est = step()
        torch.cuda.synchronize()
        zh = depth[:p].cpu()
        d = (zh - z_o)
        rel = d.abs() / z_o.abs()
        keep = rel <= 1e-3                                       # branch-flipped pairs are counted, not averaged
        try:
            conv = converged_leg(dev, native, x, x_np, oracle_mods)
        except Exception as e:
            conv = dict(error=f"{type(e).__name__}: {e}"[:300])
        more = dict(converged=conv, depth_rmse_vs_oracle_m=float(torch.sqrt((d[keep] ** 2).mean())),
                    depth_branch_flip_frac=float((~keep).float().mean()),
                    logits_relmax_vs_oracle=float((torch.cat([est[:p], est[PAIRS:PAIRS + p]]).cpu() - est_o).abs().max()
                                                  / est_o.abs().max()))

    if rank == 0:
        emit(json.dumps(headline(dp_leg, cpu, more)))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
