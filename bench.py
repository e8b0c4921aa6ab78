#!/usr/bin/env python3
"""bench.py -- patch-pairs/s through the Blurry-Edges hot path on MI355X.

One "step" = one pass of the hot path over one batch of synthetic input, BASELINE.json configs[1]:
a batch of 4096 synthetic 21x21 two-aperture patch pairs (8192 CNN patches) per GPU through
  LocalStage CNN (fp32-MFMA implicit GEMM)  ->  wedge renderer pass A (colours, ridge solve)  ->  DfD depth solve
with the input already resident in HBM.  Prints ONE JSON line (rank 0).

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Multi-GPU: patch pairs are independent, so every rank runs its own shard of 4096 pairs with NO data-path
collective (weak scaling); RCCL is used only for the barrier and the MAX-over-ranks of the elapsed time.
"""
import argparse
import json
import os
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
# opt-in experiment (BE_CONV_PRECISION=bf16x3, never the default): six bf16 MFMAs per fp32 product -> the price is a
# sixth of the dense bf16 peak (~2.5 PFLOP/s)
PEAK_BF16X3_TFLOPS = 2500.0 / 6
CPU_SAMPLE_PAIRS = 4096                 # the whole batch: ~10 s on 16 threads


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
    run_small = lambda: ols.local_stage_forward(sd, x[:64])        # warm the thread pool / allocator
    with torch.no_grad():
        run_small()
    t0 = time.perf_counter()
    est, col, z = run()
    dt = time.perf_counter() - t0
    return dict(value=p / dt, unit="patch-pairs/s", cores=ncores, kind="port",
                sample=f"{p} pairs ({2 * p} CNN patches) of the same synthetic workload, 1 run, {dt:.1f} s, "
                       f"torch {torch.__version__} CPU, {ncores} threads"), (est, col, z)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--chunk", type=int, default=0, help="LocalStage sub-batch (patches); 0 = library default")
    ap.add_argument("--layers", action="store_true", help="print the per-launch conv timing table to stderr")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch multi-GPU runs with torch.distributed.run (one process per GPU)")
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
    if args.chunk:
        native.check(native.lib().be_local_stage_set_chunk(args.chunk))

    # ---- synthetic workload, resident in HBM before the timed region (each rank: its own shard)
    x_np, z_gt = synth.synthetic_patch_pairs(PAIRS, seed=synth.SEED_DEFAULT + rank)
    sd_np = synth.local_stage_state_dict()
    model = models.LocalStage()
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd_np.items()})
    model = model.to(dev).eval()
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
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- roofline leg: the same K steps again with a hipEvent pair around every conv launch (on the launch
    #      stream = torch's current stream); dominant kernel = whichever kernel id takes the most time in the step (since v10
    #      the 25 transform-domain GEMMs of the Winograd layers, k_wino_gemm_ws / k_wino_gemm)
    roof = None
    if rank == 0:
        per_step = 15 * ((2 * PAIRS + 1023) // 1024 + 1)
        cap = per_step * args.steps
        native.profile_enable(cap)
        torch.cuda.synchronize()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        recs = native.profile_read(cap)
        native.profile_enable(0)
        if args.layers and recs:
            nl = len(recs) // args.steps
            print("launch  kernel  GFLOP      ms     TFLOP/s", file=sys.stderr)
            for i in range(nl):
                rr = recs[i::nl]
                ms_i = sum(r[3] for r in rr) / len(rr)
                print(f"{i:4d}  {rr[0][0]:5d}  {rr[0][1] / 1e9:8.2f}  {ms_i:7.4f}  {rr[0][1] / ms_i / 1e9:8.2f}", file=sys.stderr)
        by_id = {}
        for r in recs:
            by_id[r[0]] = by_id.get(r[0], 0.0) + r[3]
        dom_id = max(by_id, key=by_id.get) if by_id else 0           # dominant = the kernel with the most time in the step
        dom = [r for r in recs if r[0] == dom_id]
        conv_ms = sum(r[3] for r in recs) / args.steps
        if dom:
            peak = PEAK_FP32_MFMA_TFLOPS if model.conv_precision == "f32" else PEAK_BF16X3_TFLOPS
            avg_ms = sum(r[3] for r in dom) / len(dom)
            avg_flop = sum(r[1] for r in dom) / len(dom)
            ach = avg_flop / (avg_ms * 1e-3) / 1e12
            roof = dict(bound="mfma", kernel=native.KERNEL_NAMES[dom_id], achieved=round(ach, 2),
                        peak=round(peak, 1), unit="TFLOP/s", frac=round(ach / peak, 4),
                        traffic=None, launches_per_step=len(dom) // args.steps,
                        avg_launch_ms=round(avg_ms, 4), flop_per_launch=avg_flop,
                        algo_bytes_per_launch=sum(r[2] for r in dom) / len(dom),
                        # MFMA work actually issued (pixel-major tiles skip the zero-padding taps the algorithmic
                        # count includes; tile padding counted): the busy fraction of the matrix pipe at nominal clock
                        mfma_executed_tflops=round(sum(r[4] for r in dom) / sum(r[3] for r in dom) / 1e9, 2),
                        mfma_executed_frac=round(sum(r[4] for r in dom) / sum(r[3] for r in dom) / 1e9 / peak, 4),
                        all_conv_ms_per_step=round(conv_ms, 3),
                        end_to_end_frac=round(PAIRS * args.steps / elapsed * FLOP_PER_PAIR / (peak * 1e12), 4),
                        note="layers 1-3 run as Winograd F(3x3,3x3): achieved / frac count the FLOPs the dominant kernel "
                             "actually performs (the 25 transform-domain GEMMs: 100 multiplies per map and channel pair where "
                             "the direct form has 324); end_to_end_frac prices the reference's direct-convolution FLOPs "
                             "(775.43 MFLOP per pair) and can therefore exceed 1")
            pmc = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
            if os.path.exists(pmc):                 # HBM bytes per launch from the separate --pmc passes
                try:
                    roof["traffic"] = json.load(open(pmc)).get("bytes_per_launch")
                except Exception:
                    pass

    cpu = None
    extra = {}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu, (est_o, col_o, z_o) = cpu_baseline(x_np, sd_np)
        p = CPU_SAMPLE_PAIRS
        zh = depth[:p].cpu()
        d = (zh - z_o)
        rel = d.abs() / z_o.abs()
        keep = rel <= 1e-3                                       # branch-flipped pairs are counted, not averaged
        extra = dict(depth_rmse_vs_oracle_m=float(torch.sqrt((d[keep] ** 2).mean())),
                     depth_branch_flip_frac=float((~keep).float().mean()),
                     logits_relmax_vs_oracle=float((torch.cat([est[:p], est[PAIRS:PAIRS + p]]).cpu() - est_o).abs().max()
                                                   / est_o.abs().max()))

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        out = {
            "metric": "patch-pairs/s (local CNN + render + depth) on 21x21 synth",
            "value": round(world * PAIRS * args.steps / elapsed, 1), "unit": "patch-pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32" if model.conv_precision == "f32" else "f32 operands split exactly into 3 bf16 pieces, 6 bf16 MFMAs per product, f32 accumulate (opt-in experiment)", "data": "synthetic",
            "config": {"workload": "configs[1]: batch of 4096 synthetic 21x21 two-aperture patch pairs per GPU "
                                   "(8192 CNN patches): LocalStage inference + pass-A colour solve + depth solve",
                       "pairs_per_gpu": PAIRS, "weights": "portable-generator random init (no checkpoint offline)",
                       "sharding": "independent pairs per rank, no data-path collective"},
            "roofline": roof, "cpu_baseline": cpu,
        }
        out.update(extra)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
