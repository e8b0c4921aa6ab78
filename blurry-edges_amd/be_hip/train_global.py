"""Global-stage training loop: the build's counterpart of global_training.py:173-221 (+ global_data_pre_cal.py).

GlobalStage runs under PyTorch autograd (boundary kept: stock ops in train mode); the loss and its gradient w.r.t.
the network output come from the fused HIP GlobalLoss (utils.global_loss); the inputs of the network (the
[P,38] normalised local features, global_data_pre_cal.py:10-33) come from the HIP local pass.

    python -m be_hip.train_global --steps 20 --images 4 --batch 1
    torchrun --nproc-per-node 8 -m be_hip.train_global            (per-GPU batch, RCCL gradient all-reduce)
"""
from __future__ import annotations

import argparse
import json
import os
import time

import numpy as np

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC for RCCL, before any HIP call
import torch  # noqa: E402

from . import dp, synth

KEYS = ("color", "color_cons", "bndry_cons", "smthns", "smthns_cons", "bndry_loc", "depth")


class GammaSchedule:
    """GlobalLoss.update_gamma / final_gamma (global_training.py:25-60): three-phase interpolation of the 7 weights."""

    def __init__(self, args):
        self.rng = {k: getattr(args, "gamma_" + k) for k in KEYS}
        self.dyn = args.dynamic_epoch
        self.idx = -1
        self.gamma = {}

    def step(self, idx_update=True):
        if idx_update:
            self.idx += 1
        d = self.dyn
        if self.idx < d[0]:
            rate, case = self.idx / (d[0] - 1), 0
        elif self.idx < d[1]:
            rate, case = 1.0, 0
        elif self.idx < d[2]:
            rate, case = (self.idx - d[1]) / (d[2] - d[1] - 1), 1
        else:
            rate, case = 1.0, 1
        self.gamma = {k: r[case] + rate * (r[case + 1] - r[case]) for k, r in self.rng.items()}
        return self.gamma

    def final(self):
        self.gamma = {k: r[-1] for k, r in self.rng.items()}
        return self.gamma


def make_dataset(n_images, dev, local_module, helper_local, seed0=1898):
    """Synthetic global training set on the GPU + the pre-computed local features (global_data_pre_cal.py)."""
    from .pipeline import DepthPipeline
    pipe = DepthPipeline(local_module, None, helper_local, None)
    out = []
    for i in range(n_images):
        s = {k: torch.from_numpy(v).to(dev) for k, v in synth.synthetic_global_sample(147, 147, seed=seed0 + i).items()}
        with torch.no_grad():
            _, _, _, pm = pipe.local_pass(s["img_ny"].permute(0, 3, 1, 2).contiguous())      # [P,38]
        s["pm"] = pm
        out.append(s)
    return out


def train_step(model, helper, dcal, opt, batch, gamma, flat=None, world=1, clip=1.0):
    """One iteration of global_training.py:207-213; batch: dict of stacked GPU tensors."""
    import utils
    est = model(batch["pm"])
    opt.zero_grad(set_to_none=True)      # backward then SETS .grad (no fill, no accumulate launch per parameter)
    loss = utils.global_loss(helper, dcal, est, batch["img_gt"], batch["img_gt"], batch["bndry_dist"], batch["deri"],
                             batch["bndry_depth"], gamma, empty_mask="zero")      # a batch without depth-mask pixels skips the term
    #                                                     (the reference's 0 / 0 = NaN there would end the run)
    loss.backward()
    if flat is not None:
        dp.allreduce_mean_(dp.grads_as_flat(list(model.parameters()), flat), world)      # zero-copy when the backward wrote one buffer
    if hasattr(opt, "clip_and_step"):    # be_hip.optim.ClipAdamW(gather=True): norm + clip + AdamW in two launches (+ one gathering copy)
        opt.clip_and_step(clip)
    else:
        torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=clip, norm_type=2)
        opt.step()
    return loss.detach()


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--images", type=int, default=4)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--lr", type=float, default=1e-4)
    a = ap.parse_args(argv)
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    lr_ = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", lr_))
    torch.cuda.set_device(lr_)
    dev = torch.device("cuda", lr_)
    import models, utils
    args = utils.get_args("global_train", argv=[])
    args.batch_size = a.batch
    torch.manual_seed(1898 + rank)
    local = models.LocalStage().to(dev)
    local.load_state_dict({k: torch.from_numpy(np.asarray(v)).to(dev) for k, v in synth.local_stage_state_dict().items()})
    local.eval()
    helper = utils.PostProcessGlobalBase(args, dev)
    dcal = utils.DepthEtas(args, dev)
    data = make_dataset(a.images, dev, local, helper, seed0=1898 + 1000 * rank)
    model = models.GlobalStage(in_parameter_size=args.input_size, out_parameter_size=args.output_size, device=dev).to(dev)
    for p in model.parameters():
        if p.dim() > 1:
            torch.nn.init.xavier_normal_(p)
    from .optim import ClipAdamW
    opt = ClipAdamW(model.parameters(), lr=a.lr, gather=True)     # clip_grad_norm_ + AdamW of global_training.py:213-214, fused
    flat = dp.flat_grad_buffer(model.parameters()) if world > 1 else None
    sched = GammaSchedule(args)
    gamma = sched.final()
    model.train()
    losses = []

    def batch_of(it):
        idx = [(it * a.batch + j) % a.images for j in range(a.batch)]
        return {k: torch.stack([data[i][k] for i in idx]) for k in ("pm", "img_gt", "bndry_dist", "deri", "bndry_depth")}

    for it in range(2):                                   # warm-up (kernel loading, allocator), not timed, weights untouched
        with torch.no_grad():
            model(batch_of(it)["pm"])
    # the first two REAL steps are not timed: they grow the allocator by the per-layer training workspaces (q / k / v splits and
    # keep bits, ~270 MB per layer at batch 8) and load the backward kernels
    skip = min(2, a.steps - 1)
    for it in range(a.steps):
        if it == skip:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        losses.append(train_step(model, helper, dcal, opt, batch_of(it), gamma, flat, world))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    losses = [float(l) for l in losses]
    timed = a.steps - skip
    if rank == 0:
        print(json.dumps({"metric": "global training images/s", "value": world * a.batch * timed / dt, "n_gpus": world,
                          "ms_per_step": dt / timed * 1e3, "timed_steps": timed, "first_loss": losses[0], "last_loss": losses[-1]}))


if __name__ == "__main__":
    main()
