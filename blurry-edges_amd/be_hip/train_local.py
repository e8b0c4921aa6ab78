"""Local-stage training loop: the build's counterpart of local_training.py:68-118 (epoch loop, beta schedule,
AdamW lr 6e-5, clip-norm 1, ReduceLROnPlateau) on the HIP training kernels, optionally data-parallel.

    python -m be_hip.train_local --steps 200            (single GPU, synthetic basic-shapes patches)
    torchrun --nproc-per-node 8 -m be_hip.train_local   (per-GPU batch 64, RCCL gradient all-reduce)
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


class BetaSchedule:
    """LocalLoss.update_beta / final_beta (local_training.py:18-30): linear 0 -> max over dynamic_epoch epochs."""

    def __init__(self, max_bndry, max_smooth, dynamic_epoch):
        self.max_b, self.max_s, self.dyn, self.idx = max_bndry, max_smooth, dynamic_epoch, -1
        self.beta_b = self.beta_s = 0.0

    def step(self):
        self.idx += 1
        rate = self.idx / (self.dyn - 1) if self.idx < self.dyn else 1.0
        self.beta_b, self.beta_s = rate * self.max_b, rate * self.max_s

    def final(self):
        self.beta_b, self.beta_s = self.max_b, self.max_s


def train_step(model, helper, opt, batch, beta_b, beta_s, flat=None, world=1, clip=1.0, stats=None, sync=None):
    """One iteration of local_training.py:103-108.  batch: dict of GPU tensors (dataset layouts).
    world > 1: the gradients are averaged over the ranks before clipping.  sync = a be_hip.dp.GradSync: its buckets are each
    all-reduced on a side stream as soon as the backward has finished their slice (overlap); without one, `flat` selects
    the older path - one bucketed all-reduce after the whole backward.
    stats: optional dict that receives `grad_norm` (the total norm clip_grad_norm_ measured, a device scalar)."""
    from . import native, train
    est = model(batch["img_ny"].permute(0, 3, 1, 2))
    opt.zero_grad(set_to_none=True)      # backward then SETS .grad (no fill, no accumulate launch per parameter)
    # LocalLoss forward + analytic backward in one launch (what utils.local_loss wraps in an autograd.Function; called directly
    # here: `loss.backward()` through the Function costs a ones-fill and a `1 * grad` launch per step, and the loss kernel wraps
    # the raw angles itself in float64 - no write-back into est).  local_training.py:105 passes the clean image as both images.
    partial, dloss_dest, _ = native.local_loss(helper.render_opts(False), est.detach(), batch["img_gt"].contiguous(), batch["img_gt"].contiguous(),
                                               batch["bndry_dist"].contiguous(), batch["deri"].contiguous(), beta_b, beta_s, want_grad=True)
    loss = native.local_loss_finish(partial, beta_b, beta_s)
    if sync is not None and (world > 1 or sync.always):
        train.set_grad_hook(sync.bucket_ready, sync.groups)
        try:
            est.backward(dloss_dest)
        finally:
            train.set_grad_hook(None)
        sync.finish()                    # the compute stream waits for the last bucket; .grad now holds the mean
    else:
        est.backward(dloss_dest)
        if flat is not None and world > 1:
            dp.allreduce_mean_(dp.grads_as_flat(list(model.parameters()), flat), world)      # zero-copy when the backward wrote one buffer
    if hasattr(opt, "clip_and_step"):    # be_hip.optim.ClipAdamW: norm + clip + AdamW over the flat gradient buffer, two launches
        norm = opt.clip_and_step(clip)
    else:
        norm = torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=clip, norm_type=2)
        opt.step()
    if stats is not None:
        stats["grad_norm"] = norm
    return loss.detach()


class GraphedStep:
    """train_step captured once as a hipGraph and replayed with the batch copied into static buffers: ~150 short launches
    per step at batch 64, so removing the host launch cost is worth a quarter of the step.  beta_b / beta_s may change
    between replays (they are kernel arguments baked into the capture), so a new epoch value means a new capture - cheap
    next to an epoch, and done lazily.  The optimizer must have been built with capturable=True."""

    def __init__(self, model, helper, opt, flat=None, world=1, keep_graph=False):
        """keep_graph: keep the hipGraph itself next to its executable (native.graph_node_counts reads the launch count)."""
        self.model, self.helper, self.opt, self.flat, self.world = model, helper, opt, flat, world
        self.key = self.graph = self.static = self.loss = None
        self.warm = False
        self.keep_graph = keep_graph

    def __call__(self, batch, beta_b, beta_s):
        if not self.warm:                               # the very first step runs eagerly: it creates the optimizer state
            self.warm = True                            # and every lazily allocated buffer outside any capture
            return train_step(self.model, self.helper, self.opt, batch, beta_b, beta_s, self.flat, self.world)
        # everything that enters the captured launches as a host-side constant: loss weights, learning rates, shapes
        key = (float(beta_b), float(beta_s)) + tuple(float(g["lr"]) for g in self.opt.param_groups) + \
            tuple(tuple(v.shape) for v in batch.values())
        if key != self.key:
            self.static = {k: v.clone() for k, v in batch.items()}
            torch.cuda.synchronize()
            self.graph = torch.cuda.CUDAGraph(keep_graph=True) if self.keep_graph else torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):          # records the step; nothing executes yet
                self.loss = train_step(self.model, self.helper, self.opt, self.static, beta_b, beta_s, self.flat, self.world)
            self.key = key
        else:
            for k in self.static:
                self.static[k].copy_(batch[k])
        self.graph.replay()
        # the replay moved weights and BatchNorm statistics on the device; no tensor version changed
        self.model.invalidate_packed()
        return self.loss.clone()


class SegmentedGraphStep:
    """The data-parallel training step as hipGraph segments (buckets + 1) with the gradient buckets' all-reduces issued between them.

    The eager data-parallel step is bound by its host launches.  DEFAULT form (segments): the step is cut where the backward finishes a gradient bucket
    (be_hip.train.set_grad_hook: fc, layer3, layer2, layer1, conv1 + layer0): segment k ends there, the host replays it and
    issues bucket k's all-reduce on the side stream (GradSync.bucket_ready: event behind the segment, RCCL call), then replays
    segment k + 1, which therefore overlaps that collective; the last segment is the division by the world size, the gradient
    clipping and AdamW.  Host cost per step: buckets + 1 graph launches + one RCCL call per bucket (default: 5 + 4).
    The step runs WITHOUT the autograd engine (forward_train / the fused loss kernel / backward_train called directly): the
    engine would run the backward - and the hook that ends and begins captures - on another thread than the one that began the
    capture.  Same kernels, same order, same results as train_step (tests/test_dp_gpu.py).
    OPT-IN form (capture_collectives=True / BE_DP_CAPTURE=1, round 5): PyTorch 2.10 + RCCL 2.26 DO capture a collective
    (lab/rccl_capture_probe.py), so the whole step - bucket all-reduces included - can be ONE hipGraph.  Proven bit-identical at ONE
    rank only; between real ranks it is unproven (no multi-GPU node has been available), hence opt-in: a capture that fails on one rank
    would leave the others inside a collective."""

    def __init__(self, model, helper, opt, sync, world=1, clip=1.0, capture_collectives=None):
        """capture_collectives (round 5; default: environment BE_DP_CAPTURE, "0"): capture the bucket all-reduces INTO the graph -
        the whole data-parallel step is then ONE hipGraph: the backward's bucket hook issues the collective on GradSync's side stream
        behind an event, all of it recorded by the capture (cross-stream dependencies become graph edges), and a replay is one host
        call.  PyTorch's ProcessGroupNCCL supports capture; lab/rccl_capture_probe.py checks it on this stack."""
        self.model, self.helper, self.opt, self.sync, self.world, self.clip = model, helper, opt, sync, world, clip
        if capture_collectives is None:
            capture_collectives = os.environ.get("BE_DP_CAPTURE", "0") == "1"
        self.capture_collectives = bool(capture_collectives) and sync is not None
        self.key = self.graphs = self.static = self.loss = self.flat = self.ranges = None
        self.warm = 0
        self.stream = None

    def _step(self, batch, beta_b, beta_s, hook):
        """forward + loss + backward (+ hook per finished bucket) + [sync.finish] + clip + AdamW, no autograd engine"""
        from . import native, train
        m = self.model
        t = [v.detach() for v in m._tensor_list()]
        x = batch["img_ny"].permute(0, 3, 1, 2).to(torch.float32).contiguous()
        logits, S = train.forward_train(x, t)
        m.invalidate_packed()
        torch._foreach_add_([b for b in (mod.num_batches_tracked for mod in m.modules()
                                         if isinstance(mod, (torch.nn.BatchNorm2d, torch.nn.BatchNorm1d)))], 1)
        b = logits.shape[0]
        partial, grad_est, _ = native.local_loss(self.helper.render_opts(False), logits, batch["img_gt"].contiguous(),
                                                 batch["img_gt"].contiguous(), batch["bndry_dist"].contiguous(),
                                                 batch["deri"].contiguous(), beta_b, beta_s, want_grad=True)
        loss = native.local_loss_finish(partial, beta_b, beta_s)                                 # utils._LocalLossFn.forward
        train.set_grad_hook(hook, self.sync.groups if self.sync is not None else None)
        try:
            grads = train.backward_train(grad_est, t, S)
        finally:
            train.set_grad_hook(None)
        for p, i in zip(m.parameters(), train.TRAINABLE):
            p.grad = grads[i]
        return loss

    def _clip_step(self):
        if hasattr(self.opt, "clip_and_step"):
            self.opt.clip_and_step(self.clip)
        else:
            torch.nn.utils.clip_grad_norm_(self.model.parameters(), max_norm=self.clip, norm_type=2)
            self.opt.step()

    def _tail(self):
        flat = self.sync.finish() if self.sync is not None else None
        self._clip_step()
        return flat

    def __call__(self, batch, beta_b, beta_s):
        if self.warm < 2:                               # two eager steps: optimizer state, lazily allocated buffers, RCCL warm-up
            self.warm += 1
            loss = self._step(batch, beta_b, beta_s, self.sync.bucket_ready if self.sync is not None else None)
            self._tail()
            return loss.detach()
        key = (float(beta_b), float(beta_s)) + tuple(float(g["lr"]) for g in self.opt.param_groups) + \
            tuple(tuple(v.shape) for v in batch.values())
        if key != self.key:
            self._capture(batch, beta_b, beta_s)
            self.key = key
        else:
            for k in self.static:
                self.static[k].copy_(batch[k])
        if self.capture_collectives:
            self.graphs[0].replay()                      # forward, loss, backward, the bucket collectives, division, clip, AdamW
            self.model.invalidate_packed()
            return self.loss.clone()
        nb = len(self.graphs) - 1
        for k in range(nb):
            self.graphs[k].replay()
            if self.sync is not None:
                self.sync.bucket_ready(self.flat, *self.ranges[k])       # event behind segment k, all-reduce on the side stream
        if self.sync is not None:
            self.sync.wait()                             # the compute stream waits for the collectives (the division is captured)
            self.sync.flat = None
        self.graphs[nb].replay()
        # the replay moved weights and BatchNorm statistics on the device; no tensor version changed (as in GraphedStep)
        self.model.invalidate_packed()
        return self.loss.clone()

    def _capture(self, batch, beta_b, beta_s):
        self.static = {k: v.clone() for k, v in batch.items()}
        torch.cuda.synchronize()
        if self.capture_collectives:
            g = torch.cuda.CUDAGraph()
            seen = []

            def hook_c(flat, lo, hi):                    # bucket final: its collective goes onto the side stream, inside the capture
                seen.append((flat, lo, hi))
                self.sync.bucket_ready(flat, lo, hi)
            if self.stream is None:
                self.stream = torch.cuda.Stream()
            self.stream.wait_stream(torch.cuda.current_stream())
            timing, self.sync.timing = self.sync.timing, False           # timed events cannot be captured
            with torch.cuda.stream(self.stream):
                with torch.cuda.graph(g, stream=self.stream, capture_error_mode="thread_local"):
                    self.loss = self._step(self.static, beta_b, beta_s, hook_c)
                    self.sync.wait()                     # the capturing stream joins the side stream: every collective is an ancestor
                    flat = seen[0][0]
                    self.sync.flat = None
                    if self.world > 1:                   # GradSync.finish's condition (a one-rank group divides by nothing)
                        flat.div_(self.world)
                    self._clip_step()
            self.sync.timing = timing
            torch.cuda.current_stream().wait_stream(self.stream)
            self.graphs, self.flat, self.ranges = [g], flat, [(lo, hi) for _, lo, hi in seen]
            return
        from .dp import DEFAULT_GROUPS
        nb = len(self.sync.groups if self.sync is not None else DEFAULT_GROUPS)
        graphs = [torch.cuda.CUDAGraph() for _ in range(nb + 1)]
        pool = torch.cuda.graph_pool_handle()            # ONE memory pool: tensors made in one segment live on into the next
        seen = []
        state = dict(k=0)

        def hook(flat, lo, hi):                          # bucket k is final: segment k ends here, segment k + 1 begins
            seen.append((flat, lo, hi))
            graphs[state["k"]].capture_end()
            state["k"] += 1
            graphs[state["k"]].capture_begin(pool=pool, capture_error_mode="thread_local")
        if self.stream is None:
            self.stream = torch.cuda.Stream()
        self.stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.stream):
            # thread_local: the process group's watchdog thread polls its events while this thread captures - legal for it,
            # and only this thread's calls are policed
            graphs[0].capture_begin(pool=pool, capture_error_mode="thread_local")
            self.loss = self._step(self.static, beta_b, beta_s, hook)
            assert state["k"] == nb, f"the backward reported {state['k']} gradient buckets, {nb} expected"
            flat = seen[0][0]
            if self.sync is not None and self.world > 1:
                flat.div_(self.world)                    # GradSync.finish's division (same condition), captured
            self._clip_step()
            graphs[nb].capture_end()
        torch.cuda.current_stream().wait_stream(self.stream)
        self.graphs, self.flat, self.ranges = graphs, flat, [(lo, hi) for _, lo, hi in seen]


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--patches", type=int, default=16000)       # 8000 images x 2 (train_val_data_generator.py:188)
    ap.add_argument("--lr", type=float, default=6e-5)
    ap.add_argument("--graph", action="store_true", help="capture the whole step (fwd + loss + bwd + clip + AdamW) in a hipGraph")
    a = ap.parse_args(argv)
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    lr_ = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", lr_))
    torch.cuda.set_device(lr_)
    dev = torch.device("cuda", lr_)
    import models, utils
    args = utils.get_args("local_train", argv=[])
    model = models.LocalStage().to(dev)
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)).to(dev) for k, v in synth.local_stage_state_dict().items()})
    helper = utils.PostProcessLocalBase(args, dev)
    from .optim import ClipAdamW
    opt = ClipAdamW(model.parameters(), lr=a.lr)            # clip + AdamW over the flat gradient buffer (two launches)
    flat = None
    sync = dp.GradSync(world) if world > 1 else None       # four buckets, each all-reduced while the backward goes on
    if world > 1:
        dp.broadcast_parameters(model, src=0)               # replicas start from rank 0's weights and statistics
    data = {k: torch.from_numpy(v).to(dev) for k, v in synth.synthetic_training_patches(a.patches, seed=1869 + rank).items()}
    sched = BetaSchedule(args.beta_bndry_loc, args.beta_smthns, args.dynamic_epoch)
    sched.final()
    model.train()
    losses = []
    step_fn = None
    if a.graph and world > 1:
        # default: graph segments (buckets + 1) with the bucket all-reduces issued between them.  One whole-step capture with the
        # collectives inside works at one rank on this stack (BE_DP_CAPTURE=1) but is unproven between real ranks: opt-in
        step_fn = SegmentedGraphStep(model, helper, opt, sync, world=world)
    elif a.graph:
        # ~130 short launches per step at batch 64: replaying one captured hipGraph removes the host launch cost
        step_fn = GraphedStep(model, helper, opt)
    if step_fn is not None:                                     # eager first step(s) + the capture stay outside the clock
        for _ in range(3):
            step_fn({k: v[:a.batch] for k, v in data.items()}, sched.beta_b, sched.beta_s)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(a.steps):
        lo = (it * a.batch) % (a.patches - a.batch + 1)
        batch = {k: v[lo:lo + a.batch] for k, v in data.items()}
        if step_fn is not None:
            losses.append(step_fn(batch, sched.beta_b, sched.beta_s))
        else:
            losses.append(train_step(model, helper, opt, batch, sched.beta_b, sched.beta_s, flat, world, sync=sync))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        dp.broadcast_bn_stats(model, src=0)                 # what a checkpoint written now would hold on every rank
    if rank == 0:
        print(json.dumps({"metric": "local training patches/s", "value": world * a.batch * a.steps / dt, "n_gpus": world,
                          "ms_per_step": dt / a.steps * 1e3, "graph": bool(a.graph), "first_loss": float(losses[0]), "last_loss": float(losses[-1])}))


if __name__ == "__main__":
    main()
