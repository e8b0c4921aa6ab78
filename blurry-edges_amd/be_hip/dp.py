"""Data-parallel gradient exchange for LocalStage training (one process per GPU, torch.distributed).

The only collective of the training path: an all-reduce (sum, then / world) of the flat fp32
gradient buffer (7 254 122 elements = 29.0 MB) before clip_grad_norm_ and the optimizer step.  The HIP backward writes
the gradients of one layer group after another into consecutive slices of that buffer (fc first, conv1 last) and reports five
completion points (`GRAD_POINTS`: fc 9.5 MB, layer3 6.3, layer2 9.2, layer1 3.3, conv1 + layer0 0.6).  The exchange runs as
BUCKETS made of consecutive points (`DEFAULT_GROUPS`: four buckets - the 0.6 MB head rides with layer1's 3.3 MB, because the
last bucket is the only one whose collective is exposed and a 0.6 MB ring is latency-bound; `GradSync(groups=...)` takes any
other split, e.g. `GRAD_POINTS` itself for round 2's five), each launched on a side stream the moment its slice is final
(`GradSync`): the collective of a bucket overlaps the backward of the layers in front of it.  With RCCL over xGMI (7 point-to-point links x
~153 GB/s per GPU) buckets of several MB keep every link busy; BatchNorm statistics stay per replica during training (the
reference's batch-64 semantics); `broadcast_parameters` aligns the replicas at the start and `broadcast_bn_stats` hands
rank 0's running statistics to everyone before a checkpoint is written."""
from __future__ import annotations

import torch


def fused_adamw():
    """`fused=` argument for torch.optim.AdamW in the training loops: the single-kernel implementation (the default foreach
    one is eight multi-tensor launches per step: 0.4 ms of a 3.1 ms LocalStage step).  BE_FUSED_ADAMW=0: torch's default."""
    import os
    return True if os.environ.get("BE_FUSED_ADAMW", "1") != "0" else None


def flat_grad_buffer(params):
    """One contiguous fp32 buffer holding every .grad (allocated once; grads become views into it)."""
    params = [p for p in params if p.requires_grad]
    total = sum(p.numel() for p in params)
    buf = torch.zeros(total, dtype=torch.float32, device=params[0].device)
    off = 0
    for p in params:
        p.grad = buf[off:off + p.numel()].view_as(p)
        off += p.numel()
    return buf


def allreduce_mean_(buf: torch.Tensor, world: int, group=None, bucket_bytes: int = 16 << 20):
    """In-place average of the flat gradient buffer over the ranks, in buckets of bucket_bytes."""
    if world == 1:
        return buf
    import torch.distributed as dist
    if buf.is_cuda and dist.get_backend(group) != "nccl":      # gloo rehearsal on one GPU: no device transport in this build
        host = buf.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
        buf.copy_(host)
        buf.div_(world)
        return buf
    n = max(1, bucket_bytes // 4)
    handles = [dist.all_reduce(buf[i:i + n], op=dist.ReduceOp.SUM, group=group, async_op=True)
               for i in range(0, buf.numel(), n)]
    for h in handles:
        h.wait()
    buf.div_(world)
    return buf


def copy_grads_into(buf: torch.Tensor, params):
    """For parameters whose .grad was replaced by a fresh tensor (autograd does that when .grad is None):
    copy into the flat buffer and re-point .grad at the view."""
    off = 0
    for p in params:
        if not p.requires_grad:
            continue
        view = buf[off:off + p.numel()].view_as(p)
        if p.grad is not None and p.grad.data_ptr() != view.data_ptr():
            view.copy_(p.grad)
            p.grad = view
        off += p.numel()


def grads_as_flat(params, fallback: torch.Tensor | None = None):
    """The gradients of `params` as ONE flat tensor to all-reduce.  The HIP backward (be_hip.train.backward_train) writes every
    gradient into consecutive slices of one buffer, in parameter order, and autograd keeps those views as .grad: then this is
    that buffer, zero-copy.  Otherwise (torch autograd allocated the gradients one by one) they are copied into `fallback`
    (from flat_grad_buffer) and re-pointed at it, as copy_grads_into does."""
    params = [p for p in params if p.requires_grad]
    if params and all(p.grad is not None and p.grad.is_contiguous() for p in params):
        g0 = params[0].grad
        st, off, ok = g0.untyped_storage(), g0.storage_offset(), True
        for p in params:
            g = p.grad
            if g.untyped_storage().data_ptr() != st.data_ptr() or g.storage_offset() != off or g.dtype != g0.dtype:
                ok = False
                break
            off += g.numel()
        if ok:
            total = off - g0.storage_offset()
            return torch.empty(0, dtype=g0.dtype, device=g0.device).set_(st, g0.storage_offset(), (total,))
    if fallback is None:
        raise RuntimeError("grads_as_flat: gradients are not one flat buffer and no fallback buffer was given")
    copy_grads_into(fallback, params)
    return fallback


# ------------------------------------------------------------------------------- overlapped, bucketed gradient exchange
# tensor-index ranges (native.local_stage_pack order, 86 tensors) whose gradients are final at the backward's five completion
# points, in completion order: fc.1/fc.2/fc.4, layer3, layer2, layer1, conv1 + layer0
GRAD_POINTS = ((78, 86), (60, 78), (42, 60), (24, 42), (0, 24))
# buckets = unions of consecutive completion points.  Round 3 default: the head (0.6 MB) is merged into layer1's bucket.
DEFAULT_GROUPS = ((78, 86), (60, 78), (42, 60), (0, 42))
# two buckets (round 5 A/B: fewer hipGraph segments in SegmentedGraphStep, fewer collectives; the exposed one is 13.2 MB):
# fc + layer3 (15.8 MB) | layer2 + layer1 + layer0 + conv1 (13.2 MB)
TWO_GROUPS = ((60, 86), (0, 60))
GROUPS_BY_COUNT = {2: TWO_GROUPS, 4: DEFAULT_GROUPS, 5: GRAD_POINTS}


def check_groups(groups):
    """groups must tile [0, 86) from the tail, each a union of consecutive completion points; returns them as a tuple."""
    groups = tuple((int(a), int(b)) for a, b in groups)
    cuts = {a for a, _ in GRAD_POINTS} | {86}
    hi = 86
    for a, b in groups:
        if b != hi or a >= b or a not in cuts:
            raise ValueError(f"gradient buckets must be consecutive unions of {GRAD_POINTS} from the tail, got {groups}")
        hi = a
    if hi != 0:
        raise ValueError(f"gradient buckets must cover every tensor, got {groups}")
    return groups


def bucket_ranges(numels, groups=DEFAULT_GROUPS, trainable=None):
    """[(lo, hi)] float offsets into the flat gradient buffer, in the order the LocalStage backward completes them.
    numels: element count of each of the 86 tensors (native.local_stage_pack order); groups: tensor-index ranges (see
    DEFAULT_GROUPS / GRAD_POINTS); trainable: be_hip.train.TRAINABLE."""
    if trainable is None:
        from .train import TRAINABLE as trainable
    off, start = 0, {}
    for i in trainable:
        start[i] = off
        off += numels[i]
    start[86] = off
    first = lambda a: start[min(i for i in list(trainable) + [86] if i >= a)]
    return [(first(a), first(b)) for a, b in groups]


class GradSync:
    """All-reduce of the flat gradient buffer in buckets that start while the backward is still running.

        sync = GradSync(world)                                    # once; groups= picks the bucket boundaries
        train.set_grad_hook(sync.bucket_ready, sync.groups)       # the backward calls it with (flat, lo, hi) as each bucket becomes final
        loss.backward(); sync.finish()                            # all buckets reduced and divided by world; .grad views are the means

    GPU tensors + RCCL ("nccl"): every bucket is issued under a side stream that waits for an event recorded on the compute
    stream right behind the kernels that wrote the slice; finish() makes the compute stream wait for the collectives.
    gloo (CPU rehearsal, tests): the same calls, synchronous; GPU tensors are staged through the host because gloo has no
    device transport on this build.
    timing=True (RCCL path): each bucket also records when its slice became final (compute stream) and when its collective
    completed (side stream); `bucket_times()` then gives, per bucket, bytes and issue -> complete milliseconds of the last step -
    what a first multi-GPU run needs to attribute `exposed_comm_ms` (a bucket's time includes waiting for the bucket before it:
    the collectives of one step are serialised on the side stream)."""

    ALGORITHMS = ("allreduce", "rs_ag")

    def __init__(self, world, group=None, always=False, groups=DEFAULT_GROUPS, timing=False, algorithm="allreduce"):
        """always: run the exchange even for world == 1 (a one-rank group: the sum is the identity) - how the RCCL path is
        exercised on a single GPU (tests, bench.py at N = 1).
        algorithm: "allreduce" (default) - one `all_reduce` per bucket, RCCL chooses ring / tree; "rs_ag" - the same sum written as
        `reduce_scatter_tensor` into this rank's 1/world shard of the bucket followed by `all_gather_into_tensor` back into the
        bucket (on a full xGMI mesh a direct reduce-scatter + all-gather moves (world-1)/world of a bucket per phase
        spread over all 7 links, where a single ring pushes 2 (world-1)/world of it through each link in turn); the < world elements
        that do not divide take a tiny all_reduce.  Same sums: both give every rank the same buffer, and for world = 2 the very
        same bits.  A knob for the first run on a real 8-GPU node (`bench.py --dp-algorithm`, BE_DP_ALGORITHM)."""
        if algorithm not in self.ALGORITHMS:
            raise ValueError(f"GradSync: algorithm must be one of {self.ALGORITHMS}, got {algorithm!r}")
        self.world, self.group, self.always = world, group, always
        self.groups = check_groups(groups)
        self.timing = timing
        self.algorithm = algorithm
        self._shards = {}
        self.handles, self.flat, self.side = [], None, None
        self.bytes = 0
        self._events, self._timed_side = [], False

    def bucket_ready(self, flat, lo, hi):
        if (self.world == 1 and not self.always) or hi <= lo:
            return
        import torch.distributed as dist
        self.flat = flat
        piece = flat[lo:hi]
        self.bytes += piece.numel() * 4
        if piece.is_cuda and dist.get_backend(self.group) == "nccl":
            if self.side is None:
                self.side = torch.cuda.Stream(device=flat.device)
            ev = torch.cuda.Event(enable_timing=self.timing)
            ev.record()                                               # behind the last kernel that wrote flat[lo:hi]
            with torch.cuda.stream(self.side):
                self.side.wait_event(ev)
                hs = self._issue(piece, dist, (lo, hi))
                if self.timing:
                    for h in hs:
                        h.wait()                                      # the SIDE stream waits for the collective(s) ...
                    done = torch.cuda.Event(enable_timing=True)
                    done.record()                                     # ... so this event is their completion
                    self._events.append((piece.numel() * 4, ev, done))
                    self._timed_side = True
                else:
                    self.handles.extend(hs)
        elif piece.is_cuda:
            host = piece.cpu()
            for h in self._issue(host, dist, (lo, hi)):
                h.wait()
            piece.copy_(host)
        else:
            self.handles.extend(self._issue(piece, dist, (lo, hi)))

    def _issue(self, piece, dist, bucket):
        """the bucket's sum over the ranks, in place, as asynchronous collective(s) in issue order -> their handles.
        bucket = (lo, hi) of the piece in the flat buffer: names the scratch shard of the rs_ag form."""
        w = dist.get_world_size(self.group)
        if self.algorithm == "allreduce" or w == 1 or piece.numel() < w:
            return [dist.all_reduce(piece, op=dist.ReduceOp.SUM, group=self.group, async_op=True)]
        r = dist.get_rank(self.group)
        s = piece.numel() // w                                        # shard length; [w * s, n) is the remainder
        main = piece[:w * s]
        # one scratch shard PER BUCKET, reused every step.  Not per shard length: two equal-sized buckets of one step
        # would share it, and only the RCCL path orders bucket A's gather (which reads the shard) before bucket B's scatter (which
        # writes it) - on the gloo path the gather may still be running on a worker thread
        key = (piece.device, bucket, s)
        shard = self._shards.get(key)
        if shard is None:
            shard = self._shards[key] = torch.empty(s, dtype=piece.dtype, device=piece.device)
        hs = [dist.reduce_scatter_tensor(shard, main, op=dist.ReduceOp.SUM, group=self.group, async_op=True)]
        # collectives of one process group run in issue order (one stream under RCCL; gloo's worker queue): the gather reads the
        # shard the scatter wrote, the remainder's all_reduce touches elements neither of them does
        if not piece.is_cuda:
            hs[0].wait()
        hs.append(dist.all_gather_into_tensor(main, shard, group=self.group, async_op=True))
        if piece.numel() > w * s:
            hs.append(dist.all_reduce(piece[w * s:], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        return hs

    def wait(self):
        """The current (compute) stream waits for every collective issued so far; nothing is divided."""
        for h in self.handles:
            h.wait()                                                  # nccl: the current stream waits for the collective
        self.handles = []
        if self._timed_side:
            torch.cuda.current_stream().wait_stream(self.side)
            self._timed_side = False

    def finish(self):
        """-> the averaged flat buffer (None if nothing was exchanged)."""
        self.wait()
        flat, self.flat = self.flat, None
        if flat is not None and self.world > 1:
            flat.div_(self.world)
        return flat

    def bucket_times(self):
        """[(bytes, ms from 'slice final' to 'collective complete')] of the buckets issued since the last call (timing=True;
        synchronises on the completion events)."""
        out = []
        for nbytes, a, b in self._events:
            b.synchronize()
            out.append((nbytes, a.elapsed_time(b)))
        self._events = []
        return out


def broadcast_parameters(model, src=0, group=None):
    """Every parameter AND buffer (BatchNorm statistics, counters) of rank `src` to all ranks: replicas start identical
    (SURVEY 8e).  One-off, ~100 small broadcasts."""
    import torch.distributed as dist
    with torch.no_grad():
        for t in list(model.parameters()) + list(model.buffers()):
            if t.is_cuda and dist.get_backend(group) != "nccl":
                host = t.detach().cpu()
                dist.broadcast(host, src, group=group)
                t.copy_(host)
            else:
                dist.broadcast(t.detach(), src, group=group)
    if hasattr(model, "invalidate_packed"):
        model.invalidate_packed()


def broadcast_bn_stats(model, src=0, group=None):
    """Rank `src`'s BatchNorm running statistics to every rank (before torch.save: all replicas then write the same
    checkpoint; during training the statistics stay per replica)."""
    import torch.distributed as dist
    with torch.no_grad():
        for b in model.buffers():
            if b.is_cuda and dist.get_backend(group) != "nccl":
                host = b.detach().cpu()
                dist.broadcast(host, src, group=group)
                b.copy_(host)
            else:
                dist.broadcast(b.detach(), src, group=group)
    if hasattr(model, "invalidate_packed"):
        model.invalidate_packed()
