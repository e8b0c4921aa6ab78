"""Data-parallel gradient exchange for LocalStage training (one process per GPU, torch.distributed).

The only collective of the training path: ONE all-reduce (sum, then / world) of the flat fp32
gradient buffer (7 254 122 elements = 29.0 MB) before clip_grad_norm_ and the optimizer step.  With RCCL over
xGMI (7 point-to-point links x ~153 GB/s per GPU) the buffer is sent as a few large buckets so every link carries
several MB per phase; BatchNorm statistics stay per replica (the reference's batch-64 semantics)."""
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
