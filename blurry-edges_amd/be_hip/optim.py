"""ClipAdamW: `clip_grad_norm_(params, max_norm)` + `torch.optim.AdamW.step()` (local_training.py:107-108, torch defaults) as
three HIP launches over the ONE flat gradient buffer the LocalStage backward writes (be_hip.train.backward_train): the stock
path is 13 multi-tensor launches, 0.16 ms of a 2.5 ms step.

It is a torch.optim.Optimizer: param_groups (ReduceLROnPlateau moves `lr`), zero_grad and state_dict work as usual; state[p]
holds `exp_avg` / `exp_avg_sq` as views of two flat buffers and one shared device scalar `step`.  `load_state_dict` (a resume:
its own checkpoints or a stock `torch.optim.AdamW` one - same keys) copies the loaded moments and step count INTO those flat
buffers, which are what the kernel reads, and points state[p] back at the views (the inherited method rebuilt
`state` with fresh tensors the kernel never saw).  The kernel writes the parameters through a device pointer table; every
eager `clip_and_step` therefore bumps their version counters (`torch.autograd.graph.increment_version`, host side only) so that
caches keyed on `_version` - LocalStage's BN-folded weight pack - see the update; a replayed hipGraph cannot do that, which is
why GraphedStep / SegmentedGraphStep call `model.invalidate_packed()` themselves.  `step()`
alone is AdamW without clipping; `clip_and_step(max_norm)` is the fused tail of the training step and returns the gradient norm
before clipping (a device scalar).  Gradients that are not one flat buffer in parameter order are refused (no silent fallback)
unless the optimizer was built with `gather=True` (the GlobalStage loops: one multi-tensor copy gathers them first)."""
from __future__ import annotations

import ctypes as C

import torch

from . import native
from .native import check, dptr, lib, stream_ptr


class ClipAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, write_back=True, gather=False):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        if len(self.param_groups) != 1:
            raise ValueError("ClipAdamW: one parameter group (the training scripts of the reference use one)")
        ps = self.param_groups[0]["params"]
        if not ps or any((not p.is_cuda) or p.dtype != torch.float32 or not p.is_contiguous() for p in ps):
            raise ValueError("ClipAdamW: contiguous float32 parameters on the GPU")
        dev = ps[0].device
        self._n = sum(p.numel() for p in ps)
        self._m = torch.zeros(self._n, dtype=torch.float32, device=dev)
        self._v = torch.zeros(self._n, dtype=torch.float32, device=dev)
        self._step = torch.zeros(1, dtype=torch.float32, device=dev)
        self._norm = torch.zeros(1, dtype=torch.float32, device=dev)
        self.write_back = bool(write_back)
        self.gather, self._gbuf, self._gviews = bool(gather), None, None
        chunk = lib().be_adam_chunk()
        self._partial = torch.empty((self._n + chunk - 1) // chunk, dtype=torch.float64, device=dev)
        entries, self._offsets, off = [], [], 0
        for p in ps:
            n = p.numel()
            self._offsets.append(off)
            self.state[p] = dict(step=self._step, exp_avg=self._m[off:off + n].view_as(p), exp_avg_sq=self._v[off:off + n].view_as(p))
            for lo in range(0, n, chunk):
                c = min(chunk, n - lo)
                entries.append(native.AdamEntry(p.data_ptr() + 4 * lo, self._m.data_ptr() + 4 * (off + lo),
                                                self._v.data_ptr() + 4 * (off + lo), off + lo, c))
            off += n
        self._ptrs = tuple(p.data_ptr() for p in ps)
        arr = (native.AdamEntry * len(entries))(*entries)
        self._nentries = len(entries)
        self._table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)

    def _flat_grad(self):
        ps = self.param_groups[0]["params"]
        if tuple(p.data_ptr() for p in ps) != self._ptrs:
            raise RuntimeError("ClipAdamW: a parameter's storage moved after the optimizer was built")
        g0 = ps[0].grad
        if g0 is None:
            raise RuntimeError("ClipAdamW: no gradients (call backward first)")
        base = g0.data_ptr()
        flat = all(p.grad is not None and p.grad.data_ptr() == base + 4 * off and p.grad.is_contiguous() for p, off in zip(ps, self._offsets))
        if flat:
            return base
        if not self.gather:
            raise RuntimeError("ClipAdamW: gradients must be consecutive slices of one flat buffer in parameter order "
                               "(be_hip.train.backward_train writes them that way); build the optimizer with gather=True for a "
                               "backward that allocates them one by one, or use torch.optim.AdamW")
        # gather mode (GlobalStage: its backward returns one tensor per parameter): one multi-tensor copy into the optimizer's own
        # flat buffer, and .grad re-pointed at the views - so the clipped gradient is what .grad shows afterwards, as with
        # clip_grad_norm_
        if any(p.grad is None for p in ps):
            raise RuntimeError("ClipAdamW: a parameter has no gradient")
        if self._gbuf is None:
            self._gbuf = torch.empty(self._n, dtype=torch.float32, device=self._m.device)
            self._gviews = [self._gbuf[off:off + p.numel()].view_as(p) for p, off in zip(ps, self._offsets)]
        torch._foreach_copy_(self._gviews, [p.grad for p in ps])
        for p, v in zip(ps, self._gviews):
            p.grad = v
        return self._gbuf.data_ptr()

    @torch.no_grad()
    def clip_and_step(self, max_norm=1.0, grad_scale=1.0):
        """-> the total gradient norm before clipping (device scalar, valid until the next call)."""
        g = self.param_groups[0]
        base = self._flat_grad()
        dev = self._m.device
        o = native.ops()
        if o is not None:
            ps = self.param_groups[0]["params"]
            flat = torch.empty(0, dtype=torch.float32, device=dev).set_(ps[0].grad.untyped_storage(), ps[0].grad.storage_offset(), (self._n,))
            o.clip_adamw(self._table, self._nentries, flat, self._partial, float(max_norm), float(grad_scale), float(g["lr"]),
                         float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]), float(g["weight_decay"]), self._step, self._norm,
                         self.write_back)
            self._bump_versions()
            return self._norm[0]
        check(lib().be_clip_adamw_f32(dptr(self._table, "table", (torch.uint8,)), self._nentries, C.c_void_p(base), self._n,
                                      dptr(self._partial, "partial", (torch.float64,)), self._partial.numel(), float(max_norm),
                                      float(grad_scale), float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]),
                                      float(g["weight_decay"]), dptr(self._step), dptr(self._norm), int(self.write_back),
                                      stream_ptr(dev)), "be_clip_adamw_f32")
        self._bump_versions()
        return self._norm[0]

    def _bump_versions(self):
        if self.write_back and not torch.cuda.is_current_stream_capturing():
            torch.autograd.graph.increment_version(self.param_groups[0]["params"])

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        ps = self.param_groups[0]["params"]
        if tuple(p.data_ptr() for p in ps) != self._ptrs:
            raise RuntimeError("ClipAdamW: a parameter's storage moved after the optimizer was built")
        steps = set()
        with torch.no_grad():
            for p, off in zip(ps, self._offsets):
                st, n = self.state.get(p), p.numel()
                if not st:                                              # a parameter the checkpoint had no state for: zeros, step 0
                    self._m[off:off + n].zero_(); self._v[off:off + n].zero_()
                else:
                    self._m[off:off + n].copy_(st["exp_avg"].reshape(-1)); self._v[off:off + n].copy_(st["exp_avg_sq"].reshape(-1))
                    steps.add(float(st["step"]))
                self.state[p] = dict(step=self._step, exp_avg=self._m[off:off + n].view_as(p), exp_avg_sq=self._v[off:off + n].view_as(p))
            if len(steps) > 1:
                raise ValueError(f"ClipAdamW.load_state_dict: one shared step count expected, the checkpoint holds {sorted(steps)}")
            self._step.fill_(steps.pop() if steps else 0.0)

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        self.clip_and_step(max_norm=0.0)
        return loss
