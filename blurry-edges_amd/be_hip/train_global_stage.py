"""GlobalStage training step on the GPU: train-mode forward (dropout at the four sites of every encoder layer) and
the full backward, orchestrated over the kernels of libblurry_edges_hip (be_attn.hip, be_conv.hip, be_train.hip).

Counterpart of `est = model(pm); loss.backward()` in global_training.py:207-213.  `GlobalStageTrainFn.apply(src, pe,
seed, p, nhead, eps, *params)` is a torch.autograd.Function whose backward returns the gradient of every parameter in
the reference's layout (nn.TransformerEncoderLayer, post-norm, ReLU; models/global_stage.py:28-32), so AdamW /
clip_grad_norm_ / a gradient all-reduce work on it unchanged.  torch allocates buffers and supplies the stream.

Dropout masks are counter-based (hash of element index, seed and site number): the elementwise sites re-derive them in the
backward, the attention forward leaves one keep BIT per probability in the layer's workspace for its backward; the [L,L]
attention probabilities themselves are never stored (recomputed from the saved log-sum-exp).  Dropout sites per layer i:
  16 i + 0  attention probabilities (one sub-site per batch*head inside the kernel)
  16 i + 1  dropout1 (self-attention branch before norm1)
  16 i + 2  dropout  (after the FFN ReLU)
  16 i + 3  dropout2 (FFN branch before norm2)
"""
from __future__ import annotations

import os

import torch

from . import native
from .native import check, dptr, lib, stream_ptr
from .train import _col_sum, _new, _wgrad

PER_LAYER = 12       # in_proj w,b | out_proj w,b | linear1 w,b | linear2 w,b | norm1 w,b | norm2 w,b


def parameter_list(model):
    """The 102 tensors in the order GlobalStageTrainFn expects (= state-dict order of models/global_stage.py)."""
    out = [model.in_src_projection.weight, model.in_src_projection.bias]
    for lyr in model.encoder.layers:
        a = lyr.self_attn
        out += [a.in_proj_weight, a.in_proj_bias, a.out_proj.weight, a.out_proj.bias, lyr.linear1.weight, lyr.linear1.bias,
                lyr.linear2.weight, lyr.linear2.bias, lyr.norm1.weight, lyr.norm1.bias, lyr.norm2.weight, lyr.norm2.bias]
    out += [model.encoder.norm.weight, model.encoder.norm.bias, model.generator.weight, model.generator.bias]
    return out


def _linear(x, w, b, act=0):
    pw, pb = native.conv_pack(w.contiguous(), b.contiguous())
    return native.linear(x, pw, pb, w.shape[0], act=act)


def _dgrad_lin(dy, w, residual=None):
    """dy [T,cout] -> dx [T,cin] (+ residual) for y = x W^T."""
    cout, cin = w.shape
    dev = dy.device
    pw = _new(lib().be_conv_dgrad_packed_floats(cout, cin, 1), dev)
    pb = _new((cin + 31) // 32 * 32, dev)
    check(lib().be_conv_pack_dgrad_f32(dptr(w.contiguous()), cout, cin, 1, 0, dptr(pw), dptr(pb), stream_ptr(dev)),
          "be_conv_pack_dgrad_f32")
    return native.linear(dy, pw, pb, cin, residual=residual)


def _wgrad_lin(x, dy, w_shape):
    t = x.shape[0]
    return _wgrad(x.view(t, 1, 1, x.shape[1]), dy.view(t, 1, 1, dy.shape[1]), tuple(w_shape), 1)


class _Side:
    """The parameter gradients of the encoder's linears are off the data-gradient chain (nothing in the backward reads them) and
    their launches fill half the chip (one 128 x 128 weight tile x <= 128 row slices): they run on a side stream, one after the
    other (they share the scratch buffer), beside the attention / LayerNorm kernels of the main chain.  fork(): the side stream
    waits for everything enqueued so far; join(): the main stream waits for the side stream (end of the backward).  Operands are
    kept alive until the join (the caching allocator would otherwise hand a freed dy to the main stream while the side stream still
    reads it).  BE_GLOBAL_SIDE=0: everything on the caller's stream."""
    enabled = os.environ.get("BE_GLOBAL_SIDE", "1") != "0"
    stream = None
    scratch = None                   # the side stream's own be_train scratch (the main stream's kernels use train._Scratch meanwhile)
    keep = []

    @classmethod
    def get(cls, dev):
        if cls.stream is None or cls.stream.device != dev:
            cls.stream = torch.cuda.Stream(device=dev)
            cls.scratch = torch.empty(lib().be_train_scratch_bytes() // 4, dtype=torch.float32, device=dev)
        return cls.stream

    @classmethod
    def join(cls, dev):
        if cls.enabled and cls.stream is not None and cls.keep:
            torch.cuda.current_stream(dev).wait_stream(cls.stream)
        cls.keep = []


def _lin_param_grads(x, dy, w_shape):
    """(dW, db) of y = x W^T + b.  Channel counts that are multiples of 128 (every linear of an encoder layer): two launches
    (be_linear_param_grads_f32) on the side stream (_Side; the caller joins before it hands the gradients out); otherwise the
    weight-gradient GEMM and the column sum on their own."""
    cout, cin = w_shape
    t = x.shape[0]
    if cout % 128 == 0 and cin % 128 == 0 and t >= 256:
        from .train import _Scratch
        dev = x.device
        dw, db = _new((cout, cin), dev), _new(cout, dev)
        sc = _Scratch.get(dev)
        dyc = dy.contiguous()
        side = _Side.enabled and not torch.cuda.is_current_stream_capturing()
        if side:
            st = _Side.get(dev)
            st.wait_stream(torch.cuda.current_stream(dev))
            _Side.keep.append((x, dyc, dw, db))
            sc = _Side.scratch
            with torch.cuda.stream(st):
                check(lib().be_linear_param_grads_f32(dptr(x, "x"), dptr(dyc, "dy"), dptr(dw), dptr(db), t, cin, cout, dptr(sc),
                                                      sc.numel() * 4, stream_ptr(dev)), "be_linear_param_grads_f32")
            return dw, db
        check(lib().be_linear_param_grads_f32(dptr(x, "x"), dptr(dyc, "dy"), dptr(dw), dptr(db), t, cin, cout, dptr(sc),
                                              sc.numel() * 4, stream_ptr(dev)), "be_linear_param_grads_f32")
        return dw, db
    return _wgrad_lin(x, dy, w_shape), _col_sum(dy)


class _Packs:
    """Packed weights of every linear of one training step - forward form and data-gradient form - written by ONE launch
    (be_conv_pack_jobs_f32) at the top of the forward (71 single pack launches per step before); buffers and the device job table
    are built once per parameter set.  Keys are positions in the parameter list t."""
    cache = {}

    def __init__(self, t, cin):
        dev = t[0].device
        nl = (len(t) - 6) // PER_LAYER
        self.pad = (-cin) % 32
        d = t[0].shape[0]
        self.w_in = torch.zeros(d, cin + self.pad, dtype=torch.float32, device=dev)       # in-projection with zero input columns
        cout = t[-2].shape[0]
        self.cp = (cout + 31) // 32 * 32
        self.wg = torch.zeros(self.cp, t[-2].shape[1], dtype=torch.float32, device=dev)   # generator with zero output rows (dgrad / wgrad)
        self.fwd, self.dg, jobs = {}, {}, []

        def add(key, w, b, want_dgrad, w_dgrad=None):
            co, ci = w.shape
            pw, pb = _new(lib().be_conv_packed_floats(co, ci, 1), dev), _new((co + 31) // 32 * 32, dev)
            self.fwd[key] = (pw, pb, co)
            jobs.append(native.PackJob(dptr(w), dptr(b), None, None, None, None, dptr(pw), dptr(pb), 0.0, co, ci, 1, 0, 0))
            if want_dgrad:
                wd = w if w_dgrad is None else w_dgrad
                co2, ci2 = wd.shape
                dw, db = _new(lib().be_conv_dgrad_packed_floats(co2, ci2, 1), dev), _new((ci2 + 31) // 32 * 32, dev)
                self.dg[key] = (dw, db, ci2)
                jobs.append(native.PackJob(dptr(wd), None, None, None, None, None, dptr(dw), dptr(db), 0.0, co2, ci2, 1, 0, 1))
        add(0, self.w_in, t[1], False)
        for i in range(nl):
            base = 2 + PER_LAYER * i
            for k in (0, 2, 4, 6):
                add(base + k, t[base + k], t[base + k + 1], True)
        add(len(t) - 2, t[-2], t[-1], True, w_dgrad=self.wg)
        arr = (native.PackJob * len(jobs))(*jobs)
        self.njobs = len(jobs)
        self.table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
        self.keep = list(t)

    @classmethod
    def get(cls, t, cin):
        key = tuple(v.data_ptr() for v in t) + (cin,)
        p = cls.cache.get(key)
        if p is None:
            for v in t:
                if not v.is_contiguous():
                    raise RuntimeError("GlobalStage training: parameters must be contiguous")
            cls.cache.clear()                                           # one model at a time keeps its buffers
            p = cls.cache[key] = cls(t, cin)
        return p

    def pack(self, t, cin):
        self.w_in[:, :cin].copy_(t[0])
        self.wg[:t[-2].shape[0]].copy_(t[-2])
        dev = self.table.device
        check(lib().be_conv_pack_jobs_f32(dptr(self.table, "job table", (torch.uint8,)), self.njobs, stream_ptr(dev)),
              "be_conv_pack_jobs_f32")

    def lin(self, key, x, act=0):
        pw, pb, co = self.fwd[key]
        return native.linear(x, pw, pb, co, act=act)

    def dgrad(self, key, dy, residual=None):
        dw, db, ci = self.dg[key]
        return native.linear(dy, dw, db, ci, residual=residual)


def attention_train_fwd(qkv, B, L, H, p, seed, ws=None, l_valid=None):
    dev = qkv.device
    need = lib().be_attention_train_workspace_floats(B, L, H)
    if ws is None or ws.numel() < need:
        ws = _new(need, dev)
    o = native.ops()
    if o is not None:
        out, lse = o.attention_train_fwd(qkv, ws, B, L, L if l_valid is None else int(l_valid), H, float(p), int(seed) & 0xffffffff)
        return out, lse, ws
    out = _new((B * L, H * 16), dev)
    lse = _new((B * H, L), dev)
    check(lib().be_attention_train_fwd_f32(dptr(qkv, "qkv"), dptr(out), dptr(lse), dptr(ws), B, L, L if l_valid is None else int(l_valid),
                                           H, float(p), int(seed) & 0xffffffff, stream_ptr(dev)), "be_attention_train_fwd_f32")
    return out, lse, ws


def attention_bwd(qkv, out, lse, dout, B, L, H, p, seed, ws=None, operands_ready=False, l_valid=None):
    """operands_ready: ws is the workspace attention_train_fwd returned for THIS qkv and is untouched since."""
    dev = qkv.device
    need = lib().be_attention_train_workspace_floats(B, L, H)
    if ws is None or ws.numel() < need:
        ws, operands_ready = _new(need, dev), False
    o = native.ops()
    if o is not None:
        return o.attention_bwd(qkv, out, lse, dout.contiguous(), ws, _bwd_scratch(B, L, H, dev), bool(operands_ready), B, L,
                               L if l_valid is None else int(l_valid), H, float(p), int(seed) & 0xffffffff), ws
    dqkv = torch.empty_like(qkv)
    check(lib().be_attention_bwd_f32(dptr(qkv, "qkv"), dptr(out), dptr(lse), dptr(dout.contiguous(), "dout"), dptr(dqkv),
                                     dptr(ws), dptr(_bwd_scratch(B, L, H, dev)), int(bool(operands_ready)), B, L,
                                     L if l_valid is None else int(l_valid), H, float(p), int(seed) & 0xffffffff, stream_ptr(dev)),
          "be_attention_bwd_f32")
    return dqkv, ws


_SCRATCH = {}


def _bwd_scratch(B, L, H, dev):
    """The partial-dQ buffer of the attention backward (537 MB at batch 8): its contents mean nothing between calls, so ONE
    buffer per device serves every layer and step (calls on one stream are ordered)."""
    need = lib().be_attention_bwd_scratch_floats(B, L, H)
    key = (dev.type, dev.index)
    buf = _SCRATCH.get(key)
    if buf is None or buf.numel() < need:
        buf = _SCRATCH[key] = _new(need, dev)
    return buf


def attention_dropout_mask(B, L, H, p, seed, dev):
    m = _new((B * H, L, L), dev)
    check(lib().be_attention_dropout_mask_f32(dptr(m), B, L, H, float(p), int(seed) & 0xffffffff, stream_ptr(dev)),
          "be_attention_dropout_mask_f32")
    return m


def attention_keep_bits(B, L, H, p, seed, dev):
    """The attention dropout decisions in the packed layout the training forward leaves in its workspace, from the formula:
    int16 [B*H, L/32, L/32, 64]."""
    k = torch.empty((B * H, L // 32, L // 32, 32), dtype=torch.int32, device=dev)          # two halfwords per int32
    check(lib().be_attention_keep_bits_u16(dptr(k), B, L, H, float(p), int(seed) & 0xffffffff, stream_ptr(dev)),
          "be_attention_keep_bits_u16")
    return k.view(torch.int16)


def workspace_keep_bits(ws, B, L, H):
    """View of the keep bits inside a workspace attention_train_fwd returned."""
    off = lib().be_attention_train_keep_offset_floats(B, L, H)
    n = B * H * (L // 32) ** 2 * 64
    return ws[off:off + n // 2].view(torch.int16).view(B * H, L // 32, L // 32, 64)


def dropout(x, p, seed, site, gate=None):
    y = torch.empty_like(x)
    check(lib().be_dropout_f32(dptr(x, "x"), dptr(gate), dptr(y), x.numel(), float(p), int(seed) & 0xffffffff, int(site),
                               stream_ptr(x.device)), "be_dropout_f32")
    return y


def add_layernorm_train(x, res, gamma, beta, eps, p, seed, site):
    rows, d = x.shape
    v, y = torch.empty_like(x), torch.empty_like(x)
    check(lib().be_add_layernorm_train_f32(dptr(x, "x"), dptr(res), dptr(gamma), dptr(beta), dptr(v), dptr(y), rows, d,
                                           float(eps), float(p), int(seed) & 0xffffffff, int(site), stream_ptr(x.device)),
          "be_add_layernorm_train_f32")
    return v, y


def layernorm_bwd(dy, v, gamma, eps, p, seed, site, want_dv=True, want_dx=True):
    """-> (dv, dx, dgamma, dbeta)"""
    rows, d = v.shape
    dev = v.device
    dv = torch.empty_like(v) if want_dv else None
    dx = torch.empty_like(v) if want_dx else None
    part = _new(lib().be_layernorm_bwd_partial_floats(rows, d), dev)
    check(lib().be_layernorm_bwd_f32(dptr(dy.contiguous(), "dy"), dptr(v), dptr(gamma), dptr(dv), dptr(dx), dptr(part), rows, d,
                                     float(eps), float(p), int(seed) & 0xffffffff, int(site), stream_ptr(dev)),
          "be_layernorm_bwd_f32")
    gb = _col_sum(part.view(-1, 2 * d))
    return dv, dx, gb[:d].clone(), gb[d:].clone()


def forward_train(src, pe, seed, p, H, eps, t, l_valid=None):
    """src [B,L,cin]; t = parameter_list order (detached).  -> (out [B,L,cout], saved).
    l_valid: real tokens per sequence when the caller padded L up to a multiple of 128 (keys beyond are masked)."""
    B, L, cin = src.shape
    T = B * L
    dev = src.device
    nl = (len(t) - 6) // PER_LAYER
    packs = _Packs.get(t, cin)
    packs.pack(t, cin)                                         # every linear, both forms, one launch
    pad = packs.pad
    x0 = src.reshape(T, cin).to(torch.float32)
    if pad:
        x0 = torch.cat([x0, x0.new_zeros(T, pad)], dim=1)
    x0 = x0.contiguous()
    h = packs.lin(0, x0)
    native.add_pe_(h, pe[:L].contiguous(), B)
    S = dict(x0=x0, layers=[], shape=(B, L, cin), l_valid=l_valid, packs=packs)
    for i in range(nl):
        base = 2 + PER_LAYER * i
        wqkv, bqkv, wo, bo, w1, b1, w2, b2, g1, be1, g2, be2 = t[base:base + PER_LAYER]
        qkv = packs.lin(base, h)
        a, lse, ws = attention_train_fwd(qkv, B, L, H, p, seed + 16 * i, l_valid=l_valid)   # one workspace per layer: the split q/k/v
                                                                              # (6 x 16 B per token and head) are reused by the backward
        sa = packs.lin(base + 2, a)
        v1, h1 = add_layernorm_train(sa, h, g1, be1, eps, p, seed, 16 * i + 1)
        f = packs.lin(base + 4, h1, act=2)
        fd = dropout(f, p, seed, 16 * i + 2) if p > 0 else f
        y2 = packs.lin(base + 6, fd)
        v2, h2 = add_layernorm_train(y2, h1, g2, be2, eps, p, seed, 16 * i + 3)
        S["layers"].append((h, qkv, a, lse, v1, h1, f, v2, ws))
        h = h2
    gN, bN, wg, bg = t[-4:]
    vN, hN = add_layernorm_train(h, None, gN, bN, eps, 0.0, seed, 0)
    out = packs.lin(len(t) - 2, hN)
    S.update(vN=vN, hN=hN)
    return out.view(B, L, -1), S


def backward_train(dout, seed, p, H, eps, t, S):
    """dout [B,L,cout] -> list of gradients in the order of t."""
    B, L, cin = S["shape"]
    T = B * L
    dev = dout.device
    nl = (len(t) - 6) // PER_LAYER
    grads = [None] * len(t)
    gN, bN, wg, bg = t[-4:]
    cout = wg.shape[0]
    # generator: the 12 output columns padded to 32 so that the MFMA dgrad / wgrad paths take it
    packs = S["packs"]
    cp = packs.cp
    dy = dout.reshape(T, cout).to(torch.float32)
    dyp = torch.zeros(T, cp, dtype=torch.float32, device=dev)
    dyp[:, :cout] = dy
    grads[-2] = _wgrad_lin(S["hN"], dyp, packs.wg.shape)[:cout].contiguous()
    grads[-1] = _col_sum(dyp)[:cout].contiguous()
    d_hN = packs.dgrad(len(t) - 2, dyp)
    dh, _, grads[-4], grads[-3] = layernorm_bwd(d_hN, S["vN"], gN, eps, 0.0, seed, 0, want_dx=False)
    for i in reversed(range(nl)):
        base = 2 + PER_LAYER * i
        wqkv, bqkv, wo, bo, w1, b1, w2, b2, g1, be1, g2, be2 = t[base:base + PER_LAYER]
        h_in, qkv, a, lse, v1, h1, f, v2, ws = S["layers"][i]
        dv2, dy2, grads[base + 10], grads[base + 11] = layernorm_bwd(dh, v2, g2, eps, p, seed, 16 * i + 3)
        fd = dropout(f, p, seed, 16 * i + 2) if p > 0 else f
        grads[base + 6], grads[base + 7] = _lin_param_grads(fd, dy2, w2.shape)
        dfd = packs.dgrad(base + 6, dy2)
        df = dropout(dfd, p, seed, 16 * i + 2, gate=f)                    # dropout' and relu' in one pass
        grads[base + 4], grads[base + 5] = _lin_param_grads(h1, df, w1.shape)
        dh1 = packs.dgrad(base + 4, df, residual=dv2)
        dv1, dsa, grads[base + 8], grads[base + 9] = layernorm_bwd(dh1, v1, g1, eps, p, seed, 16 * i + 1)
        grads[base + 2], grads[base + 3] = _lin_param_grads(a, dsa, wo.shape)
        da = packs.dgrad(base + 2, dsa)
        dqkv, _ = attention_bwd(qkv, a, lse, da, B, L, H, p, seed + 16 * i, ws, operands_ready=True, l_valid=S["l_valid"])
        grads[base], grads[base + 1] = _lin_param_grads(h_in, dqkv, wqkv.shape)
        dh = packs.dgrad(base, dqkv, residual=dv1)
    x0 = S["x0"]
    grads[0] = _wgrad_lin(x0, dh, (t[0].shape[0], x0.shape[1]))[:, :cin].contiguous()
    grads[1] = _col_sum(dh)
    _Side.join(dev)                                                       # the linears' parameter gradients are complete
    return grads


class GlobalStageTrainFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src, pe, seed, p, nhead, eps, l_valid, *params):
        t = [v.detach() for v in params]
        out, S = forward_train(src.detach(), pe, int(seed), float(p), int(nhead), float(eps), t, l_valid=l_valid)
        ctx.S, ctx.t, ctx.cfg = S, t, (int(seed), float(p), int(nhead), float(eps))
        return out

    @staticmethod
    def backward(ctx, dout):
        seed, p, nhead, eps = ctx.cfg
        grads = backward_train(dout.contiguous(), seed, p, nhead, eps, ctx.t, ctx.S)
        ctx.S = None
        return (None,) * 7 + tuple(grads)
