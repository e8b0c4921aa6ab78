"""LocalStage training step on the GPU: train-mode forward (batch-statistics BatchNorm) and the full backward,
orchestrated layer by layer over the training kernels of libblurry_edges_hip (be_train.hip, be_conv.hip).

Counterpart of `est = model(x); loss.backward()` in local_training.py:103-106: `LocalStageTrainFn.apply(x, *tensors)`
is a torch.autograd.Function whose backward returns the gradient of every parameter in the reference's layout,
so torch.optim.AdamW / clip_grad_norm_ / a gradient all-reduce work on it unchanged.  No torch math is involved:
torch allocates buffers and supplies the stream.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import native
from .native import check, dptr, lib, stream_ptr

BN_EPS, BN_MOMENTUM = 1e-5, 0.1

# (name, cout, cin, k) in state-dict order; index into the 86-tensor list = 6*i
CONVS = [("conv1", 64, 3, 7),
         ("layer0.conv1", 96, 64, 3), ("layer0.conv2", 96, 96, 3), ("layer0.ds", 96, 64, 1),
         ("layer1.conv1", 256, 96, 3), ("layer1.conv2", 256, 256, 3), ("layer1.ds", 256, 96, 1),
         ("layer2.conv1", 384, 256, 3), ("layer2.conv2", 384, 384, 3), ("layer2.ds", 384, 256, 1),
         ("layer3.conv1", 256, 384, 3), ("layer3.conv2", 256, 256, 3), ("layer3.ds", 256, 384, 1)]


class _Scratch:
    buf = None

    @classmethod
    def get(cls, dev):
        if cls.buf is None or cls.buf.device != dev:
            cls.buf = torch.empty(lib().be_train_scratch_bytes() // 4, dtype=torch.float32, device=dev)
        return cls.buf


def _new(shape, dev):
    return torch.empty(shape, dtype=torch.float32, device=dev)


class _Packs:
    """Packed weights of every unit for one training step - forward form and data-gradient form - written by ONE launch
    (be_conv_pack_jobs_f32) at the top of the forward; buffers and the device job table are built once per parameter set."""
    cache = {}

    def __init__(self, t):
        dev = t[0].device
        # (fc.4 runs from the raw parameters, forward and backward: no pack)
        units = [(6 * i, CONVS[i][1], CONVS[i][2], CONVS[i][3], 0) for i in range(13)] + [(78, 1024, 2304, 1, 9)]
        self.fwd, self.dg, self.keep = {}, {}, [v for v in t]
        jobs = []
        for wi, cout, cin, ks, chw in units:
            cin_k = 3 if ks == 7 else cin                               # conv1's weight has 3 input channels
            nfl = lib().be_conv_packed_floats(cout, cin_k, ks)
            pw, pb = _new(nfl, dev), _new((cout + 31) // 32 * 32, dev)
            self.fwd[wi] = (pw, pb)
            jobs.append(native.PackJob(dptr(t[wi]), dptr(t[wi + 1]), None, None, None, None, dptr(pw), dptr(pb), 0.0,
                                       cout, cin_k, ks, chw, 0))
            if ks != 7:                                                 # conv1 needs no input gradient
                nd = lib().be_conv_dgrad_packed_floats(cout, cin, ks)
                dw, db = _new(nd, dev), _new((cin + 31) // 32 * 32, dev)
                self.dg[wi] = (dw, db)
                jobs.append(native.PackJob(dptr(t[wi]), None, None, None, None, None, dptr(dw), dptr(db), 0.0, cout, cin, ks, chw, 1))
        arr = (native.PackJob * len(jobs))(*jobs)
        self.njobs = len(jobs)
        self.table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)

    @classmethod
    def get(cls, t):
        key = tuple(v.data_ptr() for v in t)
        p = cls.cache.get(key)
        if p is None:
            for v in t:
                if not v.is_contiguous():
                    raise RuntimeError("LocalStage training: parameters must be contiguous")
            cls.cache.clear()                                           # one model at a time keeps its buffers
            p = cls.cache[key] = cls(t)
        return p

    def pack(self):
        dev = self.table.device
        check(lib().be_conv_pack_jobs_f32(dptr(self.table, "job table", (torch.uint8,)), self.njobs, stream_ptr(dev)),
              "be_conv_pack_jobs_f32")


# ---- the single-purpose (layer-level) forms of round 2: conv, BatchNorm forward / backward, weight gradient, column sum, data
#      gradient, each its own C call.  The training step no longer uses them (it calls the units below); they stay as the
#      reference chain of tests/test_train_gpu.py::test_training_unit_matches_the_single_purpose_kernels and for GlobalStage's
#      training path (_col_sum, _wgrad).
def _conv_fwd(x, packs, wi, cout, ks):
    """plain conv / linear + bias (no BatchNorm fold) with the step's packed weights: x NHWC -> y NHWC."""
    pw, pb = packs.fwd[wi]
    return native.conv_nhwc(x, pw, pb, cout, ks, act=0, scratch=_Scratch.get(x.device))


def _bn_fwd(y, gamma, beta, rm, rv, res, act):
    m, c = y.numel() // y.shape[-1], y.shape[-1]
    dev = y.device
    mean, invstd = _new(c, dev), _new(c, dev)
    s_in = torch.empty_like(y) if act else None
    out = torch.empty_like(y)
    sc = _Scratch.get(dev)
    check(lib().be_bn_train_fwd_f32(dptr(y), dptr(gamma), dptr(beta), dptr(res), BN_EPS, BN_MOMENTUM, dptr(rm), dptr(rv),
                                    dptr(mean), dptr(invstd), dptr(s_in), dptr(out), m, c, int(act), dptr(sc),
                                    sc.numel() * 4, stream_ptr(dev)), "be_bn_train_fwd_f32")
    return out, (y, mean, invstd, s_in)


def _bn_bwd(dout, saved, gamma, dgamma=None, dbeta=None):
    y, mean, invstd, s_in = saved
    m, c = y.numel() // y.shape[-1], y.shape[-1]
    dev = y.device
    ds, dy = torch.empty_like(y), torch.empty_like(y)
    dgamma = _new(c, dev) if dgamma is None else dgamma
    dbeta = _new(c, dev) if dbeta is None else dbeta
    sc = _Scratch.get(dev)
    check(lib().be_bn_train_bwd_f32(dptr(dout), dptr(s_in), dptr(y), dptr(mean), dptr(invstd), dptr(gamma), dptr(ds),
                                    dptr(dy), dptr(dgamma), dptr(dbeta), m, c, dptr(sc), sc.numel() * 4, stream_ptr(dev)),
          "be_bn_train_bwd_f32")
    return ds, dy, dgamma, dbeta


def _col_sum(a, out=None):
    m, c = a.numel() // a.shape[-1], a.shape[-1]
    out = _new(c, a.device) if out is None else out
    sc = _Scratch.get(a.device)
    check(lib().be_col_sum_f32(dptr(a), dptr(out), m, c, dptr(sc), sc.numel() * 4, stream_ptr(a.device)), "be_col_sum_f32")
    return out


def _wgrad(x, dy, w_shape, ks, chw_hw=0, out=None):
    n, h, w, cin = x.shape
    cout = dy.shape[-1]
    dw = _new(w_shape, x.device) if out is None else out
    sc = _Scratch.get(x.device)
    check(lib().be_conv_wgrad_f32(dptr(x), dptr(dy), dptr(dw), n, h, w, cin, cout, ks, chw_hw, dptr(sc), sc.numel() * 4,
                                  stream_ptr(x.device)), "be_conv_wgrad_f32")
    return dw


def _dgrad(dy, packs, wi, cin, ks):
    """dy NHWC [.., cout] -> dx NHWC [.., cin] through the transposed / mirrored pack of the step."""
    pw, pb = packs.dg[wi]
    return native.conv_nhwc(dy, pw, pb, cin, ks, act=0, scratch=_Scratch.get(dy.device))


def _pool_bwd(x, dout, k, stride, pad):
    n, h, w, c = x.shape
    dx = torch.empty_like(x)
    check(lib().be_maxpool_nhwc_bwd_f32(dptr(x), dptr(dout), dptr(dx), n, h, w, c, k, stride, pad, stream_ptr(x.device)),
          "be_maxpool_nhwc_bwd_f32")
    return dx


def _pool_fwd_idx(x, k, stride, pad):
    """max-pool that also records the winning window element per output value (one byte): (y, (idx, input shape))"""
    n, h, w, c = x.shape
    o = native.ops()
    if o is not None:
        y, idx = o.maxpool_fwd_idx(x, k, stride, pad)
        return y, (idx, (n, h, w, c))
    oh, ow = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    y = torch.empty(n, oh, ow, c, dtype=torch.float32, device=x.device)
    idx = torch.empty(n, oh, ow, c, dtype=torch.uint8, device=x.device)
    check(lib().be_maxpool_nhwc_fwd_idx_f32(dptr(x), dptr(y), dptr(idx, "idx", (torch.uint8,)), n, h, w, c, k, stride, pad,
                                            stream_ptr(x.device)), "be_maxpool_nhwc_fwd_idx_f32")
    return y, (idx, (n, h, w, c))


def _pool_bwd_idx(saved, dout, k, stride, pad):
    idx, (n, h, w, c) = saved
    o = native.ops()
    if o is not None:
        return o.maxpool_bwd_idx(idx, dout.contiguous(), h, w, k, stride, pad)
    dx = torch.empty(n, h, w, c, dtype=torch.float32, device=dout.device)
    check(lib().be_maxpool_nhwc_bwd_idx_f32(dptr(idx, "idx", (torch.uint8,)), dptr(dout), dptr(dx), n, h, w, c, k, stride, pad,
                                            stream_ptr(dout.device)), "be_maxpool_nhwc_bwd_idx_f32")
    return dx


def _unit_fwd(xin, packs, wi, cout, ks, gamma, beta, rm, rv, res, act):
    """conv / linear + BatchNorm (batch statistics, running statistics updated in place) [+ res] [+ Smish]: three launches
    (be_train_unit_fwd_f32).  Returns (out, (y, mean, invstd, s_in)) - what the backward needs."""
    n, h, w, cin = xin.shape
    dev = xin.device
    pw, pb = packs.fwd[wi]
    o = native.ops()
    if o is not None:
        out, y, mean, invstd, s_in = o.train_unit_fwd(xin, pw, pb, gamma, beta, res, rm, rv, cout, ks, bool(act), BN_EPS, BN_MOMENTUM,
                                                      _Scratch.get(dev))
        return out, (y, mean, invstd, s_in if act else None)
    y = torch.empty(n, h, w, cout, dtype=torch.float32, device=dev)
    out = torch.empty_like(y)
    s_in = torch.empty_like(y) if act else None
    mean, invstd = _new(cout, dev), _new(cout, dev)
    sc = _Scratch.get(dev)
    d = native.ConvDesc(n, h, w, cin, cout, ks, 0)
    check(lib().be_train_unit_fwd_f32(C.byref(d), dptr(xin), dptr(pw), dptr(pb), dptr(gamma), dptr(beta), dptr(res), BN_EPS,
                                      BN_MOMENTUM, dptr(rm), dptr(rv), dptr(y), dptr(mean), dptr(invstd), dptr(s_in), dptr(out),
                                      int(act), dptr(sc), sc.numel() * 4, stream_ptr(dev)), "be_train_unit_fwd_f32")
    return out, (y, mean, invstd, s_in)


def _unit_bwd(xin, dout, saved, gamma, dg_pack, dx_add, ks, chw_hw, dgamma, dbeta, dw, db):
    """backward of _unit_fwd in four launches (be_train_unit_bwd_f32): -> (ds, dx); dx is None when dg_pack is None, and
    includes dx_add (the other branch of a residual block) when given."""
    y, mean, invstd, s_in = saved
    n, h, w, cin = xin.shape
    cout = y.shape[-1]
    dev = xin.device
    o = native.ops()
    if o is not None:
        pw_, pb_ = dg_pack if dg_pack is not None else (None, None)
        ds, dx = o.train_unit_bwd(xin, dout.contiguous(), s_in, y, mean, invstd, gamma, pw_, pb_, dx_add, ks, int(chw_hw), dgamma, dbeta,
                                  dw, db, _Scratch.get(dev))
        return ds, (dx if dg_pack is not None else None)
    ds, dy = torch.empty_like(y), torch.empty_like(y)
    dx = torch.empty(n, h, w, cin, dtype=torch.float32, device=dev) if dg_pack is not None else None
    pw, pb = dg_pack if dg_pack is not None else (None, None)
    sc = _Scratch.get(dev)
    d = native.ConvDesc(n, h, w, cin, cout, ks, 0)
    check(lib().be_train_unit_bwd_f32(C.byref(d), dptr(xin), dptr(dout), dptr(s_in), dptr(y), dptr(mean), dptr(invstd), dptr(gamma),
                                      dptr(pw), dptr(pb), dptr(dx_add), int(chw_hw), dptr(ds), dptr(dy), dptr(dgamma), dptr(dbeta),
                                      dptr(dw), dptr(db), dptr(dx), dptr(sc), sc.numel() * 4, stream_ptr(dev)),
          "be_train_unit_bwd_f32")
    return ds, dx


def _unit_pair_fwd(xin, packs, ua, ub):
    """Two units on the same input that produce the same [n,h,w,cout] - a residual block's 3x3 convolution and its 1x1 downsample -
    with every launch shared (be_train_unit_pair_fwd_f32).  ua / ub = (wi, cout, ks, gamma, beta, run_mean, run_var, act).
    Returns ((out_a, saved_a), (out_b, saved_b)) as two _unit_fwd calls would, bit for bit."""
    n, h, w, cin = xin.shape
    dev = xin.device
    o = native.ops()
    if o is not None:
        (wa, ca, ka, ga, ba, rma, rva, acta), (wb, cb, kb, gb, bb, rmb, rvb, actb) = ua, ub
        r = o.train_unit_pair_fwd(xin, *packs.fwd[wa], ga, ba, rma, rva, ca, ka, bool(acta), *packs.fwd[wb], gb, bb, rmb, rvb, cb, kb,
                                  bool(actb), BN_EPS, BN_MOMENTUM, _Scratch.get(dev))
        return ((r[0], (r[1], r[2], r[3], r[4] if acta else None)), (r[5], (r[6], r[7], r[8], r[9] if actb else None)))
    sc = _Scratch.get(dev)
    structs, res = [], []
    for wi, cout, ks, gamma, beta, rm, rv, act in (ua, ub):
        pw, pb = packs.fwd[wi]
        y = torch.empty(n, h, w, cout, dtype=torch.float32, device=dev)
        out = torch.empty_like(y)
        s_in = torch.empty_like(y) if act else None
        mean, invstd = _new(cout, dev), _new(cout, dev)
        structs.append(native.TrainUnitFwd(native.ConvDesc(n, h, w, cin, cout, ks, 0), dptr(xin), dptr(pw), dptr(pb), dptr(gamma), dptr(beta),
                                           None, dptr(rm), dptr(rv), dptr(y), dptr(mean), dptr(invstd), dptr(s_in), dptr(out), int(act)))
        res.append((out, (y, mean, invstd, s_in)))
    check(lib().be_train_unit_pair_fwd_f32(C.byref(structs[0]), C.byref(structs[1]), BN_EPS, BN_MOMENTUM, dptr(sc), sc.numel() * 4,
                                           stream_ptr(dev)), "be_train_unit_pair_fwd_f32")
    return tuple(res)


def _unit_pair_bwd(xin, ua, ub):
    """Backward of _unit_pair_fwd's two units in shared launches (be_train_unit_pair_bwd_f32).
    ua / ub = (dout, saved, gamma, dg_pack, ks, dgamma, dbeta, dw, db).  Returns (ds_a, ds_b, dx) with dx = the SUM of the two
    units' input gradients."""
    n, h, w, cin = xin.shape
    dev = xin.device
    o = native.ops()
    if o is not None:
        (da, sa, ga, pa, ka, dga, dba, dwa, dbia), (db_, sb, gb, pb_, kb, dgb, dbb, dwb, dbib) = ua, ub
        r = o.train_unit_pair_bwd(xin, da.contiguous(), sa[3], sa[0], sa[1], sa[2], ga, pa[0], pa[1], ka, dga, dba, dwa, dbia,
                                  db_.contiguous(), sb[3], sb[0], sb[1], sb[2], gb, pb_[0], pb_[1], kb, dgb, dbb, dwb, dbib, _Scratch.get(dev))
        return r[0], r[1], r[2]
    sc = _Scratch.get(dev)
    structs, dss, dxs, keep = [], [], [], []
    for dout, saved, gamma, dg_pack, ks, dgamma, dbeta, dw, db in (ua, ub):
        y, mean, invstd, s_in = saved
        cout = y.shape[-1]
        dout = dout.contiguous()
        ds, dy = torch.empty_like(y), torch.empty_like(y)
        dx = torch.empty(n, h, w, cin, dtype=torch.float32, device=dev)
        keep += [dout, dy]
        structs.append(native.TrainUnitBwd(native.ConvDesc(n, h, w, cin, cout, ks, 0), dptr(xin), dptr(dout), dptr(s_in), dptr(y), dptr(mean),
                                           dptr(invstd), dptr(gamma), dptr(dg_pack[0]), dptr(dg_pack[1]), None, 0, dptr(ds), dptr(dy),
                                           dptr(dgamma), dptr(dbeta), dptr(dw), dptr(db), dptr(dx)))
        dss.append(ds)
        dxs.append(dx)
    check(lib().be_train_unit_pair_bwd_f32(C.byref(structs[0]), C.byref(structs[1]), dptr(sc), sc.numel() * 4, stream_ptr(dev)),
          "be_train_unit_pair_bwd_f32")
    return dss[0], dss[1], dxs[0]


def forward_train(x, t):
    """x [N,3,21,21]; t = the 86 tensors (native.local_stage_pack order).  Returns (logits [N,10], saved)."""
    n = x.shape[0]
    packs = _Packs.get(t)
    packs.pack()                                          # this step's weights, forward and data-gradient forms: one launch
    S = {"packs": packs}

    def unit(name, i, xin, res=None, act=True):
        _, cout, cin, ks = CONVS[i]
        w, b, g, be_, rm, rv = t[6 * i:6 * i + 6]
        out, saved = _unit_fwd(xin, packs, 6 * i, cout, ks, g, be_, rm, rv, res, act)
        S[name] = (xin, saved)
        return out

    x4 = native.patches_to_nhwc4(x)                       # NCHW, or the dataset's channels-last batch seen through permute(0,3,1,2)
    a1 = unit("conv1", 0, x4)
    p1, S["pool1"] = _pool_fwd_idx(a1, 3, 2, 1)

    def block(tag, base, xin):
        # conv1 and the downsample read the same input and are independent of each other: one set of launches for the two
        ua, ub = [(6 * i, CONVS[i][1], CONVS[i][3], t[6 * i + 2], t[6 * i + 3], t[6 * i + 4], t[6 * i + 5], act)
                  for i, act in ((base, True), (base + 2, False))]
        (tt, saved_a), (d, saved_b) = _unit_pair_fwd(xin, packs, ua, ub)
        S[tag + ".conv1"], S[tag + ".ds"] = (xin, saved_a), (xin, saved_b)
        return unit(tag + ".conv2", base + 1, tt, res=d)

    l0 = block("layer0", 1, p1)
    p2, S["pool2"] = _pool_fwd_idx(l0, 3, 2, 1)
    l1 = block("layer1", 4, p2)
    l2 = block("layer2", 7, l1)
    l3 = block("layer3", 10, l2)
    p3, S["pool3"] = _pool_fwd_idx(l3, 2, 2, 0)
    f_in = p3.reshape(n, 1, 1, 2304)
    w1, b1, g1, be1, rm1, rv1, w4, b4 = t[78:86]
    f1, saved1 = _unit_fwd(f_in, packs, 78, 1024, 1, g1, be1, rm1, rv1, None, True)
    S["fc1"] = (f_in, saved1)
    # fc.4 (1024 -> 10) from the raw parameters: one wave per output element
    out = _new((n, 10), x.device)
    check(lib().be_linear_small_fwd_f32(dptr(f1), dptr(w4), dptr(b4), dptr(out), n, 1024, 10, stream_ptr(x.device)),
          "be_linear_small_fwd_f32")
    S["fc4"] = f1
    return out, S


# trainable entries of the 86-tensor list, in order (= LocalStage.parameters() order): conv / linear weight + bias and
# BatchNorm gamma + beta of every unit; running statistics (6i+4, 6i+5, 82, 83) have no gradient
TRAINABLE = [6 * i + j for i in range(13) for j in range(4)] + [78, 79, 80, 81, 84, 85]


_GRAD_HOOK = None
_GRAD_GROUPS = None


def set_grad_hook(fn, groups=None):
    """fn(flat, lo, hi) is called by backward_train each time a gradient BUCKET's slice flat[lo:hi] of the gradient buffer is
    final (all kernels that write it are enqueued on the current stream).  groups: the buckets as tensor-index ranges, unions of
    the backward's completion points fc, layer3, layer2, layer1, conv1 + layer0 (be_hip.dp.GRAD_POINTS; default
    be_hip.dp.DEFAULT_GROUPS).  be_hip.dp.GradSync.bucket_ready starts that bucket's all-reduce there.  None removes the hook."""
    global _GRAD_HOOK, _GRAD_GROUPS
    _GRAD_HOOK = fn
    _GRAD_GROUPS = groups if fn is not None else None


def backward_train(dlogits, t, S):
    """Returns the list of 86 gradients (None for running statistics) in the order of t.  Every gradient is a view into ONE
    flat buffer, in parameter order, written by the kernels directly: autograd then stores those views as .grad, so a
    data-parallel run all-reduces that buffer as it stands (be_hip.dp.grads_as_flat) - no per-parameter copy."""
    n = dlogits.shape[0]
    dev = dlogits.device
    grads = [None] * 86
    flat = torch.empty(sum(t[i].numel() for i in TRAINABLE), dtype=torch.float32, device=dev)
    off = 0
    for i in TRAINABLE:
        grads[i] = flat[off:off + t[i].numel()].view(t[i].shape)
        off += t[i].numel()
    hook = _GRAD_HOOK
    if hook is not None:
        from .dp import DEFAULT_GROUPS, GRAD_POINTS, bucket_ranges, check_groups
        groups = check_groups(_GRAD_GROUPS if _GRAD_GROUPS is not None else DEFAULT_GROUPS)
        ranges = bucket_ranges([v.numel() for v in t], groups)
        # completion point k closes the bucket whose first tensor is the point's first tensor
        closes = {GRAD_POINTS.index(next(p for p in GRAD_POINTS if p[0] == a)): r for (a, _), r in zip(groups, ranges)}

    def done(k):
        if hook is not None and k in closes:
            hook(flat, *closes[k])
    w1, b1, g1, be1, rm1, rv1, w4, b4 = t[78:86]
    packs = S["packs"]
    # fc.4
    f1 = S["fc4"].reshape(n, 1024)
    dx = _new((n, 1024), dev)
    check(lib().be_linear_small_bwd_f32(dptr(f1), dptr(w4.contiguous()), dptr(dlogits.contiguous()), dptr(dx), dptr(grads[84]),
                                        dptr(grads[85]), n, 1024, 10, stream_ptr(dev)), "be_linear_small_bwd_f32")
    # fc.1 + BN1d + Smish
    f_in, saved1 = S["fc1"]
    _, d = _unit_bwd(f_in, dx.reshape(n, 1, 1, 1024), saved1, g1, packs.dg[78], None, 1, 9, grads[80], grads[81], grads[78], grads[79])
    done(0)                                                            # fc.1 / fc.2 / fc.4: the tail of the buffer
    d = _pool_bwd_idx(S["pool3"], d.reshape(n, 3, 3, 256), 2, 2, 0)

    def unit_bwd(name, i, dout, need_dx=True, dx_add=None):
        """dout = gradient w.r.t. the unit's OUTPUT (after Smish if any).  Returns (ds, dx [+ dx_add])."""
        _, cout, cin, ks = CONVS[i]
        xin, saved = S[name]
        return _unit_bwd(xin, dout, saved, t[6 * i + 2], packs.dg[6 * i] if need_dx else None, dx_add, ks, 0,
                         grads[6 * i + 2], grads[6 * i + 3], grads[6 * i], grads[6 * i + 1])

    def block_bwd(tag, base, dout):
        ds_, dt = unit_bwd(tag + ".conv2", base + 1, dout)            # ds_ = dout * smish'(.) = grad of the residual too
        # conv1 (gradient dt) and the downsample (gradient ds_): shared launches, both branches' input gradients summed in the last
        xin, saved_a = S[tag + ".conv1"]
        _, saved_b = S[tag + ".ds"]
        ua, ub = [(g_, sv, t[6 * i + 2], packs.dg[6 * i], CONVS[i][3], grads[6 * i + 2], grads[6 * i + 3], grads[6 * i], grads[6 * i + 1])
                  for i, g_, sv in ((base, dt, saved_a), (base + 2, ds_, saved_b))]
        return _unit_pair_bwd(xin, ua, ub)[2]

    d = block_bwd("layer3", 10, d)
    done(1)
    d = block_bwd("layer2", 7, d)
    done(2)
    d = block_bwd("layer1", 4, d)
    done(3)
    d = _pool_bwd_idx(S["pool2"], d, 3, 2, 1)
    d = block_bwd("layer0", 1, d)
    d = _pool_bwd_idx(S["pool1"], d, 3, 2, 1)
    unit_bwd("conv1", 0, d, need_dx=False)
    done(4)                                                            # conv1 + layer0: the head of the buffer
    return grads


class LocalStageTrainFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, *tensors):
        t = [v.detach() for v in tensors]
        out, S = forward_train(x.detach().to(torch.float32), t)
        ctx.S, ctx.t = S, t
        return out

    @staticmethod
    def backward(ctx, dlogits):
        grads = backward_train(dlogits.contiguous(), ctx.t, ctx.S)
        ctx.S = None
        return (None,) + tuple(grads)
