"""Oracle: LocalStage CNN as a pure function of a state-dict (TEST INFRASTRUCTURE).

Restates models/local_stage.py:4-73 of the reference with torch.nn.functional ops on CPU.
The state-dict keys are the reference's own (SURVEY.md §8b, 100 entries).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5       # nn.BatchNorm2d default used at models/local_stage.py:13,16,36,48,56
BN_MOMENTUM = 0.1


def to_torch_sd(sd_np, dtype=torch.float32):
    out = {}
    for k, v in sd_np.items():
        t = torch.from_numpy(np.asarray(v))
        out[k] = t.to(dtype) if t.is_floating_point() else t
    return out


def smish(x):
    # models/local_stage.py:4-6 : x * tanh(log(1 + sigmoid(x)))   (log(1+s), not log1p)
    return x * torch.tanh(torch.log(1 + torch.sigmoid(x)))


_RUNNING_OUT = None    # dict filled with the updated running statistics of a training forward (see local_stage_forward)


def _bn(y, sd, p, training):
    rm, rv = sd[p + ".running_mean"], sd[p + ".running_var"]
    if training:      # batch statistics; running buffers are updated on clones so the oracle stays pure
        rm, rv = rm.detach().clone(), rv.detach().clone()
        if _RUNNING_OUT is not None:
            _RUNNING_OUT[p + ".running_mean"], _RUNNING_OUT[p + ".running_var"] = rm, rv
    return F.batch_norm(y, rm, rv, sd[p + ".weight"], sd[p + ".bias"], training, BN_MOMENTUM, BN_EPS)


def conv_bn(x, sd, p, pad, training=False):
    # nn.Sequential(Conv2d, BatchNorm2d[, Smish]) : models/local_stage.py:11-17,34-37,54-56
    return _bn(F.conv2d(x, sd[p + ".0.weight"], sd[p + ".0.bias"], padding=pad), sd, p + ".1", training)


def residual_block(x, sd, p, training=False):
    # models/local_stage.py:20-28 ; every block here has a 1x1 downsample (:53-56)
    out = smish(conv_bn(x, sd, p + ".conv1", 1, training))
    out = conv_bn(out, sd, p + ".conv2", 1, training)
    res = conv_bn(x, sd, p + ".downsample", 0, training)
    return smish(out + res)


def local_stage_forward(sd, x, training=False, taps=None, running_out=None):
    """x [N,3,21,21] -> [N,10]  (models/local_stage.py:63-73).
    taps: optional dict filled with intermediate activations (NCHW) for layer-by-layer parity.
    running_out: optional dict; a training forward stores the running statistics nn.BatchNorm would have written
    (momentum 0.1, unbiased variance) under their state-dict keys - sd itself is never modified."""
    global _RUNNING_OUT
    _RUNNING_OUT = running_out if training else None
    try:
        return _forward(sd, x, training, taps)
    finally:
        _RUNNING_OUT = None


def _forward(sd, x, training, taps):
    def tap(name, t):
        if taps is not None:
            taps[name] = t
        return t
    h = tap("conv1", smish(conv_bn(x, sd, "conv1", 3, training)))                 # [N,64,21,21]
    h = tap("pool1", F.max_pool2d(h, 3, 2, 1))                                    # [N,64,11,11]
    h = tap("layer0", residual_block(h, sd, "layer0.0", training))                # [N,96,11,11]
    h = tap("pool2", F.max_pool2d(h, 3, 2, 1))                                    # [N,96,6,6]
    h = tap("layer1", residual_block(h, sd, "layer1.0", training))                # [N,256,6,6]
    h = tap("layer2", residual_block(h, sd, "layer2.0", training))                # [N,384,6,6]
    h = tap("layer3", residual_block(h, sd, "layer3.0", training))                # [N,256,6,6]
    h = tap("pool3", F.max_pool2d(h, 2, 2))                                       # [N,256,3,3]
    h = h.flatten(1)                                                              # (C,H,W) order
    h = F.linear(h, sd["fc.1.weight"], sd["fc.1.bias"])
    h = tap("fc1", smish(_bn(h, sd, "fc.2", training)))
    return F.linear(h, sd["fc.4.weight"], sd["fc.4.bias"])


# FLOP model used by bench.py's roofline: 2*MAC of convs + linears per patch.
def flops_per_patch():
    conv = [(64, 3, 7, 441), (96, 64, 3, 121), (96, 96, 3, 121), (96, 64, 1, 121),
            (256, 96, 3, 36), (256, 256, 3, 36), (256, 96, 1, 36),
            (384, 256, 3, 36), (384, 384, 3, 36), (384, 256, 1, 36),
            (256, 384, 3, 36), (256, 256, 3, 36), (256, 384, 1, 36)]
    f = sum(2 * co * ci * k * k * hw for co, ci, k, hw in conv)
    return f + 2 * 2304 * 1024 + 2 * 1024 * 10
