"""Oracle: GlobalStage forward written out operation by operation, with EXPLICIT dropout masks (TEST INFRASTRUCTURE).

Restates models/global_stage.py:22-38 = Linear(38,128) + 2-D sinusoidal position table + 8 post-norm
nn.TransformerEncoderLayer (MHA 8 x 16, dropout on the attention probabilities; x = norm1(x + dropout1(sa(x)));
x = norm2(x + dropout2(linear2(dropout(relu(linear1(x))))))) + final LayerNorm + Linear(128,12).
torch's own dropout draws from Philox and cannot be reproduced by another generator, so the masks are arguments:
with all-ones masks this is the reference in eval mode / with p = 0 (pinned by golden g12, taken from the reference's
own module in train mode with p = 0); with the masks the HIP kernels report, it is the autograd reference for the
training path.  Plain torch ops in whatever dtype the inputs carry (float64 in the tests).
"""
import math

import torch
import torch.nn.functional as F


def position_table(d_model=128, max_len=64, stride=2):
    """models/global_stage.py:8-17 (computed in float32 as the reference does; cast it afterwards if needed):
    token (i, j) of the max_len x max_len grid gets sin/cos of ROW position i*stride in channels [0, d/2) and of COLUMN
    position j*stride in channels [d/2, d), even channels sin, odd channels cos."""
    half = d_model // 2
    pos = torch.linspace(0, (max_len - 1) * stride, max_len)
    div = torch.exp(torch.arange(0, half, 2) * (-2 * math.log(10000.0) / d_model))
    pe = torch.zeros(max_len, max_len, d_model)
    for n_, k in enumerate(range(0, half, 2)):
        ang = pos * div[n_]
        for i in range(max_len):
            pe[i, :, k], pe[i, :, k + 1] = torch.sin(ang[i]), torch.cos(ang[i])
            pe[:, i, half + k], pe[:, i, half + k + 1] = torch.sin(ang[i]), torch.cos(ang[i])
    return pe.reshape(max_len * max_len, d_model)


def forward(sd, src, pe, p=0.0, masks=None, nhead=8, eps=1e-5):
    """sd: state-dict (names of models/global_stage.py) as tensors that may require grad; src [B,L,38]; pe [>=L,128].
    masks: None (no dropout) or a list with one dict per layer: attn [B*H,L,L], d1 [B*L,128], ff [B*L,256],
    d2 [B*L,128], entries in {0,1}.  -> [B,L,12]"""
    B, L, _ = src.shape
    keep = 1.0 / (1.0 - p)
    h = src @ sd["in_src_projection.weight"].T + sd["in_src_projection.bias"]
    h = h + pe[:L].to(h.dtype)[None]
    D = h.shape[-1]
    dh = D // nhead
    nl = 1 + max(int(k.split(".")[2]) for k in sd if k.startswith("encoder.layers."))
    for i in range(nl):
        g = lambda n: sd[f"encoder.layers.{i}.{n}"]
        m = masks[i] if masks is not None else None
        qkv = h @ g("self_attn.in_proj_weight").T + g("self_attn.in_proj_bias")
        q, k, v = [t.view(B, L, nhead, dh).permute(0, 2, 1, 3) for t in qkv.split(D, dim=-1)]      # [B,H,L,dh]
        s = (q @ k.transpose(-1, -2)) / math.sqrt(dh)
        pr = torch.softmax(s, dim=-1)
        if m is not None:
            pr = pr * m["attn"].view(B, nhead, L, L).to(pr.dtype) * keep
        a = (pr @ v).permute(0, 2, 1, 3).reshape(B, L, D)
        sa = a @ g("self_attn.out_proj.weight").T + g("self_attn.out_proj.bias")
        if m is not None:
            sa = sa * m["d1"].view(B, L, D).to(sa.dtype) * keep
        h = F.layer_norm(h + sa, (D,), g("norm1.weight"), g("norm1.bias"), eps)
        f = torch.relu(h @ g("linear1.weight").T + g("linear1.bias"))
        if m is not None:
            f = f * m["ff"].view(B, L, -1).to(f.dtype) * keep
        y = f @ g("linear2.weight").T + g("linear2.bias")
        if m is not None:
            y = y * m["d2"].view(B, L, D).to(y.dtype) * keep
        h = F.layer_norm(h + y, (D,), g("norm2.weight"), g("norm2.bias"), eps)
    h = F.layer_norm(h, (D,), sd["encoder.norm.weight"], sd["encoder.norm.bias"], eps)
    return h @ sd["generator.weight"].T + sd["generator.bias"]
