"""GlobalStage: transformer encoder over the 64x64 patch tokens of one 147x147 pair.

Boundary kept exactly (constructor signature, forward(src [B,L,38]) -> [B,L,12], 102-entry state-dict, the 2-D
sinusoidal position table held as a plain tensor that is NOT part of the state-dict): models/global_stage.py:6-38
of the reference.  Per SURVEY.md §2 row 4 / §8f-1 this stage runs on stock PyTorch-ROCm ops in this round
(the math SDPA backend: the flash / mem-efficient backends on ROCm are Triton-built); a hand-written HIP
attention is the next row.  It is ~2.4 % of the FLOPs of the path (18.9 of 794 MFLOP per pair).
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


def position_table(d_model, max_len, stride):
    """[max_len*max_len, d_model]: first half of the channels encodes the patch ROW position (0, stride, ...),
    second half the COLUMN position; even channels sin, odd cos (models/global_stage.py:8-17)."""
    half = d_model // 2
    pos = torch.linspace(0, (max_len - 1) * stride, max_len)
    freq = torch.exp(torch.arange(0, half, 2) * (-2 * math.log(10000.0) / d_model))          # [half/2]
    ang = pos[:, None] * freq[None, :]                                                       # [max_len, half/2]
    axis = torch.stack([torch.sin(ang), torch.cos(ang)], dim=-1).reshape(max_len, half)     # interleaved sin/cos
    table = torch.cat([axis[:, None, :].expand(max_len, max_len, half),
                       axis[None, :, :].expand(max_len, max_len, half)], dim=-1)
    return table.reshape(max_len * max_len, d_model).contiguous()


class PositionalEncoding(nn.Module):
    def __init__(self, d_model, max_len, stride, device=None):
        super().__init__()
        self.pe = position_table(d_model, max_len, stride).unsqueeze(0).to(device)   # plain attribute, as in the reference

    def forward(self, x):
        x += self.pe[:, :x.size(1), :]        # in place, as the reference does (:19)
        return x


class GlobalStage(nn.Module):
    def __init__(self, max_len=64, stride=2, in_parameter_size=38, out_parameter_size=12, d_model=128, nhead=8,
                 num_encoder_layers=8, dim_feedforward=256, layer_norm_eps=1e-5, batch_first=True, bias=True, device=None):
        super().__init__()
        self.in_src_projection = nn.Linear(in_parameter_size, d_model)
        self.positional_encoding = PositionalEncoding(d_model, max_len, stride, device=device)
        layer = nn.TransformerEncoderLayer(d_model, nhead, dim_feedforward, dropout=0.1, activation=F.relu,
                                           layer_norm_eps=layer_norm_eps, batch_first=batch_first, norm_first=False,
                                           bias=bias, device=device)
        self.encoder = nn.TransformerEncoder(layer, num_encoder_layers,
                                             nn.LayerNorm(d_model, eps=layer_norm_eps, bias=bias, device=device))
        self.generator = nn.Linear(d_model, out_parameter_size)

    def forward(self, src):
        from torch.nn.attention import sdpa_kernel, SDPBackend
        with sdpa_kernel(SDPBackend.MATH):                     # no Triton-built attention kernels
            h = self.positional_encoding(self.in_src_projection(src))
            return self.generator(self.encoder(h))
