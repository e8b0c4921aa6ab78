"""GlobalStage: transformer encoder over the 64x64 patch tokens of one 147x147 pair.

Boundary kept exactly (constructor signature, forward(src [B,L,38]) -> [B,L,12], 102-entry state-dict, the 2-D
sinusoidal position table held as a plain tensor that is NOT part of the state-dict): models/global_stage.py:6-38
of the reference.  Inference on the GPU runs on libblurry_edges_hip: the linears on the implicit-GEMM kernel, a
flash-style fp32-MFMA attention kernel whose scores never leave registers, fused residual + LayerNorm.
Training on the GPU (model.train()) runs the HIP training kernels as well: counter-based dropout at the four sites of
every layer (with or without grad mode, as the reference module), flash-style attention backward, LayerNorm / linear
backward (be_hip/train_global_stage.py).  A GPU tensor never reaches stock PyTorch ops: any sequence length up to the
position table (4096) runs on the kernels (padded to 128-token tiles, padded keys masked), other configurations raise.
CPU tensors run the module tree as it stands, which is the reference's own nn.TransformerEncoder (BASELINE configs[0]).
The stage is ~2.4 % of the FLOPs of the path (18.9 of 794 MFLOP per pair).
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
        if not src.is_cuda:
            # CPU tensors (BASELINE configs[0], "plumbing, no GPU"): the module tree IS the reference's nn.TransformerEncoder,
            # so this is the reference's own computation (math SDPA backend; the flash / mem-efficient ones are Triton-built)
            from torch.nn.attention import sdpa_kernel, SDPBackend
            with sdpa_kernel(SDPBackend.MATH):
                h = self.positional_encoding(self.in_src_projection(src))
                return self.generator(self.encoder(h))
        # GPU tensors: the HIP kernels or an error - never stock PyTorch ops
        self._require_hip_shapes()
        B, L, cin = src.shape
        if L > self.positional_encoding.pe.shape[1]:
            raise ValueError(f"GlobalStage: {L} tokens, the position table has {self.positional_encoding.pe.shape[1]} "
                             f"(models/global_stage.py:19 fails the same way)")
        # any L <= max_len^2 is legal for the reference; the kernels work on 128-token tiles: pad the sequence with zero
        # tokens and mask the padded KEYS inside the attention kernels (l_valid); every other op is per token
        Lp = (L + 127) // 128 * 128
        if Lp != L:
            src = torch.cat([src, src.new_zeros(B, Lp - L, cin)], dim=1)
        if not self.training:
            out = self._forward_hip(src, L)
        else:
            # model.train(): dropout at the four sites of every layer whatever the grad mode, as nn.TransformerEncoder does;
            # under autograd the HIP backward kernels (be_hip/train_global_stage.py).  The dropout seed is drawn from
            # torch's CPU generator, so torch.manual_seed makes a run repeatable
            from be_hip.train_global_stage import GlobalStageTrainFn, parameter_list
            lyr = self.encoder.layers[0]
            seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item()) ^ int(self.dropout_seed_salt)
            out = GlobalStageTrainFn.apply(src, self.positional_encoding.pe[0], seed, lyr.dropout.p, lyr.self_attn.num_heads,
                                           lyr.norm1.eps, L, *parameter_list(self))
        return out if Lp == L else out[:, :L]

    # xor-ed into the per-step dropout seed: a data-parallel run gives every rank its own salt (be_hip.workflow.global_train),
    # otherwise identically seeded replicas would all drop the same elements of their share of the global batch
    dropout_seed_salt = 0

    def _require_hip_shapes(self):
        lyr = self.encoder.layers[0]
        d = self.in_src_projection.out_features
        ok = (d == 128 and d // lyr.self_attn.num_heads == 16 and lyr.linear1.out_features % 32 == 0
              and lyr.self_attn.dropout == lyr.dropout.p == lyr.dropout1.p == lyr.dropout2.p)
        if not ok:
            raise NotImplementedError("GlobalStage on the GPU is built for the reference configuration: d_model 128, heads of 16, "
                                      "dim_feedforward a multiple of 32, one dropout probability (models/global_stage.py:23-32)")

    # ------------------------------------------------------------------ inference on the HIP library
    def invalidate_packed(self):
        """Drop the cached weight pack: for a caller that changes the parameters where `Tensor._version` does not see it (a
        replayed hipGraph that contains the optimizer step; cf. LocalStage.invalidate_packed)."""
        self._pk_key = None

    def train(self, mode: bool = True):
        self._pk_key = None                      # a train -> eval switch never meets a pack made before the training
        return super().train(mode)

    def _packed(self):
        """Linear weights in the implicit-GEMM layout, re-packed when a parameter changes."""
        from be_hip import native
        ps = list(self.parameters())
        key = tuple((p.data_ptr(), p._version) for p in ps)
        if getattr(self, "_pk_key", None) == key:
            return self._pk
        dev = ps[0].device

        def pack(w, b, pad_in=0):
            w = w.detach()
            if pad_in:                                       # in-projection: 38 input features padded to 64
                w = torch.cat([w, w.new_zeros(w.shape[0], pad_in)], dim=1)
            return native.conv_pack(w.contiguous(), b.detach().contiguous())
        cin = self.in_src_projection.in_features
        self._pad_in = (-cin) % 32
        pk = dict(inp=pack(self.in_src_projection.weight, self.in_src_projection.bias, self._pad_in),
                  gen=pack(self.generator.weight, self.generator.bias), layers=[])
        for lyr in self.encoder.layers:
            a = lyr.self_attn
            pk["layers"].append(dict(qkv=pack(a.in_proj_weight, a.in_proj_bias), out=pack(a.out_proj.weight, a.out_proj.bias),
                                     l1=pack(lyr.linear1.weight, lyr.linear1.bias), l2=pack(lyr.linear2.weight, lyr.linear2.bias)))
        self._pk, self._pk_key = pk, key
        return pk

    def _forward_hip(self, src, l_valid):
        from be_hip import native
        B, L, cin = src.shape
        pk = self._packed()
        d = self.in_src_projection.out_features
        H = self.encoder.layers[0].self_attn.num_heads
        x = src.reshape(B * L, cin).to(torch.float32)
        if self._pad_in:
            x = torch.cat([x, x.new_zeros(B * L, self._pad_in)], dim=1)
        h = native.linear(x.contiguous(), *pk["inp"], d)
        native.add_pe_(h, self.positional_encoding.pe[0, :L].contiguous(), B)
        ws = getattr(self, "_attn_ws", None)
        for lyr, p in zip(self.encoder.layers, pk["layers"]):
            qkv = native.linear(h, *p["qkv"], 3 * d)
            a, ws = native.attention(qkv, B, L, H, ws, l_valid=l_valid)
            y = native.linear(a, *p["out"], d, residual=h)                       # x + SA(x)
            h = native.add_layernorm(y, None, lyr.norm1.weight.detach(), lyr.norm1.bias.detach(), lyr.norm1.eps)
            f = native.linear(h, *p["l1"], lyr.linear1.out_features, act=2)      # ReLU
            y = native.linear(f, *p["l2"], d, residual=h)                        # x + FFN(x)
            h = native.add_layernorm(y, None, lyr.norm2.weight.detach(), lyr.norm2.bias.detach(), lyr.norm2.eps)
        self._attn_ws = ws
        n = self.encoder.norm
        h = native.add_layernorm(h, None, n.weight.detach(), n.bias.detach(), n.eps)
        out = native.linear(h, *pk["gen"], self.generator.out_features)
        return out.view(B, L, -1)
