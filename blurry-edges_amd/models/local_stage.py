"""LocalStage: the per-patch CNN (3x21x21 -> 10 wedge parameters) behind the reference's own interface.

Same constructor signature, module tree and state-dict layout (100 entries, incl. int64 num_batches_tracked)
as the reference's models/local_stage.py:30-62, so checkpoints written by either load into the other with
strict=True.  What differs is what runs: forward() hands the raw parameter tensors to libblurry_edges_hip
(BatchNorm folded, fp32-MFMA implicit-GEMM convs with fused Smish / residual epilogues, NHWC activations)
through the C ABI in include/blurry_edges_hip.h.  A GPU tensor runs on the HIP kernels or raises - there is no eager fallback;
a CPU tensor (BASELINE configs[0]: "PyTorch-CPU, plumbing, no GPU") runs the module tree itself, as GlobalStage and
DepthCompletion do: that tree is the reference's network, layer for layer.
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn

from be_hip import native

# (attribute, out_channels) of the four residual stages, models/local_stage.py:38-41
_STAGES = (("layer0", 96), ("layer1", 256), ("layer2", 384), ("layer3", 256))


class Smish(nn.Module):
    """x * tanh(log(1 + sigmoid(x))) (models/local_stage.py:4-6).  Inside LocalStage this is fused into the
    conv epilogues; the module exists for the state-dict / module-tree contract."""

    def forward(self, x):
        return x * torch.tanh(torch.log(1 + torch.sigmoid(x)))


def _conv_bn(cin, cout, k, pad, act):
    mods = [nn.Conv2d(cin, cout, kernel_size=k, stride=1, padding=pad), nn.BatchNorm2d(cout)]
    if act:
        mods.append(Smish())
    return nn.Sequential(*mods)


class ResidualBlock(nn.Module):
    """Module-tree twin of models/local_stage.py:8-19 (keys conv1.{0,1}, conv2.{0,1}, downsample.{0,1})."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, padding=1, downsample=None):
        super().__init__()
        if stride != 1:
            raise NotImplementedError("the HIP conv kernels are stride-1 (every block of LocalStage is)")
        self.conv1 = _conv_bn(in_channels, out_channels, kernel_size, padding, act=True)
        self.conv2 = _conv_bn(out_channels, out_channels, 3, 1, act=False)
        self.downsample = downsample
        self.activation = Smish()

    def forward(self, x):
        """CPU tensors only (BASELINE configs[0]): the module tree as it stands, i.e. the reference's own computation
        (models/local_stage.py:20-28).  On the GPU a block never runs alone: LocalStage.forward hands the whole stack to the HIP
        kernels, and a GPU tensor arriving here is refused rather than served by stock PyTorch ops."""
        if x.is_cuda:
            raise RuntimeError("ResidualBlock runs only as part of LocalStage.forward on the GPU (fused HIP kernels)")
        shortcut = x if self.downsample is None else self.downsample(x)
        return self.activation(self.conv2(self.conv1(x)) + shortcut)


class LocalStage(nn.Module):
    def __init__(self, block=ResidualBlock, layers=[1, 1, 1, 1], output_dim=10):
        super().__init__()
        if list(layers) != [1, 1, 1, 1] or output_dim != 10 or block is not ResidualBlock:
            raise NotImplementedError("the HIP LocalStage is built for the reference configuration "
                                      "(ResidualBlock, layers=[1,1,1,1], output_dim=10)")
        self.inplanes = 64
        self.conv1 = _conv_bn(3, 64, 7, 3, act=True)
        for name, planes in _STAGES:
            setattr(self, name, self._make_layer(block, planes, 1))
        self.maxpool1 = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.maxpool2 = nn.MaxPool2d(kernel_size=2, stride=2)
        self.fc = nn.Sequential(nn.Flatten(), nn.Linear(3 * 3 * 256, 1024), nn.BatchNorm1d(1024), Smish(),
                                nn.Linear(1024, output_dim))
        self._packed = None
        self._packed_key = None
        self._workspace = None

    def _make_layer(self, block, planes, blocks, kernel_size=3, stride=1, padding=1):
        down = None
        if stride != 1 or self.inplanes != planes:
            down = _conv_bn(self.inplanes, planes, 1, 0, act=False)
        stage = [block(self.inplanes, planes, kernel_size, stride, padding, down)]
        self.inplanes = planes
        stage += [block(planes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*stage)

    # ------------------------------------------------------------------ native weight hand-off
    def _tensor_list(self):
        """The 86 fp32 tensors in the order be_local_stage_pack_f32 documents."""
        out = []

        def pair(seq):
            conv, bn = seq[0], seq[1]
            out.extend([conv.weight, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var])

        pair(self.conv1)
        for name, _ in _STAGES:
            blk = getattr(self, name)[0]
            pair(blk.conv1); pair(blk.conv2); pair(blk.downsample)
        fc1, bn1, fc4 = self.fc[1], self.fc[2], self.fc[4]
        out.extend([fc1.weight, fc1.bias, bn1.weight, bn1.bias, bn1.running_mean, bn1.running_var,
                    fc4.weight, fc4.bias])
        return out

    # the arithmetic of every convolution: exact fp32 MFMA products, fp32 accumulation (bench.py reports it as `dtype`)
    conv_precision = "f32"
    # True (default): the 3x3 convolutions on the 6x6 maps run as Winograd F(6,3) x F(3,3) on 8x5 tiles (80 multiplies per map and
    # channel pair instead of 324, exact fp32 products; be_wino.hip, be_wino_math.h).  False: direct implicit-GEMM convolutions everywhere.  A per-call option of the C ABI
    # (be_local_stage_opts): instances with different settings coexist in one process.
    winograd = os.environ.get("BE_WINOGRAD", "1") != "0"
    # sub-batch (patches) the forward walks a large batch in; 0 = the library default (8192)
    chunk = 0
    # 2 (default): an eval batch of 8192 patches and more (or an image pair with 4096+ patch positions) runs as two halves
    # on two side streams - patches are independent, the halves' kernels fill each other's tails and let HBM-bound transform
    # kernels overlap matrix-bound ones: 14.0 -> 13.5 ms per 8192 patches, results bit-identical (profiles/HISTORY.md,
    # DESIGN 3.1f).  1: one stream (what the per-kernel roofline measurements use: a kernel's duration means something only
    # when nothing else shares the chip).
    streams = int(os.environ.get("BE_LOCAL_STREAMS", "2"))
    _side = None

    def invalidate_packed(self):
        """Drop the cached BN-folded weight pack.  Needed whenever parameters or running statistics change on the device
        without Python seeing it: a replayed hipGraph of the training step (be_hip.train_local.GraphedStep) updates them
        without bumping any tensor's `_version`, which is what the cache key is made of."""
        self._packed_key = None

    def train(self, mode: bool = True):
        # every train() / eval() switch re-packs on the next eval forward (one ~55 us launch): the pack folds the running
        # statistics, and those move under it during training whether or not autograd's version counters notice
        self._packed_key = None
        return super().train(mode)

    def _packed_weights(self):
        tensors = [t.detach() for t in self._tensor_list()]
        key = tuple((t.data_ptr(), t._version) for t in tensors)
        if self._packed is None or key != self._packed_key:
            self._packed = native.local_stage_pack(tensors, eps=self.conv1[1].eps)
            self._packed_key = key
        return self._packed

    def _forward_module_tree(self, x):
        """CPU tensors: BASELINE configs[0] ("PyTorch-CPU, plumbing, no GPU") - the module tree this class owns for the
        state-dict contract IS the reference's network, so running it is the reference's computation (models/local_stage.py:63-73):
        stock PyTorch in eval and train mode, autograd included.  Not a fallback of the GPU path: a GPU tensor never gets here."""
        x = self.maxpool1(self.conv1(x))
        x = self.maxpool1(self.layer0(x))
        x = self.layer3(self.layer2(self.layer1(x)))
        return self.fc(self.maxpool2(x))

    def forward(self, x):
        if not x.is_cuda:
            return self._forward_module_tree(x)
        if self.training:
            # batch-statistics BatchNorm + full backward on the HIP training kernels (be_hip/train.py);
            # running statistics and num_batches_tracked are updated in place, as nn.BatchNorm does.
            from be_hip.train import LocalStageTrainFn
            out = LocalStageTrainFn.apply(x, *self._tensor_list())
            self._packed_key = None          # running statistics changed under the packed (BN-folded) weights
            # one multi-tensor launch for the 14 counters (a `+= 1` each was 14 launches of a 270-launch step)
            torch._foreach_add_([m.num_batches_tracked for m in self.modules()
                                 if isinstance(m, (nn.BatchNorm2d, nn.BatchNorm1d))], 1)
            return out
        x = x.to(torch.float32).contiguous()
        packed = self._packed_weights()
        n = x.shape[0]
        h = (n // 2) // 128 * 128
        if self.streams >= 2 and h >= 4096 and x.is_cuda and not torch.cuda.is_current_stream_capturing():
            out = torch.empty(n, 10, dtype=torch.float32, device=x.device)
            ws = [self._workspace, getattr(self, "_workspace2", None)]

            def half(i, lo, hi):
                _, ws[i] = native.local_stage_forward(packed, x[lo:hi], out=out[lo:hi], workspace=ws[i], winograd=self.winograd,
                                                      chunk=self.chunk)
            self._on_two_streams(x.device, lambda: half(0, 0, h), lambda: half(1, h, n))
            self._workspace, self._workspace2 = ws
            return out
        out, self._workspace = native.local_stage_forward(packed, x, workspace=self._workspace,
                                                          winograd=self.winograd, chunk=self.chunk)
        return out

    def _on_two_streams(self, device, f0, f1):
        """f0 / f1 enqueue independent work; each runs on its own side stream, forked from and joined into the current one."""
        cur = torch.cuda.current_stream(device)
        if self._side is None or self._side[0].device != cur.device:
            self._side = (torch.cuda.Stream(device=device), torch.cuda.Stream(device=device))
        for s_, f in zip(self._side, (f0, f1)):
            s_.wait_stream(cur)
            with torch.cuda.stream(s_):
                f()
        for s_ in self._side:
            cur.wait_stream(s_)

    @torch.no_grad()
    def forward_image_pair(self, img, stride=2, window=None):
        """Eval forward over every 21x21 window of an image pair [2,3,H,W] (or of its block window = (top, left,
        height, width)) -> [2*P,10], in the order of nn.Unfold + permute + flatten(0,1)
        (blurry_edges_test.py:120-123) without materialising the windows."""
        if self.training:
            raise RuntimeError("LocalStage.forward_image_pair is an inference entry point; call .eval() first")
        img = img.to(torch.float32).contiguous()
        view = native.view_image_pair(img, stride, window)
        P = (((window[2] if window is not None else img.shape[2]) - native.BE_R) // stride + 1) * view.wp
        packed = self._packed_weights()
        if self.streams >= 2 and P >= 4096 and not torch.cuda.is_current_stream_capturing():
            # one aperture per side stream: the second image's windows are the same view moved by one image
            out = torch.empty(2 * P, 10, dtype=torch.float32, device=img.device)
            v2 = native.PatchView(view.base + 4 * view.s_aperture, view.s_aperture, view.s_chan, view.s_row, view.s_col, view.s_pi,
                                  view.s_pj, view.wp)
            ws = [self._workspace, getattr(self, "_workspace2", None)]

            def half(i, v):
                _, ws[i] = native.local_stage_forward_view(packed, v, P, P, img.device, out=out[i * P:(i + 1) * P], workspace=ws[i],
                                                           winograd=self.winograd, chunk=self.chunk)
            self._on_two_streams(img.device, lambda: half(0, view), lambda: half(1, v2))
            self._workspace, self._workspace2 = ws
            return out
        out, self._workspace = native.local_stage_forward_view(packed, view, P, 2 * P, img.device,
                                                               workspace=self._workspace, winograd=self.winograd, chunk=self.chunk)
        return out
