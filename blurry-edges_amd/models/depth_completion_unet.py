"""DepthCompletion (`--densify pp`): the 5-level U-Net that fills the holes of the folded depth map
(models/depth_completion_unet.py:79-113 of the reference; used at blurry_edges_test.py:141-142, 193-196).

Boundary kept: `models.DepthCompletion(n_channels=1, n_classes=1, bilinear=False)`, `forward(x [B,1,H,W]) -> [B,1,H,W]`,
and the reference's state-dict keys (`inc.double_conv.0.weight`, `down1.maxpool_conv.1.double_conv.*`, `up1.up.weight`,
`up1.conv.double_conv.*`, `outc.conv.*`), so a reference checkpoint loads with strict=True.

Inference on the GPU runs on libblurry_edges_hip: every 3x3 convolution with its BatchNorm folded and
ReLU fused on the implicit-GEMM MFMA kernel, NHWC throughout; the 2x2 stride-2 transposed convolution as a 1x1
convolution with 4*cout outputs followed by a pixel-shuffle scatter; skip tensors are written by the encoder straight
into the first half of the decoder's concatenated input, so torch.cat / F.pad never run.  CPU tensors, training mode
and the `bilinear=True` variant use the stock torch layers of this module tree.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


def _conv_bn_relu_x2(cin, cout, cmid=None):
    cmid = cmid or cout
    return nn.Sequential(nn.Conv2d(cin, cmid, 3, padding=1, bias=False), nn.BatchNorm2d(cmid), nn.ReLU(inplace=True),
                         nn.Conv2d(cmid, cout, 3, padding=1, bias=False), nn.BatchNorm2d(cout), nn.ReLU(inplace=True))


class DoubleConv(nn.Module):
    def __init__(self, in_channels, out_channels, mid_channels=None):
        super().__init__()
        self.double_conv = _conv_bn_relu_x2(in_channels, out_channels, mid_channels)

    def forward(self, x):
        return self.double_conv(x)


class Down(nn.Module):
    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.maxpool_conv = nn.Sequential(nn.MaxPool2d(2), DoubleConv(in_channels, out_channels))

    def forward(self, x):
        return self.maxpool_conv(x)


class Up(nn.Module):
    def __init__(self, in_channels, out_channels, bilinear=True):
        super().__init__()
        if bilinear:
            self.up = nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True)
            self.conv = DoubleConv(in_channels, out_channels, in_channels // 2)
        else:
            self.up = nn.ConvTranspose2d(in_channels, in_channels // 2, kernel_size=2, stride=2)
            self.conv = DoubleConv(in_channels, out_channels)

    def forward(self, below, skip):
        below = self.up(below)
        dy, dx = skip.size(2) - below.size(2), skip.size(3) - below.size(3)
        below = F.pad(below, [dx // 2, dx - dx // 2, dy // 2, dy - dy // 2])
        return self.conv(torch.cat([skip, below], dim=1))


class OutConv(nn.Module):
    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size=1)

    def forward(self, x):
        return self.conv(x)


class UNet(nn.Module):
    def __init__(self, n_channels=1, n_classes=1, bilinear=False):
        super().__init__()
        self.n_channels, self.n_classes, self.bilinear = n_channels, n_classes, bilinear
        f = 2 if bilinear else 1
        self.inc = DoubleConv(n_channels, 64)
        self.down1, self.down2, self.down3 = Down(64, 128), Down(128, 256), Down(256, 512)
        self.down4 = Down(512, 1024 // f)
        self.up1, self.up2 = Up(1024, 512 // f, bilinear), Up(512, 256 // f, bilinear)
        self.up3, self.up4 = Up(256, 128 // f, bilinear), Up(128, 64, bilinear)
        self.outc = OutConv(64, n_classes)
        self._pk = self._pk_key = None

    def forward(self, x):
        if x.is_cuda and not self.training and not self.bilinear:
            return self._forward_hip(x)
        if x.is_cuda and not self.training:
            raise NotImplementedError("DepthCompletion(bilinear=True) has no HIP path (the reference constructs the default, "
                                      "bilinear=False, blurry_edges_test.py:194)")
        x1 = self.inc(x)
        x2 = self.down1(x1)
        x3 = self.down2(x2)
        x4 = self.down3(x3)
        x5 = self.down4(x4)
        y = self.up1(x5, x4)
        y = self.up2(y, x3)
        y = self.up3(y, x2)
        y = self.up4(y, x1)
        return self.outc(y)

    # ------------------------------------------------------------------ inference on the HIP library
    def _packed(self):
        from be_hip import native
        ts = [t for t in self.state_dict().values() if t.is_floating_point()]
        key = tuple((t.data_ptr(), t._version) for t in ts)
        if self._pk_key == key:
            return self._pk

        def dc(m):                      # DoubleConv -> two (packed weight, packed bias, cout)
            out = []
            for ci, bi in ((0, 1), (3, 4)):
                conv, bn = m.double_conv[ci], m.double_conv[bi]
                w = conv.weight.detach()
                pad = (-w.shape[1]) % 32
                if pad:
                    w = torch.cat([w, w.new_zeros(w.shape[0], pad, 3, 3)], dim=1)
                pw, pb = native.conv_pack(w.contiguous(), None, bn=(bn.weight.detach(), bn.bias.detach(), bn.running_mean,
                                                                    bn.running_var), eps=bn.eps)
                out.append((pw, pb, w.shape[0]))
            return out

        def up(m):                      # ConvTranspose2d weight [cin, cout, 2, 2] -> 1x1 conv weight [(dy,dx,co), cin]
            w = m.up.weight.detach()
            cin, cout = w.shape[0], w.shape[1]
            w4 = w.permute(2, 3, 1, 0).reshape(4 * cout, cin).contiguous()
            pw, pb = native.conv_pack(w4, m.up.bias.detach().repeat(4).contiguous())
            return pw, pb, cout

        pk = dict(inc=dc(self.inc), down=[dc(d.maxpool_conv[1]) for d in (self.down1, self.down2, self.down3, self.down4)],
                  up=[(up(u), dc(u.conv)) for u in (self.up1, self.up2, self.up3, self.up4)],
                  out=native.conv_pack(self.outc.conv.weight.detach().reshape(self.n_classes, -1).contiguous(),
                                       self.outc.conv.bias.detach().contiguous()))
        self._pk, self._pk_key = pk, key
        return pk

    @torch.no_grad()
    def _forward_hip(self, x):
        from be_hip import native
        pk = self._packed()
        B, _, H, W = x.shape
        dev = x.device
        relu = 2
        # the deep levels are tiny GEMM-M (81 .. 5329 rows) with long K (up to 9216): lend scratch so their K loops split
        if getattr(self, "_scratch", None) is None or self._scratch.device != dev:
            self._scratch = torch.empty(16 << 20, dtype=torch.float32, device=dev)
        sk = self._scratch
        x0 = native.nchw_to_nhwc_pad(x.to(torch.float32).contiguous(), (self.n_channels + 31) // 32 * 32)

        def double(xin, packs, out=None):
            (w1, b1, c1), (w2, b2, c2) = packs
            t = native.conv_nhwc(xin, w1, b1, c1, 3, relu, scratch=sk)
            return native.conv_nhwc(t, w2, b2, c2, 3, relu, out=out, scratch=sk)

        # encoder: each level's output lands in channels [0, C) of the [B,h,w,2C] buffer its decoder level will read
        cats, cur, chans = [], x0, [64, 128, 256, 512]
        for lvl, c in enumerate(chans):
            h, w = (H >> lvl), (W >> lvl)
            cat = torch.zeros(B, h, w, 2 * c, dtype=torch.float32, device=dev)         # zero border = F.pad of the up branch
            double(cur, pk["inc"] if lvl == 0 else pk["down"][lvl - 1], out=cat)
            cats.append(cat)
            cur = native.maxpool_nhwc(cat, 2, 2, 0, channels=c)
        y = double(cur, pk["down"][3])                                                  # [B, H/16, W/16, 1024]
        for lvl in (3, 2, 1, 0):
            (uw, ub, uc), conv = pk["up"][3 - lvl]
            n_, h_, w_, cin = y.shape
            t = native.linear(y.view(-1, cin), uw, ub, 4 * uc).view(n_, h_, w_, 4 * uc)
            native.upconv2x2_scatter(t, cats[lvl], uc, uc)
            y = double(cats[lvl], conv)
        ow, ob = pk["out"]
        out = native.linear(y.view(-1, y.shape[-1]), ow, ob, self.n_classes)            # [B*H*W, n_classes]
        return out.view(B, H, W, self.n_classes).permute(0, 3, 1, 2).contiguous()
