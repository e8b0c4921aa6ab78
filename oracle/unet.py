"""Oracle: the depth-completion U-Net written out as array operations (TEST INFRASTRUCTURE).

Restates models/depth_completion_unet.py:8-113 of the reference (bilinear=False, eval mode): DoubleConv = two
(3x3 conv without bias, pad 1 -> BatchNorm with running statistics -> ReLU); Down = 2x2 max-pool (floor) + DoubleConv;
Up = ConvTranspose2d(k=2, s=2) of the lower level, zero-padded to the skip's size with the extra row/column at the
bottom/right (diff//2 first), concatenated AFTER the skip, + DoubleConv; 1x1 head.  The 3x3 convolution is nine shifted
matrix products, pooling / transposed convolution / padding are explicit index arithmetic, so comparing with the
reference's nn modules (golden g13) checks the layout conventions rather than restating the same calls.
"""
import torch


def conv3x3(x, w):
    """x [B,C,H,W], w [O,C,3,3], zero padding 1."""
    B, C, H, W = x.shape
    xp = x.new_zeros(B, C, H + 2, W + 2)
    xp[:, :, 1:-1, 1:-1] = x
    out = x.new_zeros(B, w.shape[0], H, W)
    for a in range(3):
        for b in range(3):
            out += torch.einsum("oc,bchw->bohw", w[:, :, a, b], xp[:, :, a:a + H, b:b + W])
    return out


def bn_relu(x, sd, p, eps=1e-5):
    g, b, m, v = (sd[p + k] for k in (".weight", ".bias", ".running_mean", ".running_var"))
    y = (x - m[None, :, None, None]) / torch.sqrt(v[None, :, None, None] + eps) * g[None, :, None, None] + b[None, :, None, None]
    return torch.clamp_min(y, 0)


def double_conv(x, sd, p):
    x = bn_relu(conv3x3(x, sd[p + ".0.weight"]), sd, p + ".1")
    return bn_relu(conv3x3(x, sd[p + ".3.weight"]), sd, p + ".4")


def maxpool2(x):
    H, W = x.shape[2] // 2 * 2, x.shape[3] // 2 * 2
    return torch.maximum(torch.maximum(x[:, :, 0:H:2, 0:W:2], x[:, :, 0:H:2, 1:W:2]),
                         torch.maximum(x[:, :, 1:H:2, 0:W:2], x[:, :, 1:H:2, 1:W:2]))


def up2x2(x, w, b):
    """ConvTranspose2d(k=2, s=2): w [Cin, Cout, 2, 2]; out[:, o, 2i+a, 2j+c] = sum_ci x[:, ci, i, j] w[ci, o, a, c] + b[o]."""
    B, C, H, W = x.shape
    out = x.new_zeros(B, w.shape[1], 2 * H, 2 * W)
    for a in range(2):
        for c in range(2):
            out[:, :, a::2, c::2] = torch.einsum("bchw,co->bohw", x, w[:, :, a, c]) + b[None, :, None, None]
    return out


def up_block(below, skip, sd, p):
    u = up2x2(below, sd[p + ".up.weight"], sd[p + ".up.bias"])
    H, W = skip.shape[2:]
    dy, dx = H - u.shape[2], W - u.shape[3]
    padded = u.new_zeros(u.shape[0], u.shape[1], H, W)
    padded[:, :, dy // 2:dy // 2 + u.shape[2], dx // 2:dx // 2 + u.shape[3]] = u
    return double_conv(torch.cat([skip, padded], dim=1), sd, p + ".conv.double_conv")


def forward(sd, x, want_levels=False):
    x1 = double_conv(x, sd, "inc.double_conv")
    xs = [x1]
    for k in range(1, 5):
        xs.append(double_conv(maxpool2(xs[-1]), sd, f"down{k}.maxpool_conv.1.double_conv"))
    y = xs[4]
    for k in range(1, 5):
        y = up_block(y, xs[4 - k], sd, f"up{k}")
    out = torch.einsum("oc,bchw->bohw", sd["outc.conv.weight"][:, :, 0, 0], y) + sd["outc.conv.bias"][None, :, None, None]
    return (out, xs) if want_levels else out
