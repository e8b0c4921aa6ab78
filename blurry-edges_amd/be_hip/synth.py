"""Portable deterministic generators for weights and synthetic patch pairs.

Everything here is a pure function of (seed, tensor name, element index) built on a
counter-based 64-bit integer hash (splitmix64), evaluated with numpy in float64 and
rounded once to float32.  No torch RNG, no numpy RNG state: the container that makes
the golden vectors and the GPU box regenerate bit-identical tensors from the seed alone
(SURVEY.md §7 step 1, §8d "synthetic patch-pair generator").

Shapes / parameter ranges follow the reference's own conventions:
  * LocalStage state-dict layout ........ models/local_stage.py:30-50 (100 entries)
  * camera constants .................... utils/args.py:14-15
  * thin-lens blur radius per aperture .. utils/data_generator.py:16-17
  * photon / read noise model ........... train_val_data_generator.py:173-181
"""
from __future__ import annotations

import math
import zlib
from collections import OrderedDict

import numpy as np
from scipy.special import erf as _erf

SEED_DEFAULT = 1869  # echoes local_training.py:73

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)
_GAMMA = np.uint64(0x9E3779B97F4A7C15)
_C1 = np.uint64(0xBF58476D1CE4E5B9)
_C2 = np.uint64(0x94D049BB133111EB)


def _mix(x: np.ndarray) -> np.ndarray:
    """splitmix64 finaliser on a uint64 array (wrap-around arithmetic)."""
    with np.errstate(over="ignore"):
        z = x + _GAMMA
        z = (z ^ (z >> np.uint64(30))) * _C1
        z = (z ^ (z >> np.uint64(27))) * _C2
        return z ^ (z >> np.uint64(31))


def _stream_key(seed: int, name: str, lane: int = 0) -> np.uint64:
    tag = zlib.crc32(name.encode("utf-8")) & 0xFFFFFFFF
    k = (int(seed) & 0xFFFFFFFF) | (tag << 32)
    k ^= (lane * 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF
    return _mix(np.array([k], dtype=np.uint64))[0]


def hash_uniform(seed: int, name: str, shape, lane: int = 0) -> np.ndarray:
    """float64 uniforms in [0,1), one per element, addressable by (seed, name, index)."""
    n = int(np.prod(shape)) if len(shape) else 1
    idx = np.arange(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        h = _mix(_mix(idx + _stream_key(seed, name, lane)))
    u = (h >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
    return u.reshape(shape)


def hash_normal(seed: int, name: str, shape) -> np.ndarray:
    """float64 standard normals (Box-Muller on two hashed uniform streams)."""
    u1 = hash_uniform(seed, name, shape, lane=1)
    u2 = hash_uniform(seed, name, shape, lane=2)
    r = np.sqrt(-2.0 * np.log(1.0 - u1))  # 1-u1 in (0,1]
    return r * np.cos(2.0 * math.pi * u2)


def f32(a: np.ndarray) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


# ---------------------------------------------------------------------------------------
# LocalStage weights (state-dict layout of models/local_stage.py:30-62)
# ---------------------------------------------------------------------------------------

# (prefix, Cout, Cin, k) of every conv+BN pair, in state-dict order.
LOCAL_STAGE_CONVS = [
    ("conv1", 64, 3, 7),
    ("layer0.0.conv1", 96, 64, 3), ("layer0.0.conv2", 96, 96, 3), ("layer0.0.downsample", 96, 64, 1),
    ("layer1.0.conv1", 256, 96, 3), ("layer1.0.conv2", 256, 256, 3), ("layer1.0.downsample", 256, 96, 1),
    ("layer2.0.conv1", 384, 256, 3), ("layer2.0.conv2", 384, 384, 3), ("layer2.0.downsample", 384, 256, 1),
    ("layer3.0.conv1", 256, 384, 3), ("layer3.0.conv2", 256, 256, 3), ("layer3.0.downsample", 256, 384, 1),
]


def _bn_entries(sd, seed, prefix, c):
    sd[f"{prefix}.weight"] = f32(0.5 + hash_uniform(seed, f"{prefix}.weight", (c,)))
    sd[f"{prefix}.bias"] = f32(-0.1 + 0.2 * hash_uniform(seed, f"{prefix}.bias", (c,)))
    sd[f"{prefix}.running_mean"] = f32(0.1 * hash_normal(seed, f"{prefix}.running_mean", (c,)))
    sd[f"{prefix}.running_var"] = f32(0.5 + hash_uniform(seed, f"{prefix}.running_var", (c,)))
    sd[f"{prefix}.num_batches_tracked"] = np.array(0, dtype=np.int64)


def local_stage_state_dict(seed: int = SEED_DEFAULT, gain: float = 1.0) -> "OrderedDict[str, np.ndarray]":
    """Random-but-reproducible LocalStage weights as numpy arrays keyed exactly like the
    reference's ``state_dict()`` (100 entries, models/local_stage.py:30-50).

    conv / linear weights ~ N(0, gain^2 * 2/(fan_in+fan_out))  (xavier-normal, as the caller in
    local_training.py:83-85 re-initialises them); biases ~ U[-0.05, 0.05];
    BN gamma ~ U[0.5,1.5], beta ~ U[-0.1,0.1], running_mean ~ N(0, 0.1^2), running_var ~ U[0.5,1.5].
    """
    sd: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for prefix, co, ci, k in LOCAL_STAGE_CONVS:
        fan_in, fan_out = ci * k * k, co * k * k
        std = gain * math.sqrt(2.0 / (fan_in + fan_out))
        sd[f"{prefix}.0.weight"] = f32(std * hash_normal(seed, f"{prefix}.0.weight", (co, ci, k, k)))
        sd[f"{prefix}.0.bias"] = f32(-0.05 + 0.1 * hash_uniform(seed, f"{prefix}.0.bias", (co,)))
        _bn_entries(sd, seed, f"{prefix}.1", co)
    std = gain * math.sqrt(2.0 / (2304 + 1024))
    sd["fc.1.weight"] = f32(std * hash_normal(seed, "fc.1.weight", (1024, 2304)))
    sd["fc.1.bias"] = f32(-0.05 + 0.1 * hash_uniform(seed, "fc.1.bias", (1024,)))
    _bn_entries(sd, seed, "fc.2", 1024)
    std = gain * math.sqrt(2.0 / (1024 + 10))
    sd["fc.4.weight"] = f32(std * hash_normal(seed, "fc.4.weight", (10, 1024)))
    sd["fc.4.bias"] = f32(-0.05 + 0.1 * hash_uniform(seed, "fc.4.bias", (10,)))
    return sd


# ---------------------------------------------------------------------------------------
# Inputs
# ---------------------------------------------------------------------------------------

def uniform_patches(n: int, seed: int = SEED_DEFAULT, name: str = "patches") -> np.ndarray:
    """[n,3,21,21] float32 in [0,1): pure-parity input."""
    return f32(hash_uniform(seed, name, (n, 3, 21, 21)))


def plausible_params10(n: int, seed: int = SEED_DEFAULT, name: str = "params10") -> np.ndarray:
    """[n,10] wedge parameters in the range the CNN is trained to emit:
    vertices in [-1.5,1.5]^2, angles in [0,2pi), eta coefficients ~ N(0.5, 0.5^2)."""
    p = np.empty((n, 10), dtype=np.float64)
    p[:, 0:4] = -1.5 + 3.0 * hash_uniform(seed, name + ".xy", (n, 4))
    p[:, 4:8] = 2.0 * math.pi * hash_uniform(seed, name + ".ang", (n, 4))
    p[:, 8:10] = 0.5 + 0.5 * hash_normal(seed, name + ".eta", (n, 2))
    return f32(p)


def glue_params10(p: int, seed: int = SEED_DEFAULT, name: str = "g15_params10") -> np.ndarray:
    """[2p,10] raw LocalStage outputs as the CNN emits them BEFORE the eval glue wraps them: plausible_params10 with
    every angle moved by a whole number of turns in [-3, 3] (blurry_edges_test.py:124 `remainder(., 2 pi)` must undo it).
    Rows 0..p-1 aperture 1, p..2p-1 aperture 2 (golden g15)."""
    q = plausible_params10(2 * p, seed, name).astype(np.float64)
    turns = np.floor(7.0 * hash_uniform(seed, name + ".turns", (2 * p, 4))) - 3.0
    q[:, 4:8] += 2.0 * math.pi * turns
    return f32(q)


def big_block_params10(k: int, seed: int = SEED_DEFAULT) -> np.ndarray:
    """[8192,10] raw LocalStage output for block k of the 36-block big-image golden (g17): rows 0..4095 aperture 1."""
    return plausible_params10(8192, seed, f"g17_p10_{k}")


def big_block_global_out(k: int, seed: int = SEED_DEFAULT) -> np.ndarray:
    """[4096,12] GlobalStage output (normalised, as the network emits it) for block k of golden g17: the inverse of the eval glue's
    de-normalisation (blurry_edges_test_big.py:161-165) applied to plausible wedge parameters."""
    p = plausible_params12(4096, seed, f"g17_p12_{k}").astype(np.float64)
    y = np.empty_like(p)
    y[:, :4] = p[:, :4] / 3.0
    y[:, 4:8] = p[:, 4:8] / math.pi - 1.0
    y[:, 8:] = p[:, 8:] - 0.5
    return f32(y)


def plausible_params12(n: int, seed: int = SEED_DEFAULT, name: str = "params12") -> np.ndarray:
    """[n,12]: 8 shared geometry + eta coefficients (w1,img1),(w2,img1),(w1,img2),(w2,img2)
    (layout of blurry_edges_test.py:36-37,44-45)."""
    p = np.empty((n, 12), dtype=np.float64)
    p[:, 0:4] = -1.5 + 3.0 * hash_uniform(seed, name + ".xy", (n, 4))
    p[:, 4:8] = 2.0 * math.pi * hash_uniform(seed, name + ".ang", (n, 4))
    p[:, 8:12] = 0.5 + 0.5 * hash_normal(seed, name + ".eta", (n, 4))
    return f32(p)


# camera constants, utils/args.py:14-15
CAM = dict(s=0.1104, rho_1=10.0, rho_2=10.2, sigma_cam=0.003, pixel_pitch=5.86e-6)
MAG = 4.0
Z_RANGE = (0.75, 1.18)   # utils/args.py:23
ALPHA_RANGE = (180.0, 200.0)
READ_SIGMA = 2.0


def blur_sigma_px(z: np.ndarray) -> np.ndarray:
    """Per-aperture Gaussian blur radius in pixels for depth z [..] -> [..,2]
    (thin-lens model of utils/data_generator.py:16-17)."""
    rhos = np.array([CAM["rho_1"], CAM["rho_2"]])
    return np.abs((1.0 / z[..., None] - rhos) * CAM["s"] + 1.0) * CAM["sigma_cam"] / CAM["pixel_pitch"] / MAG


def synthetic_patch_pairs(p: int, seed: int = SEED_DEFAULT, noise: bool = True):
    """P two-aperture 21x21 patch pairs of a blurred straight edge / corner scene.

    Returns (x [2P,3,21,21] float32, image-major as blurry_edges_test.py:121 orders them:
    rows 0..P-1 are aperture 1, rows P..2P-1 aperture 2; z [P] float32 ground-truth depth).

    Scene per pair: one or two half-planes ("wedges" with a 180 deg opening) through the patch with
    flat colours, at a single depth z ~ U[0.75,1.18] m; each aperture sees it through a Gaussian PSF
    of radius blur_sigma_px(z) (order-2 PSF of utils/data_generator.py:19-23), i.e. an erf edge;
    photon scaling alpha ~ U[180,200]; noise = Poisson(img*alpha) approximated by a hashed normal with
    matching variance + 2*N(0,1) read noise, clipped to [0,alpha], rounded, / alpha
    (train_val_data_generator.py:173-181).
    """
    name = "pairs"
    z = Z_RANGE[0] + (Z_RANGE[1] - Z_RANGE[0]) * hash_uniform(seed, name + ".z", (p,))
    sig = blur_sigma_px(z)                                   # [P,2] pixels
    ang = 2.0 * math.pi * hash_uniform(seed, name + ".ang", (p, 2))
    off = -6.0 + 12.0 * hash_uniform(seed, name + ".off", (p, 2))   # pixels from centre
    two = hash_uniform(seed, name + ".two", (p,)) < 0.5
    col = hash_uniform(seed, name + ".col", (p, 3, 3))          # [P, wedge(bg,w1,w2), rgb]
    alpha = ALPHA_RANGE[0] + (ALPHA_RANGE[1] - ALPHA_RANGE[0]) * hash_uniform(seed, name + ".alpha", (p,))

    yy, xx = np.meshgrid(np.arange(21) - 10.0, np.arange(21) - 10.0, indexing="ij")
    out = np.empty((2, p, 3, 21, 21), dtype=np.float64)
    erf = _erf
    for a in range(2):
        s = sig[:, a][:, None, None] * math.sqrt(2.0)
        d1 = (-np.sin(ang[:, 0])[:, None, None] * xx + np.cos(ang[:, 0])[:, None, None] * yy) - off[:, 0][:, None, None]
        d2 = (-np.sin(ang[:, 1])[:, None, None] * xx + np.cos(ang[:, 1])[:, None, None] * yy) - off[:, 1][:, None, None]
        h1 = 0.5 * (1.0 + erf(d1 / s))
        h2 = np.where(two[:, None, None], 0.5 * (1.0 + erf(d2 / s)), 0.0)
        u0 = (1 - h1) * (1 - h2)
        u1 = h1 * (1 - h2)
        u2 = h2
        img = (u0[:, None] * col[:, 0, :, None, None] + u1[:, None] * col[:, 1, :, None, None]
               + u2[:, None] * col[:, 2, :, None, None])
        if noise:
            lam = img * alpha[:, None, None, None]
            g1 = hash_normal(seed, f"{name}.shot{a}", img.shape)
            g2 = hash_normal(seed, f"{name}.read{a}", img.shape)
            cnt = np.rint(np.clip(lam + np.sqrt(np.maximum(lam, 0.0)) * g1 + READ_SIGMA * g2,
                                  0.0, alpha[:, None, None, None]))
            img = cnt / alpha[:, None, None, None]
        out[a] = img
    return f32(out.reshape(2 * p, 3, 21, 21)), f32(z)


def synthetic_image_pair(h: int = 147, w: int = 147, seed: int = SEED_DEFAULT, nshape: int = 6):
    """One two-aperture image pair [2,3,h,w] float32 (+ depth map [h,w]) made of nshape random
    half-plane / disc layers at random depths, back to front, each blurred per aperture by its own
    erf edge (the per-layer compositing of train_val_data_generator.py:31-116 restated analytically)."""
    name = f"img{h}x{w}"
    yy, xx = np.meshgrid(np.arange(h, dtype=np.float64), np.arange(w, dtype=np.float64), indexing="ij")
    zs = np.sort(Z_RANGE[0] + (Z_RANGE[1] - Z_RANGE[0]) * hash_uniform(seed, name + ".z", (nshape + 1,)))[::-1]
    col = hash_uniform(seed, name + ".col", (nshape + 1, 3))
    cx = hash_uniform(seed, name + ".cx", (nshape,)) * w
    cy = hash_uniform(seed, name + ".cy", (nshape,)) * h
    rad = (0.08 + 0.22 * hash_uniform(seed, name + ".r", (nshape,))) * min(h, w)
    kind = hash_uniform(seed, name + ".k", (nshape,)) < 0.5
    ang = 2.0 * math.pi * hash_uniform(seed, name + ".a", (nshape,))
    erf = _erf
    imgs = np.empty((2, 3, h, w), dtype=np.float64)
    depth = np.full((h, w), zs[0], dtype=np.float64)
    for a in range(2):
        img = np.broadcast_to(col[0][:, None, None], (3, h, w)).copy()
        for i in range(nshape):
            z = zs[i + 1]
            s = blur_sigma_px(np.array(z))[a] * math.sqrt(2.0)
            if kind[i]:
                d = rad[i] - np.sqrt((xx - cx[i]) ** 2 + (yy - cy[i]) ** 2)
            else:
                d = -math.sin(ang[i]) * (xx - cx[i]) + math.cos(ang[i]) * (yy - cy[i])
            m = 0.5 * (1.0 + erf(d / s))
            img = img * (1 - m) + m * col[i + 1][:, None, None]
            if a == 0:
                depth = np.where(d > 0, z, depth)
        imgs[a] = img
    alpha = 190.0
    g = hash_normal(seed, name + ".noise", imgs.shape)
    lam = imgs * alpha
    imgs = np.rint(np.clip(lam + np.sqrt(np.maximum(lam, 0)) * g, 0, alpha)) / alpha
    return f32(imgs), f32(depth)


# ---------------------------------------------------------------------------------------
# GlobalStage weights (state-dict layout of models/global_stage.py:23-33; SURVEY §8b, 102 entries)
# ---------------------------------------------------------------------------------------

def global_stage_state_dict(seed: int = SEED_DEFAULT, d_model=128, nlayers=8, d_ff=256, d_in=38, d_out=12):
    sd: "OrderedDict[str, np.ndarray]" = OrderedDict()

    def lin(name, o, i, gain=1.0):
        std = gain * math.sqrt(2.0 / (i + o))
        sd[name + ".weight" if not name.endswith("_weight") else name] = f32(std * hash_normal(seed, "g." + name + ".w", (o, i)))

    def vec(name, n, lo, hi):
        sd[name] = f32(lo + (hi - lo) * hash_uniform(seed, "g." + name, (n,)))

    lin("in_src_projection", d_model, d_in); vec("in_src_projection.bias", d_model, -0.05, 0.05)
    for l in range(nlayers):
        p = f"encoder.layers.{l}."
        sd[p + "self_attn.in_proj_weight"] = f32(math.sqrt(2.0 / (4 * d_model)) * hash_normal(seed, "g." + p + "inw", (3 * d_model, d_model)))
        vec(p + "self_attn.in_proj_bias", 3 * d_model, -0.05, 0.05)
        lin(p + "self_attn.out_proj", d_model, d_model); vec(p + "self_attn.out_proj.bias", d_model, -0.05, 0.05)
        lin(p + "linear1", d_ff, d_model); vec(p + "linear1.bias", d_ff, -0.05, 0.05)
        lin(p + "linear2", d_model, d_ff); vec(p + "linear2.bias", d_model, -0.05, 0.05)
        vec(p + "norm1.weight", d_model, 0.8, 1.2); vec(p + "norm1.bias", d_model, -0.1, 0.1)
        vec(p + "norm2.weight", d_model, 0.8, 1.2); vec(p + "norm2.bias", d_model, -0.1, 0.1)
    vec("encoder.norm.weight", d_model, 0.8, 1.2); vec("encoder.norm.bias", d_model, -0.1, 0.1)
    lin("generator", d_out, d_model, gain=0.5); vec("generator.bias", d_out, -0.05, 0.05)
    return sd


def unet_state_dict(seed: int = SEED_DEFAULT, n_channels=1, n_classes=1):
    """DepthCompletion U-Net (bilinear=False) in the reference's state-dict layout
    (models/depth_completion_unet.py:79-97): He-scaled conv weights, non-trivial BatchNorm statistics."""
    sd: "OrderedDict[str, np.ndarray]" = OrderedDict()

    def double(prefix, cin, cout):
        for ci, bi, a, b in ((0, 1, cin, cout), (3, 4, cout, cout)):
            sd[f"{prefix}.{ci}.weight"] = f32(math.sqrt(2.0 / (9 * a)) * hash_normal(seed, f"u.{prefix}.{ci}", (b, a, 3, 3)))
            sd[f"{prefix}.{bi}.weight"] = f32(0.8 + 0.4 * hash_uniform(seed, f"u.{prefix}.{bi}.g", (b,)))
            sd[f"{prefix}.{bi}.bias"] = f32(-0.1 + 0.2 * hash_uniform(seed, f"u.{prefix}.{bi}.b", (b,)))
            sd[f"{prefix}.{bi}.running_mean"] = f32(0.1 * hash_normal(seed, f"u.{prefix}.{bi}.m", (b,)))
            sd[f"{prefix}.{bi}.running_var"] = f32(0.5 + hash_uniform(seed, f"u.{prefix}.{bi}.v", (b,)))
            sd[f"{prefix}.{bi}.num_batches_tracked"] = np.array(0, dtype=np.int64)

    double("inc.double_conv", n_channels, 64)
    for k, (a, b) in enumerate(((64, 128), (128, 256), (256, 512), (512, 1024)), start=1):
        double(f"down{k}.maxpool_conv.1.double_conv", a, b)
    for k, (a, b) in enumerate(((1024, 512), (512, 256), (256, 128), (128, 64)), start=1):
        sd[f"up{k}.up.weight"] = f32(math.sqrt(1.0 / a) * hash_normal(seed, f"u.up{k}.w", (a, a // 2, 2, 2)))
        sd[f"up{k}.up.bias"] = f32(-0.05 + 0.1 * hash_uniform(seed, f"u.up{k}.b", (a // 2,)))
        double(f"up{k}.conv.double_conv", a, b)
    sd["outc.conv.weight"] = f32(math.sqrt(1.0 / 64) * hash_normal(seed, "u.outc.w", (n_classes, 64, 1, 1)))
    sd["outc.conv.bias"] = f32(-0.05 + 0.1 * hash_uniform(seed, "u.outc.b", (n_classes,)))
    return sd


def sparse_depth_map(h: int = 147, w: int = 147, seed: int = SEED_DEFAULT, name: str = "sparse_depth") -> np.ndarray:
    """[1,1,h,w]: a piecewise-constant depth scene in Z_RANGE with ~60 % of the pixels zeroed, the kind of map
    local2global_depth + the confidence threshold hand to DepthCompletion (blurry_edges_test.py:140-142)."""
    yy, xx = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
    u = hash_uniform(seed, name + ".p", (6, 4))
    z = np.full((h, w), Z_RANGE[1], dtype=np.float64)
    for k in range(6):
        cy, cx, r = u[k, 0] * h, u[k, 1] * w, (0.1 + 0.25 * u[k, 2]) * min(h, w)
        z[(yy - cy) ** 2 + (xx - cx) ** 2 < r * r] = Z_RANGE[0] + (Z_RANGE[1] - Z_RANGE[0]) * u[k, 3]
    keep = hash_uniform(seed, name + ".keep", (h, w)) < 0.4
    return f32(z * keep)[None, None]


def global_features(p: int = 4096, seed: int = SEED_DEFAULT, name: str = "pm") -> np.ndarray:
    """[1,P,38] normalised features in the range blurry_edges_test.py:129-132 produces ([-1,1]-ish)."""
    return f32(-1.0 + 2.0 * hash_uniform(seed, name, (1, p, 38)))


# ---------------------------------------------------------------------------------------
# Synthetic local-training set (the tensors data/dataset.py:40-47 hands to local_training.py:102-105)
# ---------------------------------------------------------------------------------------

def synthetic_training_patches(n: int, seed: int = SEED_DEFAULT):
    """n single-aperture 21x21 training patches of blurred edge / corner scenes, already divided by alpha:
    dict(img_ny [n,21,21,3], img_gt [n,21,21,3] (noise-free), bndry_dist [n,21,21] (pixels to the nearest true
    boundary), deri [n,19,19,3] (Sobel magnitude of the clean image, border cropped)) -- float32, channels-last,
    the layouts of train_val_data_generator.py:267-275 / data/dataset.py:11-19."""
    name = "trainset"
    z = Z_RANGE[0] + (Z_RANGE[1] - Z_RANGE[0]) * hash_uniform(seed, name + ".z", (n,))
    sig = blur_sigma_px(z)[:, 0]
    ang = 2.0 * math.pi * hash_uniform(seed, name + ".ang", (n, 2))
    off = -6.0 + 12.0 * hash_uniform(seed, name + ".off", (n, 2))
    two = hash_uniform(seed, name + ".two", (n,)) < 0.5
    col = hash_uniform(seed, name + ".col", (n, 3, 3))
    alpha = ALPHA_RANGE[0] + (ALPHA_RANGE[1] - ALPHA_RANGE[0]) * hash_uniform(seed, name + ".alpha", (n,))
    yy, xx = np.meshgrid(np.arange(21) - 10.0, np.arange(21) - 10.0, indexing="ij")
    s = sig[:, None, None] * math.sqrt(2.0)
    d1 = (-np.sin(ang[:, 0])[:, None, None] * xx + np.cos(ang[:, 0])[:, None, None] * yy) - off[:, 0][:, None, None]
    d2 = (-np.sin(ang[:, 1])[:, None, None] * xx + np.cos(ang[:, 1])[:, None, None] * yy) - off[:, 1][:, None, None]
    h1 = 0.5 * (1.0 + _erf(d1 / s))
    h2 = np.where(two[:, None, None], 0.5 * (1.0 + _erf(d2 / s)), 0.0)
    u = np.stack([(1 - h1) * (1 - h2), h1 * (1 - h2), h2], axis=-1)            # [n,21,21,3 wedges]
    gt = np.einsum("nhwk,nkc->nhwc", u, col)                                   # [n,21,21,3 rgb]
    lam = gt * alpha[:, None, None, None]
    g1 = hash_normal(seed, name + ".shot", gt.shape)
    g2 = hash_normal(seed, name + ".read", gt.shape)
    ny = np.rint(np.clip(lam + np.sqrt(np.maximum(lam, 0.0)) * g1 + READ_SIGMA * g2, 0.0, alpha[:, None, None, None]))
    ny = ny / alpha[:, None, None, None]
    # boundary distance: |d1| where wedge 2 does not cover, |d2| on its edge; nearest of the visible edges
    bd = np.where(two[:, None, None], np.minimum(np.where(d2 < 0, np.abs(d1), 1e3), np.abs(d2)), np.abs(d1))
    kx = np.array([[-1, 0, 1], [-2, 0, 2], [-1, 0, 1]], dtype=np.float64)
    ky = np.array([[1, 2, 1], [0, 0, 0], [-1, -2, -1]], dtype=np.float64)
    gx = sum(kx[a, b] * gt[:, a:a + 19, b:b + 19, :] for a in range(3) for b in range(3))
    gy = sum(ky[a, b] * gt[:, a:a + 19, b:b + 19, :] for a in range(3) for b in range(3))
    deri = np.sqrt(gx ** 2 + gy ** 2)
    return dict(img_ny=f32(ny), img_gt=f32(gt), bndry_dist=f32(bd), deri=f32(deri))


# ---------------------------------------------------------------------------------------
# Synthetic global-stage sample (the tensors data/dataset.py:48-55 hands to global_training.py:204-209)
# ---------------------------------------------------------------------------------------

def synthetic_global_sample(h: int = 147, w: int = 147, seed: int = SEED_DEFAULT, nshape: int = 6):
    """One training sample of the global stage, dataset layouts (channels-last, already / alpha):
    img_ny [2,h,w,3], img_gt [2,h,w,3] (noise-free), bndry_dist [h,w] (pixels to the nearest occlusion boundary),
    deri [2,h-2,w-2,3] (Sobel magnitude of the clean images), bndry_depth [h,w] (depth of the front surface at
    boundary pixels, 0 elsewhere), depth [h,w].  Scene = the layered discs / half-planes of synthetic_image_pair."""
    from scipy.ndimage import distance_transform_edt
    name = f"glob{h}x{w}"
    yy, xx = np.meshgrid(np.arange(h, dtype=np.float64), np.arange(w, dtype=np.float64), indexing="ij")
    zs = np.sort(Z_RANGE[0] + (Z_RANGE[1] - Z_RANGE[0]) * hash_uniform(seed, name + ".z", (nshape + 1,)))[::-1]
    col = hash_uniform(seed, name + ".col", (nshape + 1, 3))
    cx = hash_uniform(seed, name + ".cx", (nshape,)) * w
    cy = hash_uniform(seed, name + ".cy", (nshape,)) * h
    rad = (0.08 + 0.22 * hash_uniform(seed, name + ".r", (nshape,))) * min(h, w)
    kind = hash_uniform(seed, name + ".k", (nshape,)) < 0.5
    ang = 2.0 * math.pi * hash_uniform(seed, name + ".a", (nshape,))
    clean = np.empty((2, 3, h, w), dtype=np.float64)
    depth = np.full((h, w), zs[0], dtype=np.float64)
    layer = np.zeros((h, w), dtype=np.int64)
    for a in range(2):
        img = np.broadcast_to(col[0][:, None, None], (3, h, w)).copy()
        for i in range(nshape):
            z = zs[i + 1]
            s = blur_sigma_px(np.array(z))[a] * math.sqrt(2.0)
            if kind[i]:
                d = rad[i] - np.sqrt((xx - cx[i]) ** 2 + (yy - cy[i]) ** 2)
            else:
                d = -math.sin(ang[i]) * (xx - cx[i]) + math.cos(ang[i]) * (yy - cy[i])
            m = 0.5 * (1.0 + _erf(d / s))
            img = img * (1 - m) + m * col[i + 1][:, None, None]
            if a == 0:
                depth = np.where(d > 0, z, depth)
                layer = np.where(d > 0, i + 1, layer)
        clean[a] = img
    alpha = 190.0
    g = hash_normal(seed, name + ".noise", clean.shape)
    lam = clean * alpha
    noisy = np.rint(np.clip(lam + np.sqrt(np.maximum(lam, 0)) * g, 0, alpha)) / alpha
    edge = np.zeros((h, w), dtype=bool)
    edge[:, 1:] |= layer[:, 1:] != layer[:, :-1]
    edge[1:, :] |= layer[1:, :] != layer[:-1, :]
    bdist = distance_transform_edt(~edge) if edge.any() else np.full((h, w), float(max(h, w)))
    front = depth.copy()                                       # nearer surface of the two sides of a boundary
    front[:, 1:] = np.minimum(front[:, 1:], depth[:, :-1])
    front[1:, :] = np.minimum(front[1:, :], depth[:-1, :])
    bdepth = np.where(edge, front, 0.0)
    kx = np.array([[-1, 0, 1], [-2, 0, 2], [-1, 0, 1]], dtype=np.float64)
    ky = np.array([[1, 2, 1], [0, 0, 0], [-1, -2, -1]], dtype=np.float64)
    gx = sum(kx[a_, b_] * clean[:, :, a_:a_ + h - 2, b_:b_ + w - 2] for a_ in range(3) for b_ in range(3))
    gy = sum(ky[a_, b_] * clean[:, :, a_:a_ + h - 2, b_:b_ + w - 2] for a_ in range(3) for b_ in range(3))
    deri = np.sqrt(gx ** 2 + gy ** 2)
    return dict(img_ny=f32(noisy.transpose(0, 2, 3, 1)), img_gt=f32(clean.transpose(0, 2, 3, 1)), bndry_dist=f32(bdist),
                deri=f32(deri.transpose(0, 2, 3, 1)), bndry_depth=f32(bdepth), depth=f32(depth))


def plausible_global_output(p: int = 4096, seed: int = SEED_DEFAULT, name: str = "gout") -> np.ndarray:
    """[P,12] raw GlobalStage outputs whose de-normalisation (global_training.py:141-145) gives plausible wedges."""
    q = plausible_params12(p, seed, name).astype(np.float64)
    out = np.empty_like(q)
    out[:, :4] = q[:, :4] / 3.0
    out[:, 4:8] = q[:, 4:8] / math.pi - 1.0
    out[:, 8:] = q[:, 8:] - 0.5
    return f32(out)
