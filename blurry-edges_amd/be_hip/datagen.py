"""Synthetic-shape training data on the GPU: the counterpart of train_val_data_generator.py:31-275.

Per image: 15..25 flat-coloured circles / rotated rectangles / triangles at depths in Z_range, painted far to near;
each is blurred per aperture with the thin-lens PSF of its depth and alpha-composited; the all-in-focus image, boundary
locations, depth maps, city-block boundary distances and Sobel maps come with it; then photon + read noise, then
21x21 patches cropped near boundaries.  Arrays keep the reference's layouts, dtypes (float64) and file names, so
data.ShapeDataset reads what `save` writes.

Split of the work:
  host  : the few dozen scene parameters per image, drawn from the portable counter-based generator (be_hip.synth) in
          the order the reference draws them from np.random, and turned into integer vertices (cv2.boxPoints arithmetic +
          truncation) -- `draw_scenes`.
  GPU   : everything per pixel (be_datagen.hip through the C ABI): rasterisation, dilations, separable PSF blur +
          compositing in float64, distance transforms, Sobel, noise, patch cropping.
Rasterisation rule: the reference draws with cv2.circle / cv2.drawContours (thickness -1 and 1, default LINE_8, shift 0;
train_val_data_generator.py:58-76).  cv2 is not available offline, so `be_datagen_raster_u32` follows the ALGORITHMS those calls
run in OpenCV's modules/imgproc/src/drawing.cpp (the 2.4 - 4.5.1 form of the polygon fill rule; 4.5.2+ is recalled to round both run
ends - unverifiable offline, see oracle/datagen.py:cv_poly_masks; the masks are pinned to that restatement, not to cv2): Circle() - the midpoint walk over one octant, filled rows or the eight
symmetric points; Line() - clipLine() to the image, then the 8-connected LineIterator started from the left end point;
CollectPolyEdges() + FillEdgeCollection() - every edge drawn with Line(), the non-horizontal ones kept in 16.16 fixed point,
active for y0 <= y < y1, runs from ceil(x_left) to floor(x_right).  One workgroup per object leaves a FILL and a RING bit plane;
oracle/datagen.py holds the same rules in numpy and known answers worked by hand pin both (tests/test_host_cpu.py).  Everything
downstream of the masks follows the reference exactly and is pinned against it (golden g14).
"""
from __future__ import annotations

import ctypes as C
import math
import os

import numpy as np
import torch

from . import native, synth
from .native import check, lib, stream_ptr


def dptr(t, name="tensor"):
    return native.dptr(t, name, dtypes=(torch.float64, torch.int32, torch.int64, torch.uint8))

MAXO = 32                      # objects per image the kernels accept (num_shape < 26 in utils/args.py)
CIRCLE, RECT, TRI = 0, 1, 2


def box_points(cx, cy, w, h, angle_deg):
    """Corners of a rotated rectangle in cv2.boxPoints order (float64)."""
    a = math.radians(angle_deg)
    b, a_ = math.cos(a) * 0.5, math.sin(a) * 0.5
    p0 = (cx - a_ * h - b * w, cy + b * h - a_ * w)
    p1 = (cx + a_ * h - b * w, cy - b * h - a_ * w)
    return [p0, p1, (2 * cx - p0[0], 2 * cy - p0[1]), (2 * cx - p1[0], 2 * cy - p1[1])]


def draw_scenes(n, seed=synth.SEED_DEFAULT, img_size=(147, 147), num_shape=(15, 26), z_range=(0.75, 1.18), name="scenes"):
    """Scene parameters of n images.  Returns a dict of numpy arrays:
    nobj [n] int32, bg [n,3] float64, shape [n,MAXO,10] int32 = (type, nv, x0,y0,..,x3,y3) (circle: x0,y0 = centre,
    x1 = radius), prop [n,MAXO,4] float64 = (z, c0, c1, c2), objects sorted far -> near, plus `raw`: per image the
    list of values in the order train_val_data_generator.py:33-75 consumes np.random (for replaying the reference)."""
    H, W = img_size
    u = lambda tag, shape: synth.hash_uniform(seed, f"{name}.{tag}", shape)
    nobj = (num_shape[0] + np.floor(u("n", (n,)) * (num_shape[1] - num_shape[0]))).astype(np.int32)
    if nobj.max() > MAXO:
        raise ValueError(f"draw_scenes: at most {MAXO} objects per image")
    bg = np.floor(u("bg", (n, 3)) * 255)
    kind = np.floor(u("kind", (n, MAXO)) * 3).astype(np.int32)
    col = np.floor(u("col", (n, MAXO, 3)) * 255)
    zs = z_range[0] + (z_range[1] - z_range[0]) * u("z", (n, MAXO))
    ctr = u("ctr", (n, MAXO, 2)) * np.array([W, H], dtype=np.float64)
    par = u("par", (n, MAXO, 4))
    max_size = max(H, W) * 0.8
    shape = np.zeros((n, MAXO, 10), dtype=np.int32)
    prop = np.zeros((n, MAXO, 4), dtype=np.float64)
    raw = []
    for i in range(n):
        k = int(nobj[i])
        z = -np.sort(-zs[i, :k])                                           # far first (ascending sort, then reversed, :44-45)
        prop[i, :k, 0], prop[i, :k, 1:] = z, col[i, :k]
        r = dict(bg=bg[i].astype(np.int64), kind_col=np.concatenate([kind[i, :k, None], col[i, :k]], axis=1).astype(np.int64),
                 z=zs[i, :k, None].copy(), ctr=ctr[i, :k].copy(), per_obj=[])
        for o in range(k):
            cx, cy = ctr[i, o]
            if kind[i, o] == CIRCLE:
                rad = int(math.floor(par[i, o, 0] * int(max_size / 2)))
                shape[i, o, :5] = (CIRCLE, 0, int(cx), int(cy), rad)
                r["per_obj"].append(rad)
            elif kind[i, o] == RECT:
                sa = par[i, o, :3] * np.array([max_size, max_size, 180.0])
                pts = box_points(cx, cy, sa[0], sa[1], sa[2])
                shape[i, o, :2] = (RECT, 4)
                shape[i, o, 2:10] = np.array(pts, dtype=np.float64).astype(np.int64).reshape(-1)
                r["per_obj"].append(sa)
            else:
                sa = par[i, o, :4] * np.array([max_size, 2 * np.pi, 2 * np.pi, 2 * np.pi])
                vx, vy = cx + sa[0] * np.cos(sa[1:]), cy + sa[0] * np.sin(sa[1:])
                shape[i, o, :2] = (TRI, 3)
                shape[i, o, 2:8] = np.stack([vx, vy], axis=1).astype(np.int64).reshape(-1)
                r["per_obj"].append(sa)
        raw.append(r)
    return dict(nobj=nobj, bg=bg, shape=shape, prop=prop, raw=raw, img_size=(H, W))


def kernel_sigmas(prop, nobj, cam=None):
    """[n,MAXO,2] blur radius in pixels per object and aperture (utils/data_generator.py:16-17)."""
    cam = cam or dict(s=0.1104, rho=(10.0, 10.2), sigma_cam=0.003, pixel_pitch=5.86e-6, mag=4)
    z = np.where(prop[..., 0] > 0, prop[..., 0], 1.0)
    rho = np.asarray(cam["rho"], dtype=np.float64)
    return np.abs((1 / z[..., None] - rho) * cam["s"] + 1) * cam["sigma_cam"] / cam["pixel_pitch"] / cam["mag"]


def _t(a, dev, dt):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dt).to(dev)


def rasterize(shape, nobj, dev, H=147, W=147):
    """The FILL / RING masks of be_datagen_raster_u32, unpacked: shape [n,MAXO,10] int32, nobj [n] -> bool [n,MAXO,2,H,W]
    (what cv2.circle / cv2.drawContours with thickness -1 / 1 paint; objects >= nobj[i] are empty)."""
    shape_t, nobj_t = _t(np.asarray(shape), dev, torch.int32), _t(np.asarray(nobj), dev, torch.int32)
    n = int(nobj_t.shape[0])
    rw = (W + 31) // 32
    planes = torch.empty(lib().be_datagen_raster_words(n, H, W, MAXO), dtype=torch.int32, device=dev)
    check(lib().be_datagen_raster_u32(dptr(shape_t), dptr(nobj_t), n, H, W, MAXO, dptr(planes), stream_ptr(dev)), "be_datagen_raster_u32")
    words = planes.view(n, MAXO, 2, H, rw, 1)
    bits = (words >> torch.arange(32, device=planes.device, dtype=torch.int32)) & 1          # [..., rw, 32], bit x & 31 of word x >> 5
    return bits.reshape(n, MAXO, 2, H, rw * 32)[..., :W].bool()


def generate(scenes, dev, alpha_range=(180.0, 200.0), sigma_read=2.0, seed=synth.SEED_DEFAULT, cam=None, z_far=1.18):
    """Run the GPU generator on `draw_scenes` output.  Returns float64 GPU tensors with the reference's layouts:
    images [n,2,H,W,3] (clean, 0..255, rounded), images_aif [n,H,W,3] (/255), boundary_locations [n,H,W] (0/255),
    image_depths, boundary_depths, boundary_distances [n,H,W], derivative_maps [n,2,H,W,3] (/255),
    alphas [n], images_gt, images_ny [n,2,H,W,3]  (train_val_data_generator.py:132-163, 165-185)."""
    n = int(scenes["nobj"].shape[0])
    H, W = scenes["img_size"]
    f64 = torch.float64
    shape = _t(scenes["shape"], dev, torch.int32)
    prop = _t(scenes["prop"], dev, f64)
    nobj = _t(scenes["nobj"], dev, torch.int32)
    bg = _t(scenes["bg"], dev, f64)
    sig = _t(kernel_sigmas(scenes["prop"], scenes["nobj"], cam), dev, f64)
    new = lambda *s: torch.empty(*s, dtype=f64, device=dev)
    out = dict(images=new(n, 2, H, W, 3), images_aif=new(n, H, W, 3), boundary_locations=new(n, H, W),
               image_depths=new(n, H, W), boundary_depths=new(n, H, W), boundary_distances=new(n, H, W),
               derivative_maps=new(n, 2, H, W, 3), images_gt=new(n, 2, H, W, 3), images_ny=new(n, 2, H, W, 3))
    st = stream_ptr(dev)
    planes = torch.empty(lib().be_datagen_raster_words(n, H, W, MAXO), dtype=torch.int32, device=dev)   # FILL / RING bit planes
    check(lib().be_datagen_raster_u32(dptr(shape), dptr(nobj), n, H, W, MAXO, dptr(planes), st), "be_datagen_raster_u32")
    check(lib().be_datagen_scene_f64(dptr(planes), dptr(prop), dptr(nobj), dptr(bg), n, H, W, MAXO, float(z_far),
                                     dptr(out["images_aif"]), dptr(out["boundary_locations"]), dptr(out["image_depths"]),
                                     dptr(out["boundary_depths"]), st), "be_datagen_scene_f64")
    scratch = torch.empty(lib().be_datagen_blur_scratch_bytes(n, H, W), dtype=torch.uint8, device=dev)
    check(lib().be_datagen_blur_composite_f64(dptr(planes), dptr(prop), dptr(nobj), dptr(bg), dptr(sig), n, H, W, MAXO,
                                              int(scenes["nobj"].max()), dptr(out["images"]), dptr(scratch), scratch.numel(), st),
          "be_datagen_blur_composite_f64")
    check(lib().be_datagen_finish_f64(dptr(out["images"]), dptr(out["boundary_locations"]), dptr(out["boundary_distances"]),
                                      dptr(out["derivative_maps"]), n, H, W, dptr(scratch), scratch.numel(), st),
          "be_datagen_finish_f64")
    alphas = alpha_range[0] + (alpha_range[1] - alpha_range[0]) * synth.hash_uniform(seed, "datagen.alpha", (n,))
    out["alphas"] = _t(alphas, dev, f64)
    check(lib().be_datagen_noise_f64(dptr(out["images"]), dptr(out["alphas"]), float(sigma_read), int(seed) & 0xffffffff, n,
                                     2 * H * W * 3, dptr(out["images_gt"]), dptr(out["images_ny"]), st), "be_datagen_noise_f64")
    return out


def crop_patches(data, n_patch, seed=synth.SEED_DEFAULT, R=21):
    """Patches near boundaries (train_val_data_generator.py:187-275): candidates = pixels within R//2+1 (Chebyshev) of a
    boundary and at least R//2 from the image border; n_patch of them drawn uniformly without replacement over the whole
    set, each with a random aperture.  Returns float64 GPU tensors: patches_aif, patches_gt, patches_ny, derivative_maps
    [n_patch,R,R,3], boundary_locations, image_depths, boundary_depths, boundary_distances [n_patch,R,R], alphas [n_patch]."""
    bl = data["boundary_locations"]
    n, H, W = bl.shape
    dev = bl.device
    half = R // 2
    cand = torch.empty(n, H, W, dtype=torch.uint8, device=dev)
    st = stream_ptr(dev)
    check(lib().be_datagen_candidates_f64(dptr(bl), dptr(cand), n, H, W, half + 1, half, st), "be_datagen_candidates_f64")
    idx = torch.nonzero(cand.view(-1), as_tuple=False).view(-1)                 # candidate list (index plumbing)
    if idx.numel() < n_patch:
        raise RuntimeError(f"crop_patches: only {idx.numel()} candidate pixels for {n_patch} patches")
    key = torch.from_numpy(synth.hash_uniform(seed, "datagen.crop", (idx.numel(),))).to(dev)
    pick = idx[torch.topk(key, n_patch, largest=False, sorted=True).indices].to(torch.int64).contiguous()
    aper = torch.from_numpy(np.floor(synth.hash_uniform(seed, "datagen.aper", (n_patch,)) * 2).astype(np.int32)).to(dev)
    f64 = torch.float64
    new = lambda *s: torch.empty(*s, dtype=f64, device=dev)
    out = dict(patches_aif=new(n_patch, R, R, 3), patches_gt=new(n_patch, R, R, 3), patches_ny=new(n_patch, R, R, 3),
               derivative_maps=new(n_patch, R, R, 3), boundary_locations=new(n_patch, R, R), image_depths=new(n_patch, R, R),
               boundary_depths=new(n_patch, R, R), boundary_distances=new(n_patch, R, R), alphas=new(n_patch))
    ptrs = (C.c_void_p * 6)(*[dptr(data[k]).value for k in ("images_aif", "images_gt", "images_ny", "derivative_maps",
                                                      "image_depths", "boundary_depths")])
    outs = (C.c_void_p * 9)(*[dptr(out[k]).value for k in ("patches_aif", "patches_gt", "patches_ny", "derivative_maps",
                                                     "image_depths", "boundary_depths", "boundary_locations",
                                                     "boundary_distances", "alphas")])
    check(lib().be_datagen_crop_f64(ptrs, dptr(bl), dptr(data["alphas"]), dptr(pick), dptr(aper), n_patch, n, H, W, R, outs, st),
          "be_datagen_crop_f64")
    out["index"] = pick
    out["aperture"] = aper
    return out


_PATCH_FILES = dict(patches_aif="patches_aif", patches_gt="patches_gt", patches_ny="patches_ny",
                    boundary_locations="boundary_locations", image_depths="image_depths", boundary_depths="boundary_depths",
                    boundary_distances="boundary_distances", derivative_maps="derivative_maps", alphas="alphas")
_IMAGE_FILES = ("images_aif", "boundary_locations", "image_depths", "boundary_depths", "boundary_distances",
                "derivative_maps", "alphas", "images_gt", "images_ny")


def save(data, patches, data_path, partition):
    """Write the .npy files of train_val_data_generator.py:158-163,183-185,267-275 (float64, same names)."""
    os.makedirs(os.path.join(data_path, "patches"), exist_ok=True)
    for k in _IMAGE_FILES:
        np.save(os.path.join(data_path, f"{k}_{partition}.npy"), data[k].cpu().numpy())
    if patches is not None:
        for k, stem in _PATCH_FILES.items():
            np.save(os.path.join(data_path, "patches", f"{stem}_{partition}.npy"), patches[k].cpu().numpy())


def main(argv=None):
    """`python -m be_hip.datagen [--num_sample_train N --num_sample_val M --data_path DIR ...]`: the build's
    train_val_data_generator.py (same arguments, utils/args.py mode 'data_gen_train_val'; same output files)."""
    import time
    import utils
    a = utils.get_args("data_gen_train_val", argv=argv)
    dev = torch.device(a.cuda if torch.cuda.is_available() else "cpu")
    for partition, n in (("train", a.num_sample_train), ("val", a.num_sample_val)):
        t0 = time.perf_counter()
        chunks, patches = [], []
        for first in range(0, n, 1024):                                    # bounded device footprint: 1024 images at a time
            m = min(1024, n - first)
            sc = draw_scenes(m, seed=1869 + first, img_size=tuple(a.img_size), num_shape=tuple(a.num_shape),
                             z_range=tuple(a.Z_range), name=f"scenes.{partition}")
            d = generate(sc, dev, alpha_range=tuple(a.alpha), sigma_read=a.sigma, seed=1869 + first, z_far=a.Z_range[1],
                         cam=dict(s=a.cam_params['s'], rho=(a.cam_params['rho_1'], a.cam_params['rho_2']),
                                  sigma_cam=a.cam_params['sigma_cam'], pixel_pitch=a.cam_params['pixel_pitch'], mag=a.mag))
            p = crop_patches(d, 2 * m, seed=1869 + first, R=a.R)
            chunks.append({k: d[k].cpu() for k in _IMAGE_FILES})
            patches.append({k: p[k].cpu() for k in _PATCH_FILES})
        data = {k: torch.cat([c[k] for c in chunks]) for k in _IMAGE_FILES}
        pat = {k: torch.cat([c[k] for c in patches]) for k in _PATCH_FILES}
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        save(data, pat, a.data_path, partition)
        print(f"{partition}: {n} image pairs + {2 * n} patches in {dt:.2f} s ({n / dt:.0f} images/s) -> {a.data_path}")


if __name__ == "__main__":
    main()
