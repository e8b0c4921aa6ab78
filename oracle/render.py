"""Oracle: blurred-wedge patch renderer, ridge colour solve, masks, loss (TEST INFRASTRUCTURE).

Flat "one row per patch" layout throughout: params [N,8|10|12], patches [N,3,21,21].  The
reference's global layout ([B,C,21,21,Hp,Wp], patch index fastest) is a pure permutation of this.
Restates utils/postprocessing_loss.py:7-128, blurry_edges_test.py:19-79, local_training.py:32-52.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

from . import depth as odepth

R = 21
TWO_PI = 2 * math.pi
DELTA = 0.07                      # normalized_gaussian default, postprocessing_loss.py:97
ALPHA_LAMBDA = 5e-3               # utils/args.py:13
# postprocessing_loss.py:14 -> 4.862025; the reference holds it as a float32 tensor (lambda*eye(3), :122),
# so every dtype of this oracle uses the float32-rounded value.
LAMBDA_RIDGE = float(torch.tensor((ALPHA_LAMBDA * R ** 2) ** 2, dtype=torch.float32))


def pixel_grid(dtype=torch.float32):
    """x = column coordinate, y = row coordinate, linspace(-1,1,21) each (postprocessing_loss.py:15-17)."""
    lin = torch.linspace(-1.0, 1.0, R)              # built in float32 like the reference, then cast
    y, x = torch.meshgrid([lin, lin], indexing="ij")
    return x.to(dtype), y.to(dtype)


def _edge_and_axial(px, py, vx, vy, ang):
    s, c = torch.sin(ang), torch.cos(ang)
    edge = -s * (px - vx) + c * (py - vy)           # dist4edge  :26-27
    axial = c * (px - vx) + s * (py - vy)           # dist4axial :29-30
    return edge, axial


def _ray_distance(px, py, vx, vy, ang, w=1.0):
    # :50-76 -- behind the vertex the distance is measured to the vertex itself
    edge, axial = _edge_and_axial(px, py, vx, vy, ang)
    sgn = torch.where(edge < 0, -torch.ones_like(edge), torch.ones_like(edge))
    return torch.where(axial < 0, torch.sqrt(edge ** 2 + (axial * w) ** 2) * sgn, edge)


def params2dists(params8: torch.Tensor, w: float = 1.0) -> torch.Tensor:
    """[N,8] (x0,y0,x1,y1,theta1,phi1,theta2,phi2) -> signed wedge distances [N,2,21,21]
    (postprocessing_loss.py:43-86)."""
    dt = params8.dtype
    px, py = pixel_grid(dt)
    px, py = px[None], py[None]
    col = lambda i: params8[:, i].reshape(-1, 1, 1)
    x0, y0, x1, y1, t1, f1, t2, f2 = (col(i) for i in range(8))
    pi = torch.pi
    sg1 = torch.where(torch.remainder(f1, 2 * pi) < pi, torch.ones_like(f1), -torch.ones_like(f1))   # :46
    sg2 = torch.where(torch.remainder(f2, 2 * pi) < pi, torch.ones_like(f2), -torch.ones_like(f2))   # :47
    d11 = _ray_distance(px, py, x0, y0, t1, w)
    d12 = _ray_distance(px, py, x0, y0, t1 + f1, w)
    d21 = _ray_distance(px, py, x1, y1, t2, w)
    d22 = _ray_distance(px, py, x1, y1, t2 + f2, w)
    one = torch.ones_like(d11)
    in1 = sg1 * torch.where((sg1 * d11 > 0) & (sg1 * d12 < 0), one, -one)      # strict  :80
    in2 = sg2 * torch.where((sg2 * d21 >= 0) & (sg2 * d22 <= 0), one, -one)    # closed  :81
    dist1 = torch.min(d11.abs(), d12.abs()) * in1
    dist2 = torch.min(d21.abs(), d22.abs()) * in2
    return torch.stack([dist1, dist2], dim=1)


def params2etas(p: torch.Tensor) -> torch.Tensor:
    return 10 ** (torch.erf(p) * 2 - 2)                                          # :88-89


def dists2indicators(dists: torch.Tensor, etas: torch.Tensor) -> torch.Tensor:
    """dists [N,2,21,21], etas [N,2] -> (u0,u1,u2) [N,3,21,21]  (:91-95).
    sqrt(2) is the float32 value of sqrt(int 2) as in the reference, promoted with the data."""
    root2 = torch.sqrt(torch.tensor(2)).to(dists.dtype)
    h = 0.5 * (1.0 + torch.erf(dists / (root2 * etas[:, :, None, None])))
    h1, h2 = h[:, 0], h[:, 1]
    return torch.stack([(1.0 - h1) * (1.0 - h2), h1 * (1.0 - h2), h2], dim=1)


def normalized_gaussian(x, delta=DELTA):
    return torch.exp(-x ** 2 / delta ** 2)                                        # :97-98


def inverse_3by3_cayley(A: torch.Tensor) -> torch.Tensor:
    """The reference's Cayley-Hamilton inverse (:104-112, adj from :127-128) -- float32-unstable."""
    eye = torch.eye(3, dtype=A.dtype)
    tr = lambda M: torch.diagonal(M, dim1=-2, dim2=-1).sum(-1)
    A2 = A @ A
    A3 = A2 @ A
    t1, t2, t3 = tr(A), tr(A2), tr(A3)
    det = (t1 ** 3 - 3 * t1 * t2 + 2 * t3) / 6
    adj = A2 - t1[..., None, None] * A + ((t1 ** 2 - t2) / 2)[..., None, None] * eye
    return adj / det[..., None, None]


def ridge_colors(wedge_sets, image_sets, lam=LAMBDA_RIDGE, inverse="cayley"):
    """Ridge regression of wedge colours.
    wedge_sets: list of [N,3,21,21] (one per aperture image that shares the colours),
    image_sets: list of [N,3,21,21] pixel data in the same order.
    Rows = pixels (row-major) of set 0, then set 1 (blurry_edges_test.py:19-28; local_training.py:37-41).
    Returns (colours [N,3(rgb),3(wedge)], AtA+lam*I [N,3,3], Aty [N,3(wedge),3(rgb)])."""
    A = torch.cat([w.flatten(2).transpose(1, 2) for w in wedge_sets], dim=1)      # [N,rows,3]
    y = torch.cat([im.flatten(2).transpose(1, 2) for im in image_sets], dim=1)    # [N,rows,3(rgb)]
    At = A.transpose(1, 2)
    G = At @ A + lam * torch.eye(3, dtype=A.dtype)
    b = At @ y
    if inverse == "cayley":
        C = inverse_3by3_cayley(G) @ b
    else:
        C = torch.linalg.solve(G, b)
    return C.transpose(1, 2), G, b


def composite(wedges: torch.Tensor, colors: torch.Tensor) -> torch.Tensor:
    """patch[c,px] = sum_k u_k[px] * C[c,k]   (local_training.py:41; blurry_edges_test.py:40-41)."""
    return torch.einsum("nkhw,nck->nchw", wedges, colors)


def boundary_distance(dists):
    d1, d2 = dists[:, 0], dists[:, 1]
    return torch.where(d2 >= 0, d2, torch.where(d1.abs() < d2.abs(), d1.abs(), d2.abs()))   # :59-60 (test)


def boundary_map(dists):
    return normalized_gaussian(boundary_distance(dists))


def depth_mask(dists, densify=None):
    """int32 [N,21,21] in {0,1,2}  (blurry_edges_test.py:47-54)."""
    d1, d2 = dists[:, 0], dists[:, 1]
    if densify == "w":
        m = (d1 > 0).to(torch.int32)
        m2 = (d2 > 0).to(torch.int32) * 2
        return torch.where(m2 == 2, m2, m)
    m = (normalized_gaussian(d1) > 0.5).to(torch.int32)
    m2 = (normalized_gaussian(d2) > 0.5).to(torch.int32) * 2
    return torch.where((m2 == 2) | (d2 >= 0), m2, m)


# --------------------------------------------------------------------------- composite passes

def wrap_angles10(p):
    """blurry_edges_test.py:123-127: angles -> remainder(., 2pi)."""
    q = p.clone()
    q[:, 4:8] = torch.remainder(p[:, 4:8], 2 * torch.pi)
    return q


def render_pass_a(params10, patches, inverse="cayley"):
    """Colours-only pass for one image (blurry_edges_test.py:128 -> :81-92 -> :30-34):
    params10 [N,10] (angles already wrapped by the caller), patches [N,3,21,21]
    -> dict(dists, etas, wedges, G, b, colors [N,3,3], recon [N,3,21,21])."""
    dists = params2dists(params10[:, :8])
    etas = params2etas(params10[:, 8:10])
    wedges = dists2indicators(dists, etas)
    colors, G, b = ridge_colors([wedges], [patches], inverse=inverse)
    return dict(dists=dists, etas=etas, wedges=wedges, G=G, b=b, colors=colors,
                recon=composite(wedges, colors))


def local_depth(c, params10_img1, params10_img2):
    """Config-2 composition: eta_k per image from the two CNN outputs of a pair,
    depth_k = etas2depth(eta_k(img1), eta_k(img2)), k=1,2.  -> [P,2]."""
    e1 = params2etas(params10_img1[:, 8:10])
    e2 = params2etas(params10_img2[:, 8:10])
    return torch.stack([odepth.etas2depth(c, e1[:, 0], e2[:, 0]),
                        odepth.etas2depth(c, e1[:, 1], e2[:, 1])], dim=1)


def render_pass_b(c, params12, patches1, patches2, rho_prime=10.39, densify=None, inverse="cayley"):
    """Full pass for a pair (blurry_edges_test.py:36-74).  params12 [P,12] de-normalised,
    patches1/2 [P,3,21,21] = the two aperture images of each patch position."""
    dists = params2dists(params12[:, :8])
    etas = params2etas(params12[:, 8:12])                       # (w1,i1),(w2,i1),(w1,i2),(w2,i2)
    w1 = dists2indicators(dists, etas[:, 0:2])
    w2 = dists2indicators(dists, etas[:, 2:4])
    colors, G, b = ridge_colors([w1, w2], [patches1, patches2], inverse=inverse)
    p1, p2 = composite(w1, colors), composite(w2, colors)
    z1 = odepth.etas2depth(c, etas[:, 0], etas[:, 2])
    z2 = odepth.etas2depth(c, etas[:, 1], etas[:, 3])
    mask = depth_mask(dists, densify)
    zmap = torch.where(mask == 1, z1[:, None, None],
                       torch.where(mask == 2, z2[:, None, None], mask.to(z1.dtype)))
    bnd = boundary_map(dists)
    tiny = torch.full_like(etas[:, 0:2], 1e-4)
    shp = composite(dists2indicators(dists, tiny), colors)
    s1 = odepth.depth2sigma(c, z1, rho_prime)
    s2 = odepth.depth2sigma(c, z2, rho_prime)
    s1 = torch.where((mask == 1).sum(dim=(1, 2)) > 0, s1, torch.full_like(s1, 1e-4))
    s2 = torch.where((mask == 2).sum(dim=(1, 2)) > 0, s2, torch.full_like(s2, 1e-4))
    refoc = composite(dists2indicators(dists, torch.stack([s1, s2], dim=1)), colors)
    return dict(dists=dists, etas=etas, colors=colors, G=G, b=b, patches1=p1, patches2=p2,
                shpd=shp, refoc=refoc, boundary=bnd, depth_map=zmap, depth_mask=mask,
                depth1=z1, depth2=z2, sig_refoc=torch.stack([s1, s2], dim=1))


# --------------------------------------------------------------------------- local training loss

def image_derivative(img):
    """Per-channel Sobel magnitude, valid padding (postprocessing_loss.py:19-20,114-117)."""
    kx = torch.tensor([[-1, 0, 1], [-2, 0, 2], [-1, 0, 1]], dtype=img.dtype)
    ky = torch.tensor([[1, 2, 1], [0, 0, 0], [-1, -2, -1]], dtype=img.dtype)
    kx = kx[None, None].repeat(3, 1, 1, 1)
    ky = ky[None, None].repeat(3, 1, 1, 1)
    return torch.sqrt(F.conv2d(img, kx, groups=3) ** 2 + F.conv2d(img, ky, groups=3) ** 2 + 1e-8)


def local_loss(est, img_fit, gt_img, bndry_dist, deri, beta_b=1e-3, beta_s=5e-4, inverse="cayley"):
    """local_training.py:32-52.  est [B,10] (not modified here; the reference wraps angles in place),
    img_fit / gt_img [B,21,21,3] channels-last, bndry_dist [B,21,21], deri [B,19,19,3]."""
    p = torch.cat([est[:, :4], torch.remainder(est[:, 4:8], 2 * torch.pi), est[:, 8:]], dim=1)
    dists = params2dists(p[:, :8])
    wedges = dists2indicators(dists, params2etas(p[:, 8:10]))
    colors, _, _ = ridge_colors([wedges], [img_fit.permute(0, 3, 1, 2)], inverse=inverse)
    patches = composite(wedges, colors)
    bnd = boundary_map(dists)
    loss = ((gt_img - patches.permute(0, 2, 3, 1)) ** 2).sum(-1).mean() \
        + beta_b * ((bndry_dist * bnd) ** 2).mean() \
        + beta_s * ((deri.permute(0, 3, 1, 2) - image_derivative(patches)) ** 2).sum(1).mean()
    return loss, patches, bnd
