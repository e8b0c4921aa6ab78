"""Oracle: overlapping-patch unfold / fold and the big-image block tiler (TEST INFRASTRUCTURE).

Index math written out directly (no nn.Unfold/nn.Fold) so the golden comparison against the
reference's torch.nn.Unfold / Fold (blurry_edges_test.py:120-121; utils/postprocessing_loss.py:137-173;
blurry_edges_test_big.py:116-125,142-183) is a real check of the ordering.
"""
from __future__ import annotations

import math

import torch

R = 21


def grid_dims(H, W, stride=2, r=R):
    return (H - r) // stride + 1, (W - r) // stride + 1            # postprocessing_loss.py:137-138


def unfold_patches(img: torch.Tensor, stride=2, r=R) -> torch.Tensor:
    """img [B,C,H,W] -> [B, Hp*Wp, C, r, r]; patch (i,j) covers rows stride*i.., cols stride*j..,
    flat index i*Wp+j  (the ordering blurry_edges_test.py:120-121 gets from nn.Unfold); r = 21, or 19 for the
    derivative maps of global_training.py:113-116."""
    B, C, H, W = img.shape
    Hp, Wp = grid_dims(H, W, stride, r)
    out = img.new_empty(B, Hp, Wp, C, r, r)
    for a in range(r):
        for c in range(r):
            out[:, :, :, :, a, c] = img[:, :, a:a + stride * Hp:stride, c:c + stride * Wp:stride].permute(0, 2, 3, 1)
    return out.reshape(B, Hp * Wp, C, r, r)


def fold_sum(patches: torch.Tensor, H, W, stride=2) -> torch.Tensor:
    """patches [B, Hp*Wp, C, 21, 21] -> overlap SUM [B,C,H,W] (what nn.Fold computes)."""
    B, P, C = patches.shape[:3]
    Hp, Wp = grid_dims(H, W, stride)
    assert P == Hp * Wp
    p = patches.reshape(B, Hp, Wp, C, R, R)
    out = patches.new_zeros(B, C, H, W)
    for r in range(R):
        for c in range(R):
            out[:, :, r:r + stride * Hp:stride, c:c + stride * Wp:stride] += p[:, :, :, :, r, c].permute(0, 3, 1, 2)
    return out


def num_patches(H, W, stride=2, dtype=torch.float32):
    Hp, Wp = grid_dims(H, W, stride)
    return fold_sum(torch.ones(1, Hp * Wp, 1, R, R, dtype=dtype), H, W, stride)[0, 0]   # :139-143


def fold_mean(patches, H, W, stride=2):
    """local2global_color / local2global_bndry (postprocessing_loss.py:151-164)."""
    return fold_sum(patches, H, W, stride) / num_patches(H, W, stride, patches.dtype)


def fold_depth(depth_map, depth_mask, H, W, stride=2):
    """local2global_depth (:166-173): depth_map [B,P,21,21] float, depth_mask [B,P,21,21] int
    -> (depth [B,H,W], confidence [B,H,W])."""
    cnt = fold_sum((depth_mask > 0).to(depth_map.dtype)[:, :, None], H, W, stride)[:, 0]
    conf = cnt / num_patches(H, W, stride, depth_map.dtype)
    s = fold_sum(depth_map[:, :, None], H, W, stride)[:, 0]
    return s / torch.where(cnt > 0, cnt, torch.ones_like(cnt)), conf


def big_tiler(big=587, img=147, stride=2, n_margin=10):
    """Block tiling of blurry_edges_test_big.py:116-125,142-183.
    Returns dict(block_stride, n_block, big_grid, blocks=[(bi,bj, top,left, (vs,ve,hs,he) kept window in the
    block's 64x64 patch grid, (Vs,Ve,Hs,He) destination window in the big patch grid)])."""
    block_stride = img - R + stride - 2 * stride * n_margin                          # 88
    n_block = math.ceil((big - R - 2 * stride * n_margin + stride) / block_stride)   # 6
    hp = (img - R) // stride + 1                                                     # 64
    big_hp = (big - R) // stride + 1                                                 # 284
    step = block_stride // stride                                                    # 44 patches
    blocks = []
    for bi in range(n_block):
        for bj in range(n_block):
            top, left = bi * block_stride, bj * block_stride
            vs = 0 if bi == 0 else n_margin
            ve = hp if bi == n_block - 1 else hp - n_margin
            hs = 0 if bj == 0 else n_margin
            he = hp if bj == n_block - 1 else hp - n_margin
            Vs, Hs = bi * step + vs, bj * step + hs
            blocks.append((bi, bj, top, left, (vs, ve, hs, he), (Vs, Vs + ve - vs, Hs, Hs + he - hs)))
    return dict(block_stride=block_stride, n_block=n_block, big_grid=big_hp, blocks=blocks)
