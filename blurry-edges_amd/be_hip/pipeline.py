"""End-to-end depth estimation for one image pair: the build's counterpart of the reference's evaluation
harness (depth_estimator in blurry_edges_test.py:102-145 and blurry_edges_test_big.py:113-192).

image pair [2,3,H,W]  ->  unfold  ->  LocalStage (HIP)  ->  pass-A colours (HIP)  ->  feature normalisation (HIP)
  ->  GlobalStage (HIP)  ->  de-normalisation (HIP)  ->  pass-B records (HIP)
  ->  owner-computes fold (HIP)  ->  six maps + thresholded depth (or, with densify='pp', the DepthCompletion U-Net
  on the folded depth map, blurry_edges_test.py:141-142).
Nothing here computes on the CPU; tensors stay on the GPU until the caller asks for them.
"""
from __future__ import annotations

import math

import torch

from . import native


def params_src_layout(pm):
    """pm [P,38] = [P, (aperture, 19)] -> [2,P,19], the per-image layout global_data_pre_cal.py:27-32 saves in
    params_src_*.npy (and global_training.py reads back)."""
    return pm.view(-1, 2, 19).permute(1, 0, 2)


class DepthPipeline:
    def __init__(self, local_module, global_module, helper, depth_cal, rho_prime=10.39, densify=None, stride=2,
                 densify_pp_module=None):
        """helper: a utils.PostProcessGlobalBase (render options); depth_cal: utils.DepthEtas;
        densify_pp_module: a models.DepthCompletion in eval mode, required for densify='pp'."""
        if densify not in (None, "w", "pp"):
            raise ValueError(f"densify must be None, 'w' or 'pp' (utils/args.py:40), got {densify!r}")
        if densify == "pp" and densify_pp_module is None:
            raise ValueError("densify='pp' needs densify_pp_module (models.DepthCompletion, blurry_edges_test.py:193-196)")
        self.local, self.globl, self.pp = local_module, global_module, densify_pp_module
        self.helper, self.dcal = helper, depth_cal
        self.rho_prime, self.densify, self.stride = rho_prime, densify, stride
        self.depth_thres = 0.0 if densify == "w" else 0.05          # blurry_edges_test.py:109-112
        # global_module / depth_cal may be None when only local_pass is used (global_data_pre_cal.py counterpart)

    # ---- stages --------------------------------------------------------------------------------------
    def local_pass(self, img, window=None):
        """img [2,3,H,W] -> (view of the patch grid, est10 [2P,10], colors [2P,3,3], pm [P,38]).  Both kernels gather
        their 21x21 windows from the image; the unfolded [2,P,3,21,21] tensor of blurry_edges_test.py:120-121 is
        never written."""
        view = native.view_image_pair(img, self.stride, window)
        est10 = self.local.forward_image_pair(img, self.stride, window)
        colors, _ = native.render_colors_view(self.helper.render_opts(wrap_angles=True), est10, view, est10.shape[0] // 2)
        pm = native.local_features(est10, colors)
        return view, est10, colors, pm

    def global_pass(self, pm):
        """pm [P,38] -> est12 [P,12] (de-normalised wedge parameters)."""
        y = self.globl(pm.unsqueeze(0))
        return native.global_denorm(y[0])

    def records(self, est12, img, want=(), window=None):
        opts = self.helper.render_opts(wrap_angles=False)
        return native.render_full(opts, self.dcal.consts, self.rho_prime, self.densify == "w", est12,
                                  native.view_image_pair(img, self.stride, window), want=want, pixels=img)

    # ---- one 147x147 pair (blurry_edges_test.py:117-145) ---------------------------------------------
    @torch.no_grad()
    def __call__(self, img):
        img = img.contiguous()
        _, _, H, W = img.shape
        hp, wp = (H - native.BE_R) // self.stride + 1, (W - native.BE_R) // self.stride + 1
        _, est10, colors, pm = self.local_pass(img)
        est12 = self.global_pass(pm)
        rec, _ = self.records(est12, img)
        maps = native.fold_records(self.helper.render_opts(False), rec, hp, wp, H, W, self.stride, self.densify == "w")
        if self.densify == "pp":
            maps["depth_map"] = self.pp(maps["depth"][None, None])[0, 0]
        else:
            maps["depth_map"] = torch.where(maps["conf"] > self.depth_thres, maps["depth"], torch.zeros_like(maps["depth"]))
        maps.update(est10=est10, colors_a=colors, est12=est12, records=rec)
        return maps

    # ---- big image: 147x147 blocks with margin patches dropped (blurry_edges_test_big.py:116-189) -----
    @staticmethod
    def big_windows(H, W, block=147, n_margin=10, stride=2, R=21):
        """Block schedule of the big-image tiler (blurry_edges_test_big.py:118-119,166-177): a list of
        ((top, left, block, block) pixel window, (vs, ve, hs, he) kept rows/cols of the block's patch grid,
        (Vs, Hs) where that kept window starts in the big patch grid)."""
        bstride = block - R + stride - 2 * stride * n_margin                        # 88
        nb_v = math.ceil((H - R - 2 * stride * n_margin + stride) / bstride)
        nb_h = math.ceil((W - R - 2 * stride * n_margin + stride) / bstride)
        hp = (block - R) // stride + 1                                              # 64
        step = hp - 2 * n_margin                                                    # 44 patches = bstride / stride
        out = []
        for bi in range(nb_v):
            for bj in range(nb_h):
                vs = 0 if bi == 0 else n_margin
                ve = hp if bi == nb_v - 1 else hp - n_margin
                hs = 0 if bj == 0 else n_margin
                he = hp if bj == nb_h - 1 else hp - n_margin
                out.append(((bi * bstride, bj * bstride, block, block), (vs, ve, hs, he), (bi * step + vs, bj * step + hs)))
        return out

    @torch.no_grad()
    def run_big(self, img, block=147, n_margin=10, rank=0, world=1, group=None, dedup=True):
        """rank / world: this process handles the blocks shard.my_blocks gives it; the record grid is completed with one
        all-reduce (shard.assemble_records) and every rank folds the full image.  world = 1: no communication.

        dedup=True (default): the reference calls the CNN once per 147x147 block (blurry_edges_test_big.py:142-165): 36 x 8192 =
        294 912 patches for the 2 x 284 x 284 = 161 312 DISTINCT windows of a 587x587 pair (neighbouring blocks share their 20
        margin rows / columns of patches).  LocalStage, pass A and the feature glue are per-patch and position-independent bit for
        bit, so here they run ONCE over the whole patch grid and every block gathers its 64x64 rows of `pm` from it: same
        numbers, 45 % fewer patches.  With world > 1 the local pass is sharded by rows of the patch grid (shard.row_range) and the
        feature grid is completed by one all-reduce before the blocks - which GlobalStage's attention keeps whole - are dealt out.
        dedup=False: the reference's schedule, one local pass per block."""
        from . import shard
        img = img.contiguous()
        _, _, H, W = img.shape
        s, R = self.stride, native.BE_R
        hp = (block - R) // s + 1
        HP, WP = (H - R) // s + 1, (W - R) // s + 1                                 # 284
        # blocks are windows of the big image: no cropped copies, no unfolded copies
        big = torch.zeros(HP * WP, native.RECORD_FLOATS, dtype=torch.float32, device=img.device).view(HP, WP, -1)
        wins = self.big_windows(H, W, block, n_margin, s, R)
        mine = shard.my_blocks(len(wins), rank, world)
        if dedup:
            r0, r1 = shard.row_range(HP, rank, world)
            if world > 1:
                grid = torch.zeros(HP, WP, 38, dtype=torch.float32, device=img.device)
                if r1 > r0:
                    grid[r0:r1] = self.local_pass(img, (r0 * s, 0, (r1 - r0 - 1) * s + R, W))[3].view(r1 - r0, WP, 38)
                grid = shard.assemble_records(grid, group)                          # x + 0 is exact: rows are owned once
            else:
                grid = self.local_pass(img)[3].view(HP, WP, 38)
            # a block whose pixel window starts at (top, left) owns the patch rows top/s .. top/s + hp - 1 of the big grid
            feats = [grid[wins[k][0][0] // s:wins[k][0][0] // s + hp, wins[k][0][1] // s:wins[k][0][1] // s + hp].reshape(hp * hp, 38)
                     for k in mine]
        else:
            # local stage block by block (8192 patches each = one CNN sub-batch)
            feats = [self.local_pass(img, wins[k][0])[3] for k in mine]
        # GlobalStage on groups of blocks in one batch (attention at batch 1 leaves three quarters of the SIMD slots empty),
        # then pass B per block
        est = []
        for g0 in range(0, len(mine), 12):
            y = self.globl(torch.stack(feats[g0:g0 + 12]))                          # [g,4096,12]
            est.extend(native.global_denorm(y[i]) for i in range(y.shape[0]))
        for k, est12 in zip(mine, est):
            win, (vs, ve, hs, he), (Vs, Hs) = wins[k]
            rec, _ = self.records(est12, img, window=win)
            big[Vs:Vs + ve - vs, Hs:Hs + he - hs] = rec.view(hp, hp, -1)[vs:ve, hs:he]
        if world > 1:
            big = shard.assemble_records(big, group)
        maps = native.fold_records(self.helper.render_opts(False), big.view(HP * WP, -1), HP, WP, H, W, s, self.densify == "w")
        if self.densify == "pp":
            maps["depth_map"] = self.pp(maps["depth"][None, None])[0, 0]
        else:                                                                       # blurry_edges_test_big.py:189
            maps["depth_map"] = torch.where(maps["conf"] > 0.05, maps["depth"], torch.zeros_like(maps["depth"]))
        return maps
