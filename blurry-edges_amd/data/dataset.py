"""Datasets over the .npy files the generators write (data/dataset.py:6-73 of the reference; file names and array
layouts of train_val_data_generator.py:158-163,183-185,267-275 and test_data_generator.py).  Items are divided by the
per-sample photon level alpha exactly where the reference does it, so the training loops see the same tensors.

mode 'local'      -> (img_ny/alpha [21,21,3], img_gt/alpha [21,21,3], bndry_dist [21,21], deri [19,19,3])
mode 'global_pre' -> img_ny/alpha [2,H,W,3]
mode 'global'     -> (params_src [P,38], img_ny/alpha, img_gt/alpha [2,H,W,3], bndry_dist [H,W], deri [2,H-2,W-2,3],
                      bndry_depth [H,W])
TestDataset       -> (images_ny/alpha [2,H,W,3], depth_map [H,W])
"""
import os

import numpy as np
import torch
from torch.utils.data import Dataset

_FILES = {
    'local': dict(img_ny='patches_ny', img_gt='patches_gt', alpha='alphas', bndry_dist='boundary_distances', deri='derivative_maps'),
    'global_pre': dict(img_ny='images_ny', alpha='alphas'),
    'global': dict(input_param='params_src', img_ny='images_ny', img_gt='images_gt', deri='derivative_maps',
                   bndry_dist='boundary_distances', bndry_depth='boundary_depths', alpha='alphas'),
}
_ORDER = {'local': ('img_ny', 'img_gt', 'bndry_dist', 'deri'), 'global_pre': ('img_ny',),
          'global': ('input_param', 'img_ny', 'img_gt', 'bndry_dist', 'deri', 'bndry_depth')}
_PER_ALPHA = ('img_ny', 'img_gt')


class ShapeDataset(Dataset):
    def __init__(self, device, data_path='.', train=False, mode='local'):
        if mode not in _FILES:
            raise ValueError(f"ShapeDataset: unknown mode {mode!r}")
        part = 'train' if train else 'val'
        self.mode, self.device = mode, device
        for attr, stem in _FILES[mode].items():
            arr = torch.from_numpy(np.load(os.path.join(data_path, f'{stem}_{part}.npy'))).float()
            if attr == 'deri':                        # the Sobel maps lose their one-pixel border (valid convolution)
                arr = arr[..., 1:-1, 1:-1, :]
            setattr(self, attr, arr)

    def __len__(self):
        return self.img_ny.shape[0]

    def __getitem__(self, idx):
        alpha = self.alpha[idx].to(self.device)
        out = []
        for attr in _ORDER[self.mode]:
            t = getattr(self, attr)[idx, ...].to(self.device)
            out.append(t / alpha if attr in _PER_ALPHA else t)
        return out[0] if len(out) == 1 else tuple(out)

    # ---- extension (not in the reference): whole batches gathered on the device --------------------------------------
    def batches(self, batch_size, shuffle=False, drop_last=True, generator=None, rank=0, world=1):
        """Yields what DataLoader(self, batch_size, shuffle, drop_last) would collate (same tuple order, same division by
        alpha), but with the arrays resident on `device` and one gather per batch instead of one host-to-device copy per
        sample and tensor: at batch 64 the per-sample path costs more than the training step itself."""
        if getattr(self, "_resident", None) is None:
            self._resident = {a: getattr(self, a).to(self.device) for a in tuple(_ORDER[self.mode]) + ("alpha",)}
        r = self._resident
        # data parallel (world > 1): every rank draws the SAME permutation and materialises only its own batches - batch i goes
        # to rank i % world, and the batches past the last full round of `world` are dropped so that all ranks take equally many
        # steps (each step ends in a collective)
        index_lists = list(_batches(self, batch_size, shuffle, drop_last, generator))
        if world > 1:
            index_lists = index_lists[:len(index_lists) - len(index_lists) % world][rank::world]
        for idx in index_lists:
            idx = idx.to(self.device)
            al = r["alpha"][idx]
            out = []
            for attr in _ORDER[self.mode]:
                t = r[attr][idx]
                out.append(t / al.view(-1, *([1] * (t.dim() - 1))) if attr in _PER_ALPHA else t)
            yield out[0] if len(out) == 1 else tuple(out)


def _batches(ds, batch_size, shuffle, drop_last, generator):
    n = len(ds)
    order = torch.randperm(n, generator=generator) if shuffle else torch.arange(n)
    stop = n - n % batch_size if drop_last else n
    for lo in range(0, stop, batch_size):
        yield order[lo:lo + batch_size]


class TestDataset(Dataset):
    __test__ = False                                   # not a pytest class

    def __init__(self, device, data_path='.'):
        load = lambda stem: torch.from_numpy(np.load(os.path.join(data_path, stem + '.npy'))).float()
        self.ny_img, self.depth_map, self.alpha = load('images_ny'), load('depth_maps'), load('alphas')
        self.device = device

    def __len__(self):
        return self.ny_img.shape[0]

    def __getitem__(self, idx):
        alpha = self.alpha[idx].to(self.device)
        return self.ny_img[idx, ...].to(self.device) / alpha, self.depth_map[idx, :, :].to(self.device)
