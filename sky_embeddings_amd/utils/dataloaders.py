"""utils/dataloaders.py mirror for the HDF5 cutout path (H5Dataset, MaskGenerator,
build_h5_dataloader, get_augmentations).  The augmentation pipeline (utils/dataloaders.py:14-106) is
``sky_embeddings_amd.augment.Augmenter``: torchvision's parameter draws on the host, the arithmetic in one HIP launch.  The
FITS tile sampler is out of scope.

Differences from the reference, none of which change a returned value:
  * the file is parsed once and datasets are memory-mapped (hdf5_lite) instead of re-opening
    the HDF5 file per item (utils/dataloaders.py:289);
  * ``open_h5`` is the single HDF5 access point (h5py is not required).
"""
from __future__ import annotations

import numpy as np
import torch

from .. import hdf5_lite


def open_h5(path):
    return hdf5_lite.File(path, "r")


def extract_center(array, n):
    """utils/dataloaders.py: central n x n crop of a [C,H,W] (or [..,H,W]) array."""
    h, w = array.shape[-2:]
    top, left = (h - n) // 2, (w - n) // 2
    return array[..., top:top + n, left:left + n]


class MaskGenerator:
    """utils/dataloaders.py:155-219 -- SimMIM per-channel random patch masks (ratio = U(0,1)*max)."""

    def __init__(self, input_size=192, patch_size=4, max_mask_ratio=0.9, num_mask_chans=1):
        self.input_size, self.patch_size = input_size, patch_size
        self.max_mask_ratio, self.num_mask_chans = max_mask_ratio, num_mask_chans
        self.n_patches = self.input_size // self.patch_size
        self.token_count = self.n_patches ** 2

    def __call__(self):
        mask_ratio = torch.rand(1).item() * self.max_mask_ratio
        mask_count = int(torch.ceil(torch.tensor(self.token_count * mask_ratio)).item())
        masks = torch.zeros((self.num_mask_chans, self.token_count), dtype=torch.int)
        for i in range(self.num_mask_chans):
            masks[i, torch.randperm(self.token_count)[:mask_count]] = 1
        masks = masks.view(self.num_mask_chans, self.n_patches, self.n_patches)
        masks = masks.repeat_interleave(self.patch_size, dim=1).repeat_interleave(self.patch_size, dim=2)
        return masks.squeeze(0) if self.num_mask_chans == 1 else masks


class H5Dataset(torch.utils.data.Dataset):
    """utils/dataloaders.py:221-328: (cutout f32 [C,H,W] clipped at pixel_min, mask, ra_dec f32[2][, labels])."""

    def __init__(self, data_file, img_size, patch_size, num_channels, max_mask_ratio, num_patches=None, label_keys=None,
                 transform=None, pixel_min=-3., pixel_max=None, indices=None):
        self.data_file, self.transform, self.img_size = data_file, transform, img_size
        self.num_patches, self.label_keys = num_patches, label_keys
        self.pixel_min, self.pixel_max, self.indices = pixel_min, pixel_max, indices
        self.max_mask_ratio = max_mask_ratio
        self.mask_generator = (MaskGenerator(input_size=img_size, patch_size=patch_size, max_mask_ratio=max_mask_ratio,
                                             num_mask_chans=num_channels) if max_mask_ratio is not None else None)
        self._f = None

    def _file(self):
        if self._f is None:  # opened lazily so that DataLoader workers each map the file themselves
            self._f = open_h5(self.data_file)
        return self._f

    def __getstate__(self):
        d = dict(self.__dict__)
        d["_f"] = None
        return d

    def __len__(self):
        if self.indices is not None:
            return len(self.indices)
        return len(self._file()['cutouts'])

    def __getitem__(self, idx):
        if self.indices is not None:
            idx = self.indices[idx]
        f = self._file()
        cutout = f['cutouts'][idx]
        if self.pixel_min is not None:
            cutout[cutout < self.pixel_min] = self.pixel_min      # NaN compares False: preserved
        if self.pixel_max is not None:
            cutout[cutout > self.pixel_max] = self.pixel_max
        if (np.array(cutout.shape[1:]) > self.img_size).any():
            cutout = np.ascontiguousarray(extract_center(cutout, self.img_size))
        ra_dec = torch.from_numpy(np.asarray([f['ra'][idx], f['dec'][idx]]).astype(np.float32))
        labels = None
        if self.label_keys is not None:
            labels = [f[k][idx] for k in self.label_keys]
            if 'class' in self.label_keys:
                labels = torch.from_numpy(np.asarray(labels).astype(np.int64)).long()
            else:
                labels = torch.from_numpy(np.asarray(labels).astype(np.float32))
        cutout = torch.from_numpy(cutout).to(torch.float32)
        if self.transform is not None:
            cutout = self.transform(cutout)
        mask = self.mask_generator() if self.mask_generator is not None else torch.zeros_like(cutout)
        if self.label_keys is None:
            return cutout, mask, ra_dec
        return cutout, mask, ra_dec, labels


def get_augmentations(img_size=64, flip=True, crop=True, brightness=0.8, noise=0.01, nan_channels=2):
    """utils/dataloaders.py:90-106: the target-augmentation pipeline; the returned object maps one [C, H, W] tensor to an
    augmented copy (like the reference's v2.Compose) and a whole batch to ``1 + A`` copies per sample with ``.batch``."""
    from ..augment import Augmenter
    return Augmenter(img_size=img_size, flip=flip, crop=crop, brightness=brightness, noise=noise, nan_channels=nan_channels)


def build_h5_dataloader(filename, batch_size, num_workers, patch_size=8, num_channels=5, max_mask_ratio=None,
                        label_keys=None, img_size=64, num_patches=None, augment=False, brightness=0.8, noise=0.01,
                        nan_channels=2, shuffle=True, indices=None, transforms=None, sampler=None):
    """utils/dataloaders.py:134-153 (+ optional ``sampler`` for one-process-per-GPU sharding)."""
    if (transforms is None) and augment:
        transforms = get_augmentations(img_size=img_size, brightness=brightness, noise=noise, nan_channels=nan_channels)
    dataset = H5Dataset(filename, img_size=img_size, patch_size=patch_size, num_channels=num_channels,
                        max_mask_ratio=max_mask_ratio, num_patches=num_patches, label_keys=label_keys,
                        transform=transforms, indices=indices)
    return torch.utils.data.DataLoader(dataset, batch_size=batch_size, shuffle=shuffle if sampler is None else False,
                                       sampler=sampler, num_workers=num_workers, pin_memory=torch.cuda.is_available())


def build_fits_dataloader(*args, **kwargs):
    raise NotImplementedError("FITS tile streaming (utils/dataloaders.py:538-654) needs astropy and survey tiles: out "
                              "of scope (SURVEY.md §2 row 5); use train_data_file = <cutouts>.h5")
