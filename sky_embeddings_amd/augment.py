"""Target augmentations of the similarity search (utils/dataloaders.py:14-106, applied by utils/eval_fns.py:88-108 and
similarity_search.py:160) without torchvision: the random PARAMETERS are drawn on the host exactly as torchvision's
transforms draw them (RandomHorizontalFlip / RandomVerticalFlip p = 0.5, RandomResizedCrop.get_params with its ten
attempts and central-crop fallback, the reference's brightness / noise / channel-NaN transforms), the ARITHMETIC runs in
one HIP launch per batch (csrc/augment.hip): every sample followed by its ``num_augmentations`` augmented copies.
"""
from __future__ import annotations

import math

import torch

from . import ops


class Augmenter:
    """``get_augmentations(img_size, flip, crop, brightness, noise, nan_channels)`` of the reference as an object.
    ``batch(samples, A)`` is the product path (device, whole batch); calling the object on one [C, H, W] (or [B, C, H, W])
    tensor returns one augmented copy of it, like the reference's ``v2.Compose`` pipeline does."""

    def __init__(self, img_size=64, flip=True, crop=True, brightness=0.8, noise=0.01, nan_channels=2, scale=(0.8, 1.0),
                 ratio=(0.9, 1.1), seed=None, device="cuda"):
        self.img_size, self.flip, self.crop = img_size, flip, crop
        self.brightness, self.noise, self.nan_channels = brightness, noise, nan_channels
        self.scale, self.ratio = scale, ratio
        self.device = torch.device(device)
        self.gen = torch.Generator()
        if seed is not None:
            self.gen.manual_seed(seed)

    # -- parameter draws (host) ---------------------------------------------------------------------------------------
    def _uniform(self, n, lo, hi):
        return torch.rand(n, generator=self.gen, dtype=torch.float64) * (hi - lo) + lo

    def draw(self, n, C, S):
        """-> (params float32 [n, 8] = {flip_h, flip_v, top, left, h, w, brightness, sigma}, nan_mask int32 [n])."""
        p = torch.zeros(n, 8, dtype=torch.float64)
        if self.flip:
            p[:, 0] = (torch.rand(n, generator=self.gen) < 0.5).double()
            p[:, 1] = (torch.rand(n, generator=self.gen) < 0.5).double()
        top, left = torch.zeros(n, dtype=torch.int64), torch.zeros(n, dtype=torch.int64)
        h, w = torch.full((n,), S, dtype=torch.int64), torch.full((n,), S, dtype=torch.int64)
        if self.crop:
            # torchvision RandomResizedCrop.get_params: ten attempts, then a central crop with the ratio clamped
            todo = torch.ones(n, dtype=torch.bool)
            log_r = (math.log(self.ratio[0]), math.log(self.ratio[1]))
            for _ in range(10):
                area = S * S * self._uniform(n, self.scale[0], self.scale[1])
                aspect = torch.exp(self._uniform(n, log_r[0], log_r[1]))
                wc = torch.round(torch.sqrt(area * aspect)).long()
                hc = torch.round(torch.sqrt(area / aspect)).long()
                ok = todo & (wc > 0) & (wc <= S) & (hc > 0) & (hc <= S)
                i = (torch.rand(n, generator=self.gen, dtype=torch.float64) * (S - hc + 1).clamp(min=1)).long()
                j = (torch.rand(n, generator=self.gen, dtype=torch.float64) * (S - wc + 1).clamp(min=1)).long()
                top[ok], left[ok], h[ok], w[ok] = i[ok], j[ok], hc[ok], wc[ok]
                todo &= ~ok
            if bool(todo.any()):
                in_ratio = 1.0
                if in_ratio < min(self.ratio):
                    wf, hf = S, int(round(S / min(self.ratio)))
                elif in_ratio > max(self.ratio):
                    hf, wf = S, int(round(S * max(self.ratio)))
                else:
                    wf, hf = S, S
                top[todo], left[todo], h[todo], w[todo] = (S - hf) // 2, (S - wf) // 2, hf, wf
        p[:, 2], p[:, 3], p[:, 4], p[:, 5] = top.double(), left.double(), h.double(), w.double()
        p[:, 6] = self._uniform(n, self.brightness, 1.0 / self.brightness) if self.brightness is not None else 1.0
        p[:, 7] = self._uniform(n, 0.0, self.noise) if self.noise is not None else 0.0
        nan_mask = torch.zeros(n, dtype=torch.int32)
        if self.nan_channels is not None:
            if self.nan_channels > C:
                raise ValueError(f"max_channels must be <= the number of channels ({self.nan_channels} > {C})")
            count = torch.randint(0, self.nan_channels + 1, (n,), generator=self.gen)
            order = torch.rand(n, C, generator=self.gen).argsort(dim=1)          # a random permutation of the channels per row
            for c in range(C):
                nan_mask |= ((order[:, c] < count).int() << c)                 # channel c is among the first `count` of the permutation
        return p.float(), nan_mask

    # -- device path ----------------------------------------------------------------------------------------------------
    def batch(self, samples, num_augmentations, params=None, nan_mask=None, noise=None):
        """samples [B, C, S, S] -> [B * (1 + A), C, S, S] on the device: sample b, then its A augmented copies."""
        x = samples.to(self.device, torch.float32).contiguous()
        B, C, S, S2 = x.shape
        assert S == S2, "square cutouts"
        A = int(num_augmentations)
        n = B * (1 + A)
        if params is None:
            params, nan_mask = self.draw(n, C, S)
        if noise is None and self.noise is not None:
            noise = torch.randn(n, C, S, S, device=self.device)
        out = torch.empty(n, C, S, S, device=self.device)
        ops.augment(x, out, params.to(self.device).contiguous(), nan_mask.to(self.device).contiguous(),
                    None if noise is None else noise.to(self.device).contiguous(), A)
        return out

    def __call__(self, img):
        single = img.dim() == 3
        x = img.unsqueeze(0) if single else img
        out = self.batch(x, 1).view(x.shape[0], 2, *x.shape[1:])[:, 1]
        out = out.to(img.device)
        return out[0] if single else out
