"""CPU restatement of the reference's target-augmentation pipeline (utils/dataloaders.py:14-106) for GIVEN random
parameters.  TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.

torchvision is absent from the build image, so the reference's own pipeline cannot be executed here ("parity unpinned" for
this row: third-party arithmetic).  What torchvision's tensor path does for each transform is restated with the torch ops
it calls: ``hflip`` / ``vflip`` = ``Tensor.flip``; ``resized_crop`` = slice + ``torch.nn.functional.interpolate(mode=
"bilinear", align_corners=False, antialias=True)``; the brightness / noise / channel-NaN transforms are the reference's
own three-liners (utils/dataloaders.py:14-86)."""
from __future__ import annotations

import torch
import torch.nn.functional as F


def augment(imgs, params, nan_mask, noise, A):
    """imgs [B, C, S, S]; params [B * (1 + A), 8] = {flip_h, flip_v, top, left, h, w, brightness, sigma}; nan_mask [B * (1 + A)];
    noise [B * (1 + A), C, S, S] or None  ->  [B * (1 + A), C, S, S] (copy 0 of each sample unchanged)."""
    B, C, S, _ = imgs.shape
    out = torch.empty(B * (1 + A), C, S, S)
    for b in range(B):
        for a in range(1 + A):
            n = b * (1 + A) + a
            x = imgs[b].clone()
            if a == 0:
                out[n] = x
                continue
            p = params[n]
            if p[0] != 0:
                x = x.flip(-1)                                   # v2.RandomHorizontalFlip
            if p[1] != 0:
                x = x.flip(-2)                                   # v2.RandomVerticalFlip
            i, j, h, w = int(p[2]), int(p[3]), int(p[4]), int(p[5])
            x = F.interpolate(x[None, :, i:i + h, j:j + w], size=(S, S), mode="bilinear", align_corners=False, antialias=True)[0]
            x = x * p[6]                                         # RandomBrightnessAdjust
            if noise is not None:
                x = x + noise[n] * p[7]                          # RandomNoise
            for c in range(C):
                if (int(nan_mask[n]) >> c) & 1:                  # RandomChannelNaN
                    x[c] = float("nan")
            out[n] = x
    return out
