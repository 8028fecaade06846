"""utils/vit.py mirror -- API ONLY: ``build_model`` + ``forward_features`` so
that ``similarity_search.py`` can route through a downstream ViT built on a pretrained MAE encoder
(``pretained_mae`` in the ini, similarity_search.py:104-118).  The encoder is the same Block stack as
the MAE's and runs in the same HIP kernels; predictor heads, fine-tuning optimisers and layer-wise
LR decay belong to the downstream-predictor workload and are out of scope.
"""
from __future__ import annotations

import os
from collections import defaultdict

import torch

from ..engine import MAEEngine
from ..model_config import MODEL_TYPES, config_for
from .mim_vit import _DataParallelShim, _PatchEmbedInfo, _compute_dtype
from .misc import str2bool

_ENCODER_KEYS = ("cls_token", "pos_embed", "patch_mask_values", "patch_embed.", "blocks.", "norm.", "ra_dec_embed.")


class VisionTransformer:
    """utils/vit.py:258-393 front end + encoder: input norm, NaN fill, patch embed, pos embed, cls
    token (+ RA/Dec token, utils/vit.py:374-378), Blocks, final norm.  Tokens stay in raster order (no random shuffling here)."""

    def __init__(self, cfg, device, compute_dtype):
        self.cfg = cfg
        self.engine = MAEEngine(cfg, device=device, compute_dtype=compute_dtype)
        self.patch_embed = _PatchEmbedInfo(cfg)
        self.num_extra_tokens, self.attn_pool, self.simmim = cfg.num_extra_tokens, None, False
        self.in_chans, self.pixel_mean, self.pixel_std = cfg.in_chans, cfg.pixel_mean, cfg.pixel_std
        self._ramp = None

    def eval(self):
        return self

    def train(self, mode=True):
        if mode:
            raise NotImplementedError("training downstream predictors is out of scope (SURVEY.md §2 rows 13-16)")
        return self

    def load_encoder_state(self, sd):
        own = self.engine.state_dict()
        picked = {k: v for k, v in sd.items() if k in own and k.startswith(_ENCODER_KEYS)}
        missing = [k for k in own if k.startswith(_ENCODER_KEYS) and k not in picked]
        if missing:
            raise RuntimeError(f"checkpoint lacks encoder tensors, e.g. {missing[:4]}")
        self.engine.load_state_dict({**{k: v for k, v in own.items()}, **picked})

    def forward_features(self, x, ra_dec=None, mask=None, reshape_out=False):
        """utils/vit.py:344-388 -> (tokens [B, extra+L, D], None, None); extra = cls (+ the RA/Dec token)."""
        x = x.to(self.engine.device, torch.float32).contiguous()
        B, L = x.shape[0], self.cfg.num_patches
        if self._ramp is None or self._ramp.shape[0] != B:
            self._ramp = torch.arange(L, device=x.device, dtype=torch.float32).repeat(B, 1).contiguous() / L
        lat, _, _ = self.engine.forward_features(x, 0.0, self._ramp, ra_dec=ra_dec)
        lat = lat.clone()
        if reshape_out:
            lat = lat[:, self.num_extra_tokens:]
            H = W = int(L ** 0.5)
            lat = lat.permute(0, 2, 1).reshape(B, -1, H, W)
        return lat, None, None


def build_model(config, mae_config, model_filename, mae_filename, device, build_optimizer=False):
    """utils/vit.py:21-195 signature; returns (model, losses, cur_iter)."""
    if build_optimizer:
        raise NotImplementedError("predictor fine-tuning / linear probing is out of scope (SURVEY.md §2 rows 13-16)")
    arch = mae_config['ARCHITECTURE']
    model_type = arch['model_type']
    base = {"simmim": "base", "mimlarge": "large", "mimhuge": "huge"}.get(model_type, model_type)
    if base not in MODEL_TYPES:
        raise KeyError(model_type)
    cfg = config_for(base, img_size=int(config['ARCHITECTURE']['img_size']), patch_size=int(arch['patch_size']),
                     in_chans=int(arch['num_channels']), embed_dim=int(arch['embed_dim']),
                     pixel_mean=float(arch['pixel_mean']), pixel_std=float(arch['pixel_std']),
                     ra_dec=str2bool(arch.get('ra_dec', 'False')))
    model = VisionTransformer(cfg, device, _compute_dtype(mae_config))
    losses, cur_iter = defaultdict(list), 1
    for fn, is_own in ((model_filename, True), (mae_filename, False)):
        if fn and fn != 'None' and os.path.exists(fn):
            ck = torch.load(fn, map_location="cpu", weights_only=False)
            model.load_encoder_state(ck['model'])
            if is_own:
                losses, cur_iter = defaultdict(list, dict(ck['losses'])), ck['batch_iters'] + 1
            break
    return _DataParallelShim(model), losses, cur_iter
