"""Architecture description shared by the host-side mirror of ``utils/mim_vit.py``.

Product code (no oracle imports).  State-dict names/shapes/order follow the reference module
(utils/mim_vit.py:206-283 with timm Block sub-module names) so checkpoints interchange.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, replace

import numpy as np


@dataclass(frozen=True)
class MAEConfig:
    """Constructor surface of ``MaskedAutoencoderViT`` (utils/mim_vit.py:185-189)."""
    img_size: int = 64
    patch_size: int = 16
    in_chans: int = 5
    embed_dim: int = 768
    depth: int = 12
    num_heads: int = 12
    decoder_embed_dim: int = 512
    decoder_depth: int = 8
    decoder_num_heads: int = 16
    mlp_ratio: float = 4.0
    norm_pix_loss: bool = True
    loss_fn: str = "mse"       # exact 'mse' -> MSE, anything else -> L1 (utils/mim_vit.py:502)
    pixel_mean: float = 0.0
    pixel_std: float = 1.0
    simmim: bool = False
    attn_pool: bool = False
    ra_dec: bool = False
    ln_eps: float = 1e-6

    @property
    def grid(self):
        return self.img_size // self.patch_size

    @property
    def num_patches(self):
        return self.grid * self.grid

    @property
    def patch_dim(self):
        return self.patch_size * self.patch_size * self.in_chans

    @property
    def num_extra_tokens(self):
        return 2 if self.ra_dec else 1

    @property
    def head_up(self):
        """Up-sampling factor of the SimMIM head (Conv1x1 + PixelShuffle): the patch size, or the whole image behind an
        attention pool (utils/mim_vit.py:250)."""
        return self.img_size if self.attn_pool else self.patch_size

    @property
    def head_dim(self):
        return self.head_up * self.head_up * self.in_chans


# utils/mim_vit.py:561-612 factories.  'tiny' is a build extension for the
# BASELINE.json configs[0] plumbing case: ViT-Tiny encoder (D=192, 3 heads) + MAE default decoder.
MODEL_TYPES = {
    "base": dict(depth=12, num_heads=12, decoder_embed_dim=512, decoder_depth=8, decoder_num_heads=16, simmim=False),
    "large": dict(depth=24, num_heads=16, decoder_embed_dim=512, decoder_depth=8, decoder_num_heads=16, simmim=False),
    "huge": dict(depth=32, num_heads=16, decoder_embed_dim=512, decoder_depth=8, decoder_num_heads=16, simmim=False),
    "simmim": dict(depth=12, num_heads=12, decoder_embed_dim=512, decoder_depth=8, decoder_num_heads=16, simmim=True),
    "mimlarge": dict(depth=24, num_heads=16, decoder_embed_dim=512, decoder_depth=8, decoder_num_heads=16, simmim=True),
    "mimhuge": dict(depth=32, num_heads=16, decoder_embed_dim=512, decoder_depth=8, decoder_num_heads=16, simmim=True),
    "maesimple": dict(depth=12, num_heads=12, decoder_embed_dim=512, decoder_depth=1, decoder_num_heads=1, simmim=False),
    "tiny": dict(depth=12, num_heads=3, decoder_embed_dim=512, decoder_depth=8, decoder_num_heads=16, simmim=False),
}


def config_for(model_type: str, **kw) -> MAEConfig:
    if model_type not in MODEL_TYPES:
        raise KeyError(f"unknown model_type {model_type!r}; known: {sorted(MODEL_TYPES)}")
    return replace(MAEConfig(**MODEL_TYPES[model_type]), **kw)


def sincos_pos_embed(embed_dim: int, grid_size: int, cls_token: bool = True, ra_dec: bool = False) -> np.ndarray:
    """Fixed 2-D sin-cos table of utils/pos_embed.py:20-86: float64 frequencies, first half of the
    channels encodes the column index, second half the row index, each as [sin | cos]; zero rows
    for the (ra_dec and) cls token(s)."""
    def one_d(dim, pos):
        omega = 1.0 / 10000 ** (np.arange(dim // 2, dtype=np.float64) / (dim / 2.0))
        out = np.einsum("m,d->md", pos.reshape(-1).astype(np.float64), omega)
        return np.concatenate([np.sin(out), np.cos(out)], axis=1)
    gh = np.arange(grid_size, dtype=np.float32)
    gw = np.arange(grid_size, dtype=np.float32)
    grid = np.stack(np.meshgrid(gw, gh), axis=0).reshape(2, 1, grid_size, grid_size)
    emb = np.concatenate([one_d(embed_dim // 2, grid[0]), one_d(embed_dim // 2, grid[1])], axis=1)
    n_extra = int(ra_dec) + int(cls_token)
    if n_extra:
        emb = np.concatenate([np.zeros([n_extra, embed_dim]), emb], axis=0)
    return emb


def _block(prefix, dim, hidden):
    return [
        (f"{prefix}.norm1.weight", (dim,)), (f"{prefix}.norm1.bias", (dim,)),
        (f"{prefix}.attn.qkv.weight", (3 * dim, dim)), (f"{prefix}.attn.qkv.bias", (3 * dim,)),
        (f"{prefix}.attn.proj.weight", (dim, dim)), (f"{prefix}.attn.proj.bias", (dim,)),
        (f"{prefix}.norm2.weight", (dim,)), (f"{prefix}.norm2.bias", (dim,)),
        (f"{prefix}.mlp.fc1.weight", (hidden, dim)), (f"{prefix}.mlp.fc1.bias", (hidden,)),
        (f"{prefix}.mlp.fc2.weight", (dim, hidden)), (f"{prefix}.mlp.fc2.bias", (dim,)),
    ]


SH_FEATURES, SIREN_HIDDEN = 25, 8   # LocationEncoder("siren", legendre_polys=5, dim_hidden=8, num_layers=1): mim_vit.py:211-215


def state_layout(cfg: MAEConfig):
    """Ordered (name, shape) == reference ``state_dict()`` (utils/mim_vit.py:206-283; MAE and SimMIM modes).
    SimMIM head: Conv1x1 to ``up**2 * C`` channels + PixelShuffle(up) with up = patch_size -- the reference writes
    ``tile_size`` (utils/mim_vit.py:255), which only gives an image-shaped prediction when img_size == patch_size**2,
    where the two coincide."""
    D, Dd, p, C = cfg.embed_dim, cfg.decoder_embed_dim, cfg.patch_size, cfg.in_chans
    L, E = cfg.num_patches, cfg.num_extra_tokens
    out = [("cls_token", (1, 1, D)), ("pos_embed", (1, L + E, D)), ("patch_mask_values", (C, p, p))]
    if cfg.simmim:
        out.append(("mask_token", (1, 1, 1)))
    else:
        out += [("mask_token", (1, 1, Dd)), ("decoder_pos_embed", (1, L + E, Dd))]
    out += [("patch_embed.proj.weight", (D, C, p, p)), ("patch_embed.proj.bias", (D,))]
    if cfg.ra_dec:
        out += [("ra_dec_embed.neural_network.layers.0.weight", (SIREN_HIDDEN, SH_FEATURES)),
                ("ra_dec_embed.neural_network.layers.0.bias", (SIREN_HIDDEN,)),
                ("ra_dec_embed.neural_network.last_layer.weight", (D, SIREN_HIDDEN)),
                ("ra_dec_embed.neural_network.last_layer.bias", (D,))]
    for i in range(cfg.depth):
        out += _block(f"blocks.{i}", D, int(D * cfg.mlp_ratio))
    out += [("norm.weight", (D,)), ("norm.bias", (D,))]
    if cfg.simmim:
        if cfg.attn_pool:   # timm AttentionPoolLatent (utils/mim_vit.py:246-249): latent_len 1, q / kv / proj, norm, Mlp
            hid = int(D * cfg.mlp_ratio)
            out += [("attn_pool.latent", (1, 1, D)), ("attn_pool.q.weight", (D, D)), ("attn_pool.q.bias", (D,)),
                    ("attn_pool.kv.weight", (2 * D, D)), ("attn_pool.kv.bias", (2 * D,)),
                    ("attn_pool.proj.weight", (D, D)), ("attn_pool.proj.bias", (D,)),
                    ("attn_pool.norm.weight", (D,)), ("attn_pool.norm.bias", (D,)),
                    ("attn_pool.mlp.fc1.weight", (hid, D)), ("attn_pool.mlp.fc1.bias", (hid,)),
                    ("attn_pool.mlp.fc2.weight", (D, hid)), ("attn_pool.mlp.fc2.bias", (D,))]
        out += [("decoder.0.weight", (cfg.head_dim, D, 1, 1)), ("decoder.0.bias", (cfg.head_dim,))]
        return out
    out += [("decoder_embed.weight", (Dd, D)), ("decoder_embed.bias", (Dd,))]
    for i in range(cfg.decoder_depth):
        out += _block(f"decoder_blocks.{i}", Dd, int(Dd * cfg.mlp_ratio))
    out += [("decoder_norm.weight", (Dd,)), ("decoder_norm.bias", (Dd,)),
            ("decoder_pred.weight", (cfg.patch_dim, Dd)), ("decoder_pred.bias", (cfg.patch_dim,))]
    return out


FROZEN = ("pos_embed", "decoder_pos_embed")  # requires_grad=False parameters (utils/mim_vit.py:228,273)


def not_optimised(cfg: MAEConfig):
    """State tensors the optimiser never touches: the frozen tables, and SimMIM's (1,1,1) ``mask_token``, which no
    forward uses (utils/mim_vit.py:263) -- its .grad stays None, so torch's AdamW skips it (no update, no decay)."""
    return FROZEN + (("mask_token",) if cfg.simmim else ())


def weight_decay_split(cfg: MAEConfig):
    """timm ``param_groups_weight_decay`` as called at utils/mim_vit.py:126: no decay iff
    ndim <= 1 or the name ends with '.bias'; frozen / never-used tensors are not optimised."""
    decay, no_decay = [], []
    skip = not_optimised(cfg)
    for name, shape in state_layout(cfg):
        if name in skip:
            continue
        (no_decay if (len(shape) <= 1 or name.endswith(".bias")) else decay).append(name)
    return decay, no_decay


def xavier_bound(shape):
    fan_out, fan_in = shape[0], int(np.prod(shape[1:]))
    return math.sqrt(6.0 / (fan_in + fan_out))
