"""CPU restatement (torch fp32, no timm) of the reference MIM/MAE pretraining path.

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.

Every function cites the reference lines it restates (paths relative to
``/root/reference``).  The model is expressed functionally over a flat
``state`` dict whose keys/shapes are the reference checkpoint's state-dict
names, so goldens captured from the reference load directly.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from dataclasses import dataclass, replace
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------
# configuration (utils/mim_vit.py:185-189 constructor args + :561-612 factories)
# --------------------------------------------------------------------------
@dataclass(frozen=True)
class MAEConfig:
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
    loss_fn: str = "mse"          # exact lowercase 'mse' => MSE, anything else L1 (mim_vit.py:502)
    pixel_mean: float = 0.0
    pixel_std: float = 1.0
    simmim: bool = False
    ra_dec: bool = False          # RA/Dec token from the LocationEncoder (mim_vit.py:209-216, location_encoder.py)
    attn_pool: bool = False       # SimMIM only: timm AttentionPoolLatent after the blocks, head up-samples to the image (mim_vit.py:246-250)
    ln_eps: float = 1e-6          # partial(nn.LayerNorm, eps=1e-6), mim_vit.py:565

    @property
    def grid(self) -> int:
        return self.img_size // self.patch_size

    @property
    def num_patches(self) -> int:
        return self.grid * self.grid

    @property
    def patch_dim(self) -> int:
        return self.patch_size * self.patch_size * self.in_chans

    @property
    def num_extra_tokens(self) -> int:
        return 2 if self.ra_dec else 1


# model_type -> (depth, heads, dec_dim, dec_depth, dec_heads, simmim); mim_vit.py:561-612.
# 'tiny' is a build extension: ViT-Tiny encoder with the MAE default decoder.
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
    return replace(MAEConfig(**MODEL_TYPES[model_type]), **kw)


# --------------------------------------------------------------------------
# fixed 2-D sin-cos table (utils/pos_embed.py:20-86)
# --------------------------------------------------------------------------
def _sincos_1d(embed_dim: int, pos: np.ndarray) -> np.ndarray:
    # pos_embed.py:66-86: omega in float64, [sin | cos] halves
    omega = np.arange(embed_dim // 2, dtype=np.float64)
    omega /= embed_dim / 2.0
    omega = 1.0 / 10000 ** omega
    out = np.einsum("m,d->md", pos.reshape(-1).astype(np.float64), omega)
    return np.concatenate([np.sin(out), np.cos(out)], axis=1)


def sincos_pos_embed(embed_dim: int, grid_size: int, cls_token: bool = True, ra_dec: bool = False) -> np.ndarray:
    """pos_embed.py:20-39 -- meshgrid(w, h): first half of the channels encodes
    the w (column) index, second half the h (row) index; zero rows prepended
    for the ra_dec and cls tokens."""
    gh = np.arange(grid_size, dtype=np.float32)
    gw = np.arange(grid_size, dtype=np.float32)
    grid = np.stack(np.meshgrid(gw, gh), axis=0).reshape(2, 1, grid_size, grid_size)
    emb = np.concatenate([_sincos_1d(embed_dim // 2, grid[0]), _sincos_1d(embed_dim // 2, grid[1])], axis=1)
    if ra_dec:
        emb = np.concatenate([np.zeros([1, embed_dim]), emb], axis=0)
    if cls_token:
        emb = np.concatenate([np.zeros([1, embed_dim]), emb], axis=0)
    return emb


# --------------------------------------------------------------------------
# parameters
# --------------------------------------------------------------------------
def _xavier(shape, gen):
    fan_out, fan_in = shape[0], int(np.prod(shape[1:]))
    a = math.sqrt(6.0 / (fan_in + fan_out))
    return (torch.rand(shape, generator=gen, dtype=torch.float32) * 2 - 1) * a


def _block_names(prefix: str, dim: int, hidden: int):
    return [
        (f"{prefix}.norm1.weight", (dim,)), (f"{prefix}.norm1.bias", (dim,)),
        (f"{prefix}.attn.qkv.weight", (3 * dim, dim)), (f"{prefix}.attn.qkv.bias", (3 * dim,)),
        (f"{prefix}.attn.proj.weight", (dim, dim)), (f"{prefix}.attn.proj.bias", (dim,)),
        (f"{prefix}.norm2.weight", (dim,)), (f"{prefix}.norm2.bias", (dim,)),
        (f"{prefix}.mlp.fc1.weight", (hidden, dim)), (f"{prefix}.mlp.fc1.bias", (hidden,)),
        (f"{prefix}.mlp.fc2.weight", (dim, hidden)), (f"{prefix}.mlp.fc2.bias", (dim,)),
    ]


def state_layout(cfg: MAEConfig):
    """Ordered (name, shape) list == the reference module's state_dict order
    (mim_vit.py:206-283; timm Block sub-module names)."""
    D, Dd, p, C = cfg.embed_dim, cfg.decoder_embed_dim, cfg.patch_size, cfg.in_chans
    L, E = cfg.num_patches, cfg.num_extra_tokens
    out = [("cls_token", (1, 1, D)), ("pos_embed", (1, L + E, D)), ("patch_mask_values", (C, p, p))]
    if cfg.simmim:
        out.append(("mask_token", (1, 1, 1)))
    else:
        out += [("mask_token", (1, 1, Dd)), ("decoder_pos_embed", (1, L + E, Dd))]
    out += [("patch_embed.proj.weight", (D, C, p, p)), ("patch_embed.proj.bias", (D,))]
    if cfg.ra_dec:   # LocationEncoder("siren", legendre_polys=5, dim_hidden=8, num_layers=1, num_classes=D), mim_vit.py:211-215
        out += [("ra_dec_embed.neural_network.layers.0.weight", (SIREN_HIDDEN, SH_FEATURES)),
                ("ra_dec_embed.neural_network.layers.0.bias", (SIREN_HIDDEN,)),
                ("ra_dec_embed.neural_network.last_layer.weight", (D, SIREN_HIDDEN)),
                ("ra_dec_embed.neural_network.last_layer.bias", (D,))]
    for i in range(cfg.depth):
        out += _block_names(f"blocks.{i}", D, int(D * cfg.mlp_ratio))
    out += [("norm.weight", (D,)), ("norm.bias", (D,))]
    if cfg.simmim:
        up = cfg.patch_size  # build deviation from mim_vit.py:255 (tile_size); identical when H == p*p
        if cfg.attn_pool:    # mim_vit.py:246-250: one pooled token per image, the head up-samples it to the whole image
            hid = int(D * cfg.mlp_ratio)
            out += [("attn_pool.latent", (1, 1, D)), ("attn_pool.q.weight", (D, D)), ("attn_pool.q.bias", (D,)),
                    ("attn_pool.kv.weight", (2 * D, D)), ("attn_pool.kv.bias", (2 * D,)),
                    ("attn_pool.proj.weight", (D, D)), ("attn_pool.proj.bias", (D,)),
                    ("attn_pool.norm.weight", (D,)), ("attn_pool.norm.bias", (D,)),
                    ("attn_pool.mlp.fc1.weight", (hid, D)), ("attn_pool.mlp.fc1.bias", (hid,)),
                    ("attn_pool.mlp.fc2.weight", (D, hid)), ("attn_pool.mlp.fc2.bias", (D,))]
            up = cfg.img_size
        out += [("decoder.0.weight", (up * up * C, D, 1, 1)), ("decoder.0.bias", (up * up * C,))]
    else:
        out += [("decoder_embed.weight", (Dd, D)), ("decoder_embed.bias", (Dd,))]
        for i in range(cfg.decoder_depth):
            out += _block_names(f"decoder_blocks.{i}", Dd, int(Dd * cfg.mlp_ratio))
        out += [("decoder_norm.weight", (Dd,)), ("decoder_norm.bias", (Dd,)),
                ("decoder_pred.weight", (cfg.patch_dim, Dd)), ("decoder_pred.bias", (cfg.patch_dim,))]
    return out


FROZEN = ("pos_embed", "decoder_pos_embed")  # requires_grad=False, mim_vit.py:228,273


def init_state(cfg: MAEConfig, seed: int = 0) -> "OrderedDict[str, torch.Tensor]":
    """mim_vit.py:290-324 initialisation *distribution* (sincos tables, xavier on
    every linear and on the conv viewed [D,-1], N(0,.02) tokens, LN 1/0,
    patch_mask_values zeros).  The RNG stream is this oracle's own: fixtures,
    not seeds, carry reference weights."""
    gen = torch.Generator().manual_seed(seed)
    st = OrderedDict()
    for name, shape in state_layout(cfg):
        if name == "pos_embed":
            t = torch.from_numpy(sincos_pos_embed(shape[-1], cfg.grid, True, cfg.ra_dec)).float().unsqueeze(0)
        elif name == "decoder_pos_embed":
            t = torch.from_numpy(sincos_pos_embed(shape[-1], cfg.grid, True, cfg.ra_dec)).float().unsqueeze(0)
        elif name in ("cls_token", "mask_token"):
            t = torch.randn(shape, generator=gen) * 0.02
        elif name == "attn_pool.latent":       # timm AttentionPoolLatent.init_weights: trunc_normal_tf_(std = dim ** -0.5)
            t = (torch.randn(shape, generator=gen) * shape[-1] ** -0.5).clamp_(-2 * shape[-1] ** -0.5, 2 * shape[-1] ** -0.5)
        elif name.startswith("ra_dec_embed."):
            # Siren.init_ (location_encoder.py:41-49): first layer U(-1/dim_in, 1/dim_in); last layer
            # U(-sqrt(6/dim_in)/w0, +) with w0 = 1; biases drawn from the same range
            dim_in = SH_FEATURES if ".layers.0." in name else SIREN_HIDDEN
            w_std = (1.0 / dim_in) if ".layers.0." in name else math.sqrt(6.0 / dim_in)
            t = (torch.rand(shape, generator=gen) * 2 - 1) * w_std
        elif name == "patch_mask_values":
            t = torch.zeros(shape)
        elif name.endswith("norm1.weight") or name.endswith("norm2.weight") or name.endswith("norm.weight"):
            t = torch.ones(shape)
        elif name.endswith(".bias"):
            t = torch.zeros(shape)
        elif name == "decoder.0.weight":
            # nn.Conv2d default init is kaiming-uniform(a=sqrt(5)) == U(-1/sqrt(fan_in), 1/sqrt(fan_in))
            b = 1.0 / math.sqrt(shape[1])
            t = (torch.rand(shape, generator=gen) * 2 - 1) * b
        else:
            t = _xavier(shape, gen)
        st[name] = t.contiguous()
    return st


def weight_decay_split(cfg: MAEConfig):
    """timm ``param_groups_weight_decay`` semantics used at mim_vit.py:126:
    no decay iff ``p.ndim <= 1`` or name ends with '.bias'; frozen tensors are
    not optimised."""
    decay, no_decay = [], []
    for name, shape in state_layout(cfg):
        if name in FROZEN or (cfg.simmim and name == "mask_token"):
            # SimMIM never uses its (1,1,1) mask_token (mim_vit.py:263): its .grad stays None and torch's AdamW skips
            # it entirely (no update, no weight decay) -- pinned by the simmim goldens' state_after3/mask_token
            continue
        (no_decay if (len(shape) <= 1 or name.endswith(".bias")) else decay).append(name)
    return decay, no_decay


# --------------------------------------------------------------------------
# RA/Dec token: spherical harmonics (closed form) -> one Siren layer -> linear
# (utils/location_encoder.py:138-243; mim_vit.py:209-216 builds it with legendre_polys=5, dim_hidden=8, num_layers=1)
# --------------------------------------------------------------------------
SH_L = 5
SH_FEATURES = SH_L * SH_L
SIREN_HIDDEN = 8
SIREN_W0_FIRST = 30.0


def _assoc_legendre(l, m, x):  # location_encoder.py:138-155
    pmm = torch.ones_like(x)
    if m > 0:
        somx2 = torch.sqrt((1 - x) * (1 + x))
        fact = 1.0
        for _ in range(1, m + 1):
            pmm = pmm * (-fact) * somx2
            fact += 2.0
    if l == m:
        return pmm
    pmmp1 = x * (2.0 * m + 1.0) * pmm
    if l == m + 1:
        return pmmp1
    pll = torch.zeros_like(x)
    for ll in range(m + 2, l + 1):
        pll = ((2.0 * ll - 1.0) * x * pmmp1 - (ll + m - 1.0) * pmm) / (ll - m)
        pmm, pmmp1 = pmmp1, pll
    return pll


def _sh_norm(l, m):  # location_encoder.py:157-159
    return math.sqrt((2.0 * l + 1.0) * math.factorial(l - m) / (4 * math.pi * math.factorial(l + m)))


def spherical_harmonics(ra_dec):
    """location_encoder.py:161-206: phi = deg2rad(ra), theta = deg2rad(dec + 90); features ordered l = 0..4, m = -l..l."""
    phi, theta = torch.deg2rad(ra_dec[:, 0]), torch.deg2rad(ra_dec[:, 1] + 90)
    ct = torch.cos(theta)
    Y = []
    for l in range(SH_L):
        for m in range(-l, l + 1):
            if m == 0:
                y = _sh_norm(l, 0) * _assoc_legendre(l, 0, ct)
            elif m > 0:
                y = math.sqrt(2.0) * _sh_norm(l, m) * torch.cos(m * phi) * _assoc_legendre(l, m, ct)
            else:
                y = math.sqrt(2.0) * _sh_norm(l, -m) * torch.sin(-m * phi) * _assoc_legendre(l, -m, ct)
            Y.append(y)
    return torch.stack(Y, dim=-1)


def location_encoder(st, ra_dec):
    """SirenNet(dim_in=25, dim_hidden=8, num_layers=1, dim_out=D): sin(30 * (W0 sh + b0)) then a plain linear
    (the last Siren layer's activation is Identity, location_encoder.py:84-85)."""
    sh = spherical_harmonics(ra_dec)
    h = torch.sin(SIREN_W0_FIRST * F.linear(sh, st["ra_dec_embed.neural_network.layers.0.weight"],
                                            st["ra_dec_embed.neural_network.layers.0.bias"]))
    return F.linear(h, st["ra_dec_embed.neural_network.last_layer.weight"], st["ra_dec_embed.neural_network.last_layer.bias"])


# --------------------------------------------------------------------------
# forward pieces
# --------------------------------------------------------------------------
def norm_inputs(x, cfg):  # mim_vit.py:523-524
    return (x - cfg.pixel_mean) / cfg.pixel_std


def patchify(imgs, cfg):  # mim_vit.py:326-338 : (N,C,H,W) -> (N,L,p*p*C) ordered (py,px,c)
    p = cfg.patch_size
    h = w = imgs.shape[2] // p
    x = imgs.reshape(imgs.shape[0], cfg.in_chans, h, p, w, p)
    x = torch.einsum("nchpwq->nhwpqc", x)
    return x.reshape(imgs.shape[0], h * w, p * p * cfg.in_chans)


def unpatchify(x, cfg):  # mim_vit.py:340-352
    p = cfg.patch_size
    h = w = int(round(x.shape[1] ** 0.5))
    x = x.reshape(x.shape[0], h, w, p, p, cfg.in_chans)
    x = torch.einsum("nhwpqc->nchpwq", x)
    return x.reshape(x.shape[0], cfg.in_chans, h * p, h * p)


def patch_mean_and_var(t):  # mim_vit.py:614-627 : NaN-aware mean and *biased* variance
    ok = ~torch.isnan(t)
    cnt = ok.sum(dim=-1, keepdim=True)
    mean = torch.where(ok, t, torch.zeros((), dtype=t.dtype)).sum(dim=-1, keepdim=True) / cnt
    d2 = torch.where(ok, t - mean, torch.zeros((), dtype=t.dtype)) ** 2
    var = d2.sum(dim=-1, keepdim=True) / cnt
    return mean, var


def random_masking_from_noise(x, mask_ratio, noise):
    """mim_vit.py:354-379 with the noise supplied (SURVEY §7 'RNG parity').
    Ties in ``noise`` are broken by lower index (stable argsort): the build's
    documented contract; torch.argsort's own tie order is unspecified."""
    N, L, D = x.shape
    len_keep = int(L * (1 - mask_ratio))
    ids_shuffle = torch.argsort(noise, dim=1, stable=True)
    ids_restore = torch.argsort(ids_shuffle, dim=1, stable=True)
    ids_keep = ids_shuffle[:, :len_keep]
    x_masked = torch.gather(x, 1, ids_keep.unsqueeze(-1).repeat(1, 1, D))
    mask = torch.ones(N, L, dtype=x.dtype)
    mask[:, :len_keep] = 0
    mask = torch.gather(mask, 1, ids_restore)
    return x_masked, mask, ids_restore


def layer_norm(x, w, b, eps):
    return F.layer_norm(x, (x.shape[-1],), w, b, eps)


# TEST-ONLY operand-rounding hook (tools/operand_rounding_study.py, tests/test_oracle_golden.py): when set, every
# contraction of the path -- the linear layers, the patch-embedding conv (as the GEMM the HIP path runs), the two attention
# products -- goes through ``OPERAND_HOOK(a, b, kind)`` = a @ b with kind in {"aw", "aa"} (activation x weight-transposed /
# activation x activation), so that the effect of a GEMM operand format (bf16, fp16, split hi + lo) on loss / pred / gradients
# can be measured against the unrounded fp32 restatement.  None (the default) leaves every line below as the plain torch op.
OPERAND_HOOK = None


def linear(x, w, b=None):
    if OPERAND_HOOK is None:
        return F.linear(x, w, b)
    y = OPERAND_HOOK(x, w.t(), "aw")
    return y if b is None else y + b


def _amm(a, b):
    return a @ b if OPERAND_HOOK is None else OPERAND_HOOK(a, b, "aa")


def block(x, st, prefix, num_heads, eps):
    """timm ``Block`` as the reference instantiates it (mim_vit.py:231-233):
    pre-LN, qkv bias, softmax(q k^T * hd^-0.5) v, exact-erf GELU MLP, no
    dropout / drop-path / LayerScale / qk-norm (SURVEY §8c stand-in contract).
    THIRD-PARTY ARITHMETIC, parity unpinned (timm, no pinned version); checked against an independent implementation of the
    same layer (transformers' ViTLayer) in tests/test_oracle_golden.py."""
    B, N, D = x.shape
    hd = D // num_heads
    h = layer_norm(x, st[f"{prefix}.norm1.weight"], st[f"{prefix}.norm1.bias"], eps)
    qkv = linear(h, st[f"{prefix}.attn.qkv.weight"], st[f"{prefix}.attn.qkv.bias"])
    qkv = qkv.reshape(B, N, 3, num_heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    att = _amm(q * hd ** -0.5, k.transpose(-2, -1))
    att = att.softmax(dim=-1)
    o = _amm(att, v).transpose(1, 2).reshape(B, N, D)
    x = x + linear(o, st[f"{prefix}.attn.proj.weight"], st[f"{prefix}.attn.proj.bias"])
    h = layer_norm(x, st[f"{prefix}.norm2.weight"], st[f"{prefix}.norm2.bias"], eps)
    h = F.gelu(linear(h, st[f"{prefix}.mlp.fc1.weight"], st[f"{prefix}.mlp.fc1.bias"]))
    x = x + linear(h, st[f"{prefix}.mlp.fc2.weight"], st[f"{prefix}.mlp.fc2.bias"])
    return x


def attention_pool_latent(x, st, num_heads, eps, prefix="attn_pool"):
    """timm.layers.AttentionPoolLatent as the reference builds it (mim_vit.py:246-249: latent_len = 1, qkv_bias, no q/k norm,
    no positional table, pool_type 'token'): ONE learned query attends over all tokens; then x + Mlp(LayerNorm(x)).
    x [B, N, D] -> [B, D].  (timm is absent: restated from its published forward, like the Block.)"""
    B, N, D = x.shape
    hd = D // num_heads
    q = F.linear(st[f"{prefix}.latent"].expand(B, -1, -1), st[f"{prefix}.q.weight"], st[f"{prefix}.q.bias"])
    q = q.reshape(B, 1, num_heads, hd).transpose(1, 2)                                  # [B, H, 1, hd]
    kv = F.linear(x, st[f"{prefix}.kv.weight"], st[f"{prefix}.kv.bias"]).reshape(B, N, 2, num_heads, hd).permute(2, 0, 3, 1, 4)
    k, v = kv.unbind(0)                                                                 # [B, H, N, hd]
    att = ((q * hd ** -0.5) @ k.transpose(-2, -1)).softmax(dim=-1)
    o = (att @ v).transpose(1, 2).reshape(B, 1, D)
    o = F.linear(o, st[f"{prefix}.proj.weight"], st[f"{prefix}.proj.bias"])
    h = layer_norm(o, st[f"{prefix}.norm.weight"], st[f"{prefix}.norm.bias"], eps)
    h = F.linear(F.gelu(F.linear(h, st[f"{prefix}.mlp.fc1.weight"], st[f"{prefix}.mlp.fc1.bias"])),
                 st[f"{prefix}.mlp.fc2.weight"], st[f"{prefix}.mlp.fc2.bias"])
    return (o + h)[:, 0]


def forward_features(st, x, cfg: MAEConfig, mask_ratio=0.0, noise=None, mask=None, reshape_out=True, ra_dec=None):
    """mim_vit.py:381-438."""
    B = x.shape[0]
    E = cfg.num_extra_tokens
    x = norm_inputs(x, cfg)
    pmv = st["patch_mask_values"].repeat(1, cfg.grid, cfg.grid).expand(B, -1, -1, -1)
    x = torch.where(torch.isnan(x), pmv, x)
    ids_restore = None
    if cfg.simmim and mask is not None:
        x = x * (1 - mask) + pmv * mask
    # timm PatchEmbed == Conv2d(k=s=p) -> flatten(2).transpose(1,2)  (mim_vit.py:206,402)
    if OPERAND_HOOK is None:
        x = F.conv2d(x, st["patch_embed.proj.weight"], st["patch_embed.proj.bias"], stride=cfg.patch_size)
        x = x.flatten(2).transpose(1, 2)
    else:   # the same contraction as the [B L, C p p] x [C p p, D] GEMM the HIP path runs
        p_, g_ = cfg.patch_size, cfg.grid
        rows = x.reshape(B, cfg.in_chans, g_, p_, g_, p_).permute(0, 2, 4, 1, 3, 5).reshape(B, g_ * g_, -1)
        x = linear(rows, st["patch_embed.proj.weight"].reshape(cfg.embed_dim, -1), st["patch_embed.proj.bias"])
    x = x + st["pos_embed"][:, E:, :]
    if not cfg.simmim:
        if noise is None:
            noise = torch.rand(B, x.shape[1])
        x, mask, ids_restore = random_masking_from_noise(x, mask_ratio, noise)
    if cfg.ra_dec:
        tok = location_encoder(st, ra_dec) + st["pos_embed"][:, 1]          # mim_vit.py:410-414
        x = torch.cat((tok.unsqueeze(1), x), dim=1)
    cls = (st["cls_token"] + st["pos_embed"][:, :1, :]).expand(B, -1, -1)
    x = torch.cat((cls, x), dim=1)
    for i in range(cfg.depth):
        x = block(x, st, f"blocks.{i}", cfg.num_heads, cfg.ln_eps)
    if cfg.simmim and cfg.attn_pool:
        x = attention_pool_latent(x, st, cfg.num_heads, cfg.ln_eps).unsqueeze(1)      # mim_vit.py:426-427
    x = layer_norm(x, st["norm.weight"], st["norm.bias"], cfg.ln_eps)
    if cfg.simmim and reshape_out:
        if not cfg.attn_pool:
            x = x[:, E:]
        Bb, L, C = x.shape
        H = W = int(L ** 0.5)
        x = x.permute(0, 2, 1).reshape(Bb, C, H, W)
    return x, mask, ids_restore


def forward_decoder(st, x, ids_restore, cfg: MAEConfig):
    """mim_vit.py:440-471."""
    E = cfg.num_extra_tokens
    if cfg.simmim:
        x = F.conv2d(x, st["decoder.0.weight"], st["decoder.0.bias"])
        # reference: tile_size (== patch_size when H == p*p), img_size behind an attention pool (mim_vit.py:250)
        return F.pixel_shuffle(x, cfg.img_size if cfg.attn_pool else cfg.patch_size)
    x = linear(x, st["decoder_embed.weight"], st["decoder_embed.bias"])
    n_mask = ids_restore.shape[1] + E - x.shape[1]
    mask_tokens = st["mask_token"].repeat(x.shape[0], n_mask, 1)
    x_ = torch.cat([x[:, E:, :], mask_tokens], dim=1)
    x_ = torch.gather(x_, 1, ids_restore.unsqueeze(-1).repeat(1, 1, x.shape[2]))
    x = torch.cat([x[:, :E, :], x_], dim=1)
    x = x + st["decoder_pos_embed"]
    for i in range(cfg.decoder_depth):
        x = block(x, st, f"decoder_blocks.{i}", cfg.decoder_num_heads, cfg.ln_eps)
    x = layer_norm(x, st["decoder_norm.weight"], st["decoder_norm.bias"], cfg.ln_eps)
    x = linear(x, st["decoder_pred.weight"], st["decoder_pred.bias"])
    return x[:, E:, :]


def forward_loss(imgs, pred, mask, cfg: MAEConfig, nan_safe: bool = False):
    """mim_vit.py:473-521 (imgs already input-normalised).

    ``nan_safe=False`` is the faithful restatement: when a target element is NaN
    the reference's forward value is finite (the NaN is zeroed after the fact,
    mim_vit.py:509-515) but its BACKWARD is NaN for every parameter
    (d/dpred (t-pred)^2 = -2 (NaN) * 0; pinned by golden mae_tiny_B_nan /
    mae_tiny_D_l1, whose reference gradients are all-NaN).  ``nan_safe=True``
    gives the evidently intended gradient (NaN target elements contribute zero)
    with a bit-identical forward value; the HIP path implements that and
    DESIGN.md lists it as a deviation."""
    if cfg.simmim:
        valid = (~torch.isnan(imgs)).to(imgs.dtype)
        mask = valid * mask
        if cfg.norm_pix_loss:
            t = patchify(imgs, cfg)
            mean, var = patch_mean_and_var(t)
            t = (t - mean) / (var + 1.0e-6) ** 0.5
            imgs = unpatchify(t, cfg)
    else:
        imgs = patchify(imgs, cfg)
        if cfg.norm_pix_loss:
            mean, var = patch_mean_and_var(imgs)
            imgs = (imgs - mean) / (var + 1.0e-6) ** 0.5
    diff = imgs - pred
    bad = torch.isnan(diff)
    if nan_safe:
        diff = torch.where(bad, torch.zeros_like(diff), diff)
    if cfg.loss_fn == "mse":
        loss = diff ** 2
    else:
        loss = diff.abs()
    # mim_vit.py:509-512: exclude NaN elements from numerator and denominator
    nan_mask = torch.where(bad, 0, 1)
    if nan_mask.shape != mask.shape:
        mask = mask.unsqueeze(2)
    mask = nan_mask * mask
    loss = torch.nan_to_num(loss, nan=0.0)
    avg_scale = mask.sum() / mask.numel() * loss.numel()
    return (loss * mask).sum() / (avg_scale + 1e-5)


def forward(st, imgs, cfg: MAEConfig, mask_ratio=0.75, noise=None, mask=None, nan_safe=False, ra_dec=None):
    """mim_vit.py:552-559 -> (loss, pred, mask, ids_restore, latent)."""
    latent, mask, ids_restore = forward_features(st, imgs, cfg, mask_ratio=mask_ratio, noise=noise, mask=mask, ra_dec=ra_dec)
    pred = forward_decoder(st, latent, ids_restore, cfg)
    loss = forward_loss(norm_inputs(imgs, cfg).detach(), pred, mask, cfg, nan_safe=nan_safe)
    return loss, pred, mask, ids_restore, latent


def loss_and_grads(st, imgs, cfg, mask_ratio=0.75, noise=None, mask=None, nan_safe=False, ra_dec=None):
    """run_iter's ``loss.backward()`` (utils/pretrain_fns.py:26-34) via torch autograd on CPU."""
    leaf = OrderedDict()
    for k, v in st.items():
        leaf[k] = v.detach().clone().requires_grad_(k not in FROZEN)
    loss, pred, mask_out, ids_restore, latent = forward(leaf, imgs, cfg, mask_ratio, noise, mask, nan_safe, ra_dec)
    loss.backward()
    grads = OrderedDict((k, (v.grad if v.grad is not None else torch.zeros_like(v)))
                        for k, v in leaf.items() if k not in FROZEN)
    return loss.detach(), pred.detach(), mask_out, ids_restore, latent.detach(), grads


# --------------------------------------------------------------------------
# optimiser + schedule (mim_vit.py:119-144, pretrain_fns.py:34-41)
# --------------------------------------------------------------------------
def cosine_lr(step: int, init_lr: float, total_iters: int, final_lr_factor: float) -> float:
    """Closed form of CosineAnnealingLR(T_max=total_iters, eta_min=init_lr/final_lr_factor)
    after ``step`` scheduler steps (mim_vit.py:142-144)."""
    eta_min = init_lr / final_lr_factor
    return eta_min + (init_lr - eta_min) * (1 + math.cos(math.pi * step / total_iters)) / 2


def adamw_step(p, g, m, v, step, lr, wd, beta1=0.9, beta2=0.95, eps=1e-8):
    """One torch.optim.AdamW update in its single-tensor op order, fp32, in place.
    step is 1-based."""
    p.mul_(1 - lr * wd)
    m.mul_(beta1).add_(g, alpha=1 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-(lr / bc1))


class Trainer:
    """run_iter-equivalent loop state on CPU: AdamW(betas=(0.9,0.95)) + cosine LR."""

    def __init__(self, cfg, st, init_lr=1e-4, weight_decay=0.05, total_iters=1000, final_lr_factor=1e7):
        self.cfg, self.st = cfg, st
        self.decay, self.no_decay = weight_decay_split(cfg)
        self.m = {k: torch.zeros_like(st[k]) for k in self.decay + self.no_decay}
        self.v = {k: torch.zeros_like(st[k]) for k in self.decay + self.no_decay}
        self.t = 0
        self.init_lr, self.wd, self.total, self.flf = init_lr, weight_decay, total_iters, final_lr_factor

    def lr(self):
        return cosine_lr(self.t, self.init_lr, self.total, self.flf)

    def step(self, imgs, mask_ratio=0.75, noise=None, mask=None, ra_dec=None):
        loss, pred, mask_o, ids, latent, grads = loss_and_grads(self.st, imgs, self.cfg, mask_ratio, noise, mask,
                                                                 ra_dec=ra_dec)
        lr = self.lr()
        self.t += 1
        for k in self.decay:
            adamw_step(self.st[k], grads[k], self.m[k], self.v[k], self.t, lr, self.wd)
        for k in self.no_decay:
            adamw_step(self.st[k], grads[k], self.m[k], self.v[k], self.t, lr, 0.0)
        return loss, pred, mask_o, grads


def simmim_mask_from_noise(noise, ratio_u, max_ratio, patch_size):
    """utils/dataloaders.py:197-219 (MaskGenerator.__call__) driven by explicit uniform draws: per sample
    ratio = u * max_ratio, count = int(ceil(tensor(L * ratio))); per channel the `count` patches with the smallest noise are
    masked (== randperm(L)[:count] in distribution; ties by index); upsampled to pixels.  -> float [B, C, H, W]."""
    B, C, L = noise.shape
    grid = int(round(L ** 0.5))
    out = torch.zeros(B, C, L)
    for b in range(B):
        count = int(torch.ceil(torch.tensor(L * (float(ratio_u[b]) * max_ratio))).item())
        for c in range(C):
            order = torch.argsort(noise[b, c], stable=True)
            out[b, c, order[:count]] = 1
    out = out.view(B, C, grid, grid)
    return out.repeat_interleave(patch_size, dim=2).repeat_interleave(patch_size, dim=3).contiguous()
