#!/usr/bin/env python3
"""Generate golden vectors by RUNNING THE REFERENCE on CPU in the build container.

Usage (build container only; /root/reference does not exist on the GPU box):
    python tests/golden/make_golden.py

What runs reference code unchanged:
  * utils/similarity.py  (imports with torch only)
  * utils/pos_embed.py
  * utils/mim_vit.py     -- needs ``timm`` and ``h5py`` which are absent here; they are
    replaced by the TEST-ONLY stand-ins below.  The stand-in is this
    repo's own restatement of timm's published PatchEmbed / Block / param-group
    semantics, so the Block arithmetic in these goldens is "parity unpinned" third-party
    arithmetic; everything in mim_vit.py itself (input norm, NaN fill, masking, token
    assembly, decoder un-shuffle, patchify, loss, init, optimiser wiring) is the
    reference's own code executing.

Outputs small .npz fixtures next to this file.  Only data is written: no reference
source or bytecode is copied.
"""
import os
import sys
import types
from functools import partial

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.environ.get("SKYEMB_GOLDEN_OUT") or HERE      # the reproducibility test regenerates into a temporary directory


# --------------------------------------------------------------------------
# test-only stand-ins for absent third-party modules
# --------------------------------------------------------------------------
def install_standins():
    class PatchEmbed(nn.Module):
        def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768):
            super().__init__()
            self.img_size = (img_size, img_size)
            self.patch_size = (patch_size, patch_size)
            self.grid_size = (img_size // patch_size, img_size // patch_size)
            self.num_patches = self.grid_size[0] * self.grid_size[1]
            self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size, bias=True)

        def forward(self, x):
            return self.proj(x).flatten(2).transpose(1, 2)

    class Attention(nn.Module):
        def __init__(self, dim, num_heads, qkv_bias=True):
            super().__init__()
            self.num_heads = num_heads
            self.head_dim = dim // num_heads
            self.scale = self.head_dim ** -0.5
            self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
            self.proj = nn.Linear(dim, dim)

        def forward(self, x):
            B, N, C = x.shape
            qkv = self.qkv(x).reshape(B, N, 3, self.num_heads, self.head_dim).permute(2, 0, 3, 1, 4)
            q, k, v = qkv.unbind(0)
            attn = (q * self.scale) @ k.transpose(-2, -1)
            attn = attn.softmax(dim=-1)
            x = (attn @ v).transpose(1, 2).reshape(B, N, C)
            return self.proj(x)

    class Mlp(nn.Module):
        def __init__(self, dim, hidden):
            super().__init__()
            self.fc1 = nn.Linear(dim, hidden)
            self.act = nn.GELU()
            self.fc2 = nn.Linear(hidden, dim)

        def forward(self, x):
            return self.fc2(self.act(self.fc1(x)))

    class Block(nn.Module):
        def __init__(self, dim, num_heads, mlp_ratio=4.0, qkv_bias=False, norm_layer=nn.LayerNorm):
            super().__init__()
            self.norm1 = norm_layer(dim)
            self.attn = Attention(dim, num_heads, qkv_bias)
            self.norm2 = norm_layer(dim)
            self.mlp = Mlp(dim, int(dim * mlp_ratio))

        def forward(self, x):
            x = x + self.attn(self.norm1(x))
            return x + self.mlp(self.norm2(x))

    class AttentionPoolLatent(nn.Module):
        """timm.layers.AttentionPoolLatent restated for the arguments the reference passes (mim_vit.py:246-249):
        latent_len 1, qkv_bias, no q/k norm, no positional table, pool_type 'token'; init: trunc-normal latent (std dim^-0.5)."""
        def __init__(self, in_features, num_heads=8, mlp_ratio=4.0, norm_layer=nn.LayerNorm):
            super().__init__()
            dim = in_features
            self.num_heads, self.head_dim, self.scale = num_heads, dim // num_heads, (dim // num_heads) ** -0.5
            self.latent = nn.Parameter(torch.zeros(1, 1, dim))
            self.q = nn.Linear(dim, dim, bias=True)
            self.kv = nn.Linear(dim, dim * 2, bias=True)
            self.proj = nn.Linear(dim, dim)
            self.norm = norm_layer(dim)
            self.mlp = Mlp(dim, int(dim * mlp_ratio))
            nn.init.trunc_normal_(self.latent, std=dim ** -0.5, a=-2 * dim ** -0.5, b=2 * dim ** -0.5)

        def forward(self, x):
            B, N, C = x.shape
            q = self.q(self.latent.expand(B, -1, -1)).reshape(B, 1, self.num_heads, self.head_dim).transpose(1, 2)
            kv = self.kv(x).reshape(B, N, 2, self.num_heads, self.head_dim).permute(2, 0, 3, 1, 4)
            k, v = kv.unbind(0)
            attn = ((q * self.scale) @ k.transpose(-2, -1)).softmax(dim=-1)
            x = (attn @ v).transpose(1, 2).reshape(B, 1, C)
            x = self.proj(x)
            x = x + self.mlp(self.norm(x))
            return x[:, 0]

    def param_groups_weight_decay(model, weight_decay=1e-5, no_weight_decay_list=()):
        decay, no_decay = [], []
        for name, p in model.named_parameters():
            if not p.requires_grad:
                continue
            if p.ndim <= 1 or name.endswith(".bias") or name in no_weight_decay_list:
                no_decay.append(p)
            else:
                decay.append(p)
        return [{"params": no_decay, "weight_decay": 0.0}, {"params": decay, "weight_decay": weight_decay}]

    class VisionTransformer(nn.Module):
        """timm.models.vision_transformer.VisionTransformer restated for what utils/vit.py uses (its subclass calls
        super().__init__(**kwargs) with img_size, patch_size, in_chans, num_classes, global_pool, embed_dim, depth, num_heads,
        mlp_ratio, qkv_bias, drop_rate, norm_layer and then reads patch_embed / cls_token / blocks / norm / fc_norm / head /
        forward_head / no_weight_decay): class token, final norm unless the pooled features get their own (global_pool =
        'avg' -> fc_norm), pooling = attention pool | mean of the patch tokens | class token, dropout, linear head."""
        def __init__(self, img_size=224, patch_size=16, in_chans=3, num_classes=1000, global_pool="token", embed_dim=768, depth=12,
                     num_heads=12, mlp_ratio=4.0, qkv_bias=True, drop_rate=0.0, norm_layer=None, **unused):
            super().__init__()
            assert global_pool in ("", "avg", "token", "map")
            norm_layer = norm_layer or partial(nn.LayerNorm, eps=1e-6)
            use_fc_norm = global_pool == "avg"
            self.num_classes, self.global_pool, self.num_prefix_tokens = num_classes, global_pool, 1
            self.num_features = self.embed_dim = embed_dim
            self.patch_embed = PatchEmbed(img_size, patch_size, in_chans, embed_dim)
            self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
            self.pos_embed = nn.Parameter(torch.randn(1, self.patch_embed.num_patches + 1, embed_dim) * 0.02)
            self.pos_drop = nn.Dropout(drop_rate)
            self.blocks = nn.Sequential(*[Block(embed_dim, num_heads, mlp_ratio, qkv_bias, norm_layer) for _ in range(depth)])
            self.norm = norm_layer(embed_dim) if not use_fc_norm else nn.Identity()
            self.attn_pool = AttentionPoolLatent(embed_dim, num_heads=num_heads, mlp_ratio=mlp_ratio, norm_layer=norm_layer) if global_pool == "map" else None
            self.fc_norm = norm_layer(embed_dim) if use_fc_norm else nn.Identity()
            self.head_drop = nn.Dropout(drop_rate)
            self.head = nn.Linear(embed_dim, num_classes) if num_classes > 0 else nn.Identity()
            nn.init.normal_(self.cls_token, std=1e-6)
            for m in self.modules():
                if isinstance(m, nn.Linear):
                    nn.init.trunc_normal_(m.weight, std=0.02)
                    if m.bias is not None:
                        nn.init.zeros_(m.bias)

        def no_weight_decay(self):
            return {"pos_embed", "cls_token", "dist_token"}

        def forward_head(self, x, pre_logits=False):
            if self.attn_pool is not None:
                x = self.attn_pool(x)
            elif self.global_pool == "avg":
                x = x[:, self.num_prefix_tokens:].mean(dim=1)
            elif self.global_pool:
                x = x[:, 0]
            x = self.fc_norm(x)
            x = self.head_drop(x)
            return x if pre_logits else self.head(x)

    timm = types.ModuleType("timm")
    optim = types.ModuleType("timm.optim")
    of = types.ModuleType("timm.optim.optim_factory")
    of.param_groups_weight_decay = param_groups_weight_decay
    models = types.ModuleType("timm.models")
    vt = types.ModuleType("timm.models.vision_transformer")
    vt.PatchEmbed, vt.Block, vt.VisionTransformer = PatchEmbed, Block, VisionTransformer
    layers = types.ModuleType("timm.layers")
    layers.AttentionPoolLatent = AttentionPoolLatent
    mlayers = types.ModuleType("timm.models.layers")                       # utils/vit.py:9
    mlayers.trunc_normal_ = nn.init.trunc_normal_
    models.layers = mlayers
    timm.optim, optim.optim_factory, timm.models, models.vision_transformer, timm.layers = optim, of, models, vt, layers
    for n, m in [("timm", timm), ("timm.optim", optim), ("timm.optim.optim_factory", of), ("timm.models", models),
                 ("timm.models.vision_transformer", vt), ("timm.layers", layers), ("timm.models.layers", mlayers)]:
        sys.modules[n] = m
    sys.modules["h5py"] = types.ModuleType("h5py")  # inert: never called on this path
    return param_groups_weight_decay


def sd_np(sd):
    return {k: v.detach().cpu().numpy().copy() for k, v in sd.items()}


def mae_case(mim_vit, pgwd, name, *, img, patch, C=5, D=64, depth=2, heads=4, Dd=32, ddepth=1, dheads=4,
             norm_pix=True, loss_fn="mse", nan=False, B=4, mask_ratio=0.75, steps=0, seed=0, pixel_mean=0.1,
             pixel_std=1.3, pmv_rand=False, ra_dec=False):
    torch.manual_seed(seed)
    model = mim_vit.MaskedAutoencoderViT(img_size=img, patch_size=patch, in_chans=C, embed_dim=D, depth=depth,
                                         num_heads=heads, decoder_embed_dim=Dd, decoder_depth=ddepth,
                                         decoder_num_heads=dheads, mlp_ratio=4,
                                         norm_layer=partial(nn.LayerNorm, eps=1e-6), norm_pix_loss=norm_pix,
                                         loss_fn=loss_fn, pixel_mean=pixel_mean, pixel_std=pixel_std, ra_dec=ra_dec)
    with torch.no_grad():
        # make zero-initialised tensors non-trivial so every gradient path is exercised
        g = torch.Generator().manual_seed(seed + 100)
        for n, p in model.named_parameters():
            if n.endswith(".bias") or "norm" in n:
                p.add_(torch.randn(p.shape, generator=g) * 0.05)
        if pmv_rand:
            model.patch_mask_values.copy_(torch.randn(model.patch_mask_values.shape, generator=g) * 0.5)
    g = torch.Generator().manual_seed(seed + 1)
    x = torch.randn(B, C, img, img, generator=g).clamp_(min=-3.0)
    if nan:
        x[1, 2] = float("nan")                      # a whole missing band
        x[2, 0, 3:9, 5:20] = float("nan")           # a NaN blob crossing patch borders
        x[3, 4, ::7, ::5] = float("nan")            # scattered NaN pixels
    out = {"imgs": x.numpy().copy(), "mask_ratio": np.float64(mask_ratio),
           "cfg": np.array([img, patch, C, D, depth, heads, Dd, ddepth, dheads, int(norm_pix)], dtype=np.int64),
           "loss_fn": np.array(loss_fn), "pixel_mean": np.float64(pixel_mean), "pixel_std": np.float64(pixel_std)}
    out.update({"state/" + k: v for k, v in sd_np(model.state_dict()).items()})
    rd = None
    if ra_dec:      # MAE mode with the RA/Dec token (mim_vit.py:410-414, 446-467): sky positions in degrees
        rd = torch.stack([torch.rand(B, generator=g) * 360, torch.rand(B, generator=g) * 180 - 90], dim=1)
        out["ra_dec"] = rd.numpy().copy()

    L = (img // patch) ** 2

    def fwd(s):
        # random_masking draws torch.rand(N, L) first thing after the seed (mim_vit.py:363)
        torch.manual_seed(s)
        noise = torch.rand(B, L).numpy().copy()
        torch.manual_seed(s)
        return model(x, ra_dec=rd, mask_ratio=mask_ratio), noise

    model.train(True)
    (loss, pred, mask), out["noise"] = fwd(1000)
    torch.manual_seed(1000)
    _, _, ids_restore = model.forward_features(x, ra_dec=rd, mask_ratio=mask_ratio)
    out.update(loss=loss.detach().numpy().copy(), pred=pred.detach().numpy().copy(), mask=mask.numpy().copy(),
               ids_restore=ids_restore.numpy().copy())
    # encoder-only path (eval_fns.py:115): mask_ratio=0 keeps all tokens, shuffled
    torch.manual_seed(1000)
    latent, _, ids0 = model.forward_features(x, ra_dec=rd, mask_ratio=0, reshape_out=False)
    out.update(latent_full=latent.detach().numpy().copy(), ids_restore_full=ids0.numpy().copy())
    loss.backward()
    for n, p in model.named_parameters():
        if p.requires_grad:
            out["grad/" + n] = (p.grad if p.grad is not None else torch.zeros_like(p)).numpy().copy()
    if steps:
        # run_iter-equivalent: AdamW(param_groups_weight_decay, betas=(0.9,0.95)) + CosineAnnealingLR
        # (mim_vit.py:126-144, pretrain_fns.py:34-41)
        model.zero_grad(set_to_none=True)
        init_lr, wd, total, flf = 1e-3, 0.05, 10, 1e7
        opt = torch.optim.AdamW(pgwd(model, wd), lr=init_lr, betas=(0.9, 0.95))
        sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, total, eta_min=init_lr / flf)
        noises, losses, lrs = [], [], []
        for it in range(steps):
            lrs.append(opt.param_groups[0]["lr"])
            (loss, _, _), nz = fwd(2000 + it)
            noises.append(nz)
            loss.backward()
            opt.step()
            opt.zero_grad(set_to_none=True)
            sched.step()
            losses.append(float(loss))
            if it in (0, steps - 1):
                out.update({f"state_after{it + 1}/" + k: v for k, v in sd_np(model.state_dict()).items()
                            if k not in ("pos_embed", "decoder_pos_embed")})
        out.update(step_losses=np.array(losses), step_lrs=np.array(lrs), step_noises=np.stack(noises),
                   opt_hparams=np.array([init_lr, wd, total, flf]))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print("wrote", name, "loss", float(out["loss"]))


def simmim_case(mim_vit, pgwd, name, *, img=64, patch=8, C=5, D=64, depth=2, heads=4, norm_pix=True, loss_fn="L1",
                nan=False, ra_dec=False, B=3, steps=0, seed=0, pixel_mean=0.1, pixel_std=1.3, attn_pool=False):
    """SimMIM mode (mim_vit.py:244-264, 394-399, 431-436, 469, 480-493): per-channel pixel masks, encoder over all tokens,
    Conv1x1 + PixelShuffle head.  Geometry has img == patch**2 so that the reference's ``tile_size`` upsampling equals
    the patch size (its head is only shape-valid there)."""
    assert attn_pool or img == patch * patch
    torch.manual_seed(seed)
    model = mim_vit.MaskedAutoencoderViT(img_size=img, patch_size=patch, in_chans=C, embed_dim=D, depth=depth,
                                         num_heads=heads, mlp_ratio=4, norm_layer=partial(nn.LayerNorm, eps=1e-6),
                                         norm_pix_loss=norm_pix, simmim=True, loss_fn=loss_fn, pixel_mean=pixel_mean,
                                         pixel_std=pixel_std, ra_dec=ra_dec, attn_pool=attn_pool)
    g = torch.Generator().manual_seed(seed + 100)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if (n.endswith(".bias") or "norm" in n) and "ra_dec_embed" not in n:
                p.add_(torch.randn(p.shape, generator=g) * 0.05)
        model.patch_mask_values.copy_(torch.randn(model.patch_mask_values.shape, generator=g) * 0.5)
    g = torch.Generator().manual_seed(seed + 1)
    x = torch.randn(B, C, img, img, generator=g).clamp_(min=-3.0)
    if nan:
        x[1, 2] = float("nan")
        x[2, 0, 3:9, 5:20] = float("nan")
        x[0, 4, ::7, ::5] = float("nan")
    # per-channel patch-granular random masks (MaskGenerator semantics, dataloaders.py:197-219), seeded here
    grid = img // patch
    m = (torch.rand(B, C, grid, grid, generator=g) < 0.55).float()
    mask = m.repeat_interleave(patch, dim=2).repeat_interleave(patch, dim=3).contiguous()
    radec = torch.stack([torch.rand(B, generator=g) * 360.0, torch.rand(B, generator=g) * 180.0 - 90.0], dim=1) if ra_dec else None
    out = {"imgs": x.numpy().copy(), "pixel_mask": mask.numpy().copy(),
           "cfg": np.array([img, patch, C, D, depth, heads, int(norm_pix), int(ra_dec), int(attn_pool)], dtype=np.int64),
           "loss_fn": np.array(loss_fn), "pixel_mean": np.float64(pixel_mean), "pixel_std": np.float64(pixel_std)}
    if ra_dec:
        out["ra_dec"] = radec.numpy().copy()
    out.update({"state/" + k: v for k, v in sd_np(model.state_dict()).items()})
    model.train(True)
    loss, pred, mask_out = model(x, ra_dec=radec, mask=mask)
    latent, _, _ = model.forward_features(x, ra_dec=radec, mask=mask, reshape_out=False)
    out.update(loss=loss.detach().numpy().copy(), pred=pred.detach().numpy().copy(), latent=latent.detach().numpy().copy())
    if ra_dec:
        out["ra_dec_token"] = model.ra_dec_embed(radec).detach().numpy().copy()
        out["sh_features"] = model.ra_dec_embed.positional_encoder(radec).detach().numpy().copy()
    loss.backward()
    for n, p in model.named_parameters():
        if p.requires_grad:
            out["grad/" + n] = (p.grad if p.grad is not None else torch.zeros_like(p)).numpy().copy()
    if steps:
        model.zero_grad(set_to_none=True)
        init_lr, wd, total, flf = 1e-3, 0.05, 10, 1e7
        opt = torch.optim.AdamW(pgwd(model, wd), lr=init_lr, betas=(0.9, 0.95))
        sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, total, eta_min=init_lr / flf)
        losses = []
        for it in range(steps):
            loss, _, _ = model(x, ra_dec=radec, mask=mask)
            loss.backward()
            opt.step()
            opt.zero_grad(set_to_none=True)
            sched.step()
            losses.append(float(loss))
        out.update({f"state_after{steps}/" + k: v for k, v in sd_np(model.state_dict()).items() if k != "pos_embed"})
        out.update(step_losses=np.array(losses), opt_hparams=np.array([init_lr, wd, total, flf]))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print("wrote", name, "loss", float(out["loss"]))


def unit_pieces(mim_vit, pos_embed):
    out = {}
    for D in (64, 512, 768, 1024):
        for grid in (4, 8):
            for rd in (False, True):
                out[f"sincos/{D}_{grid}_{int(rd)}"] = pos_embed.get_2d_sincos_pos_embed(D, grid, cls_token=True,
                                                                                       ra_dec=rd).astype(np.float64)
    model = mim_vit.MaskedAutoencoderViT(img_size=32, patch_size=8, in_chans=3, embed_dim=16, depth=1, num_heads=2,
                                         decoder_embed_dim=16, decoder_depth=1, decoder_num_heads=2)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 3, 32, 32, generator=g)
    pt = model.patchify(x)
    out["patchify/in"] = x.numpy()
    out["patchify/out"] = pt.numpy()
    out["patchify/roundtrip"] = model.unpatchify(pt).numpy()
    xn = pt.clone()
    xn[0, 3, 5:50] = float("nan")
    xn[1, 7, ::3] = float("nan")
    mean, var = mim_vit.patch_mean_and_var(xn)
    out["pmv/in"], out["pmv/mean"], out["pmv/var"] = xn.numpy(), mean.numpy(), var.numpy()
    for L in (16, 64):
        for ratio in (0.0, 0.6, 0.75):
            torch.manual_seed(77)
            noise = torch.rand(3, L)
            torch.manual_seed(77)
            tok = torch.arange(3 * L * 2, dtype=torch.float32).reshape(3, L, 2)
            xm, mask, ids = model.random_masking(tok, ratio)
            key = f"mask/{L}_{ratio}"
            out[key + "/noise"], out[key + "/x_masked"] = noise.numpy(), xm.numpy()
            out[key + "/mask"], out[key + "/ids_restore"] = mask.numpy(), ids.numpy()
    np.savez_compressed(os.path.join(OUT, "unit_pieces.npz"), **out)
    print("wrote unit_pieces")


def similarity_cases(sim):
    out = {}
    g = torch.Generator().manual_seed(7)
    for (T, P, N) in ((130, 1, 512), (65, 16, 128), (65, 64, 64)):
        D = 96
        tgt = torch.randn(T, P, D, generator=g) * (0.5 + torch.rand(D, generator=g)) + torch.randn(D, generator=g)
        tst = torch.randn(N, P, D, generator=g) * 1.1 + 0.2
        key = f"sim/{T}_{P}_{N}"
        out[key + "/target"], out[key + "/test"] = tgt.numpy(), tst.numpy()
        avg, w = sim.determine_target_features(tgt)
        out[key + "/avg"], out[key + "/w"] = avg.numpy(), w.numpy()
        for metric in ("cosine", "MSE", "MAE"):
            for combine in ("min", "mean", "max"):
                for uw in (True, False):
                    s = sim.compute_similarity(tgt, tst, metric=metric, combine=combine, use_weights=uw)
                    out[f"{key}/{metric}_{combine}_{int(uw)}"] = s.numpy()
    # streaming update_best_scores (similarity.py:18-35) over 8 batches; RA/Dec column 0 carries the sample index
    N, B, n_save = 512, 64, 50
    scores = torch.randn(N, generator=g)
    out["stream/scores"] = scores.numpy()
    for metric in ("cosine", "MSE"):
        best_s = torch.full((n_save,), float("-inf") if metric == "cosine" else float("inf"))
        best_rd = torch.empty((n_save, 2))
        best_x = torch.empty((n_save, 1))
        for b in range(N // B):
            idx = torch.arange(b * B, (b + 1) * B, dtype=torch.float32)
            rd = torch.stack([idx, idx], dim=1)
            best_x, best_rd, best_s = sim.update_best_scores(idx[:, None], rd, scores[b * B:(b + 1) * B], best_x,
                                                             best_rd, best_s, n_save, metric)
        out[f"stream/{metric}_best_scores"] = best_s.numpy()
        out[f"stream/{metric}_best_idx"] = best_rd[:, 0].numpy().astype(np.int64)
    # first-batch standardisation (similarity.py:98-102)
    lat = torch.randn(32, 4, 96, generator=g) * 3 + 1
    mu, sd = lat.mean(dim=(0, 1)), lat.std(dim=(0, 1), unbiased=True)
    out["std/in"], out["std/mu"], out["std/sd"] = lat.numpy(), mu.numpy(), sd.numpy()
    out["std/out"] = ((lat - mu) / (sd + 1e-8)).numpy()
    np.savez_compressed(os.path.join(OUT, "similarity.npz"), **out)
    print("wrote similarity")


class TokenStub(nn.Module):
    """Stand-in for the encoder behind mae_simsearch (the driver only needs forward_features and
    num_extra_tokens): cls row = mean of the patch rows, patch rows = fixed linear map of 4x4 pixel blocks."""
    num_extra_tokens = 1

    def __init__(self, W):
        super().__init__()
        self.W = W

    def forward_features(self, x, ra_dec=None, mask_ratio=0, mask=None, reshape_out=False):
        B, C, H, Wd = x.shape
        p = x.reshape(B, C, H // 4, 4, Wd // 4, 4).permute(0, 2, 4, 1, 3, 5).reshape(B, (H // 4) * (Wd // 4), C * 16)
        tok = p @ self.W.to(p.device)
        return torch.cat((tok.mean(dim=1, keepdim=True), tok), dim=1), None, None


def simsearch_cases(sim):
    """utils/similarity.py:37-132 (mae_simsearch) executed end to end around the stub encoder: flat and tile-nested
    loaders, every token-selection mode, cosine and MSE.  RA column carries the sample index."""
    out = {}
    g = torch.Generator().manual_seed(11)
    N, B, C, S, D, n_save = 160, 16, 2, 8, 24, 12
    W = torch.randn(C * 16, D, generator=g) / 4
    x = torch.randn(N, C, S, S, generator=g)
    x[40:48] = x[3:11] * 1.01 + 0.01          # near-duplicates of target-like rows: a non-trivial ranking
    rd = torch.stack([torch.arange(N, dtype=torch.float32), torch.rand(N, generator=g)], dim=1)
    stub = TokenStub(W)
    tgt = stub.forward_features(x[3:11] + 0.05 * torch.randn(8, C, S, S, generator=g))[0]
    out["ss/W"], out["ss/x"], out["ss/ra_dec"], out["ss/target_latent"] = W.numpy(), x.numpy(), rd.numpy(), tgt.numpy()
    flat = [(x[i:i + B], torch.zeros(B), rd[i:i + B]) for i in range(0, N, B)]
    tiles = [([[x[i:i + B], x[i + B:i + 2 * B]]], [[torch.zeros(B), torch.zeros(B)]], [[rd[i:i + B], rd[i + B:i + 2 * B]]])
             for i in range(0, N, 2 * B)]
    cases = [("cos_min", dict(metric="cosine", combine="min")), ("cos_mean_nw", dict(metric="cosine", combine="mean", use_weights=False)),
             ("cos_max_pool", dict(metric="cosine", combine="min", max_pool=True)), ("cos_cls", dict(metric="cosine", combine="max", cls_token=True)),
             ("mse_mean", dict(metric="MSE", combine="mean")), ("mae_min_nb3", dict(metric="MAE", combine="min", n_batches=3))]
    import contextlib, io
    for name, kw in cases:
        for nested, loader in ((False, flat), (True, tiles)):
            with contextlib.redirect_stdout(io.StringIO()):
                bs, bl, brd, bsc = sim.mae_simsearch(stub, tgt, loader, torch.device("cpu"), nested_batches=nested, n_save=n_save,
                                                     verbose=1000, **kw)
            key = f"ss/{name}/{'tiles' if nested else 'flat'}"
            out[key + "/scores"], out[key + "/idx"] = bsc.numpy(), brd[:, 0].numpy().astype(np.int64)
            out[key + "/latent"] = bl.numpy()
            assert torch.equal(bs, x[brd[:, 0].long()])
    np.savez_compressed(os.path.join(OUT, "simsearch_driver.npz"), **out)
    print("wrote simsearch_driver")



def central_cases(sim):
    """utils/similarity.py:238-240 with n_central_patches: the reference calls utils.misc.select_centre without importing
    it (NameError).  The golden is made by binding the REFERENCE's own utils.misc.select_centre into the module's namespace
    -- the import the file lacks -- and running compute_similarity unchanged."""
    import importlib
    misc = importlib.import_module("utils.misc")
    assert misc.__file__.startswith(REF), misc.__file__
    sim.select_centre = misc.select_centre
    out = {}
    g = torch.Generator().manual_seed(23)
    for (T, L, N, D, n) in ((9, 16, 24, 32, 4), (5, 64, 12, 16, 16), (7, 64, 12, 16, 4)):
        tgt = torch.randn(T, L, D, generator=g) * 2 + 0.5
        tst = torch.randn(N, L, D, generator=g)
        key = f"central/{T}_{L}_{N}_{n}"
        out[key + "/target"], out[key + "/test"] = tgt.numpy(), tst.numpy()
        for metric in ("cosine", "MSE", "MAE"):
            for combine in ("min", "mean", "max"):
                out[f"{key}/{metric}_{combine}"] = sim.compute_similarity(tgt, tst, metric=metric, combine=combine, use_weights=True,
                                                                          n_central_patches=n).numpy()
    del sim.select_centre
    np.savez_compressed(os.path.join(OUT, "similarity_central.npz"), **out)
    print("wrote similarity_central")


def maskgen_cases():
    """utils/dataloaders.py:155-219 (MaskGenerator) executed as is.  The module imports h5py, torchvision and astropy at the
    top (all absent here, none touched by MaskGenerator): inert module objects stand in for the import statements only.
    For each seed the first torch.rand(1) after the seed is the generator's ratio draw: the golden pins
    count = ceil(L * ratio) per channel, the per-channel independence and the pixel up-sampling; the subsets themselves
    follow torch.randperm's stream, which no device kernel reproduces."""
    for name in ("torchvision", "torchvision.transforms", "torchvision.transforms.v2", "astropy", "astropy.io", "astropy.io.fits",
                 "astropy.wcs"):
        m = types.ModuleType(name)
        m.disable_beta_transforms_warning = lambda: None
        m.v2 = m.fits = m.WCS = None
        sys.modules.setdefault(name, m)
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    sys.modules["astropy"].io = sys.modules["astropy.io"]
    import importlib
    dl = importlib.import_module("utils.dataloaders")
    assert dl.__file__.startswith(REF), dl.__file__
    out = {}
    for (size, p, C, mx) in ((64, 8, 5, 0.9), (128, 16, 5, 0.6), (64, 16, 9, 0.9)):
        gen = dl.MaskGenerator(input_size=size, patch_size=p, max_mask_ratio=mx, num_mask_chans=C)
        us, masks = [], []
        for seed in range(40):
            torch.manual_seed(seed)
            us.append(float(torch.rand(1)))
            torch.manual_seed(seed)
            masks.append(gen().numpy().astype(np.uint8))
        key = f"mg/{size}_{p}_{C}_{mx}"
        out[key + "/u"], out[key + "/masks"] = np.array(us, dtype=np.float32), np.stack(masks)
    np.savez_compressed(os.path.join(OUT, "maskgen.npz"), **out)
    print("wrote maskgen")


def predictor_cases():
    """Downstream predictor (utils/vit.py:258-393 VisionTransformer, :134-172 optimisers, utils/predictor_training_fns.py:3-61
    run_iter, utils/pos_embed.py:122-144 interpolate_pos_embed via utils/vit.py:198-256 load_model) through the timm stand-in:
    forward logits and THREE optimiser steps for
      lp_token_ce   linear probe (norm + head trained), class-token pooling, cross-entropy
      ft_avg_mse    fine-tuning with layer-wise lr decay (param_groups_lrd as utils/vit.py:141-143 calls it), mean pooling +
                    fc_norm, MSE on normalised labels, NaN pixels in the input
      fs_token_mse  "fully supervised" branch: timm's weight-decay split, one lr
      lp_map_ce     attentive probe (the shipped cls_ap_*.ini): the two-head AttentionPoolLatent + norm + head trained, encoder frozen
      ft_map_mse    fine-tuning through the attention pool (z_ft_2.ini)
      lp_map_ce_oc, ft_map_mse_oc   the same two with the optimiser EXACTLY as utils/vit.py:174-186 leaves it: a OneCycleLR is constructed
                    first (its constructor rewrites every group's lr / initial_lr to max_lr / 25 and beta1 to 0.95) and then
                    replaced by the LinearLR
    plus the checkpoint surgery of load_model on an MAE checkpoint of another image size (bicubic pos_embed interpolation)."""
    import importlib
    vit = importlib.import_module("utils.vit")
    ptf = importlib.import_module("utils.predictor_training_fns")
    lrd = importlib.import_module("lr_decay")
    assert vit.__file__.startswith(REF) and ptf.__file__.startswith(REF)
    out = {}
    img, patch, C, D, depth, heads = 32, 8, 5, 32, 2, 2
    for case, method, pool, loss_fn, ncls in (("lp_token_ce", "lp", "token", "crossentropy", 3), ("ft_avg_mse", "ft", "avg", "mse", 2),
                                              ("fs_token_mse", "fs", "token", "mse", 1), ("lp_map_ce", "lp", "map", "crossentropy", 3),
                                              ("ft_map_mse", "ft", "map", "mse", 1), ("lp_map_ce_oc", "lp", "map", "crossentropy", 3),
                                              ("ft_map_mse_oc", "ft", "map", "mse", 1)):
        torch.manual_seed({"lp_token_ce": 21, "ft_avg_mse": 22, "fs_token_mse": 23, "lp_map_ce": 24, "ft_map_mse": 25, "lp_map_ce_oc": 26,
                           "ft_map_mse_oc": 27}[case])
        label_means, label_stds = ([0.5, -1.0][:ncls], [2.0, 0.5][:ncls]) if loss_fn == "mse" else ([0.0], [1.0])
        model = vit.VisionTransformer(label_means, label_stds, 0.1, 1.7, False, ra_dec=False, depth=depth, num_heads=heads, mlp_ratio=4,
                                      qkv_bias=True, norm_layer=partial(nn.LayerNorm, eps=1e-6), img_size=img, in_chans=C, embed_dim=D,
                                      patch_size=patch, num_classes=ncls, global_pool=pool, drop_rate=0.0)
        with torch.no_grad():                                   # a "pre-trained" state: nothing at its init value
            for n, prm in model.named_parameters():
                if prm.ndim == 1 and "norm" in n and n.endswith("weight"):
                    prm.copy_(1.0 + 0.1 * torch.randn_like(prm))
                elif n == "pos_embed":
                    prm.copy_(torch.randn_like(prm) * 0.05)
                else:
                    prm.copy_(torch.randn_like(prm) * (0.3 if prm.ndim == 1 else 0.08))
        model = nn.DataParallel(model)
        init_lr, weight_decay, layer_decay, total, final_lr_factor = 2e-3, 0.03, 0.7, 50, 100.0
        if method == "ft":                                      # utils/vit.py:138-143 (positional call: weight_decay lands in init_lr)
            groups, max_lr = lrd.param_groups_lrd(model.module, weight_decay, no_weight_decay_list=model.module.no_weight_decay(),
                                                  layer_decay=layer_decay)
            opt = torch.optim.AdamW(groups)
        elif method == "lp":                                    # utils/vit.py:145-160
            comps = [model.module.norm, model.module.fc_norm, model.module.head]
            if pool == "map":                                   # utils/vit.py:148-149: the attentive probe of the shipped cls_ap_* configs
                comps.append(model.module.attn_pool)
            opt = torch.optim.AdamW([{"params": m.parameters()} for m in comps], lr=init_lr, weight_decay=weight_decay)
            for prm in model.module.parameters():
                prm.requires_grad = False
            for m in comps:
                for prm in m.parameters():
                    prm.requires_grad = True
        else:                                                   # utils/vit.py:162-171 (the split sees the DataParallel wrapper's names)
            import timm.optim.optim_factory as of
            opt = torch.optim.AdamW(of.param_groups_weight_decay(model, weight_decay), lr=init_lr)
        if case.endswith("_oc"):                                # utils/vit.py:174-182, argument for argument; the object is discarded
            torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=max_lr if method == "ft" else init_lr, total_steps=int(total), pct_start=0.05,
                                                anneal_strategy='cos', cycle_momentum=True, base_momentum=0.85, max_momentum=0.95,
                                                div_factor=25.0, final_div_factor=final_lr_factor, three_phase=False)
        sched = torch.optim.lr_scheduler.LinearLR(opt, start_factor=1.0, end_factor=1 / final_lr_factor, total_iters=total)
        if case.endswith("_oc"):
            out[f"{case}/opt_groups"] = np.array([[gp["lr"], gp["initial_lr"], gp["betas"][0], gp["betas"][1], gp["weight_decay"]]
                                                  for gp in opt.param_groups])
        out[f"{case}/hyper"] = np.array([init_lr, weight_decay, layer_decay, total, final_lr_factor])
        out[f"{case}/cfg"] = np.array([img, patch, C, D, depth, heads, ncls])
        out[f"{case}/label_means"], out[f"{case}/label_stds"] = np.array(label_means, np.float32), np.array(label_stds, np.float32)
        for k, v in sd_np(model.module.state_dict()).items():
            out[f"{case}/state/{k}"] = v
        B = 4
        g = torch.Generator().manual_seed(5)
        xs = torch.randn(3, B, C, img, img, generator=g)
        if case == "ft_avg_mse":
            xs[:, 1, 2, 4:9, 3:7] = float("nan")
        labels = torch.randint(0, ncls, (3, B, 1), generator=g) if loss_fn == "crossentropy" else torch.randn(3, B, ncls, generator=g)
        out[f"{case}/x"], out[f"{case}/labels"] = xs.numpy(), labels.numpy()
        model.eval()
        with torch.no_grad():
            out[f"{case}/logits0"] = model(xs[0]).numpy()
        from collections import defaultdict
        cp = defaultdict(list)
        for it in range(3):
            model, opt, sched, cp = ptf.run_iter(model, xs[it], None, None, labels[it], opt, sched, cp, loss_fn=loss_fn, mode="train")
            if it != 1:                                          # (states after the first and the third step)
                for k, v in sd_np(model.module.state_dict()).items():
                    out[f"{case}/step{it}/{k}"] = v
        out[f"{case}/train_loss"] = np.array(cp["train_loss"])
        out[f"{case}/train_metric"] = np.array(cp["train_acc" if loss_fn == "crossentropy" else "train_mae"])
        out[f"{case}/lr_after"] = np.array([gp["lr"] for gp in opt.param_groups])
        if case.endswith("_oc"):
            # the optimiser / scheduler state dicts' STRUCTURE (what a predictor checkpoint of the reference holds)
            osd = opt.state_dict()
            out[f"{case}/opt_state_ids"] = np.array(sorted(osd["state"].keys()))
            out[f"{case}/opt_group_param_ids"] = np.array([len(gp["params"]) for gp in osd["param_groups"]])
            first = osd["state"][osd["param_groups"][0]["params"][0]]
            out[f"{case}/opt_state_keys"] = np.array(sorted(first.keys()))
            out[f"{case}/opt_state_step"] = np.array(float(first["step"]))
            out[f"{case}/sched_keys"] = np.array(sorted(sched.state_dict().keys()))
        print("wrote predictor", case, "losses", cp["train_loss"])
    # checkpoint surgery: an MAE checkpoint made at 32x32 (16 patches) loaded into a 48x48 model (36 patches)
    torch.manual_seed(31)
    src = {"pos_embed": torch.randn(1, 17, D), "cls_token": torch.randn(1, 1, D), "head.weight": torch.randn(7, D), "head.bias": torch.randn(7)}
    big = vit.VisionTransformer([0.0], [1.0], 0.0, 1.0, False, ra_dec=False, depth=1, num_heads=heads, mlp_ratio=4, qkv_bias=True,
                                norm_layer=partial(nn.LayerNorm, eps=1e-6), img_size=48, in_chans=C, embed_dim=D, patch_size=patch,
                                num_classes=2, global_pool="token", drop_rate=0.0)
    ck = {k: v.clone() for k, v in src.items()}
    pe = importlib.import_module("pos_embed")
    pe.interpolate_pos_embed(big, ck)
    out["surgery/pos_embed_in"], out["surgery/pos_embed_out"] = src["pos_embed"].numpy(), ck["pos_embed"].numpy()
    ck2 = {k: v.clone() for k, v in src.items()}
    ck2["pos_embed"] = torch.randn(1, 37, D)
    small = vit.VisionTransformer([0.0], [1.0], 0.0, 1.0, False, ra_dec=False, depth=1, num_heads=heads, mlp_ratio=4, qkv_bias=True,
                                  norm_layer=partial(nn.LayerNorm, eps=1e-6), img_size=32, in_chans=C, embed_dim=D, patch_size=patch,
                                  num_classes=2, global_pool="token", drop_rate=0.0)
    out["surgery/crop_in"] = ck2["pos_embed"].numpy().copy()
    pe.crop_pos_embed(small, ck2)
    out["surgery/crop_out"] = ck2["pos_embed"].numpy()
    np.savez_compressed(os.path.join(OUT, "predictor.npz"), **out)


def main():
    pgwd = install_standins()
    # this repo ships a drop-in ``utils`` package of the same name: keep it off the path so that the REFERENCE is imported
    repo = os.path.dirname(os.path.dirname(HERE))
    sys.path[:] = [p for p in sys.path if os.path.abspath(p or os.getcwd()) != repo]
    sys.path.insert(0, REF)
    sys.path.insert(0, os.path.join(REF, "utils"))
    import importlib
    mim_vit = importlib.import_module("utils.mim_vit")
    assert mim_vit.__file__.startswith(REF), mim_vit.__file__
    sim = importlib.import_module("utils.similarity")
    pos_embed = importlib.import_module("utils.pos_embed")
    torch.set_num_threads(4)
    only = sys.argv[1:]                      # e.g. `make_golden.py simsearch` regenerates one family
    if "simsearch" in only or not only:
        simsearch_cases(sim)
    if "central" in only or not only:
        central_cases(sim)
    if "maskgen" in only or not only:
        maskgen_cases()
    if "mae_radec" in only or not only:
        # I: MAE mode WITH the RA/Dec token (two extra tokens through encoder and decoder), three optimiser steps
        mae_case(mim_vit, pgwd, "mae_tiny_I_radec", img=64, patch=16, D=32, heads=2, Dd=32, dheads=2, norm_pix=True, loss_fn="mse",
                 nan=True, ra_dec=True, seed=12, steps=3)
    if "predictor" in only or not only:
        predictor_cases()
    if "attnpool" in only or not only:
        # J: SimMIM behind timm's AttentionPoolLatent (mim_vit.py:246-250, 426-427): one pooled token per image, head up-samples
        # it to the whole image; NaNs, RA/Dec token, three optimiser steps.  (32 x 32 cutouts keep the D x img^2 C head small.)
        simmim_case(mim_vit, pgwd, "simmim_tiny_J_attnpool", img=32, patch=8, norm_pix=True, loss_fn="L1", nan=True, ra_dec=True,
                    D=32, heads=2, seed=13, steps=3, attn_pool=True)
    if only:
        return
    unit_pieces(mim_vit, pos_embed)
    similarity_cases(sim)
    # A: BASELINE geometry (64/16, 5 bands), clean input, 3 optimiser steps
    mae_case(mim_vit, pgwd, "mae_tiny_A", img=64, patch=16, norm_pix=True, loss_fn="mse", steps=3)
    # B: NaN bands / pixels + non-zero patch_mask_values
    mae_case(mim_vit, pgwd, "mae_tiny_B_nan", img=32, patch=8, norm_pix=True, loss_fn="mse", nan=True, pmv_rand=True,
             seed=3)
    # C: no norm-pix, D: L1 (any loss_fn != 'mse')
    mae_case(mim_vit, pgwd, "mae_tiny_C_nonorm", img=32, patch=8, norm_pix=False, loss_fn="mse", seed=4)
    mae_case(mim_vit, pgwd, "mae_tiny_D_l1", img=32, patch=8, norm_pix=True, loss_fn="L1", nan=True, seed=5)
    # E: reference's usual patch size 8 on 64x64 (L=64, 17 kept) with mask_ratio 0.6
    mae_case(mim_vit, pgwd, "mae_tiny_E_p8", img=64, patch=8, D=32, Dd=16, heads=2, dheads=2, mask_ratio=0.6, seed=6)
    # F-H: SimMIM mode (64/8 geometry): L1 + norm-pix with NaNs (the shipped configs' loss), plain MSE, RA/Dec token
    simmim_case(mim_vit, pgwd, "simmim_tiny_F_l1_nan", norm_pix=True, loss_fn="L1", nan=True, seed=7, steps=3)
    simmim_case(mim_vit, pgwd, "simmim_tiny_G_mse", norm_pix=False, loss_fn="mse", seed=8)
    simmim_case(mim_vit, pgwd, "simmim_tiny_H_radec", norm_pix=True, loss_fn="L1", nan=True, ra_dec=True, D=32, heads=2,
                seed=9, steps=3)


if __name__ == "__main__":
    main()
