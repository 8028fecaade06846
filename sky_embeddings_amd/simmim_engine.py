"""SimMIM mode of the MIM path (utils/mim_vit.py:244-264, 394-399, 410-414, 431-436, 469, 480-493) on the same kernels:
per-channel pixel masks blended into the input, every patch token encoded (plus the optional RA/Dec token), a linear
head per token (= Conv1x1 + PixelShuffle) and a pixel-wise masked loss.  The launch schedule reuses MAEEngine's
transformer-block forward / backward, LayerNorm batching and grouped weight-gradient launches.

Sequence layout: Ne = E + L tokens per sample, E = 1 (cls) or 2 (cls, RA/Dec); rows b*Ne + {0: cls, 1: RA/Dec, E + l: patch l}.
The head runs over all M = B*Ne rows (the E extra rows are ignored by the loss, whose gradient for them is zero): one
plain GEMM instead of a gather.

``attn_pool = True`` (utils/mim_vit.py:246-250, 426-427): after the blocks, timm's AttentionPoolLatent pools each sample's
tokens into ONE row (a learned query attends over them; then x + Mlp(LayerNorm(x))), the final norm and the head run on
[B, D] rows, and the head up-samples each row to the whole image (PixelShuffle(img_size) = the loss kernel's index map with
one "patch" the size of the image).
"""
from __future__ import annotations

import numpy as np
import torch

from . import ops
from .engine import MAEEngine
from .model_config import MAEConfig
from .ops import KC, RC


def simmim_stage_ranges(store, cfg: MAEConfig, n_encoder_groups: int = 3):
    """Backward stages [head, encoder block groups (top first) ..., embedding] -> slices of the flat gradient buffer
    that are final after each (cf. engine.stage_gradient_ranges)."""
    off = store.offsets
    # (layout order: the pool precedes the head.  Without the pool the head's weight gradient may be a problem of the last block's
    # grouped launch -- _extra_wgrad_layers -- so it belongs to the top encoder stage and the head stage finishes no decayed tensor)
    head0 = off["attn_pool.latent"] if cfg.attn_pool else store.n_decay
    assert cfg.attn_pool or off["decoder.0.weight"] + int(np.prod(store.shapes["decoder.0.weight"])) <= store.n_decay
    ranges = [[(head0, store.n_decay)] if head0 < store.n_decay else []]
    bounds = sorted({round(cfg.depth * k / n_encoder_groups) for k in range(n_encoder_groups + 1)}, reverse=True)
    groups = []
    for hi, lo in zip(bounds[:-1], bounds[1:]):
        groups.append((hi, lo))
        end = head0 if hi == cfg.depth else off[f"blocks.{hi}.attn.qkv.weight"]
        ranges.append([(off[f"blocks.{lo}.attn.qkv.weight"], end)])
    ranges.append([(0, off["blocks.0.attn.qkv.weight"]), (store.n_decay, store.n)])
    return groups, ranges


class SimMIMEngine(MAEEngine):
    _modes = "simmim"

    def __init__(self, cfg: MAEConfig, device="cuda", compute_dtype=torch.bfloat16, seed=None):
        assert cfg.simmim, "SimMIMEngine serves simmim=True configurations (MAE mode: MAEEngine)"
        super().__init__(cfg, device=device, compute_dtype=compute_dtype, seed=seed)

    # ------------------------------------------------------------------ buffers
    def _workspace(self, B, keep, train):
        key = (B, keep, train)
        if key in self._ws:
            return self._ws[key]
        cfg, dev, T = self.cfg, self.device, self.dtype
        L, D, pv, E = cfg.num_patches, cfg.embed_dim, cfg.patch_dim, cfg.num_extra_tokens
        Ne = E + L
        M = B * Ne
        f32 = dict(device=dev, dtype=torch.float32)
        lp = dict(device=dev, dtype=T)
        w = {}
        ar = torch.arange(L, device=dev)
        w["pe_dst"] = (torch.arange(B, device=dev)[:, None] * Ne + E + ar[None, :]).to(torch.int32).contiguous()
        w["pe_tab"] = ar[None, :].expand(B, L).to(torch.int32).contiguous()
        w["patches"] = torch.empty(B * L, pv, **lp)
        w["latent32"] = torch.empty(M, D, **f32)
        hidden = int(D * cfg.mlp_ratio)

        def block_bufs():
            return dict(ln1=torch.empty(M, D, **lp), mean1=torch.empty(M, **f32), rstd1=torch.empty(M, **f32),
                        qkv=torch.empty(M, 3 * D, **lp), att=torch.empty(M, D, **lp), xmid=torch.empty(M, D, **f32),
                        ln2=torch.empty(M, D, **lp), mean2=torch.empty(M, **f32), rstd2=torch.empty(M, **f32),
                        hpre=torch.empty(M, hidden, **lp), hact=torch.empty(M, hidden, **lp))

        w["enc"] = [block_bufs() for _ in range(cfg.depth if train else 1)]
        w["xs"] = [torch.empty(M, D, **f32) for _ in range((cfg.depth + 1) if train else 2)]
        w["lat_lp"] = torch.empty(M, D, **lp)
        w["lat_mean"], w["lat_rstd"] = torch.empty(M, **f32), torch.empty(M, **f32)
        w["sh"], w["z"], w["dz"] = torch.empty(B, 25, **f32), torch.empty(B, 8, **f32), torch.empty(B, 8, **f32)
        pool = cfg.attn_pool
        R = B if pool else M                     # rows the final norm and the head see
        pvh = cfg.head_dim
        if pool:
            H = cfg.num_heads
            w["ap_x"] = torch.empty(M, D, **lp)                       # block output in the compute dtype (kv projection operand)
            w["ap_kv"] = torch.empty(M, 2 * D, **lp)
            w["ap_q"] = torch.empty(D, **f32)
            w["ap_prob"] = torch.empty(B, H, Ne, **f32)
            w["ap_o"] = torch.empty(B, D, **lp)
            w["ap_y"] = torch.empty(B, D, **f32)                      # proj output (residual stream of the pool)
            w["ap_ln"] = torch.empty(B, D, **lp)
            w["ap_mean"], w["ap_rstd"] = torch.empty(B, **f32), torch.empty(B, **f32)
            w["ap_hpre"], w["ap_hact"] = torch.empty(B, hidden, **lp), torch.empty(B, hidden, **lp)
            w["ap_z"] = torch.empty(B, D, **f32)
            w["lat_lp"] = torch.empty(B, D, **lp)
            w["latent32"] = torch.empty(B, D, **f32)
            w["lat_mean"], w["lat_rstd"] = torch.empty(B, **f32), torch.empty(B, **f32)
        if train:
            w["pred_tok"] = torch.empty(R, pvh, **f32)
            w["pred_img"] = torch.empty(B, cfg.in_chans, cfg.img_size, cfg.img_size, **f32)
            w["loss"] = torch.zeros(1, **f32)
            w["loss_ws"] = torch.empty(4 * B * L + 4, **f32)
            w["dpred"] = torch.empty(R, pvh, **lp)
            if pool:
                w["ap_gz"], w["ap_gz_lp"] = torch.empty(B, D, **f32), torch.empty(B, D, **lp)
                w["ap_gy"], w["ap_gy_lp"] = torch.empty(B, D, **f32), torch.empty(B, D, **lp)
                w["ap_dln"], w["ap_dh"] = torch.empty(B, D, **lp), torch.empty(B, hidden, **lp)
                w["ap_do"] = torch.empty(B, D, **lp)
                w["ap_dkv"] = torch.empty(M, 2 * D, **lp)
                w["ap_dq"], w["ap_dq_ws"] = torch.empty(B, D, **f32), torch.empty(D, **f32)
            w["g"] = torch.empty(M * D, **f32)
            w["g_lp"] = torch.empty(M * D, **lp)
            w["g_lp2"] = torch.empty(M * D, **lp)
            w["dln"] = torch.empty(M * D, **lp)
            w["datt"] = torch.empty(M * D, **lp)
            w["dh"] = torch.empty(M * hidden, **lp)
            w["dqkv"] = torch.empty(3 * M * D, **lp)
            # second scratch set (see MAEEngine._workspace): consecutive blocks alternate between the two
            w["g_lp_b"], w["g_lp2_b"] = torch.empty(M * D, **lp), torch.empty(M * D, **lp)
            w["dh_b"], w["dqkv_b"] = torch.empty(M * hidden, **lp), torch.empty(3 * M * D, **lp)
            w["dT"] = torch.empty(B * L, D, **lp)
            w["drows"] = torch.empty(B * L, pv, **f32)
            w["pmv_part"] = torch.empty(B, pv, **f32)
            w["rs_part"] = torch.empty(256, D, **f32)
            w["splitk_ws"] = self._splitk_ws
            order = [("ln", "norm", R, D)] + ([("ln", "attn_pool.norm", B, D)] if pool else [])
            for i in reversed(range(cfg.depth)):
                order.append(("block", f"blocks.{i}", w["enc"][i], M, D))
            self._build_reduce_table(w, order)
            w["wgrad_groups"] = {}            # (built after the reduce table: every group carries its block's norm1 backward)
            if self.dtype in ops.LP_DTYPES:
                for i, bufs in enumerate(w["enc"]):
                    w["wgrad_groups"][f"blocks.{i}"] = self._make_wgrad_group(f"blocks.{i}", bufs, M, D, w)
        if train and getattr(self, "_fused_adamw", None) is not None and w.get("wgrad_groups"):
            self._build_adamw_groups(w)
        if train and getattr(self, "_g16", None) is not None and w.get("wgrad_groups"):
            self._build_variant_groups(w, "g16")
        self._ws[key] = w
        return w

    # ------------------------------------------------------------------ forward
    def _check_simmim_inputs(self, imgs, mask, ra_dec):
        cfg = self.cfg
        assert imgs.is_cuda and imgs.dtype == torch.float32 and imgs.is_contiguous()
        B, C, H, W = imgs.shape
        assert (C, H, W) == (cfg.in_chans, cfg.img_size, cfg.img_size), f"bad cutout shape {tuple(imgs.shape)}"
        if mask is not None:
            assert mask.shape == imgs.shape and mask.is_cuda, "SimMIM pixel mask must be [B,C,H,W] on the device"
            mask = mask.to(torch.float32).contiguous()
        if cfg.ra_dec:
            assert ra_dec is not None and tuple(ra_dec.shape) == (B, 2), "ra_dec=True models need ra_dec [B,2] (degrees)"
            ra_dec = ra_dec.to(device=imgs.device, dtype=torch.float32).contiguous()
        return mask, ra_dec

    def _encoder_fwd_simmim(self, imgs, mask, ra_dec, w, train):
        """utils/mim_vit.py:381-429 with simmim=True: blend, embed all L patches, (RA/Dec,) cls, blocks, norm."""
        cfg, st = self.cfg, self.store
        B = imgs.shape[0]
        L, D, pv, E = cfg.num_patches, cfg.embed_dim, cfg.patch_dim, cfg.num_extra_tokens
        Ne = E + L
        M = B * Ne
        ops.patch_gather_blend(imgs, st.param("patch_mask_values"), None, mask, w["patches"], cfg.patch_size, L,
                               cfg.pixel_mean, cfg.pixel_std)
        xs = w["xs"]
        x0 = xs[0]
        pos = st.frozen["pos_embed"].view(-1, D)
        ops.gemm(w["patches"], st.lp("patch_embed.proj.weight"), M=B * L, N=D, K=pv, bias=st.param("patch_embed.proj.bias"),
                 table=pos[E:], tab_row=w["pe_tab"], ldt=D, dst_row=w["pe_dst"], out_f32=x0, ldo32=D,
                 prefetch=self._pf("fwd", "patch_embed.proj.weight", B * L))
        x0.view(B, Ne, D)[:, 0, :] = st.param("cls_token").view(D) + pos[0]        # utils/mim_vit.py:417-419 (host glue)
        if cfg.ra_dec:
            P = st.param
            ops.radec_token_fwd(ra_dec, P("ra_dec_embed.neural_network.layers.0.weight"),
                                P("ra_dec_embed.neural_network.layers.0.bias"), P("ra_dec_embed.neural_network.last_layer.weight"),
                                P("ra_dec_embed.neural_network.last_layer.bias"), pos[1], x0.view(-1)[D:], Ne * D, B, D,
                                w["sh"], w["z"])
        for i in range(cfg.depth):
            if train:
                self._block_fwd(xs[i], xs[i + 1], w["enc"][i], f"blocks.{i}", M, D, cfg.num_heads, B, Ne)
            else:
                self._block_fwd(xs[i % 2], xs[(i + 1) % 2], w["enc"][0], f"blocks.{i}", M, D, cfg.num_heads, B, Ne)
        x_last = xs[cfg.depth] if train else xs[cfg.depth % 2]
        if cfg.attn_pool:
            self._pool_fwd(x_last, w, B, Ne)
            ops.layernorm_fwd(w["ap_z"], st.param("norm.weight"), st.param("norm.bias"), w["lat_lp"], w["lat_mean"], w["lat_rstd"],
                              B, D, cfg.ln_eps, y32=w["latent32"])
            return x_last
        ops.layernorm_fwd(x_last, st.param("norm.weight"), st.param("norm.bias"), w["lat_lp"], w["lat_mean"], w["lat_rstd"],
                          M, D, cfg.ln_eps, y32=w["latent32"])
        return x_last

    def _pool_fwd(self, x_last, w, B, Ne):
        """timm AttentionPoolLatent (utils/mim_vit.py:426-427): [B*Ne, D] fp32 block output -> w["ap_z"] [B, D] fp32."""
        cfg, st = self.cfg, self.store
        D, H = cfg.embed_dim, cfg.num_heads
        M, hidden = B * Ne, int(D * cfg.mlp_ratio)
        P, LP = st.param, st.lp
        ops.cast(x_last, w["ap_x"], M * D)
        ops.gemm(w["ap_x"], LP("attn_pool.kv.weight"), M=M, N=2 * D, K=D, bias=P("attn_pool.kv.bias"), out=w["ap_kv"])
        ops.attnpool_q(P("attn_pool.latent"), P("attn_pool.q.weight"), P("attn_pool.q.bias"), w["ap_q"])
        ops.attnpool_fwd(w["ap_q"], w["ap_kv"], w["ap_o"], w["ap_prob"], B, Ne, H, D // H)
        ops.gemm(w["ap_o"], LP("attn_pool.proj.weight"), M=B, N=D, K=D, bias=P("attn_pool.proj.bias"), out_f32=w["ap_y"])
        ops.layernorm_fwd(w["ap_y"], P("attn_pool.norm.weight"), P("attn_pool.norm.bias"), w["ap_ln"], w["ap_mean"], w["ap_rstd"],
                          B, D, cfg.ln_eps)
        ops.gemm(w["ap_ln"], LP("attn_pool.mlp.fc1.weight"), M=B, N=hidden, K=D, bias=P("attn_pool.mlp.fc1.bias"), act=ops.ACT_GELU,
                 out=w["ap_hact"], out2=w["ap_hpre"])
        ops.gemm(w["ap_hact"], LP("attn_pool.mlp.fc2.weight"), M=B, N=D, K=hidden, bias=P("attn_pool.mlp.fc2.bias"), resid=w["ap_y"],
                 ldr=D, out_f32=w["ap_z"], ws=self._splitk_ws)

    def forward_features(self, imgs, mask_ratio=0.0, noise=None, mask=None, ra_dec=None):
        """utils/mim_vit.py:381-438 (reshape_out=False): -> (latent fp32 [B, E+L, D], mask, None); tokens keep their order."""
        cfg = self.cfg
        mask, ra_dec = self._check_simmim_inputs(imgs, mask, ra_dec)
        B = imgs.shape[0]
        w = self._workspace(B, cfg.num_patches, False)
        self._encoder_fwd_simmim(imgs, mask, ra_dec, w, False)
        if cfg.attn_pool:
            return w["latent32"].view(B, 1, cfg.embed_dim), mask, None            # one pooled token per image
        return w["latent32"].view(B, cfg.num_extra_tokens + cfg.num_patches, cfg.embed_dim), mask, None

    def forward_train(self, imgs, mask=None, ra_dec=None):
        """utils/mim_vit.py:552-559 with simmim=True: -> (loss [1], pred fp32 [B,C,H,W], pixel mask)."""
        cfg, st = self.cfg, self.store
        assert mask is not None, "SimMIM training needs the per-pixel mask (MaskGenerator, utils/dataloaders.py:155-219)"
        mask, ra_dec = self._check_simmim_inputs(imgs, mask, ra_dec)
        B = imgs.shape[0]
        L, D, pv, E = cfg.num_patches, cfg.embed_dim, cfg.patch_dim, cfg.num_extra_tokens
        M = B * (E + L)
        w = self._workspace(B, L, True)
        self.plan_loss_scale(self.expected_masked_elements(B, None))
        self._encoder_fwd_simmim(imgs, mask, ra_dec, w, True)
        # head: Conv1x1 D -> p*p*C per token; PixelShuffle(p) is the loss kernel's index map (utils/mim_vit.py:254-261,469)
        pool = cfg.attn_pool         # one row per image, laid out like the image (PixelShuffle(img_size), utils/mim_vit.py:250)
        ops.gemm(w["lat_lp"], st.lp("decoder.0.weight"), M=B if pool else M, N=cfg.head_dim, K=D, bias=st.param("decoder.0.bias"),
                 out_f32=w["pred_tok"])
        ops.simmim_pixel_loss(imgs, w["pred_tok"], mask, w["loss"], w["dpred"], self.code, w["pred_img"], w["loss_ws"],
                              cfg.patch_size, 0 if pool else E, cfg.pixel_mean, cfg.pixel_std, cfg.norm_pix_loss, cfg.loss_fn != "mse",
                              pooled=pool, dscale=self.loss_scale)
        self._last = (imgs, B, L, mask)
        return w["loss"], w["pred_img"], mask

    def expected_masked_elements(self, B, mask_ratio):
        """SimMIM: the ratio is drawn per sample, U(0, max_mask_ratio) -- the scale is planned for a quarter of the pixels (any
        power of two within a few binades serves: see MAEEngine.__init__)."""
        cfg = self.cfg
        return B * cfg.in_chans * cfg.img_size * cfg.img_size // 4

    # ------------------------------------------------------------------ backward
    def _ctx(self):
        assert self._last is not None, "backward() without forward_train()"
        imgs, B, L, mask = self._last
        return imgs, B, mask, self._ws[(B, L, True)]

    def _decoder_weight_chain(self):
        # (prefetch hints, engine._pf: behind the blocks the forward chain reads the pixel head; the pooled variant's small GEMMs name nothing)
        return [] if self.cfg.attn_pool else ["decoder.0.weight"]

    def _extra_wgrad_layers(self, prefix, M, w):
        """The pixel head's weight gradient (Conv2d 1x1 = a linear over the token rows, utils/mim_vit.py:244-249) as a fifth problem
        of the last block's grouped weight-gradient launch -- the first grouped launch of backward, same token rows (engine.py
        _extra_wgrad_layers).  Not with the pooled head: its rows are the images, not the tokens."""
        import os
        cfg = self.cfg
        if (os.environ.get("SKYEMB_FOLD_WGRADS", "1") == "0" or self._side is not None or cfg.attn_pool or "dpred" not in w
                or prefix != f"blocks.{cfg.depth - 1}" or w["dpred"].shape[0] != M):
            return []
        return [(w["dpred"], w["lat_lp"], "decoder.0", w["dpred"].shape[1], cfg.embed_dim)]

    def backward_decoder(self):
        """Stage 0: the head (decoder.0) and the final norm; leaves d(block output) in g / g_lp."""
        imgs, B, mask, w = self._ctx()
        cfg = self.cfg
        D, pv = cfg.embed_dim, cfg.patch_dim
        M = B * (cfg.num_extra_tokens + cfg.num_patches)
        self._ln_first = self._ln_count = 0
        g, g_lp = w["g"][:M * D].view(M, D), w["g_lp"][:M * D].view(M, D)
        if cfg.attn_pool:
            self._pool_bwd(w, B, M // B, g, g_lp)
            self._end_stage(w)
            return
        dln = w["dln"][:M * D].view(M, D)
        self._linear_bwd(w["dpred"], w["lat_lp"], "decoder.0.weight", "decoder.0.bias", M, pv, D, w, dx_out=dln,
                         wgrad="decoder.0" not in w.get("folded_wgrads", ()))
        self._ln_bwd(dln, w["xs"][cfg.depth], "norm", w["lat_mean"], w["lat_rstd"], None, g, g_lp, M, D, w)
        self._end_stage(w)

    def _pool_bwd(self, w, B, Ne, g, g_lp):
        """Head, final norm and the attention pool, backwards; leaves d(block output) in g (fp32) / g_lp."""
        cfg, st = self.cfg, self.store
        D, H = cfg.embed_dim, cfg.num_heads
        M, hidden = B * Ne, int(D * cfg.mlp_ratio)
        P, G = st.param, st.grad
        self._linear_bwd(w["dpred"], w["lat_lp"], "decoder.0.weight", "decoder.0.bias", B, cfg.head_dim, D, w, dx_out=w["ap_dln"])
        self._ln_bwd(w["ap_dln"], w["ap_z"], "norm", w["lat_mean"], w["lat_rstd"], None, w["ap_gz"], w["ap_gz_lp"], B, D, w)
        # z = y + fc2(gelu(fc1(ln(y))))
        self._linear_bwd(w["ap_gz_lp"], w["ap_hact"], "attn_pool.mlp.fc2.weight", "attn_pool.mlp.fc2.bias", B, D, hidden, w,
                         dx_out=w["ap_dh"], dx_act=ops.ACT_DGELU, dx_aux=w["ap_hpre"])
        self._linear_bwd(w["ap_dh"], w["ap_ln"], "attn_pool.mlp.fc1.weight", "attn_pool.mlp.fc1.bias", B, hidden, D, w,
                         dx_out=w["ap_dln"])
        self._ln_bwd(w["ap_dln"], w["ap_y"], "attn_pool.norm", w["ap_mean"], w["ap_rstd"], w["ap_gz"], w["ap_gy"], w["ap_gy_lp"],
                     B, D, w)
        # y = proj(pool(q, kv(x)))
        self._linear_bwd(w["ap_gy_lp"], w["ap_o"], "attn_pool.proj.weight", "attn_pool.proj.bias", B, D, D, w, dx_out=w["ap_do"])
        ops.attnpool_bwd(w["ap_q"], w["ap_kv"], w["ap_do"], w["ap_prob"], w["ap_dkv"], w["ap_dq"], B, Ne, H, D // H)
        ops.attnpool_q_bwd(w["ap_dq"], P("attn_pool.latent"), P("attn_pool.q.weight"), G("attn_pool.q.weight"), G("attn_pool.q.bias"),
                           G("attn_pool.latent"), w["ap_dq_ws"])
        self._wgrad(w["ap_dkv"], w["ap_x"], 2 * D, D, M, G("attn_pool.kv.weight"), G("attn_pool.kv.bias"), w)
        ops.gemm(w["ap_dkv"], st.lp("attn_pool.kv.weight"), M=M, N=D, K=2 * D, a_layout=KC, b_layout=RC, lda=2 * D, ldb=D,
                 out_f32=g, ldo32=D, out=g_lp, ws=w["splitk_ws"])

    def backward_encoder(self, hi=None, lo=0):
        imgs, B, mask, w = self._ctx()
        cfg = self.cfg
        D, Ne = cfg.embed_dim, cfg.num_extra_tokens + cfg.num_patches
        M = B * Ne
        g, g_lp = w["g"][:M * D].view(M, D), w["g_lp"][:M * D].view(M, D)
        for i in reversed(range(lo, cfg.depth if hi is None else hi)):
            self._block_bwd(w["xs"][i], w["enc"][i], f"blocks.{i}", M, D, cfg.num_heads, B, Ne, g, g_lp, w)
        self._end_stage(w)

    def backward_embed(self):
        """Last stage: cls token, RA/Dec encoder, patch embedding, patch_mask_values (g = d xs[0])."""
        imgs, B, mask, w = self._ctx()
        cfg, st = self.cfg, self.store
        L, D, pv, E = cfg.num_patches, cfg.embed_dim, cfg.patch_dim, cfg.num_extra_tokens
        Ne = E + L
        M = B * Ne
        g = w["g"][:M * D].view(M, D)
        ops.rowsum_select(g, D, None, 0, 1, Ne, B, D, w["rs_part"], st.grad("cls_token").view(D))
        if cfg.ra_dec:
            G = st.grad
            ops.radec_token_bwd(g.view(-1)[D:], Ne * D, st.param("ra_dec_embed.neural_network.last_layer.weight"), w["sh"],
                                w["z"], w["dz"], G("ra_dec_embed.neural_network.layers.0.weight"),
                                G("ra_dec_embed.neural_network.layers.0.bias"), G("ra_dec_embed.neural_network.last_layer.weight"),
                                G("ra_dec_embed.neural_network.last_layer.bias"), B, D)
        ops.gather_rows(g, w["pe_dst"], None, w["dT"], B * L, D)
        self._wgrad(w["dT"], w["patches"], D, pv, B * L, st.grad("patch_embed.proj.weight"), st.grad("patch_embed.proj.bias"), w)
        ops.gemm(w["dT"], st.lp("patch_embed.proj.weight"), M=B * L, N=pv, K=D, a_layout=KC, b_layout=RC, lda=D, ldb=pv,
                 out_f32=w["drows"])
        ops.patch_gather_bwd_pmv_blend(imgs, None, mask, w["drows"], w["pmv_part"], st.grad("patch_mask_values"),
                                       cfg.patch_size, L)
        self._end_stage(w, last=True)

    def backward(self):
        self.backward_decoder()
        self.backward_encoder()
        self.backward_embed()

    def backward_stages(self, n_encoder_groups=3):
        groups, ranges = simmim_stage_ranges(self.store, self.cfg, n_encoder_groups)
        stages = [(self.backward_decoder, ranges[0])]
        for k, (hi, lo) in enumerate(groups):
            stages.append(((lambda h=hi, l=lo: self.backward_encoder(h, l)), ranges[1 + k]))
        stages.append((self.backward_embed, ranges[-1]))
        return stages

    # ------------------------------------------------------------------ accounting
    def flops_per_image(self, mask_ratio=0.0):
        """(executed, reference-algorithmic) forward+backward FLOPs per image, 3x-forward convention."""
        cfg = self.cfg
        L, D, pv, Ne = cfg.num_patches, cfg.embed_dim, cfg.patch_dim, cfg.num_extra_tokens + cfg.num_patches
        hidden = int(D * cfg.mlp_ratio)
        hd = D // cfg.num_heads
        blk = 2 * Ne * D * (3 * D + D + 2 * hidden) + 4 * Ne * Ne * hd * cfg.num_heads
        fwd = 2 * L * pv * D + cfg.depth * blk + 2 * L * D * pv
        fwd_exec = fwd + 2 * cfg.num_extra_tokens * D * pv          # the head also runs over the extra rows
        return 3.0 * fwd_exec, 3.0 * fwd
