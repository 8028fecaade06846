"""SimMIM mode of the MIM path (utils/mim_vit.py:244-264, 394-399, 410-414, 431-436, 469, 480-493) on the same kernels:
per-channel pixel masks blended into the input, every patch token encoded (plus the optional RA/Dec token), a linear
head per token (= Conv1x1 + PixelShuffle) and a pixel-wise masked loss.  The launch schedule reuses MAEEngine's
transformer-block forward / backward, LayerNorm batching and grouped weight-gradient launches.

Sequence layout: Ne = E + L tokens per sample, E = 1 (cls) or 2 (cls, RA/Dec); rows b*Ne + {0: cls, 1: RA/Dec, E + l: patch l}.
The head runs over all M = B*Ne rows (the E extra rows are ignored by the loss, whose gradient for them is zero): one
plain GEMM instead of a gather.
"""
from __future__ import annotations

import torch

from . import ops
from .engine import MAEEngine
from .model_config import MAEConfig
from .ops import KC, RC


def simmim_stage_ranges(store, cfg: MAEConfig, n_encoder_groups: int = 3):
    """Backward stages [head, encoder block groups (top first) ..., embedding] -> slices of the flat gradient buffer
    that are final after each (cf. engine.stage_gradient_ranges)."""
    off = store.offsets
    head0 = off["decoder.0.weight"]
    ranges = [[(head0, store.n_decay)]]
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
        if train:
            w["pred_tok"] = torch.empty(M, pv, **f32)
            w["pred_img"] = torch.empty(B, cfg.in_chans, cfg.img_size, cfg.img_size, **f32)
            w["loss"] = torch.zeros(1, **f32)
            w["loss_ws"] = torch.empty(4 * B * L + 4, **f32)
            w["dpred"] = torch.empty(M, pv, **lp)
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
            w["rs_part"] = torch.empty(64, D, **f32)
            w["splitk_ws"] = self._splitk_ws
            order = [("norm", M, D)]
            for i in reversed(range(cfg.depth)):
                order += [(f"blocks.{i}.norm2", M, D), (f"blocks.{i}.norm1", M, D)]
            entries = []
            w["ln_index"], w["ln_parts"] = {}, []
            for k, (name, M_, D_) in enumerate(order):
                nb = ops.layernorm_bwd_blocks(M_)
                part = torch.empty(2, nb, D_, **f32)
                w["ln_index"][name] = k
                w["ln_parts"].append(part)
                entries.append((part, self.store.grad(f"{name}.weight"), self.store.grad(f"{name}.bias"), nb, D_))
            w["ln_items"] = ops.ln_reduce_items(entries, dev)
            w["ln_max_D"] = D
            w["wgrad_groups"] = {}
            if self.dtype == torch.bfloat16:
                for i, bufs in enumerate(w["enc"]):
                    w["wgrad_groups"][f"blocks.{i}"] = self._make_wgrad_group(f"blocks.{i}", bufs, M, D, w)
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
                 table=pos[E:], tab_row=w["pe_tab"], ldt=D, dst_row=w["pe_dst"], out_f32=x0, ldo32=D)
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
        ops.layernorm_fwd(x_last, st.param("norm.weight"), st.param("norm.bias"), w["lat_lp"], w["lat_mean"], w["lat_rstd"],
                          M, D, cfg.ln_eps, y32=w["latent32"])
        return x_last

    def forward_features(self, imgs, mask_ratio=0.0, noise=None, mask=None, ra_dec=None):
        """utils/mim_vit.py:381-438 (reshape_out=False): -> (latent fp32 [B, E+L, D], mask, None); tokens keep their order."""
        cfg = self.cfg
        mask, ra_dec = self._check_simmim_inputs(imgs, mask, ra_dec)
        B = imgs.shape[0]
        w = self._workspace(B, cfg.num_patches, False)
        self._encoder_fwd_simmim(imgs, mask, ra_dec, w, False)
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
        self._encoder_fwd_simmim(imgs, mask, ra_dec, w, True)
        # head: Conv1x1 D -> p*p*C per token; PixelShuffle(p) is the loss kernel's index map (utils/mim_vit.py:254-261,469)
        ops.gemm(w["lat_lp"], st.lp("decoder.0.weight"), M=M, N=pv, K=D, bias=st.param("decoder.0.bias"), out_f32=w["pred_tok"])
        ops.simmim_pixel_loss(imgs, w["pred_tok"], mask, w["loss"], w["dpred"], self.code, w["pred_img"], w["loss_ws"],
                              cfg.patch_size, E, cfg.pixel_mean, cfg.pixel_std, cfg.norm_pix_loss, cfg.loss_fn != "mse")
        self._last = (imgs, B, L, mask)
        return w["loss"], w["pred_img"], mask

    # ------------------------------------------------------------------ backward
    def _ctx(self):
        assert self._last is not None, "backward() without forward_train()"
        imgs, B, L, mask = self._last
        return imgs, B, mask, self._ws[(B, L, True)]

    def backward_decoder(self):
        """Stage 0: the head (decoder.0) and the final norm; leaves d(block output) in g / g_lp."""
        imgs, B, mask, w = self._ctx()
        cfg = self.cfg
        D, pv = cfg.embed_dim, cfg.patch_dim
        M = B * (cfg.num_extra_tokens + cfg.num_patches)
        self._ln_first = self._ln_count = 0
        dln = w["dln"][:M * D].view(M, D)
        self._linear_bwd(w["dpred"], w["lat_lp"], "decoder.0.weight", "decoder.0.bias", M, pv, D, w, dx_out=dln)
        g, g_lp = w["g"][:M * D].view(M, D), w["g_lp"][:M * D].view(M, D)
        self._ln_bwd(dln, w["xs"][cfg.depth], "norm", w["lat_mean"], w["lat_rstd"], None, g, g_lp, M, D, w)
        self._end_stage(w)

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
