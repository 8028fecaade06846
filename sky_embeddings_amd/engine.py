"""MAE pretraining engine: explicit forward / backward / optimiser schedule over libskyemb.

This replaces the reference's autograd graph (utils/mim_vit.py:552-559 forward,
utils/pretrain_fns.py:26-41 backward + AdamW) with a fixed launch sequence of hand-written
gfx950 kernels.  Python only sequences the launches and owns the buffers (torch tensors);
no torch op touches activations on the hot path, so a whole step can be captured in a HIP graph.

Memory layout (HBM, per process):
  * parameters: ONE flat fp32 buffer ``p`` laid out [decayed tensors | non-decayed tensors],
    every tensor padded to a multiple of 8 elements; ``g`` (grads), ``m``, ``v`` (Adam state)
    mirror it; ``p_lp`` is the bf16 (or fp32) shadow the GEMMs read, refreshed by the AdamW kernel.
    DDP all-reduces contiguous slices of ``g``.
  * activations: residual stream fp32 [tokens, D]; GEMM operands in the compute dtype;
    per-block tensors saved for backward (LN inputs + stats, qkv, attention out, MLP pre/post).
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np
import torch

from . import ops
from ._lib import ACT_DGELU, ACT_GELU, BF16, F32, KC, RC
from .model_config import (FROZEN, SH_FEATURES, SIREN_HIDDEN, MAEConfig, not_optimised, sincos_pos_embed, state_layout,
                           weight_decay_split, xavier_bound)


def _pad8(n):
    return (n + 7) // 8 * 8


class ParamStore:
    """Flat parameter / gradient / optimiser-state buffers with named views."""

    def __init__(self, cfg: MAEConfig, device, lp_dtype):
        self.cfg = cfg
        shapes = dict(state_layout(cfg))
        decay, no_decay = weight_decay_split(cfg)
        self.decay, self.no_decay = decay, no_decay
        self.order = decay + no_decay
        self.offsets = {}
        off = 0
        for name in decay:
            self.offsets[name] = off
            off += _pad8(int(np.prod(shapes[name])))
        self.n_decay = off
        for name in no_decay:
            self.offsets[name] = off
            off += _pad8(int(np.prod(shapes[name])))
        self.n = off
        self.shapes = shapes
        self.p = torch.zeros(self.n, device=device, dtype=torch.float32)
        self.g = torch.zeros(self.n, device=device, dtype=torch.float32)
        self.m = torch.zeros(self.n, device=device, dtype=torch.float32)
        self.v = torch.zeros(self.n, device=device, dtype=torch.float32)
        self.p_lp = torch.zeros(self.n, device=device, dtype=lp_dtype)
        # tensors of the state dict that no optimiser step touches (sincos tables; SimMIM's unused mask_token)
        self.frozen = {k: torch.zeros(shapes[k], device=device, dtype=torch.float32) for k in not_optimised(cfg) if k in shapes}

    def _view(self, buf, name):
        o = self.offsets[name]
        return buf[o:o + int(np.prod(self.shapes[name]))].view(self.shapes[name])

    def param(self, name):
        return self.frozen[name] if name in self.frozen else self._view(self.p, name)

    def grad(self, name):
        return self._view(self.g, name)

    def lp(self, name):
        return self._view(self.p_lp, name)

    def refresh_lp(self):
        ops.cast(self.p, self.p_lp, self.n)


def stage_gradient_ranges(store: ParamStore, cfg: MAEConfig, n_encoder_groups: int = 3):
    """Slices of the flat gradient buffer that are FINAL after each backward stage
    [decoder, encoder block groups (top first) ..., embedding].  Decayed weights sit in layout order, so
    every stage owns one contiguous run of them; the small non-decayed tail (biases, LayerNorm) and
    the front tensors (cls, patch_mask_values, mask_token, patch embedding) go with the last stage.
    Returns (encoder groups [(hi, lo), ...], ranges per stage)."""
    off = store.offsets
    # decoder_embed.weight, the first tensor of the decoder's run, goes with the TOP ENCODER stage: its gradient may be a problem of
    # blocks.{depth-1}'s grouped weight-gradient launch (MAEEngine._extra_wgrad_layers), i.e. written after the decoder stage ended
    dec0 = off["decoder_embed.weight"] + _pad8(int(np.prod(store.shapes["decoder_embed.weight"])))
    ranges = [[(dec0, store.n_decay)]]
    bounds = sorted({round(cfg.depth * k / n_encoder_groups) for k in range(n_encoder_groups + 1)}, reverse=True)
    groups = []
    for hi, lo in zip(bounds[:-1], bounds[1:]):
        groups.append((hi, lo))
        end = dec0 if hi == cfg.depth else off[f"blocks.{hi}.attn.qkv.weight"]
        ranges.append([(off[f"blocks.{lo}.attn.qkv.weight"], end)])
    ranges.append([(0, off["blocks.0.attn.qkv.weight"]), (store.n_decay, store.n)])
    return groups, ranges


class MAEEngine:
    _modes = "mae"      # SimMIMEngine (simmim_engine.py) handles simmim=True and the RA/Dec token

    def __init__(self, cfg: MAEConfig, device="cuda", compute_dtype=torch.bfloat16, seed=None):
        if cfg.attn_pool and not cfg.simmim:
            raise ValueError("attn_pool exists in SimMIM mode only (utils/mim_vit.py:244-254, :282)")
        if cfg.simmim and self._modes == "mae":
            raise NotImplementedError("simmim=True is served by sky_embeddings_amd.simmim_engine.SimMIMEngine")
        assert cfg.embed_dim % cfg.num_heads == 0 and cfg.decoder_embed_dim % cfg.decoder_num_heads == 0
        assert cfg.embed_dim % 8 == 0 and cfg.decoder_embed_dim % 8 == 0 and cfg.patch_size % 4 == 0
        self.cfg = cfg
        self.device = torch.device(device)
        self.dtype = compute_dtype
        self.code = ops.dtype_code(compute_dtype)
        # Loss scale of the backward pass (a power of two; 1 outside the fp16 mode).  fp16 data gradients underflow without one
        # (d loss / d pred ~ 2 / masked elements ~ 5e-7 at B = 256: below fp16's normal range); every step of backward is linear in
        # d loss / d pred, so the loss kernel multiplies that by `loss_scale` (dscale of skyemb_masked_patch_loss) and the optimiser
        # divides it out (FusedAdamW.grad_scale, skyemb_adamw_desc.grad_scale): the flat gradient buffer holds loss_scale x the
        # gradients -- `grad(name)` of this class returns them unscaled.  The scale follows the batch: 2^round(log2(expected masked
        # elements / 64)), at most 2^16, so that d loss / d pred enters backward at ~0.03 whatever the batch (2^16 at B = 256 and at
        # mim_19's B = 128 -- profiles/r06_operand_rounding.json: with it fp16 gradients sit at the format's rounding floor, 1e-3;
        # without, 2.6e-2 -- 2^11 at B = 8: a fixed 2^16 would put d loss / d pred of a four-image batch at ~10 and the products
        # behind it past fp16's 65504).  SKYEMB_LOSS_SCALE or assigning `engine.loss_scale` fixes it by hand.
        import os
        env = os.environ.get("SKYEMB_LOSS_SCALE")
        self._loss_scale_auto = compute_dtype == torch.float16 and env is None
        self._loss_scale = float(env) if (env is not None and compute_dtype == torch.float16) else (65536.0 if compute_dtype == torch.float16 else 1.0)
        assert self._loss_scale > 0 and math.log2(self._loss_scale).is_integer(), "loss_scale must be a power of two"
        self.fold_decoder_wgrads = True      # utils.vit's predictor turns it off: it never runs backward_decoder (_extra_wgrad_layers)
        self.store = ParamStore(cfg, self.device, compute_dtype)
        self._ws = {}
        self._last = None
        # fp32 scratch for split-K GEMM launches (partial slabs; every launch on the stream reuses it)
        self._splitk_ws = torch.zeros(8 * 1024 * 1024, device=self.device, dtype=torch.float32)
        self._side, self._pending, self._group_done = None, {}, None
        self._ln_first = self._ln_count = 0      # LayerNorms of the running backward stage awaiting their batched reduce
        self.initialize_weights(seed)

    # ------------------------------------------------------------------ parameters
    def initialize_weights(self, seed=None):
        """utils/mim_vit.py:290-324: sincos tables, xavier_uniform on every Linear and on the conv
        weight viewed [D,-1], N(0,0.02) cls/mask tokens, LayerNorm 1/0, zero Linear biases,
        patch_mask_values zeros; the conv bias keeps nn.Conv2d's default U(+-1/sqrt(fan_in))."""
        cfg, st = self.cfg, self.store
        gen = torch.Generator().manual_seed(seed if seed is not None else int(torch.initial_seed() % (2 ** 31)))
        for name in st.order:
            shape = st.shapes[name]
            if name in ("cls_token", "mask_token"):
                t = torch.randn(shape, generator=gen) * 0.02
            elif name == "attn_pool.latent":     # timm AttentionPoolLatent.init_weights: trunc_normal_tf_(std = dim ** -0.5)
                sd_ = shape[-1] ** -0.5
                t = (torch.randn(shape, generator=gen) * sd_).clamp_(-2 * sd_, 2 * sd_)
            elif name == "patch_mask_values":
                t = torch.zeros(shape)
            elif name == "patch_embed.proj.bias" or name.startswith("decoder.0."):
                # nn.Conv2d default init, untouched by _init_weights: U(+-1/sqrt(fan_in)) for weight and bias
                b = 1.0 / math.sqrt(cfg.patch_dim if name.startswith("patch_embed") else cfg.embed_dim)
                t = (torch.rand(shape, generator=gen) * 2 - 1) * b
            elif name.startswith("ra_dec_embed."):
                # Siren.init_ (utils/location_encoder.py:41-49): first layer U(+-1/dim_in); last U(+-sqrt(6/dim_in)) (w0 = 1)
                first = ".layers.0." in name
                w_std = (1.0 / SH_FEATURES) if first else math.sqrt(6.0 / SIREN_HIDDEN)
                t = (torch.rand(shape, generator=gen) * 2 - 1) * w_std
            elif name.endswith("norm1.weight") or name.endswith("norm2.weight") or name.endswith("norm.weight"):
                t = torch.ones(shape)
            elif name.endswith(".bias"):
                t = torch.zeros(shape)
            else:
                t = (torch.rand(shape, generator=gen) * 2 - 1) * xavier_bound(shape)
            st.param(name).copy_(t)
        for k in st.frozen:
            if k in FROZEN:
                tab = sincos_pos_embed(st.shapes[k][-1], cfg.grid, True, cfg.ra_dec)
                st.frozen[k].copy_(torch.from_numpy(tab).float().unsqueeze(0))
            else:
                st.frozen[k].copy_(torch.randn(st.shapes[k], generator=gen) * 0.02)   # SimMIM's unused mask_token
        st.refresh_lp()

    @property
    def loss_scale(self):
        return self._loss_scale

    @loss_scale.setter
    def loss_scale(self, v):
        assert v > 0 and math.log2(v).is_integer(), "loss_scale must be a power of two"
        self._loss_scale, self._loss_scale_auto = float(v), False

    def plan_loss_scale(self, masked_elements):
        """fp16 mode: the scale for a batch whose loss averages over ~`masked_elements` pixels (see __init__); called by forward_train
        with the batch's expected count, and by TrainStep before it bakes the scale into the fused optimiser launches."""
        if self._loss_scale_auto:
            e = round(math.log2(max(float(masked_elements), 64.0) / 64.0))
            self._loss_scale = float(2 ** min(max(e, 0), 16))
        return self._loss_scale

    def expected_masked_elements(self, B, mask_ratio):
        """Pixels the loss of a B-image batch averages over (MAE: the masked patches)."""
        cfg = self.cfg
        return B * (cfg.num_patches - int(cfg.num_patches * (1 - mask_ratio))) * cfg.patch_dim

    def grad(self, name):
        """d loss / d parameter `name` of the last backward(): the flat gradient buffer's view without the loss scale."""
        g = self.store.grad(name)
        return g if self.loss_scale == 1.0 else g / self.loss_scale

    def state_dict(self):
        out = OrderedDict()
        for name, _ in state_layout(self.cfg):
            out[name] = self.store.param(name)
        return out

    def load_state_dict(self, sd, strict=True):
        names = [n for n, _ in state_layout(self.cfg)]
        missing = [n for n in names if n not in sd]
        unexpected = [k for k in sd if k not in names]
        if strict and (missing or unexpected):
            raise RuntimeError(f"load_state_dict: missing {missing[:5]} unexpected {unexpected[:5]}")
        for n in names:
            if n in sd:
                self.store.param(n).copy_(torch.as_tensor(sd[n]).to(torch.float32).reshape(self.store.shapes[n]))
        self.store.refresh_lp()

    # ------------------------------------------------------------------ workspaces
    def _workspace(self, B, keep, train):
        key = (B, keep, train)
        if key in self._ws:
            return self._ws[key]
        cfg, dev, T = self.cfg, self.device, self.dtype
        L, D, Dd, pv, E = cfg.num_patches, cfg.embed_dim, cfg.decoder_embed_dim, cfg.patch_dim, cfg.num_extra_tokens
        Ne, Nd = E + keep, E + L            # E extra tokens first: cls (, RA/Dec: utils/mim_vit.py:410-419)
        Me, Md = B * Ne, B * Nd
        f32 = dict(device=dev, dtype=torch.float32)
        lp = dict(device=dev, dtype=T)
        i32 = dict(device=dev, dtype=torch.int32)
        w = {}
        w["sh"], w["z"], w["dz"] = torch.empty(B, 25, **f32), torch.empty(B, 8, **f32), torch.empty(B, 8, **f32)   # RA/Dec encoder
        w["ids_restore"] = torch.empty(B, L, device=dev, dtype=torch.int64)
        w["mask"] = torch.empty(B, L, **f32)
        w["ids_keep"] = torch.empty(B, keep, **i32)
        w["dec_dst"] = torch.empty(B, Ne, **i32)
        w["dec_tab"] = torch.empty(B, Ne, **i32)
        w["pe_dst"] = (torch.arange(B, device=dev)[:, None] * Ne + E + torch.arange(keep, device=dev)[None, :]).to(torch.int32).contiguous()
        w["patches"] = torch.empty(B * keep, pv, **lp)
        w["latent32"] = torch.empty(Me, D, **f32)

        def block_bufs(M, dim):
            return dict(ln1=torch.empty(M, dim, **lp), mean1=torch.empty(M, **f32), rstd1=torch.empty(M, **f32),
                        qkv=torch.empty(M, 3 * dim, **lp), att=torch.empty(M, dim, **lp), xmid=torch.empty(M, dim, **f32),
                        ln2=torch.empty(M, dim, **lp), mean2=torch.empty(M, **f32), rstd2=torch.empty(M, **f32),
                        hpre=torch.empty(M, int(dim * cfg.mlp_ratio), **lp), hact=torch.empty(M, int(dim * cfg.mlp_ratio), **lp))

        n_enc_sets = cfg.depth if train else 1
        w["enc"] = [block_bufs(Me, D) for _ in range(n_enc_sets)]
        w["xs"] = [torch.empty(Me, D, **f32) for _ in range((cfg.depth + 1) if train else 2)]
        w["lat_lp"] = torch.empty(Me, D, **lp)
        w["lat_mean"], w["lat_rstd"] = torch.empty(Me, **f32), torch.empty(Me, **f32)
        if train:
            w["dec"] = [block_bufs(Md, Dd) for _ in range(cfg.decoder_depth)]
            w["xd"] = [torch.empty(Md, Dd, **f32) for _ in range(cfg.decoder_depth + 1)]
            w["dlat_lp"] = torch.empty(Md, Dd, **lp)
            w["dlat_mean"], w["dlat_rstd"] = torch.empty(Md, **f32), torch.empty(Md, **f32)
            w["pred"] = torch.empty(B, Nd, pv, **f32)
            w["loss"] = torch.zeros(1, **f32)
            w["loss_ws"] = torch.empty(4 * B * L + 4, **f32)
            w["dpred"] = torch.empty(Md, pv, **lp)
            # backward temporaries (sized for the larger of encoder / decoder)
            Mx = max(Me, Md)
            Dx = max(D, Dd)
            Hx = max(Me * int(D * cfg.mlp_ratio), Md * int(Dd * cfg.mlp_ratio))
            w["g"] = torch.empty(Mx * Dx, **f32)
            w["g_lp"] = torch.empty(Mx * Dx, **lp)
            w["g_lp2"] = torch.empty(Mx * Dx, **lp)   # d(xmid) copy: keeps g_lp (fc2's dy) alive for the grouped wgrad launch
            # second set of the four dy buffers a block's grouped weight-gradient launch reads: consecutive blocks
            # alternate between the sets, so that launch can run on the side stream under the NEXT block's dgrad chain
            w["g_lp_b"], w["g_lp2_b"] = torch.empty(Mx * Dx, **lp), torch.empty(Mx * Dx, **lp)
            w["dh_b"], w["dqkv_b"] = torch.empty(Hx, **lp), torch.empty(3 * Mx * Dx, **lp)
            w["dln"] = torch.empty(Mx * Dx, **lp)
            w["datt"] = torch.empty(Mx * Dx, **lp)
            w["dh"] = torch.empty(Hx, **lp)
            w["dqkv"] = torch.empty(3 * Mx * Dx, **lp)
            # LayerNorm dgamma/dbeta: every LN keeps its own partial sums; ONE batched launch per backward stage
            # finishes them (42 reduce launches -> 3-6).  Table order = the order backward visits the LNs.
            order = [("ln", "decoder_norm", Md, Dd)]
            for i in reversed(range(cfg.decoder_depth)):
                order.append(("block", f"decoder_blocks.{i}", w["dec"][i], Md, Dd))
            order.append(("ln", "norm", Me, D))
            for i in reversed(range(cfg.depth)):
                order.append(("block", f"blocks.{i}", w["enc"][i], Me, D))
            self._build_reduce_table(w, order)
            w["dE"] = torch.empty(Me, Dd, **lp)
            # the four weight-gradient GEMMs of a block as ONE grouped launch (ops.GemmGroup): together they fill the
            # chip, so none needs split-K slabs or a reduce launch.  bf16 path only; pointers are fixed from here on.
            # (the block's norm1 backward rides in the same launch as a side job: _make_wgrad_group)
            w["splitk_ws"] = self._splitk_ws
            w["wgrad_groups"] = {}
            if self.dtype in ops.LP_DTYPES:
                for tag, blocks, M_, dim in (("blocks", w["enc"], Me, D), ("decoder_blocks", w["dec"], Md, Dd)):
                    for i, bufs in enumerate(blocks):
                        w["wgrad_groups"][f"{tag}.{i}"] = self._make_wgrad_group(f"{tag}.{i}", bufs, M_, dim, w)
            w["dT"] = torch.empty(B * keep, D, **lp)
            w["drows"] = torch.empty(B * keep, pv, **f32)
            w["pmv_part"] = torch.empty(B, pv, **f32)
            w["rs_part"] = torch.empty(256, max(D, Dd), **f32)
        if train and getattr(self, "_fused_adamw", None) is not None and w.get("wgrad_groups"):
            self._build_adamw_groups(w)
        if train and getattr(self, "_g16", None) is not None and w.get("wgrad_groups"):
            self._build_variant_groups(w, "g16")
        self._ws[key] = w
        return w

    def _build_reduce_table(self, w, order):
        """Device table of the batched column reduces of backward (ops.layernorm_bwd_reduce_batch), in the order backward visits
        its entries: ("ln", name, M, D) = one LayerNorm's dgamma / dbeta partial sums; ("block", prefix, bufs, M, dim) = a
        transformer block: norm2, then norm1."""
        f32 = dict(device=self.device, dtype=torch.float32)
        entries = []
        w["ln_index"], w["ln_parts"] = {}, {}

        def ln(name, M_, D_):
            nb = ops.layernorm_bwd_blocks(M_)
            part = torch.empty(2, nb, D_, **f32)
            w["ln_index"][name] = len(entries)
            w["ln_parts"][name] = part
            entries.append((part, self.store.grad(f"{name}.weight"), self.store.grad(f"{name}.bias"), nb, D_))
        for item in order:
            if item[0] == "ln":
                ln(*item[1:])
                continue
            _, prefix, bufs, M_, dim = item
            ln(f"{prefix}.norm2", M_, dim)
            ln(f"{prefix}.norm1", M_, dim)
        w["ln_items"] = ops.ln_reduce_items(entries, self.device)

    # ------------------------------------------------------------------ prefetch hints
    # Between two uses of a weight matrix a step moves far more than the 256 MB memory-side cache holds (activations, 2.9 GB of
    # optimiser state), so every GEMM of the chain found its weights in HBM and paid that latency inside its first k-steps
    # (tools/ubench/cold_weights_probe.py: [1280 x 3072 x 768] 12.2 us with the weights cached, 14.6 from HBM -- the in-step figure).
    # Every GEMM launch therefore names the weights the NEXT GEMM of the chain will read (skyemb_gemm_args.prefetch): its workgroups
    # touch them before their own first loads, and the lines are in the memory-side cache when the next launch asks.
    # Measured (bench.py extra.optimizer_placement, interleaved): config A 5.08 -> 4.74 ms; forward hints alone 4.95, backward alone 4.90.
    # WEIGHTS only: naming saved activations as well (the dGELU operand, norm2's input, the attention backward's qkv rows, the grouped
    # launch's operands) made the step slower again (4.83; 4.79 for the grouped launch's operand alone) -- those are streamed by
    # bandwidth-type kernels that gain nothing, while the extra lines sit in front of the naming launch's own first stage.
    def _weight_chain(self):
        """Weight names in the order the forward GEMMs read them (backward's data gradients read them in reverse)."""
        cfg = self.cfg
        blk = lambda p: [f"{p}.attn.qkv.weight", f"{p}.attn.proj.weight", f"{p}.mlp.fc1.weight", f"{p}.mlp.fc2.weight"]
        names = ["patch_embed.proj.weight"]
        for i in range(cfg.depth):
            names += blk(f"blocks.{i}")
        names += self._decoder_weight_chain()
        return [n for n in names if n in self.store.offsets]

    def _decoder_weight_chain(self):
        cfg = self.cfg
        names = ["decoder_embed.weight"]
        for i in range(cfg.decoder_depth):
            names += [f"decoder_blocks.{i}.attn.qkv.weight", f"decoder_blocks.{i}.attn.proj.weight", f"decoder_blocks.{i}.mlp.fc1.weight",
                      f"decoder_blocks.{i}.mlp.fc2.weight"]
        return names + ["decoder_pred.weight"]

    PF_MAX_ROWS = 8192     # token rows above which launches name nothing: ViT-L at B = 128 (8320 rows) has k-loops of tens of us that
                           # re-read a weight panel from the L2 thirty times -- its first touch is amortised, and the hint's lines in front of
                           # every workgroup's first stage cost more than they return there (mim_19: 24.91 without, 25.05 ms with)

    def _pf(self, direction, wname, rows=0):
        """The bf16 weights the GEMM after the one reading `wname` will read (forward chain / backward's data-gradient chain), or None.
        rows: token rows of the naming launch."""
        import os
        if rows >= int(os.environ.get("SKYEMB_PF_MAX_ROWS", self.PF_MAX_ROWS)):
            return None
        maps = getattr(self, "_pf_maps", None)
        if maps is None:
            chain = self._weight_chain()
            maps = self._pf_maps = {"fwd": dict(zip(chain[:-1], chain[1:])), "bwd": dict(zip(chain[1:], chain[:-1]))}
        if self.dtype not in ops.LP_DTYPES:
            return None
        only = os.environ.get("SKYEMB_PF_ONLY", "")        # (bench.py's A/B: hints in one direction only)
        if only and only != direction:
            return None
        nxt = maps[direction].get(wname)
        return None if nxt is None else self.store.lp(nxt)

    # ------------------------------------------------------------------ forward pieces
    def _embed(self, imgs, noise, keep, w, ra_dec=None):
        """a3-a6: mask from noise, fused normalise/NaN-fill/gather of the kept patches, patch-embed
        GEMM with bias + positional rows scattered into the token sequence, (RA/Dec token,) cls row."""
        cfg, st = self.cfg, self.store
        B = imgs.shape[0]
        L, D, pv, E = cfg.num_patches, cfg.embed_dim, cfg.patch_dim, cfg.num_extra_tokens
        Ne = E + keep
        ops.random_mask_from_noise(noise, keep, w["ids_restore"], w["mask"], w["ids_keep"], w["dec_dst"], w["dec_tab"], n_extra=E)
        ops.patch_gather(imgs, st.param("patch_mask_values"), w["ids_keep"], w["patches"], cfg.patch_size, keep,
                         cfg.pixel_mean, cfg.pixel_std)
        x0 = w["xs"][0]
        pos = st.frozen["pos_embed"].view(-1, D)
        ops.gemm(w["patches"], st.lp("patch_embed.proj.weight"), M=B * keep, N=D, K=pv,
                 bias=st.param("patch_embed.proj.bias"), table=pos[E:], tab_row=w["ids_keep"], ldt=D,
                 dst_row=w["pe_dst"], out_f32=x0, ldo32=D, prefetch=self._pf("fwd", "patch_embed.proj.weight", B * keep))
        # cls_token + pos_embed[:, :1] (utils/mim_vit.py:417-419): B tiny row copies (host glue)
        x0.view(B, Ne, D)[:, 0, :] = st.param("cls_token").view(D) + pos[0]
        if cfg.ra_dec:
            # LocationEncoder(ra_dec) + pos_embed[:, 1] right behind the cls token (utils/mim_vit.py:410-414)
            P = st.param
            ops.radec_token_fwd(ra_dec, P("ra_dec_embed.neural_network.layers.0.weight"),
                                P("ra_dec_embed.neural_network.layers.0.bias"), P("ra_dec_embed.neural_network.last_layer.weight"),
                                P("ra_dec_embed.neural_network.last_layer.bias"), pos[1], x0.view(-1)[D:], Ne * D, B, D,
                                w["sh"], w["z"])
        return x0

    def _block_fwd(self, x_in, x_out, bufs, prefix, M, dim, heads, Bsz, N):
        st, eps = self.store, self.cfg.ln_eps
        hd = dim // heads
        hidden = bufs["hpre"].shape[1]
        P, LP = st.param, st.lp
        ops.layernorm_fwd(x_in, P(f"{prefix}.norm1.weight"), P(f"{prefix}.norm1.bias"), bufs["ln1"], bufs["mean1"],
                          bufs["rstd1"], M, dim, eps)
        PF = lambda name: self._pf("fwd", f"{prefix}.{name}.weight", M)     # (the next GEMM's weights: _pf)
        ops.gemm(bufs["ln1"], LP(f"{prefix}.attn.qkv.weight"), M=M, N=3 * dim, K=dim, bias=P(f"{prefix}.attn.qkv.bias"),
                 out=bufs["qkv"], prefetch=PF("attn.qkv"))
        ops.mha_fwd(bufs["qkv"], bufs["att"], Bsz, N, heads, hd)
        ops.gemm(bufs["att"], LP(f"{prefix}.attn.proj.weight"), M=M, N=dim, K=dim, bias=P(f"{prefix}.attn.proj.bias"),
                 resid=x_in, ldr=dim, out_f32=bufs["xmid"], ws=self._splitk_ws, prefetch=PF("attn.proj"))
        ops.layernorm_fwd(bufs["xmid"], P(f"{prefix}.norm2.weight"), P(f"{prefix}.norm2.bias"), bufs["ln2"],
                          bufs["mean2"], bufs["rstd2"], M, dim, eps)
        ops.gemm(bufs["ln2"], LP(f"{prefix}.mlp.fc1.weight"), M=M, N=hidden, K=dim, bias=P(f"{prefix}.mlp.fc1.bias"),
                 act=ACT_GELU, out=bufs["hact"], out2=bufs["hpre"], prefetch=PF("mlp.fc1"))
        ops.gemm(bufs["hact"], LP(f"{prefix}.mlp.fc2.weight"), M=M, N=dim, K=hidden, bias=P(f"{prefix}.mlp.fc2.bias"),
                 resid=bufs["xmid"], ldr=dim, out_f32=x_out, ws=self._splitk_ws, prefetch=PF("mlp.fc2"))

    def _encoder_fwd(self, imgs, noise, keep, w, train, ra_dec=None):
        cfg, st = self.cfg, self.store
        B = imgs.shape[0]
        D, Ne = cfg.embed_dim, cfg.num_extra_tokens + keep
        Me = B * Ne
        self._embed(imgs, noise, keep, w, ra_dec)
        xs = w["xs"]
        for i in range(cfg.depth):
            if train:
                self._block_fwd(xs[i], xs[i + 1], w["enc"][i], f"blocks.{i}", Me, D, cfg.num_heads, B, Ne)
            else:
                self._block_fwd(xs[i % 2], xs[(i + 1) % 2], w["enc"][0], f"blocks.{i}", Me, D, cfg.num_heads, B, Ne)
        x_last = xs[cfg.depth] if train else xs[cfg.depth % 2]
        ops.layernorm_fwd(x_last, st.param("norm.weight"), st.param("norm.bias"), w["lat_lp"], w["lat_mean"],
                          w["lat_rstd"], Me, D, cfg.ln_eps, y32=w["latent32"])
        return x_last

    # ------------------------------------------------------------------ public forward paths
    def _check_ra_dec(self, imgs, ra_dec):
        if not self.cfg.ra_dec:
            return None
        assert ra_dec is not None and tuple(ra_dec.shape) == (imgs.shape[0], 2), "ra_dec=True models need ra_dec [B,2] (degrees)"
        return ra_dec.to(device=imgs.device, dtype=torch.float32).contiguous()

    def _check_inputs(self, imgs, noise):
        cfg = self.cfg
        assert imgs.is_cuda and imgs.dtype == torch.float32 and imgs.is_contiguous()
        B, C, H, W = imgs.shape
        assert (C, H, W) == (cfg.in_chans, cfg.img_size, cfg.img_size), f"bad cutout shape {tuple(imgs.shape)}"
        if noise is None:
            noise = torch.rand(B, cfg.num_patches, device=imgs.device)  # utils/mim_vit.py:363
        assert noise.shape == (B, cfg.num_patches) and noise.is_cuda and noise.dtype == torch.float32
        return noise.contiguous()

    def forward_features(self, imgs, mask_ratio=0.0, noise=None, ra_dec=None):
        """utils/mim_vit.py:381-438 (MAE mode): -> (latent fp32 [B, E+keep, D], mask [B,L], ids_restore [B,L])."""
        cfg = self.cfg
        noise = self._check_inputs(imgs, noise)
        ra_dec = self._check_ra_dec(imgs, ra_dec)
        B = imgs.shape[0]
        keep = int(cfg.num_patches * (1 - mask_ratio))
        w = self._workspace(B, keep, False)
        self._encoder_fwd(imgs, noise, keep, w, False, ra_dec)
        return w["latent32"].view(B, cfg.num_extra_tokens + keep, cfg.embed_dim), w["mask"], w["ids_restore"]

    def forward_train(self, imgs, mask_ratio=0.75, noise=None, ra_dec=None):
        """utils/mim_vit.py:552-559: -> (loss [1] device tensor, pred fp32 view [B, L, pv], mask [B, L]).
        Activations are saved for :meth:`backward`."""
        cfg, st = self.cfg, self.store
        noise = self._check_inputs(imgs, noise)
        ra_dec = self._check_ra_dec(imgs, ra_dec)
        B = imgs.shape[0]
        L, D, Dd, pv, E = cfg.num_patches, cfg.embed_dim, cfg.decoder_embed_dim, cfg.patch_dim, cfg.num_extra_tokens
        keep = int(L * (1 - mask_ratio))
        assert keep >= 1, "mask_ratio leaves no visible patch"
        Ne, Nd = E + keep, E + L
        Me, Md = B * Ne, B * Nd
        w = self._workspace(B, keep, True)
        self.plan_loss_scale(B * (L - keep) * pv)
        self._encoder_fwd(imgs, noise, keep, w, True, ra_dec)
        # ---- decoder (utils/mim_vit.py:440-467)
        xd = w["xd"]
        dpos = st.frozen["decoder_pos_embed"].view(-1, Dd)
        ops.gemm(w["lat_lp"], st.lp("decoder_embed.weight"), M=Me, N=Dd, K=D, bias=st.param("decoder_embed.bias"),
                 table=dpos, tab_row=w["dec_tab"], ldt=Dd, dst_row=w["dec_dst"], out_f32=xd[0], ldo32=Dd,
                 prefetch=self._pf("fwd", "decoder_embed.weight", Me))
        ops.fill_mask_tokens(xd[0], w["mask"], st.param("mask_token"), dpos, B, L, Dd, n_extra=E)
        for i in range(cfg.decoder_depth):
            self._block_fwd(xd[i], xd[i + 1], w["dec"][i], f"decoder_blocks.{i}", Md, Dd, cfg.decoder_num_heads, B, Nd)
        ops.layernorm_fwd(xd[-1], st.param("decoder_norm.weight"), st.param("decoder_norm.bias"), w["dlat_lp"],
                          w["dlat_mean"], w["dlat_rstd"], Md, Dd, cfg.ln_eps)
        ops.gemm(w["dlat_lp"], st.lp("decoder_pred.weight"), M=Md, N=pv, K=Dd, bias=st.param("decoder_pred.bias"),
                 out_f32=w["pred"])
        # ---- loss + d loss / d pred (utils/mim_vit.py:473-521)
        ops.masked_patch_loss(imgs, w["pred"], w["mask"], w["loss"], w["dpred"], None, self.code, w["loss_ws"],
                              cfg.patch_size, E, cfg.pixel_mean, cfg.pixel_std, cfg.norm_pix_loss, cfg.loss_fn != "mse",
                              dscale=self.loss_scale)
        self._last = (imgs, B, keep)
        return w["loss"], w["pred"][:, E:, :], w["mask"]

    # ------------------------------------------------------------------ backward
    def _linear_bwd(self, dy, x_in, wname, bname, M, N, K, w, dx_out=None, dx_act=0, dx_aux=None, wgrad=True, prefetch="chain"):
        """dy [M,N] (lp), x_in [M,K] (lp): dW[N,K] = dy^T x, db = colsum(dy), optional dx = dy W."""
        st = self.store
        # wgrad; the bias gradient (column sums of dy) rides along in the same launch
        if wgrad:
            self._wgrad(dy, x_in, N, K, M, st.grad(wname), st.grad(bname), w)
        if dx_out is not None:
            self._before_write(dx_out)
            # (prefetch: the weights of the next data gradient of the chain -- _pf -- unless the caller names a tensor or None)
            ops.gemm(dy, st.lp(wname), M=M, N=K, K=N, a_layout=KC, b_layout=RC, lda=N, ldb=K, act=dx_act, aux=dx_aux,
                     ldaux=K, out=dx_out, ws=w["splitk_ws"], prefetch=self._pf("bwd", wname, M) if isinstance(prefetch, str) else prefetch)

    # -- weight gradients on a side stream: nothing downstream in backward depends on them, so they fill the
    # ramp / tail bubbles of the dgrad chain.  dy lives in scratch that later layers overwrite: every writer of such a
    # buffer first waits for the wgrad that still reads it (_before_write).
    def enable_wgrad_overlap(self, on=True):
        if bool(on) != (self._side is not None):
            # workspaces are planned for one of the two schedules (side workgroups and folded problems of the grouped launches exist
            # only without the side stream: _norm1_side_record, _extra_wgrad_layers): rebuild them on the next forward
            if self._ws and getattr(self, "_graph_captures", 0):
                # HIP graphs captured by an earlier TrainStep hold raw pointers into those workspaces (group blobs, scratch):
                # dropping them would leave that step replaying freed memory
                raise RuntimeError("enable_wgrad_overlap: the weight-gradient schedule cannot change while HIP graphs captured on this "
                                   "engine's workspaces exist (build a new engine, or construct every TrainStep with the same wgrad_overlap)")
            self._ws, self._last = {}, None
        if on and self._side is None:
            self._side = torch.cuda.Stream(device=self.device)
            self._splitk_ws_side = torch.zeros_like(self._splitk_ws)
        elif not on:
            self._side = None
        self._pending = {}

    def _wgrad(self, dy, x_in, M, N, K, dW, db, w):
        side = self._side
        if side is None:
            ops.gemm(dy, x_in, M=M, N=N, K=K, a_layout=RC, b_layout=RC, lda=M, ldb=N, out_f32=dW, colsum_a=db,
                     ws=w["splitk_ws"])
            return
        ready = torch.cuda.Event()
        ready.record()
        side.wait_event(ready)
        with torch.cuda.stream(side):
            ops.gemm(dy, x_in, M=M, N=N, K=K, a_layout=RC, b_layout=RC, lda=M, ldb=N, out_f32=dW, colsum_a=db,
                     ws=self._splitk_ws_side)
            done = torch.cuda.Event()
            done.record()
        self._pending[dy.data_ptr()] = done

    def _before_write(self, buf):
        if self._side is not None:
            ev = self._pending.pop(buf.data_ptr(), None)
            if ev is not None:
                torch.cuda.current_stream().wait_event(ev)

    def _end_stage(self, w, last=False):
        """End of a backward stage: finish the dgamma/dbeta of the LayerNorms it visited (one launch), join the
        weight-gradient stream."""
        if self._side is not None:                            # (the grouped launches' bias partial sums are reduced below)
            torch.cuda.current_stream().wait_stream(self._side)
            self._pending.clear()
            self._group_done = None
        if self._ln_count:
            ops.layernorm_bwd_reduce_batch(w["ln_items"], self._ln_first, self._ln_count)
        self._ln_first, self._ln_count = (0, 0) if last else (self._ln_first + self._ln_count, 0)

    def _ln_bwd(self, dy, x, prefix, mean, rstd, g_in, g, g_lp, M, dim, w):
        st = self.store
        k = w["ln_index"][prefix]
        assert k == self._ln_first + self._ln_count, "LayerNorm backward visited out of table order"
        self._ln_count += 1
        self._before_write(g_lp)
        ops.layernorm_bwd(dy, x, st.param(f"{prefix}.weight"), mean, rstd, g_in, g, g_lp, w["ln_parts"][prefix], None, None,
                          M, dim, self.code)

    def _bwd_set(self, prefix):
        """Scratch set (0 / 1) of a block's backward: blocks alternate, the top block of a stack uses set 0."""
        stack, i = prefix.rsplit(".", 1)
        depth = self.cfg.depth if stack == "blocks" else self.cfg.decoder_depth
        return (depth - 1 - int(i)) & 1

    @staticmethod
    def _scratch(w, key, s, n):
        return (w[key] if s == 0 or (key + "_b") not in w else w[key + "_b"])[:n]

    def _wgrad_layers(self, prefix, bufs, M, dim, w):
        """(dy, x_in, weight, bias, N_out, K_in) of the four linear layers of a block, as backward sees them."""
        hidden = bufs["hpre"].shape[1]
        s = self._bwd_set(prefix)
        g_lp = self._scratch(w, "g_lp", s, M * dim).view(M, dim)
        g_lp2 = self._scratch(w, "g_lp2", s, M * dim).view(M, dim)
        dh = self._scratch(w, "dh", s, M * hidden).view(M, hidden)
        dqkv = self._scratch(w, "dqkv", s, 3 * M * dim).view(M, 3 * dim)
        return [(g_lp, bufs["hact"], f"{prefix}.mlp.fc2", dim, hidden), (dh, bufs["ln2"], f"{prefix}.mlp.fc1", hidden, dim),
                (g_lp2, bufs["att"], f"{prefix}.attn.proj", dim, dim), (dqkv, bufs["ln1"], f"{prefix}.attn.qkv", 3 * dim, dim)]

    def _make_wgrad_group(self, prefix, bufs, M, dim, w, adamw=None, g16=None, side=None):
        st = self.store

        def dst(name, n_out, k_in):
            # fp32 into the flat gradient buffer, or (data-parallel runs with bf16 communication) straight into the bf16 mirror
            # the all-reduce sums: the same rounding the cast kernel applied to the stored fp32 value, one pass less over it
            if g16 is None:
                return dict(out_f32=st.grad(f"{name}.weight"))
            o = st.offsets[f"{name}.weight"]
            return dict(out=g16[o:o + n_out * k_in].view(n_out, k_in))
        layers = self._wgrad_layers(prefix, bufs, M, dim, w) + self._extra_wgrad_layers(prefix, M, w)
        import os
        # (experiments: SKYEMB_WGRAD_TILE_ENC / _DEC force the grouped launch's tile code for one stack)
        tile = int(os.environ.get("SKYEMB_WGRAD_TILE_DEC" if prefix.startswith("decoder") else "SKYEMB_WGRAD_TILE_ENC", "0"))
        # (the launch sits between the data-gradient chains of two blocks: its first problem touches the weights the next chain starts
        # with -- the block below's fc2 -- and this block's qkv data gradient, right in front of it, names nothing: _block_bwd)
        hint = self._pf("bwd", f"{prefix}.attn.qkv.weight", M)
        args = [ops.gemm_args(dy, x_in, M=n_out, N=k_in, K=M, a_layout=RC, b_layout=RC, lda=n_out, ldb=k_in,
                              colsum_a=st.grad(f"{name}.bias"), prefetch=hint if j == 0 else None, **dst(name, n_out, k_in))
                for j, (dy, x_in, name, n_out, k_in) in enumerate(layers)]
        grp = ops.GemmGroup(args, self.device, tile=tile, adamw=adamw, side=side, ln_bwd=self._norm1_side_record(prefix, bufs, M, dim, w))
        if grp.ok:
            grp.extra_layers = [name for _, _, name, _, _ in layers[4:]]
            if adamw is None and g16 is None:                 # (the plain group is planned first: it decides what the call sites skip)
                w.setdefault("folded_wgrads", set()).update(grp.extra_layers)
        return grp if grp.ok else None

    def _extra_wgrad_layers(self, prefix, M, w):
        """Single weight gradients that nothing in backward waits for, folded into a block's grouped launch as further problems
        (same token rows = same contraction length; round 5): `decoder_pred` into the first decoder block's launch, `decoder_embed`
        into the first encoder block's.  As launches of their own (a split-K GEMM + its reduce each) they were 25 and 14 us of the
        step's dependent chain.  -> [(dy, x_in, layer name, n_out, k_in)]; the names are recorded in w['folded_wgrads']."""
        import os
        if (os.environ.get("SKYEMB_FOLD_WGRADS", "1") == "0" or self._side is not None or "dpred" not in w
                or not self.fold_decoder_wgrads):
            # (fold_decoder_wgrads = False: a caller that runs backward_encoder / backward_embed WITHOUT backward_decoder -- the
            # downstream predictor, whose token count makes Me == Md -- would otherwise multiply an uninitialised dE every step)
            return []
        cfg = self.cfg
        out = []
        if prefix == f"decoder_blocks.{cfg.decoder_depth - 1}" and w["dpred"].shape[0] == M:
            out.append((w["dpred"], w["dlat_lp"], "decoder_pred", cfg.patch_dim, cfg.decoder_embed_dim))
        if prefix == f"blocks.{cfg.depth - 1}" and "dE" in w and w["dE"].shape[0] == M:
            out.append((w["dE"], w["lat_lp"], "decoder_embed", cfg.decoder_embed_dim, cfg.embed_dim))
        return out

    def _block_input(self, prefix, w):
        """fp32 residual stream a block reads (saved by the training forward)."""
        tag, i = prefix.rsplit(".", 1)
        return (w["xs"] if tag == "blocks" else w["xd"])[int(i)]

    def _norm1_side_record(self, prefix, bufs, M, dim, w):
        """The backward of the block's norm1 as a side job of its grouped weight-gradient launch (skyemb_gemm_group_attach_ln_bwd):
        it needs the qkv data gradient (in w['dln'] by then) and the residual gradient, not the weight gradients -- as a launch
        of its own it sat BEHIND the grouped launch, 7-11 us at ViT-B and 29 us at ViT-L per block on the step's critical path.
        Same buffers _block_bwd would hand ops.layernorm_bwd; None when the schedule cannot take it (weight gradients on a side
        stream: the next block would read the residual gradient before this launch is done; SKYEMB_LN_SIDE=0)."""
        import os
        if self._side is not None or os.environ.get("SKYEMB_LN_SIDE", "1") == "0" or "ln_parts" not in w:
            return None
        s = self._bwd_set(prefix)
        g = w["g"][:M * dim].view(M, dim)
        return dict(dy=w["dln"][:M * dim].view(M, dim), x=self._block_input(prefix, w), gamma=self.store.param(f"{prefix}.norm1.weight"),
                    mean=bufs["mean1"], rstd=bufs["rstd1"], g_in=g, g_out=g, g_lp=self._scratch(w, "g_lp", s ^ 1, M * dim).view(M, dim),
                    part=w["ln_parts"][f"{prefix}.norm1"], M=M, D=dim)

    # -- optimiser step fused into the weight-gradient launches (one process per replica: TrainStep(fused_adamw=True)) --
    def enable_fused_adamw(self, optimizer, on=True, side=None):
        """The grouped weight-gradient launches of the transformer blocks carry the AdamW step of the blocks' weight matrices
        (include/skyemb.h).  Epilogue form: a launch steps its OWN four matrices in its epilogue instead of storing their
        gradients (skyemb_gemm_group_plan_adamw: the gradient buffer is then not written for them).  Side form: a launch
        stores its gradients, and the launch that FOLLOWS it in the backward pass carries their step as a side job of extra
        workgroups (skyemb_gemm_group_plan_side_adamw: HBM-bound work beside the k-loops instead of behind them).  side = None:
        SKYEMB_ADAMW_SIDE, default "auto" -- the side form where the carrying launch leaves a quarter of the device's workgroup
        slots free (the 256 x 256 groups of a ViT-L block: 192 tiles for 256 units, mim_19 26.5 -> 25.5 ms; the ViT-B decoder's
        384 tiles for 512 slots), the epilogue form elsewhere (the ViT-B encoder's launches fill the chip and the side form
        LOSES there).  Either way fused_adamw_ranges(workspace) are the slices of the flat buffers these launches update -- the
        caller runs the ordinary AdamW on the rest (embeddings, biases, LayerNorms, the single weight gradients)."""
        self._fused_adamw = None
        if not on:
            return
        import os
        # placement policy: "auto" (default) = side jobs where the carrying launch leaves compute units idle (256 x 256 tiles:
        # ViT-L), epilogue elsewhere; "1" / True = side jobs everywhere; "0" / False = epilogue everywhere; "dec" / "enc" =
        # side jobs carried by the decoder's / encoder's launches only (experiments)
        self._adamw_side = os.environ.get("SKYEMB_ADAMW_SIDE", "auto") if side is None else ("1" if side is True else "0" if side is False else str(side))
        assert self._adamw_side in ("auto", "0", "1", "dec", "enc"), self._adamw_side
        self._adamw_side_blocks = int(os.environ.get("SKYEMB_SIDE_BLOCKS", "256"))
        assert self.dtype in ops.LP_DTYPES, "the fused optimiser step exists on the 16-bit paths (grouped weight gradients)"
        from ._lib import AdamwDesc
        st = self.store
        b1, b2 = optimizer.defaults["betas"]
        d = AdamwDesc()
        d.g_base, d.p, d.m, d.v, d.p_lp = (t.data_ptr() for t in (st.g, st.p, st.m, st.v, st.p_lp))
        d.hyper = optimizer.hyper_device.data_ptr()
        d.n_decay = st.n_decay
        d.beta1, d.beta2, d.eps = b1, b2, optimizer.defaults["eps"]
        d.weight_decay, d.grad_scale = optimizer.param_groups[1]["weight_decay"], optimizer.grad_scale
        self._fused_adamw = d
        for w in self._ws.values():                           # workspaces built before the switch
            if "wgrad_groups" in w:
                self._build_adamw_groups(w)

    # -- data-parallel runs with bf16 gradient communication: the block weights' gradients go straight into the bf16 mirror --
    def enable_grad_mirror(self, g16):
        """The grouped weight-gradient launches write bf16 into `g16` (the flat mirror the all-reduce sums) instead of fp32 into
        the gradient buffer; grad_mirror_ranges(workspace) are the slices they cover -- the caller casts only the rest."""
        self._g16 = g16
        for w in self._ws.values():
            if "wgrad_groups" in w:
                self._build_variant_groups(w, "g16")

    def grad_mirror_ranges(self, w):
        return list(w.get("g16_ranges", []))

    def fused_adamw_ranges(self, w):
        """Slices of the flat buffers the fused launches of workspace `w` update (merged, ascending)."""
        return list(w.get("fused_ranges", []))

    def _build_adamw_groups(self, w):
        self._build_variant_groups(w, "adamw")

    def _wgrad_launch_order(self, w):
        """Prefixes of the blocks in the order backward() launches their grouped weight gradients."""
        order = [f"decoder_blocks.{i}" for i in reversed(range(len(w.get("dec", []))))]
        return order + [f"blocks.{i}" for i in reversed(range(self.cfg.depth))]

    def _block_weight_span(self, prefix):
        """[lo, hi) of the flat buffers holding a block's four weight matrices (layout order keeps them together)."""
        st = self.store
        spans = sorted((st.offsets[f"{prefix}.{n}.weight"], st.offsets[f"{prefix}.{n}.weight"] + _pad8(int(np.prod(st.shapes[f"{prefix}.{n}.weight"]))))
                       for n in ("attn.qkv", "attn.proj", "mlp.fc1", "mlp.fc2"))
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:])), "a block's weight matrices are not contiguous in the flat buffers"
        return spans[0][0], spans[-1][1]

    def _build_variant_groups(self, w, kind):
        """A second set of the blocks' grouped weight-gradient launches: kind 'adamw' (optimiser step carried by the launches -- in
        the epilogue, or as the side job of the following launch: enable_fused_adamw; ranges in w['fused_ranges']) or 'g16' (bf16
        output into the communication mirror; w['g16_ranges'])."""
        st = self.store
        groups, spans = {}, []
        order = [p for p in self._wgrad_launch_order(w) if w["wgrad_groups"].get(p) is not None]
        # carry[k]: launch k carries the step of launch k-1's block as a side job (launch k-1 then STORES its gradients)
        carry = [False] * len(order)
        if kind == "adamw":
            mode = getattr(self, "_adamw_side", "0")
            ncu = torch.cuda.get_device_properties(self.device).multi_processor_count
            for k in range(1, len(order)):
                grp0 = w["wgrad_groups"][order[k]]
                info = grp0.info
                # "auto": the carrying launch must leave a quarter of the device's workgroup slots free (slots per compute unit
                # follow from the tile's LDS ring: one 256 x 256, two 128 x 128 / 128 x 64, three 64 x 64).  Measured (bench.py
                # extra.optimizer_placement): ViT-L, 192 tiles of 256 x 256 on 256 units: 26.5 -> 25.5 ms per step; ViT-B decoder,
                # 384 tiles of 128 x 64 on 512 slots: 5.24 -> 5.21; ViT-B encoder, 440 of 512: side jobs LOSE (5.24 -> 5.47 with
                # every launch carrying one: they start when the tiles end, and move 34 instead of 26 bytes per parameter)
                slots = ncu * {256256: 1, 128128: 2, 9128128: 1, 128064: 2, 64064: 3}.get(info.tile, 1)
                carry[k] = (mode == "1" or (mode == "auto" and grp0.tile_blocks <= 0.76 * slots) or
                            (mode == "dec" and order[k].startswith("decoder_blocks")) or (mode == "enc" and order[k].startswith("blocks")))
        w["adamw_side_launches"] = sum(carry)
        for k, prefix in enumerate(order):
            tag, i = prefix.rsplit(".", 1)
            bufs = (w["enc"] if tag == "blocks" else w["dec"])[int(i)]
            M, dim = bufs["ln1"].shape
            stores = k + 1 < len(order) and carry[k + 1]
            if kind == "adamw" and (carry[k] or stores):
                lo, hi = self._block_weight_span(order[k - 1]) if carry[k] else (0, 0)
                side = (not stores, lo, hi, self._adamw_side_blocks if hi > lo else 0)
                grp = self._make_wgrad_group(prefix, bufs, M, dim, w, adamw=self._fused_adamw, side=side)
                side_mode = True
            elif kind == "adamw":
                grp = self._make_wgrad_group(prefix, bufs, M, dim, w, adamw=self._fused_adamw)
                side_mode = False
            else:
                grp = self._make_wgrad_group(prefix, bufs, M, dim, w, g16=self._g16)
                side_mode = False
            if grp is None:
                assert not side_mode, f"side optimiser jobs: the grouped launch of {prefix} could not be planned"
                continue
            groups[prefix] = grp
            for name in ("attn.qkv", "attn.proj", "mlp.fc1", "mlp.fc2"):
                o = st.offsets[f"{prefix}.{name}.weight"]
                spans.append((o, o + _pad8(int(np.prod(st.shapes[f"{prefix}.{name}.weight"])))))
            # folded single weight gradients: written to the mirror / stepped in the epilogue with the launch's own tiles; where the
            # launch only STORES gradients they stay with the ordinary AdamW launch like every tensor outside the blocks
            if kind != "adamw" or not stores:
                for name in getattr(grp, "extra_layers", []):
                    o = st.offsets[f"{name}.weight"]
                    spans.append((o, o + _pad8(int(np.prod(st.shapes[f"{name}.weight"])))))
        spans.sort()
        merged = []
        for s_, e_ in spans:
            if merged and merged[-1][1] == s_:
                merged[-1] = (merged[-1][0], e_)
            else:
                merged.append((s_, e_))
        w["wgrad_groups_adamw" if kind == "adamw" else "wgrad_groups_g16"] = groups
        w["fused_ranges" if kind == "adamw" else "g16_ranges"] = merged

    def _block_bwd(self, x_in, bufs, prefix, M, dim, heads, Bsz, N, g, g_lp, w):
        """g holds d(block output) on entry and d(block input) on exit (fp32); its compute-dtype copy is read from the
        block's scratch set and written to the other set (the next block's)."""
        hd = dim // heads
        hidden = bufs["hpre"].shape[1]
        s = self._bwd_set(prefix)
        g_lp = self._scratch(w, "g_lp", s, M * dim).view(M, dim)
        g_lp_next = self._scratch(w, "g_lp", s ^ 1, M * dim).view(M, dim)
        dh = self._scratch(w, "dh", s, M * hidden).view(M, hidden)
        dqkv = self._scratch(w, "dqkv", s, 3 * M * dim).view(M, 3 * dim)
        dln = w["dln"][:M * dim].view(M, dim)
        datt = w["datt"][:M * dim].view(M, dim)
        # (only inside a TrainStep that owns the optimiser step: `_fused_active` is raised around ITS launches, so that
        # engine.backward() called by anybody else -- loss.backward() of the module API, tests -- stores plain gradients)
        fused = (getattr(self, "_fused_active", False) and getattr(self, "_fused_adamw", None) is not None
                 and prefix in w.get("wgrad_groups_adamw", {}))
        mirror = (not fused and getattr(self, "_g16_active", False) and getattr(self, "_g16", None) is not None
                  and prefix in w.get("wgrad_groups_g16", {}))
        group = (w["wgrad_groups_adamw"] if fused else w["wgrad_groups_g16"] if mirror else w["wgrad_groups"]).get(prefix)
        single = group is None                      # weight gradients launch by launch (fp32 mode)
        g_mid = g_lp if single else self._scratch(w, "g_lp2", s, M * dim).view(M, dim)
        # MLP: x_out = xmid + fc2(gelu(fc1(ln2(xmid))))
        self._linear_bwd(g_lp, bufs["hact"], f"{prefix}.mlp.fc2.weight", f"{prefix}.mlp.fc2.bias", M, dim, hidden, w,
                         dx_out=dh, dx_act=ACT_DGELU, dx_aux=bufs["hpre"], wgrad=single)
        self._linear_bwd(dh, bufs["ln2"], f"{prefix}.mlp.fc1.weight", f"{prefix}.mlp.fc1.bias", M, hidden, dim, w,
                         dx_out=dln, wgrad=single)
        self._ln_bwd(dln, bufs["xmid"], f"{prefix}.norm2", bufs["mean2"], bufs["rstd2"], g, g, g_mid, M, dim, w)
        # attention: xmid = x_in + proj(mha(qkv(ln1(x_in))))
        self._linear_bwd(g_mid, bufs["att"], f"{prefix}.attn.proj.weight", f"{prefix}.attn.proj.bias", M, dim, dim, w,
                         dx_out=datt, wgrad=single)
        self._before_write(dqkv)
        ops.mha_bwd(bufs["qkv"], datt, dqkv, Bsz, N, heads, hd)
        self._linear_bwd(dqkv, bufs["ln1"], f"{prefix}.attn.qkv.weight", f"{prefix}.attn.qkv.bias", M, 3 * dim, dim, w,
                         dx_out=dln, wgrad=single, prefetch="chain" if group is None else None)
        prev_done = self._group_done
        if group is not None:
            # all four dW / db of the block in ONE launch.  With the side stream it runs under the next block's dgrad
            # chain (a bandwidth-bound launch next to a chain of latency-bound ones); its four dy live in this block's
            # scratch set, which nothing writes before the norm1 backward of the NEXT block (waited for below)
            if self._side is None:
                group.launch()
            else:
                ready = torch.cuda.Event()
                ready.record()
                self._side.wait_event(ready)
                with torch.cuda.stream(self._side):
                    group.launch()
                    self._group_done = torch.cuda.Event()
                    self._group_done.record()
        if prev_done is not None:
            torch.cuda.current_stream().wait_event(prev_done)   # the previous block's launch still reads g_lp_next
        if group is not None and group.ln_side:
            # norm1's backward rode in the grouped launch (side workgroups): only the stage's reduce table moves on
            assert w["ln_index"][f"{prefix}.norm1"] == self._ln_first + self._ln_count, "LayerNorm backward visited out of table order"
            assert x_in.data_ptr() == self._block_input(prefix, w).data_ptr()
            self._ln_count += 1
        else:
            self._ln_bwd(dln, x_in, f"{prefix}.norm1", bufs["mean1"], bufs["rstd1"], g, g, g_lp_next, M, dim, w)

    def _last_key(self):
        """Workspace key of the last forward_train() call."""
        assert self._last is not None, "no forward_train() yet"
        return (self._last[1], self._last[2], True)

    def _bwd_ctx(self):
        assert self._last is not None, "backward() without forward_train()"
        imgs, B, keep = self._last
        cfg = self.cfg
        L, E = cfg.num_patches, cfg.num_extra_tokens
        return imgs, B, keep, E + keep, E + L, self._ws[(B, keep, True)]

    def backward_decoder(self):
        """Stage 0 of backward: decoder_pred ... decoder_embed (+ mask_token); leaves d latent in w['dln']."""
        imgs, B, keep, Ne, Nd, w = self._bwd_ctx()
        cfg, st = self.cfg, self.store
        L, D, Dd, pv = cfg.num_patches, cfg.embed_dim, cfg.decoder_embed_dim, cfg.patch_dim
        Me, Md = B * Ne, B * Nd
        self._ln_first = self._ln_count = 0
        dln = w["dln"][:Md * Dd].view(Md, Dd)
        # (decoder_pred's weight gradient rides in the first decoder block's grouped launch when that launch exists: _extra_wgrad_layers)
        self._linear_bwd(w["dpred"], w["dlat_lp"], "decoder_pred.weight", "decoder_pred.bias", Md, pv, Dd, w, dx_out=dln,
                         wgrad="decoder_pred" not in w.get("folded_wgrads", ()))
        g = w["g"][:Md * Dd].view(Md, Dd)
        g_lp = w["g_lp"][:Md * Dd].view(Md, Dd)
        self._ln_bwd(dln, w["xd"][-1], "decoder_norm", w["dlat_mean"], w["dlat_rstd"], None, g, g_lp, Md, Dd, w)
        for i in reversed(range(cfg.decoder_depth)):
            self._block_bwd(w["xd"][i], w["dec"][i], f"decoder_blocks.{i}", Md, Dd, cfg.decoder_num_heads, B, Nd, g,
                            g_lp, w)
        # mask token, decoder_embed (g = d xd[0])
        ops.rowsum_select(g, Dd, w["mask"], cfg.num_extra_tokens, L, Nd, B * L, Dd, w["rs_part"], st.grad("mask_token").view(Dd))
        ops.gather_rows(g, w["dec_dst"], None, w["dE"], Me, Dd)
        dln_e = w["dln"][:Me * D].view(Me, D)
        self._linear_bwd(w["dE"], w["lat_lp"], "decoder_embed.weight", "decoder_embed.bias", Me, Dd, D, w, dx_out=dln_e,
                         wgrad="decoder_embed" not in w.get("folded_wgrads", ()))     # (... in the first encoder block's)
        self._end_stage(w)

    def backward_encoder(self, hi=None, lo=0):
        """Encoder blocks hi-1 ... lo (hi=None: from the top, including the final norm)."""
        imgs, B, keep, Ne, Nd, w = self._bwd_ctx()
        cfg = self.cfg
        D = cfg.embed_dim
        Me = B * Ne
        g = w["g"][:Me * D].view(Me, D)
        g_lp = w["g_lp"][:Me * D].view(Me, D)
        if hi is None:
            hi = cfg.depth
            dln_e = w["dln"][:Me * D].view(Me, D)
            self._ln_bwd(dln_e, w["xs"][cfg.depth], "norm", w["lat_mean"], w["lat_rstd"], None, g, g_lp, Me, D, w)
        for i in reversed(range(lo, hi)):
            self._block_bwd(w["xs"][i], w["enc"][i], f"blocks.{i}", Me, D, cfg.num_heads, B, Ne, g, g_lp, w)
        self._end_stage(w)

    def backward_embed(self):
        """Last stage: cls token, patch embedding, patch_mask_values (g = d xs[0])."""
        imgs, B, keep, Ne, Nd, w = self._bwd_ctx()
        cfg, st = self.cfg, self.store
        D, pv = cfg.embed_dim, cfg.patch_dim
        Me = B * Ne
        g = w["g"][:Me * D].view(Me, D)
        ops.rowsum_select(g, D, None, 0, 1, Ne, B, D, w["rs_part"], st.grad("cls_token").view(D))
        if cfg.ra_dec:
            G = st.grad
            ops.radec_token_bwd(g.view(-1)[D:], Ne * D, st.param("ra_dec_embed.neural_network.last_layer.weight"), w["sh"],
                                w["z"], w["dz"], G("ra_dec_embed.neural_network.layers.0.weight"),
                                G("ra_dec_embed.neural_network.layers.0.bias"), G("ra_dec_embed.neural_network.last_layer.weight"),
                                G("ra_dec_embed.neural_network.last_layer.bias"), B, D)
        ops.gather_rows(g, w["pe_dst"], None, w["dT"], B * keep, D)
        self._wgrad(w["dT"], w["patches"], D, pv, B * keep, st.grad("patch_embed.proj.weight"),
                    st.grad("patch_embed.proj.bias"), w)
        ops.gemm(w["dT"], st.lp("patch_embed.proj.weight"), M=B * keep, N=pv, K=D, a_layout=KC, b_layout=RC, lda=D,
                 ldb=pv, out_f32=w["drows"])
        ops.patch_gather_bwd_pmv(imgs, w["ids_keep"], w["drows"], w["pmv_part"], st.grad("patch_mask_values"),
                                 cfg.patch_size, keep)
        self._end_stage(w, last=True)

    def backward(self):
        """Gradients of the last :meth:`forward_train` loss into the flat ``g`` buffer (every
        trainable tensor is written exactly once, so no zeroing pass is needed)."""
        self.backward_decoder()
        self.backward_encoder()
        self.backward_embed()

    def backward_stages(self, n_encoder_groups=3):
        """Backward cut into stages for gradient all-reduce overlap (one process per GPU): a list of
        (callable, [(start, end), ...]) -- see :func:`stage_gradient_ranges`."""
        cfg = self.cfg
        groups, ranges = stage_gradient_ranges(self.store, cfg, n_encoder_groups)
        stages = [(self.backward_decoder, ranges[0])]
        for k, (hi, lo) in enumerate(groups):
            stages.append(((lambda h=hi, l=lo, top=(k == 0): self.backward_encoder(None if top else h, l)), ranges[1 + k]))
        stages.append((self.backward_embed, ranges[-1]))
        return stages

    # ------------------------------------------------------------------ accounting
    def flops_per_image(self, mask_ratio=0.75):
        """(executed, reference-algorithmic) forward+backward FLOPs per image, 3x-forward
        convention of SURVEY.md §8: the reference embeds all L patches before masking, the
        build only the kept ones (identical results)."""
        cfg = self.cfg
        L, D, Dd, pv = cfg.num_patches, cfg.embed_dim, cfg.decoder_embed_dim, cfg.patch_dim
        keep = int(L * (1 - mask_ratio))
        Ne, Nd = cfg.num_extra_tokens + keep, cfg.num_extra_tokens + L
        r = cfg.mlp_ratio

        def blocks(n, d, depth, heads):
            lin = 2 * n * d * (3 * d + d + 2 * r * d)
            att = 2 * 2 * n * n * d
            return depth * (lin + att)
        common = blocks(Ne, D, cfg.depth, cfg.num_heads) + 2 * Ne * D * Dd + blocks(Nd, Dd, cfg.decoder_depth,
                                                                                  cfg.decoder_num_heads) + 2 * Nd * Dd * pv
        executed = 3 * (common + 2 * keep * pv * D)
        algorithmic = 3 * (common + 2 * L * pv * D)
        return executed, algorithmic
