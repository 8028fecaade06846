"""Prediction head of the downstream ViT (utils/vit.py:302-310, 390-393 + timm's ``VisionTransformer.forward_head``) on the
library's kernels, forward AND backward: pooling of the encoder's tokens -- class token, mean of the patch tokens + ``fc_norm``,
or timm's ``AttentionPoolLatent`` with two heads (what every shipped predictor ini asks for: ``global_pool = map``) -- and the
linear classifier / regressor.

Kernels: ``skyemb_gemm`` (kv projection over every token, proj, fc1 + GELU, fc2 + residual, the head; data and weight
gradients with the bias gradient riding in the weight-gradient launch), ``skyemb_attnpool_q / _fwd / _bwd / _q_bwd`` (one
learned query per head, one wave per (sample, head)), ``skyemb_layernorm_fwd / _bwd``, ``skyemb_cast``; the optimiser steps
the head's flat buffers with ``skyemb_adamw`` (utils.vit.PredictorOptimizer).  Torch only slices / copies rows here (class
token, token mean) and owns the memory; the loss on the [B, num_classes] predictions stays in the caller
(utils/predictor_training_fns.py).

Layout: like the engine's ParamStore -- one flat fp32 buffer ``p`` with named views (timm's tensor names), ``g`` / ``m`` /
``v`` beside it and the compute-dtype shadow ``p_lp`` the GEMMs read.  ``head.weight`` / ``head.bias`` are stored with their
class count padded to a multiple of 8 (zero rows: the GEMM's vector width); the named views show the real rows only.
"""
from __future__ import annotations

from collections import OrderedDict

import numpy as np
import torch

from . import ops
from .ops import ACT_DGELU, ACT_GELU, KC, RC

POOL_HEADS = 2          # utils/vit.py:305: AttentionPoolLatent(embed_dim, num_heads=2, ...)


def _pad8(n):
    return (n + 7) // 8 * 8


class HeadStore:
    """Flat parameter / gradient / Adam-state buffers of the head with named views (cf. engine.ParamStore)."""

    def __init__(self, shapes: "OrderedDict[str, tuple]", alloc: dict, device, lp_dtype):
        """shapes: the tensors as the state dict shows them; alloc: element counts that differ from prod(shape) (padded rows)."""
        self.shapes = OrderedDict(shapes)
        self.offsets, self.sizes = {}, {}
        off = 0
        for name, shape in self.shapes.items():
            self.offsets[name] = off
            self.sizes[name] = _pad8(int(alloc.get(name, int(np.prod(shape)))))
            off += self.sizes[name]
        self.n = max(off, 8)
        f32 = dict(device=device, dtype=torch.float32)
        self.p, self.g = torch.zeros(self.n, **f32), torch.zeros(self.n, **f32)
        self.m, self.v = torch.zeros(self.n, **f32), torch.zeros(self.n, **f32)
        self.p_lp = torch.zeros(self.n, device=device, dtype=lp_dtype)

    def _view(self, buf, name):
        o = self.offsets[name]
        return buf[o:o + int(np.prod(self.shapes[name]))].view(self.shapes[name])

    def raw(self, buf, name):
        """The tensor's whole padded run of `buf` (what the GEMMs and the optimiser address)."""
        o = self.offsets[name]
        return buf[o:o + self.sizes[name]]

    def param(self, name):
        return self._view(self.p, name)

    def grad(self, name):
        return self._view(self.g, name)

    def refresh_lp(self):
        ops.cast(self.p, self.p_lp, self.n)


class PredictorHead:
    """global_pool in {'', 'token', 'avg', 'map'} + Linear(D, num_classes); see the module docstring."""

    def __init__(self, D, mlp_ratio, eps, global_pool, num_classes, device, compute_dtype, gen, splitk_ws=None):
        self.D, self.hidden, self.eps = int(D), int(D * mlp_ratio), float(eps)
        self.pool, self.C, self.Cp = global_pool, int(num_classes), _pad8(int(num_classes)) if num_classes > 0 else 0
        self.device, self.dtype, self.code = torch.device(device), compute_dtype, ops.dtype_code(compute_dtype)
        self._splitk_ws = splitk_ws
        D, hid = self.D, self.hidden
        shapes, alloc = OrderedDict(), {}
        if global_pool == 'avg':
            shapes["fc_norm.weight"], shapes["fc_norm.bias"] = (D,), (D,)
        if global_pool == 'map':
            assert D % POOL_HEADS == 0 and (D // POOL_HEADS) % 4 == 0, "attention pool: embed_dim must split into two heads of 4k columns"
            shapes["attn_pool.latent"] = (1, 1, D)
            for name, (o, i) in (("q", (D, D)), ("kv", (2 * D, D)), ("proj", (D, D))):
                shapes[f"attn_pool.{name}.weight"], shapes[f"attn_pool.{name}.bias"] = (o, i), (o,)
            shapes["attn_pool.norm.weight"], shapes["attn_pool.norm.bias"] = (D,), (D,)
            shapes["attn_pool.mlp.fc1.weight"], shapes["attn_pool.mlp.fc1.bias"] = (hid, D), (hid,)
            shapes["attn_pool.mlp.fc2.weight"], shapes["attn_pool.mlp.fc2.bias"] = (D, hid), (D,)
        if self.C > 0:
            shapes["head.weight"], shapes["head.bias"] = (self.C, D), (self.C,)
            alloc["head.weight"], alloc["head.bias"] = self.Cp * D, self.Cp
        self.store = HeadStore(shapes, alloc, self.device, compute_dtype)
        self.tensors = OrderedDict((k, self.store.param(k)) for k in shapes)
        self._ws = {}
        # timm VisionTransformer(drop_rate=...): `head_drop`, an nn.Dropout on the pooled features in front of the classifier
        # (forward_head: fc_norm -> head_drop -> head), active in training mode.  The caller (utils.vit.VisionTransformer) sets both.
        self.drop_rate, self.training = 0.0, False
        self._init(gen)

    def _init(self, gen):
        """timm's initialisation of these modules: trunc-normal(0.02) Linear weights, zero biases, LayerNorm 1 / 0, trunc-normal
        latent with std D^-0.5 (AttentionPoolLatent.init_weights)."""
        D = self.D
        for name, t in self.tensors.items():
            if name == "attn_pool.latent":
                s = D ** -0.5
                t.copy_((torch.randn(t.shape, generator=gen) * s).clamp_(-2 * s, 2 * s))
            elif name.endswith("norm.weight"):
                t.fill_(1.0)
            elif name.endswith(".bias"):
                t.zero_()
            else:
                t.copy_((torch.randn(t.shape, generator=gen) * 0.02).clamp_(-0.04, 0.04))
        self.store.refresh_lp()

    # ---- buffers --------------------------------------------------------------------------------------------------------------
    def _workspace(self, B, Ne):
        key = (B, Ne)
        if key in self._ws:
            return self._ws[key]
        D, hid, dev = self.D, self.hidden, self.device
        f32, lp = dict(device=dev, dtype=torch.float32), dict(device=dev, dtype=self.dtype)
        M = B * Ne
        w = {"feat": torch.zeros(B, D, **f32), "z": torch.zeros(B, D, **f32), "z_lp": torch.zeros(B, D, **lp),
             "gz": torch.zeros(B, D, **f32), "gz_lp": torch.zeros(B, D, **lp)}
        if self.C > 0:
            w["logits"] = torch.zeros(B, self.Cp, **f32)
            w["dlog"] = torch.zeros(B, self.Cp, **lp)
        nb = ops.layernorm_bwd_blocks(B)
        if self.pool == 'avg':
            w["mean"], w["rstd"] = torch.zeros(B, **f32), torch.zeros(B, **f32)
            w["part"] = torch.zeros(2, nb, D, **f32)
            w["dfeat"] = torch.zeros(B, D, **f32)
        if self.pool == 'map':
            H = POOL_HEADS
            w["x_lp"] = torch.zeros(M, D, **lp)                            # (inference on a caller's token tensor)
            w["kv"], w["dkv"] = torch.zeros(M, 2 * D, **lp), torch.zeros(M, 2 * D, **lp)
            w["q"], w["prob"] = torch.zeros(D, **f32), torch.zeros(B, H, Ne, **f32)
            w["o"], w["do"] = torch.zeros(B, D, **lp), torch.zeros(B, D, **lp)
            w["y"] = torch.zeros(B, D, **f32)
            w["ln"], w["dln"] = torch.zeros(B, D, **lp), torch.zeros(B, D, **lp)
            w["mean"], w["rstd"] = torch.zeros(B, **f32), torch.zeros(B, **f32)
            w["hpre"], w["hact"], w["dh"] = (torch.zeros(B, hid, **lp) for _ in range(3))
            w["gy"], w["gy_lp"] = torch.zeros(B, D, **f32), torch.zeros(B, D, **lp)
            w["dq"], w["dq_ws"] = torch.zeros(B, D, **f32), torch.zeros(D, **f32)
            w["part"] = torch.zeros(2, nb, D, **f32)
        self._ws[key] = w
        return w

    # ---- forward --------------------------------------------------------------------------------------------------------------
    def _pool_map_fwd(self, x_lp, w, B, Ne):
        """timm AttentionPoolLatent.forward (latent_len 1, pool 'token', no q / k norm, no positional table): x_lp [B*Ne, D] in the
        compute dtype (the final norm's output) -> w['z'] [B, D] fp32."""
        D, hid, H = self.D, self.hidden, POOL_HEADS
        P, LP = self.store.param, (lambda n: self.store.raw(self.store.p_lp, n))
        M = B * Ne
        ops.gemm(x_lp, LP("attn_pool.kv.weight"), M=M, N=2 * D, K=D, bias=P("attn_pool.kv.bias"), out=w["kv"])
        ops.attnpool_q(P("attn_pool.latent"), P("attn_pool.q.weight"), P("attn_pool.q.bias"), w["q"])
        ops.attnpool_fwd(w["q"], w["kv"], w["o"], w["prob"], B, Ne, H, D // H)
        ops.gemm(w["o"], LP("attn_pool.proj.weight"), M=B, N=D, K=D, bias=P("attn_pool.proj.bias"), out_f32=w["y"])
        ops.layernorm_fwd(w["y"], P("attn_pool.norm.weight"), P("attn_pool.norm.bias"), w["ln"], w["mean"], w["rstd"], B, D, self.eps)
        ops.gemm(w["ln"], LP("attn_pool.mlp.fc1.weight"), M=B, N=hid, K=D, bias=P("attn_pool.mlp.fc1.bias"), act=ACT_GELU,
                 out=w["hact"], out2=w["hpre"])
        ops.gemm(w["hact"], LP("attn_pool.mlp.fc2.weight"), M=B, N=D, K=hid, bias=P("attn_pool.mlp.fc2.bias"), resid=w["y"], ldr=D,
                 out_f32=w["z"], ws=self._splitk_ws)

    def forward(self, B, Ne, tokens_lp=None, feat=None, pre_logits=False):
        """tokens_lp [B*Ne, D] (compute dtype; 'map') or feat [B, D] fp32 (class-token rows / the patch-token mean) -> predictions
        [B, num_classes] fp32 (a view of the workspace), or the pooled features with pre_logits / num_classes == 0."""
        w = self._workspace(B, Ne)
        D = self.D
        P = self.store.param
        if self.pool == 'map':
            self._pool_map_fwd(tokens_lp, w, B, Ne)
        elif self.pool == 'avg':
            w["feat"].copy_(feat)
            ops.layernorm_fwd(w["feat"], P("fc_norm.weight"), P("fc_norm.bias"), w["z_lp"], w["mean"], w["rstd"], B, D, self.eps, y32=w["z"])
        else:
            w["z"].copy_(feat)
        w["drop"] = None
        if self.training and self.drop_rate > 0.0:
            # inverted dropout as nn.Dropout applies it; the keep mask is drawn with torch's generator on the device (the individual
            # draws differ from the reference's stream, the distribution does not) and kept for backward.  [B, D] elementwise: glue.
            keep = 1.0 - self.drop_rate
            w["drop"] = (torch.rand(B, D, device=self.device) < keep).to(torch.float32).div_(keep)
            w["z"].mul_(w["drop"])
        if pre_logits or self.C <= 0:
            return w["z"]
        if self.pool != 'avg' or w["drop"] is not None:
            ops.cast(w["z"], w["z_lp"], B * D)
        st = self.store
        ops.gemm(w["z_lp"], st.raw(st.p_lp, "head.weight"), M=B, N=self.Cp, K=D, bias=st.raw(st.p, "head.bias"), out_f32=w["logits"])
        return w["logits"][:, :self.C]

    # ---- backward -------------------------------------------------------------------------------------------------------------
    def backward(self, dlogits, B, Ne, tokens_lp=None, dtokens_lp=None):
        """d loss / d predictions [B, num_classes] -> every head gradient in ``store.g``; returns d features [B, D] fp32 (token /
        avg: gradient of the pooled-from rows) or, for 'map', writes d loss / d every token into dtokens_lp [B*Ne, D] (compute
        dtype: the final norm's incoming gradient in the engine) and returns None."""
        w = self._workspace(B, Ne)
        st = self.store
        D, hid, H = self.D, self.hidden, POOL_HEADS
        P, G = st.param, st.grad
        LP = lambda n: st.raw(st.p_lp, n)                                  # noqa: E731
        w["dlog"].zero_()
        w["dlog"][:, :self.C].copy_(dlogits)
        # head: dW = dlogits^T z, db = column sums, dz = dlogits W
        ops.gemm(w["dlog"], w["z_lp"], M=self.Cp, N=D, K=B, a_layout=RC, b_layout=RC, lda=self.Cp, ldb=D,
                 out_f32=st.raw(st.g, "head.weight").view(self.Cp, D), colsum_a=st.raw(st.g, "head.bias"))
        ops.gemm(w["dlog"], LP("head.weight"), M=B, N=D, K=self.Cp, a_layout=KC, b_layout=RC, lda=self.Cp, ldb=D,
                 out_f32=w["gz"], out=w["gz_lp"])
        if w.get("drop") is not None:                                      # d loss / d (features before the dropout)
            w["gz"].mul_(w["drop"])
            ops.cast(w["gz"], w["gz_lp"], B * D)
        if self.pool == 'avg':
            ops.layernorm_bwd(w["gz_lp"], w["feat"], P("fc_norm.weight"), w["mean"], w["rstd"], None, w["dfeat"], None, w["part"],
                              G("fc_norm.weight"), G("fc_norm.bias"), B, D, self.code)
            return w["dfeat"]
        if self.pool != 'map':
            return w["gz"]
        M = B * Ne

        def linear_bwd(dy, x_in, name, n_out, k_in, rows, **dx):
            """dW [n_out, k_in] = dy^T x (+ db = column sums of dy) and d x = dy W into dx['out'] with the fused epilogue asked for."""
            ops.gemm(dy, x_in, M=n_out, N=k_in, K=rows, a_layout=RC, b_layout=RC, lda=n_out, ldb=k_in, out_f32=G(f"{name}.weight"),
                     colsum_a=G(f"{name}.bias"), ws=self._splitk_ws)
            ops.gemm(dy, LP(f"{name}.weight"), M=rows, N=k_in, K=n_out, a_layout=KC, b_layout=RC, lda=n_out, ldb=k_in,
                     ws=self._splitk_ws, **dx)
        # z = y + fc2(gelu(fc1(norm(y))))
        linear_bwd(w["gz_lp"], w["hact"], "attn_pool.mlp.fc2", D, hid, B, act=ACT_DGELU, aux=w["hpre"], ldaux=hid, out=w["dh"])
        linear_bwd(w["dh"], w["ln"], "attn_pool.mlp.fc1", hid, D, B, out=w["dln"])
        ops.layernorm_bwd(w["dln"], w["y"], P("attn_pool.norm.weight"), w["mean"], w["rstd"], w["gz"], w["gy"], w["gy_lp"], w["part"],
                          G("attn_pool.norm.weight"), G("attn_pool.norm.bias"), B, D, self.code)
        # y = proj(pool(q, kv(tokens)))
        linear_bwd(w["gy_lp"], w["o"], "attn_pool.proj", D, D, B, out=w["do"])
        ops.attnpool_bwd(w["q"], w["kv"], w["do"], w["prob"], w["dkv"], w["dq"], B, Ne, H, D // H)
        ops.attnpool_q_bwd(w["dq"], P("attn_pool.latent"), P("attn_pool.q.weight"), G("attn_pool.q.weight"), G("attn_pool.q.bias"),
                           G("attn_pool.latent"), w["dq_ws"])
        linear_bwd(w["dkv"], tokens_lp, "attn_pool.kv", 2 * D, D, M, out=dtokens_lp)
        return None
