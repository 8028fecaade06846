"""utils/vit.py mirror: the downstream ViT -- a pre-trained MAE / SimMIM encoder with a prediction head -- built, loaded and
TRAINED through the reference's API (``build_model`` with ``build_optimizer``, ``load_model``, ``model.module.*``).

The encoder (patch embedding, Blocks, final norm) runs forward AND backward in the HIP engine (``MAEEngine``: the same kernels
as pre-training; tokens in raster order, nothing masked), and so does the head -- pooling, ``fc_norm``, one Linear
(``sky_embeddings_amd.predictor_head.PredictorHead``: GEMM, attention-pool, LayerNorm and AdamW kernels of the library on the
head's own flat buffers).  One autograd node joins the predictions to the caller's loss: its backward runs the head's backward
schedule and then the engine's.
Training methods of utils/vit.py:134-172: ``ft`` (fine-tuning with layer-wise lr decay), ``lp`` (linear probe: final norm,
``fc_norm`` and head only; the encoder backward is not run at all), anything else "fully supervised" (timm's weight-decay split,
one lr); schedule = the LinearLR the reference ends up with -- starting from what its discarded OneCycleLR left in the
optimiser: lr = init_lr / 25, beta1 = 0.95 (utils/vit.py:174-186, ``apply_onecycle_side_effects``).
Pooling: ``token`` (class token), ``avg`` (mean of the patch tokens + ``fc_norm``; timm then has no final norm), ``map`` (timm's
``AttentionPoolLatent`` with two heads, as every shipped predictor config asks for: one learned query over the encoder's tokens,
projection, LayerNorm + MLP residual; its gradient with respect to ALL tokens enters the engine's backward as the final norm's
incoming gradient), ``''``.
"""
from __future__ import annotations

import ast

import math
import os
from collections import OrderedDict, defaultdict

import numpy as np
import torch

from .. import ops
from ..engine import MAEEngine
from ..model_config import MODEL_TYPES, config_for
from ..predictor_head import PredictorHead
from .lr_decay import param_groups_lrd
from .mim_vit import _DataParallelShim, _PatchEmbedInfo, _compute_dtype
from .misc import str2bool
from .pos_embed import interpolate_pos_embed

_ENCODER_KEYS = ("cls_token", "pos_embed", "patch_mask_values", "patch_embed.", "blocks.", "norm.", "ra_dec_embed.")


class _Predict(torch.autograd.Function):
    """HIP encoder + head forward -> predictions [B, num_classes]; backward hands d predictions to the head's backward schedule and
    what that leaves (d features / d tokens) to the engine's (parameter gradients land in the flat gradient buffers) -- or only to
    the final norm's when the encoder is frozen (linear probe)."""

    @staticmethod
    def forward(ctx, hook, model, x, ra_dec):
        ctx.model = model
        return model._predict_train(x, ra_dec)

    @staticmethod
    def backward(ctx, dpred):
        ctx.model._predict_backward(dpred.contiguous())
        return torch.zeros(1, device=dpred.device), None, None, None


class VisionTransformer:
    """utils/vit.py:258-393."""

    def __init__(self, cfg, device, compute_dtype, num_classes=0, global_pool='token', label_means=(0.0,), label_stds=(1.0,),
                 drop_rate=0.0, seed=None):
        if global_pool not in ('', 'avg', 'token', 'map'):
            raise ValueError(f"global_pool = {global_pool!r}")
        self.cfg = cfg
        if compute_dtype == torch.float16:
            # the predictor's loss lives in the caller's torch code: its backward enters the engine unscaled, and fp16 data gradients
            # need the static loss scale the MIM engines apply in their own loss kernels
            raise NotImplementedError("the downstream predictor runs in bf16 or f32 (compute_dtype = f16 is a pretraining mode)")
        self.engine = MAEEngine(cfg, device=device, compute_dtype=compute_dtype, seed=seed)
        self.engine.fold_decoder_wgrads = False      # backward_decoder never runs here: its folded weight gradients would read garbage
        dev = self.engine.device
        self.patch_embed = _PatchEmbedInfo(cfg)
        self.num_extra_tokens, self.attn_pool, self.simmim = cfg.num_extra_tokens, None, False
        self.in_chans, self.pixel_mean, self.pixel_std, self.ra_dec = cfg.in_chans, cfg.pixel_mean, cfg.pixel_std, cfg.ra_dec
        self.global_pool, self.num_classes, self.drop_rate = global_pool, int(num_classes), float(drop_rate)
        self.label_means = torch.tensor(label_means, dtype=torch.float32)
        self.label_stds = torch.tensor(label_stds, dtype=torch.float32)
        self.tile_size = cfg.img_size // cfg.patch_size
        self.num_blocks = cfg.depth
        D = cfg.embed_dim
        gen = torch.Generator().manual_seed(0 if seed is None else seed)
        # head-side tensors under timm's names: views of the head's flat fp32 buffer (predictor_head.PredictorHead).  'avg' pooling
        # normalises the POOLED features (fc_norm) and has no final norm; 'map' = timm AttentionPoolLatent(embed_dim, num_heads=2,
        # mlp_ratio, norm_layer) (utils/vit.py:303-309)
        self.pool_heads = 2
        self._head_mod = PredictorHead(D, cfg.mlp_ratio, cfg.ln_eps, global_pool, self.num_classes, dev, compute_dtype, gen,
                                       splitk_ws=self.engine._splitk_ws)
        self.head = self._head_mod.tensors
        self.training = False
        self.frozen_encoder = False                # linear probe: only norm / fc_norm / head receive gradients
        self.trainable = None                      # names with requires_grad (None = everything but pos_embed)
        self._hook = torch.zeros(1, device=dev, requires_grad=True)
        self._ramp = None
        self.pos_embed = self.engine.store.frozen["pos_embed"]

    # ---- nn.Module-like surface ---------------------------------------------------------------------------------------------
    def eval(self):
        return self.train(False)

    def train(self, mode=True):
        self.training = bool(mode)
        # ARCHITECTURE.dropout (utils/vit.py:40, 55-123 -> timm's `drop_rate`): dropout on the pooled features in front of the classifier
        # (timm >= 0.9 `head_drop`; the positional / projection / attention dropouts have rates of their own there, which the
        # reference leaves at 0), active in training mode only
        self._head_mod.drop_rate, self._head_mod.training = self.drop_rate, self.training
        return self

    def no_weight_decay(self):
        return {'pos_embed', 'cls_token', 'dist_token'}

    def _encoder_names(self):
        names = [n for n in self.engine.store.order if n.startswith(_ENCODER_KEYS)]
        if self.global_pool == 'avg':
            names = [n for n in names if not n.startswith("norm.")]      # timm: nn.Identity() when fc_norm is used
        return names

    def state_dict(self):
        """The reference module's tensors under its names (the engine also owns an MAE decoder nobody uses here: not listed)."""
        own = self.engine.state_dict()
        out = OrderedDict((k, v) for k, v in own.items() if k.startswith(_ENCODER_KEYS) and (self.global_pool != 'avg' or not k.startswith("norm.")))
        out.update(self.head)
        return out

    def trainable_tensors(self):
        """[(name, ndim)] of the tensors an optimiser may touch, in state-dict order (pos_embed is a fixed table)."""
        sd = self.state_dict()
        return [(k, v.dim()) for k, v in sd.items() if k != 'pos_embed' and (self.trainable is None or k in self.trainable)]

    def load_state_dict(self, sd, strict=True):
        own = self.state_dict()
        missing = [k for k in own if k not in sd]
        unexpected = [k for k in sd if k not in own]
        if strict and (missing or unexpected):
            raise RuntimeError(f"load_state_dict: missing {missing[:5]} unexpected {unexpected[:5]}")
        eng_sd = self.engine.state_dict()
        self.engine.load_state_dict({**eng_sd, **{k: v for k, v in sd.items() if k in eng_sd and k in own}})
        for k in self.head:
            if k in sd:
                self.head[k].copy_(torch.as_tensor(sd[k]).to(torch.float32).reshape(self.head[k].shape))
        self.sync_head()
        return type("Keys", (), {"missing_keys": missing, "unexpected_keys": unexpected})()

    def sync_head(self):
        """Refresh the compute-dtype shadow the head's GEMMs read (after anything wrote the fp32 head tensors directly)."""
        self._head_mod.store.refresh_lp()

    def load_encoder_state(self, sd):
        own = self.engine.state_dict()
        picked = {k: v for k, v in sd.items() if k in own and k.startswith(_ENCODER_KEYS)}
        missing = [k for k in self._encoder_names() if k not in picked]
        if missing:
            raise RuntimeError(f"checkpoint lacks encoder tensors, e.g. {missing[:4]}")
        self.engine.load_state_dict({**{k: v for k, v in own.items()}, **picked})

    # ---- labels (utils/vit.py:315-342) --------------------------------------------------------------------------------------
    def norm_inputs(self, x):
        return (x - self.pixel_mean) / self.pixel_std

    def normalize_labels(self, labels):
        return (labels - self.label_means.to(labels.device)) / self.label_stds.to(labels.device)

    def denormalize_labels(self, labels):
        return labels * self.label_stds.to(labels.device) + self.label_means.to(labels.device)

    # ---- encoder ------------------------------------------------------------------------------------------------------------
    def _inputs(self, x, ra_dec):
        x = x.to(self.engine.device, torch.float32).contiguous()
        B, L = x.shape[0], self.cfg.num_patches
        if self._ramp is None or self._ramp.shape[0] != B:
            self._ramp = torch.arange(L, device=x.device, dtype=torch.float32).repeat(B, 1).contiguous() / L
        return x, self.engine._check_ra_dec(x, ra_dec)

    def forward_features(self, x, ra_dec=None, mask=None, reshape_out=False):
        """utils/vit.py:344-388 -> (tokens [B, extra+L, D], None, None); extra = cls (+ the RA/Dec token)."""
        if mask is not None:
            raise NotImplementedError("utils.vit forward_features with a pixel mask is not built (the predictors pass none)")
        x, ra_dec = self._inputs(x, ra_dec)
        B, L = x.shape[0], self.cfg.num_patches
        lat, _, _ = self.engine.forward_features(x, 0.0, self._ramp, ra_dec=ra_dec)
        if self.global_pool == 'avg':          # no final norm in this configuration: the stream itself
            w = self.engine._ws[(B, L, False)]
            lat = w["xs"][self.cfg.depth % 2].view(B, -1, self.cfg.embed_dim)
        lat = lat.clone()
        if reshape_out:
            lat = lat[:, self.num_extra_tokens:]
            H = W = int(L ** 0.5)
            lat = lat.permute(0, 2, 1).reshape(B, -1, H, W)
        return lat, None, None

    def _predict_train(self, x, ra_dec):
        """Encoder forward with activations kept, then the head: -> predictions [B, num_classes] (pooled features when the model
        has no classifier)."""
        eng, cfg, hd = self.engine, self.cfg, self._head_mod
        B, L = x.shape[0], cfg.num_patches
        w = eng._workspace(B, L, True)
        eng._encoder_fwd(x, self._ramp, L, w, True, ra_dec)
        eng._last = (x, B, L)
        Ne, D = cfg.num_extra_tokens + L, cfg.embed_dim
        if self.global_pool == 'map':
            out = hd.forward(B, Ne, tokens_lp=w["lat_lp"])          # the final norm's output in the compute dtype: every token
        elif self.global_pool == 'avg':
            out = hd.forward(B, Ne, feat=w["xs"][cfg.depth].view(B, Ne, D)[:, 1:].mean(dim=1))
        elif self.global_pool == 'token':
            out = hd.forward(B, Ne, feat=w["latent32"].view(B, Ne, D)[:, 0])
        else:
            raise NotImplementedError("training needs a pooled head (global_pool = token | avg | map)")
        return out.clone()

    def _predict_backward(self, dpred):
        eng, cfg, hd = self.engine, self.cfg, self._head_mod
        x, B, L = eng._last
        w = eng._ws[(B, L, True)]
        Ne, D = cfg.num_extra_tokens + L, cfg.embed_dim
        Me = B * Ne
        st = eng.store
        dlat = w["dln"][:Me * D].view(Me, D)                      # the final norm's incoming gradient (compute dtype)
        if self.num_classes <= 0:
            raise NotImplementedError("backward through a model without a classifier")
        dfeat = hd.backward(dpred, B, Ne, tokens_lp=w["lat_lp"], dtokens_lp=dlat)
        if self.global_pool == 'avg':
            if self.frozen_encoder:
                return                                 # nothing inside the engine trains (fc_norm and the head live outside it)
            g = w["g"][:Me * D].view(B, Ne, D)
            g.zero_()
            g[:, 1:] = (dfeat / (Ne - 1)).unsqueeze(1)
            w["g_lp"][:Me * D].view(B, Ne, D).copy_(g)
            eng._ln_first, eng._ln_count = w["ln_index"][f"blocks.{cfg.depth - 1}.norm2"], 0
            eng.backward_encoder(hi=cfg.depth, lo=0)   # (hi given: the final norm is not part of this configuration)
            eng.backward_embed()
            return
        if self.global_pool != 'map':                  # class token: d loss / d (row 0 of every sample), zero elsewhere
            dlat.zero_()
            dlat.view(B, Ne, D)[:, 0] = dfeat
        if self.frozen_encoder:
            # linear probe: the final norm's own gradients, nothing below it (utils/vit.py:145-160)
            tmp = w["g"][:Me * D].view(Me, D)
            ops.layernorm_bwd(dlat, w["xs"][cfg.depth], st.param("norm.weight"), w["lat_mean"], w["lat_rstd"], None, tmp, None,
                              w["ln_parts"]["norm"], st.grad("norm.weight"), st.grad("norm.bias"), Me, D, eng.code)
            return
        eng._ln_first, eng._ln_count = w["ln_index"]["norm"], 0
        eng.backward_encoder()
        eng.backward_embed()

    def forward_head(self, x, pre_logits=False):
        """timm VisionTransformer.forward_head on a token tensor [B, N, D] (inference): pooling, fc_norm, classifier -- the head's
        forward schedule on the library's kernels."""
        hd = self._head_mod
        B, N, D = x.shape
        x = x.to(self.engine.device, torch.float32).contiguous()
        if self.global_pool == 'map':
            w = hd._workspace(B, N)
            ops.cast(x.view(-1), w["x_lp"], B * N * D)
            return hd.forward(B, N, tokens_lp=w["x_lp"], pre_logits=pre_logits).clone()
        if self.global_pool == 'avg':
            return hd.forward(B, N, feat=x[:, 1:].mean(dim=1), pre_logits=pre_logits).clone()     # timm: ONE prefix token (the RA/Dec token is averaged in)
        if self.global_pool == 'token':
            return hd.forward(B, N, feat=x[:, 0], pre_logits=pre_logits).clone()
        return x

    def forward(self, x, mask=None, ra_dec=None):
        """utils/vit.py:390-393 -> predictions [B, num_classes].  In training mode the result is the output of the autograd node
        whose backward runs the head's and the engine's backward schedules."""
        if self.training and torch.is_grad_enabled():
            x, ra_dec = self._inputs(x, ra_dec)
            return _Predict.apply(self._hook, self, x, ra_dec)
        with torch.no_grad():
            tokens, _, _ = self.forward_features(x, ra_dec=ra_dec)
            return self.forward_head(tokens)

    __call__ = forward


class PredictorOptimizer:
    """torch.optim.AdamW over named parameter groups of a downstream ViT: every tensor -- the engine's and the head's -- is
    stepped by the AdamW kernel on its slice of the flat buffers it lives in (fp32 master + the compute-dtype shadow the GEMMs
    read).  groups: [{'params': [names], 'lr', 'weight_decay'(, 'betas')}]; eps = torch's default, as the reference.
    state_dict / load_state_dict use torch.optim.AdamW's layout (integer parameter ids in group order, per-parameter
    'step' / 'exp_avg' / 'exp_avg_sq'), so predictor checkpoints interchange with the reference's."""

    def __init__(self, model, groups, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        self.model = model
        self.defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        self.param_groups = []
        for g in groups:
            g = dict(g)
            g.setdefault("lr", lr)
            g.setdefault("weight_decay", weight_decay)
            g.setdefault("betas", tuple(betas))
            g["initial_lr"] = g["lr"]
            self.param_groups.append(g)
        self.step_count = 0

    def zero_grad(self, set_to_none=True):
        """Nothing to do: every backward overwrites the gradient buffers (engine and head) it computes."""

    def _buffers(self, name):
        """(p, g, m, v, p_lp) slices of the flat buffers `name` lives in, padded run included."""
        hs = self.model._head_mod.store
        if name in hs.offsets:
            o, n = hs.offsets[name], hs.sizes[name]
            return tuple(b[o:o + n] for b in (hs.p, hs.g, hs.m, hs.v, hs.p_lp)), hs
        st = self.model.engine.store
        o = st.offsets[name]
        n = (int(np.prod(st.shapes[name])) + 7) // 8 * 8
        return tuple(b[o:o + n] for b in (st.p, st.g, st.m, st.v, st.p_lp)), st

    def step(self):
        self.step_count += 1
        t = self.step_count
        eps = self.defaults["eps"]
        for g in self.param_groups:
            lr, wd = g["lr"], g["weight_decay"]
            b1, b2 = g["betas"]
            bc1, bc2 = 1.0 - b1 ** t, 1.0 - b2 ** t
            for name in g["params"]:
                (p, gr, m, v, p_lp), _ = self._buffers(name)
                n = p.numel()
                ops.adamw(p, gr, m, v, p_lp, n, n if wd != 0.0 else 0, None, b1, b2, eps, wd, lr=lr, bc1=bc1, bc2=bc2)

    def _names(self):
        return [n for g in self.param_groups for n in g["params"]]

    def state_dict(self):
        """torch.optim.AdamW.state_dict(): {'state': {id: {'step', 'exp_avg', 'exp_avg_sq'}}, 'param_groups': [... 'params': [ids]]};
        'param_names' (ids -> names) rides along for readers of this repo and is ignored by torch."""
        names = self._names()
        state = {}
        if self.step_count:
            for i, n in enumerate(names):
                _, store = self._buffers(n)
                state[i] = {"step": torch.tensor(float(self.step_count)), "exp_avg": store._view(store.m, n).detach().clone().cpu(),
                            "exp_avg_sq": store._view(store.v, n).detach().clone().cpu()}
        groups, k = [], 0
        for g in self.param_groups:
            d = {key: v for key, v in g.items() if key != "params"}
            d.update(eps=self.defaults["eps"], amsgrad=False, maximize=False, foreach=None, capturable=False, differentiable=False,
                     fused=None, decoupled_weight_decay=True)
            d["params"] = list(range(k, k + len(g["params"])))
            k += len(g["params"])
            groups.append(d)
        return {"state": state, "param_groups": groups, "param_names": names}

    def load_state_dict(self, sd):
        """Accepts torch's layout (the reference's checkpoints: ids in group order) and this repo's earlier name-keyed one."""
        names = self._names()
        steps = []
        for key, s in sd["state"].items():
            n = names[int(key)] if not isinstance(key, str) else key
            _, store = self._buffers(n)
            store._view(store.m, n).copy_(torch.as_tensor(s["exp_avg"]).reshape(store.shapes[n]))
            store._view(store.v, n).copy_(torch.as_tensor(s["exp_avg_sq"]).reshape(store.shapes[n]))
            if "step" in s:
                steps.append(int(float(s["step"])))
        self.step_count = int(sd["step"]) if "step" in sd else (max(steps) if steps else 0)
        for g, saved in zip(self.param_groups, sd["param_groups"]):
            for k in ("lr", "initial_lr", "weight_decay", "betas"):
                if k in saved:
                    g[k] = tuple(saved[k]) if k == "betas" else saved[k]


def apply_onecycle_side_effects(optimizer, max_lr, div_factor=25.0, max_momentum=0.95, final_div_factor=1e4):
    """utils/vit.py:174-182 constructs a OneCycleLR and throws it away (:183-186 overwrites it with LinearLR) -- but the constructor
    has already rewritten the optimiser: every group's ``initial_lr = lr = max_lr / div_factor`` and, with cycle_momentum,
    ``betas = (max_momentum, beta2)``; LinearLR then keeps that ``initial_lr`` (torch's LRScheduler uses setdefault).  So the
    reference trains from init_lr / 25 with beta1 = 0.95.  max_lr: one value or one per group (the fine-tuning branch hands over
    the groups' own scaled rates)."""
    per_group = list(max_lr) if isinstance(max_lr, (list, tuple)) else [max_lr] * len(optimizer.param_groups)
    assert len(per_group) == len(optimizer.param_groups)
    for g, mx in zip(optimizer.param_groups, per_group):
        g["initial_lr"] = g["lr"] = mx / div_factor
        g["max_lr"], g["min_lr"] = mx, mx / div_factor / final_div_factor     # (bookkeeping torch leaves in the group; unused afterwards)
        g["betas"] = (max_momentum, g["betas"][1])
        g["max_momentum"], g["base_momentum"] = max_momentum, 0.85


class LinearLR:
    """torch.optim.lr_scheduler.LinearLR(optimizer, start_factor, end_factor, total_iters) in closed form."""

    def __init__(self, optimizer, start_factor=1.0, end_factor=1.0, total_iters=5):
        self.optimizer, self.start_factor, self.end_factor, self.total_iters = optimizer, start_factor, end_factor, int(total_iters)
        self.base_lrs = [g["initial_lr"] for g in optimizer.param_groups]
        self.last_epoch = 0
        self._apply()

    def _factor(self):
        t = min(self.last_epoch, self.total_iters)
        return self.start_factor + (self.end_factor - self.start_factor) * t / self.total_iters

    def _apply(self):
        for g, base in zip(self.optimizer.param_groups, self.base_lrs):
            g["lr"] = base * self._factor()

    def step(self):
        self.last_epoch += 1
        self._apply()

    def get_last_lr(self):
        return [g["lr"] for g in self.optimizer.param_groups]

    def state_dict(self):
        """torch's keys (base_lrs, last_epoch, _step_count, _last_lr, ...)."""
        return {"start_factor": self.start_factor, "end_factor": self.end_factor, "total_iters": self.total_iters,
                "base_lrs": list(self.base_lrs), "last_epoch": self.last_epoch, "_step_count": self.last_epoch + 1,
                "_is_initial": False, "_get_lr_called_within_step": False, "_last_lr": self.get_last_lr()}

    def load_state_dict(self, sd):
        self.last_epoch = int(sd["last_epoch"])
        if "base_lrs" in sd and len(sd["base_lrs"]) == len(self.base_lrs):
            self.base_lrs = [float(v) for v in sd["base_lrs"]]
        self._apply()


def build_optimizer(model, train_method, init_lr, weight_decay, layer_decay):
    """utils/vit.py:134-172 on a VisionTransformer of this module (``model`` = the object behind ``.module``)."""
    if train_method in ('finetune', 'ft'):
        print('\nUsing the fine-tuning training method...')
        # utils/vit.py:141-143 passes (model, weight_decay, ...) POSITIONALLY to param_groups_lrd(model, init_lr, weight_decay=0.05, ...):
        # the configured weight decay becomes the base lr of the groups and their weight decay stays 0.05.  Mirrored as written.
        groups, _ = param_groups_lrd(model, weight_decay, no_weight_decay_list=model.no_weight_decay(), layer_decay=layer_decay)
        return PredictorOptimizer(model, groups)                 # (the groups' own rates are what utils/vit.py:174 hands OneCycleLR)
    if train_method in ('linearprobe', 'lp'):
        print('\nUsing the linear probing training method...')
        comps = ["norm.", "fc_norm.", "head."]
        if model.global_pool == 'map':
            comps.append("attn_pool.")
        names = [k for k, _ in model.trainable_tensors()]
        groups = [{"params": [k for k in names if k.startswith(c)]} for c in comps]
        model.trainable = {k for g in groups for k in g["params"]}      # everything else: requires_grad = False
        model.frozen_encoder = True
        # (one group per component, as the reference builds them -- fc_norm's is empty unless the pooling is 'avg')
        return PredictorOptimizer(model, groups, lr=init_lr, weight_decay=weight_decay)
    print('\nUsing the fully supervised training method...')
    # timm's param_groups_weight_decay: 1-D tensors and biases are not decayed, everything else is (cls_token and patch_mask_values too)
    no_decay = [k for k, nd in model.trainable_tensors() if nd <= 1 or k.endswith(".bias")]
    decay = [k for k, nd in model.trainable_tensors() if not (nd <= 1 or k.endswith(".bias"))]
    return PredictorOptimizer(model, [{"params": no_decay, "weight_decay": 0.0}, {"params": decay, "weight_decay": weight_decay}], lr=init_lr)


def build_model(config, mae_config, model_filename, mae_filename, device, build_optimizer=False):
    """utils/vit.py:21-195 -> (model, losses, cur_iter[, optimizer, lr_scheduler])."""
    arch = mae_config['ARCHITECTURE']
    model_type = arch['model_type']
    base = {"simmim": "base", "mimlarge": "large", "mimhuge": "huge"}.get(model_type, model_type)
    if base not in MODEL_TYPES:
        raise KeyError(model_type)
    cfg = config_for(base, img_size=int(config['ARCHITECTURE']['img_size']), patch_size=int(arch['patch_size']),
                     in_chans=int(arch['num_channels']), embed_dim=int(arch['embed_dim']),
                     pixel_mean=float(arch['pixel_mean']), pixel_std=float(arch['pixel_std']),
                     ra_dec=str2bool(arch.get('ra_dec', 'False')))
    data = config['DATA'] if 'DATA' in config else {}
    if 'num_classes' in data:
        num_labels = int(data['num_classes'])
    elif 'label_keys' in data:
        num_labels = len(ast.literal_eval(data['label_keys']))
        if 'TRAINING' in config and str2bool(config['TRAINING'].get('use_label_errs', 'False')):
            num_labels = num_labels // 2
    else:
        num_labels = 0                           # similarity_search.py routes through forward_features only
    dt = _compute_dtype(mae_config, default='bf16')
    if dt == torch.float16:      # (a pretraining ini that names f16: the predictor's loss is the caller's torch code -- no loss scale -- so bf16)
        dt = torch.bfloat16
    model = VisionTransformer(cfg, device, dt, num_classes=num_labels,
                              global_pool=config['ARCHITECTURE'].get('global_pool', 'token'),
                              # utils/vit.py:38-39 as written: the LENGTHS of the configured lists -- 1 for every shipped ini, so
                              # labels are normalised as (y - 1) / 1 whatever the ini says (label_stds = [0] in all cls_*.ini)
                              label_means=len(ast.literal_eval(data['label_means'])) if 'label_means' in data else 1,
                              label_stds=len(ast.literal_eval(data['label_stds'])) if 'label_stds' in data else 1,
                              drop_rate=float(ast.literal_eval(config['ARCHITECTURE'].get('dropout', '0.0'))))
    model = _DataParallelShim(model)
    if not build_optimizer:
        return load_model(model, model_filename, mae_filename)
    tr = config['TRAINING']
    total = int(float(tr['total_batch_iters']))
    optimizer = globals()['build_optimizer'](model.module, tr['train_method'], float(tr['init_lr']), float(tr['weight_decay']),
                                             float(tr['layer_decay']))
    # utils/vit.py:174-182: the discarded OneCycleLR's constructor (max_lr = the groups' scaled rates when fine-tuning -- the
    # `init_lr` param_groups_lrd returned -- else the configured init_lr)
    finetune = tr['train_method'] in ('finetune', 'ft')
    apply_onecycle_side_effects(optimizer, [g["lr"] for g in optimizer.param_groups] if finetune else float(tr['init_lr']),
                                final_div_factor=float(tr['final_lr_factor']))
    lr_scheduler = LinearLR(optimizer, start_factor=1.0, end_factor=1 / float(tr['final_lr_factor']), total_iters=total)
    model, losses, cur_iter = load_model(model, model_filename, mae_filename, optimizer, lr_scheduler)
    return model, losses, cur_iter, optimizer, lr_scheduler


def load_model(model, model_filename, mae_filename='None', optimizer=None, lr_scheduler=None):
    """utils/vit.py:198-256: resume the predictor's own checkpoint, or start from a pre-trained MAE checkpoint (head tensors of
    another shape dropped, positional table interpolated to this image size, head weight re-initialised small), or fresh."""
    m = model.module
    if model_filename and model_filename != 'None' and os.path.exists(model_filename):
        print('\nLoading saved model weights...')
        ck = torch.load(model_filename, map_location="cpu", weights_only=False)
        losses, cur_iter = defaultdict(list, dict(ck['losses'])), ck['batch_iters'] + 1
        if optimizer is not None:
            optimizer.load_state_dict(ck['optimizer'])
        if lr_scheduler is not None:
            lr_scheduler.load_state_dict(ck['lr_scheduler'])
        sd = dict(ck['model'])
        interpolate_pos_embed(m, sd)
        own = m.state_dict()
        if all(k in sd for k in own):
            m.load_state_dict({k: sd[k] for k in own})
        else:
            m.load_encoder_state(sd)               # (an encoder-only checkpoint written by the similarity-search route)
    elif mae_filename and mae_filename != 'None' and os.path.exists(mae_filename):
        print('\nLoading pre-trained MAE model weights...')
        sd = dict(torch.load(mae_filename, map_location="cpu", weights_only=False)['model'])
        own = m.state_dict()
        for k in ('head.weight', 'head.bias'):
            if k in sd and k in own and tuple(sd[k].shape) != tuple(own[k].shape):
                print(f"Removing key {k} from pretrained checkpoint")
                del sd[k]
        interpolate_pos_embed(m, sd)
        m.load_state_dict({k: v for k, v in sd.items() if k in own}, strict=False)
        if "head.weight" in m.head:
            torch.nn.init.trunc_normal_(m.head["head.weight"], std=2e-5)
            m.sync_head()
        losses, cur_iter = defaultdict(list), 1
    else:
        print('\nStarting fresh model to train...')
        losses, cur_iter = defaultdict(list), 1
    return model, losses, cur_iter
