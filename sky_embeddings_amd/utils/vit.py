"""utils/vit.py mirror: the downstream ViT -- a pre-trained MAE / SimMIM encoder with a prediction head -- built, loaded and
TRAINED through the reference's API (``build_model`` with ``build_optimizer``, ``load_model``, ``model.module.*``).

The encoder (patch embedding, Blocks, final norm) runs forward AND backward in the HIP engine (``MAEEngine``: the same kernels
as pre-training; tokens in raster order, nothing masked); the head -- pooling, ``fc_norm``, one Linear -- is a few kilobytes of
torch on the same device, joined to the engine by an autograd node whose backward runs the engine's backward schedule.
Training methods of utils/vit.py:134-172: ``ft`` (fine-tuning with layer-wise lr decay), ``lp`` (linear probe: final norm,
``fc_norm`` and head only; the encoder backward is not run at all), anything else "fully supervised" (timm's weight-decay split,
one lr); schedule = the LinearLR the reference ends up with (its OneCycleLR is overwritten, utils/vit.py:174-186).
Pooling: ``token`` (class token), ``avg`` (mean of the patch tokens + ``fc_norm``; timm then has no final norm), ``map`` (timm's
``AttentionPoolLatent`` with two heads, as every shipped predictor config asks for: one learned query over the encoder's tokens,
projection, LayerNorm + MLP residual -- evaluated in torch like the rest of the head, ~1.5 % of the encoder's FLOPs; its gradient
with respect to ALL tokens enters the engine's backward), ``''``.
"""
from __future__ import annotations

import math
import os
from collections import OrderedDict, defaultdict

import numpy as np
import torch

from .. import ops
from ..engine import MAEEngine
from ..model_config import MODEL_TYPES, config_for
from .lr_decay import param_groups_lrd
from .mim_vit import _DataParallelShim, _PatchEmbedInfo, _compute_dtype
from .misc import str2bool
from .pos_embed import interpolate_pos_embed

_ENCODER_KEYS = ("cls_token", "pos_embed", "patch_mask_values", "patch_embed.", "blocks.", "norm.", "ra_dec_embed.")


class _EncoderFeatures(torch.autograd.Function):
    """HIP encoder forward -> the pooled-from tensor (class-token rows after the final norm, or the patch-token mean of the
    un-normalised stream); backward hands d features to the engine's backward schedule (parameter gradients land in the flat
    gradient buffer) -- or only to the final norm's when the encoder is frozen (linear probe)."""

    @staticmethod
    def forward(ctx, hook, model, x, ra_dec):
        ctx.model = model
        return model._encode_train(x, ra_dec)

    @staticmethod
    def backward(ctx, dfeat):
        ctx.model._encode_backward(dfeat.contiguous())
        return torch.zeros(1, device=dfeat.device), None, None, None


class VisionTransformer:
    """utils/vit.py:258-393."""

    def __init__(self, cfg, device, compute_dtype, num_classes=0, global_pool='token', label_means=(0.0,), label_stds=(1.0,),
                 drop_rate=0.0, seed=None):
        if global_pool not in ('', 'avg', 'token', 'map'):
            raise ValueError(f"global_pool = {global_pool!r}")
        self.cfg = cfg
        self.engine = MAEEngine(cfg, device=device, compute_dtype=compute_dtype, seed=seed)
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
        # head-side tensors (torch, fp32): timm's names.  'avg' pooling normalises the POOLED features (fc_norm) and has no final norm
        self.head = OrderedDict()
        if global_pool == 'avg':
            self.head["fc_norm.weight"], self.head["fc_norm.bias"] = torch.ones(D, device=dev), torch.zeros(D, device=dev)
        if global_pool == 'map':
            # timm AttentionPoolLatent(embed_dim, num_heads=2, mlp_ratio, norm_layer) (utils/vit.py:303-309): trunc-normal latent, Linear init
            hid = int(D * cfg.mlp_ratio)

            def lin(o, i):
                return (torch.randn(o, i, generator=gen) * 0.02).clamp_(-0.04, 0.04).to(dev), torch.zeros(o, device=dev)
            self.pool_heads = 2
            self.head["attn_pool.latent"] = (torch.randn(1, 1, D, generator=gen) * D ** -0.5).clamp_(-2 * D ** -0.5, 2 * D ** -0.5).to(dev)
            for name, (o, i) in (("q", (D, D)), ("kv", (2 * D, D)), ("proj", (D, D))):
                self.head[f"attn_pool.{name}.weight"], self.head[f"attn_pool.{name}.bias"] = lin(o, i)
            self.head["attn_pool.norm.weight"], self.head["attn_pool.norm.bias"] = torch.ones(D, device=dev), torch.zeros(D, device=dev)
            self.head["attn_pool.mlp.fc1.weight"], self.head["attn_pool.mlp.fc1.bias"] = lin(hid, D)
            self.head["attn_pool.mlp.fc2.weight"], self.head["attn_pool.mlp.fc2.bias"] = lin(D, hid)
        if self.num_classes > 0:
            self.head["head.weight"] = (torch.randn(self.num_classes, D, generator=gen) * 0.02).clamp_(-0.04, 0.04).to(dev)
            self.head["head.bias"] = torch.zeros(self.num_classes, device=dev)
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
        if mode and self.drop_rate != 0.0:
            raise NotImplementedError("dropout in the downstream predictor (ARCHITECTURE.dropout != 0) is not built")
        self.training = bool(mode)
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
        return type("Keys", (), {"missing_keys": missing, "unexpected_keys": unexpected})()

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

    def _pool(self, tokens):
        if self.global_pool == 'avg':
            return tokens[:, 1:].mean(dim=1)      # timm: x[:, num_prefix_tokens:] with ONE prefix token (the RA/Dec token is averaged in)
        if self.global_pool == 'token':
            return tokens[:, 0]
        if self.global_pool == 'map':
            return self._attn_pool(tokens)
        return tokens

    def _attn_pool(self, x):
        """timm.layers.AttentionPoolLatent.forward (latent_len 1, pool 'token', no q / k norm, no positional table)."""
        F, P = torch.nn.functional, self.head
        B, N, C = x.shape
        H = self.pool_heads
        hd = C // H
        q = F.linear(P["attn_pool.latent"].expand(B, -1, -1), P["attn_pool.q.weight"], P["attn_pool.q.bias"]).reshape(B, 1, H, hd).transpose(1, 2)
        kv = F.linear(x, P["attn_pool.kv.weight"], P["attn_pool.kv.bias"]).reshape(B, N, 2, H, hd).permute(2, 0, 3, 1, 4)
        k, v = kv.unbind(0)
        attn = ((q * hd ** -0.5) @ k.transpose(-2, -1)).softmax(dim=-1)
        y = F.linear((attn @ v).transpose(1, 2).reshape(B, 1, C), P["attn_pool.proj.weight"], P["attn_pool.proj.bias"])
        z = F.layer_norm(y, (C,), P["attn_pool.norm.weight"], P["attn_pool.norm.bias"], 1e-6)
        z = F.linear(F.gelu(F.linear(z, P["attn_pool.mlp.fc1.weight"], P["attn_pool.mlp.fc1.bias"])), P["attn_pool.mlp.fc2.weight"], P["attn_pool.mlp.fc2.bias"])
        return (y + z)[:, 0]

    def _encode_train(self, x, ra_dec):
        """Encoder forward with activations kept; -> what the head starts from: [B, D] (token / avg: already pooled) or every
        token [B, N, D] (map: the attention pool is part of the torch head)."""
        eng, cfg = self.engine, self.cfg
        B, L = x.shape[0], cfg.num_patches
        w = eng._workspace(B, L, True)
        eng._encoder_fwd(x, self._ramp, L, w, True, ra_dec)
        eng._last = (x, B, L)
        Ne, D = cfg.num_extra_tokens + L, cfg.embed_dim
        if self.global_pool == 'avg':
            return w["xs"][cfg.depth].view(B, Ne, D)[:, 1:].mean(dim=1)
        if self.global_pool == 'map':
            return w["latent32"].view(B, Ne, D).clone()
        return w["latent32"].view(B, Ne, D)[:, 0].clone()

    def _encode_backward(self, dfeat):
        eng, cfg = self.engine, self.cfg
        x, B, L = eng._last
        w = eng._ws[(B, L, True)]
        Ne, D = cfg.num_extra_tokens + L, cfg.embed_dim
        Me = B * Ne
        st = eng.store
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
        dlat = w["dln"][:Me * D].view(B, Ne, D)
        if self.global_pool == 'map':
            dlat.copy_(dfeat)                          # d loss / d every token
        else:
            dlat.zero_()
            dlat[:, 0] = dfeat
        if self.frozen_encoder:
            # linear probe: the final norm's own gradients, nothing below it (utils/vit.py:145-160)
            tmp = w["g"][:Me * D].view(Me, D)
            ops.layernorm_bwd(dlat.view(Me, D), w["xs"][cfg.depth], st.param("norm.weight"), w["lat_mean"], w["lat_rstd"], None, tmp, None,
                              w["ln_parts"]["norm"], st.grad("norm.weight"), st.grad("norm.bias"), Me, D, eng.code)
            return
        eng._ln_first, eng._ln_count = w["ln_index"]["norm"], 0
        eng.backward_encoder()
        eng.backward_embed()

    def forward_head(self, x, pre_logits=False):
        """timm VisionTransformer.forward_head on a token tensor [B, N, D] (inference)."""
        return self._head(self._pool(x), pre_logits)

    def _head(self, f, pre_logits=False):
        if f.dim() == 3:                                # (training path with global_pool = map: f holds every token)
            f = self._attn_pool(f)
        if self.global_pool == 'avg':
            f = torch.nn.functional.layer_norm(f, (f.shape[-1],), self.head["fc_norm.weight"], self.head["fc_norm.bias"], 1e-6)
        if pre_logits or self.num_classes <= 0:
            return f
        return torch.nn.functional.linear(f, self.head["head.weight"], self.head["head.bias"])

    def forward(self, x, mask=None, ra_dec=None):
        """utils/vit.py:390-393 -> predictions [B, num_classes].  In training mode the result carries the autograd graph of the
        head and the node that runs the engine's backward."""
        if self.training and torch.is_grad_enabled():
            x, ra_dec = self._inputs(x, ra_dec)
            for k, v in self.head.items():
                v.requires_grad_(self.trainable is None or k in self.trainable)
            f = _EncoderFeatures.apply(self._hook, self, x, ra_dec)
            return self._head(f)
        with torch.no_grad():
            tokens, _, _ = self.forward_features(x, ra_dec=ra_dec)
            return self.forward_head(tokens)

    __call__ = forward


class PredictorOptimizer:
    """torch.optim.AdamW over named parameter groups of a downstream ViT: engine tensors are stepped by the AdamW kernel on
    their slices of the flat buffers (fp32 master + the compute-dtype shadow the GEMMs read), head-side tensors by the same
    formula in torch.  groups: [{'params': [names], 'lr', 'weight_decay'}]; betas / eps = torch's defaults, as the reference."""

    def __init__(self, model, groups, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        self.model = model
        self.defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        self.param_groups = []
        for g in groups:
            g = dict(g)
            g.setdefault("lr", lr)
            g.setdefault("weight_decay", weight_decay)
            g["initial_lr"] = g["lr"]
            self.param_groups.append(g)
        self.step_count = 0
        self.head_state = {}

    def zero_grad(self, set_to_none=True):
        for v in self.model.head.values():
            v.grad = None

    def step(self):
        self.step_count += 1
        t = self.step_count
        b1, b2 = self.defaults["betas"]
        eps = self.defaults["eps"]
        bc1, bc2 = 1.0 - b1 ** t, 1.0 - b2 ** t
        st = self.model.engine.store
        for g in self.param_groups:
            lr, wd = g["lr"], g["weight_decay"]
            for name in g["params"]:
                if name in self.model.head:
                    p = self.model.head[name]
                    if p.grad is None:
                        continue
                    s = self.head_state.setdefault(name, dict(m=torch.zeros_like(p), v=torch.zeros_like(p)))
                    with torch.no_grad():
                        p.mul_(1.0 - lr * wd)
                        s["m"].mul_(b1).add_(p.grad, alpha=1.0 - b1)
                        s["v"].mul_(b2).addcmul_(p.grad, p.grad, value=1.0 - b2)
                        p.addcdiv_(s["m"], (s["v"].sqrt() / math.sqrt(bc2)).add_(eps), value=-lr / bc1)
                    continue
                o = st.offsets[name]
                n = (int(np.prod(st.shapes[name])) + 7) // 8 * 8
                sl = slice(o, o + n)
                ops.adamw(st.p[sl], st.g[sl], st.m[sl], st.v[sl], st.p_lp[sl], n, n if wd != 0.0 else 0, None, b1, b2, eps, wd,
                          lr=lr, bc1=bc1, bc2=bc2)

    def state_dict(self):
        st = self.model.engine.store
        eng = {n: dict(exp_avg=st._view(st.m, n).detach().clone(), exp_avg_sq=st._view(st.v, n).detach().clone())
               for g in self.param_groups for n in g["params"] if n not in self.model.head} if self.step_count else {}
        head = {n: dict(exp_avg=s["m"].clone(), exp_avg_sq=s["v"].clone()) for n, s in self.head_state.items()}
        return {"state": {**eng, **head}, "step": self.step_count,
                "param_groups": [{k: (list(v) if k == "params" else v) for k, v in g.items()} for g in self.param_groups]}

    def load_state_dict(self, sd):
        st = self.model.engine.store
        self.step_count = int(sd.get("step", 0))
        for n, s in sd["state"].items():
            if n in self.model.head:
                self.head_state[n] = dict(m=s["exp_avg"].to(self.model.head[n].device).clone(), v=s["exp_avg_sq"].to(self.model.head[n].device).clone())
            else:
                st._view(st.m, n).copy_(s["exp_avg"])
                st._view(st.v, n).copy_(s["exp_avg_sq"])
        for g, saved in zip(self.param_groups, sd["param_groups"]):
            g.update({k: v for k, v in saved.items() if k != "params"})


class LinearLR:
    """torch.optim.lr_scheduler.LinearLR(optimizer, start_factor, end_factor, total_iters) in closed form."""

    def __init__(self, optimizer, start_factor=1.0, end_factor=1.0, total_iters=5):
        self.optimizer, self.start_factor, self.end_factor, self.total_iters = optimizer, start_factor, end_factor, int(total_iters)
        self.last_epoch = 0
        self._apply()

    def _factor(self):
        t = min(self.last_epoch, self.total_iters)
        return self.start_factor + (self.end_factor - self.start_factor) * t / self.total_iters

    def _apply(self):
        for g in self.optimizer.param_groups:
            g["lr"] = g["initial_lr"] * self._factor()

    def step(self):
        self.last_epoch += 1
        self._apply()

    def get_last_lr(self):
        return [g["lr"] for g in self.optimizer.param_groups]

    def state_dict(self):
        return {"start_factor": self.start_factor, "end_factor": self.end_factor, "total_iters": self.total_iters,
                "last_epoch": self.last_epoch, "_last_lr": self.get_last_lr()}

    def load_state_dict(self, sd):
        self.last_epoch = int(sd["last_epoch"])
        self._apply()


def build_optimizer(model, train_method, init_lr, weight_decay, layer_decay):
    """utils/vit.py:134-172 on a VisionTransformer of this module (``model`` = the object behind ``.module``)."""
    if train_method in ('finetune', 'ft'):
        print('\nUsing the fine-tuning training method...')
        # utils/vit.py:141-143 passes (model, weight_decay, ...) POSITIONALLY to param_groups_lrd(model, init_lr, weight_decay=0.05, ...):
        # the configured weight decay becomes the base lr of the groups and their weight decay stays 0.05.  Mirrored as written.
        groups, _ = param_groups_lrd(model, weight_decay, no_weight_decay_list=model.no_weight_decay(), layer_decay=layer_decay)
        return PredictorOptimizer(model, groups)
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
        num_labels = len(eval(data['label_keys']))
        if 'TRAINING' in config and str2bool(config['TRAINING'].get('use_label_errs', 'False')):
            num_labels = num_labels // 2
    else:
        num_labels = 0                           # similarity_search.py routes through forward_features only
    model = VisionTransformer(cfg, device, _compute_dtype(mae_config), num_classes=num_labels,
                              global_pool=config['ARCHITECTURE'].get('global_pool', 'token'),
                              label_means=eval(data['label_means']) if 'label_means' in data else (0.0,),
                              label_stds=eval(data['label_stds']) if 'label_stds' in data else (1.0,),
                              drop_rate=float(eval(config['ARCHITECTURE'].get('dropout', '0.0'))))
    model = _DataParallelShim(model)
    if not build_optimizer:
        return load_model(model, model_filename, mae_filename)
    tr = config['TRAINING']
    total = int(float(tr['total_batch_iters']))
    optimizer = globals()['build_optimizer'](model.module, tr['train_method'], float(tr['init_lr']), float(tr['weight_decay']),
                                             float(tr['layer_decay']))
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
        losses, cur_iter = defaultdict(list), 1
    else:
        print('\nStarting fresh model to train...')
        losses, cur_iter = defaultdict(list), 1
    return model, losses, cur_iter
