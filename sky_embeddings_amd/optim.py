"""Optimiser + schedule of the pretraining path (utils/mim_vit.py:119-144) on the flat buffers.

``FusedAdamW`` = torch.optim.AdamW(param_groups_weight_decay(model, wd), lr, betas=(0.9, 0.95))
as ONE kernel launch over the engine's flat parameter buffer; ``CosineLR`` =
torch.optim.lr_scheduler.CosineAnnealingLR(T_max, eta_min) in closed form.  Both expose
``step / zero_grad / state_dict / load_state_dict`` with torch-compatible state layouts so that
checkpoints written by either implementation load in the other.
"""
from __future__ import annotations

import math

import numpy as np
import torch

from . import ops


class FusedAdamW:
    def __init__(self, engine, lr=1e-4, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.05, process_group=None):
        self.engine = engine
        self.store = engine.store
        self.defaults = dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay)
        # group order == timm param_groups_weight_decay: [no_decay (wd 0), decay (wd)]
        self.param_groups = [dict(lr=lr, initial_lr=lr, betas=tuple(betas), eps=eps, weight_decay=0.0),
                             dict(lr=lr, initial_lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay)]
        self.step_count = 0
        # gradients are multiplied by grad_scale on their way into the update: base_grad_scale (1 / world size: the DDP mean) over the
        # engine's loss scale (fp16 mode: the flat gradient buffer holds loss_scale x the gradients; the scale follows the batch)
        self.base_grad_scale = 1.0
        # graph mode: step scalars live in device memory (kernel arguments are frozen in a HIP graph)
        self.hyper_device = None
        self.process_group = process_group
        # gradient source of the update: the engine's flat fp32 buffer, or (set by TrainStep for bf16 gradient
        # communication) a flat bf16 copy of it that the all-reduce summed over the ranks
        self.grad_buffer = None

    @property
    def grad_scale(self):
        return self.base_grad_scale / getattr(self.engine, "loss_scale", 1.0)

    @grad_scale.setter
    def grad_scale(self, v):          # (callers that set the whole factor: kept for the predictor / tests; the loss scale is divided back in)
        self.base_grad_scale = float(v) * getattr(self.engine, "loss_scale", 1.0)

    @property
    def lr(self):
        return self.param_groups[0]["lr"]

    def set_lr(self, lr):
        for g in self.param_groups:
            g["lr"] = lr

    def zero_grad(self, set_to_none=True):
        # every gradient is overwritten by the next backward(); nothing to clear
        return None

    def step(self):
        self.begin_step()
        self.apply_range(0, self.store.n)

    def use_device_scalars(self, device):
        """Keep (lr, 1-b1^t, 1-b2^t) of the running step in device memory (kernel arguments are frozen inside a captured HIP
        graph; the optimiser step fused into the weight-gradient launches reads them from there)."""
        if self.hyper_device is None:
            self.hyper_device = torch.zeros(4, device=device, dtype=torch.float32)
        self.write_scalars(max(self.step_count, 1))
        return self.hyper_device

    def write_scalars(self, t):
        """Device scalars of optimiser step t: a one-thread kernel launch on the current stream whose ARGUMENTS carry the values
        (ordered before the launches that follow; nothing on the host is read later, so the host may run steps ahead)."""
        lr, bc1, bc2 = self.step_scalars(t)
        ops.set_scalars(self.hyper_device, lr, bc1, bc2)

    def begin_step(self):
        """Open optimiser step t; follow with apply_range() calls that together cover [0, n) exactly once (a backward
        stage's slices can be updated while later stages still compute: see TrainStep)."""
        self.step_count += 1
        if self.hyper_device is not None:
            # device-resident step scalars (a fused TrainStep switched them on): every path that opens a step refreshes them,
            # or an eager optimizer.step() after graph-mode steps would read the scalars of an older step
            self.write_scalars(self.step_count)

    def apply_range(self, start, end, grad=None):
        """AdamW update of flat elements [start, end) on the current stream (start, end multiples of 8: tensor bounds).
        grad: the gradients of exactly these elements in a buffer of their own (a reduce-scatter's output: TrainStep with the
        optimiser sharded over the ranks); default: the same slice of the flat gradient buffer / its 16-bit mirror."""
        b1, b2 = self.defaults["betas"]
        t = self.step_count
        st = self.store
        n = end - start
        if n <= 0:
            return
        n_decay = min(max(st.n_decay - start, 0), n)     # layout is [decayed | not decayed]
        sl = slice(start, end)
        g = st.g if self.grad_buffer is None else self.grad_buffer
        if grad is not None:
            g, sl_g = grad, slice(0, n)
        else:
            sl_g = sl
        ops.adamw(st.p[sl], g[sl_g], st.m[sl], st.v[sl], st.p_lp[sl], n, n_decay, self.hyper_device, b1, b2,
                  self.defaults["eps"], self.param_groups[1]["weight_decay"], grad_scale=self.grad_scale, zero_grad=False,
                  lr=self.lr, bc1=1.0 - b1 ** t, bc2=1.0 - b2 ** t)

    def step_scalars(self, t=None):
        """(lr, 1-b1^t, 1-b2^t) of optimiser step t (default: the next one) for graph-mode callers."""
        b1, b2 = self.defaults["betas"]
        t = self.step_count + 1 if t is None else t
        return self.lr, 1.0 - b1 ** t, 1.0 - b2 ** t

    # ---- torch.optim.AdamW-compatible (de)serialisation --------------------------------------
    def _index(self):
        """Parameter names in torch's id order: timm param_groups_weight_decay walks named_parameters() once and fills
        [no_decay, decay].  SimMIM's ``mask_token`` (requires_grad, ndim 3 -> decay group, utils/mim_vit.py:264) belongs
        to that list although no forward uses it: its id exists in ``param_groups`` and it has no ``state`` entry
        (torch keeps none for a parameter whose grad is None).  ``None`` marks such a stateless id."""
        from .model_config import FROZEN, state_layout
        st = self.store
        stateless = {n for n in st.frozen if n not in FROZEN}
        if not stateless:
            return list(st.no_decay), list(st.decay)
        no_decay, decay = [], []
        have = set(st.no_decay) | set(st.decay)
        for name, shape in state_layout(st.cfg):
            if name in have:
                (no_decay if name in st.no_decay else decay).append(name)
            elif name in stateless:
                (no_decay if (len(shape) <= 1 or name.endswith(".bias")) else decay).append(None)
        return no_decay, decay

    def state_dict(self):
        st = self.store
        no_decay, decay = self._index()
        names = no_decay + decay
        state = {}
        if self.step_count > 0:
            for i, n in enumerate(names):
                if n is None:
                    continue
                state[i] = {"step": torch.tensor(float(self.step_count)),
                            "exp_avg": st._view(st.m, n).detach().clone(),
                            "exp_avg_sq": st._view(st.v, n).detach().clone()}
        nd = len(no_decay)
        groups = []
        for gi, ids in enumerate((list(range(nd)), list(range(nd, len(names))))):
            g = dict(self.param_groups[gi])
            g.update(amsgrad=False, foreach=None, maximize=False, capturable=False, differentiable=False, fused=None,
                     params=ids)
            groups.append(g)
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, sd):
        st = self.store
        no_decay, decay = self._index()
        names = no_decay + decay
        groups = sd["param_groups"]
        assert [len(g["params"]) for g in groups] == [len(no_decay), len(decay)], \
            "optimizer state does not match this model (parameter counts per weight-decay group differ)"
        for g_new, g_old in zip(self.param_groups, groups):
            for k in ("lr", "initial_lr", "weight_decay", "eps"):
                if k in g_old:
                    g_new[k] = g_old[k]
            if "betas" in g_old:
                g_new["betas"] = tuple(g_old["betas"])
        self.defaults["betas"] = self.param_groups[0]["betas"]
        steps = set()
        for i, n in enumerate(names):
            s = sd["state"].get(i)
            if s is None or n is None:
                continue
            st._view(st.m, n).copy_(s["exp_avg"].to(torch.float32))
            st._view(st.v, n).copy_(s["exp_avg_sq"].to(torch.float32))
            steps.add(int(float(s["step"])))
        assert len(steps) <= 1, "per-parameter step counts differ; unsupported"
        self.step_count = steps.pop() if steps else 0


class CosineLR:
    """CosineAnnealingLR(optimizer, T_max, eta_min) (utils/mim_vit.py:142-144)."""

    def __init__(self, optimizer, T_max, eta_min=0.0):
        self.optimizer = optimizer
        self.T_max = int(T_max)
        self.eta_min = float(eta_min)
        self.base_lr = float(optimizer.param_groups[0].get("initial_lr", optimizer.lr))
        self.last_epoch = 0
        self._apply()

    def _lr_at(self, t):
        return self.eta_min + (self.base_lr - self.eta_min) * (1 + math.cos(math.pi * t / self.T_max)) / 2

    def _apply(self):
        self.optimizer.set_lr(self._lr_at(self.last_epoch))

    def step(self):
        self.last_epoch += 1
        self._apply()

    def get_last_lr(self):
        return [self.optimizer.lr for _ in self.optimizer.param_groups]

    def state_dict(self):
        return {"T_max": self.T_max, "eta_min": self.eta_min, "base_lrs": [self.base_lr] * 2,
                "last_epoch": self.last_epoch, "_step_count": self.last_epoch + 1, "_last_lr": self.get_last_lr()}

    def load_state_dict(self, sd):
        self.T_max = int(sd.get("T_max", self.T_max))
        self.eta_min = float(sd.get("eta_min", self.eta_min))
        if "base_lrs" in sd:
            self.base_lr = float(sd["base_lrs"][0])
        self.last_epoch = int(sd["last_epoch"])
        self._apply()
