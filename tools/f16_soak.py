#!/usr/bin/env python3
"""Soak of the fp16 training mode (round 6): the config-A MAE step for N optimiser steps on a fixed synthetic set (blobs + noise, a few
NaN pixels, one saturated source), fp16 / bf16 operands and the fp32 mode from the same seed -- loss every 250 steps, non-finite
parameters or losses abort.  The fp16 mode runs with its static loss scale; the point is that nothing overflows or drifts over thousands
of steps.  usage: f16_soak.py [steps] [batch]"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sky_embeddings_amd.engine import MAEEngine
from sky_embeddings_amd.model_config import config_for
from sky_embeddings_amd.optim import CosineLR, FusedAdamW
from sky_embeddings_amd.train_step import TrainStep
STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
rng = np.random.default_rng(7)
n = 4096
yy, xx = np.mgrid[0:64, 0:64]
imgs = rng.standard_normal((n, 5, 64, 64)).astype(np.float32) * 0.3
for i in range(n):                                     # one to three elliptical sources per cutout, colours correlated across the bands
    for _ in range(rng.integers(1, 4)):
        cy, cx, s, q = rng.uniform(16, 48), rng.uniform(16, 48), rng.uniform(1.5, 5.0), rng.uniform(0.5, 1.0)
        amp = rng.lognormal(1.0, 1.0) * (1 + 0.2 * rng.standard_normal(5))
        blob = np.exp(-(((yy - cy) / s) ** 2 + ((xx - cx) / (s * q)) ** 2) / 2).astype(np.float32)
        imgs[i] += amp[:, None, None].astype(np.float32) * blob
imgs[rng.random(imgs.shape) < 1e-4] = np.nan
imgs[0, :, 30:34, 30:34] = 3.0e4                       # a saturated star
data = torch.from_numpy(imgs).cuda().clamp_(min=-3.0)
cfg = config_for("base", img_size=64, patch_size=16, in_chans=5, embed_dim=768, norm_pix_loss=True, loss_fn="mse")
for name, dt in (("f16", torch.float16), ("bf16", torch.bfloat16), ("f32", torch.float32)):
    eng = MAEEngine(cfg, device="cuda", compute_dtype=dt, seed=0)
    opt = FusedAdamW(eng, lr=1.5e-4 * B / 256, betas=(0.9, 0.95), weight_decay=0.05)
    step = TrainStep(eng, opt, CosineLR(opt, STEPS), B, mask_ratio=0.75, use_graph=True)
    g = torch.Generator(device="cuda").manual_seed(1)
    steps = STEPS if dt != torch.float32 else min(STEPS, 500)      # (the fp32 mode is 6x slower: its first 500 steps as the reference curve)
    t0, out, acc = time.time(), [], []
    for it in range(steps):
        idx = torch.randint(0, n, (B,), device="cuda", generator=g)
        loss = step(data[idx])
        acc.append(loss)
        if (it + 1) % 250 == 0:
            v = float(torch.stack(acc).mean())
            acc = []
            out.append(v)
            if not np.isfinite(v):
                print(name, "NON-FINITE loss at step", it + 1)
                sys.exit(1)
    finite = bool(torch.isfinite(eng.store.p).all())
    print(f"{name:5s} loss scale {getattr(eng, 'loss_scale', 1.0):g}  mean loss per 250 steps: " + " ".join(f"{v:.4f}" for v in out) +
          f"  | parameters finite: {finite}  ({time.time() - t0:.0f} s)", flush=True)
    if not finite:
        sys.exit(1)
    del step, opt, eng
    torch.cuda.empty_cache()


def soak_mim19(steps):
    """The same for configs/mim_19.ini (SimMIM ViT-L/16 on 5 x 128 x 128, batch 128, 24 blocks): fp16 beside bf16."""
    import configparser
    from sky_embeddings_amd.simmim_engine import SimMIMEngine
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ini = configparser.ConfigParser()
    ini.read(os.path.join(root, "configs", "mim_19.ini"))
    a, t = ini["ARCHITECTURE"], ini["TRAINING"]
    cfg19 = config_for(a["model_type"], img_size=int(a["img_size"]), patch_size=int(a["patch_size"]), in_chans=int(a["num_channels"]),
                       embed_dim=int(a["embed_dim"]), norm_pix_loss=t.getboolean("norm_pix_loss"), loss_fn=t["loss_fn"])
    Bm = int(t["batch_size"])
    L, p = cfg19.num_patches, cfg19.patch_size
    count = int(np.ceil(L * float(t["max_mask_ratio"])))
    big = torch.nn.functional.interpolate(data[:1024], size=(cfg19.img_size, cfg19.img_size), mode="bilinear").nan_to_num_(nan=float("nan"))
    for name, dt in (("f16", torch.float16), ("bf16", torch.bfloat16)):
        eng = SimMIMEngine(cfg19, device="cuda", compute_dtype=dt, seed=0)
        opt = FusedAdamW(eng, lr=float(t["init_lr"]), betas=(0.9, 0.95), weight_decay=float(t["weight_decay"]))
        step = TrainStep(eng, opt, CosineLR(opt, steps), Bm)
        g = torch.Generator(device="cuda").manual_seed(2)
        t0, out, acc = time.time(), [], []
        for it in range(steps):
            idx = torch.randint(0, big.shape[0], (Bm,), device="cuda", generator=g)
            order = torch.rand(Bm, cfg19.in_chans, L, device="cuda", generator=g).argsort(dim=2)
            m = (order < count).float().view(Bm, cfg19.in_chans, cfg19.grid, cfg19.grid).repeat_interleave(p, 2).repeat_interleave(p, 3).contiguous()
            step.load_batch(big[idx], m)
            acc.append(step())
            if (it + 1) % 100 == 0:
                out.append(float(torch.stack(acc).mean()))
                acc = []
                if not np.isfinite(out[-1]):
                    print("mim_19", name, "NON-FINITE loss at step", it + 1)
                    sys.exit(1)
        finite = bool(torch.isfinite(eng.store.p).all())
        print(f"mim_19 {name:5s} loss scale {getattr(eng, 'loss_scale', 1.0):g}  mean loss per 100 steps: " + " ".join(f"{v:.4f}" for v in out) +
              f"  | parameters finite: {finite}  ({time.time() - t0:.0f} s)", flush=True)
        if not finite:
            sys.exit(1)
        del step, opt, eng
        torch.cuda.empty_cache()


if os.environ.get("SOAK_MIM19", "1") != "0":
    soak_mim19(int(os.environ.get("SOAK_MIM19_STEPS", "600")))
