#!/usr/bin/env python3
"""Experiment: one B=256 step vs two independent B=128 steps replayed concurrently on two streams (do the launch gaps /
latency chains of one lane fill with the other lane's work?)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sky_embeddings_amd.engine import MAEEngine
from sky_embeddings_amd.model_config import config_for
from sky_embeddings_amd.optim import CosineLR, FusedAdamW
from sky_embeddings_amd.train_step import TrainStep
cfg = config_for("base", img_size=64, patch_size=16, in_chans=5, embed_dim=768, norm_pix_loss=True, loss_fn="mse")


def make(B):
    eng = MAEEngine(cfg, device="cuda", compute_dtype=torch.bfloat16, seed=0)
    opt = FusedAdamW(eng, lr=1e-4, betas=(0.9, 0.95), weight_decay=0.05)
    return TrainStep(eng, opt, CosineLR(opt, 1_000_000), B, mask_ratio=0.75), torch.randn(B, 5, 64, 64, device="cuda").clamp_(min=-3.0)


def timeit(fn, n=30):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


one, x = make(256)
print("one lane  B=256: %.3f ms/step" % timeit(lambda: one(x)))
a, xa = make(128)
print("one lane  B=128: %.3f ms/step" % timeit(lambda: a(xa)))
b, xb = make(128)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def both():
    with torch.cuda.stream(s1):
        a(xa)
    with torch.cuda.stream(s2):
        b(xb)


print("two lanes B=128 + B=128 concurrently: %.3f ms per pair" % timeit(both))
