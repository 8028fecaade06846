#!/usr/bin/env python3
"""Step time of the reference's shipped configuration (configs/mim_32_shipped.ini: SimMIM ViT-Large/8, 9 bands, 64x64,
RA/Dec token, L1 + norm-pix) on one MI355X: forward + backward + AdamW through the HIP-graph TrainStep.
usage: python tools/simmim_bench.py [batch ...]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sky_embeddings_amd.model_config import config_for
from sky_embeddings_amd.optim import CosineLR, FusedAdamW
from sky_embeddings_amd.simmim_engine import SimMIMEngine
from sky_embeddings_amd.train_step import TrainStep
batches = [int(a) for a in sys.argv[1:]] or [32, 128]
cfg = config_for("mimlarge", img_size=64, patch_size=8, in_chans=9, embed_dim=1024, norm_pix_loss=True, loss_fn="L1", ra_dec=True)
for B in batches:
    eng = SimMIMEngine(cfg, device="cuda", compute_dtype=torch.bfloat16, seed=0)
    opt = FusedAdamW(eng, lr=1e-4, betas=(0.9, 0.95), weight_decay=0.05)
    step = TrainStep(eng, opt, CosineLR(opt, 1_000_000), B)
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn(B, 9, 64, 64, device="cuda", generator=g).clamp_(min=-3.0)
    x[:, 5:7][torch.rand(B, 2, device="cuda", generator=g) < 0.3] = float("nan")      # missing narrow bands
    m = (torch.rand(B, 9, 8, 8, device="cuda", generator=g) < 0.45).float().repeat_interleave(8, 2).repeat_interleave(8, 3).contiguous()
    rd = torch.stack([torch.rand(B, device="cuda", generator=g) * 360, torch.rand(B, device="cuda", generator=g) * 180 - 90], 1)
    for _ in range(3):
        loss = step(x, m, rd)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 20
    for _ in range(n):
        loss = step(x, m, rd)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    ex, alg = eng.flops_per_image()
    print(f"B={B}: {dt*1e3:.2f} ms/step  {B/dt:.0f} img/s  {B/dt*ex/1e12:.0f} TFLOP/s executed ({ex/1e9:.1f} GFLOP/img)  loss {float(loss):.4f}  params {eng.store.n/1e6:.1f} M")
    del step, opt, eng
    torch.cuda.empty_cache()
