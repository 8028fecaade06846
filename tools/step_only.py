#!/usr/bin/env python3
"""Only the MAE ViT-B/16 training step (HIP-graph replay), 30 timed steps -- a clean target for rocprofv3 --kernel-trace
(divide the kernel_stats totals by 33 = 2 warm-up + 1 capture + 30 replays)."""
import os, sys, time
import torch
LP = torch.bfloat16 if os.environ.get("SKYEMB_DTYPE", "f16") == "bf16" else torch.float16   # the headline's operand format (f16) unless SKYEMB_DTYPE=bf16
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sky_embeddings_amd.engine import MAEEngine
from sky_embeddings_amd.model_config import config_for
from sky_embeddings_amd.optim import CosineLR, FusedAdamW
from sky_embeddings_amd.train_step import TrainStep
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
cfg = config_for("base", img_size=64, patch_size=16, in_chans=5, embed_dim=768, norm_pix_loss=True, loss_fn="mse")
eng = MAEEngine(cfg, device="cuda", compute_dtype=LP, seed=0)
opt = FusedAdamW(eng, lr=1e-4, betas=(0.9, 0.95), weight_decay=0.05)
step = TrainStep(eng, opt, CosineLR(opt, 1_000_000), B, mask_ratio=0.75, use_graph=os.environ.get("SKYEMB_NO_GRAPH", "0") != "1")
imgs = torch.randn(B, 5, 64, 64, device="cuda").clamp_(min=-3.0)
step(imgs); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(30):
    loss = step(imgs)
torch.cuda.synchronize()
print("ms/step %.3f loss %.5f" % ((time.perf_counter() - t0) / 30 * 1e3, float(loss)))
