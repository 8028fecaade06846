#!/usr/bin/env python3
"""Three eager training steps (no HIP graph, ~1100 kernel launches) -- a small target for rocprofv3 --pmc passes, which
serialise every dispatch.  Prints a progress line per step."""
import os, sys
import torch
LP = torch.bfloat16 if os.environ.get("SKYEMB_DTYPE", "f16") == "bf16" else torch.float16   # the headline's operand format (f16) unless SKYEMB_DTYPE=bf16
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sky_embeddings_amd.engine import MAEEngine
from sky_embeddings_amd.model_config import config_for
from sky_embeddings_amd.optim import CosineLR, FusedAdamW
from sky_embeddings_amd.train_step import TrainStep
cfg = config_for("base", img_size=64, patch_size=16, in_chans=5, embed_dim=768, norm_pix_loss=True, loss_fn="mse")
eng = MAEEngine(cfg, device="cuda", compute_dtype=LP, seed=0)
opt = FusedAdamW(eng, lr=1e-4, betas=(0.9, 0.95), weight_decay=0.05)
step = TrainStep(eng, opt, CosineLR(opt, 1_000_000), 256, mask_ratio=0.75, use_graph=False)
imgs = torch.randn(256, 5, 64, 64, device="cuda").clamp_(min=-3.0)
for i in range(3):
    loss = step(imgs)
    torch.cuda.synchronize()
    print("step", i, float(loss), flush=True)
