#!/usr/bin/env python3
"""Two eager steps of configs/mim_19.ini (SimMIM ViT-L/16, 5x128x128, bs 128, bf16; no HIP graph) -- a target for rocprofv3 --pmc
passes, which serialise every dispatch.  Prints a progress line per step."""
import configparser, os, sys
import numpy as np
import torch
LP = torch.bfloat16 if os.environ.get("SKYEMB_DTYPE", "f16") == "bf16" else torch.float16   # the headline's operand format (f16) unless SKYEMB_DTYPE=bf16
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sky_embeddings_amd.model_config import config_for
from sky_embeddings_amd.optim import CosineLR, FusedAdamW
from sky_embeddings_amd.simmim_engine import SimMIMEngine
from sky_embeddings_amd.train_step import TrainStep
ini = configparser.ConfigParser()
ini.read(os.path.join(ROOT, "configs", "mim_19.ini"))
a, t = ini["ARCHITECTURE"], ini["TRAINING"]
cfg = config_for(a["model_type"], img_size=int(a["img_size"]), patch_size=int(a["patch_size"]), in_chans=int(a["num_channels"]),
                 embed_dim=int(a["embed_dim"]), norm_pix_loss=t.getboolean("norm_pix_loss"), loss_fn=t["loss_fn"])
B, dev = int(t["batch_size"]), torch.device("cuda", 0)
eng = SimMIMEngine(cfg, device=dev, compute_dtype=LP, seed=0)
opt = FusedAdamW(eng, lr=float(t["init_lr"]), betas=(0.9, 0.95), weight_decay=float(t["weight_decay"]))
step = TrainStep(eng, opt, CosineLR(opt, 1_000_000), B, use_graph=False)
g = torch.Generator(device=dev).manual_seed(19)
x = torch.randn(B, cfg.in_chans, cfg.img_size, cfg.img_size, device=dev, generator=g).clamp_(min=-3.0)
L, p = cfg.num_patches, cfg.patch_size
count = int(np.ceil(L * float(t["max_mask_ratio"])))
order = torch.rand(B, cfg.in_chans, L, device=dev, generator=g).argsort(dim=2)
m = (order < count).float().view(B, cfg.in_chans, cfg.grid, cfg.grid).repeat_interleave(p, 2).repeat_interleave(p, 3).contiguous()
step.load_batch(x, m)
for i in range(2):
    loss = step()
    torch.cuda.synchronize()
    print("step", i, float(loss), flush=True)
