#!/usr/bin/env python3
"""Ceiling of a fusion, MEASURED: the config-A step (ViT-B/16 MAE, B = 256, bf16, HIP graph) captured with one family of launches
LEFT OUT (results are garbage; the remaining kernels do the same work on whatever the buffers hold), interleaved with the full step
in one process.  What a perfect fusion of that family into its neighbours could return is at most full - without:
  mha_fwd   the 20 forward attention launches (attention in the QKV GEMM's epilogue)
  ln_fwd    the 42 LayerNorm-forward launches (item 1c: LayerNorm folded into the consumer GEMM)
  mha_bwd   the 20 backward attention launches (attention backward in the proj data-gradient launch)
  loss      the three loss launches; ln_reduce  the batched dgamma / dbeta reduces
usage: skip_ceiling.py [rounds] [f16|bf16]"""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sky_embeddings_amd import ops
from sky_embeddings_amd.engine import MAEEngine
from sky_embeddings_amd.model_config import config_for
from sky_embeddings_amd.optim import CosineLR, FusedAdamW
from sky_embeddings_amd.train_step import TrainStep

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
T = torch.bfloat16 if (len(sys.argv) > 2 and sys.argv[2] == "bf16") else torch.float16
dev = torch.device("cuda", 0)
cfg = config_for("base", img_size=64, patch_size=16, in_chans=5, embed_dim=768, norm_pix_loss=True, loss_fn="mse")
B = 256
g = torch.Generator(device="cpu").manual_seed(1234)
pool = [torch.randn(B, 5, 64, 64, generator=g).clamp_(min=-3.0).to(dev) for _ in range(2)]


def build(skip):
    saved = {k: getattr(ops, k) for k in skip}
    for k in skip:
        setattr(ops, k, lambda *a, **kw: None)
    try:
        eng = MAEEngine(cfg, device=dev, compute_dtype=T, seed=0)
        opt = FusedAdamW(eng, lr=1e-4, betas=(0.9, 0.95), weight_decay=0.05)
        step = TrainStep(eng, opt, CosineLR(opt, 1_000_000, eta_min=1e-11), B, mask_ratio=0.75, use_graph=True)
    finally:
        for k, v in saved.items():
            setattr(ops, k, v)
    return step, eng, opt


variants = {"full": (), "without_mha_fwd": ("mha_fwd",), "without_ln_fwd": ("layernorm_fwd",), "without_both": ("mha_fwd", "layernorm_fwd"),
            "without_mha_bwd": ("mha_bwd",), "without_loss": ("masked_patch_loss",), "without_ln_reduce": ("layernorm_bwd_reduce_batch",)}
steps = {k: build(v) for k, v in variants.items()}
for s, _, _ in steps.values():
    for i in range(5):
        s(pool[i % 2])
res = {k: [] for k in steps}
for _ in range(rounds):
    for name, (s, _, _) in steps.items():
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(40):
            s(pool[i % 2])
        e1.record()
        e1.synchronize()
        res[name].append(e0.elapsed_time(e1) / 40)
mean = {k: sum(v) / len(v) for k, v in res.items()}
print(json.dumps(dict(ms_per_step=res, mean_ms=mean, ceiling_ms={k: mean["full"] - v for k, v in mean.items() if k != "full"},
                      launches_left_out={"without_mha_fwd": 20, "without_ln_fwd": 42, "without_both": 62, "without_mha_bwd": 20, "without_loss": 3,
                                         "without_ln_reduce": 2}, dtype=str(T))))
