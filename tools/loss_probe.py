#!/usr/bin/env python3
"""us per call of the MAE loss (skyemb_masked_patch_loss: two launches) at config A's shape, 20 calls per HIP graph; and by rocprofv3
when run under it.  usage: loss_probe.py [f16|bf16]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sky_embeddings_amd import ops
T = torch.bfloat16 if (len(sys.argv) > 1 and sys.argv[1] == "bf16") else torch.float16
B, C, H, p, L = 256, 5, 64, 16, 16
pv = C * p * p
g = torch.Generator(device="cuda").manual_seed(0)
imgs = torch.randn(B, C, H, H, device="cuda", generator=g).clamp_(min=-3)
pred = torch.randn(B, L + 1, pv, device="cuda", generator=g)
mask = (torch.rand(B, L, device="cuda", generator=g) < 0.75).float()
loss = torch.zeros(1, device="cuda")
ws = torch.zeros(4 * B * L + 4, device="cuda")
dp = torch.empty(B * (L + 1), pv, device="cuda", dtype=T)
f = lambda: ops.masked_patch_loss(imgs, pred, mask, loss, dp, None, ops.dtype_code(T), ws, p, 1, 0.0, 1.0, True, False, dscale=65536.0)
for _ in range(3):
    f()
torch.cuda.synchronize()
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    for _ in range(20):
        f()
gr.replay()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(10):
    gr.replay()
e.record()
e.synchronize()
print(f"masked_patch_loss B={B}: {s.elapsed_time(e) / 200 * 1e3:.2f} us per call (two launches), loss {float(loss):.6f}")
