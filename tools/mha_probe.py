#!/usr/bin/env python3
"""us per launch of the attention kernels at the step's shapes (20 launches per HIP graph).  SKYEMB_MHA_WPB=1|4 forces the waves per
workgroup of the packed (N <= 32) MFMA kernels (read once per process: run the script once per setting).
usage: mha_probe.py [f16|bf16]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sky_embeddings_amd import ops
T = torch.bfloat16 if (len(sys.argv) > 1 and sys.argv[1] == "bf16") else torch.float16


def timeit(f):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20):
            f()
    g.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10):
        g.replay()
    e.record()
    e.synchronize()
    return s.elapsed_time(e) / 200 * 1e3


print(f"# SKYEMB_MHA_WPB={os.environ.get('SKYEMB_MHA_WPB', 'auto')} {T}")
for tag, B, N, H, hd in (("config A encoder", 256, 5, 12, 64), ("config A decoder", 256, 17, 16, 32), ("B=64 encoder", 64, 5, 12, 64),
                         ("mim_19", 128, 65, 16, 64), ("predictor ViT-B 17 tokens", 256, 17, 12, 64)):
    D = H * hd
    qkv = torch.randn(B, N, 3 * D, device="cuda").to(T)
    dout = torch.randn(B, N, D, device="cuda").to(T)
    out, dqkv = torch.empty_like(dout), torch.empty_like(qkv)
    print(f"{tag:28s} B {B} N {N} H {H} hd {hd}: fwd {timeit(lambda: ops.mha_fwd(qkv, out, B, N, H, hd)):6.2f} us  "
          f"bwd {timeit(lambda: ops.mha_bwd(qkv, dout, dqkv, B, N, H, hd)):6.2f} us", flush=True)
