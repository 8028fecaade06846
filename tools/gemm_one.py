#!/usr/bin/env python3
"""Runs ONE GEMM shape repeatedly (for rocprofv3 --pmc runs). usage: gemm_one.py M N K tile [iters]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sky_embeddings_amd import ops
M, N, K, tile = [int(v) for v in sys.argv[1:5]]
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 20
x = torch.randn(M, K, device="cuda").bfloat16()
w = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
bias = torch.zeros(N, device="cuda")
y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
for _ in range(iters):
    ops.gemm(x, w, M=M, N=N, K=K, bias=bias, out=y, tile=tile)
torch.cuda.synchronize()
print("done")
