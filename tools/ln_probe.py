#!/usr/bin/env python3
"""us per launch of LayerNorm forward / backward at the step's shapes (20 launches per HIP graph) and the HBM rate of the backward.
usage: ln_probe.py [f16|bf16]"""
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


for M, D in ((8320, 1024), (4352, 512), (1280, 768)):
    x = torch.randn(M, D, device="cuda")
    gam, bet = torch.randn(D, device="cuda"), torch.randn(D, device="cuda")
    y = torch.empty(M, D, device="cuda", dtype=T)
    mean, rstd = torch.empty(M, device="cuda"), torch.empty(M, device="cuda")
    dy = torch.randn(M, D, device="cuda").to(T)
    g = torch.randn(M, D, device="cuda")
    glp = torch.empty(M, D, device="cuda", dtype=T)
    nblk = ops.layernorm_bwd_blocks(M)
    part = torch.empty(2, nblk, D, device="cuda")
    tf = timeit(lambda: ops.layernorm_fwd(x, gam, bet, y, mean, rstd, M, D, 1e-6))
    tb = timeit(lambda: ops.layernorm_bwd(dy, x, gam, mean, rstd, g, g, glp, part, None, None, M, D, ops.dtype_code(T)))
    byt = M * D * (2 + 4 + 4 + 4 + 2)
    print(f"[{M} x {D}] blocks {nblk}: fwd {tf:6.2f} us  bwd {tb:6.2f} us = {byt / tb / 1e6:5.2f} TB/s", flush=True)
