#!/usr/bin/env python3
"""Launch anatomy of the pipelined GEMM: time vs K at fixed M, N (slope = steady-state k-step cost, intercept = launch +
pipeline fill + epilogue).  usage: python tools/gemm_anatomy.py [tile]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sky_embeddings_amd import ops
tile = int(sys.argv[1]) if len(sys.argv) > 1 else 0
T = torch.bfloat16
def timeit(f, iters=50):
    for _ in range(5): f()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): f()
    e.record(); e.synchronize()
    return s.elapsed_time(e) / iters * 1e3
for M, N in ((1280, 2304), (1280, 768), (4352, 2048), (4352, 512)):
    out = []
    for K in (64, 128, 256, 512, 1024, 2048, 4096):
        x = torch.randn(M, K, device="cuda").to(T); w = torch.randn(N, K, device="cuda").to(T)
        b = torch.zeros(N, device="cuda"); y = torch.empty(M, N, device="cuda", dtype=T)
        f = lambda: ops.gemm(x, w, M=M, N=N, K=K, bias=b, out=y, tile=tile, split_k=1)
        out.append((K, timeit(f)))
    tiles = ((M + 63) // 64) * ((N + 63) // 64)
    slope = (out[-1][1] - out[-2][1]) / ((out[-1][0] - out[-2][0]) / 64)
    print(f"M={M} N={N} tiles64={tiles}: " + "  ".join(f"K={k}:{t:.1f}us" for k, t in out) + f"  | slope {slope*1e3:.0f} ns/kstep -> L2->LDS {tiles*16384/slope/1e6:.1f} TB/s (64^2 tiles)")
# empty-ish kernel launch cost: tiny GEMM
x = torch.randn(64, 64, device="cuda").to(T); w = torch.randn(64, 64, device="cuda").to(T); y = torch.empty(64, 64, device="cuda", dtype=T)
print("64x64x64 launch: %.1f us" % timeit(lambda: ops.gemm(x, w, M=64, N=64, K=64, out=y, split_k=1)))
