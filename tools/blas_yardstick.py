#!/usr/bin/env python3
"""Yardstick only (not a product path): the vendor BLAS (torch.matmul -> hipBLASLt/rocBLAS) on the GEMM shapes of one
MAE ViT-B step, plain GEMM without the fused epilogues, timed with HIP events under the same eager launch floor."""
import os, sys
import torch
B = 256
Me, Md = B * 5, B * 17
layers = []
for tag, M, D, depth in (("enc", Me, 768, 12), ("dec", Md, 512, 8)):
    layers += [(f"{tag}.qkv", M, 3 * D, D, depth), (f"{tag}.proj", M, D, D, depth), (f"{tag}.fc1", M, 4 * D, D, depth), (f"{tag}.fc2", M, D, 4 * D, depth)]
layers += [("patch_embed", B * 4, 768, 1280, 1), ("dec_embed", Me, 512, 768, 1), ("dec_pred", Md, 1280, 512, 1)]
def timeit(f, iters=20):
    """us per call, 20 calls captured into one HIP graph (no host launch floor), replayed 5x"""
    for _ in range(3): f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20): f()
    g.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5): g.replay()
    e.record(); e.synchronize()
    return s.elapsed_time(e) / 100 * 1e3


tot = [0.0, 0.0, 0.0]; fl_tot = 0.0
for name, M, N, K, cnt in layers:
    x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16); w = torch.randn(N, K, device="cuda", dtype=torch.bfloat16)
    dy = torch.randn(M, N, device="cuda", dtype=torch.bfloat16)
    y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16); dx = torch.empty(M, K, device="cuda", dtype=torch.bfloat16)
    dw = torch.empty(N, K, device="cuda", dtype=torch.bfloat16)
    t = [timeit(lambda: torch.matmul(x, w.t(), out=y)), timeit(lambda: torch.matmul(dy, w, out=dx)), timeit(lambda: torch.matmul(dy.t(), x, out=dw))]
    fl = 2.0 * M * N * K
    for i in range(3): tot[i] += t[i] * cnt
    fl_tot += 3 * fl * cnt
    print(f"{name:12s} {M:5d} {N:5d} {K:5d} | fwd {t[0]:6.1f} us {fl/t[0]/1e6:5.0f} TF | dgrad {t[1]:6.1f} us {fl/t[1]/1e6:5.0f} TF | wgrad {t[2]:6.1f} us {fl/t[2]/1e6:5.0f} TF")
s = sum(tot)
print(f"vendor BLAS per-step GEMM time: fwd {tot[0]/1e3:.2f} dgrad {tot[1]/1e3:.2f} wgrad {tot[2]/1e3:.2f} total {s/1e3:.2f} ms -> {fl_tot/s/1e6:.0f} TFLOP/s")
