#!/usr/bin/env python3
"""Times every GEMM shape of one MAE ViT-B/16 step (B=256) through the C ABI with HIP events.
usage: python tools/gemm_bench.py [--tile 0|64|128] [--dtype bf16|f32] [--iters 30]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sky_embeddings_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--tile", type=int, default=0)
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--iters", type=int, default=30)
ap.add_argument("--batch", type=int, default=256)
args = ap.parse_args()
T = torch.bfloat16 if args.dtype == "bf16" else torch.float32
dev = "cuda"
B = args.batch
Me, Md = B * 5, B * 17
layers = []  # (name, M_tokens, N_out, K_in, count_per_step, epilogue)
for tag, M, D, depth in (("enc", Me, 768, 12), ("dec", Md, 512, 8)):
    layers += [(f"{tag}.qkv", M, 3 * D, D, depth, "bias"), (f"{tag}.proj", M, D, D, depth, "resid"),
               (f"{tag}.fc1", M, 4 * D, D, depth, "gelu"), (f"{tag}.fc2", M, D, 4 * D, depth, "resid")]
layers += [("patch_embed", B * 4, 768, 1280, 1, "bias"), ("dec_embed", Me, 512, 768, 1, "bias"),
           ("dec_pred", Md, 1280, 512, 1, "bias")]


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


tot = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0}
flops_tot = 0.0
print(f"{'layer':14s} {'M':>5s} {'N':>5s} {'K':>5s} | {'fwd us':>8s} {'TF':>6s} | {'dgrad us':>8s} {'TF':>6s} | {'wgrad us':>8s} {'TF':>6s}")
for name, M, N, K, cnt, epi in layers:
    x = torch.randn(M, K, device=dev).to(T)
    w = (torch.randn(N, K, device=dev) * 0.05).to(T)
    dy = torch.randn(M, N, device=dev).to(T)
    bias = torch.zeros(N, device=dev)
    y = torch.empty(M, N, device=dev, dtype=T)
    y2 = torch.empty(M, N, device=dev, dtype=T)
    y32 = torch.empty(M, N, device=dev)
    res = torch.randn(M, N, device=dev)
    dx = torch.empty(M, K, device=dev, dtype=T)
    dw = torch.empty(N, K, device=dev)
    db = torch.empty(N, device=dev)
    ws = torch.zeros(8 * 1024 * 1024, device=dev)
    aux = torch.randn(M, K, device=dev).to(T)
    if epi == "gelu":
        fwd = lambda: ops.gemm(x, w, M=M, N=N, K=K, bias=bias, act=ops.ACT_GELU, out=y, out2=y2, tile=args.tile)
    elif epi == "resid":
        fwd = lambda: ops.gemm(x, w, M=M, N=N, K=K, bias=bias, resid=res, ldr=N, out_f32=y32, tile=args.tile, ws=ws)
    else:
        fwd = lambda: ops.gemm(x, w, M=M, N=N, K=K, bias=bias, out=y, tile=args.tile)
    if name.endswith("fc2"):
        dgrad = lambda: ops.gemm(dy, w, M=M, N=K, K=N, a_layout=ops.KC, b_layout=ops.RC, lda=N, ldb=K, act=ops.ACT_DGELU,
                                 aux=aux, ldaux=K, out=dx, tile=args.tile, ws=ws)
    else:
        dgrad = lambda: ops.gemm(dy, w, M=M, N=K, K=N, a_layout=ops.KC, b_layout=ops.RC, lda=N, ldb=K, out=dx, tile=args.tile, ws=ws)
    wgrad = lambda: ops.gemm(dy, x, M=N, N=K, K=M, a_layout=ops.RC, b_layout=ops.RC, lda=N, ldb=K, out_f32=dw, colsum_a=db,
                             tile=args.tile, ws=ws)
    fl = 2.0 * M * N * K
    t = [timeit(f, args.iters) for f in (fwd, dgrad, wgrad)]
    for k_, v in zip(("fwd", "dgrad", "wgrad"), t):
        tot[k_] += v * cnt
    flops_tot += 3 * fl * cnt
    print(f"{name:14s} {M:5d} {N:5d} {K:5d} | {t[0]:8.1f} {fl/t[0]/1e6:6.0f} | {t[1]:8.1f} {fl/t[1]/1e6:6.0f} | {t[2]:8.1f} {fl/t[2]/1e6:6.0f}")
s = sum(tot.values())
print(f"per-step GEMM time: fwd {tot['fwd']/1e3:.2f} ms, dgrad {tot['dgrad']/1e3:.2f} ms, wgrad {tot['wgrad']/1e3:.2f} ms, total {s/1e3:.2f} ms "
      f"-> {flops_tot/s/1e6:.0f} TFLOP/s over {flops_tot/1e12:.3f} TFLOP")
