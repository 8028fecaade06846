#!/usr/bin/env python3
"""Yardstick only (not a product path): the GEMM shapes of one mim_19 block (ViT-L/16, 8320 token rows) -- this library's launches
as the step issues them (forward with bias / GELU / residual epilogues, data gradients against the row-contiguous weight) beside
the vendor BLAS (torch.matmul -> hipBLASLt, plain GEMMs), each 20x in one HIP graph.  Also: the fc2 data gradient with a
k-contiguous (pre-transposed) copy of the weight, the question DESIGN.md section 7 left open.
usage: vitl_yardstick.py  (one MI355X)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sky_embeddings_amd import ops
from sky_embeddings_amd.ops import ACT_DGELU, ACT_GELU, KC, RC
M, D = 8320, 1024
dev = "cuda"


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
    for _ in range(5):
        g.replay()
    e.record()
    e.synchronize()
    return s.elapsed_time(e) / 100 * 1e3


ws = torch.zeros(8 * 1024 * 1024, device=dev)
rows = []
for name, N, K, epi in (("qkv", 3 * D, D, "bias"), ("proj", D, D, "resid"), ("fc1", 4 * D, D, "gelu"), ("fc2", D, 4 * D, "resid")):
    x = torch.randn(M, K, device=dev).bfloat16()
    w = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    wt = w.t().contiguous()                       # [K, N]: the weight as a k-contiguous B operand of the data gradient
    dy = torch.randn(M, N, device=dev).bfloat16()
    bias = torch.zeros(N, device=dev)
    y, y2 = torch.empty(M, N, device=dev, dtype=torch.bfloat16), torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    y32, res = torch.empty(M, N, device=dev), torch.randn(M, N, device=dev)
    dx, aux = torch.empty(M, K, device=dev, dtype=torch.bfloat16), torch.randn(M, K, device=dev).bfloat16()
    dw = torch.empty(N, K, device=dev, dtype=torch.bfloat16)
    if epi == "gelu":
        fwd = lambda: ops.gemm(x, w, M=M, N=N, K=K, bias=bias, act=ACT_GELU, out=y, out2=y2)
    elif epi == "resid":
        fwd = lambda: ops.gemm(x, w, M=M, N=N, K=K, bias=bias, resid=res, ldr=N, out_f32=y32, ws=ws)
    else:
        fwd = lambda: ops.gemm(x, w, M=M, N=N, K=K, bias=bias, out=y)
    kw = dict(act=ACT_DGELU, aux=aux, ldaux=K) if name == "fc2" else {}
    dgrad = lambda: ops.gemm(dy, w, M=M, N=K, K=N, a_layout=KC, b_layout=RC, lda=N, ldb=K, out=dx, ws=ws, **kw)
    dgrad_kc = lambda: ops.gemm(dy, wt, M=M, N=K, K=N, a_layout=KC, b_layout=KC, lda=N, ldb=N, out=dx, ws=ws, **kw)
    fl = 2.0 * M * N * K
    # the 144-row image stepping by 130 rows x 256 columns (8320 = 64 x 130: whole rounds of 256 tiles, no row tail): explicit tile code
    T130 = 13144256
    if epi == "gelu":
        fwd130 = lambda: ops.gemm(x, w, M=M, N=N, K=K, bias=bias, act=ACT_GELU, out=y, out2=y2, tile=T130)
    elif epi == "resid":
        fwd130 = lambda: ops.gemm(x, w, M=M, N=N, K=K, bias=bias, resid=res, ldr=N, out_f32=y32, tile=T130)
    else:
        fwd130 = lambda: ops.gemm(x, w, M=M, N=N, K=K, bias=bias, out=y, tile=T130)
    dgrad130 = lambda: ops.gemm(dy, w, M=M, N=K, K=N, a_layout=KC, b_layout=RC, lda=N, ldb=K, out=dx, tile=T130, **kw)
    # numerics of the new tile against fp32 matmuls of the same bf16 operands
    fwd130()
    ref = x.float() @ w.float().t() + bias
    got = (y32 - res) if epi == "resid" else (y2.float() if epi == "gelu" else y.float())
    e_f = float((got - ref).norm() / ref.norm())
    dgrad130()
    refd = dy.float() @ w.float()
    if name == "fc2":
        z = aux.float()
        refd = refd * (0.5 * (1 + torch.erf(z / 2 ** 0.5)) + z * torch.exp(-0.5 * z * z) / (2 * 3.141592653589793) ** 0.5)
    e_d = float((dx.float() - refd).norm() / refd.norm())
    print(f"      tile 130x256 rel-L2 error: fwd {e_f:.2e} dgrad {e_d:.2e}")
    t = dict(fwd=timeit(fwd), fwd130=timeit(fwd130), dgrad=timeit(dgrad), dgrad130=timeit(dgrad130), dgrad_kcB=timeit(dgrad_kc),
             blas_fwd=timeit(lambda: torch.matmul(x, w.t(), out=y)), blas_dgrad=timeit(lambda: torch.matmul(dy, w, out=dx)),
             blas_wgrad=timeit(lambda: torch.matmul(dy.t(), x, out=dw)))
    print(f"{name:5s} [{M} x {N} x {K}] " + " | ".join(f"{k} {v:6.1f} us {fl / v / 1e6:5.0f} TF" for k, v in t.items()), flush=True)
