#!/usr/bin/env python3
"""Yardstick only (not a product path): the GEMM shapes of one mim_19 block (ViT-L/16, 8320 token rows) beside the vendor BLAS
(torch.matmul -> hipBLASLt), each 20x in one HIP graph, us per launch.

Round 6: like for like.  The step's launches carry their layer's elementwise work in the epilogue -- bias, exact-erf GELU with the
pre-activation as a second output, the fp32 residual stream (read + written), the dGELU factor -- which a plain BLAS GEMM does
not do; rounds 4-5 compared the two anyway.  Columns per shape and direction:
  step      this library as the engine launches it (tuned tile, fused epilogue)
  plain     the same kernel with a bare epilogue (16-bit output only): the k-loops, like for like with ...
  blas      ... hipBLASLt's plain GEMM
  blas+ew   hipBLASLt + the elementwise torch kernels that produce what `step` produces (bias / GELU + pre-activation / residual add
            in fp32 / dGELU multiply): what the layer costs without the fusion
  t256      (with a third argument "t256") the step's launch forced onto the 256 x 256 persistent kernel, whatever the plan picks
usage: vitl_yardstick.py [bf16|f16] [t256]  (one MI355X)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sky_embeddings_amd import ops
from sky_embeddings_amd.ops import ACT_DGELU, ACT_GELU, KC, RC
M, D = 8320, 1024
dev = "cuda"
T = torch.float16 if (len(sys.argv) > 1 and sys.argv[1] == "f16") else torch.bfloat16
T256 = len(sys.argv) > 2 and sys.argv[2] == "t256"


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
# (round 6: half a second of matrix work before the first timing -- the first group measured after start-up read 10-15 % slow:
# qkv fwd 77.4 us first, 65.7 us as the last of its group, same launch)
_wa, _wb = torch.randn(4096, 4096, device=dev).to(T), torch.randn(4096, 4096, device=dev).to(T)
for _ in range(400):
    torch.matmul(_wa, _wb)
torch.cuda.synchronize()
print(f"# operand format {T}; us per launch, 20 launches per HIP graph")
tot = dict(step=0.0, plain=0.0, blas=0.0, blas_ew=0.0)
for name, N, K, epi in (("qkv", 3 * D, D, "bias"), ("proj", D, D, "resid"), ("fc1", 4 * D, D, "gelu"), ("fc2", D, 4 * D, "resid")):
    x = torch.randn(M, K, device=dev).to(T)
    w = (torch.randn(N, K, device=dev) * 0.05).to(T)
    dy = torch.randn(M, N, device=dev).to(T)
    bias = torch.zeros(N, device=dev)
    y, y2 = torch.empty(M, N, device=dev, dtype=T), torch.empty(M, N, device=dev, dtype=T)
    y32, res = torch.empty(M, N, device=dev), torch.randn(M, N, device=dev)
    dx, aux = torch.empty(M, K, device=dev, dtype=T), torch.randn(M, K, device=dev).to(T)
    if epi == "gelu":
        fwd = lambda: ops.gemm(x, w, M=M, N=N, K=K, bias=bias, act=ACT_GELU, out=y, out2=y2)
        fwd256 = lambda: ops.gemm(x, w, M=M, N=N, K=K, bias=bias, act=ACT_GELU, out=y, out2=y2, tile=256256)

        def blas_ew():
            torch.matmul(x, w.t(), out=y2)
            y2.add_(bias.to(T))                                  # pre-activation (second output)
            torch.nn.functional.gelu(y2, approximate="none")     # activation (allocating: torch has no out= form)
    elif epi == "resid":
        fwd = lambda: ops.gemm(x, w, M=M, N=N, K=K, bias=bias, resid=res, ldr=N, out_f32=y32, ws=ws)
        fwd256 = lambda: ops.gemm(x, w, M=M, N=N, K=K, bias=bias, resid=res, ldr=N, out_f32=y32, ws=ws, tile=256256)

        def blas_ew():
            torch.matmul(x, w.t(), out=y)
            torch.add(res, y, out=y32)                           # fp32 residual stream: read + write (bias folded in would be a third op)
    else:
        fwd = lambda: ops.gemm(x, w, M=M, N=N, K=K, bias=bias, out=y)
        fwd256 = lambda: ops.gemm(x, w, M=M, N=N, K=K, bias=bias, out=y, tile=256256)

        def blas_ew():
            torch.matmul(x, w.t(), out=y)
            y.add_(bias.to(T))
    plain = lambda: ops.gemm(x, w, M=M, N=N, K=K, out=y)
    blas = lambda: torch.matmul(x, w.t(), out=y)
    kw = dict(act=ACT_DGELU, aux=aux, ldaux=K) if name == "fc2" else {}
    dgrad = lambda: ops.gemm(dy, w, M=M, N=K, K=N, a_layout=KC, b_layout=RC, lda=N, ldb=K, out=dx, ws=ws, **kw)
    dgrad256 = lambda: ops.gemm(dy, w, M=M, N=K, K=N, a_layout=KC, b_layout=RC, lda=N, ldb=K, out=dx, ws=ws, tile=256256, **kw)
    dplain = lambda: ops.gemm(dy, w, M=M, N=K, K=N, a_layout=KC, b_layout=RC, lda=N, ldb=K, out=dx, ws=ws)
    dblas = lambda: torch.matmul(dy, w, out=dx)
    if name == "fc2":
        def dblas_ew():
            torch.matmul(dy, w, out=dx)
            z = aux.float()
            dx.mul_((0.5 * (1 + torch.erf(z * 0.7071067811865476)) + z * torch.exp(-0.5 * z * z) * 0.3989422804014327).to(T))
    else:
        dblas_ew = dblas
    fl = 2.0 * M * N * K
    for tag, fs in (("fwd", (fwd, plain, blas, blas_ew, fwd256)), ("dgrad", (dgrad, dplain, dblas, dblas_ew, dgrad256))):
        t256 = timeit(fs[4]) if T256 else None
        fs = fs[:4]
        t = [timeit(f) for f in fs]
        for k_, v in zip(tot, t):
            tot[k_] += v
        print(f"{name:5s} {tag:5s} [{M} x {(N if tag == 'fwd' else K)} x {(K if tag == 'fwd' else N)}]  step {t[0]:6.1f} us {fl / t[0] / 1e6:5.0f} TF | plain {t[1]:6.1f} us "
              f"{fl / t[1] / 1e6:5.0f} TF | blas {t[2]:6.1f} us {fl / t[2] / 1e6:5.0f} TF | blas+ew {t[3]:6.1f} us" + (f" | t256 {t256:6.1f} us" if T256 else ""), flush=True)
print("sum of the eight launches (one block, forward + data gradients): " + " | ".join(f"{k} {v:6.1f} us" for k, v in tot.items()))
