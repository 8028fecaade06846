#!/usr/bin/env python3
"""What makes a dependent launch cost 4.4 us inside the step when a chain of identical tiny kernels costs 1.5 us?  Graph-replayed
chains: (a) one tiny kernel repeated; (b) eight DIFFERENT tiny-work kernels in rotation (instruction fetch); (c) a tiny kernel
behind a kernel that dirties 8 MB (write-back at the boundary); (d) a tiny kernel reading what the predecessor wrote."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sky_embeddings_amd import ops
dev = "cuda"
N = 240
def gt(fn, reps=N):
    fn(); torch.cuda.synchronize()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s): fn()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g): fn()
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); g.replay(); b.record(); b.synchronize(); one = a.elapsed_time(b)
    a.record(); g.replay(); g.replay(); g.replay(); b.record(); b.synchronize()
    return (a.elapsed_time(b) - one) / 2
src = torch.randn(1 << 22, device=dev); dst = torch.empty(1 << 22, device=dev, dtype=torch.bfloat16)
x = torch.randn(64, 768, device=dev); gam = torch.ones(768, device=dev); bet = torch.zeros(768, device=dev)
y = torch.empty(64, 768, device=dev, dtype=torch.bfloat16); mean = torch.empty(64, device=dev); rstd = torch.empty(64, device=dev)
noise = torch.rand(8, 16, device=dev); idr = torch.empty(8, 16, device=dev, dtype=torch.int64); msk = torch.empty(8, 16, device=dev)
idk = torch.empty(8, 4, device=dev, dtype=torch.int32)
sc = torch.randn(1, 1024, device=dev); fl = torch.empty(1, device=dev)
qkv = torch.randn(8 * 5, 3 * 768, device=dev).bfloat16(); att = torch.empty(8 * 5, 768, device=dev, dtype=torch.bfloat16)
A = torch.randn(64, 128, device=dev).bfloat16(); Bw = torch.randn(64, 128, device=dev).bfloat16(); C = torch.empty(64, 64, device=dev)
tiny = [lambda: ops.cast(src, dst, 8),
        lambda: ops.layernorm_fwd(x, gam, bet, y, mean, rstd, 64, 768, 1e-6),
        lambda: ops.random_mask_from_noise(noise, 4, idr, msk, idk),
        lambda: ops.kth_largest_floor(sc, 10, fl),
        lambda: ops.mha_fwd(qkv, att, 8, 5, 12, 64),
        lambda: ops.gemm(A, Bw, M=64, N=64, K=128, out_f32=C),
        lambda: ops.set_scalars(fl, 1.0),
        lambda: ops.standardise(x, gam, gam, x)]
def chain_same():
    for _ in range(N): tiny[0]()
def chain_rot():
    for i in range(N): tiny[i % 8]()
def chain_dirty():
    for i in range(N // 2):
        ops.cast(src, dst, 1 << 22)        # writes 8 MB
        tiny[0]()
def chain_big_only():
    for i in range(N // 2): ops.cast(src, dst, 1 << 22)
t_same, t_rot, t_dirty, t_big = gt(chain_same), gt(chain_rot), gt(chain_dirty), gt(chain_big_only)
each = [gt(lambda f=f: [f() for _ in range(N)]) / N * 1e3 for f in tiny]
print(f"same tiny kernel x{N}: {t_same / N * 1e3:.2f} us per launch")
print(f"eight different tiny kernels in rotation: {t_rot / N * 1e3:.2f} us per launch (each of them repeated alone: {[round(e, 2) for e in each]} -> mean {sum(each) / 8:.2f})")
print(f"8 MB writer + tiny: {t_dirty / (N // 2) * 1e3:.2f} us per pair; writer alone {t_big / (N // 2) * 1e3:.2f} us -> the tiny kernel behind it costs {(t_dirty - t_big) / (N // 2) * 1e3:.2f} us")
