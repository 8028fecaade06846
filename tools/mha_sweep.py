import os, sys
import torch
sys.path.insert(0, "/root/repo")
from sky_embeddings_amd import ops
T = torch.float16
def timeit(f):
    for _ in range(3): f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20): f()
    g.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): g.replay()
    e.record(); e.synchronize()
    return s.elapsed_time(e) / 200 * 1e3
H, hd, N = 16, 64, 65
for B in (8, 16, 32, 48, 64, 96, 128, 256):
    D = H * hd
    qkv = torch.randn(B, N, 3 * D, device="cuda").to(T)
    dout = torch.randn(B, N, D, device="cuda").to(T)
    out, dqkv = torch.empty_like(dout), torch.empty_like(qkv)
    print(f"B {B:4d} WGs {B*H:5d} ({B*H/256:.2f}/CU): fwd {timeit(lambda: ops.mha_fwd(qkv, out, B, N, H, hd)):6.2f} us  bwd {timeit(lambda: ops.mha_bwd(qkv, dout, dqkv, B, N, H, hd)):6.2f} us", flush=True)
