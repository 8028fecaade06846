#!/usr/bin/env python3
"""Launch cost against the memory a chain of tiny kernels touches: the same 8-element cast, but every launch at a different offset of a
large buffer (stride bytes apart) -- address-translation misses show as a per-launch cost that grows with the footprint."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sky_embeddings_amd import ops
N = 240
def gt(fn):
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
    return (a.elapsed_time(b) - one) / 2 / N * 1e3
big = torch.zeros(1 << 30, device="cuda")          # 4 GB of fp32
dst = torch.empty(1 << 30, device="cuda", dtype=torch.bfloat16)
for stride_mb in (0, 1, 4, 16):
    step = stride_mb * (1 << 20) // 4
    def chain():
        for i in range(N):
            o = (i * step) % ((1 << 30) - 64)
            ops.cast(big[o:], dst[o:], 8)
    print(f"stride {stride_mb:3d} MB (footprint {stride_mb * N} MB): {gt(chain):.2f} us per launch", flush=True)
# many small separate allocations (what the step's workspaces are): 240 tensors of 8 MB
bufs = [(torch.zeros(1 << 21, device="cuda"), torch.empty(1 << 21, device="cuda", dtype=torch.bfloat16)) for _ in range(N)]
def chain2():
    for a, b in bufs: ops.cast(a, b, 8)
print(f"240 separate 8 MB + 4 MB tensors: {gt(chain2):.2f} us per launch")
def chain3():
    for a, b in bufs: ops.cast(a, b, 1 << 21)
print(f"  the same, casting all 2 M elements each (12 MB of traffic): {gt(chain3):.2f} us per launch")
