#!/usr/bin/env python3
"""Cost of one dependent launch in the step's setting: a chain of N tiny kernels (skyemb_cast of 8 elements; 1 workgroup) and of
N medium ones (cast of 1 M elements), eager on a stream and replayed from a HIP graph captured through torch."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sky_embeddings_amd import ops
dev = "cuda"
src = torch.randn(1 << 20, device=dev)
dst = torch.empty(1 << 20, device=dev, dtype=torch.bfloat16)
N = 400
def chain(n_el):
    for _ in range(N):
        ops.cast(src, dst, n_el)
def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps / N * 1e3
for n_el in (8, 1 << 20):
    eager = timed(lambda: chain(n_el), reps=5)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        chain(n_el)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        chain(n_el)
    graph = timed(g.replay)
    print(f"cast of {n_el} elements: eager {eager:.2f} us per launch, graph replay {graph:.2f} us per launch")
