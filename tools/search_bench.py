#!/usr/bin/env python3
"""Many-query search alone: 1M x 768 fp32 bank, Q queries, k = 100 (BASELINE configs[3] on one GPU).
usage: python tools/search_bench.py [Q] [iters] [N]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sky_embeddings_amd.search import PreparedBank, cosine_topk
Q = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
N, D, k = int(sys.argv[3]) if len(sys.argv) > 3 else 1_000_000, 768, 100
g = torch.Generator(device="cuda").manual_seed(2024)
bank = torch.empty(N, D, device="cuda")
for s in range(0, N, 50_000):
    bank[s:s + 50_000] = torch.randn(min(50_000, N - s), D, device="cuda", generator=g)
queries = torch.randn(Q, D, device="cuda", generator=torch.Generator(device="cuda").manual_seed(2025))
w = 1.0 / (torch.rand(D, device="cuda", generator=torch.Generator(device="cuda").manual_seed(7)) + 0.5) ** 2
pb = PreparedBank(bank, w / w.sum())
torch.cuda.synchronize(); t0 = time.perf_counter()
pb.half_image(); torch.cuda.synchronize()
print("fp16 image of the bank: %.1f ms (once per bank)" % ((time.perf_counter() - t0) * 1e3))
st = {}
cosine_topk(queries, pb, k, stats=st); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(iters):
    s, i = cosine_topk(queries, pb, k, stats=st)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / iters
print("Q=%d: %.2f ms per search = %.0f queries/s  (%s, %d re-run exactly)  checksum %d" % (Q, dt * 1e3, Q / dt, st["path"], st["redone"], int(i.sum().item() % (1 << 31))))
