#!/usr/bin/env python3
"""GPU time of every launch of a small-Q search (Q = 16 / 1 over 1 M x 768, k = 100), each replayed 20x from a HIP graph
(no host gaps, no profiler): prepare_queries (wnorm), sample scorer, k-th floor, bank pass, merge."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sky_embeddings_amd import ops, search
N, D, k = 1_000_000, 768, 100
bank = torch.randn(N, D, device="cuda")
w = torch.rand(D, device="cuda") + 0.5
pb = search.PreparedBank(bank, w)
def gt(f, reps=400):
    f(); torch.cuda.synchronize()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        f()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps): f()
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); g.replay(); b.record(); b.synchronize()
    one = a.elapsed_time(b)
    a.record(); g.replay(); g.replay(); g.replay(); b.record(); b.synchronize()
    return (a.elapsed_time(b) - one) / (2 * reps) * 1e3          # (the difference removes the per-replay launch cost)
for Q in (16, 1):
    q = torch.randn(Q, D, device="cuda")
    tw, qn = torch.empty(Q, D, device="cuda"), torch.empty(Q, device="cuda")
    sb, sn = pb.sample(256 * k)
    sc = torch.empty(Q, sb.shape[0], device="cuda"); floor = torch.empty(Q, device="cuda")
    nch = ops.cosine_topk_chunks(N, Q, D, k)
    ps = torch.empty(Q, nch, k, device="cuda"); pi = torch.empty(Q, nch, k, device="cuda", dtype=torch.int64)
    os_, oi = torch.empty(Q, k, device="cuda"), torch.empty(Q, k, device="cuda", dtype=torch.int64)
    wsi = torch.empty(Q, device="cuda", dtype=torch.int32)
    t = {}
    t["wnorm(queries)"] = gt(lambda: ops.weighted_norms(q, pb.weights, qn, tw))
    t["sample scores"] = gt(lambda: ops.cosine_scores(tw, qn, sb, sn, 1e-6, sc))
    t["kth floor"] = gt(lambda: ops.kth_largest_floor(sc, k, floor))
    t["bank pass"] = gt(lambda: ops.cosine_topk(tw, qn, bank, pb.norms, k, 1e-6, 0, nch, ps, pi, floor), reps=10)
    wsf = torch.empty(Q * ((sb.shape[0] + 15) // 16), device="cuda")
    t["sample floor (2 launches)"] = gt(lambda: ops.cosine_sample_floor(tw, qn, sb, sn, k, 1e-6, wsf, floor))
    t["merge"] = gt(lambda: ops.topk_merge(ps, pi, Q, nch, k, os_, oi, wsi))
    def whole():
        ops.weighted_norms(q, pb.weights, qn, tw); ops.cosine_sample_floor(tw, qn, sb, sn, k, 1e-6, wsf, floor)
        ops.cosine_topk(tw, qn, bank, pb.norms, k, 1e-6, 0, nch, ps, pi, floor); ops.topk_merge(ps, pi, Q, nch, k, os_, oi, wsi)
    t["whole search (one graph)"] = gt(whole, reps=10)
    print(f"Q={Q}: " + "  ".join(f"{n} {v:.1f} us" for n, v in t.items()), flush=True)
