#!/usr/bin/env python3
"""Times the pieces of search.cosine_topk for Q=16 over a 1M x 768 bank (HIP events)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sky_embeddings_amd import ops, search
N, D, k, Q = 1_000_000, 768, 100, int(sys.argv[1]) if len(sys.argv) > 1 else 16
bank = torch.randn(N, D, device="cuda")
q = torch.randn(Q, D, device="cuda")
w = torch.rand(D, device="cuda") + 0.5
pb = search.PreparedBank(bank, w)
def ev(f, n=10):
    f(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); s.record()
    for _ in range(n): f()
    e.record(); e.synchronize()
    return s.elapsed_time(e) / n, (time.perf_counter() - t0) / n * 1e3
tw, qn = search.prepare_queries(q, pb.weights)
print("prepare_queries  gpu %.3f ms  wall %.3f ms" % ev(lambda: search.prepare_queries(q, pb.weights)))
print("pruning_floor    gpu %.3f ms  wall %.3f ms" % ev(lambda: search.pruning_floor(tw, qn, pb, k, 1e-6)))
thr0 = search.pruning_floor(tw, qn, pb, k, 1e-6)
nch = ops.cosine_topk_chunks(N, Q, D, k)
ps = torch.empty(Q, nch, k, device="cuda"); pi = torch.empty(Q, nch, k, device="cuda", dtype=torch.int64)
print("nlists", nch)
print("main kernel      gpu %.3f ms  wall %.3f ms" % ev(lambda: ops.cosine_topk(tw, qn, bank, pb.norms, k, 1e-6, 0, nch, ps, pi, thr0)))
os_, oi = torch.empty(Q, k, device="cuda"), torch.empty(Q, k, device="cuda", dtype=torch.int64)
wsi = torch.empty(Q, device="cuda", dtype=torch.int32)
print("merge tournament gpu %.3f ms  wall %.3f ms" % ev(lambda: ops.topk_merge(ps, pi, Q, nch, k, os_, oi)))
print("merge gather+sort gpu %.3f ms wall %.3f ms" % ev(lambda: ops.topk_merge(ps, pi, Q, nch, k, os_, oi, wsi)))
print("cosine_topk all  gpu %.3f ms  wall %.3f ms" % ev(lambda: search.cosine_topk(q, pb, k)))
print("  without prune  gpu %.3f ms  wall %.3f ms" % ev(lambda: search.cosine_topk(q, pb, k, prune=False)))
