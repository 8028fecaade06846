import os, sys, torch
sys.path.insert(0, '/root/repo')
from sky_embeddings_amd import ops
def gt(f, reps=400):
    f(); torch.cuda.synchronize()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s): f()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps): f()
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); g.replay(); b.record(); b.synchronize(); one = a.elapsed_time(b)
    a.record(); g.replay(); g.replay(); g.replay(); b.record(); b.synchronize()
    return (a.elapsed_time(b) - one) / (2 * reps) * 1e3
Q, D = 16, 768
q = torch.randn(Q, D, device="cuda"); w = torch.rand(D, device="cuda") + 0.5
tw, qn = torch.empty(Q, D, device="cuda"), torch.empty(Q, device="cuda")
sc = torch.randn(Q, 25600, device="cuda") * 0.036; floor = torch.empty(Q, device="cuda")
src = torch.randn(8, device="cuda"); dst = torch.empty(8, device="cuda", dtype=torch.bfloat16)
print("SKY_DBG", os.environ.get("SKY_DBG"), "cast(8) %.1f  wnorm %.1f  kth %.1f us" % (gt(lambda: ops.cast(src, dst, 8)), gt(lambda: ops.weighted_norms(q, w, qn, tw)), gt(lambda: ops.kth_largest_floor(sc, 100, floor))))
