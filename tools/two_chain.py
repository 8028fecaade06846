#!/usr/bin/env python3
"""Would two half-batch chains on two streams beat one full-batch chain?  The forward GEMMs of the blocks (qkv, proj, fc1, fc2) and
their data gradients, captured once as ONE chain over all token rows and once as TWO concurrent chains over half the rows each
(fork / join inside the graph).  Measurement only; nothing in the product depends on it."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sky_embeddings_amd import ops  # noqa: E402

dev = "cuda"
T = torch.bfloat16


def make_chain(M, D, depth):
    """-> list of launch closures for `depth` blocks on M token rows"""
    fs = []
    x = torch.randn(M, D, device=dev).to(T)
    for _ in range(depth):
        wq = (torch.randn(3 * D, D, device=dev) * 0.05).to(T)
        wp = (torch.randn(D, D, device=dev) * 0.05).to(T)
        w1 = (torch.randn(4 * D, D, device=dev) * 0.05).to(T)
        w2 = (torch.randn(D, 4 * D, device=dev) * 0.05).to(T)
        bq, bp, b1, b2 = (torch.zeros(n, device=dev) for n in (3 * D, D, 4 * D, D))
        qkv = torch.empty(M, 3 * D, device=dev, dtype=T)
        att = torch.randn(M, D, device=dev).to(T)
        r1, r2 = torch.randn(M, D, device=dev), torch.randn(M, D, device=dev)
        y1, y2 = torch.empty(M, D, device=dev), torch.empty(M, D, device=dev)
        h, hg = torch.empty(M, 4 * D, device=dev, dtype=T), torch.empty(M, 4 * D, device=dev, dtype=T)
        dx = torch.empty(M, D, device=dev, dtype=T)
        dh = torch.empty(M, 4 * D, device=dev, dtype=T)
        dy = torch.randn(M, D, device=dev).to(T)
        dqkv = torch.randn(M, 3 * D, device=dev).to(T)
        ws = torch.zeros(4 * 1024 * 1024, device=dev)
        fs += [lambda x=x, wq=wq, bq=bq, qkv=qkv: ops.gemm(x, wq, M=M, N=3 * D, K=D, bias=bq, out=qkv),
               lambda att=att, wp=wp, bp=bp, r1=r1, y1=y1, ws=ws: ops.gemm(att, wp, M=M, N=D, K=D, bias=bp, resid=r1, ldr=D, out_f32=y1, ws=ws),
               lambda x=x, w1=w1, b1=b1, h=h, hg=hg: ops.gemm(x, w1, M=M, N=4 * D, K=D, bias=b1, act=ops.ACT_GELU, out=h, out2=hg),
               lambda hg=hg, w2=w2, b2=b2, r2=r2, y2=y2, ws=ws: ops.gemm(hg, w2, M=M, N=D, K=4 * D, bias=b2, resid=r2, ldr=D, out_f32=y2, ws=ws),
               # data gradients: fc2 (+dGELU), fc1, proj, qkv
               lambda dy=dy, w2=w2, h=h, dh=dh, ws=ws: ops.gemm(dy, w2, M=M, N=4 * D, K=D, a_layout=ops.KC, b_layout=ops.RC, lda=D, ldb=4 * D,
                                                                 act=ops.ACT_DGELU, aux=h, ldaux=4 * D, out=dh, ws=ws),
               lambda dh=dh, w1=w1, dx=dx, ws=ws: ops.gemm(dh, w1, M=M, N=D, K=4 * D, a_layout=ops.KC, b_layout=ops.RC, lda=4 * D, ldb=D, out=dx, ws=ws),
               lambda dy=dy, wp=wp, dx=dx, ws=ws: ops.gemm(dy, wp, M=M, N=D, K=D, a_layout=ops.KC, b_layout=ops.RC, lda=D, ldb=D, out=dx, ws=ws),
               lambda dqkv=dqkv, wq=wq, dx=dx, ws=ws: ops.gemm(dqkv, wq, M=M, N=D, K=3 * D, a_layout=ops.KC, b_layout=ops.RC, lda=3 * D, ldb=D, out=dx, ws=ws)]
    return fs


def time_graph(chains):
    """chains: list of launch lists; chain 0 on the capture stream, the others on side streams between a fork and a join"""
    side = [torch.cuda.Stream() for _ in chains[1:]]
    def run():
        cur = torch.cuda.current_stream()
        for s in side:
            s.wait_stream(cur)
        for f in chains[0]:
            f()
        for s, ch in zip(side, chains[1:]):
            with torch.cuda.stream(s):
                for f in ch:
                    f()
        for s in side:
            cur.wait_stream(s)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for _ in range(2):
            run()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            run()
        g.replay(); torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(10):
                g.replay()
            e.record(); e.synchronize()
            best = min(best, s.elapsed_time(e) / 10)
    return best


for tag, M, D, depth in (("encoder", 1280, 768, 12), ("decoder", 4352, 512, 8)):
    one = time_graph([make_chain(M, D, depth)])
    two = time_graph([make_chain(M // 2, D, depth), make_chain(M // 2, D, depth)])
    both = time_graph([make_chain(M, D, depth), make_chain(M, D, depth)])
    print(f"{tag}: one chain over {M} rows {one:.3f} ms; two concurrent chains over {M // 2} rows {two:.3f} ms; "
          f"two concurrent chains over {M} rows each {both:.3f} ms (= {both / 2:.3f} per chain)", flush=True)
