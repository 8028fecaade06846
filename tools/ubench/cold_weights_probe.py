#!/usr/bin/env python3
"""Are the ViT-B GEMM launches slower inside the step than alone because their WEIGHTS come from HBM?
For the GEMM shapes of an encoder block at BASELINE configs[1] (1280 token rows), us per launch, 40 launches per HIP graph:
  warm      the same weight matrix every launch (memory-side cache hit),
  rotate60  60 different weight matrices in turn (282 MB of 4.7 MB matrices: more than the 256 MB memory-side cache -> HBM),
  rotate6   6 in turn (what tools/ubench/gemm_lab times: cold in the L2s, warm in the memory-side cache),
  rotate_all_prefetch_next   as rotate60, every launch carrying the prefetch hint for the next launch's weights.
Activations (the A operand) are the same buffer in all three."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sky_embeddings_amd import ops                                     # noqa: E402
from sky_embeddings_amd.ops import KC, RC                              # noqa: E402


def graph_time(fns, reps=5):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for f in fns[:3]:
            f()
        g = torch.cuda.CUDAGraph()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            for f in fns:
                f()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        g.replay()
        e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / len(fns))
    return best


def main():
    dev = torch.device("cuda", 0)
    M, D, H = 1280, 768, 3072
    lp = dict(device=dev, dtype=torch.bfloat16)
    # (name, rows of W, cols of W, forward?)   forward: out[M, n_out] = x[M, k_in] W[n_out, k_in]^T ; dgrad: dx[M, k_in] = dy[M, n_out] W[n_out, k_in]
    shapes = [("qkv fwd", 3 * D, D, True), ("proj fwd", D, D, True), ("fc1 fwd", H, D, True), ("fc2 fwd", D, H, True),
              ("fc2 dgrad", D, H, False), ("fc1 dgrad", H, D, False), ("proj dgrad", D, D, False), ("qkv dgrad", 3 * D, D, False)]
    out = {}
    n = 40
    for name, n_out, k_in, fwd in shapes:
        per = n_out * k_in * 2
        pool = max(60, int(300e6 // per))
        Ws = [torch.randn(n_out, k_in, **lp) * 0.02 for _ in range(pool)]
        x = torch.randn(M, k_in if fwd else n_out, **lp) * 0.1
        y = torch.empty(M, n_out if fwd else k_in, **lp)

        def launch(W, nxt=None):
            if fwd:
                return lambda: ops.gemm(x, W, M=M, N=n_out, K=k_in, out=y, prefetch=nxt)
            return lambda: ops.gemm(x, W, M=M, N=k_in, K=n_out, a_layout=KC, b_layout=RC, lda=n_out, ldb=k_in, out=y, prefetch=nxt)
        row = {"weight_MB": per / 1e6, "pool": pool,
               "warm_us": graph_time([launch(Ws[0]) for _ in range(n)]),
               "rotate6_us": graph_time([launch(Ws[i % 6]) for i in range(n)]),
               "rotate_all_us": graph_time([launch(Ws[i % pool]) for i in range(pool)]),
               # every launch touches the NEXT launch's weights (the prefetch hint of skyemb_gemm_args)
               "rotate_all_prefetch_next_us": graph_time([launch(Ws[i % pool], Ws[(i + 1) % pool]) for i in range(pool)]),
               # ... the weights of the launch after next (what a chain with a LayerNorm / attention launch in between would do)
               "rotate_all_prefetch_next2_us": graph_time([launch(Ws[i % pool], Ws[(i + 2) % pool]) for i in range(pool)]),
               "warm_with_hint_us": graph_time([launch(Ws[0], Ws[1]) for _ in range(n)])}
        out[name] = row
        print(name, json.dumps(row), flush=True)
        del Ws
        torch.cuda.empty_cache()
    path = os.environ.get("PROBE_OUT")
    if path:
        json.dump(out, open(path, "w"), indent=1)


if __name__ == "__main__":
    main()
