#!/usr/bin/env python3
"""What does the prefetch hint cost the launch that carries it?  Weights cached (the same matrix every launch); the hint names nothing /
a cached range / a different 4.7 MB range out of 300 MB every launch (lines that come from HBM, as in the step)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sky_embeddings_amd import ops                                     # noqa: E402
from sky_embeddings_amd.ops import KC, RC                              # noqa: E402
from cold_weights_probe import graph_time                              # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    M, D, H = 1280, 768, 3072
    lp = dict(device=dev, dtype=torch.bfloat16)
    pool = [torch.randn(H, D, **lp) * 0.02 for _ in range(64)]
    Dd, Hd = 512, 2048
    shapes = [("qkv fwd", M, 3 * D, D, True), ("proj fwd", M, D, D, True), ("fc1 fwd", M, H, D, True), ("fc2 fwd", M, D, H, True),
              ("fc2 dgrad", M, D, H, False), ("fc1 dgrad", M, H, D, False), ("proj dgrad", M, D, D, False), ("qkv dgrad", M, 3 * D, D, False),
              ("dec qkv fwd", 4352, 3 * Dd, Dd, True), ("dec proj fwd", 4352, Dd, Dd, True), ("dec fc1 fwd", 4352, Hd, Dd, True),
              ("dec fc2 fwd", 4352, Dd, Hd, True), ("dec fc2 dgrad", 4352, Dd, Hd, False), ("dec fc1 dgrad", 4352, Hd, Dd, False),
              ("dec proj dgrad", 4352, Dd, Dd, False), ("dec qkv dgrad", 4352, 3 * Dd, Dd, False)]
    for name, M, n_out, k_in, fwd in shapes:
        W = torch.randn(n_out, k_in, **lp) * 0.02
        x = torch.randn(M, k_in if fwd else n_out, **lp) * 0.1
        y = torch.empty(M, n_out if fwd else k_in, **lp)

        def launch(h):
            if fwd:
                return lambda: ops.gemm(x, W, M=M, N=n_out, K=k_in, out=y, prefetch=h)
            return lambda: ops.gemm(x, W, M=M, N=k_in, K=n_out, a_layout=KC, b_layout=RC, lda=n_out, ldb=k_in, out=y, prefetch=h)
        none = graph_time([launch(None) for _ in range(64)])
        warm = graph_time([launch(pool[0].view(-1)[:3 * Dd * Dd]) for _ in range(64)])
        nh = (3 * D * D) if M == 1280 else (3 * Dd * Dd)       # a typical next weight matrix: 3.5 MB (encoder) / 1.6 MB (decoder)
        cold = graph_time([launch(pool[i].view(-1)[:nh]) for i in range(64)])
        print(f"{name:15s} no hint {none:6.2f} us   hint -> cached range {warm:6.2f}   hint -> HBM range {cold:6.2f}   (+{cold - none:5.2f})", flush=True)


if __name__ == "__main__":
    main()
