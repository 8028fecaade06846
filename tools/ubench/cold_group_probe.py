#!/usr/bin/env python3
"""The grouped weight-gradient launch of a ViT-B encoder / decoder block (four problems, no optimiser): us per launch with the same
operands every launch (cached) / with the SAVED ACTIVATIONS (the x operands: written a forward pass earlier in the step) rotated through
> 256 MB (HBM), the gradients (dy: written by the launches just before) cached either way."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sky_embeddings_amd import ops                                     # noqa: E402
from sky_embeddings_amd.ops import RC                                  # noqa: E402
from cold_weights_probe import graph_time                              # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    lp = dict(device=dev, dtype=torch.bfloat16)
    for name, M, D, H in (("encoder", 1280, 768, 3072), ("decoder", 4352, 512, 2048)):
        layers = [(D, H), (H, D), (D, D), (3 * D, D)]                   # (n_out, k_in): fc2, fc1, proj, qkv
        per_set = sum(M * k for _, k in layers) * 2
        nset = int(320e6 // per_set) + 1
        dys = [torch.randn(M, n, **lp) * 0.1 for n, _ in layers]
        dws = [torch.empty(n, k, device=dev) for n, k in layers]
        dbs = [torch.empty(n, device=dev) for n, _ in layers]
        xsets = [[torch.randn(M, k, **lp) * 0.1 for _, k in layers] for _ in range(nset)]

        def group(xs, hint=None):
            args = [ops.gemm_args(dy, x, M=n, N=k, K=M, a_layout=RC, b_layout=RC, lda=n, ldb=k, out_f32=dw, colsum_a=db,
                                  prefetch=hint if j == 0 else None)
                    for j, (dy, x, dw, db, (n, k)) in enumerate(zip(dys, xs, dws, dbs, layers))]
            grp = ops.GemmGroup(args, dev)
            assert grp.ok
            return grp
        warm = [group(xsets[0]) for _ in range(nset)]
        cold = [group(xsets[i]) for i in range(nset)]
        tw = graph_time([g.launch for g in warm])
        tc = graph_time([g.launch for g in cold])
        print(f"{name}: {per_set / 1e6:.1f} MB of saved activations per block; grouped launch {tw:.2f} us cached, {tc:.2f} us from HBM", flush=True)


if __name__ == "__main__":
    main()
