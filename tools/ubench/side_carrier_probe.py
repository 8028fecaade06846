#!/usr/bin/env python3
"""Can a data-gradient launch of the ViT-B encoder's backward chain carry a slice of the AdamW step as side workgroups?
(Experiment of round 5; needs the library built with tools/ubench/side_carrier.patch -- `git apply` it, `make -C sky_embeddings_amd/csrc` --
which adds data-gradient carrier instances of the grouped launch; result: profiles/r05_side_carrier_probe.json)
For the four data gradients of an encoder block at BASELINE configs[1] (1280 token rows): us per launch, 20 launches per HIP graph,
 (a) the launch as the step issues it (skyemb_gemm),
 (b) the same problem as a one-problem grouped launch with NO side slice (what the blob costs),
 (c) with a slice of S parameters stepped by side workgroups (skyemb_gemm_group_plan_side_adamw, data-gradient carrier),
 and the AdamW kernel alone on the same slice.  Results are timings only (the optimiser state of a scratch engine is stepped)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sky_embeddings_amd import ops                                     # noqa: E402
from sky_embeddings_amd.engine import MAEEngine                        # noqa: E402
from sky_embeddings_amd.model_config import config_for                 # noqa: E402
from sky_embeddings_amd.ops import KC, RC                              # noqa: E402
from sky_embeddings_amd.optim import FusedAdamW                        # noqa: E402


def graph_time(fn, n=20, reps=5):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
        g = torch.cuda.CUDAGraph()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n):
                fn()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        g.replay()
        e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / n)
    return best


def main():
    dev = torch.device("cuda", 0)
    cfg = config_for("base", img_size=64, patch_size=16, in_chans=5, embed_dim=768, norm_pix_loss=True, loss_fn="mse")
    eng = MAEEngine(cfg, device=dev, compute_dtype=torch.bfloat16, seed=0)
    opt = FusedAdamW(eng, lr=1e-4, weight_decay=0.05)
    opt.use_device_scalars(dev)
    eng.enable_fused_adamw(opt, True)
    ad = eng._fused_adamw
    opt.begin_step()
    st = eng.store
    M, D, H = 1280, 768, 3072
    lp = dict(device=dev, dtype=torch.bfloat16)
    lo0 = st.offsets["blocks.3.attn.qkv.weight"]
    shapes = [("fc2 dgrad", D, H, 6128064), ("fc1 dgrad", H, D, 9064064), ("proj dgrad", D, D, 9064064), ("qkv dgrad", 3 * D, D, 9064064)]
    out = {}
    for name, n_out, k_in, tile in shapes:
        dy = torch.randn(M, n_out, **lp) * 0.1
        W = torch.randn(n_out, k_in, **lp) * 0.02
        dx = torch.empty(M, k_in, **lp)
        row = {"plain_us": graph_time(lambda: ops.gemm(dy, W, M=M, N=k_in, K=n_out, a_layout=KC, b_layout=RC, lda=n_out, ldb=k_in, out=dx))}
        for t in ([tile] if tile == 6128064 else [9064064, 10064064]):
            for S_mb, blocks in ((0, 0), (8, 64), (16, 128), (32, 128), (32, 256), (48, 256)):
                nparam = (S_mb * (1 << 20) // 26) // 8 * 8
                args = [ops.gemm_args(dy, W, M=M, N=k_in, K=n_out, a_layout=KC, b_layout=RC, lda=n_out, ldb=k_in, out=dx)]
                grp = ops.GemmGroup(args, dev, tile=t, adamw=ad, side=(False, lo0, lo0 + nparam, blocks))
                assert grp.ok, (name, t)
                row[f"tile{t}_side{S_mb}MB_{blocks}wg_us"] = graph_time(grp.launch)
        for S_mb in (8, 16, 32, 48):
            nparam = (S_mb * (1 << 20) // 26) // 8 * 8
            row[f"adamw_alone_{S_mb}MB_us"] = graph_time(lambda: opt.apply_range(lo0, lo0 + nparam))
        out[name] = row
        print(name, json.dumps(row), flush=True)
    path = os.environ.get("PROBE_OUT")
    if path:
        json.dump(out, open(path, "w"), indent=1)


if __name__ == "__main__":
    main()
