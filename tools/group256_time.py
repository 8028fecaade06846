#!/usr/bin/env python3
"""The four weight gradients of a ViT-L block (mim_19: 8320 token rows) as one grouped launch: time per launch by tile code,
with and without the optimiser step.  usage: python tools/group256_time.py [tile ...]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sky_embeddings_amd import ops
from sky_embeddings_amd._lib import RC, AdamwDesc
DEV = torch.device("cuda")
T, dim, hid = 8320, 1024, 4096
shapes = [(dim, hid), (hid, dim), (dim, dim), (3 * dim, dim)]
g = torch.Generator(device="cuda").manual_seed(1)
sets = []
for r in range(4):      # operand sets of four different blocks: cold in L2
    dys = [torch.randn(T, o, device=DEV, generator=g).bfloat16() for o, _ in shapes]
    xs = [torch.randn(T, i, device=DEV, generator=g).bfloat16() for _, i in shapes]
    sets.append((dys, xs))
sizes = [o * i for o, i in shapes]
n = sum(sizes)
flat = torch.zeros(4 * n, device=DEV)
p, m, v = torch.randn(4 * n, device=DEV), torch.zeros(4 * n, device=DEV), torch.zeros(4 * n, device=DEV)
plp = torch.empty(4 * n, device=DEV, dtype=torch.bfloat16)
hyper = torch.tensor([1e-3, 0.1, 0.05, 0.0], device=DEV)
d = AdamwDesc()
d.g_base, d.p, d.m, d.v, d.p_lp, d.hyper = (t.data_ptr() for t in (flat, p, m, v, plp, hyper))
d.n_decay, d.beta1, d.beta2, d.eps, d.weight_decay, d.grad_scale = 4 * n, 0.9, 0.95, 1e-8, 0.05, 1.0
dbs = [torch.zeros(o, device=DEV) for o, _ in shapes]
for tile in [int(a) for a in sys.argv[1:]] or [128128, 256256]:
    for adam in (False, True):
        groups = []
        for r, (dys, xs) in enumerate(sets):
            off, args = r * n, []
            for j, (o, i) in enumerate(shapes):
                args.append(ops.gemm_args(dys[j], xs[j], M=o, N=i, K=T, a_layout=RC, b_layout=RC, lda=o, ldb=i,
                                          out_f32=flat[off:off + sizes[j]].view(o, i), colsum_a=dbs[j]))
                off += sizes[j]
            groups.append(ops.GemmGroup(args, DEV, tile=tile, adamw=d if adam else None))
        assert all(x.ok for x in groups)
        for x in groups:
            x.launch()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for it in range(20):
            groups[it % 4].launch()
        e1.record()
        torch.cuda.synchronize()
        print("tile %d adamw %d blocks %d: %.1f us per launch" % (tile, adam, groups[0].total_blocks, e0.elapsed_time(e1) * 50), flush=True)
