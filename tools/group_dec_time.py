#!/usr/bin/env python3
"""The four weight gradients of a decoder block of config A (4352 token rows, dim 512) or of an encoder block (1280 rows, dim 768) as
one grouped launch: time per launch by tile code.  usage: python tools/group_dec_time.py dec|enc [tile ...]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sky_embeddings_amd import ops
from sky_embeddings_amd._lib import RC, AdamwDesc
DEV = torch.device("cuda")
which = sys.argv[1] if len(sys.argv) > 1 else "dec"
T, dim = (4352, 512) if which == "dec" else (1280, 768)
hid = 4 * dim
shapes = [(dim, hid), (hid, dim), (dim, dim), (3 * dim, dim)]
g = torch.Generator(device="cuda").manual_seed(1)
NSET = 8
sets = []
for r in range(NSET):
    dys = [torch.randn(T, o, device=DEV, generator=g).bfloat16() for o, _ in shapes]
    xs = [torch.randn(T, i, device=DEV, generator=g).bfloat16() for _, i in shapes]
    sets.append((dys, xs))
sizes = [o * i for o, i in shapes]
n = sum(sizes)
flat = torch.zeros(NSET * n, device=DEV)
p, m, v = torch.randn(NSET * n, device=DEV), torch.zeros(NSET * n, device=DEV), torch.zeros(NSET * n, device=DEV)
plp = torch.empty(NSET * n, device=DEV, dtype=torch.bfloat16)
hyper = torch.tensor([1e-3, 0.1, 0.05, 0.0], device=DEV)
d = AdamwDesc()
d.g_base, d.p, d.m, d.v, d.p_lp, d.hyper = (t.data_ptr() for t in (flat, p, m, v, plp, hyper))
d.n_decay, d.beta1, d.beta2, d.eps, d.weight_decay, d.grad_scale = NSET * n, 0.9, 0.95, 1e-8, 0.05, 1.0
dbs = [torch.zeros(o, device=DEV) for o, _ in shapes]
for tile in [int(a) for a in sys.argv[2:]] or [64064, 128064, 128128, 9128128]:
    for adam in (False, True):
        groups = []
        for r, (dys, xs) in enumerate(sets):
            off, args = r * n, []
            for j, (o, i) in enumerate(shapes):
                args.append(ops.gemm_args(dys[j], xs[j], M=o, N=i, K=T, a_layout=RC, b_layout=RC, lda=o, ldb=i,
                                          out_f32=flat[off:off + sizes[j]].view(o, i), colsum_a=dbs[j]))
                off += sizes[j]
            groups.append(ops.GemmGroup(args, DEV, tile=tile, adamw=d if adam else None))
        if not all(x.ok for x in groups):
            print("tile %d adamw %d: not built" % (tile, adam), flush=True)
            continue
        for x in groups:
            x.launch()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for it in range(40):
            groups[it % NSET].launch()
        e1.record()
        torch.cuda.synchronize()
        print("%s tile %d adamw %d blocks %d: %.1f us per launch" % (which, tile, adam, groups[0].total_blocks, e0.elapsed_time(e1) * 25), flush=True)
