#!/usr/bin/env python3
"""One-rank RCCL self-test: process-group init on this GPU and the collectives the training step issues (bf16 / fp32
all-reduce of buffer slices, async handles, the top-k all-gather), under the same launcher the driver uses:
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 tools/rccl_selftest.py

--sweep (any N ranks; needs only the launcher): times the gradient all-reduces of ONE data-parallel ViT-B step as TrainStep issues
them (the per-stage slices of the flat gradient buffer, bf16 and fp32: bench.py `extra.staged.bytes_per_stage`) and whole-buffer
reduces in buckets of 8 / 32 / 128 MB.  RCCL's algorithm and protocol are process-wide environment knobs: sweep them from the shell,
    for a in Ring Tree; do for p in Simple LL128; do NCCL_ALGO=$a NCCL_PROTO=$p python -m torch.distributed.run ... --sweep; done; done
One JSON line on rank 0 (algorithm bandwidth = bytes / time; bus bandwidth = x 2 (N - 1) / N)."""
import os, sys
import torch
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
local = int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(local)
dist.init_process_group("nccl", device_id=torch.device("cuda", local))
g = torch.ones(1 << 22, device="cuda", dtype=torch.bfloat16)
works = [dist.all_reduce(g[s:s + (1 << 20)], async_op=True) for s in range(0, 1 << 22, 1 << 20)]
for w in works:
    w.wait()
f = torch.ones(1 << 20, device="cuda")
dist.all_reduce(f)
from sky_embeddings_amd.distributed import gather_topk
s, i = gather_topk(torch.randn(4, 8, device="cuda"), torch.arange(32, device="cuda").reshape(4, 8), dist.get_world_size(), None)
torch.cuda.synchronize()
assert float(g.float().sum()) == float(1 << 22) * dist.get_world_size() and s.shape == (4, dist.get_world_size(), 8)
if "--sweep" in sys.argv:
    import json
    import time
    dev, rank, world = torch.device("cuda", local), dist.get_rank(), dist.get_world_size()
    stages_bf16 = [52428800, 28311552, 28311552, 28311552, 28311552, 28311552, 28311552, 2327552]     # bytes per stage, ViT-B (bench.py)
    total = sum(stages_bf16) // 2                                                                   # elements of the flat buffer

    def timed(fn, reps=10):
        fn(); torch.cuda.synchronize(); dist.barrier()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        dt = torch.tensor([(time.perf_counter() - t0) / reps], device=dev, dtype=torch.float64)
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
        return float(dt)
    out = {"world": world, "NCCL_ALGO": os.environ.get("NCCL_ALGO"), "NCCL_PROTO": os.environ.get("NCCL_PROTO"), "cases": []}
    for name, dtype, esz in (("bf16", torch.bfloat16, 2), ("f32", torch.float32, 4)):
        buf = torch.ones(total, device=dev, dtype=dtype)
        bounds, o = [], 0
        for b in stages_bf16:
            bounds.append((o, o + b // 2)); o += b // 2

        def per_stage():
            hs = [dist.all_reduce(buf[s:e], async_op=True) for s, e in bounds]
            for h in hs: h.wait()
        t = timed(per_stage)
        nbytes = total * esz
        out["cases"].append(dict(what=f"8 per-stage slices, {name}", bytes=nbytes, ms=t * 1e3, alg_gbs=nbytes / t / 1e9,
                                 bus_gbs=nbytes / t / 1e9 * 2 * (world - 1) / world))
        for mb in (8, 32, 128):
            n = mb * (1 << 20) // esz

            def bucketed():
                hs = [dist.all_reduce(buf[s:min(total, s + n)], async_op=True) for s in range(0, total, n)]
                for h in hs: h.wait()
            t = timed(bucketed)
            out["cases"].append(dict(what=f"whole buffer in {mb} MB buckets, {name}", bytes=nbytes, ms=t * 1e3, alg_gbs=nbytes / t / 1e9,
                                     bus_gbs=nbytes / t / 1e9 * 2 * (world - 1) / world))
        del buf
    if rank == 0:
        print(json.dumps(out))
dist.barrier()
dist.destroy_process_group()
print("rccl self-test ok: world", os.environ.get("WORLD_SIZE", "1"))
