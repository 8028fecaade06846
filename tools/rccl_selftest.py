#!/usr/bin/env python3
"""One-rank RCCL self-test: process-group init on this GPU and the collectives the training step issues (bf16 / fp32
all-reduce of buffer slices, async handles, the top-k all-gather), under the same launcher the driver uses:
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 tools/rccl_selftest.py"""
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
dist.barrier()
dist.destroy_process_group()
print("rccl self-test ok: world", os.environ.get("WORLD_SIZE", "1"))
