"""One process per GPU over RCCL/xGMI (torch.distributed backend "nccl" is RCCL on ROCm).

Replaces the reference's single-process ``nn.DataParallel`` (utils/mim_vit.py:117): weights are
replicated, each rank trains on its own shard of the minibatch stream, gradients are averaged with
an all-reduce of the engine's flat gradient buffer (bucketed contiguous slices), AdamW runs
replicated.  With equal shard sizes this equals DataParallel's mean of per-replica losses
(SURVEY.md §8e).  The host logic here is backend-agnostic so it is covered by gloo tests on CPU.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def local_device_index(local_rank: int) -> int:
    """The GPU of this rank: its LOCAL_RANK, wrapped onto the devices that exist -- two gloo ranks rehearsing the N > 1 path
    on a one-GPU box both get cuda:0 (an RCCL world needs one device per rank; ``init_from_env`` refuses it otherwise)."""
    n = torch.cuda.device_count()
    return local_rank % n if n > 0 else 0


def init_from_env(backend: str | None = None):
    """-> (rank, world, local_rank).  No-op single process when WORLD_SIZE is unset / 1.

    Backend: the argument, else ``SKYEMB_DIST_BACKEND`` (``nccl`` | ``gloo``), else RCCL ("nccl") when a GPU is visible and
    gloo otherwise.  gloo with CUDA tensors is the rehearsal mode (collectives staged through the host): same schedule, same
    bucketing, any number of ranks per GPU."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = os.environ.get("SKYEMB_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend not in ("nccl", "gloo"):
            raise ValueError(f"SKYEMB_DIST_BACKEND / backend must be 'nccl' or 'gloo', not {backend!r}")
        if backend == "nccl":
            if local >= torch.cuda.device_count():
                raise RuntimeError(f"RCCL needs one GPU per rank: LOCAL_RANK {local} but {torch.cuda.device_count()} device(s) visible "
                                   f"(SKYEMB_DIST_BACKEND=gloo rehearses several ranks on one GPU)")
            torch.cuda.set_device(local)
            dist.init_process_group(backend, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    return rank, world, local


_HOST_BARRIERS = 0


def host_barrier(timeout_s: float = 6 * 3600.0, poll_s: float = 0.05):
    """Barrier through the rendezvous store instead of a collective: ranks wait on the HOST, so a long rank-0-only
    phase (the scikit-learn linear probe between training steps) cannot run into the RCCL watchdog that a pending
    all-reduce or NCCL barrier would trip after its 10-minute default.  No-op in a single process."""
    global _HOST_BARRIERS
    if not dist.is_initialized() or dist.get_world_size() <= 1:
        return
    import time
    store = dist.distributed_c10d._get_default_store()
    key = f"skyemb/host_barrier/{_HOST_BARRIERS}"
    _HOST_BARRIERS += 1
    world = dist.get_world_size()
    arrived = store.add(key, 1)
    t0 = time.time()
    while arrived < world:
        if time.time() - t0 > timeout_s:
            raise RuntimeError(f"host_barrier: {arrived} of {world} ranks arrived within {timeout_s:.0f} s")
        time.sleep(poll_s)
        arrived = store.add(key, 0)


_AGREEMENTS = 0


def agree(flag: bool) -> bool:
    """Rank 0's `flag` on every rank, through the rendezvous store (host side, no collective, no device sync): decisions that depend
    on a rank's wall clock -- the time-based checkpoint -- must be taken together once a collective hangs on them (the sharded
    optimiser's state gather).  Every rank must call it the same number of times.  Single process: the flag itself."""
    global _AGREEMENTS
    if not dist.is_initialized() or dist.get_world_size() <= 1:
        return bool(flag)
    store = dist.distributed_c10d._get_default_store()
    key = f"skyemb/agree/{_AGREEMENTS}"
    _AGREEMENTS += 1
    if dist.get_rank() == 0:
        store.set(key, b"1" if flag else b"0")
        return bool(flag)
    return store.get(key) == b"1"


def bucket_bounds(n: int, bucket_elems: int):
    """Contiguous [start, end) slices of a flat buffer, each a multiple of 8 elements except the last."""
    bucket_elems = max(8, bucket_elems // 8 * 8)
    return [(s, min(n, s + bucket_elems)) for s in range(0, n, bucket_elems)]


def allreduce_flat_gradients(g: torch.Tensor, world: int, bucket_elems: int = 16 * 1024 * 1024, group=None,
                             async_op: bool = False):
    """Sum-all-reduce the flat gradient buffer in buckets (64 MB fp32 by default: xGMI is
    point-to-point, large buckets amortise the per-collective latency).  Averaging (1/world) is
    folded into the AdamW kernel's ``grad_scale``.  Returns the work handles when async."""
    if world <= 1:
        return []
    works = []
    for s, e in bucket_bounds(g.numel(), bucket_elems):
        w = dist.all_reduce(g[s:e], op=dist.ReduceOp.SUM, group=group, async_op=async_op)
        if async_op:
            works.append(w)
    return works


def shard_chunk(length: int, world: int) -> int:
    """Elements per rank of a flat range cut into `world` equal owner chunks (multiples of 8: the kernels' piece size); the
    < 8 * world elements behind world * chunk are the range's TAIL, which stays replicated."""
    return (length // world) // 8 * 8


def reduce_scatter_range(g: torch.Tensor, s: int, e: int, rank: int, world: int, own: torch.Tensor, group=None):
    """Sum over the ranks of g[s:e], every rank receiving only ITS chunk: own[:c] <- sum_r g_r[s + rank c : s + (rank + 1) c]
    (c = shard_chunk(e - s, world)); the tail g[s + world c : e] is all-reduced in place.  Same bytes over xGMI as the all-reduce
    it replaces (a ring all-reduce IS this reduce-scatter followed by an all-gather).  -> async work handles.
    gloo (the rehearsal backend; several ranks on one GPU, collectives staged through the host) has no reduce-scatter for device
    tensors: there the whole range is all-reduced and `finish()` of the returned handle copies this rank's chunk out."""
    c = shard_chunk(e - s, world)
    works = []
    if dist.get_backend(group) == "gloo":
        w = dist.all_reduce(g[s:e], op=dist.ReduceOp.SUM, group=group, async_op=True)
        works.append(_Then(w, (lambda: own[:c].copy_(g[s + rank * c:s + (rank + 1) * c])) if c > 0 else None))
        return works
    if c > 0:
        works.append(dist.reduce_scatter_tensor(own[:c], g[s:s + world * c], op=dist.ReduceOp.SUM, group=group, async_op=True))
    if s + world * c < e:
        works.append(dist.all_reduce(g[s + world * c:e], op=dist.ReduceOp.SUM, group=group, async_op=True))
    return works


class _Then:
    """An async work handle with a follow-up on the waiting stream (wait() keeps torch.distributed's meaning)."""

    def __init__(self, work, then=None):
        self.work, self.then = work, then

    def wait(self):
        self.work.wait()
        if self.then is not None:
            self.then()
            self.then = None


def all_gather_range(buf: torch.Tensor, s: int, e: int, rank: int, world: int, group=None):
    """buf[s : s + world c] <- every rank's chunk buf[s + r c : s + (r + 1) c] (in place: each rank's input is its own slot of the
    output, as NCCL / RCCL define the in-place all-gather).  -> async work handle, or None for an empty chunk."""
    c = shard_chunk(e - s, world)
    if c == 0:
        return None
    if dist.get_backend(group) == "gloo":
        outs = [buf[s + r * c:s + (r + 1) * c] for r in range(world)]
        return dist.all_gather(outs, buf[s + rank * c:s + (rank + 1) * c].clone(), group=group, async_op=True)
    return dist.all_gather_into_tensor(buf[s:s + world * c], buf[s + rank * c:s + (rank + 1) * c], group=group, async_op=True)


def shard_rows(n_rows: int, rank: int, world: int):
    """Contiguous row shard [lo, hi) of a bank / dataset for this rank (last shards may be shorter)."""
    per = (n_rows + world - 1) // world
    lo = min(n_rows, rank * per)
    return lo, min(n_rows, lo + per)


def gather_topk(scores: torch.Tensor, idx: torch.Tensor, world: int, group=None):
    """All-gather per-rank [Q,k] results into [Q, world, k] lists ready for the k-way merge."""
    if world <= 1:
        return scores.unsqueeze(1), idx.unsqueeze(1)
    return _gather_lists(scores, idx, world, group)


def _gather_lists(scores, idx, world, group=None):
    """The collective part of :func:`gather_topk` (also called with world = 1 by the one-rank RCCL test)."""
    Q, k = scores.shape
    gs = torch.empty(world, Q, k, dtype=scores.dtype, device=scores.device)
    gi = torch.empty(world, Q, k, dtype=idx.dtype, device=idx.device)
    if dist.get_backend(group) == "gloo":
        ls = [gs[r] for r in range(world)]
        li = [gi[r] for r in range(world)]
        dist.all_gather(ls, scores.contiguous(), group=group)
        dist.all_gather(li, idx.contiguous(), group=group)
    else:
        dist.all_gather_into_tensor(gs, scores.contiguous(), group=group)
        dist.all_gather_into_tensor(gi, idx.contiguous(), group=group)
    return gs.permute(1, 0, 2).contiguous(), gi.permute(1, 0, 2).contiguous()


class DistributedIndexSampler(torch.utils.data.Sampler):
    """Disjoint per-rank index shards of a dataset, reshuffled per epoch with a shared seed."""

    def __init__(self, n: int, rank: int, world: int, shuffle: bool = True, seed: int = 0):
        self.n, self.rank, self.world, self.shuffle, self.seed, self.epoch = n, rank, world, shuffle, seed, 0
        self.per = n // world  # drop the remainder so every rank sees equally many samples

    def set_epoch(self, epoch: int):
        self.epoch = epoch

    def __len__(self):
        return self.per

    def __iter__(self):
        if self.shuffle:
            g = torch.Generator().manual_seed(self.seed + self.epoch)
            order = torch.randperm(self.n, generator=g)
        else:
            order = torch.arange(self.n)
        return iter(order[self.rank * self.per:(self.rank + 1) * self.per].tolist())
