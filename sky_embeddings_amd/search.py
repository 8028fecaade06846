"""Batched weighted-cosine top-k over an embedding bank resident in HBM.

Build-level formulation of the reference search (utils/similarity.py: one mean target vector +
inverse-variance weights scored against streamed batches, best ``n_save`` kept by cat+argsort):
``cosine_topk(queries[Q,D], bank[N,D], k, weights)`` returns the exact top-k (score desc, index
asc) per query; Q = 1 with ``weights`` reproduces ``compute_similarity(metric='cosine')`` +
``update_best_scores``.  Multi-GPU: the bank is sharded by rows (one process per GPU), every rank
scores its shard, the per-rank [Q,k] results are all-gathered over RCCL and merged with the same
order, so the result is identical to the single-GPU one.
"""
from __future__ import annotations

import torch

from . import ops


class PreparedBank:
    """Bank rows + their weighted norms (recomputed only when the weights change)."""

    def __init__(self, bank: torch.Tensor, weights: torch.Tensor | None = None, idx_offset: int = 0):
        assert bank.is_cuda and bank.dtype == torch.float32 and bank.is_contiguous() and bank.dim() == 2
        self.bank, self.idx_offset = bank, int(idx_offset)
        self.norms = torch.empty(bank.shape[0], device=bank.device)
        self._sample = None
        self._half = None
        self.set_weights(weights)

    def set_weights(self, weights):
        self.weights = None if weights is None else weights.to(self.bank.device, torch.float32).contiguous()
        ops.weighted_norms(self.bank, self.weights, self.norms)
        self._sample = None
        self._half = None

    def half_image(self):
        """(bank16, rowp): the fp16 image + per-row constants the prefiltered many-query search runs on; built on first
        use (one pass over the bank, +50 % bank memory) and kept until the weights change."""
        if self._half is None:
            self._half = ops.bank16_prepare(self.bank, self.norms)
        return self._half

    def sample(self, rows: int):
        """A strided row sample (bank rows + norms) used to derive the pruning floor of a search."""
        N = self.bank.shape[0]
        rows = min(rows, N)
        if self._sample is None or self._sample[0].shape[0] != rows:
            idx = torch.arange(rows, device=self.bank.device) * (N // rows)
            self._sample = (self.bank.index_select(0, idx).contiguous(), self.norms.index_select(0, idx).contiguous())
        return self._sample


def standardise_(bank: torch.Tensor, mean: torch.Tensor, std: torch.Tensor, out: torch.Tensor | None = None):
    """(x - mean) / (std + 1e-8) row-wise (utils/similarity.py:101-102); in place by default."""
    out = bank if out is None else out
    ops.standardise(bank, mean.contiguous(), std.contiguous(), out)
    return out


def prepare_queries(queries: torch.Tensor, weights: torch.Tensor | None):
    Q, D = queries.shape
    tw = torch.empty(Q, D, device=queries.device)
    qn = torch.empty(Q, device=queries.device)
    ops.weighted_norms(queries.contiguous(), weights, qn, tw)
    return tw, qn


def _local_topk(tw, qn, bank, norms, k, eps, idx_offset, thr0=None):
    Q, D = tw.shape
    N = bank.shape[0]
    dev = tw.device
    nch = ops.cosine_topk_chunks(N, Q, D, k)
    ps = torch.empty(Q, nch, k, device=dev)
    pi = torch.empty(Q, nch, k, device=dev, dtype=torch.int64)
    ops.cosine_topk(tw, qn, bank, norms, k, eps, idx_offset, nch, ps, pi, thr0)
    out_s = torch.empty(Q, k, device=dev)
    out_i = torch.empty(Q, k, device=dev, dtype=torch.int64)
    ops.topk_merge(ps, pi, Q, nch, k, out_s, out_i, torch.empty(Q, device=dev, dtype=torch.int32))
    return out_s, out_i


def pruning_floor(tw, qn, pb: "PreparedBank", k: int, eps: float, sample_rows: int | None = None):
    """Per-query score floor for the main pass: the k-th best score over a row SAMPLE is a lower
    bound of the k-th best over the whole bank, so rows scoring below it can never enter the
    result.  Returned one ulp lower (the kernels keep rows STRICTLY above the floor, ties included
    this way).  None when the bank is too small for the extra pass to pay."""
    N = pb.bank.shape[0]
    if sample_rows is None:
        sample_rows = 256 * k
    if N < 8 * sample_rows:
        return None
    sb, sn = pb.sample(sample_rows)
    if ops.sample_floor_applicable(tw.shape[0], sb.shape[0], tw.shape[1], k, tw, sb):
        # Q <= 16: tile maxima of the sample scores + a one-wave selection (two short launches, no [Q, sample] matrix)
        floor = torch.empty(tw.shape[0], device=tw.device)
        ws = torch.empty(tw.shape[0] * ((sb.shape[0] + 15) // 16), device=tw.device)
        ops.cosine_sample_floor(tw, qn, sb, sn, k, eps, ws, floor)
        return floor
    sc = torch.empty(tw.shape[0], sb.shape[0], device=tw.device)
    ops.cosine_scores(tw, qn, sb, sn, eps, sc)          # [Q, sample] score matrix (small)
    floor = torch.empty(tw.shape[0], device=tw.device)
    ops.kth_largest_floor(sc, k, floor)                 # k-th best of the sample, one ulp lower (radix select, one launch)
    return floor


def _prefilter_enabled():
    import os
    return os.environ.get("SKYEMB_TOPK_PREFILTER", "1") != "0"


def _local_topk_prefiltered(tw, qn, pb: "PreparedBank", k, eps):
    """Two-stage exact top-k of this rank's shard (csrc/topk_prefilter.hip); queries the second stage could not certify
    (redo flags) go through the exact fp32 kernel."""
    Q = tw.shape[0]
    dev = tw.device
    bank16, rowp = pb.half_image()
    out_s = torch.empty(Q, k, device=dev)
    out_i = torch.empty(Q, k, device=dev, dtype=torch.int64)
    redo = torch.empty(Q, device=dev, dtype=torch.int32)
    ops.cosine_topk_prefiltered(tw, qn, pb.bank, pb.norms, bank16, rowp, k, eps, pb.idx_offset, out_s, out_i, redo)
    again = torch.nonzero(redo).squeeze(1)                 # host sync: a handful of bytes per search
    if again.numel():
        tw2, qn2 = tw.index_select(0, again).contiguous(), qn.index_select(0, again).contiguous()
        thr0 = pruning_floor(tw2, qn2, pb, k, eps)
        s2, i2 = _local_topk(tw2, qn2, pb.bank, pb.norms, k, eps, pb.idx_offset, thr0)
        out_s.index_copy_(0, again, s2)
        out_i.index_copy_(0, again, i2)
    return out_s, out_i, int(again.numel())


def cosine_topk(queries: torch.Tensor, bank, k: int, weights: torch.Tensor | None = None, eps: float = 1e-6,
                process_group=None, world_size: int = 1, prune: bool = True, stats: dict | None = None):
    """-> (scores f32 [Q,k], indices i64 [Q,k]).  ``bank`` is a [N,D] tensor or a PreparedBank
    (this rank's shard; ``idx_offset`` = first global row of the shard).  More than 16 queries take the two-stage
    path (fp16 matrix-core prefilter with a proven error bound, exact fp32 re-score of the survivors: same results bit
    for bit); SKYEMB_TOPK_PREFILTER=0 keeps every search on the exact fp32 kernels."""
    pb = bank if isinstance(bank, PreparedBank) else PreparedBank(bank, weights)
    q = queries.to(pb.bank.device, torch.float32).contiguous()
    Q, D = q.shape
    N = pb.bank.shape[0]
    assert D == pb.bank.shape[1]
    if k < 1:
        raise ValueError(f"cosine_topk: k = {k}")
    if world_size == 1 and k > N:
        raise ValueError(f"cosine_topk: k = {k} exceeds the {N} rows of the bank")
    if Q == 0:                                      # nothing to search for (an empty target list): empty result, no launch
        return (torch.empty(0, k, device=q.device), torch.empty(0, k, device=q.device, dtype=torch.int64))
    tw, qn = prepare_queries(q, pb.weights)
    if _prefilter_enabled() and ops.topk_prefilter_applicable(Q, N, D, k):
        out_s, out_i, n_redo = _local_topk_prefiltered(tw, qn, pb, k, eps)
        if stats is not None:
            stats.update(path="prefiltered", redone=n_redo)
    else:
        thr0 = pruning_floor(tw, qn, pb, k, eps) if prune else None
        out_s, out_i = _local_topk(tw, qn, pb.bank, pb.norms, k, eps, pb.idx_offset, thr0)
        if stats is not None:
            stats.update(path="exact", redone=0)
    if world_size > 1:
        from .distributed import gather_topk
        gs, gi = gather_topk(out_s, out_i, world_size, process_group)   # RCCL all-gather -> [Q, world, k]
        ops.topk_merge(gs, gi, Q, world_size, k, out_s, out_i)
    return out_s, out_i


def cosine_scores(queries: torch.Tensor, bank, weights: torch.Tensor | None = None, eps: float = 1e-6):
    """Plain [Q,N] score matrix (reference-shaped path with several patches per sample)."""
    pb = bank if isinstance(bank, PreparedBank) else PreparedBank(bank, weights)
    q = queries.to(pb.bank.device, torch.float32).contiguous()
    tw, qn = prepare_queries(q, pb.weights)
    out = torch.empty(q.shape[0], pb.bank.shape[0], device=q.device)
    ops.cosine_scores(tw, qn, pb.bank, pb.norms, eps, out)
    return out
