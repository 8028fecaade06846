"""GPU: the host-level search API (sky_embeddings_amd.search) -- sample-derived pruning floor, the
bank-streaming kernel (Q <= 16) and the tiled kernel -- against the C oracle, bit for bit."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import similarity_oracle as so


@pytest.mark.parametrize("Q,N,D,k", [(1, 30000, 768, 20), (3, 21000, 128, 10), (16, 26000, 64, 16), (40, 21000, 128, 10),
                                     (2, 900, 64, 100)])
@pytest.mark.parametrize("prune", [True, False])
def test_cosine_topk_with_and_without_pruning_floor(Q, N, D, k, prune):
    from sky_embeddings_amd import search
    rng = np.random.default_rng(N + Q)
    q = rng.standard_normal((Q, D), dtype=np.float32)
    x = rng.standard_normal((N, D), dtype=np.float32)
    # plant exact duplicates (ties broken by lower index) among the best rows of query 0, inside and outside the sample
    sc0 = so.cosine_scores_np(q[:1], x, None)[0]
    best = np.argsort(-sc0)[:3]
    x[N - 1] = x[best[0]]
    x[N // 2 + 1] = x[best[1]]
    x[7] = x[best[2]]
    w = rng.random(D, dtype=np.float32) + 0.1
    w /= w.sum()
    ref_s, ref_i = so.cosine_topk_np(q, x, k, w)
    pb = search.PreparedBank(torch.from_numpy(x).cuda(), torch.from_numpy(w).cuda())
    s, i = search.cosine_topk(torch.from_numpy(q).cuda(), pb, k, prune=prune)
    assert np.array_equal(i.cpu().numpy(), ref_i)
    assert np.array_equal(s.cpu().numpy(), ref_s)
    if prune and N >= 8 * 256 * k:
        tw, qn = search.prepare_queries(torch.from_numpy(q).cuda(), pb.weights)
        floor = search.pruning_floor(tw, qn, pb, k, 1e-6)
        assert floor is not None and bool((floor.cpu().numpy() < ref_s[:, k - 1]).all())   # a valid lower bound


def test_sharded_bank_merge_equals_single_bank():
    """Two 'ranks' in one process: per-shard top-k (global indices via idx_offset) + k-way merge == whole bank."""
    from sky_embeddings_amd import ops, search
    rng = np.random.default_rng(3)
    Q, N, D, k = 5, 24000, 128, 12
    q = rng.standard_normal((Q, D), dtype=np.float32)
    x = rng.standard_normal((N, D), dtype=np.float32)
    x[20000] = x[100]
    qd = torch.from_numpy(q).cuda()
    whole_s, whole_i = search.cosine_topk(qd, torch.from_numpy(x).cuda(), k)
    parts = []
    for lo, hi in ((0, 12000), (12000, 24000)):
        pb = search.PreparedBank(torch.from_numpy(x[lo:hi]).cuda(), None, idx_offset=lo)
        parts.append(search.cosine_topk(qd, pb, k))
    gs = torch.stack([p[0] for p in parts], dim=1).contiguous()
    gi = torch.stack([p[1] for p in parts], dim=1).contiguous()
    out_s, out_i = torch.empty(Q, k, device="cuda"), torch.empty(Q, k, device="cuda", dtype=torch.int64)
    ops.topk_merge(gs, gi, Q, 2, k, out_s, out_i)
    assert torch.equal(out_i, whole_i) and torch.equal(out_s, whole_s)
    ref_s, ref_i = so.cosine_topk_np(q, x, k, None)
    assert np.array_equal(out_i.cpu().numpy(), ref_i)
