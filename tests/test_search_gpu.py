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


def test_full_size_bank_1m_x_768_against_oracle():
    """BASELINE configs[3] at full size on one GPU: 1,000,000 x 768 fp32 bank, k = 100, weighted.  The Q = 16 launch
    (bank-streaming kernel) and a 32-query sample of the Q = 10,000 launch (tiled many-query kernel, whatever stages it
    runs) must equal oracle/topk_oracle.c bit for bit: scores and indices.  ALL 10,000 answers of the two-stage (prefiltered)
    path are then compared with the exact single-stage kernel's on the same bank (0.2 s on the device; itself oracle-checked
    through the sample and the Q = 16 launch)."""
    import os
    from sky_embeddings_amd import search
    N, D, k, Q = 1_000_000, 768, 100, 10_000
    g = torch.Generator(device="cuda").manual_seed(2024)
    bank = torch.empty(N, D, device="cuda")
    for s0 in range(0, N, 50_000):
        bank[s0:s0 + 50_000] = torch.randn(50_000, D, device="cuda", generator=g)
    # exact duplicates of good rows far apart in the bank: ties must resolve to the lower index at full size too
    bank[N - 3] = bank[17]
    bank[N // 2 + 5] = bank[123_456]
    queries = torch.randn(Q, D, device="cuda", generator=torch.Generator(device="cuda").manual_seed(2025))
    queries[5] = bank[17] + 0.01 * queries[5]          # a query whose best rows are the planted duplicates
    queries[4242] = bank[123_456] + 0.01 * queries[4242]
    w = 1.0 / (torch.rand(D, device="cuda", generator=torch.Generator(device="cuda").manual_seed(7)) + 0.5) ** 2
    w = w / w.sum()
    pb = search.PreparedBank(bank, w)
    s16, i16 = search.cosine_topk(queries[:16], pb, k)
    stL = {}
    sL, iL = search.cosine_topk(queries, pb, k, stats=stL)
    assert stL["path"] == "prefiltered"
    os.environ["SKYEMB_TOPK_PREFILTER"] = "0"
    try:
        stE = {}
        sE, iE = search.cosine_topk(queries, pb, k, stats=stE)
    finally:
        del os.environ["SKYEMB_TOPK_PREFILTER"]
    torch.cuda.synchronize()
    assert stE["path"] == "exact"
    assert torch.equal(iL, iE) and torch.equal(sL, sE), (iL != iE).any(dim=1).nonzero()[:5]      # every one of the 10,000 queries
    sample = np.unique(np.concatenate(([0, 5, 15, 16, 63, 64, 4242, Q - 1], np.random.default_rng(1).integers(0, Q, 24))))[:32]
    x = bank.cpu().numpy()
    qh = queries.cpu().numpy()
    wh = w.cpu().numpy()
    ref_s, ref_i = so.cosine_topk_np(qh[:16], x, k, wh)
    assert np.array_equal(i16.cpu().numpy(), ref_i) and np.array_equal(s16.cpu().numpy(), ref_s)
    assert ref_i[5, 0] == 17 and (N - 3) in ref_i[5, :2]
    ref_s, ref_i = so.cosine_topk_np(qh[sample], x, k, wh)
    got_i, got_s = iL.cpu().numpy()[sample], sL.cpu().numpy()[sample]
    assert np.array_equal(got_i, ref_i), np.argwhere(got_i != ref_i)[:5]
    assert np.array_equal(got_s, ref_s)
    # and the two kernels agree with each other on the queries both served
    assert torch.equal(iL[:16], i16) and torch.equal(sL[:16], s16)


def _oracle_equal(q, x, w, k, **kw):
    from sky_embeddings_amd import search
    stats = {}
    pb = search.PreparedBank(torch.from_numpy(x).cuda(), None if w is None else torch.from_numpy(w).cuda())
    s, i = search.cosine_topk(torch.from_numpy(q).cuda(), pb, k, stats=stats, **kw)
    ref_s, ref_i = so.cosine_topk_np(q, x, k, w)
    assert np.array_equal(i.cpu().numpy(), ref_i), (stats, np.argwhere(i.cpu().numpy() != ref_i)[:5])
    assert np.array_equal(s.cpu().numpy(), ref_s), stats
    return stats


@pytest.mark.parametrize("Q,N,D,k,weighted", [(64, 2048, 128, 10, True), (130, 21000, 128, 100, True), (300, 60000, 768, 100, False),
                                              (77, 9000, 192, 150, True), (1000, 30000, 160, 1, True), (17, 12000, 768, 100, True),
                                              (33, 5000, 128, 7, False)])
def test_prefiltered_many_query_topk_is_bit_exact(Q, N, D, k, weighted):
    """Two-stage path (fp16 matrix-core prefilter with a proven bound -> exact fp32 re-score) == oracle, bit for bit:
    ragged query / bank tiles, k up to the re-score quota, duplicates inside and across bank slices."""
    rng = np.random.default_rng(Q * 7 + N)
    q = rng.standard_normal((Q, D), dtype=np.float32)
    x = (rng.standard_normal((N, D)) * rng.uniform(0.2, 5.0, size=(N, 1))).astype(np.float32)   # rows of very different scale
    sc0 = so.cosine_scores_np(q[:1], x, None)[0]
    best = np.argsort(-sc0)[:3]
    x[N - 1], x[N // 2 + 1], x[7] = x[best[0]], x[best[1]], x[best[2]]      # exact ties: lower index first
    w = None
    if weighted:
        w = (1.0 / (rng.random(D, dtype=np.float32) + 0.05) ** 2).astype(np.float32)   # weights spread over 2.5 decades
        w /= w.sum()
    stats = _oracle_equal(q, x, w, k)
    assert stats["path"] == "prefiltered" and stats["redone"] <= Q // 10


def test_prefiltered_topk_certification_failures_fall_back_to_the_exact_kernel():
    """A bank crowded around the k-th score (hundreds of near-duplicates of the query's best rows), rows with NaN / Inf /
    all zeros, a zero query: the second stage cannot certify every answer from its re-score quota -- those queries are
    flagged and re-run by the exact kernel; results stay bit-identical to the oracle."""
    rng = np.random.default_rng(5)
    Q, N, D, k = 96, 12000, 128, 60
    q = rng.standard_normal((Q, D), dtype=np.float32)
    x = rng.standard_normal((N, D), dtype=np.float32)
    x[1000:1600] = q[3] + 1e-5 * rng.standard_normal((600, D)).astype(np.float32)    # 600 rows within ~1e-7 of each other for query 3
    x[2000:2300] = 2.5 * q[10] + 1e-6 * rng.standard_normal((300, D)).astype(np.float32)
    x[50] = np.nan
    x[51, 7] = np.inf
    x[52] = 0.0
    x[53, :] = 1e-30                                                                  # underflows in the fp16 image
    q[20] = 0.0
    w = rng.random(D, dtype=np.float32) + 0.1
    w /= w.sum()
    stats = _oracle_equal(q, x, w, k)
    assert stats["path"] == "prefiltered" and stats["redone"] >= 2                    # queries 3 and 10 at least


def test_prefiltered_denormal_scale_rows_and_queries_reach_the_exact_stage():
    """A non-zero bank row whose largest |x| is below ~2^-113 (its power-of-two scale would overflow) scores ~0 exactly and
    belongs in the top-k of a query whose k-th best score is NEGATIVE; a query that small is answered by the exact kernel.
    Before the clamp the row's fp16 image was inf / NaN and the row was silently dropped."""
    rng = np.random.default_rng(17)
    Q, N, D, k = 40, 6000, 128, 5
    base = rng.standard_normal(D).astype(np.float32)
    q = (base + 0.3 * rng.standard_normal((Q, D))).astype(np.float32)
    x = (-np.abs(rng.uniform(0.5, 2.0, size=(N, 1))) * base + 0.3 * rng.standard_normal((N, D))).astype(np.float32)   # anti-aligned bank
    tiny = np.float32(2.0 ** -120)
    x[123] = tiny * rng.standard_normal(D).astype(np.float32)        # denormal-scale rows: exact score ~ 0 > every other row's
    x[4567] = tiny * np.abs(base)
    q[9] = np.float32(2.0 ** -125) * q[9]                             # denormal-scale query
    ref_s, ref_i = so.cosine_topk_np(q, x, k, None)
    assert (ref_s[0] < 0).sum() >= k - 2 and {123, 4567} <= set(ref_i[0].tolist())    # the case the advisor described
    stats = _oracle_equal(q, x, None, k)
    assert stats["path"] == "prefiltered" and stats["redone"] >= 1


def test_prefiltered_candidate_floods_are_flagged_not_lost():
    """Tens of thousands of rows that all score within the fp16 interval of a query's k-th best, placed where the staged
    phases (LDS staging list of 2048 candidates per 256 x 256 item, per-workgroup regions) meet them: the lists overflow, the
    flooded queries are flagged and re-run by the exact kernel, everything stays bit-identical to the oracle."""
    rng = np.random.default_rng(11)
    Q, N, D, k = 300, 60000, 128, 50
    q = rng.standard_normal((Q, D), dtype=np.float32)
    x = rng.standard_normal((N, D), dtype=np.float32)
    x[20000:50000] = q[7] + 1e-4 * rng.standard_normal((30000, D)).astype(np.float32)     # 30 000 near-copies of query 7
    x[52000:56000] = -3.0 * q[200] + 1e-4 * rng.standard_normal((4000, D)).astype(np.float32)   # anti-aligned: never candidates
    stats = _oracle_equal(q, x, None, k)
    assert stats["path"] == "prefiltered" and 1 <= stats["redone"] <= 30


def test_prefiltered_two_pass_variant_is_bit_exact_too(monkeypatch):
    """SKYEMB_PREFILTER_LO=1: the first stage multiplies the query as hi + lo fp16 halves (tighter intervals, 128-query tiles);
    same exact results."""
    monkeypatch.setenv("SKYEMB_PREFILTER_LO", "1")
    rng = np.random.default_rng(12)
    Q, N, D, k = 200, 30000, 256, 100
    q = rng.standard_normal((Q, D), dtype=np.float32)
    x = (rng.standard_normal((N, D)) * rng.uniform(0.2, 5.0, size=(N, 1))).astype(np.float32)
    w = (1.0 / (rng.random(D, dtype=np.float32) + 0.05) ** 2).astype(np.float32)
    stats = _oracle_equal(q, x, w / w.sum(), k)
    assert stats["path"] == "prefiltered" and stats["redone"] <= Q // 10


def test_prefiltered_equals_exact_kernel_at_mid_size():
    """200k x 256 bank, 512 queries: the two device paths agree bit for bit (no CPU oracle at this size)."""
    import os
    from sky_embeddings_amd import search
    g = torch.Generator(device="cuda").manual_seed(3)
    bank = torch.randn(200_000, 256, device="cuda", generator=g)
    queries = torch.randn(512, 256, device="cuda", generator=g)
    w = torch.rand(256, device="cuda", generator=g) + 0.1
    pb = search.PreparedBank(bank, w / w.sum())
    st = {}
    s1, i1 = search.cosine_topk(queries, pb, 100, stats=st)
    assert st["path"] == "prefiltered" and st["redone"] == 0
    os.environ["SKYEMB_TOPK_PREFILTER"] = "0"
    try:
        s2, i2 = search.cosine_topk(queries, pb, 100, stats=st)
    finally:
        del os.environ["SKYEMB_TOPK_PREFILTER"]
    assert st["path"] == "exact"
    assert torch.equal(i1, i2) and torch.equal(s1, s2)


def test_degenerate_search_inputs():
    """No queries -> empty lists; k outside [1, rows of the bank] -> ValueError (nothing is launched)."""
    from sky_embeddings_amd import search
    bank = torch.randn(50, 64, device="cuda")
    s, i = search.cosine_topk(torch.empty(0, 64, device="cuda"), bank, 5)
    assert s.shape == (0, 5) and i.shape == (0, 5) and i.dtype == torch.int64
    s, i = search.cosine_topk(torch.randn(3, 64, device="cuda"), bank, 50)            # k == N: every row, best first
    assert torch.isfinite(s).all() and sorted(i[0].tolist()) == list(range(50))
    for k in (0, 51):
        with pytest.raises(ValueError):
            search.cosine_topk(torch.randn(3, 64, device="cuda"), bank, k)


def test_sample_floor_is_a_valid_and_tight_lower_bound():
    """skyemb_cosine_sample_floor (tile maxima of the sample's scores + one-wave selection) never exceeds the k-th best score of
    the sample itself (so of any bank containing it), equals the k-th largest tile maximum minus one ulp, and sits within a
    few ranks of the sample's own k-th best."""
    from sky_embeddings_amd import ops
    rng = np.random.default_rng(3)
    S, D, k = 25600, 768, 100
    x = rng.standard_normal((S, D), dtype=np.float32)
    w = (rng.random(D, dtype=np.float32) + 0.5)
    w /= w.sum()
    for Q in (1, 16, 5):
        q = rng.standard_normal((Q, D), dtype=np.float32)
        sc = so.cosine_scores_np(q, x, w)                                  # oracle scores [Q, S]
        xd, qd, wd = torch.from_numpy(x).cuda(), torch.from_numpy(q).cuda(), torch.from_numpy(w).cuda()
        xn, qn, tw = torch.empty(S, device="cuda"), torch.empty(Q, device="cuda"), torch.empty(Q, D, device="cuda")
        ops.weighted_norms(xd, wd, xn)
        ops.weighted_norms(qd, wd, qn, tw)
        floor, ws = torch.empty(Q, device="cuda"), torch.empty(Q * (S // 16), device="cuda")
        ops.cosine_sample_floor(tw, qn, xd, xn, k, 1e-6, ws, floor)
        got = floor.cpu().numpy()
        tile_max = sc.reshape(Q, S // 16, 16).max(axis=2)
        want = np.nextafter(np.sort(tile_max, axis=1)[:, ::-1][:, k - 1], np.float32(-np.inf))
        assert np.array_equal(got, want)
        kth_best = np.sort(sc, axis=1)[:, ::-1][:, k - 1]
        assert (got < kth_best).all()
        rank_of_floor = (sc > got[:, None]).sum(axis=1)                   # rows of the sample above the floor
        assert (rank_of_floor >= k).all() and (rank_of_floor <= k + 12).all(), rank_of_floor


def test_small_q_search_with_the_streaming_kernels_switched_off(tmp_path):
    """SKYEMB_TOPK_STREAM=0 (route everything to the simpler kernels) must still give a pruned, bit-exact search:
    the sample floor then comes from ``skyemb_cosine_scores`` + ``skyemb_kth_largest_floor`` (the library's own predicate
    ``skyemb_cosine_sample_floor_applicable`` says so), not from a call that rejects its arguments.  The switch is read once per
    process, hence the child process."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import numpy as np, torch
from oracle import similarity_oracle as so
from sky_embeddings_amd import ops, search
Q, N, D, k = 3, 21000, 128, 10
assert not ops.sample_floor_applicable(Q, 256 * k, D, k)
rng = np.random.default_rng(5)
q = rng.standard_normal((Q, D), dtype=np.float32); x = rng.standard_normal((N, D), dtype=np.float32)
w = rng.random(D, dtype=np.float32) + 0.1
pb = search.PreparedBank(torch.from_numpy(x).cuda(), torch.from_numpy(w).cuda())
tw, qn = search.prepare_queries(torch.from_numpy(q).cuda(), pb.weights)
assert search.pruning_floor(tw, qn, pb, k, 1e-6) is not None
s, i = search.cosine_topk(torch.from_numpy(q).cuda(), pb, k)
rs, ri = so.cosine_topk_np(q, x, k, w)
assert np.array_equal(i.cpu().numpy(), ri) and np.array_equal(s.cpu().numpy(), rs)
print("stream-off search ok")
'''
    env = dict(os.environ, SKYEMB_TOPK_STREAM="0", PYTHONPATH=root)
    out = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "stream-off search ok" in out.stdout, out.stdout[-1500:] + out.stderr[-3000:]


def test_sample_floor_predicate_follows_the_shape_limits():
    from sky_embeddings_amd import ops
    assert ops.sample_floor_applicable(16, 25600, 768, 100)
    assert not ops.sample_floor_applicable(17, 25600, 768, 100)        # more than 16 queries
    assert not ops.sample_floor_applicable(4, 25600, 96, 100)          # D % 64
    assert not ops.sample_floor_applicable(4, 25600, 768, 2000)        # k above the number of 16-row tiles
    assert not ops.sample_floor_applicable(4, 16 * 4096, 768, 100)     # more than 2048 tiles
    t = torch.empty(64 * 768 + 1, device="cuda")[1:].view(64, 768)     # rows 4 bytes off a 16-byte boundary
    assert not ops.sample_floor_applicable(4, 25600, 768, 100, t)
