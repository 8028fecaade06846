"""GPU: the N > 1 training step really executed -- two processes, one TrainStep each (HIP-graph stages, per-stage
gradient all-reduce, fused AdamW), against ONE process on the concatenated batch.  On a box with two GPUs the ranks use
RCCL on their own device; on a one-GPU box both ranks share cuda:0 and the collectives go through gloo (the schedule,
the bucketing, the bf16 / fp32 gradient communication and the optimiser are the same code either way)."""
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

STEPS, B_RANK, WORLD, LR = 3, 4, 2, 1e-3


def _data(world=WORLD):
    g = torch.Generator().manual_seed(42)
    imgs = torch.randn(world * B_RANK, 5, 64, 64, generator=g).clamp_(min=-3.0)
    noise = torch.rand(STEPS, world * B_RANK, 16, generator=g)
    return imgs, noise


def _make(dev, B, world, dtype=torch.bfloat16, **kw):
    from sky_embeddings_amd.engine import MAEEngine
    from sky_embeddings_amd.model_config import config_for
    from sky_embeddings_amd.optim import CosineLR, FusedAdamW
    from sky_embeddings_amd.train_step import TrainStep
    cfg = config_for("tiny", img_size=64, patch_size=16, in_chans=5, embed_dim=192)
    eng = MAEEngine(cfg, device=dev, compute_dtype=dtype, seed=1)
    opt = FusedAdamW(eng, lr=LR, betas=(0.9, 0.95), weight_decay=0.05)
    step = TrainStep(eng, opt, CosineLR(opt, 100), B, world_size=world, external_noise=True, **kw)        # default staging: 6 encoder groups with N > 1, 3 with one rank
    return eng, step


def _rank_main(rank, port, out_dir, grad_comm, use_nccl, world=WORLD, shard=None, dtype=torch.bfloat16):
    import torch.distributed as dist
    dev = torch.device("cuda", rank if use_nccl else 0)
    torch.cuda.set_device(dev)
    if use_nccl:
        dist.init_process_group("nccl", rank=rank, world_size=world, init_method=f"tcp://127.0.0.1:{port}", device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world, init_method=f"tcp://127.0.0.1:{port}")
    imgs, noise = _data(world)
    rows = slice(rank * B_RANK, (rank + 1) * B_RANK)
    eng, step = _make(dev, B_RANK, world, dtype=dtype, grad_comm=grad_comm, shard_optimizer=shard)
    if dtype == torch.float16:       # fp16 mode: 16-bit mirror in fp16, sums of world x 2^16 x gradients, the optimiser divides both out
        assert step.g16 is None or step.g16.dtype == torch.float16
        assert step.optimizer.grad_scale == 1.0 / (world * eng.loss_scale) and eng.loss_scale == 1024.0     # 4 images: 61 k masked pixels / 64
    assert step.staged and len(step.stages) >= 4          # decoder | encoder groups | embedding: comm overlaps backward
    # (default with N > 1: the optimiser sharded over the ranks -- reduce-scatter, AdamW on the owned chunks, all-gather of the shadow)
    assert step.shard_optimizer == (shard is not False) and step.shard_world == world and step.shard_rank == rank
    losses = []
    for it in range(STEPS):
        step.noise.copy_(noise[it][rows])
        losses.append(float(step(imgs[rows].to(dev))))
    if step.shard_optimizer:
        # a chunk's fp32 master weights are current on its owner only (the other ranks hold the gathered 16-bit shadow)
        own_s, own_e = next(iter(step._own))
        stale = not torch.equal(eng.store.p_lp[own_s:own_e].float(), eng.store.p[own_s:own_e].to(eng.store.p_lp.dtype).float())
        assert stale, "no non-owned chunk went stale: the optimiser was not sharded"
        step.gather_full_state()
    torch.cuda.synchronize(dev)
    assert torch.equal(eng.store.p_lp.float(), eng.store.p.to(eng.store.p_lp.dtype).float())       # shadow == rounded master, everywhere
    torch.save({"losses": losses, "p": eng.store.p.cpu(), "m": eng.store.m.cpu()}, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


# world 4 = the most ranks a one-GPU box lets a test start next to the test process itself (six GPU processes per card)
# (four ranks cost a minute of process start-up each: one world-4 case; fp32 and 16-bit mirrors, the replicated schedule and fp16 compute at world 2)
@pytest.mark.parametrize("grad_comm,world,shard,dtype", [("f32", 2, None, "bf16"), ("bf16", 4, None, "bf16"), ("bf16", 2, False, "bf16"),
                                                         ("bf16", 2, None, "f16")])
def test_n_rank_training_step_matches_one_rank_on_the_concatenated_batch(tmp_path, grad_comm, world, shard, dtype):
    """shard = None: the default of an N > 1 job, the optimiser sharded over the ranks; False: the replicated schedule (all-reduce,
    every rank steps everything).  Both against ONE rank on the concatenated batch, to the same bars."""
    import torch.multiprocessing as mp
    use_nccl = torch.cuda.device_count() >= world
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    dtype = torch.float16 if dtype == "f16" else torch.bfloat16      # ("bf16" grad_comm = the 16-bit mirror in the compute dtype's format)
    mp.spawn(_rank_main, args=(port, str(tmp_path), grad_comm, use_nccl, world, shard, dtype), nprocs=world, join=True)
    r = [torch.load(tmp_path / f"rank{k}.pt") for k in range(world)]
    # the replicas stay identical, bit for bit (sharded: after the gather of the owners' fp32 state)
    assert all(torch.equal(r[0]["p"], rk["p"]) and torch.equal(r[0]["m"], rk["m"]) for rk in r[1:])
    # one process, the whole batch, fp32 gradients
    imgs, noise = _data(world)
    eng, step = _make(torch.device("cuda", 0), world * B_RANK, 1, dtype=dtype)
    p0 = eng.store.p.cpu().clone()
    ref_losses = []
    for it in range(STEPS):
        step.noise.copy_(noise[it])
        ref_losses.append(float(step(imgs.cuda())))
    torch.cuda.synchronize()
    ref = eng.store.p.cpu()
    mean_losses = np.mean([rk["losses"] for rk in r], axis=0)
    # MAE mode masks the same number of patches per sample, so the mean of the rank losses is the global loss and the mean of
    # the rank gradients the global gradient
    # (bars = 2x the errors measured on MI355X, profiles/r03_parity_errors.json: losses 1.0e-5 / 4.5e-5 relative, 99.98 % of the
    # parameters inside the band, largest deviation 1.36 lr-steps)
    assert np.allclose(mean_losses, ref_losses, rtol=2e-5 if grad_comm == "f32" else 1e-4), (mean_losses, ref_losses)
    moved = (ref - p0).abs()
    err = (r[0]["p"] - ref).abs()
    # Adam moves every element by ~lr per step; summation order (f32) or bf16 rounding of the gradients can change the
    # move of an element whose gradient is rounding noise, but not the bulk
    frac_close = float((err <= (0.05 if grad_comm == "f32" else 0.25) * LR * STEPS).float().mean())
    from tests.helpers import record_parity
    record_parity(f"ddp_{world}_ranks_vs_one_{grad_comm}" + ("_replicated" if shard is False else "_sharded_optimizer") + ("_f16" if dtype == torch.float16 else ""),
                  dict(loss_rel_max=float(np.max(np.abs(mean_losses - np.asarray(ref_losses)) / np.abs(ref_losses))),
                       frac_within_band=frac_close, band_in_lr_steps=0.05 if grad_comm == "f32" else 0.25,
                       err_max_in_lr_steps=float(err.max()) / (LR * STEPS), err_mean_in_lr_steps=float(err.mean()) / (LR * STEPS),
                       comm_dtype=grad_comm, backend="nccl" if use_nccl else "gloo (both ranks on cuda:0)"))
    assert frac_close > (0.9995 if world == 2 else 0.999), frac_close
    assert float(err.max()) <= 2.5 * LR * STEPS and float(moved.max()) > 0.5 * LR


# ----------------------------------------------------------------------------------------------------------------
# RCCL itself, with the one rank a one-GPU box allows (two ranks on one card are refused by RCCL: "duplicate GPU"): the "nccl"
# branches of the package -- process-group start-up with device_id, bf16 / fp32 all-reduce of mirror slices issued async between the
# stage graphs, their waits on the optimiser stream, all_gather_into_tensor of the search lists, barrier + float64 MAX as bench.py
# uses them -- run on the real library; a sum over one rank is the identity, so every result must equal the collective-free run
# ----------------------------------------------------------------------------------------------------------------
def _rccl_one_rank_main(rank, port, out_dir):
    import torch.distributed as dist
    from sky_embeddings_amd import distributed as skd
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, init_method=f"tcp://127.0.0.1:{port}", device_id=dev)
    assert dist.get_backend() == "nccl"
    imgs, noise = _data(1)
    res = {}
    # (default schedule of an N > 1 job: waits on the compute stream, one AdamW launch; and the optimiser-stream overlap)
    # ... and (round 6) the sharded optimiser's collectives over the world of one: reduce_scatter_tensor into the owner's buffer,
    # the in-place all_gather_into_tensor of the 16-bit shadow, the tails' all-reduce -- copies with one rank, issued on RCCL
    for comm, overlap, shard in (("bf16", False, False), ("bf16", True, False), ("f32", False, False), ("bf16", False, True), ("f32", False, True)):
        for force in ("1", "0"):
            os.environ["SKYEMB_DIST_FORCE"] = force
            eng, step = _make(dev, B_RANK, 1, staged=True, grad_comm=comm, n_encoder_groups=6, optimizer_overlap=overlap,
                              shard_optimizer=bool(shard and force == "1"))
            assert step.shard_optimizer == bool(shard and force == "1")
            assert step.collectives == (force == "1") and step.staged and len(step.stages) >= 4
            assert (step.g16 is not None) == (comm == "bf16") and step.optimizer_overlap == overlap
            losses = []
            for it in range(STEPS):
                step.noise.copy_(noise[it][:B_RANK])
                losses.append(float(step(imgs[:B_RANK].to(dev))))
            torch.cuda.synchronize(dev)
            res[force] = (losses, eng.store.p.clone(), eng.store.m.clone())
            del eng, step
        a, b = res["1"], res["0"]
        assert a[0] == b[0] and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]), (comm, overlap, shard)
    # the search lists: all_gather_into_tensor (the branch gloo never takes), [Q, k] -> [Q, world, k]
    g = torch.Generator().manual_seed(5)
    scores = torch.rand(33, 100, generator=g).to(dev)
    idx = torch.randint(0, 10 ** 6, (33, 100), generator=g).to(dev)
    gs, gi = skd._gather_lists(scores, idx, 1)
    assert gs.shape == (33, 1, 100) and torch.equal(gs[:, 0], scores) and torch.equal(gi[:, 0], idx)
    # bench.py's bracket: barrier, then the MAX over ranks of a float64 wall time
    dist.barrier()
    t = torch.tensor([1.25], device=dev, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert float(t) == 1.25
    works = skd.allreduce_flat_gradients(torch.ones(3 * 1024 * 1024 + 5, device=dev), 2, bucket_elems=1024 * 1024, async_op=True)
    assert len(works) == 4
    for w in works:
        w.wait()
    with open(os.path.join(out_dir, "ok"), "w") as f:
        f.write("ok")
    dist.destroy_process_group()


def test_rccl_one_rank_runs_the_nccl_branches(tmp_path):
    import torch.multiprocessing as mp
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    mp.spawn(_rccl_one_rank_main, args=(port, str(tmp_path)), nprocs=1, join=True)
    assert (tmp_path / "ok").exists()


# ----------------------------------------------------------------------------------------------------------------
# world size 8 (BASELINE configs[2] / [3]) walked in ONE process: a one-GPU box admits six GPU processes, so the eight ranks'
# arithmetic is run rank after rank on the same card -- every kernel and every reduction order of the 8-rank job, no collective
# ----------------------------------------------------------------------------------------------------------------
# bars = 2x the drift measured on MI355X (profiles/r05_parity_errors.json)
# (measured: 99.988 % of the parameters within a quarter lr-step between the two sums, 99.933 % against one rank; largest 1.45 lr-steps; losses 6.3e-5)
W8_FRAC_BAR, W8_MAX_BAR_LR_STEPS, W8_LOSS_BAR = 0.9986, 3.0, 1.3e-4


def test_world_8_gradient_sums_bf16_against_fp32_three_steps():
    """Eight ranks' gradients of three optimiser steps, summed (a) in fp32 -- nn.DataParallel's reduction, utils/mim_vit.py:117 -- and
    (b) the way the default bf16 gradient communication does: every rank's gradient rounded to bf16 (what the weight-gradient
    launches write into the mirror), partial sums rounded to bf16 after every addition (a ring all-reduce's seven roundings), AdamW
    reading the bf16 sum with grad_scale = 1/8.  Recorded and enforced: the parameter drift between the two after 3 steps, and
    both against one rank on the concatenated batch."""
    from tests.helpers import record_parity
    world = 8
    imgs, noise = _data(world)
    dev = torch.device("cuda", 0)
    finals, losses = {}, {}
    for comm in ("f32", "bf16"):
        eng, step = _make(dev, B_RANK, 1, fused_adamw=False, use_graph=False)
        opt = step.optimizer
        opt.grad_scale = 1.0 / world
        g16 = torch.zeros(eng.store.n, device=dev, dtype=torch.bfloat16)
        ls = []
        for it in range(STEPS):
            acc32 = torch.zeros_like(eng.store.g)
            acc16 = None
            rank_losses = []
            for rk in range(world):
                rows = slice(rk * B_RANK, (rk + 1) * B_RANK)
                loss, _, _ = eng.forward_train(imgs[rows].to(dev), 0.75, noise[it][rows].to(dev))
                eng.backward()
                rank_losses.append(float(loss))
                if comm == "f32":
                    acc32 += eng.store.g
                else:
                    mine = eng.store.g.bfloat16()
                    acc16 = mine if acc16 is None else (acc16.float() + mine.float()).bfloat16()
            ls.append(float(np.mean(rank_losses)))
            if comm == "f32":
                eng.store.g.copy_(acc32)
                opt.grad_buffer = None
            else:
                g16.copy_(acc16)
                opt.grad_buffer = g16
            opt.step()
            opt.grad_buffer = None
            step.scheduler.step()
        torch.cuda.synchronize()
        finals[comm], losses[comm] = eng.store.p.cpu().clone(), ls
        del eng, step
    eng, step = _make(dev, world * B_RANK, 1)
    p0 = eng.store.p.cpu().clone()
    ref_losses = []
    for it in range(STEPS):
        step.noise.copy_(noise[it])
        ref_losses.append(float(step(imgs.cuda())))
    torch.cuda.synchronize()
    ref = eng.store.p.cpu()
    band = LR * STEPS
    rec = {}
    for name, a, b in (("bf16_vs_f32", finals["bf16"], finals["f32"]), ("f32_vs_one_rank", finals["f32"], ref), ("bf16_vs_one_rank", finals["bf16"], ref)):
        err = (a - b).abs()
        rec[name] = dict(frac_within_quarter_lr_step=float((err <= 0.25 * band).float().mean()), err_max_in_lr_steps=float(err.max()) / band,
                         err_mean_in_lr_steps=float(err.mean()) / band)
    rec["loss_rel_max_bf16"] = float(np.max(np.abs(np.asarray(losses["bf16"]) - np.asarray(ref_losses)) / np.abs(ref_losses)))
    rec["loss_rel_max_f32"] = float(np.max(np.abs(np.asarray(losses["f32"]) - np.asarray(ref_losses)) / np.abs(ref_losses)))
    record_parity("ddp_world8_emulated_three_steps", rec)
    assert float((ref - p0).abs().max()) > 0.5 * LR
    for name in ("bf16_vs_f32", "f32_vs_one_rank", "bf16_vs_one_rank"):
        assert rec[name]["frac_within_quarter_lr_step"] > W8_FRAC_BAR, (name, rec[name])
        assert rec[name]["err_max_in_lr_steps"] <= W8_MAX_BAR_LR_STEPS, (name, rec[name])
    assert rec["loss_rel_max_bf16"] <= W8_LOSS_BAR and rec["loss_rel_max_f32"] <= W8_LOSS_BAR, rec


@pytest.mark.parametrize("Q", [16, 512])
def test_world_8_sharded_search_at_the_real_shard_size(Q):
    """BASELINE configs[3]'s split at its real shard size: 8 shards of 125,000 x 768 rows (the sizes at which the per-rank path choices
    -- streaming kernel + sample floor for Q <= 16, the two-stage prefilter above -- are made), each shard searched as its rank
    would (global indices through idx_offset), the eight [Q, k] lists merged by the HIP k-way merge: == the single 1M-row bank bit
    for bit, == the CPU oracle on sampled queries, with one row's copies planted in FOUR different shards (ties -> ascending index)."""
    from oracle import similarity_oracle as so
    from sky_embeddings_amd import ops, search
    from sky_embeddings_amd.distributed import shard_rows
    N, D, k, world = 1_000_000, 768, 100, 8
    g = torch.Generator(device="cuda").manual_seed(88)
    bank = torch.empty(N, D, device="cuda")
    for s0 in range(0, N, 50_000):
        bank[s0:s0 + 50_000] = torch.randn(50_000, D, device="cuda", generator=g)
    queries = torch.randn(Q, D, device="cuda", generator=g)
    queries[0] = bank[31] + 0.01 * queries[0]
    for row in (130_000, 400_017, 999_990):                  # shards 1, 3 and 7 hold a copy of shard 0's row 31
        bank[row] = bank[31]
    w = 1.0 / (torch.rand(D, device="cuda", generator=g) + 0.5) ** 2
    w = w / w.sum()
    parts, paths = [], set()
    for rk in range(world):
        lo, hi = shard_rows(N, rk, world)
        assert hi - lo == 125_000
        st = {}
        pb = search.PreparedBank(bank[lo:hi], w, idx_offset=lo)
        parts.append(search.cosine_topk(queries, pb, k, stats=st))
        paths.add(st["path"])
        del pb
    assert paths == ({"exact"} if Q <= 16 else {"prefiltered"}), paths
    gs = torch.stack([p[0] for p in parts], dim=1).contiguous()
    gi = torch.stack([p[1] for p in parts], dim=1).contiguous()
    out_s, out_i = torch.empty(Q, k, device="cuda"), torch.empty(Q, k, device="cuda", dtype=torch.int64)
    ops.topk_merge(gs, gi, Q, world, k, out_s, out_i)
    whole_s, whole_i = search.cosine_topk(queries, search.PreparedBank(bank, w), k)
    torch.cuda.synchronize()
    assert torch.equal(out_i, whole_i) and torch.equal(out_s, whole_s)
    assert out_i[0, :4].tolist() == [31, 130_000, 400_017, 999_990]
    sample = np.unique(np.concatenate(([0, Q - 1], np.random.default_rng(2).integers(0, Q, 6))))
    ref_s, ref_i = so.cosine_topk_np(queries.cpu().numpy()[sample], bank.cpu().numpy(), k, w.cpu().numpy())
    assert np.array_equal(out_i.cpu().numpy()[sample], ref_i) and np.array_equal(out_s.cpu().numpy()[sample], ref_s)


# ----------------------------------------------------------------------------------------------------------------
# sharded-bank search in two processes: per-shard HIP top-k -> all-gather -> HIP merge (BASELINE configs[3]'s split)
# ----------------------------------------------------------------------------------------------------------------
def _search_case(Q):
    """Bank with exact duplicates planted ACROSS the shard boundary (ties -> lower global index) and inside one shard."""
    from oracle import similarity_oracle as so
    N, D, k = 30000 if Q <= 16 else 12000, 128, 12
    rng = np.random.default_rng(1000 + Q)
    q = rng.standard_normal((Q, D), dtype=np.float32)
    x = rng.standard_normal((N, D), dtype=np.float32)
    w = rng.random(D, dtype=np.float32) + 0.1
    best = np.argsort(-so.cosine_scores_np(q[:1], x, w)[0])[:3]
    lo_half = [b for b in best if b < N // 2]
    hi_half = [b for b in best if b >= N // 2]
    # a copy of a top row of query 0 in the OTHER shard: the two ranks return the same score, the merge must order them by index
    if lo_half:
        x[N - 5] = x[lo_half[0]]
    if hi_half:
        x[3] = x[hi_half[0]]
    x[N // 2 - 1] = x[N // 2 + 7] = x[best[0]]               # and one either side of the seam
    return q, x, w, k


def _search_rank_main(rank, port, out_dir, Q, use_nccl):
    import torch.distributed as dist
    from sky_embeddings_amd import search
    from sky_embeddings_amd.distributed import shard_rows
    dev = torch.device("cuda", rank if use_nccl else 0)
    torch.cuda.set_device(dev)
    if use_nccl:
        dist.init_process_group("nccl", rank=rank, world_size=WORLD, init_method=f"tcp://127.0.0.1:{port}", device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=WORLD, init_method=f"tcp://127.0.0.1:{port}")
    q, x, w, k = _search_case(Q)
    lo, hi = shard_rows(x.shape[0], rank, WORLD)
    pb = search.PreparedBank(torch.from_numpy(x[lo:hi]).to(dev), torch.from_numpy(w).to(dev), idx_offset=lo)
    stats = {}
    s, i = search.cosine_topk(torch.from_numpy(q).to(dev), pb, k, world_size=WORLD, stats=stats)
    torch.cuda.synchronize(dev)
    torch.save({"s": s.cpu(), "i": i.cpu(), "path": stats["path"]}, os.path.join(out_dir, f"search{Q}_rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("Q", [5, 40])           # the bank-streaming kernel (Q <= 16) and the two-stage many-query path
def test_two_rank_sharded_search_equals_single_bank_and_oracle(tmp_path, Q):
    import torch.multiprocessing as mp
    from oracle import similarity_oracle as so
    from sky_embeddings_amd import search
    use_nccl = torch.cuda.device_count() >= WORLD
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    mp.spawn(_search_rank_main, args=(port, str(tmp_path), Q, use_nccl), nprocs=WORLD, join=True)
    r = [torch.load(tmp_path / f"search{Q}_rank{k}.pt") for k in range(WORLD)]
    assert torch.equal(r[0]["i"], r[1]["i"]) and torch.equal(r[0]["s"], r[1]["s"])     # every rank holds the merged answer
    assert r[0]["path"] == ("exact" if Q <= 16 else "prefiltered")
    q, x, w, k = _search_case(Q)
    ref_s, ref_i = so.cosine_topk_np(q, x, k, w)
    assert np.array_equal(r[0]["i"].numpy(), ref_i) and np.array_equal(r[0]["s"].numpy(), ref_s)
    # the planted cross-shard duplicates really are in query 0's answer, lower index first
    sc = r[0]["s"][0].numpy()
    ties = np.nonzero(sc[1:] == sc[:-1])[0]
    assert len(ties) >= 2 and all(ref_i[0][t] < ref_i[0][t + 1] for t in ties)
    whole_s, whole_i = search.cosine_topk(torch.from_numpy(q).cuda(), torch.from_numpy(x).cuda(), k, weights=torch.from_numpy(w).cuda())
    assert torch.equal(whole_i.cpu(), r[0]["i"]) and torch.equal(whole_s.cpu(), r[0]["s"])


# ----------------------------------------------------------------------------------------------------------------
# the pretraining entry point itself under the launcher, two ranks
# ----------------------------------------------------------------------------------------------------------------
def test_pretrain_entry_point_under_the_launcher_two_ranks(tmp_path):
    """``python -m torch.distributed.run --nproc-per-node 2 pretrain_mim.py <ini>``: rank 0 writes the checkpoint, both ranks end
    with bit-equal master weights.  One GPU: both ranks on cuda:0 over gloo (SKYEMB_DIST_BACKEND); two or more: RCCL."""
    import configparser
    import subprocess
    import sys
    from sky_embeddings_amd import hdf5_lite
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    dd = tmp_path / "data"
    dd.mkdir()
    hdf5_lite.make_synthetic_cutouts(str(dd / "synthetic_cutouts_GRIZY_64_train.h5"), n=96, seed=1234)
    hdf5_lite.make_synthetic_cutouts(str(dd / "synthetic_cutouts_GRIZY_64_val.h5"), n=16, seed=4321)
    work = tmp_path / "work"
    (work / "configs").mkdir(parents=True)
    cfg = configparser.ConfigParser()
    cfg.read(os.path.join(ROOT, "configs", "mim_1.ini"))
    cfg["TRAINING"]["total_batch_iters"] = "8"
    cfg["TRAINING"]["batch_size"] = "8"
    with open(work / "configs" / "mim_t.ini", "w") as fh:
        cfg.write(fh)
    for name in ("pretrain_mim.py", "utils", "sky_embeddings_amd"):
        os.symlink(os.path.join(ROOT, name), work / name)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, PYTHONPATH=str(work), SKYEMB_SAVE_RANK_PARAMS=str(tmp_path), HSA_ENABLE_IPC_MODE_LEGACY="0")
    if torch.cuda.device_count() < WORLD:
        env["SKYEMB_DIST_BACKEND"] = "gloo"
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(WORLD),
                          "--master-addr", "127.0.0.1", "--master-port", str(port),
                          str(work / "pretrain_mim.py"), "mim_t", "-v", "4", "-ct", "0.001", "-dd", str(dd)],
                         cwd=str(work), env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    assert "2 process(es)" in out.stdout and out.stdout.count("Training complete.") == WORLD
    ck = torch.load(str(work / "models" / "mim_t.pth.tar"), map_location="cpu", weights_only=False)
    assert ck["batch_iters"] >= 8 and np.isfinite(ck["losses"]["train_loss"]).all()
    p0, p1 = torch.load(tmp_path / "rank0.pt"), torch.load(tmp_path / "rank1.pt")
    assert torch.equal(p0, p1)
    # and training happened: the optimiser state in the checkpoint carries the step count of the run
    steps = {float(v["step"]) for v in ck["optimizer"]["state"].values()}
    assert len(steps) == 1 and steps.pop() >= 8
