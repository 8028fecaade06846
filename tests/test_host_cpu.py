"""CPU: host-side logic of the product (no GPU compute): config / layout, HDF5 access + dataset
semantics, optimiser-state and schedule bookkeeping, and the N>1 paths under gloo (world_size 2)."""
import configparser
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from sky_embeddings_amd import distributed as sdist
from sky_embeddings_amd import hdf5_lite, model_config as mc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_model_layout_matches_reference_checkpoint_schema():
    cfg = mc.config_for("base", img_size=64, patch_size=16, in_chans=5, embed_dim=768)
    layout = mc.state_layout(cfg)
    assert len(layout) == 255                       # 255 tensors for MAE-B
    decay, no_decay = mc.weight_decay_split(cfg)
    assert (len(decay), len(no_decay)) == (86, 167)  # timm param_groups_weight_decay split
    n_train = sum(int(np.prod(s)) for n, s in layout if n not in mc.FROZEN)
    assert abs(n_train - 112.31e6) < 0.01e6         # 112.31 M trainable parameters
    # same names / shapes / order as the state dict captured from the reference (tiny geometry)
    z = np.load(os.path.join(ROOT, "tests", "golden", "mae_tiny_A.npz"))
    img, patch, C, D, depth, heads, Dd, ddepth, dheads, _ = [int(v) for v in z["cfg"]]
    tiny = mc.MAEConfig(img_size=img, patch_size=patch, in_chans=C, embed_dim=D, depth=depth, num_heads=heads,
                        decoder_embed_dim=Dd, decoder_depth=ddepth, decoder_num_heads=dheads)
    ref = [(k[len("state/"):], z[k].shape) for k in z.files if k.startswith("state/")]
    assert ref == [(n, tuple(s)) for n, s in mc.state_layout(tiny)]
    assert np.array_equal(mc.sincos_pos_embed(D, img // patch).astype(np.float32), z["state/pos_embed"][0])


def test_hdf5_lite_roundtrip_and_dataset_semantics(tmp_path):
    from sky_embeddings_amd.utils.dataloaders import H5Dataset, build_h5_dataloader
    path = str(tmp_path / "cut.h5")
    rng = np.random.default_rng(0)
    cut = rng.standard_normal((20, 5, 72, 72)).astype(np.float32) * 3
    cut[3, 1] = np.nan
    ra, dec = rng.uniform(0, 360, 20).astype(np.float32), rng.uniform(-90, 90, 20).astype(np.float32)
    hdf5_lite.write_datasets(path, {"cutouts": cut, "ra": ra, "dec": dec, "class": np.arange(20, dtype=np.int64)})
    with hdf5_lite.File(path) as f:
        assert sorted(f.keys()) == ["class", "cutouts", "dec", "ra"]
        assert f["cutouts"].shape == (20, 5, 72, 72) and f["cutouts"].dtype == np.float32
        assert np.array_equal(f["cutouts"][7], cut[7]) and np.array_equal(f["class"][:], np.arange(20))
        assert np.array_equal(np.asarray(f["ra"]), ra)
    ds = H5Dataset(path, img_size=64, patch_size=16, num_channels=5, max_mask_ratio=None, indices=[3, 7, 11])
    assert len(ds) == 3
    x, mask, rd = ds[0]                                  # utils/dataloaders.py:285-328 semantics
    assert x.shape == (5, 64, 64) and x.dtype == torch.float32 and mask.shape == x.shape and float(mask.sum()) == 0
    ref = cut[3].copy()
    ref[ref < -3.0] = -3.0
    ref = ref[:, 4:68, 4:68]
    assert np.array_equal(np.isnan(x.numpy()), np.isnan(ref)) and np.array_equal(np.nan_to_num(x.numpy()), np.nan_to_num(ref))
    assert torch.equal(rd, torch.tensor([ra[3], dec[3]]))
    dl = build_h5_dataloader(path, batch_size=4, num_workers=0, patch_size=16, num_channels=5, img_size=64, shuffle=False)
    xb, mb, rb = next(iter(dl))
    assert xb.shape == (4, 5, 64, 64) and rb.shape == (4, 2)
    ds2 = H5Dataset(path, img_size=64, patch_size=16, num_channels=5, max_mask_ratio=0.9)
    torch.manual_seed(0)
    _, m2, _ = ds2[0]
    assert m2.shape == (5, 64, 64) and set(m2.unique().tolist()) <= {0, 1}
    per_chan = m2.reshape(5, 4, 16, 4, 16)[:, :, 0, :, 0].reshape(5, -1).sum(1)
    assert len(set(per_chan.tolist())) == 1               # same count, different patches per channel
    # synthetic generator follows the schema
    p2 = hdf5_lite.make_synthetic_cutouts(str(tmp_path / "syn.h5"), n=16, nan_fraction=0.2, with_labels=True)
    with hdf5_lite.File(p2) as f:
        c = np.asarray(f["cutouts"])
        assert c.shape == (16, 5, 64, 64) and np.nanmin(c) >= -3.0 and np.isnan(c).any()
    with pytest.raises(hdf5_lite.H5LiteError):
        open(str(tmp_path / "bad.h5"), "wb").write(b"not hdf5 at all")
        hdf5_lite.File(str(tmp_path / "bad.h5"))


def test_product_refuses_cpu_and_never_imports_oracle():
    from sky_embeddings_amd.utils.mim_vit import build_model
    cfg = configparser.ConfigParser()
    cfg.read(os.path.join(ROOT, "configs", "mim_1.ini"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        build_model(cfg, "/nonexistent.pth.tar", torch.device("cpu"))
    # no product module imports the oracle package
    import re
    for base, _, files in os.walk(os.path.join(ROOT, "sky_embeddings_amd")):
        for fn in files:
            if fn.endswith(".py"):
                src = open(os.path.join(base, fn)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), fn
    for fn in ("pretrain_mim.py", "similarity_search.py", "sky_sim_search.py", "train_predictor.py", "test_predictor.py", "compare_predictors.py"):
        assert not re.search(r"^\s*(from|import)\s+oracle", open(os.path.join(ROOT, fn)).read(), flags=re.M)


def test_evaluation_numbers_of_the_predictor_figures():
    """utils/plotting_fns.py mirror (numbers only): photo-z metrics (:394-402) by hand, binned metrics, confusion matrix."""
    from sky_embeddings_amd.utils import plotting_fns as pf
    rng = np.random.default_rng(3)
    zt = rng.uniform(0.1, 1.8, 500)
    zp = zt + rng.normal(0, 0.03, 500) * (1 + zt)
    zp[:10] += 1.0                                        # outliers
    resid, bias, mad, frac = pf.photoz_prediction_metrics(zp, zt, threshold=0.15)
    r = (zp - zt) / (1 + zt)
    assert np.array_equal(resid, r) and bias == r.mean() and frac == (np.abs(r) > 0.15).sum() / 500
    assert mad == 1.4826 * np.median(np.abs(r - np.median(r))) and 0.02 < mad < 0.04 and frac >= 0.02
    res = pf.evaluate_z(zp, zt, n_bins=8, z_range=(0.2, 1.6), threshold=0.1, snr=rng.uniform(0, 30, 500))
    edges = np.linspace(0.2, 1.6, 9)
    b3 = np.where((edges[3] <= zt) & (zt < edges[4]))[0]
    assert res["z_bin_counts"][3] == len(b3) and np.isclose(res["z_bin_bias"][3], r[b3].mean())
    assert np.isclose(res["z_bin_frac_out"][3], (np.abs(r[b3]) > 0.1).mean()) and res["snr_bin_mids"][0] == 6.25
    empty = pf.evaluate_z(zp[:5], zt[:5], n_bins=50, z_range=(0.2, 1.6))
    assert np.isnan(empty["z_bin_bias"]).any()            # empty bins: NaN, not a crash
    t, p = np.array([0, 0, 1, 2, 2, 2.0]), np.array([0, 1, 1, 2, 0, 2])
    cm = pf.plot_conf_mat(t, p, ["galaxy", "qso", "star"], None)
    assert np.array_equal(cm, [[1, 1, 0], [0, 1, 0], [1, 0, 2]])


def test_cosine_schedule_and_ini_surface():
    from sky_embeddings_amd.optim import CosineLR

    class FakeOpt:
        def __init__(self):
            self.param_groups = [{"lr": 1e-3, "initial_lr": 1e-3}, {"lr": 1e-3, "initial_lr": 1e-3}]
            self.lr = 1e-3

        def set_lr(self, lr):
            self.lr = lr
    o = FakeOpt()
    s = CosineLR(o, 10, eta_min=1e-3 / 1e7)
    p = torch.nn.Parameter(torch.zeros(1))
    topt = torch.optim.AdamW([p], lr=1e-3)
    ts = torch.optim.lr_scheduler.CosineAnnealingLR(topt, 10, eta_min=1e-3 / 1e7)
    for _ in range(12):
        assert abs(o.lr - topt.param_groups[0]["lr"]) < 1e-15
        topt.step()
        ts.step()
        s.step()
    sd = s.state_dict()
    s2 = CosineLR(FakeOpt(), 10, eta_min=1e-10)
    s2.load_state_dict(sd)
    assert s2.last_epoch == 12 and abs(s2.optimizer.lr - o.lr) < 1e-18
    for name in ("mim_1", "mim_32"):
        cfg = configparser.ConfigParser()
        assert cfg.read(os.path.join(ROOT, "configs", name + ".ini"))
        for sec, keys in (("TRAINING", ["batch_size", "total_batch_iters", "mask_ratio", "norm_pix_loss", "weight_decay",
                                        "init_lr", "final_lr_factor", "loss_fn"]),
                          ("ARCHITECTURE", ["img_size", "num_channels", "pixel_mean", "pixel_std", "embed_dim", "patch_size",
                                            "model_type", "attn_pool", "ra_dec"]),
                          ("DATA", ["train_data_file", "val_data_file"])):
            for k in keys:
                assert k in cfg[sec], (name, sec, k)
        assert cfg["ARCHITECTURE"]["model_type"] in mc.MODEL_TYPES


# ------------------------------------------------------------------------------------ gloo, world_size 2
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    r, w, _ = sdist.init_from_env("gloo")
    assert (r, w) == (rank, world)
    from oracle import similarity_oracle as so
    # (1) DDP gradient averaging over the flat buffer in buckets == mean of per-rank gradients
    g = torch.Generator().manual_seed(100 + rank)
    n = 10_000 + 8
    grad = torch.randn(n, generator=g)
    mine = grad.clone()
    sdist.allreduce_flat_gradients(grad, world, bucket_elems=4096)
    other = torch.randn(n, generator=torch.Generator().manual_seed(100 + (1 - rank)))
    assert torch.allclose(grad * (1.0 / world), (mine + other) / 2, atol=1e-6)
    # (2) sharded bank: per-rank exact top-k on the shard + all-gather + merge == single-process top-k
    rng = np.random.default_rng(5)
    Q, N, D, k = 6, 1501, 48, 20
    q = rng.standard_normal((Q, D), dtype=np.float32)
    x = rng.standard_normal((N, D), dtype=np.float32)
    x[1400] = x[10]                                   # tie across shards -> lower global index first
    w8 = rng.random(D, dtype=np.float32) + 0.1
    lo, hi = sdist.shard_rows(N, rank, world)
    s_loc, i_loc = so.cosine_topk_np(q, x[lo:hi], k, w8)
    i_loc = np.where(i_loc >= 0, i_loc + lo, -1)
    gs, gi = sdist.gather_topk(torch.from_numpy(s_loc), torch.from_numpy(i_loc), world)
    assert gs.shape == (Q, world, k)
    cs, ci = gs.reshape(Q, -1).numpy(), gi.reshape(Q, -1).numpy()
    order = np.lexsort((ci, -cs), axis=1)[:, :k]
    ms, mi = np.take_along_axis(cs, order, 1), np.take_along_axis(ci, order, 1)
    rs, ri = so.cosine_topk_np(q, x, k, w8)
    assert np.array_equal(mi, ri) and np.array_equal(ms, rs)
    # (3) disjoint, equally sized index shards
    samp = sdist.DistributedIndexSampler(101, rank, world, shuffle=True, seed=3)
    mine_idx = torch.tensor(list(samp))
    both = [torch.zeros_like(mine_idx) for _ in range(world)]
    dist.all_gather(both, mine_idx)
    allidx = torch.cat(both)
    assert len(mine_idx) == 50 and len(set(allidx.tolist())) == 100
    # (4) store barrier: rank 1 arrives 0.5 s late, nobody leaves before it has arrived; twice (fresh key per use)
    import time
    for late in (1, 0):
        if rank == late:
            time.sleep(0.5)
        t0 = time.time()
        sdist.host_barrier(timeout_s=60.0)
        assert rank == late or time.time() - t0 >= 0.3
    dist.barrier()
    dist.destroy_process_group()
    open(os.path.join(tmp, f"ok{rank}"), "w").write("ok")


def test_two_process_gloo_paths(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert os.path.exists(tmp_path / "ok0") and os.path.exists(tmp_path / "ok1")
    assert sdist.bucket_bounds(20, 8) == [(0, 8), (8, 16), (16, 20)]
    assert sdist.shard_rows(10, 3, 4) == (9, 10) and sdist.shard_rows(10, 0, 4) == (0, 3)


def test_backward_stage_ranges_tile_the_gradient_buffer():
    """DDP overlap: the per-stage all-reduce slices cover the flat gradient buffer exactly once."""
    from sky_embeddings_amd.engine import ParamStore, stage_gradient_ranges
    for mt, groups in (("base", 3), ("tiny", 4), ("large", 5), ("base", 1)):
        cfg = mc.config_for(mt, img_size=64, patch_size=16, in_chans=5, embed_dim=192 if mt == "tiny" else 768)
        st = ParamStore(cfg, "cpu", torch.bfloat16)
        enc_groups, ranges = stage_gradient_ranges(st, cfg, groups)
        assert enc_groups[0][0] == cfg.depth and enc_groups[-1][1] == 0
        flat = sorted(r for stage in ranges for r in stage)
        assert flat[0][0] == 0 and flat[-1][1] == st.n
        for (a0, a1), (b0, b1) in zip(flat[:-1], flat[1:]):
            assert a1 == b0 and a0 < a1
        # a stage's slice holds exactly the weights of its own blocks; the top encoder stage also owns decoder_embed.weight, whose
        # gradient may ride in blocks.{depth-1}'s grouped weight-gradient launch (engine._extra_wgrad_layers)
        hi, lo = enc_groups[0]
        s, e = ranges[1][0]
        assert s == st.offsets[f"blocks.{lo}.attn.qkv.weight"]
        assert e == st.offsets["decoder_embed.weight"] + cfg.decoder_embed_dim * cfg.embed_dim == ranges[0][0][0]
        assert ranges[0][0][0] == st.offsets["decoder_blocks.0.attn.qkv.weight"]


def test_simmim_stage_ranges_put_the_head_weight_with_the_top_encoder_stage():
    """SimMIM without the pool: the head's weight gradient may be a problem of the last block's grouped launch
    (simmim_engine._extra_wgrad_layers): the head stage then finishes no decayed tensor; with the pool it keeps pool + head."""
    import dataclasses
    from sky_embeddings_amd.engine import ParamStore
    from sky_embeddings_amd.simmim_engine import simmim_stage_ranges
    base = mc.config_for("simmim", img_size=128, patch_size=16, in_chans=5, embed_dim=64, depth=4, num_heads=4)
    for cfg in (base, dataclasses.replace(base, attn_pool=True)):
        st = ParamStore(cfg, "cpu", torch.bfloat16)
        for groups in (1, 2, 4):
            enc_groups, ranges = simmim_stage_ranges(st, cfg, groups)
            flat = sorted(r for stage in ranges for r in stage)
            assert flat[0][0] == 0 and flat[-1][1] == st.n
            for (a0, a1), (b0, b1) in zip(flat[:-1], flat[1:]):
                assert a1 == b0 and a0 < a1
            o = st.offsets["decoder.0.weight"]
            owner = [k for k, stage in enumerate(ranges) for (s, e) in stage if s <= o < e]
            assert owner == [0 if cfg.attn_pool else 1]


def test_native_host_gather_rows():
    """skyemb_gather_rows_host (the feeder's minibatch gather; a HOST function of the C ABI) against numpy indexing."""
    import ctypes
    import numpy as np
    from sky_embeddings_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(0)
    src = rng.standard_normal((97, 5, 8, 8), dtype=np.float32)
    idx = rng.integers(0, 97, 40).astype(np.int64)
    for threads in (1, 3, 64):
        dst = np.full((40, 5, 8, 8), np.nan, np.float32)
        assert L.skyemb_gather_rows_host(src.ctypes.data, 5 * 8 * 8 * 4, idx.ctypes.data, 40, 97, dst.ctypes.data, threads) == 0
        assert np.array_equal(dst, src[idx])
    bad = np.array([0, 97], dtype=np.int64)
    assert L.skyemb_gather_rows_host(src.ctypes.data, 1280, bad.ctypes.data, 2, 97, dst.ctypes.data, 2) != 0
    assert b"out of range" in L.skyemb_last_error()


def test_optimizer_state_dict_interchanges_with_torch_adamw_in_simmim_mode():
    """in SimMIM mode the reference optimises ``mask_token`` (requires_grad, decay group) although no forward
    uses it, so torch's AdamW state dict holds its id without a state entry.  FusedAdamW must write and read exactly that
    layout (ids per group, which ids carry state), in both directions."""
    from types import SimpleNamespace
    from sky_embeddings_amd.engine import ParamStore
    from sky_embeddings_amd.optim import FusedAdamW
    for kind in ("simmim", "mae"):
        cfg = mc.MAEConfig(img_size=32, patch_size=8, in_chans=2, embed_dim=16, depth=1, num_heads=2, decoder_embed_dim=8,
                           decoder_depth=1, decoder_num_heads=2, simmim=(kind == "simmim"))
        # the reference module's parameters, by name, in registration (= state-dict) order
        params = {n: torch.nn.Parameter(torch.randn(s) * 0.1, requires_grad=n not in mc.FROZEN) for n, s in mc.state_layout(cfg)}
        no_decay = [p for n, p in params.items() if p.requires_grad and (p.ndim <= 1 or n.endswith(".bias"))]
        decay = [p for n, p in params.items() if p.requires_grad and not (p.ndim <= 1 or n.endswith(".bias"))]
        topt = torch.optim.AdamW([{"params": no_decay, "weight_decay": 0.0}, {"params": decay, "weight_decay": 0.05}],
                                 lr=1e-3, betas=(0.9, 0.95))
        unused = {"mask_token"} if cfg.simmim else set()
        g = torch.Generator().manual_seed(0)
        for n, p in params.items():
            if p.requires_grad and n not in unused:
                p.grad = torch.randn(p.shape, generator=g)
        topt.step()
        tsd = topt.state_dict()
        store = ParamStore(cfg, "cpu", torch.float32)
        fopt = FusedAdamW(SimpleNamespace(store=store), lr=1e-3, weight_decay=0.05)
        fopt.load_state_dict(tsd)                                   # torch -> ours
        assert fopt.step_count == 1
        for n in store.order:
            pid = [id(q) for q in no_decay + decay].index(id(params[n]))
            assert torch.equal(store._view(store.m, n), tsd["state"][pid]["exp_avg"]), n
            assert torch.equal(store._view(store.v, n), tsd["state"][pid]["exp_avg_sq"]), n
        fsd = fopt.state_dict()                                     # ours -> torch
        assert [g_["params"] for g_ in fsd["param_groups"]] == [g_["params"] for g_ in tsd["param_groups"]]
        assert sorted(fsd["state"]) == sorted(tsd["state"])
        topt2 = torch.optim.AdamW([{"params": no_decay, "weight_decay": 0.0}, {"params": decay, "weight_decay": 0.05}], lr=1e-3)
        topt2.load_state_dict(fsd)
        for pid, s in tsd["state"].items():
            assert torch.equal(topt2.state_dict()["state"][pid]["exp_avg"], s["exp_avg"])
        if cfg.simmim:
            mt = [id(q) for q in no_decay + decay].index(id(params["mask_token"]))
            assert mt not in fsd["state"] and mt in fsd["param_groups"][1]["params"]


def test_feeder_batch_count_is_equal_on_every_rank():
    """n % world != 0 must not give low ranks one more step per epoch (the collectives would pair up across
    epoch boundaries).  Pure host arithmetic of CutoutFeeder's shard / batch count."""
    import inspect
    from sky_embeddings_amd import feeder
    src = inspect.getsource(feeder.CutoutFeeder.__init__)
    assert "batches_per_epoch(len(self.indices), self.B, self.rank, self.world, drop_last)" in src
    for n, world, B in ((1030, 4, 8), (257, 2, 128), (1000, 8, 125), (7, 8, 1)):
        per_rank = [feeder.batches_per_epoch(n, B, r, world) for r in range(world)]
        longest = len(np.arange(n)[0::world])
        assert len(set(per_rank)) == 1 and per_rank[0] * B <= n // world <= longest


def test_hdf5_lite_chunked_resizable_datasets(tmp_path, monkeypatch):
    """The layout the reference's ETL writes (create_dataset(..., maxshape=(None, ...)) + resize: chunked, v1 chunk B-tree,
    h5py's automatic chunk shapes): ragged edge chunks, one-, two- and three-level B-trees, never-written datasets,
    indexed reads from the chunks, the one-time un-chunk into the contiguous cache (native threads) and its invalidation."""
    monkeypatch.setenv("SKYEMB_H5_CACHE", str(tmp_path / "cache"))
    (tmp_path / "cache").mkdir()
    assert hdf5_lite.h5py_guess_chunk((0, 5, 64, 64), 4) == (128, 1, 8, 16)       # h5py filters.guess_chunk by hand
    assert hdf5_lite.h5py_guess_chunk((0,), 4) == (1024,)
    rng = np.random.default_rng(0)
    cut = rng.standard_normal((301, 5, 64, 64), dtype=np.float32)                 # 301 rows: ragged last chunk block
    cut[3, 2] = np.nan
    ra = rng.random(301).astype(np.float32)
    big = np.arange(70001, dtype=np.int32)                                        # 4376 chunks of 16: a three-level B-tree
    odd = rng.standard_normal((37, 7, 9)).astype(np.float64)                      # chunks that divide no axis
    empty = np.zeros((0, 5, 8, 8), np.float32)
    path = str(tmp_path / "chunked.h5")
    hdf5_lite.write_datasets(path, {"cutouts": cut, "ra": ra, "big": big, "odd": odd, "empty": empty, "dec": ra[::-1].copy()},
                             chunks={"cutouts": hdf5_lite.h5py_guess_chunk((0, 5, 64, 64), 4), "ra": (1024,), "big": (16,),
                                     "odd": (5, 4, 4), "empty": (16, 1, 8, 8)})
    with hdf5_lite.File(path) as f:
        ds = f["cutouts"]
        assert ds.shape == cut.shape and ds.chunks == (128, 1, 8, 16) and f["dec"].chunks is None
        addr, off = ds.chunk_table()
        assert len(addr) == 3 * 5 * 8 * 4 and len(set(addr.tolist())) == len(addr)
        # indexed reads straight from the chunks (no cache yet)
        assert np.array_equal(ds._read_chunked(7), cut[7], equal_nan=True)
        assert np.array_equal(ds._read_chunked([300, 0, 129, 128]), cut[[300, 0, 129, 128]], equal_nan=True)
        assert np.array_equal(ds._read_chunked(slice(120, 140, 3)), cut[120:140:3], equal_nan=True)
        assert np.array_equal(ds._read_chunked(-1), cut[-1])
        with pytest.raises(IndexError):
            ds._read_chunked(301)
        assert np.array_equal(np.asarray(f["big"]), big) and np.array_equal(f["big"][[70000, 5, 4095, 4096]], big[[70000, 5, 4095, 4096]])
        assert np.array_equal(np.asarray(f["odd"]), odd) and np.array_equal(f["odd"][11], odd[11])
        assert f["empty"].shape == (0, 5, 8, 8) and np.asarray(f["empty"]).size == 0
        assert np.array_equal(np.asarray(f["ra"]), ra) and np.array_equal(f["ra"][5:9], ra[5:9])
        # the feeder's view: un-chunked once into the cache, then a plain memmap
        arr = ds._array()
        assert isinstance(arr, np.memmap) and np.array_equal(arr, cut, equal_nan=True)
        assert np.array_equal(ds[17], cut[17]) and np.array_equal(ds[[4, 3]], cut[[4, 3]], equal_nan=True)
    cache = [p for p in os.listdir(tmp_path / "cache") if p.endswith(".contig")]
    assert len(cache) >= 1
    # a second open reuses the cache; rewriting the source invalidates it
    with hdf5_lite.File(path) as f:
        assert np.array_equal(f["cutouts"]._array(), cut, equal_nan=True)
    cut2 = cut + 1.0
    hdf5_lite.write_datasets(path, {"cutouts": cut2}, chunks={"cutouts": (128, 1, 8, 16)})
    with hdf5_lite.File(path) as f:
        assert np.array_equal(f["cutouts"]._array(), cut2, equal_nan=True)
    # synthetic generator in the chunked flavour + the per-item dataset semantics on top of it
    from sky_embeddings_amd.utils.dataloaders import H5Dataset
    syn = hdf5_lite.make_synthetic_cutouts(str(tmp_path / "syn_chunked.h5"), n=40, seed=3, chunked=True)
    ref = hdf5_lite.make_synthetic_cutouts(str(tmp_path / "syn_contig.h5"), n=40, seed=3)
    a, b = H5Dataset(syn, 64, 16, 5, None), H5Dataset(ref, 64, 16, 5, None)
    for i in (0, 13, 39):
        xa, xb = a[i], b[i]
        assert torch.equal(xa[0], xb[0]) and torch.equal(xa[2], xb[2])


def test_select_centre_picks_the_central_block():
    """utils/misc.py:68-117: central sqrt(n) x sqrt(n) block of the raster-ordered patch grid (hand-checked cases)."""
    from sky_embeddings_amd.utils.misc import central_indices, select_centre
    lat = np.arange(2 * 64 * 3).reshape(2, 64, 3)                      # 8 x 8 grid
    got = select_centre(lat, 4)
    assert got.shape == (2, 4, 3) and np.array_equal(got[0, :, 0] // 3, [27, 28, 35, 36])      # rows 3-4, cols 3-4
    assert np.array_equal(select_centre(lat, 16)[1, :, 0] // 3 - 64, [r * 8 + c for r in range(2, 6) for c in range(2, 6)])
    assert np.array_equal(select_centre(np.arange(16).reshape(1, 16, 1), 4)[0, :, 0], [5, 6, 9, 10])   # 4 x 4 grid
    assert np.array_equal(central_indices(np.empty((5, 5)), 1), [[2, 2]])
    with pytest.raises(ValueError):
        central_indices(np.empty((8, 8)), 3)


_WCS_HDR = {"CTYPE1": "RA---TAN-SIP", "CTYPE2": "DEC--TAN-SIP", "CRPIX1": 120.5, "CRPIX2": 131.0, "CRVAL1": 150.1, "CRVAL2": 2.2,
            "CD1_1": -4.66e-5, "CD1_2": 1.0e-7, "CD2_1": -2.0e-7, "CD2_2": 4.66e-5, "A_ORDER": 2, "B_ORDER": 2, "A_2_0": 1.5e-7,
            "A_1_1": -3e-8, "B_0_2": 2e-7, "B_1_1": 5e-8}


def test_fits_lite_round_trip_and_world_coordinates(tmp_path):
    """fits_lite (the astropy-free access layer of the survey-tile path, utils/dataloaders.py:417-433): image HDU 1 of a file
    it wrote comes back bit for bit (float32 with NaNs; int16 with BSCALE / BZERO), header values keep their types, the raw
    view is big-endian; TAN-SIP pixel -> sky agrees with the oracle's independent formulation and with the obvious cases."""
    from oracle import tile_oracle as to
    from sky_embeddings_amd import fits_lite
    rng = np.random.default_rng(0)
    img = rng.standard_normal((37, 53)).astype(np.float32)
    img[3, 4] = np.nan
    p = fits_lite.write_image_fits(str(tmp_path / "calexp-HSC-G-9813-4,3.fits"), img, _WCS_HDR)
    h = fits_lite.read_image_hdu(p, 1)
    assert h.shape == (37, 53) and h.raw.dtype == np.dtype(">f4") and h.bitpix == -32
    assert np.array_equal(h.array().view(np.uint32), img.view(np.uint32))
    assert h.header["CTYPE1"] == "RA---TAN-SIP" and h.header["A_ORDER"] == 2 and abs(h.header["CD1_1"] + 4.66e-5) < 1e-18
    ints = rng.integers(-3000, 3000, (9, 11)).astype(np.int16)
    p2 = fits_lite.write_image_fits(str(tmp_path / "i.fits"), ints, {"BSCALE": 0.5, "BZERO": 100.0}, bitpix=16)
    assert np.array_equal(fits_lite.read_image_hdu(p2, 1).array(), ints * 0.5 + 100.0)
    with pytest.raises(IndexError):
        fits_lite.read_image_hdu(p, 3)
    w = fits_lite.TanSipWCS(h.header)
    x, y = np.array([0, 100.5, 119.5, 4000, 17]), np.array([0, 3000, 130.0, 4100, 999.25])
    ra, dec = w.all_pix2world(x, y, 0)
    ra2, dec2 = to.tan_sip_pix2world(_WCS_HDR, x, y, 0)
    assert np.abs(ra - ra2).max() < 1e-11 and np.abs(dec - dec2).max() < 1e-11
    assert abs(ra[2] - 150.1) < 1e-12 and abs(dec[2] - 2.2) < 1e-12                      # the reference pixel
    plain = dict(_WCS_HDR, CTYPE1="RA---TAN", CTYPE2="DEC--TAN", CD1_2=0.0, CD2_1=0.0)
    r, d = fits_lite.TanSipWCS(plain).all_pix2world(np.array([119.5, 119.5, 120.5]), np.array([130.0, 131.0, 130.0]), 0)
    assert abs((d[1] - d[0]) - 4.66e-5) < 1e-9                                           # one row up = one pixel scale north
    assert abs((r[2] - r[0]) * np.cos(np.deg2rad(2.2)) + 4.66e-5) < 1e-9                 # one column right = one pixel scale west
    with pytest.raises(NotImplementedError):
        fits_lite.TanSipWCS(dict(plain, CTYPE1="RA---SIN"))


def test_tile_file_discovery_and_overlap_grid(tmp_path):
    """find_HSC_bands (utils/dataloaders.py:330-379) and generate_overlap_coords (:478-505), hand-checked."""
    from sky_embeddings_amd.utils.dataloaders import find_HSC_bands, generate_overlap_coords
    for name in ("calexp-HSC-G-9813-4,3.fits", "calexp-HSC-R-9813-4,3.fits", "calexp-HSC-I-9813-4,3.fits", "calexp-HSC-G-9813-5,3.fits",
                 "HSC-G-9813-4,3.fits", "calexp-HSC-Q-9813-4,3.fits", "notes.txt", "ab.fits"):
        (tmp_path / name).write_bytes(b"")
    got = find_HSC_bands([str(tmp_path)], ["G", "R", "I"], min_bands=2, verbose=0)
    assert len(got) == 1 and [os.path.basename(f) for f in got[0]] == ["calexp-HSC-G-9813-4,3.fits", "calexp-HSC-R-9813-4,3.fits",
                                                                        "calexp-HSC-I-9813-4,3.fits"]
    got = find_HSC_bands([str(tmp_path)], ["G", "Y"], min_bands=1, verbose=0)
    assert sorted(g[1] for g in got) == ["None", "None"] and len(got) == 2
    assert [os.path.basename(g[0]) for g in find_HSC_bands([str(tmp_path)], ["G"], 1, 0, use_calexp=False)] == ["HSC-G-9813-4,3.fits"]
    c = generate_overlap_coords((10, 11), 4, 0.5)                  # step 2: rows 0..6, cols 0..6; W % 2 != 0 -> a right-edge column
    assert c[:4] == [(0, 0), (0, 2), (0, 4), (0, 6)] and (6, 7) in c and (0, 7) in c and len(c) == 16 + 4
    assert generate_overlap_coords((8, 8), 4, 0.0) == [(0, 0), (0, 4), (4, 0), (4, 4)]


def test_fits_lite_header_grammar_against_a_third_party_file():
    """The header-card grammar and the HDU walk (the parts of fits_lite every image read rests on) against a FITS file this repo
    did not write: numpy ships `recarray_from_file.fits` (STScI STSDAS TABLES, 2001: an empty primary HDU + one BINTABLE) in its
    test data.  Skipped where that file is absent."""
    import numpy
    from sky_embeddings_amd import fits_lite
    path = os.path.join(os.path.dirname(numpy.__file__), "_core", "tests", "data", "recarray_from_file.fits")
    if not os.path.exists(path):
        pytest.skip("numpy's test data is not installed")
    buf = open(path, "rb").read()
    assert len(buf) == 3 * fits_lite.BLOCK
    h0, pos = fits_lite._read_header(buf, 0)
    assert pos == fits_lite.BLOCK
    assert h0["SIMPLE"] is True and h0["BITPIX"] == 16 and h0["NAXIS"] == 0 and h0["EXTEND"] is True and h0["NEXTEND"] == 1
    assert h0["ORIGIN"] == "STScI-STSDAS/TABLES" and h0["FILENAME"] == "tb.fits"      # quoted strings, trailing blanks dropped
    assert "COMMENT" not in h0 and "HISTORY" not in h0
    assert fits_lite._data_bytes(h0) == 0                                             # NAXIS = 0: the extension follows at once
    h1, pos1 = fits_lite._read_header(buf, pos)
    assert pos1 == 2 * fits_lite.BLOCK
    assert h1["XTENSION"] == "BINTABLE" and (h1["BITPIX"], h1["NAXIS"], h1["NAXIS1"], h1["NAXIS2"]) == (8, 2, 17, 3)
    assert (h1["PCOUNT"], h1["GCOUNT"], h1["TFIELDS"]) == (0, 1, 3)
    assert [h1[f"TFORM{i}"] for i in (1, 2, 3)] == ["1D", "1J", "5A"] and [h1[f"TTYPE{i}"] for i in (1, 2, 3)] == ["a", "b", "c"]
    assert h1["TNULL2"] == -2147483647 and h1["TDISP1"] == "G25.16"
    n = fits_lite._data_bytes(h1)
    assert n == 17 * 3 and pos1 + -(-n // fits_lite.BLOCK) * fits_lite.BLOCK == len(buf)   # data padded to one block: end of file
    # the table rows decode with the formats the header names (big-endian 1D, 1J, 5A): what numpy's own test reads from this file
    rows = np.frombuffer(buf, dtype=np.dtype([("a", ">f8"), ("b", ">i4"), ("c", "S5")]), count=3, offset=pos1)
    assert np.all(np.isfinite(rows["a"])) and rows["c"].dtype.itemsize == 5
    with pytest.raises(NotImplementedError):                                           # a table is not an image HDU
        fits_lite.read_image_hdu(path, hdu=1)


def test_fits_lite_reads_lossless_tile_compressed_images(tmp_path):
    """ZIMAGE binary tables with the lossless gzip codecs (FITS 4.0 section 10; what fpack -g / the LSST stack's GZIP_SHUFFLE
    write) decode to the same pixels and keep the image header; codecs outside the subset are refused by name (Rice and quantised
    floats: tests/test_io_fixtures_cpu.py, against astropy-written files)."""
    from sky_embeddings_amd import fits_lite
    rng = np.random.default_rng(1)
    img = rng.standard_normal((37, 53)).astype(np.float32)
    img[2, 3] = np.nan
    for codec in ("GZIP_1", "GZIP_2"):
        for tile_rows in (1, 5):
            p = fits_lite.write_compressed_image_fits(str(tmp_path / f"{codec}_{tile_rows}.fits"), img, _WCS_HDR, codec, tile_rows)
            h = fits_lite.read_image_hdu(p, 1)
            assert h.shape == (37, 53) and h.bitpix == -32 and h.raw.dtype == np.dtype(">f4")
            assert np.array_equal(h.array().view(np.uint32), img.view(np.uint32)) and h.header["CTYPE1"] == "RA---TAN-SIP"
    ints = rng.integers(-5000, 5000, (20, 31)).astype(np.int32)
    p = fits_lite.write_compressed_image_fits(str(tmp_path / "i.fits"), ints, None, "GZIP_2", 4)
    assert np.array_equal(fits_lite.read_image_hdu(p, 1).array(), ints)
    raw = bytearray(open(p, "rb").read())
    at = raw.index(b"GZIP_2")
    raw[at:at + 6] = b"LZMA_1"
    (tmp_path / "r.fits").write_bytes(bytes(raw))
    with pytest.raises(NotImplementedError, match="LZMA_1"):
        fits_lite.read_image_hdu(str(tmp_path / "r.fits"), 1)


def test_host_cutout_functions_match_oracle(tmp_path):
    """random_cutouts / overlapping_cutouts / load_fits_bands (host mirrors of utils/dataloaders.py:381-536) against the
    restatement in oracle/tile_oracle.py for the same numpy draws; RA / Dec with the reference's (row, column) -> (x, y) order."""
    from oracle import tile_oracle as to
    from sky_embeddings_amd import fits_lite
    from sky_embeddings_amd.utils.dataloaders import generate_overlap_coords, load_fits_bands, overlapping_cutouts, random_cutouts
    rng = np.random.default_rng(3)
    bands = []
    for b in "GRI":
        img = rng.standard_normal((90, 120)).astype(np.float32)
        fits_lite.write_image_fits(str(tmp_path / f"calexp-HSC-{b}-1-2,3.fits"), img, _WCS_HDR)
        bands.append(img)
    tile, pix_to_radec = load_fits_bands([str(tmp_path / f"calexp-HSC-{b}-1-2,3.fits") for b in "GR"] + ["None", str(tmp_path / "calexp-HSC-I-1-2,3.fits")],
                                         return_wc=True)
    assert tile.shape == (4, 90, 120) and np.isnan(tile[2]).all() and np.array_equal(tile[3].astype(np.float32), bands[2])
    np.random.seed(9)
    cut, rd = random_cutouts(tile, 32, 11, pix_to_radec)
    np.random.seed(9)
    hs, ws = np.random.randint(0, 90 - 32 + 1, size=11), np.random.randint(0, 120 - 32 + 1, size=11)
    ref = to.cutouts_np(tile, hs, ws, 32)
    assert np.array_equal(np.nan_to_num(cut.astype(np.float32), nan=-7), np.nan_to_num(ref, nan=-7))
    ra, dec = to.tan_sip_pix2world(_WCS_HDR, hs + 16, ws + 16, 0)
    assert np.allclose(rd, np.vstack((ra, dec)).T, rtol=0, atol=1e-10)
    over = overlapping_cutouts(tile, 32, 0.5)
    coords = generate_overlap_coords((90, 120), 32, 0.5)
    assert over.shape == (len(coords), 4, 32, 32) and np.array_equal(np.nan_to_num(over[-1]), np.nan_to_num(tile[:, 58:90, 88:120]))


def _open_chunked_worker(args):
    path, cache = args
    os.environ["SKYEMB_H5_CACHE"] = cache
    from sky_embeddings_amd import hdf5_lite as h5
    with h5.File(path) as f:
        arr = f["cutouts"]._array()
        return float(np.nansum(arr[::7])), sorted(os.listdir(cache))


def test_hdf5_lite_unchunk_cache_is_built_once_by_concurrent_openers(tmp_path):
    """Every rank's feeder and every loader worker hits a chunked file at the same moment: the contiguous copy is built under
    a lock by one of them (no per-process temp copies left behind), the others wait and map the finished file."""
    import multiprocessing as mp
    cache = tmp_path / "cache"
    cache.mkdir()
    rng = np.random.default_rng(5)
    cut = rng.standard_normal((260, 5, 16, 16), dtype=np.float32)
    path = str(tmp_path / "c.h5")
    hdf5_lite.write_datasets(path, {"cutouts": cut}, chunks={"cutouts": (64, 1, 8, 16)})
    with mp.get_context("spawn").Pool(4) as pool:
        res = pool.map(_open_chunked_worker, [(path, str(cache))] * 4)
    want = float(np.nansum(cut[::7]))
    assert all(abs(r[0] - want) <= 1e-3 * abs(want) for r in res)
    left = sorted(os.listdir(cache))
    assert not [p for p in left if p.endswith(".tmp")], left
    assert [p for p in left if p.endswith(".contig")] and [p for p in left if p.endswith(".contig.json")]


def _worker8(rank, world, port, tmp):
    """The host-side logic of the N = 8 splits BASELINE configs[2] / [3] name, over gloo with eight CPU processes (no node with eight
    GPUs is available to the build, and a GPU box admits at most six GPU processes)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    r, w, _ = sdist.init_from_env("gloo")
    assert (r, w) == (rank, world) and world == 8
    from oracle import similarity_oracle as so
    # (1) bucketed all-reduce of the flat gradient buffer over eight ranks, fp32 and bf16 (the default communication dtype): the bf16
    # ring sum stays within 8 bf16 roundings of the fp32 mean
    n = 20_000 + 8
    grads = [torch.randn(n, generator=torch.Generator().manual_seed(200 + k)) for k in range(world)]
    g32 = grads[rank].clone()
    sdist.allreduce_flat_gradients(g32, world, bucket_elems=4096)
    want = torch.stack(grads).double().sum(0)
    assert torch.allclose(g32.double(), want, atol=1e-5)
    g16 = grads[rank].bfloat16()
    sdist.allreduce_flat_gradients(g16, world, bucket_elems=4096)
    rel = ((g16.double() - want).abs() / (torch.stack(grads).abs().double().sum(0) + 1e-30)).max()
    assert float(rel) < 8 * 2.0 ** -8, float(rel)
    every = [torch.zeros_like(g16) for _ in range(world)]
    dist.all_gather(every, g16)
    assert all(torch.equal(every[0], e) for e in every)                       # replicas see the same sum, bit for bit
    # (2) bank sharded eight ways with ragged last shards, ties planted in THREE different shards: gather + merge == one bank
    rng = np.random.default_rng(8)
    Q, N, D, k = 9, 12_345, 64, 25
    q = rng.standard_normal((Q, D), dtype=np.float32)
    x = rng.standard_normal((N, D), dtype=np.float32)
    w8 = rng.random(D, dtype=np.float32) + 0.1
    best = int(np.argmax(so.cosine_scores_np(q[:1], x, w8)[0]))
    for row in (37, 5_000, 12_300):                                          # shards 0, 3 and 7 hold a copy of query 0's best row
        x[row] = x[best]
    lo, hi = sdist.shard_rows(N, rank, world)
    assert hi - lo == (1544 if rank < 7 else N - 7 * 1544)
    s_loc, i_loc = so.cosine_topk_np(q, x[lo:hi], k, w8)
    i_loc = np.where(i_loc >= 0, i_loc + lo, -1)
    gs, gi = sdist.gather_topk(torch.from_numpy(s_loc), torch.from_numpy(i_loc), world)
    assert gs.shape == (Q, world, k)
    cs, ci = gs.reshape(Q, -1).numpy(), gi.reshape(Q, -1).numpy()
    order = np.lexsort((ci, -cs), axis=1)[:, :k]
    ms, mi = np.take_along_axis(cs, order, 1), np.take_along_axis(ci, order, 1)
    rs, ri = so.cosine_topk_np(q, x, k, w8)
    assert np.array_equal(mi, ri) and np.array_equal(ms, rs)
    tied = sorted({37, 5_000, 12_300, best})
    assert ri[0, :len(tied)].tolist() == tied                                # equal scores: ascending global index across shards
    # (3) index shards with a remainder: 8 x 12 of 101 samples, disjoint, a fresh permutation per epoch, the same 96 on every rank's view
    samp = sdist.DistributedIndexSampler(101, rank, world, shuffle=True, seed=3)
    seen = []
    for epoch in range(2):
        samp.set_epoch(epoch)
        mine_idx = torch.tensor(list(samp))
        allr = [torch.zeros_like(mine_idx) for _ in range(world)]
        dist.all_gather(allr, mine_idx)
        flat = torch.cat(allr).tolist()
        assert len(mine_idx) == len(samp) == 12 and len(set(flat)) == 96 and max(flat) < 101
        seen.append(flat)
    assert seen[0] != seen[1]
    # (4) the feeder's batch count is the same on every rank (the step's collectives would deadlock otherwise)
    from sky_embeddings_amd.feeder import batches_per_epoch
    counts = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(counts, torch.tensor([batches_per_epoch(1001, 16, rank, world)]))
    assert len({int(c) for c in counts}) == 1 and int(counts[0]) == 1001 // world // 16
    # (5) the sharded optimiser's collectives (TrainStep(shard_optimizer=True)): every stage range cut into 8 owner chunks --
    # reduce-scatter of the gradients (+ the replicated tail of < 64 elements), an "optimiser step" on the owned chunk only, the
    # in-place all-gather of the result -- equals all-reduce + the step on everything, for ragged range lengths and both dtypes
    for dt, tol in ((torch.float32, 1e-5), (torch.bfloat16, 8 * 2.0 ** -8)):
        ranges = [(0, 4096), (4096, 4096 + 8 * 1237), (4096 + 8 * 1237, n)]      # chunk 512 / 1232 (+ tail 40) / the rest
        gr = grads[rank].clone().to(dt)
        param = torch.arange(n, dtype=torch.float32) * 1e-3
        for (s_, e_) in ranges:
            c = sdist.shard_chunk(e_ - s_, world)
            own = torch.zeros(max(c, 8), dtype=dt)
            for wk in sdist.reduce_scatter_range(gr, s_, e_, rank, world, own):
                wk.wait()
            ref_chunk = want[s_ + rank * c:s_ + (rank + 1) * c]
            scale = torch.stack(grads).abs().double().sum(0)[s_ + rank * c:s_ + (rank + 1) * c] + 1e-30
            assert float(((own[:c].double() - ref_chunk).abs() / scale).max()) < tol
            tail = slice(s_ + world * c, e_)
            if tail.start < tail.stop:
                assert float(((gr[tail].double() - want[tail]).abs() / (torch.stack(grads).abs().double().sum(0)[tail] + 1e-30)).max()) < tol
            # "step": p -= g on the owned chunk and on the tail, then gather
            param[s_ + rank * c:s_ + (rank + 1) * c] -= own[:c].float()
            param[tail] -= gr[tail].float()
            wk = sdist.all_gather_range(param, s_, e_, rank, world)
            if wk is not None:
                wk.wait()
        every = [torch.zeros_like(param) for _ in range(world)]
        dist.all_gather(every, param)
        assert all(torch.equal(every[0], e) for e in every)                   # every rank holds the same parameters afterwards
        assert float((param.double() - (torch.arange(n, dtype=torch.float64) * 1e-3 - want)).abs().max()) < (1e-4 if dt == torch.float32 else 0.2)
    # (6) a wall-clock decision taken together: every rank gets rank 0's flag
    assert sdist.agree(rank == 0) is True and sdist.agree(rank != 0) is False
    sdist.host_barrier(timeout_s=60.0)
    dist.barrier()
    dist.destroy_process_group()
    open(os.path.join(tmp, f"ok{rank}"), "w").write("ok")


def test_eight_process_gloo_paths(tmp_path):
    """World size 8 -- the split BASELINE.json names -- over gloo on CPU: all-reduce (fp32 / bf16), sharded search with ties in three
    shards, index shards with a remainder, the feeder's batch count, the sharded optimiser's reduce-scatter / all-gather over ragged
    ranges, the store barrier and the store agreement."""
    port = _free_port()
    mp.spawn(_worker8, args=(8, port, str(tmp_path)), nprocs=8, join=True)
    assert all(os.path.exists(tmp_path / f"ok{k}") for k in range(8))


def test_predictor_schedule_and_layer_decay_groups_match_torch_and_the_reference_recipe():
    """Host logic of the downstream predictor's optimiser (utils/vit.py:134-186): LinearLR in closed form == torch's, and the
    layer-wise lr-decay grouping of utils/lr_decay.py on a model's tensor names."""
    from sky_embeddings_amd.utils.lr_decay import get_layer_id_for_vit, param_groups_lrd
    from sky_embeddings_amd.utils.vit import LinearLR

    class Opt:
        def __init__(self, lrs):
            self.param_groups = [{"lr": v, "initial_lr": v} for v in lrs]
    mine = Opt([1e-3, 2.5e-4])
    sched = LinearLR(mine, start_factor=1.0, end_factor=0.01, total_iters=7)
    prm = [torch.nn.Parameter(torch.zeros(1)), torch.nn.Parameter(torch.zeros(1))]
    ref_opt = torch.optim.AdamW([{"params": [prm[0]], "lr": 1e-3}, {"params": [prm[1]], "lr": 2.5e-4}])
    ref = torch.optim.lr_scheduler.LinearLR(ref_opt, start_factor=1.0, end_factor=0.01, total_iters=7)
    for _ in range(10):                                  # past total_iters: the factor stays at end_factor
        assert np.allclose(sched.get_last_lr(), ref.get_last_lr(), rtol=1e-12)
        ref_opt.step()
        ref.step()
        sched.step()

    class Model:
        num_blocks = 3

        def trainable_tensors(self):
            return [("cls_token", 3), ("patch_mask_values", 3), ("patch_embed.proj.weight", 4), ("patch_embed.proj.bias", 1),
                    ("blocks.0.attn.qkv.weight", 2), ("blocks.0.norm1.weight", 1), ("blocks.2.mlp.fc2.weight", 2),
                    ("norm.weight", 1), ("head.weight", 2), ("head.bias", 1)]
    groups, lrs = param_groups_lrd(Model(), 0.1, weight_decay=0.05, no_weight_decay_list={"pos_embed", "cls_token"}, layer_decay=0.5)
    by_name = {n: (g["lr"], g["weight_decay"]) for g in groups for n in g["params"]}
    assert get_layer_id_for_vit("blocks.2.mlp.fc2.weight", 4) == 3 and get_layer_id_for_vit("head.bias", 4) == 4
    assert by_name["cls_token"] == (0.1 * 0.5 ** 4, 0.0)                 # layer 0, listed as not decayed although 3-D
    assert by_name["patch_embed.proj.weight"] == (0.1 * 0.5 ** 4, 0.05) and by_name["patch_embed.proj.bias"] == (0.1 * 0.5 ** 4, 0.0)
    assert by_name["blocks.0.attn.qkv.weight"] == (0.1 * 0.5 ** 3, 0.05) and by_name["blocks.2.mlp.fc2.weight"] == (0.1 * 0.5, 0.05)
    assert by_name["patch_mask_values"] == (0.1, 0.05) and by_name["norm.weight"] == (0.1, 0.0) and by_name["head.weight"] == (0.1, 0.05)
    assert lrs == [g["lr"] for g in groups] and len(groups) == len({(round(l, 12), w) for l, w in by_name.values()})
