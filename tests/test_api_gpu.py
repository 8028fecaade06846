"""GPU: the drop-in module API (utils.mim_vit / utils.similarity / entry points) end to end."""
import configparser
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def test_similarity_module_matches_reference_goldens():
    """utils.similarity.compute_similarity (cosine through the HIP kernel) vs vectors captured from the
    reference; cosine within fp32 rounding of torch's unspecified summation order, MSE/MAE exact."""
    from sky_embeddings_amd.utils import similarity as sim
    z = np.load(os.path.join(GOLDEN, "similarity.npz"))
    for (T, P, N) in ((130, 1, 512), (65, 16, 128), (65, 64, 64)):
        key = f"sim/{T}_{P}_{N}"
        tgt, tst = torch.from_numpy(z[key + "/target"]).cuda(), torch.from_numpy(z[key + "/test"]).cuda()
        avg, w = sim.determine_target_features(tgt)
        assert np.allclose(avg.cpu().numpy(), z[key + "/avg"], rtol=1e-5, atol=1e-6)
        for metric in ("cosine", "MSE", "MAE"):
            for combine in ("min", "mean", "max"):
                for uw in (True, False):
                    s = sim.compute_similarity(tgt, tst, metric=metric, combine=combine, use_weights=uw).cpu().numpy()
                    ref = z[f"{key}/{metric}_{combine}_{int(uw)}"]
                    assert np.abs(s - ref).max() <= 2e-6 * max(1.0, np.abs(ref).max()), (key, metric, combine, uw)
    # streaming best-n (scores + which samples) == the reference's streaming result
    scores = torch.from_numpy(z["stream/scores"]).cuda()
    for metric in ("cosine", "MSE"):
        bs = torch.full((50,), float("-inf") if metric == "cosine" else float("inf"), device="cuda")
        bx, brd = torch.empty(50, 1, device="cuda"), torch.empty(50, 2, device="cuda")
        for b in range(8):
            idx = torch.arange(b * 64, (b + 1) * 64, dtype=torch.float32, device="cuda")
            bx, brd, bs = sim.update_best_scores(idx[:, None], torch.stack([idx, idx], 1), scores[b * 64:(b + 1) * 64], bx,
                                                 brd, bs, 50, metric)
        assert np.array_equal(bs.cpu().numpy(), z[f"stream/{metric}_best_scores"])
        assert np.array_equal(brd[:, 0].cpu().numpy().astype(np.int64), z[f"stream/{metric}_best_idx"])


def test_mae_simsearch_driver_matches_reference_goldens():
    """utils.similarity.mae_simsearch (pool-based driver, HIP cosine + standardise kernels) against outputs of the
    REFERENCE's mae_simsearch captured around the same stub encoder (tests/golden/make_golden.py simsearch_cases):
    flat and tile-nested loaders, every token-selection mode, cosine / MSE / MAE, an n_batches limit."""
    from sky_embeddings_amd.utils import similarity as sim
    z = np.load(os.path.join(GOLDEN, "simsearch_driver.npz"))
    W = torch.from_numpy(z["ss/W"]).cuda()

    class TokenStub(torch.nn.Module):
        num_extra_tokens = 1

        def forward_features(self, x, ra_dec=None, mask_ratio=0, mask=None, reshape_out=False):
            B, C, H, Wd = x.shape
            p = x.reshape(B, C, H // 4, 4, Wd // 4, 4).permute(0, 2, 4, 1, 3, 5).reshape(B, (H // 4) * (Wd // 4), C * 16)
            tok = p @ W
            return torch.cat((tok.mean(dim=1, keepdim=True), tok), dim=1), None, None

    x, rd, tgt = torch.from_numpy(z["ss/x"]), torch.from_numpy(z["ss/ra_dec"]), torch.from_numpy(z["ss/target_latent"])
    N, B, n_save = x.shape[0], 16, 12
    flat = [(x[i:i + B], torch.zeros(B), rd[i:i + B]) for i in range(0, N, B)]
    tiles = [([[x[i:i + B], x[i + B:i + 2 * B]]], [[torch.zeros(B), torch.zeros(B)]], [[rd[i:i + B], rd[i + B:i + 2 * B]]])
             for i in range(0, N, 2 * B)]
    cases = [("cos_min", dict(metric="cosine", combine="min")), ("cos_mean_nw", dict(metric="cosine", combine="mean", use_weights=False)),
             ("cos_max_pool", dict(metric="cosine", combine="min", max_pool=True)), ("cos_cls", dict(metric="cosine", combine="max", cls_token=True)),
             ("mse_mean", dict(metric="MSE", combine="mean")), ("mae_min_nb3", dict(metric="MAE", combine="min", n_batches=3))]
    stub = TokenStub()
    for name, kw in cases:
        for nested, loader in ((False, flat), (True, tiles)):
            bs, bl, brd, bsc = sim.mae_simsearch(stub, tgt, loader, torch.device("cuda"), nested_batches=nested, n_save=n_save,
                                                 verbose=0, **kw)
            key = f"ss/{name}/{'tiles' if nested else 'flat'}"
            ref_s, ref_i = z[key + "/scores"], z[key + "/idx"]
            got_i = brd[:, 0].cpu().numpy().astype(np.int64)
            assert np.abs(bsc.cpu().numpy() - ref_s).max() <= 5e-6 * max(1.0, np.abs(ref_s).max()), key
            # same winners in the same order wherever the reference's own score gap exceeds the rounding band
            gap_ok = np.abs(np.diff(ref_s)) > 1e-5 * max(1.0, np.abs(ref_s).max())
            firm = np.concatenate(([True], gap_ok)) & np.concatenate((gap_ok, [True]))
            assert np.array_equal(got_i[firm], ref_i[firm]), (key, got_i, ref_i)
            assert set(got_i[:-1]) <= set(ref_i) | set(got_i[~firm]), key
            assert torch.equal(bs.cpu(), x[torch.from_numpy(got_i)]), key
            assert np.allclose(bl.cpu().numpy()[firm], z[key + "/latent"][firm], rtol=1e-4, atol=1e-5), key


def _tiny_ini(tmp_path, total_iters=6, bs=8):
    cfg = configparser.ConfigParser()
    cfg.read(os.path.join(ROOT, "configs", "mim_1.ini"))
    cfg["TRAINING"]["total_batch_iters"] = str(total_iters)
    cfg["TRAINING"]["batch_size"] = str(bs)
    return cfg


def test_build_model_train_checkpoint_resume_and_search(tmp_path):
    from sky_embeddings_amd import hdf5_lite, search
    from sky_embeddings_amd.utils.dataloaders import build_h5_dataloader
    from sky_embeddings_amd.utils.eval_fns import build_embedding_bank, mae_latent, mae_predict
    from sky_embeddings_amd.utils.mim_vit import build_model
    from sky_embeddings_amd.utils.pretrain_fns import run_iter
    from sky_embeddings_amd.utils.similarity import determine_target_features, mae_simsearch
    cfg = _tiny_ini(tmp_path)
    # fp32 parity mode: the two search paths below encode the test set in separate passes whose random
    # token shuffles differ (reference behaviour); in bf16 that alone moves scores by ~1e-3
    cfg["TRAINING"]["compute_dtype"] = "f32"
    data = hdf5_lite.make_synthetic_cutouts(str(tmp_path / "d.h5"), n=96, seed=1, nan_fraction=0.02)
    fn = str(tmp_path / "m.pth.tar")
    torch.manual_seed(0)
    model, losses, cur_iter, opt, sched = build_model(cfg, fn, torch.device("cuda"), build_optimizer=True)
    assert cur_iter == 1 and model.module.patch_embed.num_patches == 16 and model.module.num_extra_tokens == 1
    dl = build_h5_dataloader(data, batch_size=8, num_workers=0, patch_size=16, num_channels=5, img_size=64, shuffle=False)
    from collections import defaultdict
    cp = defaultdict(list)
    for i, (x, m, rd) in enumerate(dl):
        model, opt, sched, cp = run_iter(model, x.cuda(), rd, m, 0.75, opt, sched, cp, 'train')
        if i == 5:
            break
    tl = [float(v) for v in cp['train_loss']]
    assert len(tl) == 6 and all(np.isfinite(tl)) and opt.step_count == 6
    # checkpoint in the reference's format, resume restores everything
    torch.save({'batch_iters': 6, 'losses': dict(losses), 'optimizer': opt.state_dict(), 'lr_scheduler': sched.state_dict(),
                'model': {k: v.cpu() for k, v in model.module.state_dict().items()}}, fn)
    model2, losses2, cur2, opt2, sched2 = build_model(cfg, fn, torch.device("cuda"), build_optimizer=True)
    assert cur2 == 7 and opt2.step_count == 6 and sched2.last_epoch == 6 and abs(opt2.lr - opt.lr) < 1e-18
    for k, v in model.module.state_dict().items():
        assert torch.equal(v, model2.module.state_dict()[k]), k
    assert torch.equal(opt.store.m, opt2.store.m) and torch.equal(opt.store.v, opt2.store.v)
    # same next step on both
    x, m, rd = next(iter(dl))
    noise = torch.rand(8, 16, device="cuda")
    l1, _, _ = model.module.forward(x.cuda(), noise=noise)
    l2, _, _ = model2.module.forward(x.cuda(), noise=noise)
    assert float(l1) == float(l2)
    # predictions / latents through the eval mirrors
    pred, masked, orig = mae_predict(model, dl, torch.device("cuda"), 0.75)
    assert pred.shape == (8, 64, 64, 5) and np.isfinite(pred[~np.isnan(orig)]).all()
    lat = mae_latent(model, dl, torch.device("cuda"), n_batches=2, remove_cls=False, verbose=0)
    assert lat.shape == (16, 17, 192)
    # 4 augmented copies per sample (flip / resized crop / brightness / noise / NaN channels on the device): copy 0 is the
    # sample itself, so its cls embedding equals the un-augmented one; the copies differ
    lat_aug, imgs_aug = mae_latent(model, dl, torch.device("cuda"), n_batches=1, remove_cls=False, verbose=0, return_images=True,
                                   apply_augmentations=True, num_augmentations=4)
    assert lat_aug.shape == (8 * 5, 17, 192) and imgs_aug.shape == (8 * 5, 5, 64, 64)
    assert torch.allclose(lat_aug[0::5, 0], lat[:8, 0], atol=3e-2, rtol=3e-2)      # bf16 model, a different batch shape
    assert not torch.allclose(lat_aug[1, 0], lat_aug[0, 0]) and bool(torch.isfinite(lat_aug).all())
    # reference-shaped streaming search == encode-once bank + fused top-k kernel
    tgt = lat[:5]
    imgs, blat, brd, bsc = mae_simsearch(model, tgt, dl, torch.device("cuda"), metric='cosine', combine='min',
                                         use_weights=True, max_pool=True, cls_token=False, nested_batches=False, n_save=10,
                                         verbose=1000)
    assert imgs.shape == (10, 5, 64, 64) and bsc.shape == (10,) and bool((bsc[:-1] >= bsc[1:]).all())
    bank = build_embedding_bank(model, dl, torch.device("cuda"), pool='max')
    assert bank.shape == (96, 192)
    first = bank[:8]
    mu, sd = first.mean(0), first.std(0, unbiased=True)
    t = tgt.cuda()[:, 1:].max(dim=1, keepdim=True).values
    t = (t - mu) / (sd + 1e-8)
    search.standardise_(bank, mu, sd)
    avg, w = determine_target_features(t)
    s, i = search.cosine_topk(avg.reshape(1, -1), bank, 10, weights=w)
    assert np.allclose(s[0].cpu().numpy(), bsc.cpu().numpy(), rtol=2e-5, atol=2e-6)
    ds = dl.dataset
    for rank_, j in enumerate(i[0].cpu().tolist()):
        ref_img = ds[j][0]
        got = imgs[rank_].cpu()
        assert torch.equal(torch.nan_to_num(got), torch.nan_to_num(ref_img)), rank_


@pytest.mark.parametrize("mode", ["mae", "simmim"])
def test_pretrain_entry_point_runs(tmp_path, mode):
    """python pretrain_mim.py <ini> on synthetic HDF5 files (BASELINE configs[0] plumbing; the SimMIM + RA/Dec flavour of
    the reference's shipped configs on a CHUNKED train file, 70 cutouts: the ragged last batch takes the eager path)."""
    from sky_embeddings_amd import hdf5_lite
    dd = tmp_path / "data"
    dd.mkdir()
    hdf5_lite.make_synthetic_cutouts(str(dd / "synthetic_cutouts_GRIZY_64_train.h5"), n=64 if mode == "mae" else 70, seed=1234,
                                     chunked=mode == "simmim", nan_fraction=0.0 if mode == "mae" else 0.05)
    hdf5_lite.make_synthetic_cutouts(str(dd / "synthetic_cutouts_GRIZY_64_val.h5"), n=16, seed=4321)
    hdf5_lite.make_synthetic_cutouts(str(dd / "probe.h5"), n=60, seed=99, with_labels=True)
    # a private copy of the ini with a short schedule, in a scratch checkout layout
    work = tmp_path / "work"
    (work / "configs").mkdir(parents=True)
    cfg = _tiny_ini(tmp_path, total_iters=5 if mode == "mae" else 12)
    if mode == "simmim":
        cfg["ARCHITECTURE"].update(model_type="simmim", patch_size="8", embed_dim="96", ra_dec="True")
        cfg["TRAINING"].update(loss_fn="L1", max_mask_ratio="0.9")
    else:   # the linear-probe validation hook of the reference's loop (pretrain_mim.py:189-192)
        cfg["DATA"].update(lp_class_data_file="probe.h5", lp_regress_data_file="probe.h5", lp_combine="pool")
    with open(work / "configs" / "mim_t.ini", "w") as fh:
        cfg.write(fh)
    for name in ("pretrain_mim.py",):
        os.symlink(os.path.join(ROOT, name), work / name)
    for name in ("utils", "sky_embeddings_amd"):
        os.symlink(os.path.join(ROOT, name), work / name)
    env = dict(os.environ, PYTHONPATH=str(work))
    out = subprocess.run([sys.executable, str(work / "pretrain_mim.py"), "mim_t", "-v", "2", "-ct", "0.001", "-dd", str(dd)],
                         cwd=str(work), env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "Training complete." in out.stdout and "Total Loss" in out.stdout
    ck = torch.load(str(work / "models" / "mim_t.pth.tar"), map_location="cpu", weights_only=False)
    assert set(ck) == {"batch_iters", "losses", "optimizer", "lr_scheduler", "model"}
    if mode == "mae":
        assert len(ck["model"]) == 255 - 0 and ck["batch_iters"] >= 5 and len(ck["losses"]["val_loss"]) >= 1
        assert "Linear Probing Results:" in out.stdout and len(ck["losses"]["val_lp_acc"]) >= 1 and len(ck["losses"]["val_lp_r2"]) >= 1
    else:
        assert "decoder.0.weight" in ck["model"] and ck["batch_iters"] >= 12 and np.isfinite(ck["losses"]["train_loss"]).all()


@pytest.mark.parametrize("attn_pool", [False, True])
def test_simmim_radec_model_through_the_module_api(tmp_path, attn_pool):
    """A SimMIM + RA/Dec configuration (what the reference's shipped inis train), and its attention-pooled variant
    (attn_pool = True: one pooled token per image, utils/mim_vit.py:246-250), through utils.mim_vit.build_model, the per-item
    loader with MaskGenerator, run_iter, checkpoint / resume, mae_latent and mae_predict."""
    from collections import defaultdict
    from sky_embeddings_amd import hdf5_lite
    from sky_embeddings_amd.utils.dataloaders import build_h5_dataloader
    from sky_embeddings_amd.utils.eval_fns import mae_latent, mae_predict
    from sky_embeddings_amd.utils.mim_vit import build_model
    from sky_embeddings_amd.utils.pretrain_fns import run_iter
    cfg = _tiny_ini(tmp_path, total_iters=20, bs=8)
    cfg["ARCHITECTURE"].update(model_type="simmim", patch_size="8", img_size="64", embed_dim="96", ra_dec="True",
                               attn_pool=str(attn_pool))
    cfg["TRAINING"].update(loss_fn="L1", norm_pix_loss="True", max_mask_ratio="0.9", compute_dtype="f32", init_lr="0.0005")
    path = hdf5_lite.make_synthetic_cutouts(str(tmp_path / "c.h5"), n=32, seed=3, nan_fraction=0.05)
    fn = str(tmp_path / "simmim.pth.tar")
    model, losses, cur_iter, opt, sched = build_model(cfg, fn, "cuda", build_optimizer=True)
    assert model.module.simmim and model.module.num_extra_tokens == 2 and bool(model.module.attn_pool) == attn_pool
    dl = build_h5_dataloader(path, batch_size=8, num_workers=0, patch_size=8, num_channels=5, max_mask_ratio=0.9, img_size=64,
                             num_patches=model.module.patch_embed.num_patches, shuffle=False)
    lc = defaultdict(list)
    for epoch in range(3):
        for samples, masks, ra_decs in dl:
            model, opt, sched, lc = run_iter(model, samples.cuda(), ra_decs, masks, None, opt, sched, lc, mode="train")
    tl = [float(v) for v in lc["train_loss"]]
    assert all(np.isfinite(tl)) and np.mean(tl[-4:]) < np.mean(tl[:4]), tl
    torch.save({"batch_iters": 12, "losses": dict(losses), "optimizer": opt.state_dict(), "lr_scheduler": sched.state_dict(),
                "model": {k: v.detach().cpu() for k, v in model.module.state_dict().items()}}, fn)
    model2, _, it2, opt2, _ = build_model(cfg, fn, "cuda", build_optimizer=True)
    assert it2 == 13 and opt2.step_count == opt.step_count
    for k, v in model.module.state_dict().items():
        assert torch.equal(v, model2.module.state_dict()[k]), k
    lat = mae_latent(model, dl, "cuda", n_batches=2, verbose=0, remove_cls=False)
    lat = lat[0] if isinstance(lat, tuple) else lat
    assert tuple(lat.shape[1:]) == ((1, 96) if attn_pool else (2 + 64, 96)) and bool(torch.isfinite(torch.as_tensor(lat)).all())
    pred, masked, orig = mae_predict(model, dl, "cuda", None)
    assert pred.shape == orig.shape == (8, 64, 64, 5)
    img_like, _, _ = model.module.forward_features(next(iter(dl))[0].cuda(), ra_dec=next(iter(dl))[2])
    assert tuple(img_like.shape) == ((8, 96, 1, 1) if attn_pool else (8, 96, 8, 8))       # reshape_out (utils/mim_vit.py:431-436)


@pytest.mark.parametrize("case", ["mae_tiny_A", "mae_tiny_B_nan", "mae_tiny_I_radec"])
def test_downstream_vit_forward_features_matches_reference_goldens(case):
    """utils.vit.VisionTransformer.forward_features (utils/vit.py:344-388: input norm, NaN fill, patch embed + positions,
    cls token, Blocks, final norm, tokens in raster order) is line for line the encoder half of the reference's
    ``mim_vit.forward_features``; the goldens hold that encoder's output with ``mask_ratio = 0`` in the SHUFFLED order the
    MAE path returns plus ``ids_restore`` -- un-shuffled, they are what the downstream ViT built on the same weights must
    produce.  Also the ``reshape_out`` layout and the build_model round trip through a checkpoint file."""
    from tests.helpers import load_case, rel_err
    from sky_embeddings_amd.model_config import MAEConfig
    from sky_embeddings_amd.utils.vit import VisionTransformer
    z, cfg, state = load_case(case)
    c = MAEConfig(img_size=cfg.img_size, patch_size=cfg.patch_size, in_chans=cfg.in_chans, embed_dim=cfg.embed_dim, depth=cfg.depth,
                  num_heads=cfg.num_heads, decoder_embed_dim=cfg.decoder_embed_dim, decoder_depth=cfg.decoder_depth,
                  decoder_num_heads=cfg.decoder_num_heads, pixel_mean=cfg.pixel_mean, pixel_std=cfg.pixel_std, ra_dec=cfg.ra_dec)
    vit = VisionTransformer(c, "cuda", torch.float32)
    vit.load_encoder_state(state)
    imgs = torch.from_numpy(z["imgs"].copy())
    ra_dec = torch.from_numpy(z["ra_dec"].copy()) if cfg.ra_dec else None     # utils/vit.py:374-378: token after cls
    E = 2 if cfg.ra_dec else 1
    tok, m, ids = vit.forward_features(imgs.cuda(), ra_dec=ra_dec)
    assert m is None and ids is None
    ref, restore = z["latent_full"], z["ids_restore_full"]
    expected = ref.copy()
    expected[:, E:] = np.take_along_axis(ref[:, E:], restore[:, :, None], axis=1)      # patch l sits at shuffled position restore[l]
    assert tok.shape == expected.shape and rel_err(tok.cpu().numpy(), expected) < 2e-5
    grid = cfg.img_size // cfg.patch_size
    img_like, _, _ = vit.forward_features(imgs.cuda(), ra_dec=ra_dec, reshape_out=True)
    assert img_like.shape == (imgs.shape[0], cfg.embed_dim, grid, grid)
    assert torch.equal(img_like, tok[:, E:].permute(0, 2, 1).reshape(imgs.shape[0], cfg.embed_dim, grid, grid))
    assert vit.train(True).training and not vit.eval().training            # training is built (tests/test_predictor_gpu.py) ...
    # ... with dropout in the head too (round 6: tests/test_predictor_gpu.py::test_predictor_head_dropout); fp16 is a pretraining mode
    vd = VisionTransformer(c, "cuda", torch.float32, num_classes=2, global_pool="map", drop_rate=0.1).train(True)
    assert vd._head_mod.drop_rate == 0.1 and vd._head_mod.training and not vd.eval()._head_mod.training
    with pytest.raises(NotImplementedError):
        VisionTransformer(c, "cuda", torch.float16, num_classes=2, global_pool="map")


def test_linear_probe_hook_on_hip_embeddings(tmp_path):
    """utils/pretrain_fns.py:52-159: the linear-probe validation hook on embeddings from the HIP encoder.  The labels are a
    function of the cutout brightness, which even a randomly initialised encoder keeps linearly decodable: the probe has to
    beat chance clearly; every ``combine`` mode gives the documented feature shape."""
    from collections import defaultdict
    from sky_embeddings_amd import hdf5_lite
    from sky_embeddings_amd.utils.dataloaders import build_h5_dataloader
    from sky_embeddings_amd.utils.mim_vit import build_model
    from sky_embeddings_amd.utils.pretrain_fns import get_embeddings, linear_probe
    rng = np.random.default_rng(5)
    n = 240
    level = rng.uniform(-1.0, 1.0, n).astype(np.float32)
    cut = (rng.standard_normal((n, 5, 64, 64), dtype=np.float32) * 0.3 + level[:, None, None, None]).astype(np.float32)
    path = str(tmp_path / "labelled.h5")
    hdf5_lite.write_datasets(path, {"cutouts": cut, "ra": rng.uniform(0, 360, n).astype(np.float32),
                                    "dec": rng.uniform(-90, 90, n).astype(np.float32),
                                    "class": np.digitize(level, [-0.33, 0.33]).astype(np.int64), "zspec": (level + 1.0).astype(np.float32)})
    cfg = _tiny_ini(tmp_path)
    cfg["TRAINING"]["compute_dtype"] = "f32"
    torch.manual_seed(0)
    model, *_ = build_model(cfg, str(tmp_path / "none.pth.tar"), torch.device("cuda"), build_optimizer=True)
    template = build_h5_dataloader(path, batch_size=8, num_workers=0, patch_size=16, num_channels=5, img_size=64, shuffle=False)
    cp = defaultdict(list)
    linear_probe(model, cp, "cuda", template, class_data_path=path, regress_data_path=path, combine="pool")
    assert set(cp) == {"train_lp_acc", "val_lp_acc", "train_lp_r2", "val_lp_r2"}
    assert cp["val_lp_acc"][0] > 0.6 and cp["train_lp_acc"][0] > 0.6          # 3 classes: chance = 1/3
    assert cp["val_lp_r2"][0] > 0.5
    D, L = model.module.engine.cfg.embed_dim, 16
    for combine, width in (("token", D), ("flatten", L * D), ("pool", D), ("centralpool", D), ("central", 4 * D), ("mean", D)):
        x, y = get_embeddings(path, model, "cuda", template, y_label="zspec", combine=combine, remove_cls=combine != "token")
        assert x.shape == (n, width) and y.shape == (n,), combine
        assert np.allclose(x.mean(axis=0), 0, atol=1e-3)                          # standard-scaled features
    x, _ = get_embeddings(path, model, "cuda", template, combine="none")
    assert x.shape == (n, L, D) and abs(float(x.mean())) < 1e-3 and abs(float(x.std()) - 1) < 1e-3


def test_pretrain_entry_point_trains_from_survey_tiles(tmp_path):
    """python pretrain_mim.py <ini> with ``train_data_paths`` (what the reference's shipped MIM configs use: survey tiles in FITS,
    pretrain_mim.py:88-103): tiles -> HBM -> windows cut on the device -> the HIP-graph step (SimMIM + RA/Dec token)."""
    from sky_embeddings_amd import hdf5_lite
    from tests.test_feeder_gpu import _make_tiles
    dd = tmp_path / "data"
    tiles = dd / "pdr3_dud"
    tiles.mkdir(parents=True)
    _make_tiles(str(tiles), n_patches=2, missing=())
    hdf5_lite.make_synthetic_cutouts(str(dd / "synthetic_cutouts_GRIZY_64_val.h5"), n=16, seed=4321)
    work = tmp_path / "work"
    (work / "configs").mkdir(parents=True)
    cfg = _tiny_ini(tmp_path, total_iters=10)
    del cfg["DATA"]["train_data_file"]
    cfg["DATA"].update(train_data_paths=repr([str(tiles)]), bands="['G','R','I','Z','Y']", min_bands="5", cutouts_per_tile="48", use_calexp="True")
    cfg["ARCHITECTURE"].update(model_type="simmim", patch_size="8", embed_dim="96", ra_dec="True")
    cfg["TRAINING"].update(loss_fn="L1", max_mask_ratio="0.9")
    with open(work / "configs" / "mim_t.ini", "w") as fh:
        cfg.write(fh)
    for name in ("pretrain_mim.py", "utils", "sky_embeddings_amd"):
        os.symlink(os.path.join(ROOT, name), work / name)
    out = subprocess.run([sys.executable, str(work / "pretrain_mim.py"), "mim_t", "-v", "4", "-ct", "0.001", "-dd", str(dd)],
                         cwd=str(work), env=dict(os.environ, PYTHONPATH=str(work)), capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "Training complete." in out.stdout and "sky patches per process" in out.stdout
    ck = torch.load(str(work / "models" / "mim_t.pth.tar"), map_location="cpu", weights_only=False)
    assert ck["batch_iters"] >= 10 and np.isfinite(ck["losses"]["train_loss"]).all()


def test_sky_sim_search_entry_point_streams_survey_tiles(tmp_path):
    """python sky_sim_search.py <ini> (sky_sim_search.py:123-173): targets from an HDF5 file, the test set = OVERLAPPING cutouts
    of every FITS tile under --test_dirs (use_overlap, overlap 0.4, nested batches), best n_save kept.  One target is a window
    cut out of one of the tiles: it must come back as the best match, at its sky position."""
    from sky_embeddings_amd import fits_lite, hdf5_lite
    from tests.test_feeder_gpu import _make_tiles
    dd = tmp_path / "data"
    tiles = dd / "pdr3_dud"
    tiles.mkdir(parents=True)
    _make_tiles(str(tiles), n_patches=2, missing=())
    # targets: windows of tile 0 at grid positions of the overlapping sampler (step = int(64 * 0.6) = 38) + a synthetic cutout
    bands = ("G", "R", "I", "Z", "Y")
    first = sorted(f for f in os.listdir(tiles) if "-G-" in f)[0]
    imgs = [fits_lite.read_image_hdu(str(tiles / first.replace("-G-", f"-{b}-")), hdu=1).array() for b in bands]
    win = np.stack([im[38:38 + 64, 76:76 + 64] for im in imgs]).astype(np.float32)
    win = np.where(win < -3.0, -3.0, win)
    rng = np.random.default_rng(3)
    # (two targets: the window and a faintly perturbed copy -- the inverse-variance weights of determine_target_features need a
    # non-degenerate target set; their mean is the window's embedding to ~1e-3)
    cut = np.stack([win, win + 0.02 * rng.standard_normal((5, 64, 64)).astype(np.float32)])
    hdf5_lite.write_datasets(str(dd / "targets.h5"), {"cutouts": cut, "ra": np.zeros(2, np.float32), "dec": np.zeros(2, np.float32)})
    work = tmp_path / "work"
    (work / "configs").mkdir(parents=True)
    cfg = _tiny_ini(tmp_path)
    cfg["TRAINING"]["compute_dtype"] = "f32"
    cfg["DATA"].update(bands="['G','R','I','Z','Y']", min_bands="5", cutouts_per_tile="48", use_calexp="True")
    with open(work / "configs" / "mim_t.ini", "w") as fh:
        cfg.write(fh)
    # (a checkpoint of a SEEDED random initialisation: the entry point would otherwise draw its own weights, and the margin asserted
    # at the end would differ from run to run)
    from sky_embeddings_amd.utils.mim_vit import build_model as build_mae
    (work / "models").mkdir()
    torch.manual_seed(20260)
    mae, _, _ = build_mae(cfg, str(tmp_path / "none.pth.tar"), torch.device("cuda"))
    torch.save({"batch_iters": 1, "losses": {}, "model": {k: v.cpu() for k, v in mae.module.state_dict().items()}}, str(work / "models" / "mim_t.pth.tar"))
    del mae
    for name in ("sky_sim_search.py", "utils", "sky_embeddings_amd"):
        os.symlink(os.path.join(ROOT, name), work / name)
    out = subprocess.run([sys.executable, str(work / "sky_sim_search.py"), "mim_t", "-tgt_fn", "targets.h5", "-tst_dirs", str(tiles), "-tgt_i", "[0,1]",
                          "-aug", "False", "-bs", "16", "-ns", "12", "-dd", str(dd)],
                         cwd=str(work), env=dict(os.environ, PYTHONPATH=str(work)), capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    z = np.load(str(work / "results" / "mim_t_targets_simsearch_results.npz"))
    assert set(z.files) == {"test_ra_decs", "test_scores", "target_images", "target_features", "test_images", "test_features"}
    assert z["test_images"].shape == (12, 5, 64, 64) and z["test_ra_decs"].shape == (12, 2) and z["test_scores"].shape == (12,)
    assert bool((z["test_scores"][:-1] >= z["test_scores"][1:]).all())
    # the window the target was cut from is the best match (a randomly initialised encoder still maps equal pixels to equal embeddings)
    assert np.array_equal(np.nan_to_num(z["test_images"][0]), np.nan_to_num(win)) and z["test_scores"][0] > 0.99
    # ... clear of every other window (noise tiles through a randomly initialised, unseeded encoder score 0.985-0.999: the margin
    # is that of the float arithmetic, not of a trained embedding)
    assert z["test_scores"][1] < z["test_scores"][0] - 3e-4


@pytest.mark.parametrize("method,loss_fn", [("lp", "crossentropy"), ("ft", "mse")])
def test_train_predictor_entry_point(tmp_path, method, loss_fn):
    """python train_predictor.py <ini> (train_predictor.py:13-270): a predictor on a pre-trained MAE checkpoint -- attentive probe
    with cross-entropy on the class labels (the shipped cls_ap_*.ini), fine-tuning with MSE on a normalised redshift (z_ft_2.ini) --
    validation, best-model and periodic checkpoints in the reference's format, resume from the best checkpoint."""
    from sky_embeddings_amd import hdf5_lite
    from sky_embeddings_amd.utils.mim_vit import build_model as build_mae
    dd = tmp_path / "data"
    dd.mkdir()
    hdf5_lite.make_synthetic_cutouts(str(dd / "train.h5"), n=64, seed=11, with_labels=True)
    # validation objects: twelve with a bright central source (S/N > 5 in every channel: the ones test_predictor.py evaluates), four without
    rng = np.random.default_rng(12)
    vcut = rng.standard_normal((16, 5, 64, 64)).astype(np.float32)
    vcut[:12, :, 28:36, 28:36] += 10.0
    hdf5_lite.write_datasets(str(dd / "val.h5"), {"cutouts": vcut, "ra": rng.uniform(0, 360, 16).astype(np.float32),
                                                  "dec": rng.uniform(-90, 90, 16).astype(np.float32), "zspec": rng.uniform(0.2, 1.6, 16).astype(np.float32),
                                                  "zspec_err": np.full(16, 0.01, np.float32), "class": rng.integers(0, 3, 16).astype(np.int64)})
    work = tmp_path / "work"
    (work / "configs").mkdir(parents=True)
    (work / "models").mkdir()
    mae_cfg = _tiny_ini(tmp_path)
    with open(work / "configs" / "mim_t.ini", "w") as fh:
        mae_cfg.write(fh)
    mae, _, _ = build_mae(mae_cfg, str(tmp_path / "none.pth.tar"), torch.device("cuda"))
    torch.save({"batch_iters": 5, "losses": {}, "model": {k: v.cpu() for k, v in mae.module.state_dict().items()}}, str(work / "models" / "mim_t.pth.tar"))
    cfg = configparser.ConfigParser()
    # (the shipped cls_*.ini all say label_means = [0], label_stds = [0]: utils/vit.py:38-39 only takes their LENGTH, so predictions
    # are never multiplied by that zero -- the confusion matrix below would be degenerate otherwise)
    cfg["DATA"] = {"train_data_file": "train.h5", "val_data_file": "val.h5", "label_means": "[0]" if loss_fn == "crossentropy" else "[1.0]",
                   "label_stds": "[0]" if loss_fn == "crossentropy" else "[0.6]"}
    cfg["DATA"].update({"label_keys": "['class']", "num_classes": "3"} if loss_fn == "crossentropy" else {"label_keys": "['zspec']"})
    cfg["TRAINING"] = {"train_method": method, "pretained_mae": "mim_t", "num_train": "40", "batch_size": "8", "total_batch_iters": "6", "layer_decay": "0.7",
                       "weight_decay": "0.05", "init_lr": "0.001", "final_lr_factor": "10", "augment": "False", "brightness": "0.8", "noise": "0.1",
                       "nan_channels": "5", "use_label_errs": "False", "loss_fn": loss_fn}
    cfg["ARCHITECTURE"] = {"img_size": "64", "global_pool": "map", "dropout": "0.0"}
    with open(work / "configs" / "pred_t.ini", "w") as fh:
        cfg.write(fh)
    for name in ("train_predictor.py", "utils", "sky_embeddings_amd"):
        os.symlink(os.path.join(ROOT, name), work / name)
    cmd = [sys.executable, str(work / "train_predictor.py"), "pred_t", "-v", "3", "-ct", "0.0005", "-dd", str(dd)]
    env = dict(os.environ, PYTHONPATH=str(work))
    out = subprocess.run(cmd, cwd=str(work), env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "Training complete." in out.stdout and "Loading pre-trained MAE model weights..." in out.stdout and "Validation Dataset" in out.stdout
    assert ("linear probing" if method == "lp" else "fine-tuning") in out.stdout and ("Accuracy" if loss_fn == "crossentropy" else "MAE") in out.stdout
    for fn in ("pred_t.pth.tar", "pred_t_best.pth.tar"):
        ck = torch.load(str(work / "models" / fn), map_location="cpu", weights_only=False)
        assert set(ck) == {"batch_iters", "losses", "optimizer", "lr_scheduler", "model"}
        assert "attn_pool.latent" in ck["model"] and "head.weight" in ck["model"] and "decoder_embed.weight" not in ck["model"]
        assert np.isfinite(ck["losses"]["val_loss"]).all() and len(ck["losses"]["batch_iters"]) >= 1
    enc_key = "blocks.0.attn.qkv.weight"
    trained = torch.load(str(work / "models" / "pred_t.pth.tar"), map_location="cpu", weights_only=False)["model"]
    same = torch.equal(trained[enc_key], mae.module.state_dict()[enc_key].cpu())
    assert same == (method == "lp")                                       # the probe leaves the encoder alone, fine-tuning moves it
    out2 = subprocess.run(cmd, cwd=str(work), env=env, capture_output=True, text=True, timeout=600)       # resume: from the best checkpoint
    assert out2.returncode == 0 and "Loading saved model weights..." in out2.stdout, out2.stdout[-1500:] + out2.stderr[-1500:]
    # ---- python test_predictor.py <ini> (test_predictor.py:12-118): the evaluation of the trained predictor; the numbers of the
    # reference's figures land in figures/*.npz
    for name in ("test_predictor.py", "compare_predictors.py"):
        os.symlink(os.path.join(ROOT, name), work / name)
    out3 = subprocess.run([sys.executable, str(work / "test_predictor.py"), "pred_t", "-dd", str(dd)], cwd=str(work), env=env,
                          capture_output=True, text=True, timeout=600)
    assert out3.returncode == 0 and "Testing complete." in out3.stdout, out3.stdout[-2000:] + out3.stderr[-2000:]
    prog = np.load(str(work / "figures" / "pred_t_best_progress.npz"))
    assert "val_loss" in prog.files and len(prog["val_loss"]) >= 1
    # ... against the predictions of the same checkpoint through the module API
    from sky_embeddings_amd.utils.vit import build_model as build_vit
    from sky_embeddings_amd.utils.dataloaders import build_h5_dataloader
    from sky_embeddings_amd.utils.eval_fns import ft_predict
    from sky_embeddings_amd.utils.misc import h5_snr
    from sky_embeddings_amd.utils import plotting_fns as pf
    model, _, _ = build_vit(cfg, mae_cfg, str(work / "models" / "pred_t_best.pth.tar"), str(work / "models" / "mim_t.pth.tar"), torch.device("cuda"))
    loader = build_h5_dataloader(str(dd / "val.h5"), batch_size=8, num_workers=1, label_keys=eval(cfg["DATA"]["label_keys"]), img_size=64,
                                 patch_size=int(mae_cfg["ARCHITECTURE"]["patch_size"]), num_channels=int(mae_cfg["ARCHITECTURE"]["num_channels"]),
                                 num_patches=model.module.patch_embed.num_patches, shuffle=False)
    tgt, pred = ft_predict(model, loader, torch.device("cuda"))
    assert tgt.shape[0] == 16 and pred.shape == (16, 3 if loss_fn == "crossentropy" else 1)
    keep = np.nanmin(h5_snr(str(dd / "val.h5"))[:, :5], axis=1) > 5
    assert int(keep.sum()) == 12
    if loss_fn == "mse":
        z = np.load(str(work / "figures" / "pred_t_redshift.npz"))
        _, bias, mad, frac = pf.photoz_prediction_metrics(pred[keep].reshape(-1), tgt[keep].reshape(-1), threshold=0.15)
        assert np.allclose([z["bias"], z["mad"], z["frac_out"]], [bias, mad, frac], rtol=1e-5, atol=1e-7)
        assert z["z_bin_counts"].shape == (8,) and z["snr_bin_counts"].shape == (8,)
        r = np.load(str(work / "figures" / "pred_t_predictions.npz"))
        assert np.allclose(r["resid"][:, 0], (pred - tgt)[keep, 0], atol=1e-6)
    else:
        z = np.load(str(work / "figures" / "pred_t_classes.npz"))
        cm = pf.confusion_matrix(tgt[keep, 0], pred[keep].argmax(1), n_classes=3)
        assert np.array_equal(z["confusion_matrix"], cm) and int(cm.sum()) == int(keep.sum())
        assert float(np.abs(pred - pred[:1]).max()) > 0                   # label_stds = [0] did not zero the predictions (see cfg["DATA"])
        # ---- python compare_predictors.py (compare_predictors.py:150-250): the families' score table; one member present here
        for src, dst in (("configs/pred_t.ini", "configs/cls_ap_012k.ini"), ("models/pred_t_best.pth.tar", "models/cls_ap_012k_best.pth.tar")):
            shutil.copy(str(work / src), str(work / dst))
        out4 = subprocess.run([sys.executable, str(work / "compare_predictors.py"), "x", "-dd", str(dd)], cwd=str(work), env=env,
                              capture_output=True, text=True, timeout=600)
        assert out4.returncode == 0 and "Testing complete." in out4.stdout, out4.stdout[-2000:] + out4.stderr[-2000:]
        sc = np.load(str(work / "figures" / "numsamples_class.npz"))["scores"]
        assert sc.shape == (5, 3, 8) and np.isclose(sc[2, 0, 0], np.mean(pred.argmax(1) == tgt[:, 0])) and np.isnan(sc[0, 0, 0])


def test_compute_similarity_central_patches_matches_reference_goldens():
    """n_central_patches (utils/similarity.py:238-240): the reference's own compute_similarity with the select_centre import it
    lacks supplied by the generator (tests/golden/make_golden.py central_cases)."""
    from sky_embeddings_amd.utils import similarity as sim
    z = np.load(os.path.join(GOLDEN, "similarity_central.npz"))
    keys = sorted({k.rsplit("/", 1)[0] for k in z.files if k.endswith("/target")})
    assert len(keys) == 3
    for key in keys:
        n = int(key.rsplit("_", 1)[1])
        tgt, tst = torch.from_numpy(z[key + "/target"]).cuda(), torch.from_numpy(z[key + "/test"]).cuda()
        for metric in ("cosine", "MSE", "MAE"):
            for combine in ("min", "mean", "max"):
                s = sim.compute_similarity(tgt, tst, metric=metric, combine=combine, n_central_patches=n).cpu().numpy()
                ref = z[f"{key}/{metric}_{combine}"]
                assert np.abs(s - ref).max() <= 2e-6 * max(1.0, np.abs(ref).max()), (key, metric, combine)
