"""GPU: SimMIM mode (per-channel pixel masks, all tokens encoded, linear head + PixelShuffle index map, pixel loss) and the
RA/Dec LocationEncoder token through the C ABI, against goldens captured from the reference (tests/golden/simmim_tiny_*).

Tolerances as in tests/test_mae_parity_gpu.py: f32 parity mode loss 2e-5 rel, pred 2e-5 rel-L2, gradients 2e-4 of the
tensor's max; bf16 mode loss 1e-2, pred 3e-2 rel-L2."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests.helpers import load_simmim_case, rel_err

CASES = ["simmim_tiny_F_l1_nan", "simmim_tiny_G_mse", "simmim_tiny_H_radec"]


def make_engine(cfg, state, dtype):
    from sky_embeddings_amd.model_config import MAEConfig
    from sky_embeddings_amd.simmim_engine import SimMIMEngine
    c = MAEConfig(img_size=cfg.img_size, patch_size=cfg.patch_size, in_chans=cfg.in_chans, embed_dim=cfg.embed_dim,
                  depth=cfg.depth, num_heads=cfg.num_heads, norm_pix_loss=cfg.norm_pix_loss, loss_fn=cfg.loss_fn,
                  pixel_mean=cfg.pixel_mean, pixel_std=cfg.pixel_std, simmim=True, ra_dec=cfg.ra_dec)
    eng = SimMIMEngine(c, device="cuda", compute_dtype=dtype, seed=0)
    eng.load_state_dict(state)
    return eng


@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_simmim_forward_backward_vs_reference_goldens(name, dtype):
    z, cfg, st, imgs, pmask, ra_dec = load_simmim_case(name)
    eng = make_engine(cfg, st, dtype)
    f32 = dtype == torch.float32
    rd = ra_dec.cuda() if ra_dec is not None else None
    loss, pred, _ = eng.forward_train(imgs.cuda(), mask=pmask.cuda(), ra_dec=rd)
    eng.backward()
    torch.cuda.synchronize()
    assert abs(float(loss) - float(z["loss"])) <= (2e-5 if f32 else 1e-2) * abs(float(z["loss"]))
    assert rel_err(pred.cpu().numpy(), z["pred"]) < (2e-5 if f32 else 3e-2)
    for k in eng.store.order:
        r = z["grad/" + k]
        g = eng.store.grad(k).cpu().numpy().reshape(r.shape)
        assert np.isfinite(g).all(), k
        scale = max(float(np.abs(r).max()), 1e-6)
        if f32:
            assert float(np.abs(g - r).max()) <= 2e-4 * scale, (k, float(np.abs(g - r).max()), scale)
        else:
            assert rel_err(g, r) < 8e-2 or float(np.abs(g - r).max()) < 6e-2 * scale, k
    # encoder-only path (utils/eval_fns.py:115 shape: tokens in order, extra tokens first)
    lat, _, _ = eng.forward_features(imgs.cuda(), mask=pmask.cuda(), ra_dec=rd)
    assert rel_err(lat.cpu().numpy(), z["latent"]) < (2e-5 if f32 else 2e-2)
    if ra_dec is not None and f32:
        w = eng._ws[(imgs.shape[0], cfg.num_patches, True)]
        assert rel_err(w["sh"].cpu().numpy(), z["sh_features"]) < 2e-6


@pytest.mark.parametrize("name", ["simmim_tiny_F_l1_nan", "simmim_tiny_H_radec"])
def test_simmim_three_optimiser_steps_match_reference(name):
    """forward + backward + fused AdamW + cosine LR, three steps, vs the reference's parameters (mask_token untouched)."""
    from sky_embeddings_amd.optim import CosineLR, FusedAdamW
    z, cfg, st, imgs, pmask, ra_dec = load_simmim_case(name)
    init_lr, wd, total, flf = [float(v) for v in z["opt_hparams"]]
    eng = make_engine(cfg, st, torch.float32)
    opt = FusedAdamW(eng, lr=init_lr, betas=(0.9, 0.95), weight_decay=wd)
    sched = CosineLR(opt, int(total), eta_min=init_lr / flf)
    x, m = imgs.cuda(), pmask.cuda()
    rd = ra_dec.cuda() if ra_dec is not None else None
    for it in range(3):
        loss, _, _ = eng.forward_train(x, mask=m, ra_dec=rd)
        eng.backward()
        opt.step()
        sched.step()
        assert abs(float(loss) - float(z["step_losses"][it])) <= 3e-5 * abs(float(z["step_losses"][it])), it
    sd = eng.state_dict()
    for k in eng.store.order:
        ref = z[f"state_after3/{k}"]
        # Adam turns rounding noise on near-zero gradients into moves of a fraction of lr (tests/test_oracle_golden.py);
        # the Siren's sin(30 z) amplifies it a little further for the RA/Dec case
        tol = 5e-6 * max(float(np.abs(ref).max()), 1e-3) + 5e-2 * init_lr
        assert float(np.abs(sd[k].cpu().numpy() - ref).max()) <= tol, k
    assert np.array_equal(sd["mask_token"].cpu().numpy(), z["state_after3/mask_token"])


def test_simmim_graph_step_equals_eager_and_staged():
    """TrainStep in SimMIM mode: HIP-graph replay (monolithic and staged) == eager execution, bit for bit."""
    from sky_embeddings_amd.optim import CosineLR, FusedAdamW
    from sky_embeddings_amd.train_step import TrainStep
    z, cfg, st, imgs, pmask, ra_dec = load_simmim_case("simmim_tiny_H_radec")
    B = 8
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, cfg.in_chans, cfg.img_size, cfg.img_size, generator=g).cuda()
    m = (torch.rand(B, cfg.in_chans, cfg.grid, cfg.grid, generator=g) < 0.5).float()
    m = m.repeat_interleave(cfg.patch_size, 2).repeat_interleave(cfg.patch_size, 3).contiguous().cuda()
    rd = torch.stack([torch.rand(B, generator=g) * 360, torch.rand(B, generator=g) * 180 - 90], 1).cuda()
    results = []
    for staged, graph in ((False, False), (False, True), (True, True)):
        eng = make_engine(cfg, st, torch.bfloat16)
        opt = FusedAdamW(eng, lr=1e-3)
        step = TrainStep(eng, opt, CosineLR(opt, 100), B, use_graph=graph, staged=staged, n_encoder_groups=2)
        for _ in range(3):
            loss = step(x, m, rd)
        torch.cuda.synchronize()
        results.append((float(loss), eng.store.g.clone(), eng.store.p.clone()))
    for r in results[1:]:
        assert r[0] == results[0][0] and torch.equal(r[1], results[0][1]) and torch.equal(r[2], results[0][2])
