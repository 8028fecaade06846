"""GPU: end-to-end MAE forward / backward / optimiser parity of the HIP path (through the C ABI)
against (i) the golden vectors captured from the reference and (ii) the CPU oracle.

Tolerances (north_star: loss and reconstructed pixels within 1e-3 relative fp32):
  * f32 parity mode (exact-fp32 MFMA): loss 2e-5 rel, pred 2e-5 rel-L2, gradients 2e-4 of the
    tensor's max -- i.e. well inside 1e-3;
  * bf16 throughput mode: loss 1e-2 rel, pred 3e-2 rel-L2 (bf16 operands, fp32 accumulation).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import mae_oracle as mo
from tests.helpers import record_parity, load_case, rel_err

CASES = ["mae_tiny_A", "mae_tiny_B_nan", "mae_tiny_C_nonorm", "mae_tiny_D_l1", "mae_tiny_E_p8", "mae_tiny_I_radec"]


def make_engine(cfg, state, dtype):
    from sky_embeddings_amd.engine import MAEEngine
    from sky_embeddings_amd.model_config import MAEConfig
    c = MAEConfig(img_size=cfg.img_size, patch_size=cfg.patch_size, in_chans=cfg.in_chans, embed_dim=cfg.embed_dim,
                  depth=cfg.depth, num_heads=cfg.num_heads, decoder_embed_dim=cfg.decoder_embed_dim,
                  decoder_depth=cfg.decoder_depth, decoder_num_heads=cfg.decoder_num_heads, norm_pix_loss=cfg.norm_pix_loss,
                  loss_fn=cfg.loss_fn, pixel_mean=cfg.pixel_mean, pixel_std=cfg.pixel_std, ra_dec=cfg.ra_dec)
    eng = MAEEngine(c, device="cuda", compute_dtype=dtype, seed=0)
    eng.load_state_dict(state)
    return eng


@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_forward_backward_vs_reference_goldens(name, dtype):
    z, cfg, st = load_case(name)
    eng = make_engine(cfg, st, dtype)
    imgs = torch.from_numpy(z["imgs"]).cuda()
    noise = torch.from_numpy(z["noise"]).cuda()
    ratio = float(z["mask_ratio"])
    rd = torch.from_numpy(z["ra_dec"]).cuda() if cfg.ra_dec else None      # case I: MAE mode with the RA/Dec token
    loss, pred, mask = eng.forward_train(imgs, ratio, noise, ra_dec=rd)
    eng.backward()
    torch.cuda.synchronize()
    # integer / index work: bit exact
    assert np.array_equal(mask.cpu().numpy(), z["mask"])
    assert np.array_equal(eng._ws[(imgs.shape[0], int(cfg.num_patches * (1 - ratio)), True)]["ids_restore"].cpu().numpy(),
                          z["ids_restore"])
    f32 = dtype == torch.float32
    assert abs(float(loss) - float(z["loss"])) <= (2e-5 if f32 else 1e-2) * abs(float(z["loss"]))
    assert rel_err(pred.cpu().numpy(), z["pred"]) < (2e-5 if f32 else 3e-2)
    # gradients: reference goldens where they are finite, else the oracle's nan-safe gradients
    nan_in = bool(np.isnan(z["imgs"]).any())
    if nan_in:
        _, _, _, _, _, ref = mo.loss_and_grads(st, torch.from_numpy(z["imgs"]), cfg, ratio, torch.from_numpy(z["noise"]),
                                               nan_safe=True, ra_dec=None if rd is None else rd.cpu())
        ref = {k: v.numpy() for k, v in ref.items()}
    else:
        ref = {k[len("grad/"):]: z[k] for k in z.files if k.startswith("grad/")}
    for k, r in ref.items():
        g = eng.store.grad(k).cpu().numpy()
        assert np.isfinite(g).all(), k
        scale = max(float(np.abs(r).max()), 1e-6)
        tol = (2e-4 if f32 else 6e-2) * scale
        if not f32:  # bf16: judge by relative L2 of the tensor (elementwise bf16 noise is expected)
            assert rel_err(g, r) < 8e-2 or float(np.abs(g - r).max()) < tol, k
        else:
            assert float(np.abs(g - r).max()) <= tol, (k, float(np.abs(g - r).max()), scale)


@pytest.mark.parametrize("case", ["mae_tiny_A", "mae_tiny_I_radec"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_encoder_only_path(dtype, case):
    z, cfg, st = load_case(case)
    eng = make_engine(cfg, st, dtype)
    imgs, noise = torch.from_numpy(z["imgs"]).cuda(), torch.from_numpy(z["noise"]).cuda()
    lat, mask, ids = eng.forward_features(imgs, 0.0, noise, ra_dec=torch.from_numpy(z["ra_dec"]).cuda() if cfg.ra_dec else None)
    assert np.array_equal(ids.cpu().numpy(), z["ids_restore_full"])
    assert rel_err(lat.cpu().numpy(), z["latent_full"]) < (2e-5 if dtype == torch.float32 else 2e-2)


def test_adamw_cosine_training_steps_match_reference():
    """Three run_iter steps (forward, backward, AdamW, cosine LR) through the public mirror API."""
    from sky_embeddings_amd.optim import CosineLR, FusedAdamW
    from sky_embeddings_amd.utils.pretrain_fns import run_iter
    z, cfg, st = load_case("mae_tiny_A")
    init_lr, wd, total, flf = [float(v) for v in z["opt_hparams"]]
    from sky_embeddings_amd.utils.mim_vit import MaskedAutoencoderViT, _DataParallelShim
    m = MaskedAutoencoderViT(img_size=cfg.img_size, patch_size=cfg.patch_size, in_chans=cfg.in_chans,
                             embed_dim=cfg.embed_dim, depth=cfg.depth, num_heads=cfg.num_heads,
                             decoder_embed_dim=cfg.decoder_embed_dim, decoder_depth=cfg.decoder_depth,
                             decoder_num_heads=cfg.decoder_num_heads, norm_pix_loss=cfg.norm_pix_loss, loss_fn=cfg.loss_fn,
                             pixel_mean=cfg.pixel_mean, pixel_std=cfg.pixel_std, compute_dtype=torch.float32)
    m.load_state_dict(st)
    model = _DataParallelShim(m)
    opt = FusedAdamW(m.engine, lr=init_lr, betas=(0.9, 0.95), weight_decay=wd)
    sched = CosineLR(opt, int(total), eta_min=init_lr / flf)
    imgs = torch.from_numpy(z["imgs"]).cuda()
    from collections import defaultdict
    losses = defaultdict(list)
    for it in range(3):
        assert abs(opt.lr - float(z["step_lrs"][it])) <= 1e-12 * init_lr
        noise = torch.from_numpy(z["step_noises"][it]).cuda()
        # run_iter has no noise argument (reference signature): inject through the engine hook
        m.forward_noise = noise
        loss, _, _ = model(imgs, mask_ratio=float(z["mask_ratio"]), noise=noise)
        loss.backward()
        opt.step()
        opt.zero_grad(set_to_none=True)
        sched.step()
        assert abs(float(loss) - float(z["step_losses"][it])) <= 2e-5 * abs(float(z["step_losses"][it])), it
        if it in (0, 2):
            sd = m.state_dict()
            for k in opt.store.order:
                ref = z[f"state_after{it + 1}/{k}"]
                # Adam's first steps move every element by ~lr * sign(g): elements whose gradient is rounding
                # noise may differ by a fraction of lr (see tests/test_oracle_golden.py)
                tol = 5e-6 * max(float(np.abs(ref).max()), 1e-3) + 2e-2 * init_lr
                assert float(np.abs(sd[k].cpu().numpy() - ref).max()) <= tol, (it, k)
    # run_iter itself (random noise): loss decreases and the bookkeeping matches the reference's
    model2, opt2, sched2, losses = run_iter(model, imgs, None, None, float(z["mask_ratio"]), opt, sched, losses, 'train')
    assert len(losses['train_loss']) == 1 and opt.step_count == 4 and sched.last_epoch == 4
    run_iter(model, imgs, None, None, float(z["mask_ratio"]), opt, sched, losses, 'val')
    assert len(losses['val_loss']) == 1 and opt.step_count == 4
    # optimizer / scheduler state round trip in torch's layout
    sd = opt.state_dict()
    assert len(sd["state"]) == len(opt.store.order) and len(sd["param_groups"]) == 2
    assert sd["param_groups"][0]["weight_decay"] == 0.0 and sd["param_groups"][1]["weight_decay"] == wd
    opt3 = FusedAdamW(m.engine, lr=init_lr, weight_decay=wd)
    m_before = opt.store.m.clone()
    opt.store.m.zero_()
    opt3.load_state_dict(sd)
    assert opt3.step_count == 4 and torch.equal(opt.store.m, m_before)


@pytest.mark.parametrize("B", [8, 256])
def test_full_size_config_a_against_oracle(B):
    """BASELINE config A (MAE ViT-B/16, 5x64x64, mask 0.75) vs the CPU oracle: B=8, and B=256 -- the batch bench.py times
    (BASELINE configs[1]); f32 parity mode and the bf16 mode the benchmark runs."""
    from sky_embeddings_amd.engine import MAEEngine
    from sky_embeddings_amd.model_config import config_for
    cfg_o = mo.config_for("base", patch_size=16, in_chans=5, img_size=64, embed_dim=768)
    st = mo.init_state(cfg_o, seed=0)
    g = torch.Generator().manual_seed(1234)
    imgs = torch.randn(B, 5, 64, 64, generator=g).clamp_(min=-3.0)
    noise = torch.rand(B, 16, generator=g)
    loss_o, pred_o, mask_o, _, _, grads_o = mo.loss_and_grads(st, imgs, cfg_o, 0.75, noise)
    # bars = 2x the errors measured on MI355X (profiles/r03_parity_errors.json); f32 is the north-star mode (1e-3)
    # measured (B = 256 / 8): f32 loss 0 / 0, pred 1.2e-6, worst gradient 2.0e-6 (patch_embed.proj.weight); bf16 loss 9.5e-5 / 1.4e-4,
    # pred 6.0e-3, worst gradient 1.07e-2
    for dtype, ltol, ptol, gtol in ((torch.float32, 1e-6, 2.5e-6, 4e-6), (torch.bfloat16, 3e-4, 1.2e-2, 2.2e-2)):
        eng = MAEEngine(config_for("base", patch_size=16, in_chans=5, img_size=64, embed_dim=768), compute_dtype=dtype, seed=0)
        eng.load_state_dict(st)
        loss, pred, mask = eng.forward_train(imgs.cuda(), 0.75, noise.cuda())
        eng.backward()
        assert torch.equal(mask.cpu(), mask_o)
        loss_rel = abs(float(loss) - float(loss_o)) / float(loss_o)
        pred_rel = rel_err(pred.cpu().numpy(), pred_o.numpy())
        grad_rel = {k: rel_err(eng.store.grad(k).cpu().numpy(), grads_o[k].numpy()) for k in eng.store.order}
        worst = max(grad_rel, key=grad_rel.get)
        record_parity(f"config_A_B{B}_{'f32' if dtype == torch.float32 else 'bf16'}",
                      dict(loss_rel=loss_rel, pred_rel_l2=pred_rel, grad_rel_l2_max=grad_rel[worst], grad_worst_tensor=worst,
                           grad_rel_l2_median=float(np.median(list(grad_rel.values())))))
        assert loss_rel <= ltol, (dtype, float(loss), float(loss_o))
        assert pred_rel < ptol
        for k, r in grad_rel.items():
            assert r < gtol, (dtype, k, r)
        del eng
        torch.cuda.empty_cache()


def test_staged_backward_equals_monolithic_and_graph_replay():
    """The DDP-overlap schedule (forward+decoder | encoder groups | embedding as separate HIP graphs) gives
    bit-identical gradients to the single-graph schedule, graph replay equals eager execution, and running the
    weight-gradient GEMMs on the side stream changes nothing."""
    from sky_embeddings_amd.engine import MAEEngine
    from sky_embeddings_amd.model_config import config_for
    from sky_embeddings_amd.optim import CosineLR, FusedAdamW
    from sky_embeddings_amd.train_step import TrainStep
    cfg = config_for("tiny", img_size=64, patch_size=16, in_chans=5, embed_dim=192)
    g = torch.Generator().manual_seed(0)
    imgs = torch.randn(8, 5, 64, 64, generator=g).cuda()
    results = []
    # (staged, graph, weight gradients on the side stream)
    # (staged, graph, weight gradients on the side stream, AdamW of finished stages on the optimiser stream)
    for staged, graph, overlap, opt_overlap in ((False, False, False, False), (False, False, True, False),
                                                (False, True, True, False), (True, True, True, True),
                                                (True, True, False, True), (True, False, False, True),
                                                (True, True, False, False)):
        eng = MAEEngine(cfg, compute_dtype=torch.bfloat16, seed=1)
        opt = FusedAdamW(eng, lr=1e-3)
        step = TrainStep(eng, opt, CosineLR(opt, 100), 8, use_graph=graph, staged=staged, n_encoder_groups=4, fused_adamw=False,
                         wgrad_overlap=overlap, optimizer_overlap=opt_overlap)
        torch.manual_seed(123)            # same masking noise stream for every schedule
        for _ in range(3):
            loss = step(imgs)
        torch.cuda.synchronize()
        results.append((float(loss), eng.store.g.clone(), eng.store.p.clone()))
    for r in results[1:]:
        assert r[0] == results[0][0]
        assert torch.equal(r[1], results[0][1]) and torch.equal(r[2], results[0][2])


def test_bf16_gradient_mirror_written_by_the_weight_gradient_launches(monkeypatch):
    """Data-parallel schedule with bf16 gradient communication: the grouped weight-gradient launches writing bf16 straight into
    the mirror give the same mirror, bit for bit, as fp32 gradients + the cast pass (the same rounding of the same accumulator),
    and so the same parameters after three steps."""
    from sky_embeddings_amd.engine import MAEEngine
    from sky_embeddings_amd.model_config import config_for
    from sky_embeddings_amd.optim import CosineLR, FusedAdamW
    from sky_embeddings_amd.train_step import TrainStep
    cfg = config_for("tiny", img_size=64, patch_size=16, in_chans=5, embed_dim=192)
    g = torch.Generator().manual_seed(0)
    imgs = torch.randn(64, 5, 64, 64, generator=g).cuda()      # 320 / 1088 token rows: the grouped launches apply
    results = []
    for direct, graph in (("0", True), ("1", True), ("1", False)):
        monkeypatch.setenv("SKYEMB_G16_DIRECT", direct)
        eng = MAEEngine(cfg, compute_dtype=torch.bfloat16, seed=1)
        opt = FusedAdamW(eng, lr=1e-3)
        step = TrainStep(eng, opt, CosineLR(opt, 100), 64, use_graph=graph, staged=True, n_encoder_groups=4, grad_comm="bf16")
        torch.manual_seed(123)
        for _ in range(3):
            loss = step(imgs)
        torch.cuda.synchronize()
        w = [w for k, w in eng._ws.items() if k[-1] is True][-1]
        covered = sum(e - s for s, e in eng.grad_mirror_ranges(w))
        assert (covered > 0.9 * eng.store.n) == (direct == "1"), (direct, covered, eng.store.n)
        results.append((float(loss), step.g16.clone(), eng.store.p.clone()))
    for r in results[1:]:
        assert r[0] == results[0][0]
        assert torch.equal(r[1], results[0][1]) and torch.equal(r[2], results[0][2])


@pytest.mark.parametrize("size", ["tiny", "base"])
def test_norm1_backward_inside_the_weight_gradient_launch_equals_its_own_launch(size, monkeypatch):
    """Every block's norm1 backward rides in the block's grouped weight-gradient launch as side workgroups
    (skyemb_gemm_group_attach_ln_bwd; SKYEMB_LN_SIDE=0 keeps it a launch of its own): same rows per four-wave block, same
    partial-sum table -- every gradient of the step bit for bit, ViT-B/16 at B = 256 (768- and 512-wide rows) and the tiny model."""
    from sky_embeddings_amd.engine import MAEEngine
    from sky_embeddings_amd.model_config import config_for
    cfg = (config_for("base", img_size=64, patch_size=16, in_chans=5, embed_dim=768, norm_pix_loss=True, loss_fn="mse") if size == "base"
           else config_for("tiny", img_size=64, patch_size=16, in_chans=5, embed_dim=192))
    B = 256 if size == "base" else 64
    g = torch.Generator().manual_seed(3)
    imgs = torch.randn(B, 5, 64, 64, generator=g).clamp_(min=-3).cuda()
    noise = torch.rand(B, 16, generator=g).cuda()
    out = []
    for side in ("1", "0"):
        monkeypatch.setenv("SKYEMB_LN_SIDE", side)
        eng = MAEEngine(cfg, compute_dtype=torch.bfloat16, seed=0)
        loss, _, _ = eng.forward_train(imgs, 0.75, noise)
        eng.backward()
        torch.cuda.synchronize()
        w = eng._ws[(B, 4, True)]
        carried = [grp.ln_side for grp in w["wgrad_groups"].values() if grp is not None]
        assert len(carried) == cfg.depth + cfg.decoder_depth and all(c == (side == "1") for c in carried), carried
        out.append((float(loss), eng.store.g.clone()))
        del eng
        torch.cuda.empty_cache()
    assert out[0][0] == out[1][0] and torch.equal(out[0][1], out[1][1])


@pytest.mark.parametrize("size", ["tiny", "base"])
def test_prefetch_hints_change_no_bit_of_the_step(size, monkeypatch):
    """Every GEMM launch of the step names the next GEMM's weights as a prefetch hint (engine._pf, skyemb_gemm_args.prefetch);
    SKYEMB_PREFETCH=0 builds the same launches without it: loss and every gradient bit for bit, ViT-B/16 at B = 256 and the tiny model."""
    from sky_embeddings_amd.engine import MAEEngine
    from sky_embeddings_amd.model_config import config_for
    cfg = (config_for("base", img_size=64, patch_size=16, in_chans=5, embed_dim=768, norm_pix_loss=True, loss_fn="mse") if size == "base"
           else config_for("tiny", img_size=64, patch_size=16, in_chans=5, embed_dim=192))
    B = 256 if size == "base" else 64
    g = torch.Generator().manual_seed(6)
    imgs = torch.randn(B, 5, 64, 64, generator=g).clamp_(min=-3).cuda()
    noise = torch.rand(B, 16, generator=g).cuda()
    out = []
    for hints in ("1", "0"):
        monkeypatch.setenv("SKYEMB_PREFETCH", hints)
        eng = MAEEngine(cfg, compute_dtype=torch.bfloat16, seed=0)
        assert (eng._pf("fwd", "blocks.0.mlp.fc2.weight").data_ptr() == eng.store.lp("blocks.1.attn.qkv.weight").data_ptr()
                and eng._pf("bwd", "blocks.1.attn.qkv.weight").data_ptr() == eng.store.lp("blocks.0.mlp.fc2.weight").data_ptr()
                and eng._pf("fwd", "decoder_pred.weight") is None and eng._pf("bwd", "patch_embed.proj.weight") is None)
        loss, pred, _ = eng.forward_train(imgs, 0.75, noise)
        eng.backward()
        torch.cuda.synchronize()
        out.append((float(loss), pred.clone(), eng.store.g.clone()))
        del eng
        torch.cuda.empty_cache()
    assert out[0][0] == out[1][0] and torch.equal(out[0][1], out[1][1]) and torch.equal(out[0][2], out[1][2])


@pytest.mark.parametrize("size", ["tiny", "base"])
def test_single_weight_gradients_folded_into_the_grouped_launches(size, monkeypatch):
    """decoder_pred's and decoder_embed's weight gradients ride as fifth problems in the first decoder / encoder block's grouped
    launch (engine._extra_wgrad_layers; SKYEMB_FOLD_WGRADS=0 keeps their own split-K launches).  The two forms add the same fp32
    products in a different order: those four tensors agree to 2e-6 of their norm, every other gradient bit for bit."""
    from sky_embeddings_amd.engine import MAEEngine
    from sky_embeddings_amd.model_config import config_for
    cfg = (config_for("base", img_size=64, patch_size=16, in_chans=5, embed_dim=768, norm_pix_loss=True, loss_fn="mse") if size == "base"
           else config_for("tiny", img_size=64, patch_size=16, in_chans=5, embed_dim=192))
    B = 256 if size == "base" else 64
    g = torch.Generator().manual_seed(4)
    imgs = torch.randn(B, 5, 64, 64, generator=g).clamp_(min=-3).cuda()
    noise = torch.rand(B, 16, generator=g).cuda()
    out = []
    for fold in ("1", "0"):
        monkeypatch.setenv("SKYEMB_FOLD_WGRADS", fold)
        eng = MAEEngine(cfg, compute_dtype=torch.bfloat16, seed=0)
        loss, _, _ = eng.forward_train(imgs, 0.75, noise)
        eng.backward()
        torch.cuda.synchronize()
        w = eng._ws[(B, 4, True)]
        assert w.get("folded_wgrads", set()) == ({"decoder_pred", "decoder_embed"} if fold == "1" else set())
        out.append((float(loss), {k: eng.store.grad(k).clone() for k in eng.store.offsets}))
        del eng
        torch.cuda.empty_cache()
    assert out[0][0] == out[1][0]
    for k, a in out[0][1].items():
        b = out[1][1][k]
        if k.split(".")[0] in ("decoder_pred", "decoder_embed"):
            assert float((a - b).norm() / b.norm()) < 2e-6, k
        else:
            assert torch.equal(a, b), k


@pytest.mark.parametrize("side", [False, True])
def test_fused_adamw_step_at_the_benchmark_size(side):
    """The same bit-equality at BASELINE configs[1] (ViT-B/16, B = 256): the 128x128 (encoder) and 128x64 (decoder) grouped
    launches with the optimiser step in their epilogue (side = False) or as the side job of the following block's launch
    (side = True: the default) against the separate AdamW launch, four steps, no host sync in between."""
    from sky_embeddings_amd.engine import MAEEngine
    from sky_embeddings_amd.model_config import config_for
    from sky_embeddings_amd.optim import CosineLR, FusedAdamW
    from sky_embeddings_amd.train_step import TrainStep
    cfg = config_for("base", img_size=64, patch_size=16, in_chans=5, embed_dim=768, norm_pix_loss=True, loss_fn="mse")
    imgs = torch.randn(256, 5, 64, 64, generator=torch.Generator().manual_seed(0)).clamp_(min=-3).cuda()
    res = []
    for fused in (False, True):
        eng = MAEEngine(cfg, compute_dtype=torch.bfloat16, seed=0)
        opt = FusedAdamW(eng, lr=1e-4, weight_decay=0.05)
        step = TrainStep(eng, opt, CosineLR(opt, 1000), 256, fused_adamw=fused, adamw_side=side)
        assert not fused or step.adamw_side == ("1" if side else "0")
        torch.manual_seed(5)
        ls = [step(imgs).clone() for _ in range(4)]
        torch.cuda.synchronize()
        res.append(([float(l) for l in ls], eng.store.p.clone(), eng.store.m.clone(), eng.store.v.clone(), eng.store.p_lp.clone()))
        if fused:
            assert sum(e - s_ for s_, e in step._rest_ranges) < 0.06 * eng.store.n     # 94 % of the parameters step in the GEMM epilogues
        del step, opt, eng
        torch.cuda.empty_cache()
    assert res[0][0] == res[1][0]
    for k in range(1, 5):
        assert torch.equal(res[0][k], res[1][k]), k


@pytest.mark.parametrize("side", [False, True])
@pytest.mark.parametrize("graph", [True, False])
def test_fused_adamw_step_equals_the_separate_optimiser_launch(graph, side):
    """TrainStep(fused_adamw=True): the AdamW step of every transformer block's weight matrices runs in the epilogue of the
    block's grouped weight-gradient launch.  Parameters, both moments and the bf16 shadow are bit-identical to the schedule
    with the separate optimiser launch after ten steps; engine.backward() outside the step still stores plain gradients."""
    from sky_embeddings_amd.engine import MAEEngine
    from sky_embeddings_amd.model_config import config_for
    from sky_embeddings_amd.optim import CosineLR, FusedAdamW
    from sky_embeddings_amd.train_step import TrainStep
    cfg = config_for("tiny", img_size=64, patch_size=16, in_chans=5, embed_dim=192)
    g = torch.Generator().manual_seed(0)
    imgs = torch.randn(64, 5, 64, 64, generator=g).cuda()       # 320 / 1088 token rows: multiples of 64, so the grouped launches exist
    out = []
    for fused in (False, True):
        eng = MAEEngine(cfg, compute_dtype=torch.bfloat16, seed=1)
        opt = FusedAdamW(eng, lr=1e-3, weight_decay=0.05)
        step = TrainStep(eng, opt, CosineLR(opt, 100), 64, use_graph=graph, fused_adamw=fused, adamw_side=side)
        assert step.fused_adamw == fused
        torch.manual_seed(123)
        # ten steps WITHOUT a host sync in between: the host runs ahead of the GPU, so step t's launches must still read step t's
        # scalars (a single pinned staging slot handed them step t + 1's)
        dev_losses = [step(imgs).clone() for _ in range(10)]
        torch.cuda.synchronize()
        losses = [float(l) for l in dev_losses]
        st = eng.store
        out.append((losses, st.p.clone(), st.m.clone(), st.v.clone(), st.p_lp.clone(), opt.step_count, opt))
        if fused:
            # the fused tensors moved (their gradients were never stored), the optimiser counted three steps
            assert opt.step_count == 10 and len(step._rest_ranges) >= 2
            assert sum(e - s_ for s_, e in step._rest_ranges) < 0.2 * st.n          # the block weights are the bulk
            # plain backward outside the step: gradients of a fused tensor are written again
            st.g.fill_(float("nan"))
            eng.forward_train(imgs, 0.75, torch.rand(64, 16, device="cuda"))
            eng.backward()
            assert bool(torch.isfinite(st.grad("blocks.0.attn.qkv.weight")).all())
            # ... and an eager optimizer.step() after the graph-mode steps uses ITS step's scalars (they live in a device buffer
            # now): identical to the same step on the engine that never fused
            opt.step()
            out[0][-1].step()
            torch.cuda.synchronize()
            # (both engines now hold step 11 applied to their own -- different -- gradients: compare the bias correction through
            # a tensor whose gradient is the same on both: none is, so check the scalars themselves)
            lr, bc1, bc2 = opt.step_scalars(opt.step_count)
            got = opt.hyper_device.cpu().numpy()
            assert abs(got[0] - lr) <= 1e-12 + 1e-7 * lr and abs(got[1] - bc1) <= 1e-7 and abs(got[2] - bc2) <= 1e-7
    a, b = out
    assert a[0] == b[0] and a[5] == b[5]
    for k in range(1, 5):
        assert torch.equal(a[k], b[k]), k
