"""CPU: the oracle (oracle/*.py, oracle/topk_oracle.c) against vectors captured from the reference."""
import numpy as np
import pytest
import torch

from oracle import mae_oracle as mo
from oracle import similarity_oracle as so
from tests.helpers import GOLDEN, load_case, load_simmim_case, rel_err

CASES = ["mae_tiny_A", "mae_tiny_B_nan", "mae_tiny_C_nonorm", "mae_tiny_D_l1", "mae_tiny_E_p8"]


@pytest.mark.parametrize("name", CASES)
def test_forward_backward_matches_reference(name):
    z, cfg, st = load_case(name)
    # state layout == reference state_dict (names, shapes, order)
    ref_keys = [k[len("state/"):] for k in z.files if k.startswith("state/")]
    assert ref_keys == [n for n, _ in mo.state_layout(cfg)]
    imgs, noise = torch.from_numpy(z["imgs"]), torch.from_numpy(z["noise"])
    loss, pred, mask, ids, latent, grads = mo.loss_and_grads(st, imgs, cfg, float(z["mask_ratio"]), noise)
    assert np.array_equal(mask.numpy(), z["mask"])
    assert np.array_equal(ids.numpy(), z["ids_restore"])
    assert abs(float(loss) - float(z["loss"])) <= 2e-6 * abs(float(z["loss"]))
    assert rel_err(pred.numpy(), z["pred"]) < 2e-6
    nan_input = bool(np.isnan(z["imgs"]).any())
    poisoned = nan_input and cfg.loss_fn == "mse"

    def check(gr):
        for k, g in gr.items():
            ref = z["grad/" + k]
            tol = 2e-5 * max(np.abs(ref).max(), 1e-8) + 1e-9
            assert np.abs(g.numpy() - ref).max() <= tol, k

    if poisoned:
        # reference behaviour pinned: with MSE, NaN targets poison the whole backward
        # (forward_loss docstring); L1's backward is sign-based and stays finite.
        for k, g in grads.items():
            assert np.array_equal(np.isnan(g.numpy()), np.isnan(z["grad/" + k])), k
        assert np.isnan(z["grad/cls_token"]).all() and np.isnan(z["grad/blocks.0.mlp.fc1.weight"]).all()
    else:
        check(grads)
    if nan_input:
        loss2, pred2, _, _, _, g2 = mo.loss_and_grads(st, imgs, cfg, float(z["mask_ratio"]), noise, nan_safe=True)
        assert float(loss2) == float(loss) and np.array_equal(pred2.numpy(), pred.numpy())
        assert all(bool(torch.isfinite(v).all()) for v in g2.values())
        if not poisoned:
            check(g2)  # nan_safe == the reference wherever the reference is finite
    # encoder-only path (mask_ratio=0 keeps every token, shuffled)
    lat, _, ids0 = mo.forward_features(st, imgs, cfg, 0.0, noise, reshape_out=False)
    assert np.array_equal(ids0.numpy(), z["ids_restore_full"])
    assert rel_err(lat.numpy(), z["latent_full"]) < 2e-6


@pytest.mark.parametrize("name", ["simmim_tiny_F_l1_nan", "simmim_tiny_G_mse", "simmim_tiny_H_radec", "simmim_tiny_J_attnpool"])
def test_simmim_mode_matches_reference(name):
    """SimMIM mode (per-channel pixel masks, all tokens encoded, Conv1x1 + PixelShuffle head, pixel loss), the
    RA/Dec token (spherical harmonics -> Siren -> linear) and the attention-pooled variant (case J: one pooled token per image,
    head up-samples to the image), forward / backward / three AdamW steps."""
    z, cfg, st, imgs, pmask, ra_dec = load_simmim_case(name)
    assert [k[len("state/"):] for k in z.files if k.startswith("state/")] == [n for n, _ in mo.state_layout(cfg)]
    loss, pred, _, _, latent, grads = mo.loss_and_grads(st, imgs, cfg, mask=pmask, ra_dec=ra_dec)
    assert abs(float(loss) - float(z["loss"])) <= 2e-6 * abs(float(z["loss"]))
    assert rel_err(pred.numpy(), z["pred"]) < 2e-6
    lat, _, _ = mo.forward_features(st, imgs, cfg, mask=pmask, ra_dec=ra_dec, reshape_out=False)
    assert rel_err(lat.numpy(), z["latent"]) < 2e-6
    for k, g in grads.items():
        ref = z["grad/" + k]
        assert np.abs(g.numpy() - ref).max() <= 2e-5 * max(np.abs(ref).max(), 1e-8) + 1e-9, k
    if ra_dec is not None:
        assert rel_err(mo.spherical_harmonics(ra_dec).numpy(), z["sh_features"]) < 1e-6
        assert rel_err(mo.location_encoder(st, ra_dec).numpy(), z["ra_dec_token"]) < 1e-6
    # nan_safe (what the HIP path implements) == the reference here: SimMIM zeroes NaN targets through its mask
    loss2, _, _, _, _, g2 = mo.loss_and_grads(st, imgs, cfg, mask=pmask, ra_dec=ra_dec, nan_safe=True)
    assert float(loss2) == float(loss)
    for k in grads:
        assert np.abs(g2[k].numpy() - z["grad/" + k]).max() <= 2e-5 * max(np.abs(z["grad/" + k]).max(), 1e-8) + 1e-9, k
    if "step_losses" in z.files:
        init_lr, wd, total, flf = [float(v) for v in z["opt_hparams"]]
        tr = mo.Trainer(cfg, st, init_lr=init_lr, weight_decay=wd, total_iters=int(total), final_lr_factor=flf)
        for it in range(3):
            l, *_ = tr.step(imgs, mask=pmask, ra_dec=ra_dec)
            assert abs(float(l) - float(z["step_losses"][it])) <= 1e-5 * abs(float(z["step_losses"][it]))
        for k in tr.decay + tr.no_decay:
            ref = z[f"state_after3/{k}"]
            assert np.abs(st[k].numpy() - ref).max() <= 3e-6 * max(np.abs(ref).max(), 1e-3) + 5e-3 * init_lr, k
        assert np.array_equal(st["mask_token"].numpy(), z["state_after3/mask_token"])   # never touched by the optimiser


def test_adamw_cosine_steps_match_reference():
    z, cfg, st = load_case("mae_tiny_A")
    init_lr, wd, total, flf = [float(v) for v in z["opt_hparams"]]
    tr = mo.Trainer(cfg, st, init_lr=init_lr, weight_decay=wd, total_iters=int(total), final_lr_factor=flf)
    imgs = torch.from_numpy(z["imgs"])
    for it in range(3):
        assert abs(tr.lr() - float(z["step_lrs"][it])) <= 1e-12 * init_lr + 1e-18
        loss, *_ = tr.step(imgs, float(z["mask_ratio"]), torch.from_numpy(z["step_noises"][it]))
        assert abs(float(loss) - float(z["step_losses"][it])) <= 5e-6 * abs(float(z["step_losses"][it]))
        if it in (0, 2):
            for k in tr.decay + tr.no_decay:
                ref = z[f"state_after{it + 1}/{k}"]
                # Adam turns rounding noise on ~zero gradients (e.g. the key bias, to which softmax is
                # invariant) into +-lr-sized moves, so allow a small fraction of lr absolute.
                tol = 3e-6 * max(np.abs(ref).max(), 1e-3) + (0 if it == 0 else 5e-3 * init_lr)
                assert np.abs(st[k].numpy() - ref).max() <= tol, (it, k)
    # timm param_groups_weight_decay split (86 / 167 for MAE-B per SURVEY §8a a11)
    d, nd = mo.weight_decay_split(mo.config_for("base", patch_size=16, in_chans=5))
    assert (len(d), len(nd)) == (86, 167)


def test_unit_pieces():
    z = np.load(f"{GOLDEN}/unit_pieces.npz")
    for D in (64, 512, 768, 1024):
        for grid in (4, 8):
            for rd in (False, True):
                got = mo.sincos_pos_embed(D, grid, True, rd)
                assert np.array_equal(got, z[f"sincos/{D}_{grid}_{int(rd)}"])
    cfg = mo.MAEConfig(img_size=32, patch_size=8, in_chans=3)
    x = torch.from_numpy(z["patchify/in"])
    pt = mo.patchify(x, cfg)
    assert np.array_equal(pt.numpy(), z["patchify/out"])
    assert np.array_equal(mo.unpatchify(pt, cfg).numpy(), z["patchify/roundtrip"])
    mean, var = mo.patch_mean_and_var(torch.from_numpy(z["pmv/in"]))
    assert np.array_equal(mean.numpy(), z["pmv/mean"]) and np.array_equal(var.numpy(), z["pmv/var"])
    for L in (16, 64):
        for ratio in (0.0, 0.6, 0.75):
            key = f"mask/{L}_{ratio}"
            tok = torch.arange(3 * L * 2, dtype=torch.float32).reshape(3, L, 2)
            xm, mask, ids = mo.random_masking_from_noise(tok, ratio, torch.from_numpy(z[key + "/noise"]))
            assert np.array_equal(xm.numpy(), z[key + "/x_masked"])
            assert np.array_equal(mask.numpy(), z[key + "/mask"])
            assert np.array_equal(ids.numpy(), z[key + "/ids_restore"])


def test_similarity_matches_reference():
    z = np.load(f"{GOLDEN}/similarity.npz")
    for (T, P, N) in ((130, 1, 512), (65, 16, 128), (65, 64, 64)):
        key = f"sim/{T}_{P}_{N}"
        tgt, tst = torch.from_numpy(z[key + "/target"]), torch.from_numpy(z[key + "/test"])
        avg, w = so.determine_target_features(tgt)
        assert np.array_equal(avg.numpy(), z[key + "/avg"]) and np.array_equal(w.numpy(), z[key + "/w"])
        for metric in ("cosine", "MSE", "MAE"):
            for combine in ("min", "mean", "max"):
                for uw in (True, False):
                    s = so.compute_similarity(tgt, tst, metric=metric, combine=combine, use_weights=uw)
                    assert np.array_equal(s.numpy(), z[f"{key}/{metric}_{combine}_{int(uw)}"])
    # streaming best-n == reference's streaming result (scores and sample tags)
    scores = torch.from_numpy(z["stream/scores"])
    for metric in ("cosine", "MSE"):
        bs = torch.full((50,), float("-inf") if metric == "cosine" else float("inf"))
        bt = torch.full((50,), -1, dtype=torch.int64)
        for b in range(8):
            sl = slice(b * 64, (b + 1) * 64)
            bs, bt = so.update_best_scores(scores[sl], torch.arange(b * 64, (b + 1) * 64), bs, bt, 50, metric)
        assert np.array_equal(bs.numpy(), z[f"stream/{metric}_best_scores"])
        assert np.array_equal(bt.numpy(), z[f"stream/{metric}_best_idx"])
    mu, sd = so.standardise_first_batch(torch.from_numpy(z["std/in"]))
    assert np.array_equal(mu.numpy(), z["std/mu"]) and np.array_equal(sd.numpy(), z["std/sd"])
    got = so.standardise_np(z["std/in"].reshape(-1, 96), z["std/mu"], z["std/sd"])
    assert np.array_equal(got, z["std/out"].reshape(-1, 96))


def test_similarity_central_patches_matches_reference():
    z = np.load(f"{GOLDEN}/similarity_central.npz")
    keys = sorted({k.rsplit("/", 1)[0] for k in z.files if k.endswith("/target")})
    assert len(keys) == 3
    for key in keys:
        n = int(key.rsplit("_", 1)[1])
        tgt, tst = torch.from_numpy(z[key + "/target"]), torch.from_numpy(z[key + "/test"])
        for metric in ("cosine", "MSE", "MAE"):
            for combine in ("min", "mean", "max"):
                s = so.compute_similarity(tgt, tst, metric=metric, combine=combine, n_central_patches=n)
                assert np.array_equal(s.numpy(), z[f"{key}/{metric}_{combine}"]), (key, metric, combine)


def test_c_topk_oracle_against_reference_formula():
    """The fixed-order C contract stays within fp32 rounding of the reference's torch formula and
    its top-k equals a stable sort of its own scores."""
    z = np.load(f"{GOLDEN}/similarity.npz")
    tgt, tst = torch.from_numpy(z["sim/130_1_512/target"]), torch.from_numpy(z["sim/130_1_512/test"])
    avg, w = so.determine_target_features(tgt)
    ref = z["sim/130_1_512/cosine_min_1"]  # P == 1 so 'min' is the identity
    sc = so.cosine_scores_np(avg[None].numpy(), tst[:, 0].numpy(), w.numpy())
    assert np.abs(sc[0] - ref).max() < 5e-7
    s, i = so.cosine_topk_np(avg[None].numpy(), tst[:, 0].numpy(), 20, w.numpy())
    order = np.argsort(-sc[0], kind="stable")[:20]
    assert np.array_equal(i[0], order) and np.array_equal(s[0], sc[0][order])
    # exact ties -> lower index first; k > N pads with (-inf, -1)
    q = np.ones((1, 8), np.float32)
    bank = np.ones((5, 8), np.float32)
    bank[3] = -1
    s, i = so.cosine_topk_np(q, bank, 7)
    assert i[0].tolist() == [0, 1, 2, 4, 3, -1, -1]
    assert np.isneginf(s[0][5:]).all()


def test_mask_generator_oracle_matches_reference_counts():
    """tests/golden/maskgen.npz holds masks drawn by the reference's own MaskGenerator (utils/dataloaders.py:155-219) and the
    ratio draw behind each: the oracle driven by that draw must mask the same number of patches in every channel, and the
    reference's masks must be whole patches, independent per channel."""
    z = np.load(GOLDEN + "/maskgen.npz")
    keys = sorted({k.rsplit("/", 1)[0] for k in z.files})
    assert len(keys) == 3
    for key in keys:
        size, p, C, mx = key.split("/")[1].split("_")
        size, p, C, mx = int(size), int(p), int(C), float(mx)
        u, masks = torch.from_numpy(z[key + "/u"]), torch.from_numpy(z[key + "/masks"]).float()
        L = (size // p) ** 2
        assert masks.shape == (len(u), C, size, size)
        ours = mo.simmim_mask_from_noise(torch.rand(len(u), C, L, generator=torch.Generator().manual_seed(1)), u, mx, p)
        patches = masks.view(len(u), C, size // p, p, size // p, p)
        assert torch.equal(patches, patches[:, :, :, :1, :, :1].expand_as(patches))           # whole patches only
        ref_count = masks[:, :, ::p, ::p].sum(dim=(2, 3))
        assert torch.equal(ours[:, :, ::p, ::p].sum(dim=(2, 3)), ref_count)
        assert bool((ref_count == ref_count[:, :1]).all())                                     # same count in every channel
        differs = (masks[:, 0] != masks[:, 1]).flatten(1).any(dim=1)
        assert bool(differs[ref_count[:, 0] > 0].float().mean() > 0.8)                         # channels masked independently


def test_block_arithmetic_against_an_independent_implementation():
    """The transformer block the oracle restates (timm's `Block`, which is not installed: SURVEY 8c stand-in contract) against an
    INDEPENDENT published implementation of the same pre-norm ViT layer that is installed: Hugging Face `transformers`
    `ViTLayer` (separate q / k / v projections, eager attention, exact-erf GELU, LayerNorm eps from the config).  Same weights,
    same input: outputs AND input / weight gradients agree to fp32 rounding.  Pins the block's arithmetic (LayerNorm placement,
    head split of the fused qkv, score scaling, softmax axis, residuals, GELU flavour) to third-party code; the timm wiring
    around it (cls token, pos-embed, final norm) stays pinned by the reference-generated goldens only."""
    tr = pytest.importorskip("transformers")
    from transformers import ViTConfig
    from transformers.models.vit.modeling_vit import ViTLayer
    D, H, hidden, eps = 96, 6, 384, 1e-6
    cfg = ViTConfig(hidden_size=D, num_attention_heads=H, intermediate_size=hidden, hidden_act="gelu", layer_norm_eps=eps,
                    qkv_bias=True, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    cfg._attn_implementation = "eager"
    layer = ViTLayer(cfg).eval()
    g = torch.Generator().manual_seed(11)
    st = {}
    for name, shape in (("norm1.weight", (D,)), ("norm1.bias", (D,)), ("attn.qkv.weight", (3 * D, D)), ("attn.qkv.bias", (3 * D,)),
                        ("attn.proj.weight", (D, D)), ("attn.proj.bias", (D,)), ("norm2.weight", (D,)), ("norm2.bias", (D,)),
                        ("mlp.fc1.weight", (hidden, D)), ("mlp.fc1.bias", (hidden,)), ("mlp.fc2.weight", (D, hidden)),
                        ("mlp.fc2.bias", (D,))):
        t = torch.randn(*shape, generator=g) * (0.2 if name.endswith("weight") and len(shape) == 2 else 0.5)
        if name in ("norm1.weight", "norm2.weight"):
            t = 1.0 + 0.3 * t
        st[f"b.{name}"] = t.requires_grad_(True)
    names = dict(layer.named_parameters())
    if "attention.q_proj.weight" in names:                      # transformers >= 5
        qkv = ("attention.q_proj", "attention.k_proj", "attention.v_proj")
        proj, fc1, fc2 = "attention.o_proj", "mlp.fc1", "mlp.fc2"
    else:                                                       # transformers 4.x module names
        qkv = ("attention.attention.query", "attention.attention.key", "attention.attention.value")
        proj, fc1, fc2 = "attention.output.dense", "intermediate.dense", "output.dense"
    with torch.no_grad():
        for i, mod in enumerate(qkv):                           # the fused qkv rows are [q | k | v], each [heads x head_dim]
            names[f"{mod}.weight"].copy_(st["b.attn.qkv.weight"][i * D:(i + 1) * D])
            names[f"{mod}.bias"].copy_(st["b.attn.qkv.bias"][i * D:(i + 1) * D])
        for mod, key in ((proj, "attn.proj"), (fc1, "mlp.fc1"), (fc2, "mlp.fc2"), ("layernorm_before", "norm1"), ("layernorm_after", "norm2")):
            names[f"{mod}.weight"].copy_(st[f"b.{key}.weight"])
            names[f"{mod}.bias"].copy_(st[f"b.{key}.bias"])
    x = torch.randn(3, 17, D, generator=g)
    dy = torch.randn(3, 17, D, generator=g)
    xo = x.clone().requires_grad_(True)
    yo = mo.block(xo, st, "b", H, eps)
    yo.backward(dy)
    xh = x.clone().requires_grad_(True)
    yh = layer(xh)
    yh = yh[0] if isinstance(yh, tuple) else yh
    yh.backward(dy)
    assert rel_err(yo.detach().numpy(), yh.detach().numpy()) < 2e-6
    assert rel_err(xo.grad.numpy(), xh.grad.numpy()) < 5e-6
    assert rel_err(st["b.mlp.fc2.weight"].grad.numpy(), names[f"{fc2}.weight"].grad.numpy()) < 5e-6
    assert rel_err(st["b.norm1.weight"].grad.numpy(), names["layernorm_before.weight"].grad.numpy()) < 5e-6
    gq = torch.cat([names[f"{m}.weight"].grad for m in qkv])
    assert rel_err(st["b.attn.qkv.weight"].grad.numpy(), gq.numpy()) < 5e-6
