"""GPU: the downstream predictor (utils.vit + utils.predictor_training_fns mirrors) against goldens made by the reference's own
``VisionTransformer`` / ``run_iter`` / ``param_groups_lrd`` / ``interpolate_pos_embed`` through the timm stand-in
(tests/golden/make_golden.py predictor): forward predictions, three optimiser steps of the linear-probe, fine-tuning (layer-wise
lr decay) and fully-supervised methods -- encoder forward AND backward in the HIP engine -- and the checkpoint surgery."""
import os
from collections import defaultdict

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = {"lp_token_ce": ("lp", "token", "crossentropy"), "ft_avg_mse": ("ft", "avg", "mse"), "fs_token_mse": ("fs", "token", "mse"),
         "lp_map_ce": ("lp", "map", "crossentropy"), "ft_map_mse": ("ft", "map", "mse"),
         # ... and with the optimiser as utils/vit.py:174-186 really leaves it (lr / 25 and beta1 = 0.95 from the discarded OneCycleLR)
         "lp_map_ce_oc": ("lp", "map", "crossentropy"), "ft_map_mse_oc": ("ft", "map", "mse")}


def build(z, case, dtype):
    from sky_embeddings_amd.model_config import MAEConfig
    from sky_embeddings_amd.utils.mim_vit import _DataParallelShim
    from sky_embeddings_amd.utils.vit import VisionTransformer
    img, patch, C, D, depth, heads, ncls = [int(v) for v in z[f"{case}/cfg"]]
    cfg = MAEConfig(img_size=img, patch_size=patch, in_chans=C, embed_dim=D, depth=depth, num_heads=heads, decoder_embed_dim=16,
                    decoder_depth=1, decoder_num_heads=2, pixel_mean=0.1, pixel_std=1.7)
    m = VisionTransformer(cfg, "cuda", dtype, num_classes=ncls, global_pool=CASES[case][1],
                          label_means=z[f"{case}/label_means"].tolist(), label_stds=z[f"{case}/label_stds"].tolist())
    pre = f"{case}/state/"
    sd = {k[len(pre):]: torch.from_numpy(z[k]) for k in z.files if k.startswith(pre)}
    assert sorted(sd) == sorted(m.state_dict())                       # the reference module's tensor names, nothing more or less
    m.load_state_dict(sd)
    return _DataParallelShim(m), m


@pytest.mark.parametrize("case", list(CASES))
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_predictor_forward_matches_reference(case, dtype):
    z = np.load(os.path.join(ROOT, "tests", "golden", "predictor.npz"))
    model, m = build(z, case, dtype)
    model.eval()
    out = model(torch.from_numpy(z[f"{case}/x"][0]).cuda()).cpu().numpy()
    ref = z[f"{case}/logits0"]
    err = float(np.abs(out - ref).max()) / max(float(np.abs(ref).max()), 1e-6)
    assert err < (2e-5 if dtype == torch.float32 else 3e-2), err


@pytest.mark.parametrize("case", list(CASES))
def test_predictor_training_steps_match_reference(case):
    """run_iter x 3 (forward, loss, backward through the HIP encoder, AdamW per parameter group, LinearLR) in the exact-fp32 mode."""
    from sky_embeddings_amd.utils.predictor_training_fns import run_iter
    from sky_embeddings_amd.utils.vit import LinearLR, build_optimizer
    z = np.load(os.path.join(ROOT, "tests", "golden", "predictor.npz"))
    method, pool, loss_fn = CASES[case]
    model, m = build(z, case, torch.float32)
    init_lr, wd, layer_decay, total, flf = [float(v) for v in z[f"{case}/hyper"]]
    opt = build_optimizer(m, method, init_lr, wd, layer_decay)
    if case.endswith("_oc"):
        from sky_embeddings_amd.utils.vit import apply_onecycle_side_effects
        apply_onecycle_side_effects(opt, [g["lr"] for g in opt.param_groups] if method == "ft" else init_lr)
        got = np.array([[g["lr"], g["initial_lr"], g["betas"][0], g["betas"][1], g["weight_decay"]] for g in opt.param_groups])
        assert got.shape == z[f"{case}/opt_groups"].shape and np.allclose(got, z[f"{case}/opt_groups"], rtol=1e-12, atol=0)
    sched = LinearLR(opt, start_factor=1.0, end_factor=1 / flf, total_iters=int(total))
    if method == "ft" and not case.endswith("_oc"):
        # utils/vit.py:141-143 as written: the groups' base lr is the configured WEIGHT DECAY, scaled per layer; decay 0.05 / 0
        lrs = sorted({round(g["initial_lr"], 12) for g in opt.param_groups})
        assert lrs == sorted({round(wd * layer_decay ** (m.num_blocks + 1 - i), 12) for i in range(m.num_blocks + 2)})
        assert {g["weight_decay"] for g in opt.param_groups} == {0.0, 0.05}
    pool_names = {k for k in m.head if k.startswith("attn_pool.")}
    assert (len(pool_names) == 13) == (pool == "map")
    if method == "lp":
        assert m.frozen_encoder and m.trainable == {"norm.weight", "norm.bias", "head.weight", "head.bias"} | pool_names
    x, labels = torch.from_numpy(z[f"{case}/x"]), torch.from_numpy(z[f"{case}/labels"])
    before = {k: v.detach().clone() for k, v in m.state_dict().items()}
    cp = defaultdict(list)
    for it in range(3):
        model, opt, sched, cp = run_iter(model, x[it].cuda(), None, None, labels[it].cuda(), opt, sched, cp, loss_fn=loss_fn, mode="train")
        assert abs(cp["train_loss"][-1] - float(z[f"{case}/train_loss"][it])) <= 5e-5 * abs(float(z[f"{case}/train_loss"][it])), (it, cp["train_loss"])
        if it in (0, 2):
            sd = m.state_dict()
            for k, v in sd.items():
                ref = z[f"{case}/step{it}/{k}"]
                lr_k = max(g["initial_lr"] for g in opt.param_groups if k in g["params"]) if any(k in g["params"] for g in opt.param_groups) else 0.0
                # Adam's first steps move every element by ~lr * sign(g): elements whose gradient is rounding noise may differ by a
                # fraction of lr (tests/test_mae_parity_gpu.py)
                tol = 5e-6 * max(float(np.abs(ref).max()), 1e-3) + 3e-2 * lr_k
                diff = np.abs(v.detach().cpu().numpy().reshape(ref.shape) - ref)
                if k.endswith("attn.qkv.bias") or k == "attn_pool.kv.bias":
                    # the KEY bias has an identically zero gradient (softmax is invariant to a shift of every score of a row): what
                    # either implementation computes for it is rounding noise, which Adam's normalisation turns into steps of the
                    # order of lr with an arbitrary sign -- bounded here, not compared
                    D = ref.shape[0] // (3 if k.endswith("qkv.bias") else 2)
                    lo = D if k.endswith("qkv.bias") else 0            # [q | k | v] and [k | v]
                    assert float(diff[lo:lo + D].max()) <= 1.5 * (it + 1) * lr_k, (it, k)
                    diff = np.concatenate([diff[:lo], diff[lo + D:]])
                assert float(diff.max()) <= tol, (it, k, float(diff.max()), tol)
    metric = cp["train_acc" if loss_fn == "crossentropy" else "train_mae"]
    assert np.allclose(metric, z[f"{case}/train_metric"], rtol=1e-4, atol=1e-6)
    assert np.allclose(sorted(sched.get_last_lr()), sorted(z[f"{case}/lr_after"].tolist()), rtol=1e-6)
    after = m.state_dict()
    moved = {k for k in after if not torch.equal(after[k], before[k])}
    if method == "lp":
        assert moved == {"norm.weight", "norm.bias", "head.weight", "head.bias"} | pool_names     # the encoder stayed frozen
    else:
        assert "pos_embed" not in moved and {"cls_token", "patch_mask_values", "patch_embed.proj.weight", "blocks.0.attn.qkv.weight",
                                              "blocks.1.mlp.fc2.bias", "head.weight"} <= moved
    # validation mode changes nothing; optimiser / scheduler state round trip
    run_iter(model, x[0].cuda(), None, None, labels[0].cuda(), opt, sched, cp, loss_fn=loss_fn, mode="val")
    assert len(cp["val_loss"]) == 1 and opt.step_count == 3
    sd_o, sd_s = opt.state_dict(), sched.state_dict()
    if case.endswith("_oc"):
        # the layout of torch.optim.AdamW / LinearLR state dicts, as the reference's predictor checkpoints hold them
        assert sorted(sd_o["state"].keys()) == z[f"{case}/opt_state_ids"].tolist()
        assert [len(g["params"]) for g in sd_o["param_groups"]] == z[f"{case}/opt_group_param_ids"].tolist()
        assert sorted(sd_o["state"][0].keys()) == z[f"{case}/opt_state_keys"].tolist() and float(sd_o["state"][0]["step"]) == float(z[f"{case}/opt_state_step"])
        assert set(z[f"{case}/sched_keys"].tolist()) <= set(sd_s.keys())
        # ... and torch's own AdamW accepts it
        tp = [torch.nn.Parameter(torch.zeros(tuple(v.shape))) for k, v in m.state_dict().items() if k in opt._names()]
        by_name = dict(zip([k for k in m.state_dict() if k in opt._names()], tp))
        topt = torch.optim.AdamW([{"params": [by_name[n] for n in g["params"]]} for g in opt.param_groups])
        topt.load_state_dict({k: v for k, v in sd_o.items() if k in ("state", "param_groups")})
        assert topt.param_groups[0]["betas"][0] == 0.95 and torch.equal(topt.state[by_name[opt._names()[0]]]["exp_avg"], sd_o["state"][0]["exp_avg"])
    opt2 = build_optimizer(m, method, init_lr, wd, layer_decay)
    sched2 = LinearLR(opt2, start_factor=1.0, end_factor=1 / flf, total_iters=int(total))
    opt2.load_state_dict(sd_o)
    sched2.load_state_dict(sd_s)
    assert opt2.step_count == 3 and sched2.get_last_lr() == sched.get_last_lr()


def test_predictor_bf16_training_step_runs_and_learns():
    from sky_embeddings_amd.utils.predictor_training_fns import run_iter
    from sky_embeddings_amd.utils.vit import LinearLR, build_optimizer
    z = np.load(os.path.join(ROOT, "tests", "golden", "predictor.npz"))
    case = "fs_token_mse"
    model, m = build(z, case, torch.bfloat16)
    opt = build_optimizer(m, "fs", 2e-3, 0.03, 0.7)
    sched = LinearLR(opt, 1.0, 0.01, 50)
    x, labels = torch.from_numpy(z[f"{case}/x"][0]).cuda(), torch.from_numpy(z[f"{case}/labels"][0]).cuda()
    cp = defaultdict(list)
    for _ in range(12):
        run_iter(model, x, None, None, labels, opt, sched, cp, loss_fn="mse", mode="train")
    assert np.isfinite(cp["train_loss"]).all() and cp["train_loss"][-1] < 0.5 * cp["train_loss"][0]


@pytest.mark.parametrize("case", ["fs_token_mse", "ft_avg_mse", "ft_map_mse"])
def test_predictor_head_dropout(case):
    """ARCHITECTURE.dropout (utils/vit.py:40 -> timm's `drop_rate` = dropout on the pooled features in front of the classifier): off in
    eval mode (same predictions as without it), on in training mode -- the predictions are the classifier applied to the kept, rescaled
    features and the gradient reaching the features carries the same mask -- for the three pooling modes.  The keep mask comes from
    torch's device generator: the reference's individual draws cannot be reproduced, the operation can."""
    z = np.load(os.path.join(ROOT, "tests", "golden", "predictor.npz"))
    _, m = build(z, case, torch.float32)
    x = torch.from_numpy(z[f"{case}/x"][0]).cuda()
    m.eval()
    with torch.no_grad():
        ref = m(x).clone()
    m.drop_rate = 0.5
    m.eval()
    with torch.no_grad():
        assert torch.equal(m(x), ref)                                  # eval: no dropout
    m.train(True)
    hd = m._head_mod
    pred = m(x)
    B = x.shape[0]
    w = hd._ws[next(k for k in hd._ws if k[0] == B)]
    mask = w["drop"].clone()
    assert mask is not None and set(mask.unique().tolist()) <= {0.0, 2.0} and 0.3 < float((mask > 0).float().mean()) < 0.7
    W, b = hd.tensors["head.weight"], hd.tensors["head.bias"]
    want = w["z"] @ W.t() + b                                          # w["z"]: the features AFTER the mask
    assert float((pred.detach() - want).abs().max()) < 1e-4 * max(1.0, float(want.abs().max()))
    assert bool((w["z"][mask == 0] == 0).all())
    pred.sum().backward()
    torch.cuda.synchronize()
    assert bool((w["gz"][mask == 0] == 0).all()) and bool(torch.isfinite(w["gz"]).all())
    m.train(False)


def test_checkpoint_surgery_matches_reference(tmp_path):
    """interpolate_pos_embed / crop_pos_embed on the reference's inputs, and load_model from an MAE checkpoint of another size."""
    import configparser
    from sky_embeddings_amd.utils import pos_embed as pe
    z = np.load(os.path.join(ROOT, "tests", "golden", "predictor.npz"))

    class Stub:
        def __init__(self, n_patches, n_tokens, D):
            self.patch_embed = type("P", (), {"num_patches": n_patches})()
            self.pos_embed = torch.zeros(1, n_tokens, D)
    ck = {"pos_embed": torch.from_numpy(z["surgery/pos_embed_in"])}
    pe.interpolate_pos_embed(Stub(36, 37, ck["pos_embed"].shape[-1]), ck)
    assert np.allclose(ck["pos_embed"].numpy(), z["surgery/pos_embed_out"], rtol=1e-6, atol=1e-7)
    ck = {"pos_embed": torch.from_numpy(z["surgery/crop_in"])}
    pe.crop_pos_embed(Stub(16, 17, ck["pos_embed"].shape[-1]), ck)
    assert np.array_equal(ck["pos_embed"].numpy(), z["surgery/crop_out"])
    # build_model: an MAE checkpoint trained at 64x64 feeds a predictor at 32x32 (pos_embed 16 + 1 -> 4 + 1 rows, bicubic)
    from sky_embeddings_amd.utils.mim_vit import build_model as build_mae
    from sky_embeddings_amd.utils.vit import build_model
    mae_cfg = configparser.ConfigParser()
    mae_cfg.read(os.path.join(ROOT, "configs", "mim_1.ini"))
    mae_cfg["TRAINING"]["compute_dtype"] = "f32"
    mae, _, _ = build_mae(mae_cfg, str(tmp_path / "none.pth.tar"), torch.device("cuda"))
    mae_file = str(tmp_path / "mae.pth.tar")
    torch.save({"batch_iters": 3, "losses": {}, "model": {k: v.cpu() for k, v in mae.module.state_dict().items()}}, mae_file)
    cfg = configparser.ConfigParser()
    cfg["ARCHITECTURE"] = {"img_size": "32", "global_pool": "token", "dropout": "0.0"}
    cfg["DATA"] = {"num_classes": "3", "label_means": "[0]", "label_stds": "[1]"}
    cfg["TRAINING"] = {"total_batch_iters": "20", "init_lr": "1e-3", "weight_decay": "0.05", "final_lr_factor": "100", "train_method": "ft",
                       "layer_decay": "0.75", "use_label_errs": "False"}
    model, losses, cur_iter, opt, sched = build_model(cfg, mae_cfg, str(tmp_path / "pred.pth.tar"), mae_file, torch.device("cuda"),
                                                      build_optimizer=True)
    sd = model.module.state_dict()
    assert cur_iter == 1 and sd["pos_embed"].shape == (1, 5, int(mae_cfg["ARCHITECTURE"]["embed_dim"])) and sd["head.weight"].shape[0] == 3
    assert torch.equal(sd["blocks.0.attn.qkv.weight"].cpu(), mae.module.state_dict()["blocks.0.attn.qkv.weight"].cpu())
    assert float(sd["head.weight"].abs().max()) < 1e-3                # trunc_normal_(std = 2e-5)
    ref_table = {"pos_embed": mae.module.state_dict()["pos_embed"].cpu().clone()}
    pe.interpolate_pos_embed(Stub(4, 5, sd["pos_embed"].shape[-1]), ref_table)
    assert torch.allclose(sd["pos_embed"].cpu(), ref_table["pos_embed"])
    out = model(torch.randn(2, 5, 32, 32).cuda())
    assert out.shape == (2, 3) and bool(torch.isfinite(out).all())
