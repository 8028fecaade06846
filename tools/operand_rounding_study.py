"""Which GEMM operand rounding produces the bf16 mode's 6e-3 pixel error, and which operand format lands under north_star's
1e-3?  CPU only (the oracle with its test-only OPERAND_HOOK): config A, B = 32 slice of bench.py's synthetic batch, seed-0
reference init -- the same sample as bench.py's `parity` leg.

Every contraction of the path (linear layers, patch embedding, the two attention products) is replaced by a product of
ROUNDED operands with fp32 accumulation, forward and backward (dx = r(dy) r(W), dW = r(dy)^T r(x)), per operand class:
activation-class operands (LayerNorm / GELU / attention outputs, q, k, v, P, every incoming gradient) and weight-class ones.

formats: f32 | bf16 | f16 | bf16x2 (hi + lo, both bf16: ~16 mantissa bits) | f16x2 (~22 bits) | bf16x3
`gscale`: gradients are multiplied by it before rounding and divided after (a static loss scale: fp16 gradients underflow
without one; a power of two changes nothing else).

    python tools/operand_rounding_study.py [--out profiles/r06_operand_rounding.json]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import mae_oracle as mo  # noqa: E402


def rnd(x, fmt):
    if fmt == "f32":
        return x
    if fmt == "bf16":
        return x.bfloat16().float()
    if fmt == "f16":
        return x.half().float()
    base = {"bf16x2": (torch.bfloat16, 2), "f16x2": (torch.float16, 2), "bf16x3": (torch.bfloat16, 3)}[fmt]
    out = torch.zeros_like(x)
    rest = x
    for _ in range(base[1]):
        piece = rest.to(base[0]).float()
        out = out + piece
        rest = rest - piece
    return out


class RoundedMM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, kind, fa, fw, gscale):
        fb = fw if kind == "aw" else fa
        ctx.save_for_backward(a, b)
        ctx.cfg = (kind, fa, fw, gscale)
        return rnd(a, fa) @ rnd(b, fb)

    @staticmethod
    def backward(ctx, dy):
        a, b = ctx.saved_tensors
        kind, fa, fw, gscale = ctx.cfg
        fb = fw if kind == "aw" else fa
        dyr = rnd(dy * gscale, fa) / gscale
        da = dyr @ rnd(b, fb).transpose(-1, -2)
        db = rnd(a, fa).transpose(-1, -2) @ dyr
        while db.dim() > b.dim():
            db = db.sum(0)
        return da, db, None, None, None, None


def make_hook(fa, fw, gscale):
    def hook(a, b, kind):
        return RoundedMM.apply(a, b, kind, fa, fw, float(gscale))
    return hook


def rel(a, b):
    a, b = a.double().numpy(), b.double().numpy()
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-300))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="")
    ap.add_argument("--batch", type=int, default=32)
    args = ap.parse_args()
    cfg = mo.config_for("base", patch_size=16, in_chans=5, img_size=64, embed_dim=768)
    st = mo.init_state(cfg, seed=0)
    g = torch.Generator().manual_seed(1234)
    imgs = torch.randn(args.batch, 5, 64, 64, generator=g).clamp_(min=-3.0)
    noise = torch.rand(args.batch, 16, generator=g)
    mo.OPERAND_HOOK = None
    loss0, pred0, _, _, _, grads0 = mo.loss_and_grads(st, imgs, cfg, 0.75, noise)
    variants = [
        ("f32 through the hook (control)", "f32", "f32", 1),
        ("(i) bf16 activations x bf16 weights (the timed mode)", "bf16", "bf16", 1),
        ("bf16 activations x f32 weights", "bf16", "f32", 1),
        ("f32 activations x bf16 weights", "f32", "bf16", 1),
        ("(ii) f16 x f16, loss scale 2^16", "f16", "f16", 2 ** 16),
        ("f16 x f16, no loss scale", "f16", "f16", 1),
        ("f16 activations x f32 weights", "f16", "f32", 2 ** 16),
        ("f32 activations x f16 weights", "f32", "f16", 2 ** 16),
        ("(iii) bf16 activations x split bf16 weights (hi + lo)", "bf16", "bf16x2", 1),
        ("f16 activations x split f16 weights", "f16", "f16x2", 2 ** 16),
        ("(iv) split bf16 x split bf16 (three products)", "bf16x2", "bf16x2", 1),
        ("split f16 x f16 weights", "f16x2", "f16", 2 ** 16),
        ("bf16x3 x bf16x3", "bf16x3", "bf16x3", 1),
    ]
    rows = []
    for name, fa, fw, gs in variants:
        t0 = time.time()
        mo.OPERAND_HOOK = make_hook(fa, fw, gs)
        try:
            loss, pred, _, _, _, grads = mo.loss_and_grads(st, imgs, cfg, 0.75, noise)
        finally:
            mo.OPERAND_HOOK = None
        gr = {k: rel(grads[k], grads0[k]) for k in grads0 if float(grads0[k].abs().max()) > 0}
        row = dict(variant=name, activations=fa, weights=fw, loss_scale=gs,
                   loss_rel=abs(float(loss) - float(loss0)) / float(loss0), pred_rel_l2=rel(pred, pred0),
                   grad_rel_l2_max=max(gr.values()), grad_rel_l2_median=float(np.median(list(gr.values()))),
                   grads_finite=bool(all(torch.isfinite(v).all() for v in grads.values())))
        rows.append(row)
        print(f"{name:58s} loss {row['loss_rel']:.2e}  pred {row['pred_rel_l2']:.2e}  grad max {row['grad_rel_l2_max']:.2e} "
              f"median {row['grad_rel_l2_median']:.2e}  finite {row['grads_finite']}  ({time.time() - t0:.0f} s)", flush=True)
    out = dict(sample=f"config A (MAE ViT-B/16, 5x64x64, mask 0.75), B = {args.batch} slice of bench.py's synthetic batch, seed-0 init; "
                      "oracle/mae_oracle.py with OPERAND_HOOK against the same oracle unrounded; fp32 accumulation",
               tolerance="north_star: loss and reconstructed pixels within 1e-3 relative", rows=rows)
    if args.out:
        with open(args.out, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
