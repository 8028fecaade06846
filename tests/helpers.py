"""Shared helpers for the parity tests (load goldens into oracle structures)."""
import os
from collections import OrderedDict

import numpy as np
import torch

from oracle import mae_oracle as mo

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_case(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    img, patch, C, D, depth, heads, Dd, ddepth, dheads, norm_pix = [int(v) for v in z["cfg"]]
    cfg = mo.MAEConfig(img_size=img, patch_size=patch, in_chans=C, embed_dim=D, depth=depth, num_heads=heads,
                       decoder_embed_dim=Dd, decoder_depth=ddepth, decoder_num_heads=dheads,
                       norm_pix_loss=bool(norm_pix), loss_fn=str(z["loss_fn"]), pixel_mean=float(z["pixel_mean"]),
                       pixel_std=float(z["pixel_std"]), ra_dec="ra_dec" in z.files)
    state = OrderedDict()
    for name_, _shape in mo.state_layout(cfg):
        state[name_] = torch.from_numpy(z["state/" + name_].copy())
    return z, cfg, state


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


def load_simmim_case(name):
    """SimMIM-mode goldens (tests/golden/make_golden.py simmim_case): -> (npz, cfg, state, imgs, pixel_mask, ra_dec|None)."""
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    img, patch, C, D, depth, heads, norm_pix, rd = [int(v) for v in z["cfg"][:8]]
    pool = len(z["cfg"]) > 8 and bool(z["cfg"][8])                  # attention pooling (case J)
    cfg = mo.config_for("simmim", img_size=img, patch_size=patch, in_chans=C, embed_dim=D, depth=depth, num_heads=heads,
                        norm_pix_loss=bool(norm_pix), loss_fn=str(z["loss_fn"]), pixel_mean=float(z["pixel_mean"]),
                        pixel_std=float(z["pixel_std"]), ra_dec=bool(rd), attn_pool=pool)
    state = OrderedDict()
    for name_, _shape in mo.state_layout(cfg):
        state[name_] = torch.from_numpy(z["state/" + name_].copy())
    ra_dec = torch.from_numpy(z["ra_dec"].copy()) if rd else None
    return z, cfg, state, torch.from_numpy(z["imgs"].copy()), torch.from_numpy(z["pixel_mask"].copy()), ra_dec


def record_parity(name, values):
    """Achieved errors of a parity test, merged into gpurun_out/parity_errors.json (scratch; `tests/parity_report.py` copies
    the file into profiles/ after a GPU run) and printed: the bars in the tests are set at 2x these figures."""
    import json
    import os
    root = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, "gpurun_out", "parity_errors.json")
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        data = json.load(open(path)) if os.path.exists(path) else {}
        data[name] = values
        json.dump(data, open(path, "w"), indent=1, sort_keys=True)
    except OSError:
        pass
    print(f"[parity] {name}: {values}")
