"""CPU restatement of the reference similarity-search arithmetic (utils/similarity.py).

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.

Two layers:
  * reference-shaped functions (one target set -> one mean vector + inverse
    variance weights, streamed batches, running best-n) restated in torch
    fp32 -- pinned by goldens captured from ``utils/similarity.py`` itself
    (the module imports cleanly);
  * the build-level batched ``cosine_topk`` contract (queries[Q,D] x bank[N,D]
    -> top-k (score, index)), whose *bit-exact* definition is the C file
    ``oracle/topk_oracle.c`` (fixed fp32 fma-chain order); ``cosine_topk_np``
    below is the ctypes front end to it.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))


# --------------------------------------------------------------------------
# reference-shaped layer
# --------------------------------------------------------------------------
def determine_target_features(target_latent):
    """utils/similarity.py:134-147: mean vector + normalised inverse-variance
    (unbiased std) weights over the flattened target set."""
    t = target_latent.reshape(-1, target_latent.shape[-1])
    avg = torch.mean(t, dim=0)
    w = 1 / torch.std(t, dim=0) ** 2
    w = w / torch.sum(w)
    return avg, w


def weighted_cosine_similarity(target_feats, test_feats, weights, eps=1e-6):
    """utils/similarity.py:149-172."""
    dot = torch.sum(weights * target_feats * test_feats, dim=-1)
    mt = torch.sqrt(torch.sum(weights * target_feats ** 2, dim=-1))
    mx = torch.sqrt(torch.sum(weights * test_feats ** 2, dim=-1))
    return dot / (mt * mx + eps)


def weighted_MSE(target_feats, test_feats, weights):
    """utils/similarity.py:174-192."""
    return torch.mean((target_feats - test_feats) ** 2 * weights / torch.sum(weights), dim=-1)


def weighted_MAE(target_feats, test_feats, weights):
    """utils/similarity.py:194-212."""
    return torch.mean(torch.abs(target_feats - test_feats) * weights / torch.sum(weights), dim=-1)


def select_centre(latent, n_patches):
    """utils/misc.py:68-117: the central sqrt(n) x sqrt(n) block of a raster-ordered square grid of patch tokens."""
    side, k = int(latent.shape[1] ** 0.5), int(n_patches ** 0.5)
    assert k * k == n_patches, "n must be a perfect square"
    r0 = side // 2 - k // 2
    idx = [(r0 + i) * side + (r0 + j) for i in range(k) for j in range(k)]
    return latent[:, idx]


def compute_similarity(target_latent, test_latent, metric="MAE", combine="mean", use_weights=True, n_top_sims=None,
                       n_central_patches=None):
    """utils/similarity.py:214-268.  ``n_central_patches`` (:238-240) raises NameError in the reference (select_centre is
    not imported there); pinned here by goldens made with that import supplied (similarity_central.npz)."""
    largest = metric == "cosine"
    if n_central_patches is not None:
        target_latent = select_centre(target_latent, n_central_patches)
    tgt, w = determine_target_features(target_latent)
    if not use_weights:
        w = torch.ones_like(w)
    if metric == "MAE":
        s = weighted_MAE(tgt, test_latent, w)
    elif metric == "MSE":
        s = weighted_MSE(tgt, test_latent, w)
    else:
        s = weighted_cosine_similarity(tgt, test_latent, w)
    if n_top_sims is not None:
        s = torch.topk(s, k=n_top_sims, dim=1, largest=largest).values
    if combine == "mean":
        return torch.mean(s, dim=1)
    if combine == "min":
        return torch.min(s, dim=1).values
    return torch.max(s, dim=1).values


def update_best_scores(scores, tags, best_scores, best_tags, n_save, metric):
    """utils/similarity.py:18-35 on (score, tag) pairs; the reference carries
    whole images + RA/Dec, here the tag is the global sample index.  Ties are
    ordered by lower tag (the build's contract; reference argsort is unstable)."""
    cs = torch.cat((best_scores, scores))
    ct = torch.cat((best_tags, tags))
    key = -cs if metric == "cosine" else cs
    order = np.lexsort((ct.numpy(), key.numpy()))
    order = torch.from_numpy(order)[:n_save]
    return cs[order], ct[order]


def standardise_first_batch(first_batch_latent):
    """utils/similarity.py:98-100: mean / unbiased std over (batch, patch)."""
    return first_batch_latent.mean(dim=(0, 1)), first_batch_latent.std(dim=(0, 1), unbiased=True)


# --------------------------------------------------------------------------
# build-level batched contract (C oracle)
# --------------------------------------------------------------------------
_LIB = None


def build_c_oracle(force=False):
    so = os.path.join(_HERE, "libskyemb_oracle.so")
    src = os.path.join(_HERE, "topk_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "libskyemb_oracle.so"])
    return so


def _lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libskyemb_oracle.so")
        if not os.path.exists(so):
            build_c_oracle()
        L = ctypes.CDLL(so)
        f32p, i64p = ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int64)
        L.skyemb_oracle_cosine_topk.argtypes = [f32p, f32p, f32p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
                                                ctypes.c_int64, ctypes.c_float, f32p, i64p, ctypes.c_int]
        L.skyemb_oracle_cosine_topk.restype = ctypes.c_int
        L.skyemb_oracle_cosine_scores.argtypes = [f32p, f32p, f32p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
                                                  ctypes.c_float, f32p, ctypes.c_int]
        L.skyemb_oracle_cosine_scores.restype = ctypes.c_int
        L.skyemb_oracle_standardise.argtypes = [f32p, f32p, f32p, ctypes.c_int64, ctypes.c_int64, f32p, ctypes.c_int]
        L.skyemb_oracle_standardise.restype = ctypes.c_int
        L.skyemb_oracle_num_threads.restype = ctypes.c_int
        _LIB = L
    return _LIB


def _p(a, t=ctypes.c_float):
    return a.ctypes.data_as(ctypes.POINTER(t))


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def num_threads():
    return int(_lib().skyemb_oracle_num_threads())


def cosine_scores_np(queries, bank, weights=None, eps=1e-6, threads=0):
    q, x = _f32(queries), _f32(bank)
    Q, D = q.shape
    N = x.shape[0]
    w = _f32(weights) if weights is not None else np.ones(D, np.float32)
    out = np.empty((Q, N), np.float32)
    rc = _lib().skyemb_oracle_cosine_scores(_p(q), _p(x), _p(w), Q, N, D, eps, _p(out), threads)
    assert rc == 0
    return out


def cosine_topk_np(queries, bank, k, weights=None, eps=1e-6, threads=0):
    """Exact top-k by (score desc, index asc) of the fixed-order fp32 weighted
    cosine; returns (scores[Q,k] f32, idx[Q,k] i64)."""
    q, x = _f32(queries), _f32(bank)
    Q, D = q.shape
    N = x.shape[0]
    w = _f32(weights) if weights is not None else np.ones(D, np.float32)
    s = np.empty((Q, k), np.float32)
    i = np.empty((Q, k), np.int64)
    rc = _lib().skyemb_oracle_cosine_topk(_p(q), _p(x), _p(w), Q, N, D, k, eps, _p(s), _p(i, ctypes.c_int64), threads)
    assert rc == 0
    return s, i


def standardise_np(x, mu, sigma, threads=0):
    """(x - mu) / (sigma + 1e-8), fp32, IEEE divide (utils/similarity.py:101-102)."""
    x = _f32(x)
    out = np.empty_like(x)
    N, D = x.shape
    rc = _lib().skyemb_oracle_standardise(_p(x), _p(_f32(mu)), _p(_f32(sigma)), N, D, _p(out), threads)
    assert rc == 0
    return out
