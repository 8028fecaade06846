"""utils/similarity.py mirror: same functions, argument meaning and return values.

The weighted-cosine scoring (the reference's default metric, utils/similarity.py:149-172) runs in the
HIP kernels of ``sky_embeddings_amd.search``; the MSE / MAE metrics stay torch glue.
The streaming driver keeps its running best-``n_save`` set in a device-resident pool with
threshold-filtered insertion (:class:`_BestPool`): once the pool is full only samples that beat its
worst entry move any image bytes, instead of the reference's cat + argsort + gather of every image of
the pool per batch.  Ties are ordered by arrival (the reference's unstable argsort leaves them
unspecified).
"""
from __future__ import annotations

import time

import torch

from .. import search


def get_train_samples(dataloader, nested_batches):
    """utils/similarity.py:4-14: flat loaders yield (samples, masks, ra_decs); tile loaders yield one
    tile whose three members hold a leading singleton dim around a list of batches."""
    if not nested_batches:
        yield from dataloader
        return
    for tile in dataloader:
        yield from zip(*(member[0] for member in tile))


def _ranking(scores, metric, n):
    """Positions of the best ``n`` scores, best first (cosine: larger is better; MSE / MAE: smaller);
    equal scores keep their order of appearance."""
    return torch.sort(scores, descending=(metric == 'cosine'), stable=True).indices[:n]


def update_best_scores(samples, ra_decs, similarity_scores, best_samples, best_ra_decs, best_scores, n_save, metric):
    """utils/similarity.py:18-35: merge a scored batch into the running best ``n_save`` (incumbents win ties)."""
    keep = _ranking(torch.cat((best_scores, similarity_scores)), metric, n_save)
    n_old = best_scores.shape[0]

    def merged(old, new):
        out = torch.empty((keep.shape[0], *new.shape[1:]), dtype=new.dtype, device=new.device)
        from_old = keep < n_old
        out[from_old] = old[keep[from_old]].to(new.dtype)
        out[~from_old] = new[keep[~from_old] - n_old]
        return out
    return merged(best_samples, samples), merged(best_ra_decs, ra_decs), merged(best_scores, similarity_scores)


def determine_target_features(target_latent):
    """utils/similarity.py:134-147: per-feature mean of the target set and inverse-variance weights
    (unbiased variance) normalised to sum to one."""
    feats = target_latent.reshape(-1, target_latent.shape[-1])
    inv_var = torch.std(feats, dim=0, unbiased=True).square().reciprocal()    # std**2, as the reference rounds it
    return feats.mean(dim=0), inv_var / inv_var.sum()


def weighted_cosine_similarity(target_feats, test_feats, weights, eps=1e-6):
    """utils/similarity.py:149-172 on the GPU kernel: target_feats [D], test_feats [B,P,D] (or
    [B,D]) -> [B,P] ([B]).  Fixed fp32 summation order (oracle/topk_oracle.c)."""
    shp = test_feats.shape[:-1]
    rows = test_feats.reshape(-1, test_feats.shape[-1]).to(torch.float32).contiguous()
    if not rows.is_cuda:
        raise RuntimeError("weighted_cosine_similarity runs in the HIP kernel: tensors must be on the GPU")
    s = search.cosine_scores(target_feats.reshape(1, -1).to(rows.device), rows, weights.to(rows.device), eps=eps)
    return s.reshape(shp)


def weighted_MSE(target_feats, test_feats, weights):
    """utils/similarity.py:174-192."""
    return torch.mean((target_feats - test_feats) ** 2 * weights / torch.sum(weights), dim=-1)


def weighted_MAE(target_feats, test_feats, weights):
    """utils/similarity.py:194-212."""
    return torch.mean(torch.abs(target_feats - test_feats) * weights / torch.sum(weights), dim=-1)


def compute_similarity(target_latent, test_latent, metric='MAE', combine='mean', use_weights=True,
                       n_central_patches=None, n_top_sims=None):
    """utils/similarity.py:214-268."""
    largest = metric == 'cosine'
    if n_central_patches is not None:
        # utils/similarity.py:238-240 calls utils.misc.select_centre without importing it (NameError in the reference);
        # the intended behaviour -- target features from the central patch tokens only -- is what runs here
        from .misc import select_centre
        target_latent = select_centre(target_latent, n_central_patches)
    target_latent, feat_weights = determine_target_features(target_latent)
    if not use_weights:
        feat_weights = torch.ones_like(feat_weights)
    if metric == 'MAE':
        test_similarity = weighted_MAE(target_latent, test_latent, feat_weights)
    elif metric == 'MSE':
        test_similarity = weighted_MSE(target_latent, test_latent, feat_weights)
    elif metric == 'cosine':
        test_similarity = weighted_cosine_similarity(target_latent, test_latent, feat_weights)
    else:
        raise ValueError(f"unknown metric {metric!r}")
    if n_top_sims is not None:
        test_similarity = torch.topk(test_similarity, k=n_top_sims, dim=1, largest=largest).values
    if combine == 'mean':
        return torch.mean(test_similarity, dim=1)
    if combine == 'min':
        return torch.min(test_similarity, dim=1).values
    return torch.max(test_similarity, dim=1).values


class _BestPool:
    """Device-resident running best-``n`` set of (sample, ra_dec, score).  ``key`` = score oriented so that
    larger is better.  Until the pool is full every batch is appended; afterwards only batch members that
    beat the pool's worst key are appended, and the pool is cut back to ``n`` by a stable sort."""

    def __init__(self, n, larger_is_better):
        self.n, self.sign = int(n), (1.0 if larger_is_better else -1.0)
        self.key = self.samples = self.ra_decs = None
        self.floor = None                       # worst key of a full pool

    def push(self, samples, ra_decs, scores):
        key = scores.to(torch.float32) * self.sign
        if self.floor is not None:
            sel = torch.nonzero(key > self.floor).squeeze(1)     # equal keys lose to the incumbents (arrival order)
            if sel.numel() == 0:
                return
            key, samples, ra_decs = key[sel], samples[sel], ra_decs[sel]
        if self.key is None:
            self.key, self.samples, self.ra_decs = key, samples.clone(), ra_decs.clone()
        else:
            self.key = torch.cat((self.key, key))
            self.samples = torch.cat((self.samples, samples))
            self.ra_decs = torch.cat((self.ra_decs, ra_decs.to(self.ra_decs.dtype)))
        if self.key.shape[0] >= self.n:
            self._cut()

    def _cut(self):
        order = torch.sort(self.key, descending=True, stable=True).indices[:self.n]
        self.key, self.samples, self.ra_decs = self.key[order], self.samples[order], self.ra_decs[order]
        if self.key.shape[0] == self.n:
            self.floor = self.key[-1]

    def result(self):
        """(samples, ra_decs, scores), exactly ``n`` rows, best first; unfilled slots carry the reference's
        initial score (-inf / +inf, utils/similarity.py:66) and zeros."""
        self._cut()
        m = self.key.shape[0]
        if m < self.n:
            pad = self.n - m
            self.key = torch.cat((self.key, self.key.new_full((pad,), float('-inf'))))
            self.samples = torch.cat((self.samples, self.samples.new_zeros((pad, *self.samples.shape[1:]))))
            self.ra_decs = torch.cat((self.ra_decs, self.ra_decs.new_zeros((pad, 2))))
        return self.samples, self.ra_decs, self.key * self.sign


def _pick_tokens(latent, num_extra_tokens, cls_token, max_pool):
    """utils/similarity.py:54-63 / 87-95: the cls token alone, or the patch tokens (optionally max-pooled to one)."""
    if cls_token:
        return latent[:, :1]
    latent = latent[:, num_extra_tokens:]
    return latent.amax(dim=1, keepdim=True) if max_pool else latent


def mae_simsearch(model, target_latent, dataloader, device, n_batches=None, metric='cosine', combine='min',
                  use_weights=True, max_pool=False, cls_token=False, nested_batches=True, n_save=256, verbose=100):
    """utils/similarity.py:37-132: stream the test set through the encoder, score every batch against the
    target set, keep the best ``n_save``; returns (samples, latents, ra_decs, scores).  ``n_batches`` limits
    flat loaders only, as in the reference."""
    net = getattr(model, 'module', model)
    model.eval()
    limit = None if nested_batches else (len(dataloader) if n_batches is None else n_batches)
    if verbose:
        what = f'{len(dataloader)} tiles' if nested_batches else f'{min(len(dataloader), limit)} batches'
        print(f'Similarity search over {what} (metric {metric}, combine {combine}, keeping {n_save})')
    extra = net.num_extra_tokens
    target = _pick_tokens(target_latent.to(device, non_blocking=True), extra, cls_token, max_pool)
    pool = _BestPool(n_save, metric == 'cosine')
    scale = None
    started = time.time()
    with torch.no_grad():
        for seen, (samples, _masks, ra_decs) in enumerate(get_train_samples(dataloader, nested_batches), 1):
            samples = samples.to(device, non_blocking=True)
            ra_decs = ra_decs.to(device, non_blocking=True)
            feats = _pick_tokens(net.forward_features(samples, ra_dec=ra_decs, reshape_out=False)[0], extra, cls_token,
                                 max_pool)
            if scale is None:
                # the first batch fixes the feature scale of the whole search (utils/similarity.py:98-101)
                scale = (feats.mean(dim=(0, 1)), feats.std(dim=(0, 1), unbiased=True))
                target = (target - scale[0]) / (scale[1] + 1e-8)
            rows = feats.reshape(-1, feats.shape[-1]).to(torch.float32).contiguous()
            feats = search.standardise_(rows, *scale).view(feats.shape)
            pool.push(samples, ra_decs, compute_similarity(target, feats, metric=metric, combine=combine,
                                                           use_weights=use_weights))
            if verbose and seen % verbose == 0:
                print(f'  {seen} batches scored, {(time.time() - started) / seen:0.3f} s per batch', end='\r')
            if limit is not None and seen >= limit:
                break
        if scale is None:
            raise ValueError("mae_simsearch: the dataloader yielded no batch")
        best_samples, best_ra_decs, best_scores = pool.result()
        best_latent = net.forward_features(best_samples, ra_dec=best_ra_decs, reshape_out=False)[0]
    return best_samples, best_latent, best_ra_decs, best_scores
