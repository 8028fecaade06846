"""utils/similarity.py mirror: same functions, argument meaning and return values.

The weighted-cosine scoring (the reference's default metric, utils/similarity.py:149-172) and the
"keep the best n_save" selection run in the HIP kernels of ``sky_embeddings_amd.search``; the
MSE / MAE metrics and the small bookkeeping stay torch glue.  Ties are
ordered by lower sample index (the reference's unstable argsort leaves them unspecified).
"""
from __future__ import annotations

import time

import torch

from .. import search


def get_train_samples(dataloader, nested_batches):
    """utils/similarity.py:4-14."""
    if nested_batches:
        for sample_batches, masks, ra_decs in dataloader:
            for samples, mask, ra_dec in zip(sample_batches[0], masks[0], ra_decs[0]):
                yield samples, mask, ra_dec
    else:
        for samples, mask, ra_dec in dataloader:
            yield samples, mask, ra_dec


def update_best_scores(samples, ra_decs, similarity_scores, best_samples, best_ra_decs, best_scores, n_save, metric):
    """utils/similarity.py:18-35 (stable ordering: earlier entries win ties)."""
    combined_scores = torch.cat((best_scores, similarity_scores), dim=0)
    combined_samples = torch.cat((best_samples, samples), dim=0)
    combined_ra_decs = torch.cat((best_ra_decs, ra_decs), dim=0)
    sorted_indices = torch.argsort(combined_scores, descending=(metric == 'cosine'), stable=True)[:n_save]
    return combined_samples[sorted_indices], combined_ra_decs[sorted_indices], combined_scores[sorted_indices]


def determine_target_features(target_latent):
    """utils/similarity.py:134-147: mean feature vector + normalised inverse-variance weights."""
    target_latent = target_latent.reshape(-1, target_latent.shape[-1])
    avg_feat = torch.mean(target_latent, dim=0)
    weight_feat = 1 / torch.std(target_latent, dim=0) ** 2
    weight_feat = weight_feat / torch.sum(weight_feat)
    return avg_feat, weight_feat


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
        raise NotImplementedError("n_central_patches: the reference calls an un-imported select_centre here "
                                  "(utils/similarity.py:240, NameError); not supported")
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


def mae_simsearch(model, target_latent, dataloader, device, n_batches=None, metric='cosine', combine='min',
                  use_weights=True, max_pool=False, cls_token=False, nested_batches=True, n_save=256, verbose=100):
    """utils/similarity.py:37-132: stream the test set through the encoder, score every batch
    against the target set, keep the best ``n_save``; returns (samples, latents, ra_decs, scores)."""
    if not nested_batches:
        if n_batches is None:
            n_batches = len(dataloader)
        print(f'Performing similarity search on {min(len(dataloader), n_batches)} batches...')
    else:
        print(f'Performing similarity search on {len(dataloader)} tiles...')
    model.eval()
    mod = model.module if hasattr(model, 'module') else model
    num_extra_tokens = mod.num_extra_tokens
    target_latent = target_latent.to(device, non_blocking=True)
    if cls_token:
        target_latent = target_latent[:, :1]
    else:
        target_latent = target_latent[:, num_extra_tokens:]
        if max_pool:
            target_latent, _ = torch.max(target_latent, dim=1, keepdim=True)
    best_ra_decs = torch.empty((n_save, 2), device=device)
    best_scores = torch.full((n_save,), float('-inf') if metric == 'cosine' else float('inf'), device=device)
    time_start = time.time()
    with torch.no_grad():
        for i, (samples, masks, ra_decs) in enumerate(get_train_samples(dataloader, nested_batches)):
            samples = samples.to(device, non_blocking=True)
            ra_decs = ra_decs.to(device, non_blocking=True)
            if i == 0:
                best_samples = torch.empty((n_save, *samples.shape[1:]), device=device)
            test_latent, _, _ = mod.forward_features(samples, ra_dec=ra_decs, reshape_out=False)
            if cls_token:
                test_latent = test_latent[:, :1]
            else:
                test_latent = test_latent[:, num_extra_tokens:]
                if max_pool:
                    test_latent, _ = torch.max(test_latent, dim=1, keepdim=True)
            if i == 0:  # first batch defines the feature scale (utils/similarity.py:98-101)
                mean_feats = test_latent.mean(dim=(0, 1))
                std_feats = test_latent.std(dim=(0, 1), unbiased=True)
                target_latent = (target_latent - mean_feats) / (std_feats + 1e-8)
            flat = test_latent.reshape(-1, test_latent.shape[-1]).contiguous()
            search.standardise_(flat, mean_feats, std_feats)
            test_latent = flat.view(test_latent.shape)
            test_similarity = compute_similarity(target_latent, test_latent, metric=metric, combine=combine,
                                                 use_weights=use_weights)
            best_samples, best_ra_decs, best_scores = update_best_scores(samples, ra_decs, test_similarity, best_samples,
                                                                         best_ra_decs, best_scores, n_save, metric)
            if not nested_batches:
                if (i + 1) % verbose == 0:
                    print(f'Processed {i+1}/{n_batches} image batches...', end='\r')
                if (i + 1) >= n_batches:
                    break
            elif (i + 1) % verbose == 0:
                print(f'Processed {i+1} image batches ({(time.time() - time_start)/(i+1):0.2f} seconds per batch)...',
                      end='\r')
        best_latent, _, _ = mod.forward_features(best_samples, ra_dec=best_ra_decs, reshape_out=False)
    return best_samples, best_latent, best_ra_decs, best_scores
