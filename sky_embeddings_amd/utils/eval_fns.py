"""utils/eval_fns.py mirror: ``mae_latent`` (encode a dataset into the embeddings that are searched, optionally with the
target augmentations), ``mae_predict`` (reconstructions), and the bank builder of SURVEY.md §8f rank 1.
``ft_predict``: the predictions of a trained downstream predictor."""
from __future__ import annotations

import numpy as np
import torch


def _net(model):
    return getattr(model, 'module', model)


def _nhwc(t):
    return t.permute(0, 2, 3, 1)


def mae_predict(model, dataloader, device, mask_ratio, single_batch=True):
    """utils/eval_fns.py:9-70 -> (pred_imgs, masked_inputs, orig_imgs), NHWC numpy arrays in input units: predictions at
    the masked pixels, the input elsewhere; ``masked_inputs`` = the input with the masked pixels set to NaN."""
    net = _net(model)
    model.eval()
    if not single_batch:
        print(f'Reconstructing {len(dataloader)} batches')
    out = ([], [], [])
    with torch.no_grad():
        for samples, mask, ra_decs in dataloader:
            samples = samples.to(device, non_blocking=True)
            _, pred, mask = model(samples, ra_dec=ra_decs, mask_ratio=mask_ratio, mask=mask)
            if net.simmim:
                pred = pred.clone()                     # image-shaped already (SimMIM head + PixelShuffle index map)
            else:
                # per-patch rows -> image; the [B, L] patch mask -> pixel mask through the same un-patchify
                per_patch = net.patch_embed.patch_size[0] ** 2 * net.in_chans
                pred = net.unpatchify(pred)
                mask = net.unpatchify(mask.detach()[:, :, None].expand(-1, -1, per_patch))
            hidden = _nhwc(mask) == 1
            orig = _nhwc(samples)
            recon = torch.where(hidden, _nhwc(net.denorm_imgs(samples, pred)), orig)
            shown = torch.where(hidden, torch.full_like(orig, float('nan')), orig)
            for dst, t in zip(out, (recon, shown, orig)):
                dst.append(t.cpu().numpy())
            if single_batch:
                break
    return tuple(np.concatenate(parts) for parts in out)


def mae_latent(model, dataloader, device, n_batches=None, return_images=False, verbose=1, apply_augmentations=False,
               num_augmentations=16, remove_cls=True, augmentations=None):
    """utils/eval_fns.py:72-140: encoder-only forward of every batch -> latents [N, tokens, D] on the host.  With
    ``apply_augmentations`` every sample is followed by ``num_augmentations`` augmented copies (flip, resized crop,
    brightness, noise, NaN channels: utils/dataloaders.py:90-106) and its RA/Dec repeated for each -- produced by ONE
    device launch per batch (``Augmenter.batch``) instead of the reference's per-copy Python loop.  ``augmentations``
    overrides the pipeline (an ``Augmenter``, or any callable on one [C, H, W] tensor)."""
    net = _net(model)
    model.eval()
    limit = len(dataloader) if n_batches is None else min(n_batches, len(dataloader))
    if verbose > 0:
        print(f'Encoding {limit} batches' + (f' (+ {num_augmentations} augmented copies per sample)' if apply_augmentations else ''))
    if apply_augmentations and augmentations is None:
        from .dataloaders import get_augmentations
        augmentations = get_augmentations()         # crops are resized back to the sample's own size
    keep_from = 0 if (not remove_cls or net.attn_pool) else net.num_extra_tokens
    latents, images = [], []
    with torch.no_grad():
        for done, (samples, _masks, ra_decs) in enumerate(dataloader, 1):
            if apply_augmentations:
                copies = 1 + num_augmentations
                if hasattr(augmentations, 'batch'):
                    samples = augmentations.batch(samples, num_augmentations)
                else:   # a plain per-sample callable: original first, then its copies
                    samples = torch.stack([s if a == 0 else augmentations(s.clone()) for s in samples for a in range(copies)])
                ra_decs = ra_decs.repeat_interleave(copies, dim=0)
            samples = samples.to(device, non_blocking=True)
            latent = net.forward_features(samples, ra_dec=ra_decs, mask=None, reshape_out=False)[0]
            latents.append(latent[:, keep_from:].detach().cpu())
            if return_images:
                images.append(samples.detach().cpu())
            if done >= limit:
                break
    latents = torch.cat(latents)
    return (latents, torch.cat(images)) if return_images else latents


def build_embedding_bank(model, dataloader, device, pool='max', n_batches=None):
    """Encode a dataset ONCE into a resident [N, D] fp32 bank (cls token, or max / mean pool over
    the patch tokens -- all permutation invariant, so the reference's shuffled token order does not
    matter).  The reference re-encodes every test image per search (utils/similarity.py:81)."""
    model.eval()
    net = _net(model)
    reduce = {'cls': lambda t: t[:, 0].clone(), 'max': lambda t: t[:, net.num_extra_tokens:].amax(dim=1),
              'mean': lambda t: t[:, net.num_extra_tokens:].mean(dim=1)}
    if pool not in reduce:
        raise ValueError(pool)
    rows = []
    with torch.no_grad():
        for done, (samples, _masks, ra_decs) in enumerate(dataloader, 1):
            rows.append(reduce[pool](net.forward_features(samples.to(device, non_blocking=True), ra_dec=ra_decs, reshape_out=False)[0]))
            if n_batches is not None and done >= n_batches:
                break
    return torch.cat(rows).contiguous()


def ft_predict(model, dataloader, device, num_batches=None, return_images=False, use_label_errs=False):
    """utils/eval_fns.py:142-192: predictions of a trained predictor (utils.vit) over a labelled loader, in label units
    (``denormalize_labels``) -> (tgt_labels, pred_labels[, images]) as numpy arrays.  With label errors the second half of the
    label columns is dropped.  (The reference stops after num_batches + 1 batches: kept.)"""
    model.eval()
    net = _net(model)
    tgt_labels, pred_labels, images = [], [], []
    if num_batches is None:
        num_batches = len(dataloader)
    print(f'Running predictions on {num_batches} batches...')
    with torch.no_grad():
        for i, (samples, masks, ra_decs, labels) in enumerate(dataloader):
            samples = samples.to(device, non_blocking=True)
            labels = labels.to(device, non_blocking=True)
            if use_label_errs:
                labels = labels[:, :labels.size(1) // 2]
            out = net.denormalize_labels(model(samples, mask=None, ra_dec=ra_decs.to(device, non_blocking=True)))
            tgt_labels.append(labels.cpu().numpy())
            pred_labels.append(out.float().cpu().numpy())
            if return_images:
                images.append(samples.cpu().numpy())
            if i == num_batches:
                break
    tgt_labels, pred_labels = np.concatenate(tgt_labels), np.concatenate(pred_labels)
    if return_images:
        return tgt_labels, pred_labels, np.concatenate(images)
    return tgt_labels, pred_labels
