"""utils/eval_fns.py mirror: ``mae_latent`` (encode a dataset into the embeddings that are
searched), ``mae_predict`` (reconstructions), and the bank builder of SURVEY.md §8f rank 1.
``ft_predict`` (downstream predictor) is out of scope."""
from __future__ import annotations

import numpy as np
import torch


def _mod(model):
    return model.module if hasattr(model, 'module') else model


def mae_predict(model, dataloader, device, mask_ratio, single_batch=True):
    """utils/eval_fns.py:9-70: (pred_imgs, masked_inputs, orig_imgs) as NHWC numpy arrays."""
    if not single_batch:
        print('Predicting on %i batches...' % (len(dataloader)))
    model.eval()
    mod = _mod(model)
    pred_imgs, mask_imgs, orig_imgs = [], [], []
    with torch.no_grad():
        for samples, mask, ra_decs in dataloader:
            samples = samples.to(device, non_blocking=True)
            loss, pred, mask = model(samples, ra_dec=ra_decs, mask_ratio=mask_ratio, mask=mask)
            if not mod.simmim:   # MAE: patch rows -> image; SimMIM predictions and masks are images already
                pred = mod.unpatchify(pred)
                mask = mask.detach().unsqueeze(-1).repeat(1, 1, mod.patch_embed.patch_size[0] ** 2 * mod.in_chans)
                mask = mod.unpatchify(mask)
            else:
                pred = pred.clone()
            pred = mod.denorm_imgs(samples, pred)
            pred = torch.einsum('nchw->nhwc', pred).detach().clone()
            mask = torch.einsum('nchw->nhwc', mask).detach()
            samples = torch.einsum('nchw->nhwc', samples)
            pred[mask == 0] = samples[mask == 0]
            masked_samples = samples.detach().clone()
            masked_samples[mask == 1] = torch.nan
            pred_imgs.append(pred.cpu().numpy())
            mask_imgs.append(masked_samples.cpu().numpy())
            orig_imgs.append(samples.cpu().numpy())
            if single_batch:
                break
    return np.concatenate(pred_imgs), np.concatenate(mask_imgs), np.concatenate(orig_imgs)


def mae_latent(model, dataloader, device, n_batches=None, return_images=False, verbose=1, apply_augmentations=False,
               num_augmentations=16, remove_cls=True, augmentations=None):
    """utils/eval_fns.py:72-140.  ``augmentations`` (callable on a [C,H,W] tensor) replaces the
    reference's torchvision pipeline, which is not available here."""
    if n_batches is None:
        n_batches = len(dataloader)
    if verbose > 0:
        print(f'Encoding {min(len(dataloader), n_batches)} batches...')
    model.eval()
    mod = _mod(model)
    if apply_augmentations and augmentations is None:
        raise NotImplementedError("apply_augmentations=True needs augmentations=<callable> (torchvision is absent)")
    latents, images = [], []
    with torch.no_grad():
        for samples, masks, ra_decs in dataloader:
            if apply_augmentations:
                aug_s, aug_r = [], []
                for idx, sample in enumerate(samples):
                    aug_s.append(sample.unsqueeze(0))
                    aug_r.append(ra_decs[idx].unsqueeze(0))
                    for _ in range(num_augmentations):
                        aug_s.append(augmentations(sample).unsqueeze(0))
                        aug_r.append(ra_decs[idx].unsqueeze(0))
                samples, ra_decs = torch.cat(aug_s, dim=0), torch.cat(aug_r, dim=0)
            samples = samples.to(device, non_blocking=True)
            latent, _, _ = mod.forward_features(samples, ra_dec=ra_decs, mask=None, reshape_out=False)
            if mod.attn_pool:
                remove_cls = False
            if remove_cls:
                latent = latent[:, mod.num_extra_tokens:]
            latents.append(latent.detach().cpu())
            if return_images:
                images.append(samples.detach().cpu())
            if len(latents) >= n_batches:
                break
    if return_images:
        return torch.cat(latents), torch.cat(images)
    return torch.cat(latents)


def build_embedding_bank(model, dataloader, device, pool='max', n_batches=None):
    """Encode a dataset ONCE into a resident [N, D] fp32 bank (cls token, or max / mean pool over
    the patch tokens -- all permutation invariant, so the reference's shuffled token order does not
    matter).  The reference re-encodes every test image per search (utils/similarity.py:81)."""
    model.eval()
    mod = _mod(model)
    rows = []
    with torch.no_grad():
        for i, (samples, masks, ra_decs) in enumerate(dataloader):
            latent, _, _ = mod.forward_features(samples.to(device, non_blocking=True), reshape_out=False)
            if pool == 'cls':
                rows.append(latent[:, 0].clone())
            elif pool == 'max':
                rows.append(latent[:, mod.num_extra_tokens:].max(dim=1).values)
            elif pool == 'mean':
                rows.append(latent[:, mod.num_extra_tokens:].mean(dim=1))
            else:
                raise ValueError(pool)
            if n_batches is not None and i + 1 >= n_batches:
                break
    return torch.cat(rows).contiguous()


def ft_predict(*a, **k):
    raise NotImplementedError("ft_predict belongs to the downstream-predictor workload (out of scope, SURVEY.md §2)")
