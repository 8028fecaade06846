"""utils/pretrain_fns.py mirror: ``run_iter`` (one forward / backward / AdamW / LR step) and the linear-probe validation
hook (``linear_probe`` / ``get_embeddings``: the encoder runs on the HIP path, the probes themselves are the reference's
scikit-learn estimators on the host)."""
import numpy as np
import torch


def run_iter(model, samples, ra_decs, masks, mask_ratio, optimizer, lr_scheduler, losses_cp, mode='train'):
    """utils/pretrain_fns.py:17-50 -- same signature, same return tuple."""
    model.train(mode == 'train')
    if 'train' in mode:
        loss, _, _ = model(samples, ra_dec=ra_decs, mask_ratio=mask_ratio, mask=masks)
        if loss.numel() > 1:
            loss = loss.unsqueeze(0).mean()
        loss.backward()
        optimizer.step()
        optimizer.zero_grad(set_to_none=True)
        lr_scheduler.step()
        # the reference appends float(loss) (a device sync per step, utils/pretrain_fns.py:44); keeping the
        # 0-d tensor defers the sync to the first time the value is read (same values)
        losses_cp['train_loss'].append(loss.detach())
    else:
        with torch.no_grad():
            loss, _, _ = model(samples, ra_dec=ra_decs, mask_ratio=mask_ratio, mask=masks)
        if loss.numel() > 1:
            loss = loss.unsqueeze(0).mean()
        losses_cp['val_loss'].append(loss.detach())
    return model, optimizer, lr_scheduler, losses_cp


def _probe_scores(x, y, estimator, score):
    """80/20 split with the reference's seed, fit, -> (train score, held-out score)."""
    from sklearn.model_selection import train_test_split
    x_fit, x_held, y_fit, y_held = train_test_split(x, y, test_size=0.2, random_state=42)
    estimator.fit(x_fit, y_fit)
    return float(score(y_fit, estimator.predict(x_fit))), float(score(y_held, estimator.predict(x_held)))


def linear_probe(model, losses_cp, device, dataloader_template, class_data_path=None, regress_data_path=None, combine='central',
                 remove_cls=True):
    """utils/pretrain_fns.py:52-105: quality of the embeddings by a quick linear model on top of them -- multinomial logistic
    regression on the ``class`` labels (accuracy -> ``train_lp_acc`` / ``val_lp_acc``) and an elastic net on ``zspec``
    (R2 -> ``train_lp_r2`` / ``val_lp_r2``), same estimators, hyper-parameters and split seed as the reference."""
    from sklearn.linear_model import ElasticNet, LogisticRegression
    from sklearn.metrics import accuracy_score, r2_score
    if combine == 'token':
        remove_cls = False
    model.train(False)
    if class_data_path:
        x, y = get_embeddings(class_data_path, model, device, dataloader_template, y_label='class', combine=combine,
                              remove_cls=remove_cls)
        # lbfgs fits the multinomial model for multi-class labels (what the reference's multi_class='multinomial' selects)
        fit, held = _probe_scores(x, y, LogisticRegression(solver='lbfgs', max_iter=10000, C=0.01, random_state=42), accuracy_score)
        losses_cp['train_lp_acc'].append(fit)
        losses_cp['val_lp_acc'].append(held)
    if regress_data_path:
        x, y = get_embeddings(regress_data_path, model, device, dataloader_template, y_label='zspec', combine=combine,
                              remove_cls=remove_cls)
        fit, held = _probe_scores(x, y, ElasticNet(alpha=0.0001, l1_ratio=0.9, max_iter=10000, random_state=42), r2_score)
        losses_cp['train_lp_r2'].append(fit)
        losses_cp['val_lp_r2'].append(held)


def get_embeddings(data_path, model, device, dataloader_template, y_label='class', combine='central', remove_cls=True):
    """utils/pretrain_fns.py:107-159 -> (x [n, features], y [n]): encode every cutout of ``data_path`` (batch 64, file order),
    reduce the tokens per ``combine`` (token | flatten | pool | centralpool | central | mean | anything else = global
    standardisation of the raw tokens), standard-scale the features."""
    from sklearn.preprocessing import StandardScaler
    from .dataloaders import build_h5_dataloader, open_h5
    from .eval_fns import mae_latent
    from .misc import select_centre
    net = getattr(model, 'module', model)
    template = dataloader_template.dataset
    loader = build_h5_dataloader(data_path, batch_size=64, num_workers=dataloader_template.num_workers, img_size=template.img_size,
                                 num_patches=template.num_patches, patch_size=net.patch_embed.patch_size[0],
                                 num_channels=net.in_chans, max_mask_ratio=None, shuffle=False)
    tokens = mae_latent(model, loader, device, verbose=0, remove_cls=remove_cls).numpy()
    with open_h5(data_path) as f:
        y = np.asarray(f[y_label][:])
    if net.attn_pool:
        combine = 'flatten'          # an attention-pooled encoder returns a single feature row
    n = tokens.shape[0]
    reducers = {'token': lambda t: t[:, :1].reshape(n, -1), 'flatten': lambda t: t.reshape(n, -1), 'pool': lambda t: t.max(axis=1),
                'centralpool': lambda t: select_centre(t, 16).max(axis=1), 'central': lambda t: select_centre(t, 4).reshape(n, -1),
                'mean': lambda t: t.mean(axis=1)}
    if combine in reducers:
        return StandardScaler().fit_transform(reducers[combine](tokens)), y
    return (tokens - np.nanmean(tokens)) / np.nanstd(tokens), y
