"""utils/predictor_training_fns.py mirror: one training / validation iteration of a downstream predictor."""
import torch


def run_iter(model, samples, masks, ra_decs, labels, optimizer, lr_scheduler, losses_cp, loss_fn='mse', label_uncertainties=None,
             mode='train'):
    """utils/predictor_training_fns.py:3-61.  The forward pass runs the HIP encoder; ``loss.backward()`` reaches it through
    the autograd node utils.vit.VisionTransformer.forward installs (its backward runs the engine's backward schedule)."""
    model.train(mode == 'train')
    with torch.set_grad_enabled(mode == 'train'):
        model_output = model(samples, mask=masks, ra_dec=ra_decs)
        labels = labels.to(model_output.device)
        if 'crossentropy' in loss_fn.lower():
            loss = torch.nn.CrossEntropyLoss()(model_output, labels.squeeze(1))
            metric = (torch.max(model_output, 1)[1] == labels.squeeze(1)).float().mean()
        if 'mse' in loss_fn.lower():
            labels = model.module.normalize_labels(labels)
            if label_uncertainties is None:
                loss = torch.nn.MSELoss()(model_output, labels)
            else:
                weights = 1.0 / (label_uncertainties.to(model_output.device) + 1e-5)      # inverse uncertainties as weights
                loss = (torch.nn.functional.mse_loss(model_output, labels, reduction='none') * weights).mean()
            metric = torch.nn.L1Loss()(model_output, labels)
    key = 'acc' if 'crossentropy' in loss_fn.lower() else 'mae'
    if 'train' in mode:
        loss.backward()
        optimizer.step()
        optimizer.zero_grad(set_to_none=True)
        lr_scheduler.step()
        losses_cp['train_loss'].append(float(loss.detach()))
        losses_cp['train_' + key].append(float(metric))
    else:
        losses_cp['val_loss'].append(float(loss.detach()))
        losses_cp['val_' + key].append(float(metric))
    return model, optimizer, lr_scheduler, losses_cp
