"""utils/pretrain_fns.py mirror: ``run_iter`` (one forward / backward / AdamW / LR step).

The linear-probe evaluation of the reference (sklearn, CPU) is out of scope."""
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


def linear_probe(*args, **kwargs):
    raise NotImplementedError("linear_probe (sklearn CPU evaluation, utils/pretrain_fns.py:52-159) is outside the "
                              "hot path this package accelerates (SURVEY.md §2 row 4)")
