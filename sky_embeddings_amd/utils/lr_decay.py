"""utils/lr_decay.py mirror: parameter groups with layer-wise learning-rate decay (BEiT / MAE fine-tuning recipe), over the
NAMES and shapes of a model's trainable tensors -- the models here are not ``nn.Module``s, their optimisers take names."""
from __future__ import annotations


def get_layer_id_for_vit(name, num_layers):
    """utils/lr_decay.py:60-74: embeddings are layer 0, ``blocks.i`` layer i + 1, an input norm layer 1, everything else
    (final norm, head, patch_mask_values, the RA/Dec encoder) the last layer."""
    if name in ('cls_token', 'pos_embed'):
        return 0
    if name.startswith('patch_embed'):
        return 0
    if name.startswith('blocks'):
        return int(name.split('.')[1]) + 1
    if 'input_norm' in name:
        return 1
    return num_layers


def param_groups_lrd(model, init_lr, weight_decay=0.05, no_weight_decay_list=(), layer_decay=.75):
    """utils/lr_decay.py:14-57.  ``model`` offers ``num_blocks`` and ``trainable_tensors() -> [(name, ndim)]``.
    -> (groups [{'lr', 'weight_decay', 'params': [names]}], [lr per group]): one group per (layer, decayed?) in first-seen
    order, lr = init_lr * layer_decay ** (num_layers - layer); 1-D tensors and the listed names are not decayed."""
    groups, order = {}, []
    num_layers = model.num_blocks + 1
    scales = [layer_decay ** (num_layers - i) for i in range(num_layers + 1)]
    for name, ndim in model.trainable_tensors():
        no_decay = ndim == 1 or name in no_weight_decay_list or 'input_norm' in name
        layer = get_layer_id_for_vit(name, num_layers)
        key = "layer_%d_%s" % (layer, "no_decay" if no_decay else "decay")
        if key not in groups:
            order.append(key)
            groups[key] = {"lr": init_lr * scales[layer], "weight_decay": 0. if no_decay else weight_decay, "params": []}
        groups[key]["params"].append(name)
    out = [groups[k] for k in order]
    return out, [g["lr"] for g in out]
