"""utils/pos_embed.py mirror: the fixed 2-D sin-cos table (init-time constant) and the checkpoint surgery that fits a
pre-trained positional table to another image size."""
import numpy as np
import torch

from ..model_config import sincos_pos_embed


def get_2d_sincos_pos_embed(embed_dim, grid_size, cls_token=False, ra_dec=False):
    """utils/pos_embed.py:20-39."""
    return sincos_pos_embed(embed_dim, grid_size, cls_token=cls_token, ra_dec=ra_dec)


def _sizes(model, checkpoint_model):
    table = checkpoint_model['pos_embed']
    num_patches = model.patch_embed.num_patches
    num_extra = model.pos_embed.shape[-2] - num_patches
    return table, num_extra, int((table.shape[-2] - num_extra) ** 0.5), int(num_patches ** 0.5)


def crop_pos_embed(model, checkpoint_model):
    """utils/pos_embed.py:89-115: keep the CENTRAL new_size x new_size patch positions of the checkpoint's table (in place)."""
    table, num_extra, orig, new = _sizes(model, checkpoint_model)
    if orig != new:
        print("Cropping the central %dx%d positional embeddings from the original %dx%d." % (new, new, orig, orig))
        border = int((orig - new) / 2)
        idx = np.arange(orig * orig).reshape(orig, orig)[border:-border, border:-border].flatten()
        checkpoint_model['pos_embed'] = torch.cat((table[:, :num_extra], table[:, num_extra:][:, idx]), dim=1)


def interpolate_pos_embed(model, checkpoint_model):
    """utils/pos_embed.py:122-144 (DeiT): bicubic resampling of the patch positions to the model's grid, extra tokens kept."""
    if 'pos_embed' not in checkpoint_model:
        return
    table, num_extra, orig, new = _sizes(model, checkpoint_model)
    if orig != new:
        print("Position interpolate from %dx%d to %dx%d" % (orig, orig, new, new))
        dim = table.shape[-1]
        pos = torch.as_tensor(table[:, num_extra:]).float().reshape(-1, orig, orig, dim).permute(0, 3, 1, 2)
        pos = torch.nn.functional.interpolate(pos, size=(new, new), mode='bicubic', align_corners=False)
        checkpoint_model['pos_embed'] = torch.cat((torch.as_tensor(table[:, :num_extra]).float(), pos.permute(0, 2, 3, 1).flatten(1, 2)), dim=1)
