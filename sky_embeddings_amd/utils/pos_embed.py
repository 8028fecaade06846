"""utils/pos_embed.py mirror: the fixed 2-D sin-cos table (init-time constant)."""
from ..model_config import sincos_pos_embed


def get_2d_sincos_pos_embed(embed_dim, grid_size, cls_token=False, ra_dec=False):
    """utils/pos_embed.py:20-39."""
    return sincos_pos_embed(embed_dim, grid_size, cls_token=cls_token, ra_dec=ra_dec)
