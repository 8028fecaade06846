"""Drop-in module path of the reference (``utils.pos_embed``): re-exports the MI355X-native mirror."""
from sky_embeddings_amd.utils.pos_embed import *  # noqa: F401,F403
from sky_embeddings_amd.utils import pos_embed as _impl

__all__ = [n for n in dir(_impl) if not n.startswith("_")]
