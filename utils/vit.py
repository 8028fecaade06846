"""Drop-in module path of the reference (``utils.vit``): re-exports the MI355X-native mirror."""
from sky_embeddings_amd.utils.vit import *  # noqa: F401,F403
from sky_embeddings_amd.utils import vit as _impl

__all__ = [n for n in dir(_impl) if not n.startswith("_")]
