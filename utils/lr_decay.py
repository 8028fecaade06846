"""Drop-in module path of the reference (``utils.lr_decay``): re-exports the MI355X-native mirror."""
from sky_embeddings_amd.utils.lr_decay import *  # noqa: F401,F403
from sky_embeddings_amd.utils import lr_decay as _impl

__all__ = [n for n in dir(_impl) if not n.startswith("_")]
