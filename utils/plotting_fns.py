"""Drop-in module path of the reference (``utils.plotting_fns``): re-exports the MI355X-native mirror (numbers, no figures)."""
from sky_embeddings_amd.utils.plotting_fns import *  # noqa: F401,F403
from sky_embeddings_amd.utils import plotting_fns as _impl

__all__ = [n for n in dir(_impl) if not n.startswith("_")]
