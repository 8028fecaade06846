"""Drop-in module path of the reference (``utils.predictor_training_fns``): re-exports the MI355X-native mirror."""
from sky_embeddings_amd.utils.predictor_training_fns import *  # noqa: F401,F403
from sky_embeddings_amd.utils import predictor_training_fns as _impl

__all__ = [n for n in dir(_impl) if not n.startswith("_")]
