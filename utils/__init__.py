"""Reference-compatible import path (``from utils.mim_vit import build_model`` ...)."""
