"""sky_embeddings_amd -- MI355X-native hot path of teaghan/sky_embeddings.

Masked-image-modelling (MAE) pretraining of ViTs on 5-band HSC cutouts and weighted-cosine
top-k search over the resulting embeddings, as hand-written gfx950 HIP kernels behind the
reference's own module API (``utils.mim_vit``, ``utils.pretrain_fns``, ``utils.similarity`` ...).
"""
__version__ = "0.1.0"
