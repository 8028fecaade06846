"""Drop-in mirror of the reference ``utils/mim_vit.py`` public surface for the MAE path.

Same entry points and argument meaning (``build_model``, ``load_model``, ``MaskedAutoencoderViT``
attributes reached through ``model.module``), same checkpoint format; the arithmetic runs in
hand-written gfx950 kernels (``sky_embeddings_amd.engine``) instead of torch/timm autograd.

Build extensions (all optional, defaults reproduce the reference):
  * ``model_type = tiny`` (BASELINE.json configs[0]); tolerant defaults for ini keys older
    configs lack (``attn_pool``, ``ra_dec``);
  * ``noise=`` keyword on ``forward`` / ``forward_features`` to supply the masking noise
    (reference draws ``torch.rand`` internally, utils/mim_vit.py:363);
  * ``[TRAINING] compute_dtype = f16|bf16|f32`` (default f16 = IEEE-half MFMA operands with a static loss scale: the throughput mode
    that holds loss / reconstructed pixels within 1e-3 of the fp32 reference; bf16 = the same kernels with bf16 operands (6e-3);
    f32 = exact-fp32 MFMA parity mode).
Deviation: NaN target pixels contribute a ZERO gradient (the reference's MSE backward is NaN
there).  SimMIM configurations (``model_type`` simmim / mimlarge / mimhuge, with or without the RA/Dec
token) run on ``sky_embeddings_amd.simmim_engine.SimMIMEngine``.
"""
from __future__ import annotations

import os
from collections import defaultdict

import torch

from ..engine import MAEEngine
from ..model_config import MODEL_TYPES, MAEConfig, config_for
from ..optim import CosineLR, FusedAdamW
from .misc import str2bool


class _PatchEmbedInfo:
    """Attribute surface of timm ``PatchEmbed`` that callers read (pretrain_mim.py:82,
    utils/pretrain_fns.py:118-119, utils/eval_fns.py:38)."""

    def __init__(self, cfg: MAEConfig):
        self.img_size = (cfg.img_size, cfg.img_size)
        self.patch_size = (cfg.patch_size, cfg.patch_size)
        self.grid_size = (cfg.grid, cfg.grid)
        self.num_patches = cfg.num_patches


class _LossBackward(torch.autograd.Function):
    """Lets callers keep writing ``loss.backward()`` (utils/pretrain_fns.py:34): the backward
    of this node runs the engine's explicit backward schedule into the flat gradient buffer."""

    @staticmethod
    def forward(ctx, hook, engine, loss):
        ctx.engine = engine
        return loss.detach().clone().reshape(())

    @staticmethod
    def backward(ctx, grad_out):
        ctx.engine.backward()
        return torch.zeros(1, device=grad_out.device), None, None


class MaskedAutoencoderViT:
    """Masked Autoencoder with VisionTransformer backbone (utils/mim_vit.py:183-559): MAE mode (MAEEngine) and SimMIM
    mode (SimMIMEngine), each with the optional RA/Dec token."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=1024, depth=24, num_heads=16,
                 decoder_embed_dim=512, decoder_depth=8, decoder_num_heads=16, mlp_ratio=4., norm_layer=None,
                 norm_pix_loss=False, simmim=False, loss_fn='mse', pixel_mean=0, pixel_std=1., attn_pool=False,
                 ra_dec=False, device="cuda", compute_dtype=torch.float16, seed=None):
        self.cfg = MAEConfig(img_size=img_size, patch_size=patch_size, in_chans=in_chans, embed_dim=embed_dim, depth=depth,
                             num_heads=num_heads, decoder_embed_dim=decoder_embed_dim, decoder_depth=decoder_depth,
                             decoder_num_heads=decoder_num_heads, mlp_ratio=mlp_ratio, norm_pix_loss=norm_pix_loss,
                             loss_fn=loss_fn, pixel_mean=float(pixel_mean), pixel_std=float(pixel_std), simmim=simmim,
                             attn_pool=attn_pool, ra_dec=ra_dec)
        self.simmim, self.loss_fn = simmim, loss_fn
        self.pixel_mean, self.pixel_std = pixel_mean, pixel_std
        self.norm_pix_loss, self.in_chans, self.ra_dec = norm_pix_loss, in_chans, ra_dec
        self.attn_pool = bool(attn_pool) and bool(simmim)       # utils/mim_vit.py:246-254: SimMIM only (:282 for MAE)
        self.num_extra_tokens = 2 if ra_dec else 1
        self.tile_size = img_size // patch_size
        self.patch_embed = _PatchEmbedInfo(self.cfg)
        if simmim:
            from ..simmim_engine import SimMIMEngine
            self.engine = SimMIMEngine(self.cfg, device=device, compute_dtype=compute_dtype, seed=seed)
        else:
            self.engine = MAEEngine(self.cfg, device=device, compute_dtype=compute_dtype, seed=seed)
        self.training = True
        self._hook = torch.zeros(1, device=self.engine.device, requires_grad=True)

    # ---- nn.Module-like surface ---------------------------------------------------------------
    def to(self, device):
        assert torch.device(device).type == self.engine.device.type, "the engine lives on the GPU it was built on"
        return self

    def train(self, mode=True):
        self.training = bool(mode)
        return self

    def eval(self):
        return self.train(False)

    def state_dict(self):
        return self.engine.state_dict()

    def load_state_dict(self, sd, strict=True):
        self.engine.load_state_dict(sd, strict=strict)

    def parameters(self):
        return [self.engine.store.param(n) for n in self.engine.store.order]

    def named_parameters(self):
        return [(n, self.engine.store.param(n)) for n, _ in self.engine.state_dict().items()]

    # ---- utils/mim_vit.py:326-352 (torch views; not on the hot path) ----------------------------
    def patchify(self, imgs):
        p = self.patch_embed.patch_size[0]
        assert imgs.shape[2] == imgs.shape[3] and imgs.shape[2] % p == 0
        h = w = imgs.shape[2] // p
        x = imgs.reshape(shape=(imgs.shape[0], self.in_chans, h, p, w, p))
        x = torch.einsum('nchpwq->nhwpqc', x)
        return x.reshape(shape=(imgs.shape[0], h * w, p ** 2 * self.in_chans))

    def unpatchify(self, x):
        p = self.patch_embed.patch_size[0]
        h = w = int(x.shape[1] ** .5)
        assert h * w == x.shape[1]
        x = x.reshape(shape=(x.shape[0], h, w, p, p, self.in_chans))
        x = torch.einsum('nhwpqc->nchpwq', x)
        return x.reshape(shape=(x.shape[0], self.in_chans, h * p, h * p))

    def norm_inputs(self, x):
        return (x - self.pixel_mean) / self.pixel_std

    def denorm_imgs(self, orig_imgs, x):
        if self.norm_pix_loss:
            x = undo_pixel_norm(orig_imgs, x, self)
        return x * self.pixel_std + self.pixel_mean

    # ---- hot path -----------------------------------------------------------------------------
    def _prep(self, x):
        return x.to(self.engine.device, torch.float32).contiguous()

    def forward_features(self, x, ra_dec=None, mask_ratio=0, mask=None, reshape_out=True, noise=None):
        """utils/mim_vit.py:381-438.  MAE mode -> (latent [B, 1+keep, D], mask [B,L], ids_restore [B,L]); as in the
        reference, mask_ratio=0 keeps every patch but in SHUFFLED order.  SimMIM mode -> tokens in
        order, (latent, pixel mask, None); reshape_out=True drops the extra tokens and returns [B, D, h, w]."""
        if self.simmim:
            m = None if mask is None else mask.to(self.engine.device)
            latent, m, _ = self.engine.forward_features(self._prep(x), mask=m, ra_dec=ra_dec)
            latent = latent.clone()
            if reshape_out:
                if not self.attn_pool:                       # a pooled latent is one token: [B, D, 1, 1] (utils/mim_vit.py:431-436)
                    latent = latent[:, self.num_extra_tokens:]
                B, L, C = latent.shape
                H = W = int(L ** 0.5)
                latent = latent.permute(0, 2, 1).reshape(B, C, H, W)
            return latent, m, None
        latent, m, ids = self.engine.forward_features(self._prep(x), mask_ratio=mask_ratio, noise=noise, ra_dec=ra_dec)
        return latent.clone(), m.clone(), ids.clone()

    def forward(self, imgs, ra_dec=None, mask_ratio=0.75, mask=None, denorm_out=False, noise=None):
        """utils/mim_vit.py:552-559 -> (loss, pred, mask): MAE pred [B,L,p*p*C], mask [B,L]; SimMIM pred [B,C,H,W] and
        the per-pixel mask it was given."""
        if self.simmim:
            loss, pred, m = self.engine.forward_train(self._prep(imgs), mask=mask.to(self.engine.device), ra_dec=ra_dec)
        else:
            loss, pred, m = self.engine.forward_train(self._prep(imgs), mask_ratio=mask_ratio, noise=noise, ra_dec=ra_dec)
        if torch.is_grad_enabled():
            loss = _LossBackward.apply(self._hook, self.engine, loss)
        else:
            loss = loss.detach().clone().reshape(())
        return loss, pred, m

    __call__ = forward


class _DataParallelShim:
    """Stands where ``nn.DataParallel(model)`` stands in the reference (utils/mim_vit.py:117):
    callers reach the model through ``.module``.  Multi-GPU is one process per GPU with an RCCL
    gradient all-reduce (sky_embeddings_amd.distributed), not intra-process replication."""

    def __init__(self, module):
        self.module = module

    def __call__(self, *a, **k):
        return self.module.forward(*a, **k)

    def train(self, mode=True):
        self.module.train(mode)
        return self

    def eval(self):
        return self.train(False)

    def parameters(self):
        return self.module.parameters()

    def named_parameters(self):
        return self.module.named_parameters()

    def state_dict(self):
        return self.module.state_dict()


def _compute_dtype(config, default='f16'):
    """[TRAINING] compute_dtype (or SKYEMB_DTYPE): GEMM operand format of the engines.  Default f16: of the two 16-bit formats (same
    kernels, same MFMA rate) it is the one whose results stay inside the reference tolerance -- loss and reconstructed pixels within
    1e-3 of the fp32 path (7e-4 / 8e-6 at config A; bf16: 6e-3 / 9e-5)."""
    name = os.environ.get("SKYEMB_DTYPE") or config['TRAINING'].get('compute_dtype', default)
    name = name.lower()
    if name in ("bf16", "bfloat16"):
        return torch.bfloat16
    if name in ("f16", "fp16", "float16", "half"):
        return torch.float16
    if name in ("f32", "fp32", "float32"):
        return torch.float32
    raise ValueError(f"compute_dtype must be f16, bf16 or f32, got {name!r}")


def build_model(config, model_filename, device, build_optimizer=False):
    """utils/mim_vit.py:19-151: same signature and return tuple."""
    norm_pix_loss = str2bool(config['TRAINING']['norm_pix_loss'])
    img_size = int(config['ARCHITECTURE']['img_size'])
    pixel_mean = float(config['ARCHITECTURE']['pixel_mean'])
    pixel_std = float(config['ARCHITECTURE']['pixel_std'])
    num_channels = int(config['ARCHITECTURE']['num_channels'])
    embed_dim = int(config['ARCHITECTURE']['embed_dim'])
    patch_size = int(config['ARCHITECTURE']['patch_size'])
    model_type = config['ARCHITECTURE']['model_type']
    loss_fn = config['TRAINING']['loss_fn']
    attn_pool = str2bool(config['ARCHITECTURE'].get('attn_pool', 'False'))   # absent in older inis
    ra_dec = str2bool(config['ARCHITECTURE'].get('ra_dec', 'False'))

    if model_type not in MODEL_TYPES:
        raise KeyError(f"model_type {model_type!r} is not one of {sorted(MODEL_TYPES)}")
    arch = MODEL_TYPES[model_type]
    if torch.device(device).type != "cuda":
        raise RuntimeError("sky_embeddings_amd runs its hot path in HIP kernels only: a GPU device is required "
                           "(there is no CPU fallback)")
    model = MaskedAutoencoderViT(img_size=img_size, patch_size=patch_size, in_chans=num_channels, embed_dim=embed_dim,
                                 depth=arch["depth"], num_heads=arch["num_heads"],
                                 decoder_embed_dim=arch["decoder_embed_dim"], decoder_depth=arch["decoder_depth"],
                                 decoder_num_heads=arch["decoder_num_heads"], mlp_ratio=4,
                                 norm_pix_loss=norm_pix_loss, simmim=arch["simmim"], loss_fn=loss_fn,
                                 pixel_mean=pixel_mean, pixel_std=pixel_std, attn_pool=attn_pool, ra_dec=ra_dec,
                                 device=device, compute_dtype=_compute_dtype(config))
    model = _DataParallelShim(model)

    if build_optimizer:
        total_batch_iters = int(float(config['TRAINING']['total_batch_iters']))
        weight_decay = float(config['TRAINING']['weight_decay'])
        init_lr = float(config['TRAINING']['init_lr'])
        final_lr_factor = float(config['TRAINING']['final_lr_factor'])
        optimizer = FusedAdamW(model.module.engine, lr=init_lr, betas=(0.9, 0.95), weight_decay=weight_decay)
        lr_scheduler = CosineLR(optimizer, int(total_batch_iters), eta_min=init_lr / final_lr_factor)
        model, losses, cur_iter = load_model(model, model_filename, optimizer, lr_scheduler)
        return model, losses, cur_iter, optimizer, lr_scheduler
    model, losses, cur_iter = load_model(model, model_filename)
    return model, losses, cur_iter


def load_model(model, model_filename, optimizer=None, lr_scheduler=None):
    """utils/mim_vit.py:154-181: resume from ``models/<name>.pth.tar`` when it exists."""
    if os.path.exists(model_filename):
        print('\nLoading saved model weights...')
        checkpoint = torch.load(model_filename, map_location=lambda storage, loc: storage, weights_only=False)
        losses = defaultdict(list, dict(checkpoint['losses']))
        cur_iter = checkpoint['batch_iters'] + 1
        if optimizer is not None:
            optimizer.load_state_dict(checkpoint['optimizer'])
        if lr_scheduler is not None:
            lr_scheduler.load_state_dict(checkpoint['lr_scheduler'])
        model.module.load_state_dict(checkpoint['model'])
    else:
        print('\nStarting fresh model to train...')
        losses = defaultdict(list)
        cur_iter = 1
    return model, losses, cur_iter


def patch_mean_and_var(imgs):
    """utils/mim_vit.py:614-627 (torch; used by denorm_imgs for visualisation only)."""
    ok = ~torch.isnan(imgs)
    cnt = ok.sum(dim=-1, keepdim=True)
    zero = torch.tensor(0.0, device=imgs.device)
    mean = torch.where(ok, imgs, zero).sum(dim=-1, keepdim=True) / cnt
    var = (torch.where(ok, imgs - mean, zero) ** 2).sum(dim=-1, keepdim=True) / cnt
    return mean, var


def undo_pixel_norm(original_images, normalized_images, model):
    """utils/mim_vit.py:629-649."""
    original_images = model.patchify(original_images)
    normalized_images = model.patchify(normalized_images)
    mean, var = patch_mean_and_var(original_images)
    return model.unpatchify(normalized_images * (var + 1.e-6) ** .5 + mean)


def mae_vit_base(**kw):
    return MaskedAutoencoderViT(**{**MODEL_TYPES_KW("base"), **kw})


def MODEL_TYPES_KW(name):
    a = MODEL_TYPES[name]
    return dict(depth=a["depth"], num_heads=a["num_heads"], decoder_embed_dim=a["decoder_embed_dim"],
                decoder_depth=a["decoder_depth"], decoder_num_heads=a["decoder_num_heads"], mlp_ratio=4,
                simmim=a["simmim"])
