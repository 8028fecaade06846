"""Thin torch-tensor front end of the C ABI (pointers + sizes in, nothing allocated inside).

PyTorch is used only for device memory and streams; every op below runs a hand-written gfx950
kernel from libskyemb.so on torch's current stream and raises if the library is missing.
"""
from __future__ import annotations

import ctypes
import os

import torch

from . import _lib
from ._lib import ACT_DGELU, ACT_GELU, ACT_NONE, BF16, F16, F32, KC, RC, AdamwDesc, GemmArgs, GemmGroupInfo, LnBwdSide, check, lib

TORCH_DTYPE = {BF16: torch.bfloat16, F32: torch.float32, F16: torch.float16}
LP_DTYPES = (torch.bfloat16, torch.float16)     # the two 16-bit operand formats of the MFMA kernels (SKYEMB_BF16 / SKYEMB_F16)


def dtype_code(t: torch.dtype) -> int:
    if t == torch.bfloat16:
        return BF16
    if t == torch.float32:
        return F32
    if t == torch.float16:
        return F16
    raise TypeError(f"unsupported activation dtype {t}")


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    if t is None:
        return None
    assert t.is_cuda, "libskyemb ops need device tensors (there is no CPU fallback)"
    return t.data_ptr()


# SKYEMB_PREFETCH=0: no prefetch hints in the launches' arguments (read whenever an argument struct is built -- once per launch of a
# captured step -- so that bench.py can A/B the hint inside one process)
def _prefetch_on():
    import os
    return os.environ.get("SKYEMB_PREFETCH", "1") != "0"


def gemm_args(A, B, *, M, N, K, a_layout=KC, b_layout=KC, lda=None, ldb=None, alpha=1.0, bias=None, table=None,
              tab_row=None, ldt=0, dst_row=None, resid=None, ldr=0, aux=None, ldaux=0, act=ACT_NONE, out_f32=None, ldo32=0,
              out=None, ldo=0, out2=None, ldo2=0, tile=0, colsum_a=None, ws=None, split_k=0, prefetch=None):
    """skyemb_gemm_args for C[M,N] = alpha * A[M,K] B[N,K]^T with fused epilogue (see include/skyemb.h).  prefetch: a tensor a LATER
    launch reads (the next layer's weights): touched by this launch so that it is in the memory-side cache by then (a hint)."""
    g = GemmArgs()
    g.A, g.B = _p(A), _p(B)
    g.lda = lda if lda is not None else (K if a_layout == KC else M)
    g.ldb = ldb if ldb is not None else (K if b_layout == KC else N)
    g.a_layout, g.b_layout = a_layout, b_layout
    g.M, g.N, g.K = M, N, K
    g.dtype = dtype_code(A.dtype)
    assert B.dtype == A.dtype
    g.alpha = alpha
    g.bias, g.table, g.tab_row, g.ldt, g.dst_row = _p(bias), _p(table), _p(tab_row), ldt, _p(dst_row)
    g.resid, g.ldr, g.aux, g.ldaux, g.act = _p(resid), ldr, _p(aux), ldaux, act
    g.out_f32, g.ldo32, g.out, g.ldo, g.out2, g.ldo2 = _p(out_f32), ldo32 or N, _p(out), ldo or N, _p(out2), ldo2 or N
    g.tile = tile
    g.colsum_a = _p(colsum_a)
    g.ws, g.ws_bytes, g.split_k = _p(ws), (ws.numel() * ws.element_size() if ws is not None else 0), split_k
    if prefetch is not None and _prefetch_on():
        g.prefetch, g.prefetch_bytes = _p(prefetch), prefetch.numel() * prefetch.element_size() // 4 * 4
    return g


def gemm(A, B, **kw):
    """C[M,N] = alpha * A[M,K] B[N,K]^T with fused epilogue (see include/skyemb.h)."""
    g = gemm_args(A, B, **kw)
    check(lib().skyemb_gemm(ctypes.byref(g), _stream()), "skyemb_gemm")


class GemmGroup:
    """Several independent GEMMs as ONE launch (skyemb_gemm_group_*): one tile shape (``tile`` = 0 lets the plan choose),
    data-gradient (KC.RC) and weight-gradient (RC.RC) problems may share a launch.  Built once from gemm_args(...)
    structs -- the device blob holds the raw pointers, so the operand buffers must stay allocated -- and replayed with
    launch().  `ok` is False when a problem is outside the grouped subset (launch them singly)."""

    def __init__(self, args, device, tile=0, adamw=None, side=None, ln_bwd=None):
        """adamw: an _lib.AdamwDesc -- the optimiser step fused into the launch's epilogue (weight-gradient groups of a
        single-process run: skyemb_gemm_group_plan_adamw).  side = (own_step, lo, hi, blocks) with adamw: the step of the flat
        slice [lo, hi) rides in the launch as a side job of `blocks` extra workgroups (skyemb_gemm_group_plan_side_adamw);
        own_step: the launch's own tiles are stepped in their epilogue too, else stored as gradients.  ln_bwd = dict(dy, x, gamma,
        mean, rstd, g_in, g_out, g_lp, part, M, D): a LayerNorm backward that does not depend on the launch's tiles (a block's
        norm1) rides in it as a side job instead of a launch of its own (skyemb_gemm_group_attach_ln_bwd); `ln_side` says whether
        the plan took it (else the caller launches ops.layernorm_bwd as before)."""
        n = len(args)
        arr = (GemmArgs * n)(*args)
        nbytes = lib().skyemb_gemm_group_blob_bytes(n)
        host = torch.zeros(nbytes, dtype=torch.uint8)
        self.info = GemmGroupInfo()
        if side is not None:
            own, lo, hi, blocks = side
            rc = lib().skyemb_gemm_group_plan_side_adamw(arr, n, tile, ctypes.byref(adamw), int(bool(own)), int(lo), int(hi), int(blocks),
                                                         host.data_ptr(), nbytes, ctypes.byref(self.info))
        elif adamw is not None:
            rc = lib().skyemb_gemm_group_plan_adamw(arr, n, tile, ctypes.byref(adamw), host.data_ptr(), nbytes, ctypes.byref(self.info))
        else:
            rc = lib().skyemb_gemm_group_plan(arr, n, tile, host.data_ptr(), nbytes, ctypes.byref(self.info))
        self.ok = rc == 0
        if rc > 0:
            check(rc, "skyemb_gemm_group_plan")
        # workgroups that compute TILES (the grid also holds the side jobs' workgroups)
        self.tile_blocks = self.info.total_blocks - (int(side[3]) if side is not None else 0)
        self.ln_side = False
        if self.ok and ln_bwd is not None and ln_bwd["dy"].dtype in LP_DTYPES:
            rec = LnBwdSide()
            for k in ("dy", "x", "gamma", "mean", "rstd", "g_in", "g_out", "g_lp", "part"):
                setattr(rec, k, _p(ln_bwd[k]))
            rec.M, rec.D = ln_bwd["M"], ln_bwd["D"]
            rc2 = lib().skyemb_gemm_group_attach_ln_bwd(host.data_ptr(), nbytes, ctypes.byref(self.info), ctypes.byref(rec))
            if rc2 > 0:
                check(rc2, "skyemb_gemm_group_attach_ln_bwd")
            self.ln_side = rc2 == 0
        self.total_blocks = self.info.total_blocks
        self.blob = host.to(device) if self.ok else None

    def launch(self):
        check(lib().skyemb_gemm_group_launch(self.blob.data_ptr(), ctypes.byref(self.info), _stream()), "skyemb_gemm_group_launch")


GEMM_COUNT_NAMES = ("fallback", "pipe", "tile256", "group", "group256", "splitk")


def gemm_launch_counts(reset=False):
    """{family: launches so far in this process} (include/skyemb.h, SKYEMB_GEMM_COUNT_*): a diagnostic the parity tests read."""
    buf = (ctypes.c_longlong * 8)()
    check(lib().skyemb_gemm_launch_counts(ctypes.cast(buf, ctypes.c_void_p), 8, int(reset)), "skyemb_gemm_launch_counts")
    return dict(zip(GEMM_COUNT_NAMES, list(buf)))


def colsum(X, M, N, out, ldx=None):
    code = dtype_code(X.dtype)
    check(lib().skyemb_colsum(_p(X), code, ldx if ldx is not None else N, M, N, _p(out), _stream()), "skyemb_colsum")


def random_mask_from_noise(noise, keep, ids_restore, mask, ids_keep, dec_dst=None, dec_tab=None, n_extra=1):
    B, L = noise.shape
    check(lib().skyemb_random_mask_from_noise(_p(noise), B, L, keep, _p(ids_restore), _p(mask), _p(ids_keep),
                                              _p(dec_dst), _p(dec_tab), n_extra, _stream()), "skyemb_random_mask_from_noise")


def augment(imgs, out, params, nan_mask, noise, A):
    B, C, S, _ = imgs.shape
    check(lib().skyemb_augment(_p(imgs), _p(out), _p(params), _p(nan_mask), _p(noise), B, C, S, A, _stream()), "skyemb_augment")


def tile_cutouts(tile, big_endian, h0, w0, S, out, lo=None, hi=None):
    """out [n, C, S, S] = windows of the HBM-resident survey tile [C, H, W] (4-byte words; big_endian [C] int32 marks planes
    still in FITS byte order), clipped at lo / hi (NaN kept)."""
    C, H, W = tile.shape
    check(lib().skyemb_tile_cutouts(_p(tile), _p(big_endian), C, H, W, _p(h0), _p(w0), h0.numel(), S,
                                    0.0 if lo is None else float(lo), 0.0 if hi is None else float(hi), int(lo is not None),
                                    int(hi is not None), _p(out), _stream()), "skyemb_tile_cutouts")


def attnpool_q(latent, Wq, bq, q):
    """q [D] = Wq latent + bq: the sample-independent query of the attention pool (timm AttentionPoolLatent)."""
    check(lib().skyemb_attnpool_q(_p(latent), _p(Wq), _p(bq), _p(q), q.numel(), _stream()), "skyemb_attnpool_q")


def attnpool_fwd(q, kv, out, prob, B, N, H, hd):
    check(lib().skyemb_attnpool_fwd(_p(q), _p(kv), dtype_code(kv.dtype), _p(out), _p(prob), B, N, H, hd, _stream()), "skyemb_attnpool_fwd")


def attnpool_bwd(q, kv, dout, prob, dkv, dq_part, B, N, H, hd):
    check(lib().skyemb_attnpool_bwd(_p(q), _p(kv), dtype_code(kv.dtype), _p(dout), _p(prob), _p(dkv), _p(dq_part), B, N, H, hd,
                                    _stream()), "skyemb_attnpool_bwd")


def attnpool_q_bwd(dq_part, latent, Wq, dWq, dbq, dlatent, ws):
    B, D = dq_part.shape
    check(lib().skyemb_attnpool_q_bwd(_p(dq_part), B, _p(latent), _p(Wq), _p(dWq), _p(dbq), _p(dlatent), _p(ws), D, _stream()),
          "skyemb_attnpool_q_bwd")


def simmim_mask_from_noise(noise, ratio_u, max_ratio, grid, p, out_mask):
    """Per-channel random patch masks for SimMIM (utils/dataloaders.py:197-219) from uniform draws: noise [B,C,L], ratio_u [B]."""
    B, C, L = noise.shape
    check(lib().skyemb_simmim_mask_from_noise(_p(noise), _p(ratio_u), max_ratio, B, C, L, grid, p, _p(out_mask), _stream()),
          "skyemb_simmim_mask_from_noise")


def patch_gather(imgs, pmv, ids_keep, out, p, keep, pixel_mean, pixel_std):
    B, C, H, W = imgs.shape
    check(lib().skyemb_patch_gather(_p(imgs), _p(pmv), _p(ids_keep), _p(out), dtype_code(out.dtype), B, C, H, W, p,
                                    keep, pixel_mean, pixel_std, _stream()), "skyemb_patch_gather")


def patch_gather_bwd_pmv(imgs, ids_keep, drows, partial, dpmv, p, keep):
    B, C, H, W = imgs.shape
    check(lib().skyemb_patch_gather_bwd_pmv(_p(imgs), _p(ids_keep), _p(drows), _p(partial), _p(dpmv), B, C, H, W, p,
                                            keep, _stream()), "skyemb_patch_gather_bwd_pmv")


def patch_gather_blend(imgs, pmv, ids_keep, pixel_mask, out, p, keep, pixel_mean, pixel_std):
    """SimMIM input path: NaN fill, then x * (1 - mask) + pmv * mask (pixel_mask None == patch_gather)."""
    B, C, H, W = imgs.shape
    check(lib().skyemb_patch_gather_blend(_p(imgs), _p(pmv), _p(ids_keep), _p(pixel_mask), _p(out), dtype_code(out.dtype), B, C,
                                          H, W, p, keep, pixel_mean, pixel_std, _stream()), "skyemb_patch_gather_blend")


def patch_gather_bwd_pmv_blend(imgs, ids_keep, pixel_mask, drows, partial, dpmv, p, keep):
    B, C, H, W = imgs.shape
    check(lib().skyemb_patch_gather_bwd_pmv_blend(_p(imgs), _p(ids_keep), _p(pixel_mask), _p(drows), _p(partial), _p(dpmv), B,
                                                  C, H, W, p, keep, _stream()), "skyemb_patch_gather_bwd_pmv_blend")


def radec_token_fwd(ra_dec, W0, b0, W1, b1, pos_row, x_rows, row_stride, B, D, sh, z):
    """x_rows: fp32 view starting at the first RA/Dec token row; consecutive samples are row_stride floats apart."""
    check(lib().skyemb_radec_token_fwd(_p(ra_dec), _p(W0), _p(b0), _p(W1), _p(b1), _p(pos_row), _p(x_rows), row_stride, B, D,
                                       _p(sh), _p(z), _stream()), "skyemb_radec_token_fwd")


def radec_token_bwd(g_rows, row_stride, W1, sh, z, dz_ws, dW0, db0, dW1, db1, B, D):
    check(lib().skyemb_radec_token_bwd(_p(g_rows), row_stride, _p(W1), _p(sh), _p(z), _p(dz_ws), _p(dW0), _p(db0), _p(dW1),
                                       _p(db1), B, D, _stream()), "skyemb_radec_token_bwd")


def simmim_pixel_loss(imgs, pred_tok, pixel_mask, loss, dpred_tok, dtype, pred_img, ws, p, extra, pixel_mean, pixel_std,
                      norm_pix, loss_l1, pooled=False, dscale=1.0):
    B, C, H, W = imgs.shape
    check(lib().skyemb_simmim_pixel_loss(_p(imgs), _p(pred_tok), _p(pixel_mask), _p(loss), _p(dpred_tok), dtype, _p(pred_img),
                                         _p(ws), B, C, H, W, p, extra, pixel_mean, pixel_std, int(norm_pix), int(loss_l1),
                                         int(pooled), float(dscale), _stream()), "skyemb_simmim_pixel_loss")


def layernorm_fwd(x, gamma, beta, y, mean, rstd, M, D, eps, y32=None, dtype=None):
    code = dtype if dtype is not None else dtype_code(y.dtype)
    check(lib().skyemb_layernorm_fwd(_p(x), _p(gamma), _p(beta), _p(y), _p(y32), code, _p(mean), _p(rstd), M, D, eps,
                                     _stream()), "skyemb_layernorm_fwd")


def layernorm_bwd_blocks(M):
    return lib().skyemb_layernorm_bwd_blocks(M)


def layernorm_bwd(dy, x, gamma, mean, rstd, g_in, g_out, g_lp, part, dgamma, dbeta, M, D, dtype):
    dy_f32 = 1 if (dy.dtype == torch.float32 and dtype in (BF16, F16)) else 0
    check(lib().skyemb_layernorm_bwd(_p(dy), dy_f32, dtype, _p(x), _p(gamma), _p(mean), _p(rstd), _p(g_in), _p(g_out),
                                     _p(g_lp), _p(part), _p(dgamma), _p(dbeta), M, D, _stream()), "skyemb_layernorm_bwd")


def ln_reduce_items(entries, device):
    """Device tables of a batched column reduce from [(part, dgamma, dbeta, nblk, D), ...]; dbeta None = a single vector
    (part [nblk, D] -> dgamma: the bias-gradient partial sums of a grouped weight-gradient launch).
    -> (items: skyemb_ln_reduce_item x n (32 bytes each), blocks: int32 pairs {item, x | y << 16} per workgroup,
        first_block: [n + 1] host prefix of the items' workgroups)."""
    rows = [[part.data_ptr(), dg.data_ptr(), db.data_ptr() if db is not None else 0, nblk + (D << 32)] for part, dg, db, nblk, D in entries]
    blocks, first = [], [0]
    for k, (part, dg, db, nblk, D) in enumerate(entries):
        for y in range(2 if db is not None else 1):
            blocks += [[k, x | (y << 16)] for x in range((D + 31) // 32)]
        first.append(len(blocks))
    return (torch.tensor(rows, dtype=torch.int64).to(device), torch.tensor(blocks, dtype=torch.int32).to(device), first)


def layernorm_bwd_reduce_batch(table, first, count):
    """Items [first, first + count) of a table made by ln_reduce_items, one launch."""
    items, blocks, first_block = table
    b0, b1 = first_block[first], first_block[first + count]
    check(lib().skyemb_layernorm_bwd_reduce_batch(items.data_ptr(), blocks.data_ptr() + 8 * b0, b1 - b0, _stream()),
          "skyemb_layernorm_bwd_reduce_batch")


def mha_fwd(qkv, out, B, N, H, hd):
    check(lib().skyemb_mha_fwd(_p(qkv), _p(out), dtype_code(qkv.dtype), B, N, H, hd, _stream()), "skyemb_mha_fwd")


def mha_bwd(qkv, dout, dqkv, B, N, H, hd):
    check(lib().skyemb_mha_bwd(_p(qkv), _p(dout), _p(dqkv), dtype_code(qkv.dtype), B, N, H, hd, _stream()),
          "skyemb_mha_bwd")


def fill_mask_tokens(x, mask, mask_token, dec_pos, B, L, Dd, n_extra=1):
    check(lib().skyemb_fill_mask_tokens(_p(x), _p(mask), _p(mask_token), _p(dec_pos), B, L, Dd, n_extra, _stream()),
          "skyemb_fill_mask_tokens")


def gather_rows(src, idx, out, out_lp, n_rows, D):
    code = dtype_code(out_lp.dtype) if out_lp is not None else F32
    check(lib().skyemb_gather_rows(_p(src), _p(idx), _p(out), _p(out_lp), code, n_rows, D, _stream()),
          "skyemb_gather_rows")


def rowsum_select(src, ld, sel, row0, inner, outer_stride, n_rows, D, partial, out):
    check(lib().skyemb_rowsum_select(_p(src), ld, _p(sel), row0, inner, outer_stride, n_rows, D, _p(partial), _p(out),
                                     _stream()), "skyemb_rowsum_select")


def masked_patch_loss(imgs, pred, mask, loss, dpred, dpred32, dtype, ws, p, extra, pixel_mean, pixel_std, norm_pix,
                      loss_l1, dscale=1.0):
    B, C, H, W = imgs.shape
    check(lib().skyemb_masked_patch_loss(_p(imgs), _p(pred), _p(mask), _p(loss), _p(dpred), _p(dpred32), dtype, _p(ws),
                                         B, C, H, W, p, extra, pixel_mean, pixel_std, int(norm_pix), int(loss_l1),
                                         float(dscale), _stream()), "skyemb_masked_patch_loss")


def adamw(p, g, m, v, p_lp, n, n_decay, hyper, beta1, beta2, eps, wd, grad_scale=1.0, zero_grad=False, lr=0.0, bc1=1.0,
          bc2=1.0):
    """hyper: device fp32[4] {lr, 1-b1^t, 1-b2^t} (graph-safe) or None to pass lr/bc1/bc2 by value."""
    code = dtype_code(p_lp.dtype) if p_lp is not None else F32
    check(lib().skyemb_adamw(_p(p), _p(g), _p(m), _p(v), _p(p_lp), code, n, n_decay, _p(hyper), lr, bc1, bc2, beta1,
                             beta2, eps, wd, grad_scale, int(zero_grad), dtype_code(g.dtype), _stream()), "skyemb_adamw")


def set_scalars(dst, a, b=0.0, c=0.0, d=0.0):
    check(lib().skyemb_set_scalars(_p(dst), a, b, c, d, _stream()), "skyemb_set_scalars")


def cast(src, dst, n):
    check(lib().skyemb_cast(_p(src), _p(dst), dtype_code(dst.dtype), n, _stream()), "skyemb_cast")


def standardise(x, mu, sigma, out):
    N, D = x.shape
    check(lib().skyemb_standardise(_p(x), _p(mu), _p(sigma), _p(out), N, D, _stream()), "skyemb_standardise")


def weighted_norms(x, w, norms, xw_out=None):
    N, D = x.shape
    check(lib().skyemb_weighted_norms(_p(x), _p(w), _p(norms), _p(xw_out), N, D, _stream()), "skyemb_weighted_norms")


def cosine_topk_chunks(N, Q, D, k):
    return lib().skyemb_cosine_topk_chunks(N, Q, D, k)


def cosine_topk(tw, qn, bank, xn, k, eps, idx_offset, nchunks, part_s, part_i, thr0=None):
    Q, D = tw.shape
    N = bank.shape[0]
    check(lib().skyemb_cosine_topk(_p(tw), _p(qn), _p(bank), _p(xn), Q, N, D, k, eps, idx_offset, nchunks, _p(thr0),
                                   _p(part_s), _p(part_i), _stream()), "skyemb_cosine_topk")


def kth_largest_floor(x, k, out):
    Q, S = x.shape
    check(lib().skyemb_kth_largest_floor(_p(x), Q, S, k, _p(out), _stream()), "skyemb_kth_largest_floor")


def sample_floor_applicable(Q, S, D, k, *tensors):
    """The library's own predicate (shape limits AND the SKYEMB_TOPK_STREAM switch) plus the 16-byte alignment the streaming
    scorer needs of the tensors it reads rows from."""
    return bool(lib().skyemb_cosine_sample_floor_applicable(Q, S, D, k)) and all(t.data_ptr() % 16 == 0 for t in tensors)


def cosine_sample_floor(tw, qn, sample, sample_norms, k, eps, ws, out):
    """Pruning floor of a small-Q search from a row sample (tile maxima + one-wave selection): see include/skyemb.h."""
    Q, D = tw.shape
    S = sample.shape[0]
    check(lib().skyemb_cosine_sample_floor(_p(tw), _p(qn), _p(sample), _p(sample_norms), Q, S, D, k, eps, _p(ws), _p(out), _stream()),
          "skyemb_cosine_sample_floor")


def topk_prefilter_applicable(Q, N, D, k):
    return bool(lib().skyemb_topk_prefilter_applicable(Q, N, D, k))


def bank16_prepare(bank, xn):
    """fp16 image of the bank for the prefiltered many-query top-k: (bank16 [rows, D] half, rowp [rows, 4] float), both
    padded to whole tiles of rows (``skyemb_bank16_rowp_rows``; ``skyemb_bank16_bytes`` is the image's size)."""
    N, D = bank.shape
    rows = lib().skyemb_bank16_rowp_rows(N)
    assert lib().skyemb_bank16_bytes(N, D) == rows * D * 2
    bank16 = torch.empty(rows, D, device=bank.device, dtype=torch.float16)
    rowp = torch.empty(rows, 4, device=bank.device, dtype=torch.float32)
    check(lib().skyemb_bank16_prepare(_p(bank), _p(xn), N, D, _p(bank16), _p(rowp), _stream()), "skyemb_bank16_prepare")
    return bank16, rowp


def cosine_topk_prefiltered(tw, qn, bank, xn, bank16, rowp, k, eps, idx_offset, out_s, out_i, redo, thr0=None, ws=None):
    Q, D = tw.shape
    N = bank.shape[0]
    need = lib().skyemb_topk_prefilter_ws_bytes(Q, D, k)
    if ws is None or ws.numel() < need:
        ws = torch.empty(need, device=tw.device, dtype=torch.uint8)
    check(lib().skyemb_cosine_topk_prefiltered(_p(tw), _p(qn), Q, _p(bank), _p(xn), _p(bank16), _p(rowp), N, D, k, eps, idx_offset,
                                               _p(thr0), _p(ws), ws.numel(), _p(out_s), _p(out_i), _p(redo), _stream()),
          "skyemb_cosine_topk_prefiltered")
    return ws


def topk_merge(in_s, in_i, Q, nlists, k, out_s, out_i, ws=None):
    """ws: optional int32 [Q] scratch enabling the gather + block-sort path for many short lists."""
    check(lib().skyemb_topk_merge(_p(in_s), _p(in_i), Q, nlists, k, _p(out_s), _p(out_i), _p(ws), _stream()),
          "skyemb_topk_merge")


def cosine_scores(tw, qn, bank, xn, eps, scores):
    Q, D = tw.shape
    N = bank.shape[0]
    check(lib().skyemb_cosine_scores(_p(tw), _p(qn), _p(bank), _p(xn), Q, N, D, eps, _p(scores), _stream()),
          "skyemb_cosine_scores")
