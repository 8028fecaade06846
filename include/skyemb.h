/*
 * skyemb.h -- C ABI of libskyemb (MI355X / gfx950 hot path of sky_embeddings).
 *
 * The reference (teaghan/sky_embeddings) has no FFI layer: its hot path is Python
 * (torch + timm).  This header is the lower drop-in boundary the build inserts beneath
 * the reference's module API: every entry point below names the
 * reference lines (paths relative to the reference root) whose arithmetic it replaces.
 *
 * Conventions
 *   - plain C, raw DEVICE pointers + sizes, no torch types; caller owns every buffer
 *     (workspaces included); nothing is allocated or freed inside the library;
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it, no host
 *     synchronisation, safe under stream capture (hipGraph);
 *   - return 0 on success, non-zero on error with the text in skyemb_last_error()
 *     (thread-local);
 *   - `dtype` selects the ACTIVATION element type of the call: SKYEMB_BF16 / SKYEMB_F16
 *     (throughput modes: 16-bit operands on v_mfma_f32_16x16x32_{bf16,f16}, fp32 accumulate)
 *     or SKYEMB_F32 (parity mode: exact-fp32 v_mfma_f32_16x16x4_f32).  Statistics, losses, the
 *     residual stream, gradients of parameters and optimiser state are always fp32.
 */
#ifndef SKYEMB_H
#define SKYEMB_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SKYEMB_BF16 0
#define SKYEMB_F32 1
#define SKYEMB_F16 2 /* IEEE half operands on v_mfma_f32_16x16x32_f16 (same rate as bf16, 11-bit significand): the throughput mode
                        that holds loss and reconstructed pixels within 1e-3 of the fp32 reference.  Every
                        `dtype` argument below that accepts SKYEMB_BF16 accepts SKYEMB_F16 with the same layouts and constraints.
                        fp16 underflows where bf16 does not: the caller scales d loss / d pred (`dscale` of the loss entry points)
                        and removes the factor in skyemb_adamw's / skyemb_adamw_desc's grad_scale. */

/* operand layouts of skyemb_gemm (logical A[M,K], B[N,K]; C = A * B^T) */
#define SKYEMB_KC 0 /* k contiguous:   X(r,k) at X[r*ld + k] */
#define SKYEMB_RC 1 /* row contiguous: X(r,k) at X[k*ld + r] */

/* epilogue activations */
#define SKYEMB_ACT_NONE 0
#define SKYEMB_ACT_GELU 1  /* out = gelu(v) (exact erf); out2 = v (pre-activation)  */
#define SKYEMB_ACT_DGELU 2 /* out = v * gelu'(aux)                                  */

const char *skyemb_last_error(void);
int skyemb_version(void);
/* Measurement aid for bench.py: entry points of the kernel families in `mask` return 0 without launching (bit 0: the MFMA
 * GEMM launches), so a timed region with and without them gives the family's in-step time.  Returns the previous mask.
 * MEASUREMENT BUILD ONLY: libskyemb_measure.so (-DSKYEMB_MEASURE).  In the product library (libskyemb.so) the switch is not
 * compiled in: the call returns -1 with skyemb_last_error set and no launch can ever be skipped. */
int skyemb_debug_skip(int mask);
/* Diagnostic: launches issued so far by each GEMM kernel family of this process (out[i] for the slots below; reset != 0 zeroes
 * them after reading).  Tests use it to assert that a shape ran on the kernel it is meant to exercise (e.g. the 256 x 256 tile
 * at ViT-L width); no product path reads it. */
#define SKYEMB_GEMM_COUNT_FALLBACK 0 /* gemm.hip: register-staged kernel (fp32 parity mode, shapes outside the pipe subset) */
#define SKYEMB_GEMM_COUNT_PIPE 1     /* gemm_pipe_kernel: LDS-DMA ring, every tile shape but 256 x 256                       */
#define SKYEMB_GEMM_COUNT_256 2      /* gemm256_kernel / gemm256_wgrad_kernel: single 256 x 256 launches                     */
#define SKYEMB_GEMM_COUNT_GROUP 3    /* gemm_pipe_group_kernel: grouped launches on ring tiles                               */
#define SKYEMB_GEMM_COUNT_GROUP256 4 /* gemm256_group_kernel: grouped launches on 256 x 256 tiles                            */
#define SKYEMB_GEMM_COUNT_SPLITK 5   /* splitk_reduce_kernel                                                                 */
#define SKYEMB_GEMM_COUNT_SLOTS 8
int skyemb_gemm_launch_counts(long long *out, int n, int reset);

/* ---------------------------------------------------------------- GEMM ----
 * Replaces every nn.Linear / Conv2d(k=s=p) contraction of timm PatchEmbed / Block /
 * the MAE decoder (utils/mim_vit.py:206,231-233,269,276-281) and their autograd
 * backward (dgrad, wgrad) (utils/pretrain_fns.py:34).
 *
 *   v[m,n]   = alpha * sum_k A(m,k) * B(n,k)  + bias[n] + table[tab_row[m], n] + resid[orow, n]
 *   orow     = dst_row ? dst_row[m] : m        (orow < 0: row is not stored)
 *   out_f32[orow, n] / out[orow, n] / out2[orow, n] per `act` (see above).
 * A, B, aux, out, out2 have element type `dtype`; bias/table/resid/out_f32 are fp32.
 * Constraint: the contiguous extent of A and of B (K for KC, rows for RC) is a
 * multiple of 8 (bf16) / 4 (f32) elements and 16-byte aligned.
 */
typedef struct skyemb_gemm_args {
    const void *A;
    const void *B;
    int64_t lda, ldb;
    int32_t a_layout, b_layout; /* SKYEMB_KC / SKYEMB_RC */
    int32_t M, N, K;
    int32_t dtype;
    float alpha;
    const float *bias;          /* [N] or NULL */
    const float *table;         /* fp32 rows added by index, or NULL */
    const int32_t *tab_row;     /* [M] */
    int64_t ldt;
    const int32_t *dst_row;     /* [M] or NULL */
    const float *resid;         /* fp32 [*, ldr] or NULL (indexed by orow) */
    int64_t ldr;
    const void *aux;            /* dtype [M, ldaux] (ACT_DGELU), indexed by m */
    int64_t ldaux;
    int32_t act;
    float *out_f32;             /* or NULL */
    int64_t ldo32;
    void *out;                  /* dtype, or NULL */
    int64_t ldo;
    void *out2;                 /* dtype, or NULL (ACT_GELU pre-activation) */
    int64_t ldo2;
    int32_t tile;               /* 0 = auto, 64 or 128 */
    float *colsum_a;            /* optional, A must be RC: colsum_a[m] = sum_k A(m,k)  (bias gradient fused
                                   into the wgrad launch: A = dy, so this is sum over tokens of dy[:, m]) */
    void *ws;                   /* optional fp32 workspace enabling split-K: partial slabs [split][M][N] + [split][M]
                                   are summed by a second launch in a fixed order (deterministic), which applies
                                   the epilogue.  One workspace per stream. */
    int64_t ws_bytes;
    int32_t split_k;            /* 0 = auto (1 when ws == NULL), 1 = off, n = force n-way */
    int32_t prefetch_wgs;       /* resolved by the library (callers leave 0): workgroups behind the tiles that carry the prefetch hint */
    const void *prefetch;       /* optional hint, no effect on results: `prefetch_bytes` bytes (a multiple of 4, 4-byte aligned) that a LATER
                                   launch will read -- the next layer's weight matrix -- are touched once by this launch's workgroups
                                   (one 4-byte LDS-DMA read per 128-byte line, ahead of their first operand loads), so that they sit in
                                   the memory-side cache when that launch starts instead of coming from HBM inside its k-loops.
                                   Where the launch leaves workgroup slots of the device free, extra workgroups behind its tiles do the
                                   touching (the tiles' own request queues stay clear); else the tiles' waves do, ahead of their first loads.
                                   Honoured by the pipelined bf16 kernels (gemm_pipe.hip); ignored elsewhere. */
    int64_t prefetch_bytes;
} skyemb_gemm_args;

int skyemb_gemm(const skyemb_gemm_args *args, void *stream);

/* Grouped launch: n <= 32 independent 16-bit problems (all bf16 or all f16) run as ONE grid (the four weight-gradient GEMMs of a transformer
 * block fill the chip together: no split-K, no reduce launch).
 * skyemb_gemm_group_plan validates the problems and fills a HOST blob of skyemb_gemm_group_blob_bytes(n) bytes; the
 * caller copies it to device memory once (pointers inside are fixed) and replays skyemb_gemm_group_launch.
 * plan returns -1 (with skyemb_last_error set) when a problem is outside the subset: launch them singly then. */
typedef struct skyemb_gemm_group_info {
    int32_t total_blocks; /* grid size of the launch                                                          */
    int32_t tile;         /* tile code the plan chose (BM * 1000 + BN): 64064, 128064, 128128 or 256256        */
    int32_t class_mask;   /* operand-layout classes present: 1 KC.KC, 2 KC.RC (dgrad), 4 RC.RC (wgrad)        */
    int32_t reserved;     /* bit 0: the tiles' epilogue is the AdamW step (plan_adamw); bit 1: side jobs (plan_side_adamw, attach_ln_bwd);
                             bit 2: the problems are SKYEMB_F16 */
} skyemb_gemm_group_info;
int64_t skyemb_gemm_group_blob_bytes(int n);
/* tile = 0: chosen from the total tile count.  The problems share ONE tile shape and may mix the data-gradient
 * (KC.RC) and weight-gradient (RC.RC) layout classes: a Linear's dX and dW read the same dY and do not depend on
 * each other, so one launch computes both. */
int skyemb_gemm_group_plan(const skyemb_gemm_args *args, int n, int tile, void *blob_host, int64_t blob_bytes,
                           skyemb_gemm_group_info *info);
int skyemb_gemm_group_launch(const void *blob_dev, const skyemb_gemm_group_info *info, void *stream);
/* The same grouped weight-gradient launch with the optimiser step fused into its epilogue (one process per model replica only:
 * with N > 1 the gradients are summed over the ranks between backward and AdamW).  Every problem's out_f32 points into the flat
 * gradient buffer `g_base`; p / m / v / p_lp are the flat parameter, moment and low-precision-shadow buffers with the SAME element
 * offsets.  An output tile is then not stored as a gradient at all: the kernel reads p, m, v, applies torch.optim.AdamW's update
 * to them (utils/mim_vit.py:126-129, utils/pretrain_fns.py:36-41; bit-identical to skyemb_adamw on the stored gradient) and writes
 * p, m, v and the bf16 shadow -- 26 instead of 4 + 30 bytes per parameter through HBM, and no separate pass over those tensors.
 * hyper: device fp32[4] {lr, 1 - beta1^t, 1 - beta2^t} of the step being taken (written before the launch: graph-safe).
 * Elements of the flat buffers below n_decay are weight-decayed.  bias gradients (colsum_a) are still stored as gradients.
 * RC.RC problems only; plan it with skyemb_gemm_group_plan_adamw, launch it with skyemb_gemm_group_launch. */
typedef struct skyemb_adamw_desc {
    float *g_base, *p, *m, *v;
    void *p_lp;            /* 16-bit shadow of p, in the format of the group's problems (bf16 / f16) */
    const float *hyper;
    int64_t n_decay;
    float beta1, beta2, eps, weight_decay, grad_scale;
    int32_t enabled;       /* set by the plan */
} skyemb_adamw_desc;
int skyemb_gemm_group_plan_adamw(const skyemb_gemm_args *args, int n, int tile, const skyemb_adamw_desc *adamw, void *blob_host,
                                 int64_t blob_bytes, skyemb_gemm_group_info *info);
/* The optimiser step as a SIDE JOB of a grouped weight-gradient launch (round 5).  In the epilogue form above a tile's parameters are
 * stepped after that tile's own k-loop: every tile ends at the same moment, so the launch is a matrix phase followed by an HBM
 * phase.  Here `side_blocks` extra workgroups behind the launch's tiles step the slice [side_lo, side_hi) of the flat buffers --
 * the weights whose gradients the PREVIOUS grouped launch of the backward pass stored in g_base (skyemb_adamw on that slice, bit
 * for bit) -- while the tile workgroups multiply: they are dispatched into the slots the tiles leave free (a grouped launch
 * rarely fills the chip evenly: 192 tiles of 256 x 256 for 256 compute units at ViT-L) and into the tiles' slots as those finish.
 * own_step = 0: the launch's own tiles are stored as gradients (a later launch's side job, or skyemb_adamw, steps them);
 * own_step = 1: they are stepped in the epilogue as with skyemb_gemm_group_plan_adamw (the LAST block of a backward pass: nothing
 * follows it that could carry its step).  An empty side range (side_lo == side_hi, side_blocks = 0) is allowed.  RC.RC problems
 * only; launch with skyemb_gemm_group_launch.  info->reserved: bit 0 = own_step, bit 1 = side job. */
int skyemb_gemm_group_plan_side_adamw(const skyemb_gemm_args *args, int n, int tile, const skyemb_adamw_desc *adamw, int own_step,
                                      int64_t side_lo, int64_t side_hi, int side_blocks, void *blob_host, int64_t blob_bytes,
                                      skyemb_gemm_group_info *info);
/* A LayerNorm backward as a SIDE JOB of a planned grouped weight-gradient launch (round 5): the backward of a transformer block's
 * norm1 (timm Block: x + attn(norm1(x)); utils/mim_vit.py:231-233 through autograd) depends on the block's qkv data gradient only,
 * not on its weight gradients -- so instead of a launch of its own behind the grouped launch, its rows are taken by extra workgroups
 * inside it (bf16 compute dtype; same rows per four-wave block, same partial-sum table and same bits as skyemb_layernorm_bwd with
 * dgamma = dbeta = NULL: the caller reduces `part` as before).  Call AFTER skyemb_gemm_group_plan / _plan_adamw / _plan_side_adamw
 * on the same host blob (weight-gradient groups only); grows info->total_blocks and sets bit 1 of info->reserved.  D <= 1024 on 256 x 256 tiles, <= 768 on the ring tiles. */
typedef struct skyemb_ln_bwd_side {
    const void *dy;             /* [M, D] bf16: d loss / d (LayerNorm output) */
    const float *x, *gamma, *mean, *rstd;
    const float *g_in;          /* fp32 residual gradient added to the result (may equal g_out), or NULL */
    float *g_out;               /* [M, D] fp32 */
    void *g_lp;                 /* [M, D] bf16 copy of g_out, or NULL */
    float *part;                /* [2, skyemb_layernorm_bwd_blocks(M), D] */
    int32_t M, D;
} skyemb_ln_bwd_side;
int skyemb_gemm_group_attach_ln_bwd(void *blob_host, int64_t blob_bytes, skyemb_gemm_group_info *info, const skyemb_ln_bwd_side *ln);

/* column sums: out[n] = sum_m X[m,n]; X is `dtype` (bias gradients) or fp32 partials
 * (LayerNorm dgamma/dbeta second stage).  Replaces autograd's bias-gradient reductions. */
int skyemb_colsum(const void *X, int dtype, int64_t ldx, int M, int N, float *out, void *stream);

/* ----------------------------------------------------------- front end ----
 * utils/mim_vit.py:354-379 random_masking with the noise supplied by the caller
 * (ties -> lower index).  noise [B,L] -> ids_restore i64 [B,L], mask f32 [B,L]
 * (1 = removed), ids_keep i32 [B,keep] (first `keep` of the shuffle, in shuffle order).
 * Optional (NULL to skip) decoder un-shuffle maps for utils/mim_vit.py:446-453, one entry per
 * encoder token (the n_extra = 1 (cls) or 2 (cls, RA/Dec) extra tokens first): dec_dst i32 [B,n_extra+keep] = row of
 * the [B,n_extra+L] decoder sequence the token lands on, dec_tab i32 [B,n_extra+keep] = its decoder_pos_embed row. */
int skyemb_random_mask_from_noise(const float *noise, int B, int L, int keep, int64_t *ids_restore, float *mask,
                                  int32_t *ids_keep, int32_t *dec_dst, int32_t *dec_tab, int n_extra, void *stream);
/* SimMIM mask generator on the device (replaces utils/dataloaders.py:197-219 MaskGenerator.__call__, which runs per item in
 * the loader workers): per sample ratio = ratio_u[b] * max_ratio, count = ceil(L * ratio); per channel the `count` patches
 * with the smallest noise[b, c, :] are masked (a uniformly random subset, like randperm(L)[:count]); out_mask float 0 / 1
 * [B, C, grid*p, grid*p].  noise [B, C, L] and ratio_u [B] are uniform [0, 1) draws supplied by the caller. */
int skyemb_simmim_mask_from_noise(const float *noise, const float *ratio_u, double max_ratio, int B, int C, int L, int grid, int p,
                                  float *out_mask, void *stream);

/* utils/mim_vit.py:385-392 + the im2row half of timm PatchEmbed (:206,402): for every kept
 * patch (b, ids_keep[b,j]) write the row  out[b*keep + j, c*p*p + py*p + px] =
 * isnan(x) ? pmv[c,py,px] : (x - mean)/std   in `dtype`.  ids_keep == NULL: all L patches in order. */
int skyemb_patch_gather(const float *imgs, const float *pmv, const int32_t *ids_keep, void *out, int dtype, int B,
                        int C, int H, int W, int p, int keep, float pixel_mean, float pixel_std, void *stream);

/* SimMIM mode (utils/mim_vit.py:394-399): same with  x = x * (1 - mask) + pmv * mask  after the NaN fill;
 * pixel_mask fp32 [B,C,H,W], 1 = hidden.  The matching patch_mask_values gradient weights drows by
 * d x / d pmv = isnan(pixel) ? 1 : pixel_mask. */
int skyemb_patch_gather_blend(const float *imgs, const float *pmv, const int32_t *ids_keep, const float *pixel_mask, void *out,
                              int dtype, int B, int C, int H, int W, int p, int keep, float pixel_mean, float pixel_std,
                              void *stream);
int skyemb_patch_gather_bwd_pmv_blend(const float *imgs, const int32_t *ids_keep, const float *pixel_mask, const float *drows,
                                      float *partial, float *dpmv, int B, int C, int H, int W, int p, int keep, void *stream);

/* RA/Dec token (utils/mim_vit.py:209-216, 410-414; utils/location_encoder.py:138-243): real spherical harmonics
 * (l < 5, 25 features) -> sin(30 (W0 sh + b0)) [8] -> W1 h + b1 (+ pos_row) written to x[b * row_stride + d], d < D.
 * sh [B,25] and z [B,8] (pre-sine) are saved for the backward, which takes g = d loss / d token rows (same stride) and
 * returns the four parameter gradients (dz_ws: fp32 [B,8] scratch).  Batch reductions run in a fixed order. */
int skyemb_radec_token_fwd(const float *ra_dec, const float *W0, const float *b0, const float *W1, const float *b1,
                           const float *pos_row, float *x, int64_t row_stride, int B, int D, float *sh, float *z, void *stream);
int skyemb_radec_token_bwd(const float *g, int64_t row_stride, const float *W1, const float *sh, const float *z, float *dz_ws,
                           float *dW0, float *db0, float *dW1, float *db1, int B, int D, void *stream);

/* gradient of patch_mask_values: dpmv[c,py,px] = sum over gathered NaN pixels of drows (fp32
 * [B*keep, C*p*p]); deterministic two-stage reduction, `partial` is fp32 [B, C*p*p]. */
int skyemb_patch_gather_bwd_pmv(const float *imgs, const int32_t *ids_keep, const float *drows, float *partial,
                                float *dpmv, int B, int C, int H, int W, int p, int keep, void *stream);

/* -------------------------------------------------------- LayerNorm -------
 * nn.LayerNorm(eps=1e-6) of timm Block.norm1/norm2 and the final norms
 * (utils/mim_vit.py:236,280,429,458).  x fp32 [M,D] -> y dtype [M,D] (+ y32 fp32 copy if
 * non-NULL), mean/rstd fp32 [M]. */
int skyemb_layernorm_fwd(const float *x, const float *gamma, const float *beta, void *y, float *y32, int dtype,
                         float *mean, float *rstd, int M, int D, float eps, void *stream);

/* dx = LN'(dy); g_out = (g_in ? g_in : 0) + dx (fp32; g_out may alias g_in); g_lp = dtype copy
 * of g_out (or NULL).  dy is `dtype` (or fp32 when dy_is_f32).  dgamma/dbeta: deterministic two-stage
 * reduction through part[2, nblk, D] (fp32 workspace, nblk = skyemb_layernorm_bwd_blocks(M)). */
int skyemb_layernorm_bwd_blocks(int M);
/* dgamma == NULL: leave the partial sums in `part` for skyemb_layernorm_bwd_reduce_batch, which finishes the
 * dgamma / dbeta of many LayerNorms (e.g. all of one backward stage) in ONE launch; `items` is a DEVICE array. */
typedef struct {
    const float *part;   /* [2, nblk, D] written by skyemb_layernorm_bwd */
    float *dgamma, *dbeta;      /* dbeta == NULL: a single vector (part [nblk][D] -> dgamma) */
    int32_t nblk, D;
} skyemb_ln_reduce_item;
/* blocks: device int32 pairs {item index into `items`, x | y << 16}, one per workgroup: columns [32 x, +32) of half y (0 = dgamma,
 * 1 = dbeta) of that item.  The caller lists ceil(D / 32) pairs per half of every item of the stage (a flat list: items of
 * different widths share the launch without idle workgroups). */
int skyemb_layernorm_bwd_reduce_batch(const skyemb_ln_reduce_item *items, const int32_t *blocks, int n_blocks, void *stream);
int skyemb_layernorm_bwd(const void *dy, int dy_is_f32, int dtype, const float *x, const float *gamma,
                         const float *mean, const float *rstd, const float *g_in, float *g_out, void *g_lp,
                         float *part, float *dgamma, float *dbeta, int M, int D, void *stream);

/* -------------------------------------------------------- attention -------
 * timm Attention core / F.scaled_dot_product_attention on tiny sequences (N = 5, 17, 65, 66):
 * qkv dtype [B, N, 3, H, hd] -> out dtype [B, N, H*hd];  softmax(q k^T hd^-0.5) v in fp32.
 * Backward recomputes the probabilities from q, k. */
int skyemb_mha_fwd(const void *qkv, void *out, int dtype, int B, int N, int H, int hd, void *stream);
int skyemb_mha_bwd(const void *qkv, const void *dout, void *dqkv, int dtype, int B, int N, int H, int hd,
                   void *stream);

/* -------------------------------------------------------- decoder glue ----
 * utils/mim_vit.py:446-453: rows of the decoder sequence that hold a mask token:
 * x[b, 1+l, :] = mask_token + dec_pos[1+l] for every l with mask[b,l]==1 (x fp32 [B, 1+L, Dd]). */
int skyemb_fill_mask_tokens(float *x, const float *mask, const float *mask_token, const float *dec_pos, int B,
                            int L, int Dd, int n_extra, void *stream);
/* gather fp32 rows: out[i, :] = src[idx[i], :] (also emits a dtype copy when out_lp != NULL) */
int skyemb_gather_rows(const float *src, const int32_t *idx, float *out, void *out_lp, int dtype, int n_rows,
                       int D, void *stream);
/* selected row sum (d mask_token, d cls_token): out[d] = sum_i [sel == NULL || sel[i] != 0] src[r(i)*ld + d],
 * r(i) = row0 + (i / inner) * outer_stride + (i % inner), i in [0, n_rows); `partial` fp32 [256, D]. */
int skyemb_rowsum_select(const float *src, int64_t ld, const float *sel, int row0, int inner, int outer_stride,
                         int n_rows, int D, float *partial, float *out, void *stream);

/* ------------------------------------------------------------- loss -------
 * utils/mim_vit.py:326-338,473-521,614-627: patchify + NaN-aware per-patch mean / biased
 * variance normalisation + masked MSE (loss_l1 == 0) or L1 + NaN exclusion.
 * pred fp32 [B, Nd, pv] with the first `extra` rows of each sample ignored (cls / ra_dec).
 * Outputs: loss (fp32 scalar, device), dpred dtype/fp32 [B, Nd, pv] (extra + unmasked rows
 * zero) = dscale * d loss / d pred.  NaN target elements contribute zero gradient (DESIGN.md deviation).
 * dscale: static loss scale of the backward pass (1 = none; a power of two for SKYEMB_F16, whose data gradients would
 * underflow otherwise -- every later step of backward is linear in dpred, so the caller divides it out again through
 * grad_scale of skyemb_adamw / skyemb_adamw_desc); `loss` itself is never scaled.
 * `ws` fp32 workspace of 4*B*L + 4 floats.  p % 4 == 0, W % 4 == 0, 16-byte aligned buffers. */
int skyemb_masked_patch_loss(const float *imgs, const float *pred, const float *mask, float *loss, void *dpred,
                             float *dpred32, int dtype, float *ws, int B, int C, int H, int W, int p, int extra,
                             float pixel_mean, float pixel_std, int norm_pix, int loss_l1, float dscale, void *stream);

/* SimMIM mode (utils/mim_vit.py:469, 480-493, 497-521): pixel-wise loss on the head GEMM's token rows
 * pred_tok fp32 [B*(L+extra), C*p*p] (column c*p*p + py*p + px == PixelShuffle(p) of the Conv1x1 output) with
 * weights w = pixel_mask where the target is not NaN:  loss = sum(w*l) / (sum(w) + 1e-5), optional per-patch
 * normalisation of the target.  Outputs: loss, dpred_tok (dtype, same layout; extra rows zero; may be NULL),
 * pred_img fp32 [B,C,H,W] (the reference's `pred`; may be NULL).  ws: 4*B*L + 4 floats.
 * pooled != 0 (attention-pooled models, utils/mim_vit.py:250): pred_tok / dpred_tok are [B, C*H*W], one row per image laid
 * out like the image (PixelShuffle(img_size) of the head's output); p stays the patch size of the normalisation; extra = 0.
 * dscale: as in skyemb_masked_patch_loss (dpred_tok = dscale * d loss / d pred_tok). */
int skyemb_simmim_pixel_loss(const float *imgs, const float *pred_tok, const float *pixel_mask, float *loss, void *dpred_tok,
                             int dtype, float *pred_img, float *ws, int B, int C, int H, int W, int p, int extra,
                             float pixel_mean, float pixel_std, int norm_pix, int loss_l1, int pooled, float dscale, void *stream);

/* dst[0..3] = {a, b, c, d} by a kernel launch (the values travel as kernel arguments): how the scalars of optimiser step t --
 * lr, 1 - beta1^t, 1 - beta2^t -- reach the device buffer `hyper` of skyemb_adamw / skyemb_adamw_desc in front of a replayed HIP
 * graph (torch.optim.AdamW keeps them on the host: utils/pretrain_fns.py:36-41). */
int skyemb_set_scalars(float *dst, float a, float b, float c, float d, void *stream);
/* ----------------------------------------------------------- optimiser ----
 * torch.optim.AdamW single-tensor update order (utils/mim_vit.py:126-129,
 * utils/pretrain_fns.py:36-41) over one flat fp32 parameter buffer:
 * elements [0, n_decay) use weight decay `wd`, the rest 0.  Step scalars {lr, 1-beta1^t, 1-beta2^t}
 * come from `hyper` (device fp32[4], for launches captured in a HIP graph) when non-NULL, else
 * from the lr/bc1/bc2 arguments.  Also refreshes the dtype shadow copy `p_lp` used by the GEMMs
 * (NULL to skip), scales gradients by grad_scale (DDP averaging, and the inverse of the backward pass's loss scale)
 * and optionally zeroes g.  `g` holds `grad_dtype` elements: SKYEMB_F32 (the flat gradient buffer the kernels write), or
 * SKYEMB_BF16 / SKYEMB_F16 (the copy a 16-bit gradient all-reduce worked on: half the xGMI bytes per step). */
int skyemb_adamw(float *p, void *g, float *m, float *v, void *p_lp, int dtype, int64_t n, int64_t n_decay,
                 const float *hyper, float lr, float bc1, float bc2, float beta1, float beta2, float eps, float wd,
                 float grad_scale, int zero_grad, int grad_dtype, void *stream);
int skyemb_cast(const float *src, void *dst, int dtype, int64_t n, void *stream);

/* ------------------------------------------------------------ input feeder -
 * utils/dataloaders.py:285-328 (H5Dataset.__getitem__): the reference opens the file and reads ONE cutout per python
 * call.  Here a minibatch of rows is gathered by native threads from the memory-mapped contiguous dataset into a (pinned)
 * staging buffer -- HOST function, no stream -- copied to the device once, then clipped at pixel_min / pixel_max (NaN
 * kept) and centre-cropped on the device (utils/dataloaders.py:293-300). */
int skyemb_gather_rows_host(const void *src, int64_t row_bytes, const int64_t *idx, int64_t n, int64_t src_rows, void *dst,
                            int nthreads);
/* Chunked HDF5 dataset -> contiguous row-major image (host threads; see hdf5_lite.py).  The reference's ETL writes its
 * train / validation files chunked (data_processing/2_create_h5_files.py:70-81: maxshape=(None, ...) + resize); they are
 * un-chunked once into a cache file and then served by the mmap fast path above.  file_base / file_bytes: the mapped file;
 * chunk i: byte address chunk_addr[i], element offsets chunk_off[i*rank ..], extent chunk_dims (edge chunks stored whole). */
int skyemb_h5_unchunk_host(const void *file_base, int64_t file_bytes, const int64_t *chunk_addr, const int64_t *chunk_off,
                           int64_t nchunks, int rank, const int64_t *chunk_dims, const int64_t *dset_dims, int elem_size, void *dst,
                           int nthreads);
/* The tiles of a FITS tile-compressed image (FITS 4.0 section 10; the reference reads such survey tiles through astropy / CFITSIO,
 * utils/dataloaders.py:418) -> integer pixels in the host's byte order, native threads over the tiles; see fits_lite.py, which
 * dequantises floating-point images afterwards.  HOST function.  codec 1 = RICE_1 (what fpack and astropy's CompImageHDU write by
 * default; pixels of `bytepix` = 1, 2 or 4 bytes coded in blocks of `blocksize` differences), 2 = PLIO_1 (integer masks), 3 =
 * HCOMPRESS_1 (both int32 pixels: bytepix = 4; an HCOMPRESS tile's stream carries its own dimensions, whose product must be npix[t];
 * for HCOMPRESS `blocksize` carries the SMOOTH flag of the image, 0 or 1).
 * Tile t: bytes [off[t], off[t] + len[t]) of `base` hold npix[t] pixels; they are written to dst + dst_off[t] * bytepix (dst holds
 * dst_pixels pixels). */
int skyemb_fits_decode_tiles_host(int codec, const void *base, int64_t base_bytes, const int64_t *off, const int64_t *len,
                                  const int64_t *npix, const int64_t *dst_off, int64_t ntiles, int bytepix, int blocksize, void *dst,
                                  int64_t dst_pixels, int nthreads);
/* Quantised floating-point tiles of such an image -> float32 pixels placed in the image (FITS 4.0 section 10.2): q * ZSCALE + ZZERO,
 * or (q - r + 0.5) * ZSCALE + ZZERO with the convention's subtractive dither (method 1 / 2; 2: q == -2147483646 is exactly 0; method 0:
 * none).  rand: the convention's 10 000 random numbers (fits_lite.dither_sequence), table_row[t]: the tile's 0-based row in the table.
 * Tile t: q + q_off[t], h[t] x w[t] pixels -> out[(y0[t] + y) * W + x0[t] + x]; has_blank[t] != 0: q == blank[t] -> NaN (both may be
 * NULL).  big_endian_out != 0: floats stored byte-swapped, the layout of an uncompressed FITS image.  HOST function. */
int skyemb_fits_dequantise_tiles_host(const int32_t *q, const int64_t *q_off, const int64_t *y0, const int64_t *x0, const int64_t *h,
                                      const int64_t *w, const int64_t *table_row, int64_t ntiles, const double *zscale, const double *zzero,
                                      const int32_t *blank, const uint8_t *has_blank, const float *rand, int method, int zdither0,
                                      float *out, int64_t H, int64_t W, int big_endian_out, int nthreads);
/* Survey-tile sampler: FitsDataset.__getitem__ (utils/dataloaders.py:589-654) cuts `cutouts_per_tile` windows out of a
 * multi-band FITS tile with a python loop (random_cutouts :449-476 / overlapping_cutouts :507-536) and clips them (:618-621).
 * Here the tile [C, H, W] sits in HBM as the files' own 4-byte words (big_endian[c] != 0: plane c still holds FITS
 * big-endian floats; a missing band is a plane of NaNs) and ONE launch writes out [n, C, S, S] = tile[:, h0[i]:+S, w0[i]:+S],
 * decoded and clipped at lo / hi (NaN kept).  h0 / w0: int32 device arrays, window origins (caller guarantees they fit). */
int skyemb_tile_cutouts(const void *tile, const int *big_endian, int C, int H, int W, const int *h0, const int *w0, int n, int S,
                        float lo, float hi, int use_lo, int use_hi, float *out, void *stream);
int skyemb_clip_crop(const float *src, float *dst, int64_t n_planes, int Hs, int Ws, int size, float lo, float hi,
                     int use_lo, int use_hi, void *stream);

/* Target augmentations of the similarity search (utils/dataloaders.py:14-106 get_augmentations as applied by
 * utils/eval_fns.py:88-108: per sample the original + A augmented copies): flips, RandomResizedCrop back to S x S
 * (crop + anti-aliased bilinear resize), brightness factor, additive noise, channels set to NaN -- one launch per batch.
 *   imgs [B, C, S, S] -> out [B * (1 + A), C, S, S], copy 0 of every sample unchanged
 *   params [B * (1 + A), 8] = {flip_h, flip_v, crop_top, crop_left, crop_h, crop_w, brightness, sigma} (drawn by the caller)
 *   nan_mask [B * (1 + A)]: bit c set -> channel c is NaN;  noise [B * (1 + A), C, S, S] standard normal draws or NULL */
int skyemb_augment(const float *imgs, float *out, const float *params, const int32_t *nan_mask, const float *noise, int B, int C,
                   int S, int A, void *stream);

/* ------------------------------------------------------ similarity search -
 * utils/similarity.py:98-102: standardise bank rows in place or to `out`:
 * (x - mu) / (sigma + 1e-8). */
int skyemb_standardise(const float *x, const float *mu, const float *sigma, float *out, int64_t N, int D,
                       void *stream);
/* utils/similarity.py:166-167: weighted norms sqrt(sum_d w x^2) in the oracle's fixed fma order:
 * rows[n] (bank) ; for queries also tw = w * t (utils/similarity.py:163). w == NULL -> ones. */
int skyemb_weighted_norms(const float *x, const float *w, float *norms, float *xw_out, int64_t N, int D,
                          void *stream);
/* utils/similarity.py:149-172 + 18-35 fused: per-(query tile, bank chunk) exact partial top-k.
 *   tw [Q,D] (= w*t), qn [Q], bank [N,D], xn [N]  ->  part_s f32 / part_i i64 [Q, nchunks, k]
 * (nchunks = skyemb_cosine_topk_chunks(N, Q, D, k): bank chunks of the tiled kernel, or one list per
 * wavefront of the bank-streaming kernel used when Q <= 16).
 * thr0 (optional, [Q]): a pruning floor per query -- only rows scoring STRICTLY above it are kept.  Any value
 * below the true k-th best score is valid (e.g. nextafter(k-th best of a row sample, -inf)) and leaves
 * the result unchanged while removing most list insertions.
 * sorted by (score desc, index asc); idx = idx_offset + local row.  A list with fewer than k rows ends at its first
 * entry with a NEGATIVE index (score -inf); the slots behind that terminator are unspecified.  Then skyemb_topk_merge. */
int skyemb_cosine_topk_chunks(int64_t N, int Q, int D, int k);
int skyemb_cosine_topk(const float *tw, const float *qn, const float *bank, const float *xn, int Q, int64_t N,
                       int D, int k, float eps, int64_t idx_offset, int nchunks, const float *thr0, float *part_s,
                       int64_t *part_i, void *stream);
/* merge `nlists` sorted length-k lists per query (bank chunks, or per-rank results after the
 * RCCL all-gather): in [Q, nlists, k] -> out [Q, k].  `ws` (optional, Q ints) enables the gather + block-sort
 * path used for the many short lists of the bank-streaming kernel (per-query fallback to the tournament). */
/* out[q] = the k-th largest value of x[q, 0..S) minus one ulp (NaN ranks as -inf): the pruning floor a search derives from
 * the exact scores of a bank sample (host glue of the build; the reference has no counterpart -- it sorts everything). */
int skyemb_kth_largest_floor(const float *x, int Q, int S, int k, float *out, void *stream);
/* The same floor for Q <= 16 queries in two launches, without the [Q, S] score matrix: exact scores of the S sample rows reduced
 * on the fly to the maximum of every 16-row tile, then the k-th largest of the S / 16 maxima per query, one ulp lower (the score
 * of at least k different rows: a valid floor, and within a few ranks of the sample's own k-th best when S / 16 >> k).
 *   sample [S, D] rows of the bank, sample_norms [S] their weighted norms; ws: Q * ceil(S / 16) floats; floor_out [Q].
 * Needs D % 64 == 0, D <= 1024, k <= S / 16 <= 2048.  (Build-level helper like skyemb_kth_largest_floor: the reference sorts
 * every score, utils/similarity.py:18-35.) */
int skyemb_cosine_sample_floor(const float *tw, const float *qn, const float *sample, const float *sample_norms, int Q, int64_t S,
                               int D, int k, float eps, float *ws, float *floor_out, void *stream);
/* 1 when skyemb_cosine_sample_floor accepts (Q, S, D, k) in THIS process -- the shape limits above and the streaming kernels not
 * switched off (SKYEMB_TOPK_STREAM=0) -- else 0: the caller then uses skyemb_cosine_scores + skyemb_kth_largest_floor.  Base
 * addresses of tw and sample must additionally be 16-byte aligned (checked by the call itself). */
int skyemb_cosine_sample_floor_applicable(int Q, int64_t S, int D, int k);
int skyemb_topk_merge(const float *in_s, const int64_t *in_i, int Q, int nlists, int k, float *out_s,
                      int64_t *out_i, void *ws, void *stream);

/* Many-query exact top-k in two stages (Q >= 17): an fp16 matrix-core pass over a half-precision image of the bank with
 * a proven error bound keeps, per query, only the rows whose score could still reach its top-k; the survivors are
 * re-scored with the contract's fp32 fma chain.  Results are bit-identical to skyemb_cosine_topk + skyemb_topk_merge.
 * Replaces the same reference code as skyemb_cosine_topk (utils/similarity.py:18-35,149-172).
 *   skyemb_bank16_prepare   once per bank / weights: bank16 = fp16 rows scaled by 2^-e_i, rowp = 4 floats per row
 *                           {xn 2^-e, ||x'||_2, 2^-e, xn}; BOTH hold skyemb_bank16_rowp_rows(N) rows (N rounded up to whole
 *                           tiles: the search kernel reads whole tiles), i.e. skyemb_bank16_bytes(N, D) bytes of bank16
 *   skyemb_cosine_topk_prefiltered   whole pipeline on `stream`; out_s / out_i [Q, k] final lists; redo[q] != 0 marks a query
 *                           whose answer could not be certified (too many near-ties, candidate overflow, fewer than k
 *                           finite rows): the caller runs those through skyemb_cosine_topk.  thr0 as in skyemb_cosine_topk
 *                           (NULL: the first bank slice supplies the floor).  ws: skyemb_topk_prefilter_ws_bytes(Q, D, k). */
int skyemb_topk_prefilter_applicable(int Q, int64_t N, int D, int k);
int64_t skyemb_topk_prefilter_ws_bytes(int Q, int D, int k);
int64_t skyemb_bank16_bytes(int64_t N, int D);
int64_t skyemb_bank16_rowp_rows(int64_t N);
int skyemb_bank16_prepare(const float *bank, const float *xn, int64_t N, int D, void *bank16, float *rowp, void *stream);
int skyemb_cosine_topk_prefiltered(const float *tw, const float *qn, int Q, const float *bank, const float *xn, const void *bank16,
                                   const float *rowp, int64_t N, int D, int k, float eps, int64_t idx_offset, const float *thr0,
                                   void *ws, int64_t ws_bytes, float *out_s, int64_t *out_i, int *redo, void *stream);
/* Attention pooling with one learned query (timm AttentionPoolLatent as built at utils/mim_vit.py:246-249 and applied at
 * :426-427; SimMIM models with attn_pool = True).  q [D] = Wq latent + bq is sample-independent (skyemb_attnpool_q);
 * kv [B, N, 2, H, hd] is the kv projection's output (compute dtype); fwd: out [B, D] (compute dtype) = softmax(scale q.k) v
 * per head, prob [B, H, N] fp32 saved for backward; bwd: dkv [B, N, 2, H, hd], dq_part [B, D] fp32 (per-sample dq);
 * skyemb_attnpool_q_bwd sums dq_part over the batch -> dWq = dq latent^T, dbq = dq, dlatent = Wq^T dq (ws: D floats). */
int skyemb_attnpool_q(const float *latent, const float *Wq, const float *bq, float *q, int D, void *stream);
int skyemb_attnpool_fwd(const float *q, const void *kv, int dtype, void *out, float *prob, int B, int N, int H, int hd, void *stream);
int skyemb_attnpool_bwd(const float *q, const void *kv, int dtype, const void *dout, const float *prob, void *dkv, float *dq_part,
                        int B, int N, int H, int hd, void *stream);
int skyemb_attnpool_q_bwd(const float *dq_part, int B, const float *latent, const float *Wq, float *dWq, float *dbq, float *dlatent,
                          float *ws, int D, void *stream);
/* plain score matrix for the reference-shaped path with P>1 patches per sample
 * (utils/similarity.py:262-267 combine over patches happens on these): scores [Q, N]. */
int skyemb_cosine_scores(const float *tw, const float *qn, const float *bank, const float *xn, int Q, int64_t N,
                         int D, float eps, float *scores, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SKYEMB_H */
