// Front-end / glue kernels of the MIM path: random masking from supplied noise, fused
// normalise + NaN-fill + patch gather (im2row of the kept patches only), decoder mask-token fill,
// row gathers, small deterministic reductions.  All HBM-bound, coalesced 16-byte accesses.
#include "common.h"
#include <type_traits>

namespace {

// ---- utils/mim_vit.py:354-379 ---------------------------------------------------------------
// one wave per sample: rank[i] = #{j : noise[j] < noise[i] or (== and j < i)}  (stable argsort)
__global__ __launch_bounds__(256) void mask_kernel(const float *__restrict__ noise, int B, int L, int keep,
                                                   int64_t *__restrict__ ids_restore, float *__restrict__ mask,
                                                   int32_t *__restrict__ ids_keep, int32_t *__restrict__ dec_dst,
                                                   int32_t *__restrict__ dec_tab, int E) {
    extern __shared__ float sn[];  // [4][L]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.x * 4 + wave;
    if (b >= B) return;
    float *row = sn + wave * L;
    for (int i = lane; i < L; i += 64) row[i] = noise[(int64_t)b * L + i];
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < L; i += 64) {
        const float v = row[i];
        int rank = 0;
        for (int j = 0; j < L; ++j) {
            const float u = row[j];
            rank += (u < v || (u == v && j < i)) ? 1 : 0;
        }
        ids_restore[(int64_t)b * L + i] = rank;
        mask[(int64_t)b * L + i] = rank >= keep ? 1.0f : 0.0f;
        if (rank < keep) {
            ids_keep[(int64_t)b * keep + rank] = i;
            if (dec_dst) {  // encoder token E+rank of sample b lands on decoder row E+i (mim_vit.py:448-450; E extra tokens)
                dec_dst[(int64_t)b * (keep + E) + E + rank] = b * (L + E) + E + i;
                dec_tab[(int64_t)b * (keep + E) + E + rank] = E + i;
            }
        }
    }
    if (dec_dst && lane < E) {   // cls (and RA/Dec) rows keep their place at the front
        dec_dst[(int64_t)b * (keep + E) + lane] = b * (L + E) + lane;
        dec_tab[(int64_t)b * (keep + E) + lane] = lane;
    }
}


// ---- utils/dataloaders.py:197-219 (MaskGenerator.__call__) on the device ------------------------------------------------------
// per sample: ratio = u * max_ratio, count = ceil(L * ratio); per channel an independent uniformly random subset of `count`
// patches is masked (the reference takes randperm(L)[:count]; here: the `count` smallest of L iid uniform noise values --
// the same distribution, ties broken by index), expanded to pixels.  One wave per (sample, channel); out is float 0 / 1.
__global__ __launch_bounds__(256) void simmim_mask_kernel(const float *__restrict__ noise, const float *__restrict__ ratio_u,
                                                          double max_ratio, int BC, int C, int L, int grid, int p,
                                                          float *__restrict__ out) {
    extern __shared__ float sn[];   // [4][L] noise, then [4][L] mask bytes as floats
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int bc = blockIdx.x * 4 + wave;
    if (bc >= BC) return;
    float *row = sn + wave * 2 * L, *mk = row + L;
    for (int i = lane; i < L; i += 64) row[i] = noise[(int64_t)bc * L + i];
    __builtin_amdgcn_wave_barrier();
    const int b = bc / C;
    // int(torch.ceil(torch.tensor(token_count * mask_ratio))): product in double, rounded to fp32, then ceil
    const int count = (int)ceilf((float)((double)L * ((double)ratio_u[b] * max_ratio)));
    for (int i = lane; i < L; i += 64) {
        const float v = row[i];
        int rank = 0;
        for (int j = 0; j < L; ++j) {
            const float u = row[j];
            rank += (u < v || (u == v && j < i)) ? 1 : 0;
        }
        mk[i] = rank < count ? 1.0f : 0.0f;
    }
    __builtin_amdgcn_wave_barrier();
    const int H = grid * p;
    float *o = out + (int64_t)bc * H * H;
    for (int e = lane * 4; e < H * H; e += 256) {
        const int y = e / H, x = e - y * H;                    // 4 consecutive pixels stay inside one patch (p % 4 == 0)
        const float m = mk[(y / p) * grid + x / p];
        *(float4 *)(o + e) = make_float4(m, m, m, m);
    }
}

// ---- utils/mim_vit.py:385-392 + im2row for Conv2d(k=s=p) --------------------------------------
// grid = B*keep rows; each thread moves 4 consecutive px of one (c, py)
// pmask (SimMIM, utils/mim_vit.py:394-399; NULL in MAE mode): x = x * (1 - mask) + pmv * mask after the NaN fill
template <typename T>
__global__ __launch_bounds__(256) void patch_gather_kernel(const float *__restrict__ imgs, const float *__restrict__ pmv,
                                                           const int32_t *__restrict__ ids_keep, T *__restrict__ out,
                                                           int C, int H, int W, int p, int keep, float mean, float stdv,
                                                           const float *__restrict__ pmask) {
    const int row = blockIdx.x;
    const int b = row / keep;
    const int l = ids_keep ? ids_keep[row] : (row - b * keep);
    const int gw = W / p;
    const int y0 = (l / gw) * p, x0 = (l % gw) * p;
    const int pv4 = C * p * p / 4, p4 = p / 4;
    for (int e = threadIdx.x; e < pv4; e += 256) {
        const int px4 = e % p4, py = (e / p4) % p, c = e / (p4 * p);
        const float4 x = *(const float4 *)(imgs + (((int64_t)b * C + c) * H + y0 + py) * W + x0 + 4 * px4);
        const float4 m = *(const float4 *)(pmv + (c * p + py) * p + 4 * px4);
        float4 o;
        // (x - mean) / std exactly as the reference; NaN -> learned fill value
        o.x = x.x != x.x ? m.x : (x.x - mean) / stdv;
        o.y = x.y != x.y ? m.y : (x.y - mean) / stdv;
        o.z = x.z != x.z ? m.z : (x.z - mean) / stdv;
        o.w = x.w != x.w ? m.w : (x.w - mean) / stdv;
        if (pmask) {
            const float4 k = *(const float4 *)(pmask + (((int64_t)b * C + c) * H + y0 + py) * W + x0 + 4 * px4);
            o.x = o.x * (1.0f - k.x) + m.x * k.x;
            o.y = o.y * (1.0f - k.y) + m.y * k.y;
            o.z = o.z * (1.0f - k.z) + m.z * k.z;
            o.w = o.w * (1.0f - k.w) + m.w * k.w;
        }
        if constexpr (std::is_same<T, f16_t>::value) {
            // the one place an INPUT of any size enters the fp16 path: a saturated star of > 65504 normalised units stays the
            // largest finite half instead of becoming inf (NaNs were replaced above; fminf / fmaxf would pass the other operand)
            o.x = fminf(fmaxf(o.x, -65504.f), 65504.f);
            o.y = fminf(fmaxf(o.y, -65504.f), 65504.f);
            o.z = fminf(fmaxf(o.z, -65504.f), 65504.f);
            o.w = fminf(fmaxf(o.w, -65504.f), 65504.f);
        }
        store4<T>(out + (int64_t)row * (C * p * p) + 4 * e, o.x, o.y, o.z, o.w);
    }
}

// partial[b][e] = sum_j w * drows[b*keep+j][e],  w = d(embedded pixel)/d(pmv) = isnan(pixel) ? 1 : pixel_mask (0 in MAE mode)
__global__ __launch_bounds__(256) void pmv_partial_kernel(const float *__restrict__ imgs, const int32_t *__restrict__ ids_keep,
                                                          const float *__restrict__ drows, float *__restrict__ partial,
                                                          int C, int H, int W, int p, int keep, const float *__restrict__ pmask) {
    const int b = blockIdx.x;
    const int gw = W / p, pv = C * p * p;
    const int e = blockIdx.y * 256 + threadIdx.x;         // grid (B, ceil(pv / 256)): one patch element per thread
    if (e >= pv) return;
    const int px = e % p, py = (e / p) % p, c = e / (p * p);
    float acc = 0.f;
    for (int j0 = 0; j0 < keep; j0 += 4) {                // four patches' loads in flight (the sum stays in patch order)
        float wgt[4], d[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = j0 + u < keep ? j0 + u : keep - 1;
            const int l = ids_keep ? ids_keep[b * keep + j] : j;
            const int y = (l / gw) * p + py, x = (l % gw) * p + px;
            const int64_t pix = (((int64_t)b * C + c) * H + y) * W + x;
            const float v = imgs[pix];
            wgt[u] = j0 + u >= keep ? 0.0f : v != v ? 1.0f : (pmask ? pmask[pix] : 0.0f);
            d[u] = drows[((int64_t)b * keep + j) * pv + e];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (wgt[u] != 0.0f) acc += wgt[u] * d[u];
    }
    partial[(int64_t)b * pv + e] = acc;
}

// ---- column sums: out[n] = sum_m X[m][n]; block = 64 columns x 4 row groups -------------------
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T *__restrict__ X, int64_t ldx, int M, int N,
                                                     float *__restrict__ out) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + lane;
    float acc = 0.f;
    if (n < N) {
#pragma unroll 8
        for (int m = wave; m < M; m += 4) acc += to_f32<T>(X[(int64_t)m * ldx + n]);   // (8 loads in flight; same summation order)
    }
    red[wave][lane] = acc;
    __syncthreads();
    if (wave == 0 && n < N) out[n] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
}

// ---- utils/mim_vit.py:446-453: rows holding a mask token ----------------------------------------
__global__ __launch_bounds__(256) void fill_mask_tokens_kernel(float *__restrict__ x, const float *__restrict__ mask,
                                                               const float *__restrict__ mask_token,
                                                               const float *__restrict__ dec_pos, int L, int Dd, int E) {
    const int b = blockIdx.x / L, l = blockIdx.x % L;
    if (mask[(int64_t)b * L + l] == 0.0f) return;
    float *dst = x + ((int64_t)b * (L + E) + E + l) * Dd;
    const float *pos = dec_pos + (int64_t)(E + l) * Dd;
    for (int d = threadIdx.x * 4; d < Dd; d += 1024) {
        const float4 t = *(const float4 *)(mask_token + d), q = *(const float4 *)(pos + d);
        *(float4 *)(dst + d) = make_float4(t.x + q.x, t.y + q.y, t.z + q.z, t.w + q.w);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void gather_rows_kernel(const float *__restrict__ src, const int32_t *__restrict__ idx,
                                                          float *__restrict__ out, T *__restrict__ out_lp, int D) {
    const int r = blockIdx.x;
    const float *s = src + (int64_t)idx[r] * D;
    for (int d = threadIdx.x * 4; d < D; d += 1024) {
        const float4 v = *(const float4 *)(s + d);
        if (out) *(float4 *)(out + (int64_t)r * D + d) = v;
        if (out_lp) store4<T>(out_lp + (int64_t)r * D + d, v.x, v.y, v.z, v.w);
    }
}

// partial[blk][d] = sum over this block's selected rows
__global__ __launch_bounds__(256) void rowsum_select_kernel(const float *__restrict__ src, int64_t ld,
                                                            const float *__restrict__ sel, int row0, int inner,
                                                            int outer_stride, int n_rows, int D,
                                                            float *__restrict__ partial) {
    const int nblk = gridDim.x;
    for (int d = threadIdx.x; d < D; d += 256) {
        float acc = 0.f;
        // four rows in flight per thread (the one-row-at-a-time loop was a chain of dependent ~1 us loads: 27 us per call);
        // the additions keep the row order, so the sums are the same bits
        for (int i0 = blockIdx.x; i0 < n_rows; i0 += 4 * nblk) {
            float x[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * nblk;
                const bool take = i < n_rows && !(sel && sel[i] == 0.0f);
                const int ii = take ? i : 0;
                const int64_t r = (int64_t)row0 + (int64_t)(ii / inner) * outer_stride + (ii % inner);
                x[u] = take ? src[r * ld + d] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (i0 + u * nblk < n_rows && !(sel && sel[i0 + u * nblk] == 0.0f)) acc += x[u];
        }
        partial[(int64_t)blockIdx.x * D + d] = acc;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void cast_kernel(const float *__restrict__ src, T *__restrict__ dst, int64_t n4) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const float4 v = *(const float4 *)(src + 4 * i);
        store4<T>(dst + 4 * i, v.x, v.y, v.z, v.w);
    }
}

// feeder tail (utils/dataloaders.py:293-300): clip at pixel_min / pixel_max (NaN compares false: kept) and centre-crop a
// staged minibatch [n_planes, Hs, Ws] -> [n_planes, size, size]; one thread per output pixel quad when size % 4 == 0
__global__ __launch_bounds__(256) void clip_crop_kernel(const float *__restrict__ src, float *__restrict__ dst, int64_t n_out,
                                                        int Hs, int Ws, int size, int top, int left, float lo, float hi,
                                                        int use_lo, int use_hi) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_out; i += (int64_t)gridDim.x * 256) {
        const int x = (int)(i % size);
        const int64_t t = i / size;
        const int y = (int)(t % size);
        const int64_t plane = t / size;
        float v = src[(plane * Hs + top + y) * Ws + left + x];
        if (use_lo && v < lo) v = lo;
        if (use_hi && v > hi) v = hi;
        dst[i] = v;
    }
}

// Survey-tile sampler (utils/dataloaders.py:449-476 random_cutouts / :507-536 overlapping_cutouts + the clip at :618-621):
// the whole multi-band tile [C, H, W] is resident in HBM as the 4-byte words of the FITS files (big-endian floats where
// be[c] != 0: the bytes went from the mapped file to the device untouched); cutout i = tile[:, h0[i] : +S, w0[i] : +S],
// decoded, clipped (NaN kept: both comparisons are false) -> out [n, C, S, S].
__global__ __launch_bounds__(256) void tile_cutouts_kernel(const unsigned int *__restrict__ tile, const int *__restrict__ be, int C,
                                                           int H, int W, const int *__restrict__ h0, const int *__restrict__ w0,
                                                           int64_t n_out, int S, float lo, float hi, int use_lo, int use_hi,
                                                           float *__restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_out; i += (int64_t)gridDim.x * 256) {
        const int x = (int)(i % S);
        int64_t t = i / S;
        const int y = (int)(t % S);
        t /= S;
        const int c = (int)(t % C);
        const int64_t k = t / C;
        unsigned int u = tile[((int64_t)c * H + h0[k] + y) * W + w0[k] + x];
        if (be[c]) u = __builtin_bswap32(u);
        float v = __uint_as_float(u);
        if (use_lo && v < lo) v = lo;
        if (use_hi && v > hi) v = hi;
        out[i] = v;
    }
}

}  // namespace

extern "C" int skyemb_clip_crop(const float *src, float *dst, int64_t n_planes, int Hs, int Ws, int size, float lo, float hi,
                                int use_lo, int use_hi, void *stream) {
    SKY_CHECK_ARG(src && dst && n_planes > 0 && size > 0 && Hs >= size && Ws >= size, "skyemb_clip_crop: bad shape");
    const int64_t n_out = n_planes * size * size;
    int64_t blocks = ceil_div64(n_out, 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(clip_crop_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, dst, n_out, Hs, Ws, size,
                       (Hs - size) / 2, (Ws - size) / 2, lo, hi, use_lo, use_hi);
    SKY_LAUNCH_CHECK("skyemb_clip_crop");
    return 0;
}

extern "C" int skyemb_tile_cutouts(const void *tile, const int *big_endian, int C, int H, int W, const int *h0, const int *w0, int n,
                                   int S, float lo, float hi, int use_lo, int use_hi, float *out, void *stream) {
    SKY_CHECK_ARG(tile && big_endian && h0 && w0 && out && C > 0 && n > 0 && S > 0 && H >= S && W >= S,
                  "skyemb_tile_cutouts: bad shape C=%d H=%d W=%d n=%d S=%d", C, H, W, n, S);
    const int64_t n_out = (int64_t)n * C * S * S;
    int64_t blocks = ceil_div64(n_out, 256);
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(tile_cutouts_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const unsigned int *)tile,
                       big_endian, C, H, W, h0, w0, n_out, S, lo, hi, use_lo, use_hi, out);
    SKY_LAUNCH_CHECK("skyemb_tile_cutouts");
    return 0;
}

extern "C" int skyemb_random_mask_from_noise(const float *noise, int B, int L, int keep, int64_t *ids_restore,
                                             float *mask, int32_t *ids_keep, int32_t *dec_dst, int32_t *dec_tab,
                                             int n_extra, void *stream) {
    SKY_CHECK_ARG(B > 0 && L > 0 && keep >= 0 && keep <= L && L <= 4096 && n_extra >= 1 && n_extra <= 2,
                  "skyemb_random_mask_from_noise: bad shape B=%d L=%d keep=%d extra=%d", B, L, keep, n_extra);
    hipLaunchKernelGGL(mask_kernel, dim3((B + 3) / 4), dim3(256), (size_t)4 * L * sizeof(float), (hipStream_t)stream, noise,
                       B, L, keep, ids_restore, mask, ids_keep, dec_dst, dec_tab, n_extra);
    SKY_LAUNCH_CHECK("skyemb_random_mask_from_noise");
    return 0;
}


extern "C" int skyemb_simmim_mask_from_noise(const float *noise, const float *ratio_u, double max_ratio, int B, int C, int L, int grid,
                                             int p, float *out_mask, void *stream) {
    SKY_CHECK_ARG(noise && ratio_u && out_mask && B > 0 && C > 0 && L == grid * grid && L <= 4096 && p > 0 && p % 4 == 0 &&
                  max_ratio >= 0.0 && max_ratio <= 1.0, "skyemb_simmim_mask_from_noise: bad arguments (L=%d grid=%d p=%d)", L, grid, p);
    hipLaunchKernelGGL(simmim_mask_kernel, dim3((B * C + 3) / 4), dim3(256), (size_t)8 * L * sizeof(float), (hipStream_t)stream, noise,
                       ratio_u, max_ratio, B * C, C, L, grid, p, out_mask);
    SKY_LAUNCH_CHECK("skyemb_simmim_mask_from_noise");
    return 0;
}

static int patch_gather_launch(const float *imgs, const float *pmv, const int32_t *ids_keep, const float *pixel_mask, void *out,
                               int dtype, int B, int C, int H, int W, int p, int keep, float pixel_mean, float pixel_std,
                               void *stream) {
    SKY_CHECK_ARG(B > 0 && C > 0 && p > 0 && p % 4 == 0 && H % p == 0 && W % p == 0 && keep > 0,
                  "skyemb_patch_gather: bad geometry C=%d H=%d W=%d p=%d keep=%d", C, H, W, p, keep);
    SKY_CHECK_ARG(ids_keep || keep == (H / p) * (W / p), "skyemb_patch_gather: ids_keep == NULL needs keep == L");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == SKYEMB_BF16)
        hipLaunchKernelGGL(patch_gather_kernel<bf16_t>, dim3(B * keep), dim3(256), 0, st, imgs, pmv, ids_keep, (bf16_t *)out,
                           C, H, W, p, keep, pixel_mean, pixel_std, pixel_mask);
    else if (dtype == SKYEMB_F16)
        hipLaunchKernelGGL(patch_gather_kernel<f16_t>, dim3(B * keep), dim3(256), 0, st, imgs, pmv, ids_keep, (f16_t *)out,
                           C, H, W, p, keep, pixel_mean, pixel_std, pixel_mask);
    else
        hipLaunchKernelGGL(patch_gather_kernel<float>, dim3(B * keep), dim3(256), 0, st, imgs, pmv, ids_keep, (float *)out, C,
                           H, W, p, keep, pixel_mean, pixel_std, pixel_mask);
    SKY_LAUNCH_CHECK("skyemb_patch_gather");
    return 0;
}

extern "C" int skyemb_patch_gather(const float *imgs, const float *pmv, const int32_t *ids_keep, void *out, int dtype,
                                   int B, int C, int H, int W, int p, int keep, float pixel_mean, float pixel_std,
                                   void *stream) {
    return patch_gather_launch(imgs, pmv, ids_keep, nullptr, out, dtype, B, C, H, W, p, keep, pixel_mean, pixel_std, stream);
}

extern "C" int skyemb_patch_gather_blend(const float *imgs, const float *pmv, const int32_t *ids_keep, const float *pixel_mask,
                                         void *out, int dtype, int B, int C, int H, int W, int p, int keep, float pixel_mean,
                                         float pixel_std, void *stream) {
    return patch_gather_launch(imgs, pmv, ids_keep, pixel_mask, out, dtype, B, C, H, W, p, keep, pixel_mean, pixel_std, stream);
}

extern "C" int skyemb_colsum(const void *X, int dtype, int64_t ldx, int M, int N, float *out, void *stream) {
    SKY_CHECK_ARG(M > 0 && N > 0, "skyemb_colsum: bad shape");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == SKYEMB_BF16)
        hipLaunchKernelGGL(colsum_kernel<bf16_t>, dim3((N + 63) / 64), dim3(256), 0, st, (const bf16_t *)X, ldx, M, N, out);
    else if (dtype == SKYEMB_F16)
        hipLaunchKernelGGL(colsum_kernel<f16_t>, dim3((N + 63) / 64), dim3(256), 0, st, (const f16_t *)X, ldx, M, N, out);
    else
        hipLaunchKernelGGL(colsum_kernel<float>, dim3((N + 63) / 64), dim3(256), 0, st, (const float *)X, ldx, M, N, out);
    SKY_LAUNCH_CHECK("skyemb_colsum");
    return 0;
}

extern "C" int skyemb_patch_gather_bwd_pmv_blend(const float *imgs, const int32_t *ids_keep, const float *pixel_mask,
                                                 const float *drows, float *partial, float *dpmv, int B, int C, int H, int W,
                                                 int p, int keep, void *stream) {
    SKY_CHECK_ARG(B > 0 && C > 0 && p > 0 && keep > 0, "skyemb_patch_gather_bwd_pmv: bad geometry");
    hipLaunchKernelGGL(pmv_partial_kernel, dim3(B, (C * p * p + 255) / 256), dim3(256), 0, (hipStream_t)stream, imgs, ids_keep, drows, partial, C, H,
                       W, p, keep, pixel_mask);
    SKY_LAUNCH_CHECK("skyemb_patch_gather_bwd_pmv");
    return skyemb_colsum(partial, SKYEMB_F32, (int64_t)C * p * p, B, C * p * p, dpmv, stream);
}

extern "C" int skyemb_patch_gather_bwd_pmv(const float *imgs, const int32_t *ids_keep, const float *drows, float *partial,
                                           float *dpmv, int B, int C, int H, int W, int p, int keep, void *stream) {
    return skyemb_patch_gather_bwd_pmv_blend(imgs, ids_keep, nullptr, drows, partial, dpmv, B, C, H, W, p, keep, stream);
}

extern "C" int skyemb_fill_mask_tokens(float *x, const float *mask, const float *mask_token, const float *dec_pos, int B,
                                       int L, int Dd, int n_extra, void *stream) {
    SKY_CHECK_ARG(B > 0 && L > 0 && Dd % 4 == 0 && n_extra >= 1, "skyemb_fill_mask_tokens: bad shape");
    hipLaunchKernelGGL(fill_mask_tokens_kernel, dim3(B * L), dim3(256), 0, (hipStream_t)stream, x, mask, mask_token, dec_pos,
                       L, Dd, n_extra);
    SKY_LAUNCH_CHECK("skyemb_fill_mask_tokens");
    return 0;
}

extern "C" int skyemb_gather_rows(const float *src, const int32_t *idx, float *out, void *out_lp, int dtype, int n_rows,
                                  int D, void *stream) {
    SKY_CHECK_ARG(n_rows > 0 && D % 4 == 0, "skyemb_gather_rows: bad shape");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == SKYEMB_BF16)
        hipLaunchKernelGGL(gather_rows_kernel<bf16_t>, dim3(n_rows), dim3(256), 0, st, src, idx, out, (bf16_t *)out_lp, D);
    else if (dtype == SKYEMB_F16)
        hipLaunchKernelGGL(gather_rows_kernel<f16_t>, dim3(n_rows), dim3(256), 0, st, src, idx, out, (f16_t *)out_lp, D);
    else
        hipLaunchKernelGGL(gather_rows_kernel<float>, dim3(n_rows), dim3(256), 0, st, src, idx, out, (float *)out_lp, D);
    SKY_LAUNCH_CHECK("skyemb_gather_rows");
    return 0;
}

extern "C" int skyemb_rowsum_select(const float *src, int64_t ld, const float *sel, int row0, int inner, int outer_stride,
                                    int n_rows, int D, float *partial, float *out, void *stream) {
    SKY_CHECK_ARG(n_rows > 0 && D > 0 && inner > 0, "skyemb_rowsum_select: bad shape");
    const int nblk = 256;      // one workgroup per CU: 16 of the 4096 decoder rows each (64 workgroups walked 64 rows: 23 us per call)
    hipLaunchKernelGGL(rowsum_select_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, src, ld, sel, row0, inner,
                       outer_stride, n_rows, D, partial);
    SKY_LAUNCH_CHECK("skyemb_rowsum_select");
    return skyemb_colsum(partial, SKYEMB_F32, D, nblk, D, out, stream);
}

extern "C" int skyemb_cast(const float *src, void *dst, int dtype, int64_t n, void *stream) {
    SKY_CHECK_ARG(n > 0 && n % 4 == 0, "skyemb_cast: n must be a positive multiple of 4");
    int64_t blocks = ceil_div64(n / 4, 256);
    if (blocks > 4096) blocks = 4096;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == SKYEMB_BF16)
        hipLaunchKernelGGL(cast_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, st, src, (bf16_t *)dst, n / 4);
    else if (dtype == SKYEMB_F16)
        hipLaunchKernelGGL(cast_kernel<f16_t>, dim3((unsigned)blocks), dim3(256), 0, st, src, (f16_t *)dst, n / 4);
    else
        hipLaunchKernelGGL(cast_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, st, src, (float *)dst, n / 4);
    SKY_LAUNCH_CHECK("skyemb_cast");
    return 0;
}
