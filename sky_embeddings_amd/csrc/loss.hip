// Fused masked per-patch reconstruction loss (utils/mim_vit.py:326-338,473-521,614-627):
// patchify + NaN-aware per-patch mean / biased variance normalisation + MSE or L1 over the masked,
// non-NaN elements; forward gives the scalar loss, backward gives d loss / d pred.
//
//   pass 1 (one block per patch): target stats + per-patch (sum, count)      -> ws[(b*L+l)*4 + {0,1,2,3}]
//   finalize (one block): loss = S / (cnt/numel*numel + 1e-5), inv_den        -> ws[4*B*L + {0,1}]
//   pass 2 (one block per decoder row): dpred = mask * valid * dl/ddiff * inv_den
#include "common.h"

namespace {

__device__ __forceinline__ float block_sum(float v, float *red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// target element e = (py*p + px)*C + c of patch l  (patchify order 'nhwpqc')
__device__ __forceinline__ float target_elem(const float *__restrict__ imgs, int b, int l, int e, int C, int H, int W,
                                             int p, float mean, float stdv) {
    const int c = e % C, px = (e / C) % p, py = e / (C * p);
    const int gw = W / p;
    const int y = (l / gw) * p + py, x = (l % gw) * p + px;
    return (imgs[(((int64_t)b * C + c) * H + y) * W + x] - mean) / stdv;
}

__global__ __launch_bounds__(256) void loss_pass1(const float *__restrict__ imgs, const float *__restrict__ pred,
                                                  const float *__restrict__ mask, float *__restrict__ ws, int C, int H,
                                                  int W, int p, int L, int extra, float mean, float stdv, int norm_pix,
                                                  int loss_l1) {
    __shared__ float red[4];
    const int b = blockIdx.x / L, l = blockIdx.x % L;
    const int pv = C * p * p;
    float *o = ws + (int64_t)blockIdx.x * 4;
    if (mask[blockIdx.x] == 0.0f) {  // unmasked patches contribute nothing (mim_vit.py:518-519)
        if (threadIdx.x == 0) { o[0] = 0.f; o[1] = 0.f; o[2] = 0.f; o[3] = 0.f; }
        return;
    }
    // the patch's target elements are read ONCE into registers (<= 12 per thread: patches up to 3072 values; larger ones
    // re-read them per pass); same element -> thread mapping and summation order as the three-pass form, so the same bits
    constexpr int MAXE = 12;
    const bool cached = pv <= MAXE * 256;
    float tv[MAXE];
    if (cached) {
#pragma unroll
        for (int u = 0; u < MAXE; ++u) {
            const int e = threadIdx.x + u * 256;
            tv[u] = e < pv ? target_elem(imgs, b, l, e, C, H, W, p, mean, stdv) : 0.f;
        }
    }
    float mu = 0.f, istd = 1.f;
    if (norm_pix) {
        float s = 0.f, n = 0.f;
        if (cached) {
#pragma unroll
            for (int u = 0; u < MAXE; ++u)
                if (threadIdx.x + u * 256 < pv) { const float t = tv[u]; if (t == t) { s += t; n += 1.f; } }
        } else {
            for (int e = threadIdx.x; e < pv; e += 256) {
                const float t = target_elem(imgs, b, l, e, C, H, W, p, mean, stdv);
                if (t == t) { s += t; n += 1.f; }
            }
        }
        s = block_sum(s, red);
        n = block_sum(n, red);
        mu = s / n;
        float q = 0.f;
        if (cached) {
#pragma unroll
            for (int u = 0; u < MAXE; ++u)
                if (threadIdx.x + u * 256 < pv) { const float t = tv[u]; if (t == t) { const float d = t - mu; q += d * d; } }
        } else {
            for (int e = threadIdx.x; e < pv; e += 256) {
                const float t = target_elem(imgs, b, l, e, C, H, W, p, mean, stdv);
                if (t == t) { const float d = t - mu; q += d * d; }
            }
        }
        q = block_sum(q, red);
        istd = 1.0f / sqrtf(q / n + 1.0e-6f);
    }
    const float *pr = pred + ((int64_t)b * (L + extra) + extra + l) * pv;
    float s = 0.f, n = 0.f;
    if (cached) {
#pragma unroll
        for (int u = 0; u < MAXE; ++u) {
            const int e = threadIdx.x + u * 256;
            if (e < pv) {
                float t = tv[u];
                if (norm_pix) t = (t - mu) * istd;
                const float d = t - pr[e];
                if (d == d) { s += loss_l1 ? fabsf(d) : d * d; n += 1.f; }
            }
        }
    } else {
        for (int e = threadIdx.x; e < pv; e += 256) {
            float t = target_elem(imgs, b, l, e, C, H, W, p, mean, stdv);
            if (norm_pix) t = (t - mu) * istd;
            const float d = t - pr[e];
            if (d == d) { s += loss_l1 ? fabsf(d) : d * d; n += 1.f; }
        }
    }
    s = block_sum(s, red);
    n = block_sum(n, red);
    if (threadIdx.x == 0) { o[0] = s; o[1] = n; o[2] = mu; o[3] = istd; }
}

__global__ __launch_bounds__(256) void loss_finalize(float *__restrict__ ws, float *__restrict__ loss, int BL, float numel, float dscale) {
    __shared__ float red[4];
    float s = 0.f, n = 0.f;
    for (int i = threadIdx.x; i < BL; i += 256) { s += ws[4 * (int64_t)i]; n += ws[4 * (int64_t)i + 1]; }
    s = block_sum(s, red);
    n = block_sum(n, red);
    if (threadIdx.x == 0) {
        // avg_scale_factor = mask.sum() / mask.numel() * loss.numel()   (mim_vit.py:518)
        const float scale = n / numel * numel;
        const float inv = 1.0f / (scale + 1e-5f);
        loss[0] = s / (scale + 1e-5f);
        ws[4 * (int64_t)BL] = loss[0];
        ws[4 * (int64_t)BL + 1] = inv * dscale;   // dscale: the caller's static loss scale (a power of two; 1 = none)
    }
}

template <typename T>
__global__ __launch_bounds__(256) void loss_pass2(const float *__restrict__ imgs, const float *__restrict__ pred,
                                                  const float *__restrict__ mask, const float *__restrict__ ws,
                                                  T *__restrict__ dpred, float *__restrict__ dpred32, int C, int H, int W,
                                                  int p, int L, int extra, float mean, float stdv, int norm_pix,
                                                  int loss_l1, int BL) {
    const int Nd = L + extra;
    const int b = blockIdx.x / Nd, r = blockIdx.x % Nd;
    const int pv = C * p * p;
    const int64_t off = (int64_t)blockIdx.x * pv;
    const int l = r - extra;
    const bool live = l >= 0 && mask[(int64_t)b * L + l] != 0.0f;
    if (!live) {
        for (int e = threadIdx.x; e < pv; e += 256) {
            if (dpred) dpred[off + e] = from_f32<T>(0.f);
            if (dpred32) dpred32[off + e] = 0.f;
        }
        return;
    }
    const float inv = ws[4 * (int64_t)BL + 1];
    const float mu = ws[((int64_t)b * L + l) * 4 + 2], istd = ws[((int64_t)b * L + l) * 4 + 3];
    for (int e = threadIdx.x; e < pv; e += 256) {
        float t = target_elem(imgs, b, l, e, C, H, W, p, mean, stdv);
        if (norm_pix) t = (t - mu) * istd;
        const float d = pred[off + e] - t;  // d loss_e / d pred = 2 d (mse) | sign(d) (l1)
        float g = 0.f;
        if (d == d) g = (loss_l1 ? (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) : 2.f * d) * inv;
        if (dpred) dpred[off + e] = from_f32<T>(g);
        if (dpred32) dpred32[off + e] = g;
    }
}


// ---- SimMIM mode (utils/mim_vit.py:469, 480-493, 497-521) ---------------------------------------
// pred comes straight from the head GEMM as token rows: row b*(L+extra)+extra+l, column c*p*p + py*p + px
// (= Conv1x1 output channel c*up^2 + i*up + j of PixelShuffle(up = p), utils/mim_vit.py:254-261); the pixel-shuffled
// image pred_img [B,C,H,W] is only materialised when the caller asks for it.  Weights: w = pixel_mask where the
// target is not NaN, else 0;  loss = sum(w * l) / (sum(w) + 1e-5).

// element e' = (c*p + py)*p + px of patch l -> pixel index in [B,C,H,W]
__device__ __forceinline__ int64_t simmim_pixel(int b, int l, int e, int C, int H, int W, int p) {
    const int px = e % p, py = (e / p) % p, c = e / (p * p);
    const int gw = W / p;
    return (((int64_t)b * C + c) * H + (l / gw) * p + py) * W + (l % gw) * p + px;
}

__global__ __launch_bounds__(256) void simmim_pass1(const float *__restrict__ imgs, const float *__restrict__ pred,
                                                    const float *__restrict__ pmask, float *__restrict__ ws, int C, int H,
                                                    int W, int p, int L, int extra, float mean, float stdv, int norm_pix,
                                                    int loss_l1, int pooled) {
    __shared__ float red[4];
    const int b = blockIdx.x / L, l = blockIdx.x % L;
    const int pv = C * p * p;
    float *o = ws + (int64_t)blockIdx.x * 4;
    float mu = 0.f, istd = 1.f;
    if (norm_pix) {   // NaN-aware mean and biased variance over the whole patch vector (patch_mean_and_var)
        float s = 0.f, n = 0.f;
        for (int e = threadIdx.x; e < pv; e += 256) {
            const float t = (imgs[simmim_pixel(b, l, e, C, H, W, p)] - mean) / stdv;
            if (t == t) { s += t; n += 1.f; }
        }
        s = block_sum(s, red);
        n = block_sum(n, red);
        mu = s / n;
        float q = 0.f;
        for (int e = threadIdx.x; e < pv; e += 256) {
            const float t = (imgs[simmim_pixel(b, l, e, C, H, W, p)] - mean) / stdv;
            if (t == t) { const float d = t - mu; q += d * d; }
        }
        q = block_sum(q, red);
        istd = 1.0f / sqrtf(q / n + 1.0e-6f);
    }
    // prediction of pixel `pix`: element e of the patch's token row, or -- behind an attention pool, whose one row per image
    // is up-sampled by PixelShuffle(img_size) -- the row laid out like the image itself
    const float *pr = pred + ((int64_t)b * (L + extra) + extra + l) * pv;
    float s = 0.f, n = 0.f;
    for (int e = threadIdx.x; e < pv; e += 256) {
        const int64_t pix = simmim_pixel(b, l, e, C, H, W, p);
        float t = (imgs[pix] - mean) / stdv;
        if (norm_pix) t = (t - mu) * istd;
        const float d = t - (pooled ? pred[pix] : pr[e]);
        const float w = pmask[pix];
        if (d == d) { s += w * (loss_l1 ? fabsf(d) : d * d); n += w; }
    }
    s = block_sum(s, red);
    n = block_sum(n, red);
    if (threadIdx.x == 0) { o[0] = s; o[1] = n; o[2] = mu; o[3] = istd; }
}

template <typename T>
__global__ __launch_bounds__(256) void simmim_pass2(const float *__restrict__ imgs, const float *__restrict__ pred,
                                                    const float *__restrict__ pmask, const float *__restrict__ ws,
                                                    T *__restrict__ dpred, float *__restrict__ pred_img, int C, int H, int W,
                                                    int p, int L, int extra, float mean, float stdv, int norm_pix,
                                                    int loss_l1, int BL, int pooled) {
    const int Nd = L + extra;
    const int b = blockIdx.x / Nd, r = blockIdx.x % Nd;
    const int pv = C * p * p;
    const int64_t off = (int64_t)blockIdx.x * pv;
    const int l = r - extra;
    if (l < 0) {   // cls / RA-Dec rows take no part in the reconstruction
        if (dpred)
            for (int e = threadIdx.x; e < pv; e += 256) dpred[off + e] = from_f32<T>(0.f);
        return;
    }
    const float inv = ws[4 * (int64_t)BL + 1];
    const float mu = ws[((int64_t)b * L + l) * 4 + 2], istd = ws[((int64_t)b * L + l) * 4 + 3];
    for (int e = threadIdx.x; e < pv; e += 256) {
        const int64_t pix = simmim_pixel(b, l, e, C, H, W, p);
        const int64_t at = pooled ? pix : off + e;
        const float pe = pred[at];
        if (pred_img) pred_img[pix] = pe;
        if (!dpred) continue;
        float t = (imgs[pix] - mean) / stdv;
        if (norm_pix) t = (t - mu) * istd;
        const float d = pe - t;
        float g = 0.f;
        if (d == d) g = pmask[pix] * (loss_l1 ? (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) : 2.f * d) * inv;
        dpred[at] = from_f32<T>(g);
    }
}

}  // namespace

extern "C" int skyemb_masked_patch_loss(const float *imgs, const float *pred, const float *mask, float *loss, void *dpred,
                                        float *dpred32, int dtype, float *ws, int B, int C, int H, int W, int p,
                                        int extra, float pixel_mean, float pixel_std, int norm_pix, int loss_l1,
                                        float dscale, void *stream) {
    SKY_CHECK_ARG(B > 0 && C > 0 && p > 0 && H % p == 0 && W % p == 0 && extra >= 0, "skyemb_masked_patch_loss: bad geometry");
    SKY_CHECK_ARG(dscale > 0.f, "skyemb_masked_patch_loss: dscale must be positive (1 = no loss scale)");
    hipStream_t st = (hipStream_t)stream;
    const int L = (H / p) * (W / p), BL = B * L;
    const float numel = (float)((double)BL * C * p * p);
    hipLaunchKernelGGL(loss_pass1, dim3(BL), dim3(256), 0, st, imgs, pred, mask, ws, C, H, W, p, L, extra, pixel_mean,
                       pixel_std, norm_pix, loss_l1);
    hipLaunchKernelGGL(loss_finalize, dim3(1), dim3(256), 0, st, ws, loss, BL, numel, dscale);
    if (dpred || dpred32) {
        if (dtype == SKYEMB_BF16)
            hipLaunchKernelGGL(loss_pass2<bf16_t>, dim3(B * (L + extra)), dim3(256), 0, st, imgs, pred, mask, ws, (bf16_t *)dpred,
                               dpred32, C, H, W, p, L, extra, pixel_mean, pixel_std, norm_pix, loss_l1, BL);
        else if (dtype == SKYEMB_F16)
            hipLaunchKernelGGL(loss_pass2<f16_t>, dim3(B * (L + extra)), dim3(256), 0, st, imgs, pred, mask, ws, (f16_t *)dpred,
                               dpred32, C, H, W, p, L, extra, pixel_mean, pixel_std, norm_pix, loss_l1, BL);
        else
            hipLaunchKernelGGL(loss_pass2<float>, dim3(B * (L + extra)), dim3(256), 0, st, imgs, pred, mask, ws, (float *)dpred,
                               dpred32, C, H, W, p, L, extra, pixel_mean, pixel_std, norm_pix, loss_l1, BL);
    }
    SKY_LAUNCH_CHECK("skyemb_masked_patch_loss");
    return 0;
}

extern "C" int skyemb_simmim_pixel_loss(const float *imgs, const float *pred_tok, const float *pixel_mask, float *loss,
                                        void *dpred_tok, int dtype, float *pred_img, float *ws, int B, int C, int H, int W,
                                        int p, int extra, float pixel_mean, float pixel_std, int norm_pix, int loss_l1,
                                        int pooled, float dscale, void *stream) {
    SKY_CHECK_ARG(B > 0 && C > 0 && p > 0 && H % p == 0 && W % p == 0 && extra >= 0 && pixel_mask, "skyemb_simmim_pixel_loss: bad arguments");
    SKY_CHECK_ARG(dscale > 0.f, "skyemb_simmim_pixel_loss: dscale must be positive (1 = no loss scale)");
    SKY_CHECK_ARG(!pooled || extra == 0, "skyemb_simmim_pixel_loss: a pooled prediction has one row per image (extra = 0)");
    hipStream_t st = (hipStream_t)stream;
    const int L = (H / p) * (W / p), BL = B * L;
    hipLaunchKernelGGL(simmim_pass1, dim3(BL), dim3(256), 0, st, imgs, pred_tok, pixel_mask, ws, C, H, W, p, L, extra, pixel_mean,
                       pixel_std, norm_pix, loss_l1, pooled);
    hipLaunchKernelGGL(loss_finalize, dim3(1), dim3(256), 0, st, ws, loss, BL, 1.0f, dscale);   // scale = sum(w) (n/numel*numel)
    if (dpred_tok || pred_img) {
        if (dtype == SKYEMB_BF16)
            hipLaunchKernelGGL(simmim_pass2<bf16_t>, dim3(B * (L + extra)), dim3(256), 0, st, imgs, pred_tok, pixel_mask, ws,
                               (bf16_t *)dpred_tok, pred_img, C, H, W, p, L, extra, pixel_mean, pixel_std, norm_pix, loss_l1, BL, pooled);
        else if (dtype == SKYEMB_F16)
            hipLaunchKernelGGL(simmim_pass2<f16_t>, dim3(B * (L + extra)), dim3(256), 0, st, imgs, pred_tok, pixel_mask, ws,
                               (f16_t *)dpred_tok, pred_img, C, H, W, p, L, extra, pixel_mean, pixel_std, norm_pix, loss_l1, BL, pooled);
        else
            hipLaunchKernelGGL(simmim_pass2<float>, dim3(B * (L + extra)), dim3(256), 0, st, imgs, pred_tok, pixel_mask, ws,
                               (float *)dpred_tok, pred_img, C, H, W, p, L, extra, pixel_mean, pixel_std, norm_pix, loss_l1, BL, pooled);
    }
    SKY_LAUNCH_CHECK("skyemb_simmim_pixel_loss");
    return 0;
}
