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

// The patch's pv = C p p target elements into LDS in patchify order e = (py p + px) C + c, READ in image order (rows of p
// consecutive pixels: float4 loads of whole 64-byte rows instead of one 4-byte request per element with the channel -- 16 KB
// apart -- as the fastest index: round 6, loss_pass1 26.7 -> ~10 us at config A).  Same values as target_elem, so every sum below
// keeps its bits.  pv <= STAGE_MAX, p % 4 == 0.
constexpr int STAGE_MAX = 12 * 256;
__device__ __forceinline__ void stage_patch(const float *__restrict__ imgs, int b, int l, int C, int H, int W, int p, float mean,
                                            float stdv, float *__restrict__ lds) {
    const int gw = W / p, p4 = p >> 2, n4 = C * p * p4;
    const int y0 = (l / gw) * p, x0 = (l % gw) * p;
    for (int j = threadIdx.x; j < n4; j += 256) {
        const int px4 = j % p4, py = (j / p4) % p, c = j / (p4 * p);
        const float4 v = *(const float4 *)(imgs + (((int64_t)b * C + c) * H + y0 + py) * W + x0 + 4 * px4);
        float *d = lds + ((py * p + 4 * px4) * C + c);
        d[0] = (v.x - mean) / stdv;
        d[C] = (v.y - mean) / stdv;
        d[2 * C] = (v.z - mean) / stdv;
        d[3 * C] = (v.w - mean) / stdv;
    }
    __syncthreads();
}

// the scalar reduction of the loss (loss_finalize).  (Round 6 tried it inside pass 1, run by the last workgroup to arrive at a
// counter: 4096 returning atomic adds on one word serialise at ~90 per us -- pass 1 went from 27 to 55 us -- and a release fence per
// workgroup writes back the XCD's whole L2, 0.35 ms in all; one more 1.5 us launch boundary is the cheaper form.)
__device__ __forceinline__ void finalize_body(float *__restrict__ ws, float *__restrict__ loss, int BL, float numel, float dscale, float *red) {
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
__global__ __launch_bounds__(256) void loss_pass1(const float *__restrict__ imgs, const float *__restrict__ pred,
                                                  const float *__restrict__ mask, float *__restrict__ ws, int C, int H,
                                                  int W, int p, int L, int extra, float mean, float stdv, int norm_pix,
                                                  int loss_l1) {
    __shared__ float red[4];
    __shared__ __attribute__((aligned(16))) float patch[STAGE_MAX];
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
        stage_patch(imgs, b, l, C, H, W, p, mean, stdv, patch);
#pragma unroll
        for (int u = 0; u < MAXE; ++u) {
            const int e = threadIdx.x + u * 256;
            tv[u] = e < pv ? patch[e] : 0.f;
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
    finalize_body(ws, loss, BL, numel, dscale, red);
}

template <typename T>
__device__ __forceinline__ void store8t(T *p, const float (&v)[8]) {
    store4<T>(p, v[0], v[1], v[2], v[3]);
    store4<T>(p + 4, v[4], v[5], v[6], v[7]);
}

template <typename T>
__global__ __launch_bounds__(256) void loss_pass2(const float *__restrict__ imgs, const float *__restrict__ pred,
                                                  const float *__restrict__ mask, const float *__restrict__ ws,
                                                  T *__restrict__ dpred, float *__restrict__ dpred32, int C, int H, int W,
                                                  int p, int L, int extra, float mean, float stdv, int norm_pix,
                                                  int loss_l1, int BL) {
    __shared__ __attribute__((aligned(16))) float patch[STAGE_MAX];
    const int Nd = L + extra;
    const int b = blockIdx.x / Nd, r = blockIdx.x % Nd;
    const int pv = C * p * p;                              // a multiple of 16 (p % 4 == 0)
    const int64_t off = (int64_t)blockIdx.x * pv;
    const int l = r - extra;
    const bool live = l >= 0 && mask[(int64_t)b * L + l] != 0.0f;
    if (!live) {
        const float z[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int e = 8 * threadIdx.x; e < pv; e += 8 * 256) {
            if (dpred) store8t<T>(dpred + off + e, z);
            if (dpred32) { *(float4 *)(dpred32 + off + e) = make_float4(0.f, 0.f, 0.f, 0.f); *(float4 *)(dpred32 + off + e + 4) = make_float4(0.f, 0.f, 0.f, 0.f); }
        }
        return;
    }
    const float inv = ws[4 * (int64_t)BL + 1];
    const float mu = ws[((int64_t)b * L + l) * 4 + 2], istd = ws[((int64_t)b * L + l) * 4 + 3];
    const bool cached = pv <= STAGE_MAX;
    if (cached) stage_patch(imgs, b, l, C, H, W, p, mean, stdv, patch);
    for (int e0 = 8 * threadIdx.x; e0 < pv; e0 += 8 * 256) {  // 8 consecutive elements per thread: 32-byte loads, 16 / 32-byte stores
        const float4 p0 = *(const float4 *)(pred + off + e0), p1 = *(const float4 *)(pred + off + e0 + 4);
        const float pr[8] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w};
        float g[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float t = cached ? patch[e0 + k] : target_elem(imgs, b, l, e0 + k, C, H, W, p, mean, stdv);
            if (norm_pix) t = (t - mu) * istd;
            const float d = pr[k] - t;  // d loss_e / d pred = 2 d (mse) | sign(d) (l1)
            g[k] = 0.f;
            if (d == d) g[k] = (loss_l1 ? (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) : 2.f * d) * inv;
        }
        if (dpred) store8t<T>(dpred + off + e0, g);
        if (dpred32) { *(float4 *)(dpred32 + off + e0) = make_float4(g[0], g[1], g[2], g[3]); *(float4 *)(dpred32 + off + e0 + 4) = make_float4(g[4], g[5], g[6], g[7]); }
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
    SKY_CHECK_ARG(p % 4 == 0 && aligned16(imgs) && aligned16(pred) && (!dpred || aligned16(dpred)) && (!dpred32 || aligned16(dpred32)) && W % 4 == 0,
                  "skyemb_masked_patch_loss: patch size and image width must be multiples of 4, buffers 16-byte aligned");
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
