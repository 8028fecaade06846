// Target-augmentation pipeline of the similarity search on the device (utils/dataloaders.py:14-106 get_augmentations,
// applied per sample in utils/eval_fns.py:88-108): horizontal / vertical flip, RandomResizedCrop back to the cutout size
// (torchvision resized_crop = crop + bilinear interpolate(align_corners=False, antialias=True)), brightness factor,
// additive Gaussian noise, whole channels set to NaN.  The reference runs it per (sample, copy) in Python on the host;
// here ONE launch writes the original and its A augmented copies for a whole batch.  The random PARAMETERS are drawn by
// the caller (sky_embeddings_amd/augment.py mirrors torchvision's get_params); this kernel is the arithmetic.
//   params[n] = {flip_h, flip_v, crop_top, crop_left, crop_h, crop_w, brightness, sigma}   nan_mask[n]: bit c -> channel c = NaN
// out[n] for n = b * (1 + A) + a:  a == 0 -> the unchanged sample b;  a >= 1 -> augmented copy a.
#include "common.h"

namespace {

struct Taps {
    int lo, n;
    float w[3];
};
// torch's separable anti-aliased bilinear weights (aten UpSampleKernel.cpp, _compute_indices_min_size_weights_aa):
// scale = in / out, support = max(scale, 1), centre = scale * (o + 0.5), taps [centre - support + 0.5, centre + support + 0.5)
__device__ __forceinline__ Taps taps_of(int o, int in_size, int out_size) {
    const float scale = (float)in_size / (float)out_size;
    const float support = scale >= 1.0f ? scale : 1.0f, invscale = scale >= 1.0f ? 1.0f / scale : 1.0f;
    const float center = scale * ((float)o + 0.5f);
    Taps t;
    t.lo = (int)(center - support + 0.5f);
    if (t.lo < 0) t.lo = 0;
    int hi = (int)(center + support + 0.5f);
    if (hi > in_size) hi = in_size;
    t.n = hi - t.lo;
    if (t.n > 3) t.n = 3;                    // crops never exceed the output size here (scale <= 1): at most 3 taps
    float total = 0.f;
    for (int j = 0; j < 3; ++j) {
        float x = ((float)(j + t.lo) - center + 0.5f) * invscale;
        x = fabsf(x);
        const float w = (j < t.n && x < 1.0f) ? 1.0f - x : 0.f;
        t.w[j] = w;
        total += w;
    }
    for (int j = 0; j < 3; ++j) t.w[j] = total != 0.f ? t.w[j] / total : 0.f;
    return t;
}

__global__ __launch_bounds__(256) void augment_kernel(const float *__restrict__ imgs, float *__restrict__ out,
                                                      const float *__restrict__ params, const int *__restrict__ nan_mask,
                                                      const float *__restrict__ noise, int C, int S, int A, int64_t total) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const int x = (int)(e % S), y = (int)((e / S) % S), c = (int)((e / ((int64_t)S * S)) % C);
    const int64_t n = e / ((int64_t)S * S * C);
    const int64_t b = n / (1 + A);
    const int a = (int)(n % (1 + A));
    const float *src = imgs + (b * C + c) * (int64_t)S * S;
    if (a == 0) {
        out[e] = src[(int64_t)y * S + x];
        return;
    }
    if ((nan_mask[n] >> c) & 1) {
        out[e] = NAN;
        return;
    }
    const float *p = params + n * 8;
    const bool fh = p[0] != 0.f, fv = p[1] != 0.f;
    const int ci = (int)p[2], cj = (int)p[3], ch = (int)p[4], cw = (int)p[5];
    const Taps ty = taps_of(y, ch, S), tx = taps_of(x, cw, S);
    float v = 0.f;
    for (int ky = 0; ky < ty.n; ++ky) {
        int yy = ci + ty.lo + ky;
        yy = fv ? S - 1 - yy : yy;
        float row = 0.f;
        for (int kx = 0; kx < tx.n; ++kx) {
            int xx = cj + tx.lo + kx;
            xx = fh ? S - 1 - xx : xx;
            row += src[(int64_t)yy * S + xx] * tx.w[kx];          // horizontal pass first, as torch's separable kernel
        }
        v += row * ty.w[ky];
    }
    v *= p[6];
    if (noise) v += noise[e] * p[7];
    out[e] = v;
}

}  // namespace

extern "C" int skyemb_augment(const float *imgs, float *out, const float *params, const int32_t *nan_mask, const float *noise, int B,
                              int C, int S, int A, void *stream) {
    SKY_CHECK_ARG(imgs && out && params && nan_mask && B > 0 && C > 0 && C <= 32 && S > 0 && A >= 0, "skyemb_augment: bad arguments");
    const int64_t total = (int64_t)B * (1 + A) * C * S * S;
    hipLaunchKernelGGL(augment_kernel, dim3((unsigned)ceil_div64(total, 256)), dim3(256), 0, (hipStream_t)stream, imgs, out, params,
                       nan_mask, noise, C, S, A, total);
    SKY_LAUNCH_CHECK("skyemb_augment");
    return 0;
}
