// RA/Dec token of the encoder sequence (utils/mim_vit.py:209-216, 410-414; utils/location_encoder.py:138-243):
//   sh[25]  = real spherical harmonics l = 0..4, m = -l..l at phi = deg2rad(ra), theta = deg2rad(dec + 90)
//             (closed form: associated Legendre recursion in fp32, the reference's operation order)
//   h[8]    = sin(30 * (W0 sh + b0))                      first Siren layer (w0_initial = 30)
//   token   = W1 h + b1 (+ pos_embed[1])                  last Siren layer has an Identity activation
// Tiny (B x 25 x 8 x D): one block per sample forward; backward reduces over the batch in a fixed order.
#include "common.h"

namespace {

constexpr int SH_L = 5, NF = 25, NH = 8;
constexpr float W0_FIRST = 30.0f;

__device__ float assoc_legendre(int l, int m, float x) {
    float pmm = 1.0f;
    if (m > 0) {
        const float somx2 = sqrtf((1.0f - x) * (1.0f + x));
        float fact = 1.0f;
        for (int i = 1; i <= m; ++i) {
            pmm = pmm * (-fact) * somx2;
            fact += 2.0f;
        }
    }
    if (l == m) return pmm;
    float pmmp1 = x * (2.0f * m + 1.0f) * pmm;
    if (l == m + 1) return pmmp1;
    float pll = 0.0f;
    for (int ll = m + 2; ll <= l; ++ll) {
        pll = ((2.0f * ll - 1.0f) * x * pmmp1 - (ll + m - 1.0f) * pmm) / (float)(ll - m);
        pmm = pmmp1;
        pmmp1 = pll;
    }
    return pll;
}

__device__ double factorial_d(int n) {
    double f = 1.0;
    for (int i = 2; i <= n; ++i) f *= i;
    return f;
}

// feature f = l*l + (m + l)
__device__ float sh_feature(int f, float phi, float ct) {
    int l = 0;
    while ((l + 1) * (l + 1) <= f) ++l;
    const int m = f - l * l - l, am = m < 0 ? -m : m;
    // python floats (double) in the reference, cast to fp32 when multiplied with the tensor
    const double norm = sqrt((2.0 * l + 1.0) * factorial_d(l - am) / (4.0 * 3.14159265358979323846 * factorial_d(l + am)));
    const float P = assoc_legendre(l, am, ct);
    if (m == 0) return (float)norm * P;
    const float k = (float)(1.4142135623730951 * norm);
    return m > 0 ? k * cosf((float)m * phi) * P : k * sinf((float)(-m) * phi) * P;
}

// one block (256 threads) per sample
__global__ __launch_bounds__(256) void radec_fwd_kernel(const float *__restrict__ ra_dec, const float *__restrict__ W0,
                                                        const float *__restrict__ b0, const float *__restrict__ W1,
                                                        const float *__restrict__ b1, const float *__restrict__ pos,
                                                        float *__restrict__ x, int64_t row_stride, int D, float *__restrict__ sh_out,
                                                        float *__restrict__ z_out) {
    __shared__ float sh[NF], h[NH];
    const int b = blockIdx.x, t = threadIdx.x;
    if (t < NF) {
        const float phi = ra_dec[2 * b] * 0.017453292519943295f;            // torch.deg2rad: x * (pi / 180)
        const float theta = (ra_dec[2 * b + 1] + 90.0f) * 0.017453292519943295f;
        const float v = sh_feature(t, phi, cosf(theta));
        sh[t] = v;
        sh_out[(int64_t)b * NF + t] = v;
    }
    __syncthreads();
    if (t < NH) {
        float z = 0.f;
        for (int f = 0; f < NF; ++f) z = fmaf(sh[f], W0[t * NF + f], z);
        z += b0[t];
        z_out[(int64_t)b * NH + t] = z;
        h[t] = sinf(W0_FIRST * z);
    }
    __syncthreads();
    for (int d = t; d < D; d += 256) {
        float y = 0.f;
#pragma unroll
        for (int k = 0; k < NH; ++k) y = fmaf(h[k], W1[d * NH + k], y);
        y += b1[d];
        if (pos) y += pos[d];
        x[(int64_t)b * row_stride + d] = y;
    }
}

// dW1[d][k] = sum_b g[b][d] h[b][k], db1[d] = sum_b g[b][d];  dh[b][k] = sum_d g[b][d] W1[d][k]
__global__ __launch_bounds__(256) void radec_bwd_last_kernel(const float *__restrict__ g, int64_t row_stride, const float *__restrict__ z,
                                                             float *__restrict__ dW1, float *__restrict__ db1, int B, int D) {
    const int d = blockIdx.x * 256 + threadIdx.x;
    if (d >= D) return;
    float acc[NH], accb = 0.f;
#pragma unroll
    for (int k = 0; k < NH; ++k) acc[k] = 0.f;
    for (int b = 0; b < B; ++b) {
        const float gv = g[(int64_t)b * row_stride + d];
        accb += gv;
#pragma unroll
        for (int k = 0; k < NH; ++k) acc[k] = fmaf(gv, sinf(W0_FIRST * z[(int64_t)b * NH + k]), acc[k]);
    }
#pragma unroll
    for (int k = 0; k < NH; ++k) dW1[d * NH + k] = acc[k];
    db1[d] = accb;
}

// one block per sample: dz[b][k] = (sum_d g[b][d] W1[d][k]) * 30 cos(30 z[b][k])
__global__ __launch_bounds__(256) void radec_bwd_dz_kernel(const float *__restrict__ g, int64_t row_stride, const float *__restrict__ W1,
                                                           const float *__restrict__ z, float *__restrict__ dz, int D) {
    __shared__ float red[4][NH];
    const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    float acc[NH];
#pragma unroll
    for (int k = 0; k < NH; ++k) acc[k] = 0.f;
    for (int d = t; d < D; d += 256) {
        const float gv = g[(int64_t)b * row_stride + d];
#pragma unroll
        for (int k = 0; k < NH; ++k) acc[k] = fmaf(gv, W1[d * NH + k], acc[k]);
    }
#pragma unroll
    for (int k = 0; k < NH; ++k) {
        const float s = wave_sum(acc[k]);
        if (lane == 0) red[wave][k] = s;
    }
    __syncthreads();
    if (t < NH) {
        const float dh = (red[0][t] + red[1][t]) + (red[2][t] + red[3][t]);
        const float zz = z[(int64_t)b * NH + t];
        dz[(int64_t)b * NH + t] = dh * W0_FIRST * cosf(W0_FIRST * zz);
    }
}

// dW0[k][f] = sum_b dz[b][k] sh[b][f], db0[k] = sum_b dz[b][k]   (one thread per output, fixed order over b)
__global__ __launch_bounds__(256) void radec_bwd_first_kernel(const float *__restrict__ dz, const float *__restrict__ sh,
                                                              float *__restrict__ dW0, float *__restrict__ db0, int B) {
    const int t = threadIdx.x;
    if (t < NH * NF) {
        const int k = t / NF, f = t % NF;
        float acc = 0.f;
        for (int b = 0; b < B; ++b) acc = fmaf(dz[(int64_t)b * NH + k], sh[(int64_t)b * NF + f], acc);
        dW0[t] = acc;
    }
    if (t < NH) {
        float acc = 0.f;
        for (int b = 0; b < B; ++b) acc += dz[(int64_t)b * NH + t];
        db0[t] = acc;
    }
}

}  // namespace

extern "C" int skyemb_radec_token_fwd(const float *ra_dec, const float *W0, const float *b0, const float *W1, const float *b1,
                                      const float *pos_row, float *x, int64_t row_stride, int B, int D, float *sh, float *z,
                                      void *stream) {
    SKY_CHECK_ARG(ra_dec && W0 && b0 && W1 && b1 && x && sh && z && B > 0 && D > 0, "skyemb_radec_token_fwd: bad arguments");
    hipLaunchKernelGGL(radec_fwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, ra_dec, W0, b0, W1, b1, pos_row, x, row_stride,
                       D, sh, z);
    SKY_LAUNCH_CHECK("skyemb_radec_token_fwd");
    return 0;
}

extern "C" int skyemb_radec_token_bwd(const float *g, int64_t row_stride, const float *W1, const float *sh, const float *z,
                                      float *dz_ws, float *dW0, float *db0, float *dW1, float *db1, int B, int D, void *stream) {
    SKY_CHECK_ARG(g && W1 && sh && z && dz_ws && dW0 && db0 && dW1 && db1 && B > 0 && D > 0, "skyemb_radec_token_bwd: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(radec_bwd_last_kernel, dim3((D + 255) / 256), dim3(256), 0, st, g, row_stride, z, dW1, db1, B, D);
    hipLaunchKernelGGL(radec_bwd_dz_kernel, dim3(B), dim3(256), 0, st, g, row_stride, W1, z, dz_ws, D);
    hipLaunchKernelGGL(radec_bwd_first_kernel, dim3(1), dim3(256), 0, st, dz_ws, sh, dW0, db0, B);
    SKY_LAUNCH_CHECK("skyemb_radec_token_bwd");
    return 0;
}
