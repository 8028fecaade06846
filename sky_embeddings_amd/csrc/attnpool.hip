// Attention pooling over a sample's tokens with ONE learned query (timm AttentionPoolLatent as the reference builds it,
// utils/mim_vit.py:246-249, 426-427: latent_len = 1, no q/k norm; the SimMIM variant whose head up-samples the pooled
// token to the whole image).  The query does not depend on the sample: q = Wq latent + bq is computed once per step.
//   forward : s_j = scale q_h . k_{b,j,h}   p = softmax_j(s)   o_{b,h} = sum_j p_j v_{b,j,h}
//   backward: dp_j = do . v_j   ds_j = p_j (dp_j - sum_j p_j dp_j)   dv_j = p_j do   dk_j = scale ds_j q   dq += scale ds_j k_j
// kv is the kv-projection's output [B, N, 2, H, hd] (row = token, k then v).  One wavefront per (sample, head); everything
// is a few KB per wave: latency-bound, fp32 arithmetic, operands in the compute dtype.
#include "common.h"

namespace {

constexpr int MAXN = 256;   // tokens per sample (cls + RA/Dec + patches)
constexpr int MAXHD = 512;  // columns per head (the downstream predictor pools with TWO heads: 384 at ViT-B, 512 at ViT-L)

// q[o] = bq[o] + sum_i Wq[o][i] latent[i]; one wave per output
__global__ __launch_bounds__(256) void attnpool_q_kernel(const float *__restrict__ latent, const float *__restrict__ Wq,
                                                         const float *__restrict__ bq, float *__restrict__ q, int D) {
    const int lane = threadIdx.x & 63, o = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (o >= D) return;
    float s = 0.f;
    for (int i = lane; i < D; i += 64) s = fmaf(Wq[(int64_t)o * D + i], latent[i], s);
    s = wave_sum(s);
    if (lane == 0) q[o] = s + bq[o];
}

template <typename T>
__global__ __launch_bounds__(256) void attnpool_fwd_kernel(const float *__restrict__ q, const T *__restrict__ kv,
                                                           T *__restrict__ out, float *__restrict__ prob, int B, int N, int H,
                                                           int hd) {
    __shared__ float sq[4][MAXHD], sp[4][MAXN];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int bh = blockIdx.x * 4 + wave;
    if (bh >= B * H) return;
    const int b = bh / H, h = bh - b * H, D = H * hd;
    const float scale = rsqrtf((float)hd);
    for (int d = lane; d < hd; d += 64) sq[wave][d] = q[h * hd + d];
    __builtin_amdgcn_wave_barrier();
    const T *kb = kv + (int64_t)b * N * 2 * D + h * hd;
    float mx = -INFINITY;
    for (int j = lane; j < N; j += 64) {
        const T *kr = kb + (int64_t)j * 2 * D;
        float s = 0.f;
        for (int d = 0; d < hd; d += 4) {
            const float4 kk = load4<T>(kr + d);
            s = fmaf(sq[wave][d], kk.x, fmaf(sq[wave][d + 1], kk.y, fmaf(sq[wave][d + 2], kk.z, fmaf(sq[wave][d + 3], kk.w, s))));
        }
        s *= scale;
        sp[wave][j] = s;
        mx = fmaxf(mx, s);
    }
    mx = wave_max(mx);
    float sum = 0.f;
    for (int j = lane; j < N; j += 64) {
        const float e = __expf(sp[wave][j] - mx);
        sp[wave][j] = e;
        sum += e;
    }
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;
    for (int j = lane; j < N; j += 64) {
        const float p = sp[wave][j] * inv;
        sp[wave][j] = p;
        prob[(int64_t)bh * N + j] = p;
    }
    __builtin_amdgcn_wave_barrier();
    for (int d = lane; d < hd; d += 64) {
        float o = 0.f;
        for (int j = 0; j < N; ++j) o = fmaf(sp[wave][j], to_f32<T>(kb[(int64_t)j * 2 * D + D + d]), o);
        out[(int64_t)b * D + h * hd + d] = from_f32<T>(o);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void attnpool_bwd_kernel(const float *__restrict__ q, const T *__restrict__ kv,
                                                           const T *__restrict__ dout, const float *__restrict__ prob,
                                                           T *__restrict__ dkv, float *__restrict__ dq_part, int B, int N, int H,
                                                           int hd) {
    __shared__ float sq[4][MAXHD], sdo[4][MAXHD], sds[4][MAXN];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int bh = blockIdx.x * 4 + wave;
    if (bh >= B * H) return;
    const int b = bh / H, h = bh - b * H, D = H * hd;
    const float scale = rsqrtf((float)hd);
    for (int d = lane; d < hd; d += 64) {
        sq[wave][d] = q[h * hd + d];
        sdo[wave][d] = to_f32<T>(dout[(int64_t)b * D + h * hd + d]);
    }
    __builtin_amdgcn_wave_barrier();
    const T *kb = kv + (int64_t)b * N * 2 * D + h * hd;
    T *dkb = dkv + (int64_t)b * N * 2 * D + h * hd;
    float rs = 0.f;
    for (int j = lane; j < N; j += 64) {
        const T *vr = kb + (int64_t)j * 2 * D + D;
        float dp = 0.f;
        for (int d = 0; d < hd; d += 4) {
            const float4 vv = load4<T>(vr + d);
            dp = fmaf(sdo[wave][d], vv.x, fmaf(sdo[wave][d + 1], vv.y, fmaf(sdo[wave][d + 2], vv.z, fmaf(sdo[wave][d + 3], vv.w, dp))));
        }
        sds[wave][j] = dp;
        rs = fmaf(prob[(int64_t)bh * N + j], dp, rs);
    }
    rs = wave_sum(rs);
    for (int j = lane; j < N; j += 64) {
        const float p = prob[(int64_t)bh * N + j];
        const float ds = p * (sds[wave][j] - rs);
        sds[wave][j] = ds * scale;
        T *dk = dkb + (int64_t)j * 2 * D, *dv = dk + D;
        for (int d = 0; d < hd; d += 4) {
            store4<T>(dk + d, ds * scale * sq[wave][d], ds * scale * sq[wave][d + 1], ds * scale * sq[wave][d + 2], ds * scale * sq[wave][d + 3]);
            store4<T>(dv + d, p * sdo[wave][d], p * sdo[wave][d + 1], p * sdo[wave][d + 2], p * sdo[wave][d + 3]);
        }
    }
    __builtin_amdgcn_wave_barrier();
    for (int d = lane; d < hd; d += 64) {
        float a = 0.f;
        for (int j = 0; j < N; ++j) a = fmaf(sds[wave][j], to_f32<T>(kb[(int64_t)j * 2 * D + d]), a);
        dq_part[(int64_t)b * D + h * hd + d] = a;
    }
}

// block o: dq[o] = sum_b dq_part[b][o] (fixed order);  dWq[o][:] = dq[o] latent;  dbq[o] = dq[o];  dqv[o] = dq[o]
__global__ __launch_bounds__(256) void attnpool_q_bwd1_kernel(const float *__restrict__ dq_part, int B, const float *__restrict__ latent,
                                                              float *__restrict__ dWq, float *__restrict__ dbq,
                                                              float *__restrict__ dqv, int D) {
    __shared__ float red[256];
    const int o = blockIdx.x, tid = threadIdx.x;
    float s = 0.f;
    for (int b = tid; b < B; b += 256) s += dq_part[(int64_t)b * D + o];
    red[tid] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (tid < w) red[tid] += red[tid + w];
        __syncthreads();
    }
    const float dq = red[0];
    for (int i = tid; i < D; i += 256) dWq[(int64_t)o * D + i] = dq * latent[i];
    if (tid == 0) {
        dbq[o] = dq;
        dqv[o] = dq;
    }
}
// dlatent[i] = sum_o Wq[o][i] dq[o]
__global__ __launch_bounds__(256) void attnpool_q_bwd2_kernel(const float *__restrict__ dqv, const float *__restrict__ Wq,
                                                              float *__restrict__ dlatent, int D) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= D) return;
    float s = 0.f;
    for (int o = 0; o < D; ++o) s = fmaf(Wq[(int64_t)o * D + i], dqv[o], s);
    dlatent[i] = s;
}

}  // namespace

extern "C" int skyemb_attnpool_q(const float *latent, const float *Wq, const float *bq, float *q, int D, void *stream) {
    SKY_CHECK_ARG(latent && Wq && bq && q && D > 0, "skyemb_attnpool_q: bad arguments");
    hipLaunchKernelGGL(attnpool_q_kernel, dim3((D + 3) / 4), dim3(256), 0, (hipStream_t)stream, latent, Wq, bq, q, D);
    SKY_LAUNCH_CHECK("skyemb_attnpool_q");
    return 0;
}

extern "C" int skyemb_attnpool_fwd(const float *q, const void *kv, int dtype, void *out, float *prob, int B, int N, int H, int hd,
                                   void *stream) {
    SKY_CHECK_ARG(q && kv && out && prob && B > 0 && N > 0 && N <= MAXN && H > 0 && hd > 0 && hd <= MAXHD && hd % 4 == 0,
                  "skyemb_attnpool_fwd: bad shape B=%d N=%d H=%d hd=%d (N <= %d, hd <= %d, hd %% 4 == 0)", B, N, H, hd, MAXN, MAXHD);
    const dim3 grid((B * H + 3) / 4), block(256);
    if (dtype == SKYEMB_BF16)
        hipLaunchKernelGGL(attnpool_fwd_kernel<bf16_t>, grid, block, 0, (hipStream_t)stream, q, (const bf16_t *)kv, (bf16_t *)out, prob, B, N, H, hd);
    else if (dtype == SKYEMB_F16)
        hipLaunchKernelGGL(attnpool_fwd_kernel<f16_t>, grid, block, 0, (hipStream_t)stream, q, (const f16_t *)kv, (f16_t *)out, prob, B, N, H, hd);
    else
        hipLaunchKernelGGL(attnpool_fwd_kernel<float>, grid, block, 0, (hipStream_t)stream, q, (const float *)kv, (float *)out, prob, B, N, H, hd);
    SKY_LAUNCH_CHECK("skyemb_attnpool_fwd");
    return 0;
}

extern "C" int skyemb_attnpool_bwd(const float *q, const void *kv, int dtype, const void *dout, const float *prob, void *dkv,
                                   float *dq_part, int B, int N, int H, int hd, void *stream) {
    SKY_CHECK_ARG(q && kv && dout && prob && dkv && dq_part && B > 0 && N > 0 && N <= MAXN && H > 0 && hd > 0 && hd <= MAXHD && hd % 4 == 0,
                  "skyemb_attnpool_bwd: bad shape B=%d N=%d H=%d hd=%d", B, N, H, hd);
    const dim3 grid((B * H + 3) / 4), block(256);
    if (dtype == SKYEMB_BF16)
        hipLaunchKernelGGL(attnpool_bwd_kernel<bf16_t>, grid, block, 0, (hipStream_t)stream, q, (const bf16_t *)kv, (const bf16_t *)dout, prob,
                           (bf16_t *)dkv, dq_part, B, N, H, hd);
    else if (dtype == SKYEMB_F16)
        hipLaunchKernelGGL(attnpool_bwd_kernel<f16_t>, grid, block, 0, (hipStream_t)stream, q, (const f16_t *)kv, (const f16_t *)dout, prob,
                           (f16_t *)dkv, dq_part, B, N, H, hd);
    else
        hipLaunchKernelGGL(attnpool_bwd_kernel<float>, grid, block, 0, (hipStream_t)stream, q, (const float *)kv, (const float *)dout, prob,
                           (float *)dkv, dq_part, B, N, H, hd);
    SKY_LAUNCH_CHECK("skyemb_attnpool_bwd");
    return 0;
}

extern "C" int skyemb_attnpool_q_bwd(const float *dq_part, int B, const float *latent, const float *Wq, float *dWq, float *dbq,
                                     float *dlatent, float *ws, int D, void *stream) {
    SKY_CHECK_ARG(dq_part && latent && Wq && dWq && dbq && dlatent && ws && B > 0 && D > 0, "skyemb_attnpool_q_bwd: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(attnpool_q_bwd1_kernel, dim3(D), dim3(256), 0, st, dq_part, B, latent, dWq, dbq, ws, D);
    hipLaunchKernelGGL(attnpool_q_bwd2_kernel, dim3((D + 255) / 256), dim3(256), 0, st, (const float *)ws, Wq, dlatent, D);
    SKY_LAUNCH_CHECK("skyemb_attnpool_q_bwd");
    return 0;
}
