// Multi-head attention core for tiny sequences (N = 5 / 17 / 65 / 66 tokens): one wavefront owns one
// (sample, head); q, k, v (and dO) are staged in LDS as fp32 rows (16-byte aligned pitch), every inner
// product / accumulation runs on float4 LDS reads, softmax statistics stay in fp32, the backward
// recomputes the probabilities.  0.3 % of the model FLOPs: built for latency, not for MFMA.
//
// qkv layout is the reference's: [B, N, 3, H, hd]  (timm Attention: qkv(x).reshape(B,N,3,H,hd)).
#include "common.h"

namespace {

__device__ __forceinline__ float dot4(const float4 a, const float4 b, float acc) {
    acc = fmaf(a.x, b.x, acc);
    acc = fmaf(a.y, b.y, acc);
    acc = fmaf(a.z, b.z, acc);
    return fmaf(a.w, b.w, acc);
}

// stage one [N, hd] head (rows row_stride elements apart) as fp32 rows of `pitch` floats; 8 elements per lane-step
template <typename T>
__device__ __forceinline__ void load_head(const T *__restrict__ src, int64_t row_stride, float *dst, int N, int hd,
                                          int pitch, float scale, int lane) {
    const int v4 = hd >> 2;
    for (int e = lane; e < N * v4; e += 64) {
        const int n = e / v4, d = (e - n * v4) << 2;
        const float4 x = load4<T>(src + (int64_t)n * row_stride + d);
        *(float4 *)&dst[n * pitch + d] = make_float4(x.x * scale, x.y * scale, x.z * scale, x.w * scale);
    }
}

// P[N][NP] = softmax_j(qs_i . k_j)
__device__ __forceinline__ void scores_softmax(const float *qs, const float *k, float *P, int N, int NP, int hd, int pitch,
                                               int lane) {
    for (int e = lane; e < N * N; e += 64) {
        const int i = e / N, j = e - i * N;
        float a0 = 0.f, a1 = 0.f;
        for (int d = 0; d < hd; d += 8) {
            a0 = dot4(*(const float4 *)&qs[i * pitch + d], *(const float4 *)&k[j * pitch + d], a0);
            a1 = dot4(*(const float4 *)&qs[i * pitch + d + 4], *(const float4 *)&k[j * pitch + d + 4], a1);
        }
        P[i * NP + j] = a0 + a1;
    }
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < N; i += 64) {
        float mx = -INFINITY;
        for (int j = 0; j < N; ++j) mx = fmaxf(mx, P[i * NP + j]);
        float sum = 0.f;
        for (int j = 0; j < N; ++j) {
            const float ev = __expf(P[i * NP + j] - mx);
            P[i * NP + j] = ev;
            sum += ev;
        }
        const float inv = 1.0f / sum;
        for (int j = 0; j < N; ++j) P[i * NP + j] *= inv;
    }
    __builtin_amdgcn_wave_barrier();
}

template <typename T>
__global__ void mha_fwd_kernel(const T *__restrict__ qkv, T *__restrict__ out, int B, int N, int H, int hd, int waves,
                               int per_wave_floats) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int head_id = blockIdx.x * waves + wave;
    if (head_id >= B * H) return;
    const int b = head_id / H, h = head_id - b * H;
    const int pitch = hd + 4, D = H * hd, NP = N + 1;
    float *q = lds + (size_t)wave * per_wave_floats, *k = q + N * pitch, *v = k + N * pitch, *P = v + N * pitch;
    const T *base = qkv + (int64_t)b * N * 3 * D + h * hd;
    load_head<T>(base, 3 * D, q, N, hd, pitch, rsqrtf((float)hd), lane);
    load_head<T>(base + D, 3 * D, k, N, hd, pitch, 1.0f, lane);
    load_head<T>(base + 2 * D, 3 * D, v, N, hd, pitch, 1.0f, lane);
    __builtin_amdgcn_wave_barrier();
    scores_softmax(q, k, P, N, NP, hd, pitch, lane);
    const int v4 = hd >> 2;
    for (int e = lane; e < N * v4; e += 64) {
        const int i = e / v4, d = (e - i * v4) << 2;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int j = 0; j < N; ++j) {
            const float p = P[i * NP + j];
            const float4 x = *(const float4 *)&v[j * pitch + d];
            acc.x = fmaf(p, x.x, acc.x); acc.y = fmaf(p, x.y, acc.y); acc.z = fmaf(p, x.z, acc.z); acc.w = fmaf(p, x.w, acc.w);
        }
        store4<T>(out + ((int64_t)b * N + i) * D + h * hd + d, acc.x, acc.y, acc.z, acc.w);
    }
}

template <typename T>
__global__ void mha_bwd_kernel(const T *__restrict__ qkv, const T *__restrict__ dout, T *__restrict__ dqkv, int B,
                               int N, int H, int hd, int waves, int per_wave_floats) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int head_id = blockIdx.x * waves + wave;
    if (head_id >= B * H) return;
    const int b = head_id / H, h = head_id - b * H;
    const int pitch = hd + 4, D = H * hd, NP = N + 1;
    const float scale = rsqrtf((float)hd);
    float *q = lds + (size_t)wave * per_wave_floats, *k = q + N * pitch, *v = k + N * pitch, *dO = v + N * pitch;
    float *P = dO + N * pitch, *dS = P + N * NP;
    const T *base = qkv + (int64_t)b * N * 3 * D + h * hd;
    load_head<T>(base, 3 * D, q, N, hd, pitch, scale, lane);
    load_head<T>(base + D, 3 * D, k, N, hd, pitch, 1.0f, lane);
    load_head<T>(base + 2 * D, 3 * D, v, N, hd, pitch, 1.0f, lane);
    load_head<T>(dout + (int64_t)b * N * D + h * hd, D, dO, N, hd, pitch, 1.0f, lane);
    __builtin_amdgcn_wave_barrier();
    scores_softmax(q, k, P, N, NP, hd, pitch, lane);
    // dP = dO v^T
    for (int e = lane; e < N * N; e += 64) {
        const int i = e / N, j = e - i * N;
        float a0 = 0.f, a1 = 0.f;
        for (int d = 0; d < hd; d += 8) {
            a0 = dot4(*(const float4 *)&dO[i * pitch + d], *(const float4 *)&v[j * pitch + d], a0);
            a1 = dot4(*(const float4 *)&dO[i * pitch + d + 4], *(const float4 *)&v[j * pitch + d + 4], a1);
        }
        dS[i * NP + j] = a0 + a1;
    }
    __builtin_amdgcn_wave_barrier();
    // dS = P * (dP - rowsum(P * dP))
    for (int i = lane; i < N; i += 64) {
        float dot = 0.f;
        for (int j = 0; j < N; ++j) dot = fmaf(P[i * NP + j], dS[i * NP + j], dot);
        for (int j = 0; j < N; ++j) dS[i * NP + j] = P[i * NP + j] * (dS[i * NP + j] - dot);
    }
    __builtin_amdgcn_wave_barrier();
    T *dbase = dqkv + (int64_t)b * N * 3 * D + h * hd;
    const int v4 = hd >> 2;
    for (int e = lane; e < N * v4; e += 64) {
        const int n = e / v4, d = (e - n * v4) << 2;
        float4 aq = make_float4(0.f, 0.f, 0.f, 0.f), ak = aq, av = aq;
        for (int j = 0; j < N; ++j) {
            const float s_nj = dS[n * NP + j], s_jn = dS[j * NP + n], p_jn = P[j * NP + n];
            const float4 kj = *(const float4 *)&k[j * pitch + d], qj = *(const float4 *)&q[j * pitch + d];
            const float4 oj = *(const float4 *)&dO[j * pitch + d];
            aq.x = fmaf(s_nj, kj.x, aq.x); aq.y = fmaf(s_nj, kj.y, aq.y); aq.z = fmaf(s_nj, kj.z, aq.z); aq.w = fmaf(s_nj, kj.w, aq.w);
            ak.x = fmaf(s_jn, qj.x, ak.x); ak.y = fmaf(s_jn, qj.y, ak.y); ak.z = fmaf(s_jn, qj.z, ak.z); ak.w = fmaf(s_jn, qj.w, ak.w);
            av.x = fmaf(p_jn, oj.x, av.x); av.y = fmaf(p_jn, oj.y, av.y); av.z = fmaf(p_jn, oj.z, av.z); av.w = fmaf(p_jn, oj.w, av.w);
        }
        const int64_t o = (int64_t)n * 3 * D + d;
        store4<T>(dbase + o, aq.x * scale, aq.y * scale, aq.z * scale, aq.w * scale);   // dq = scale * dS k
        store4<T>(dbase + o + D, ak.x, ak.y, ak.z, ak.w);                                // dk = dS^T (scale q)
        store4<T>(dbase + o + 2 * D, av.x, av.y, av.z, av.w);                            // dv = P^T dO
    }
}

struct Plan {
    int waves, per_wave_floats;
    size_t smem;
};

Plan make_plan(int N, int hd, bool bwd) {
    Plan p;
    const int pitch = hd + 4, NP = N + 1;
    p.per_wave_floats = (bwd ? 4 : 3) * N * pitch + (bwd ? 2 : 1) * N * NP;
    p.per_wave_floats = (p.per_wave_floats + 3) & ~3;
    const size_t per = (size_t)p.per_wave_floats * 4;
    int w = (int)(65536 / per);
    if (w > 4) w = 4;
    if (w < 1) w = 1;
    p.waves = w;
    p.smem = per * w;
    return p;
}

template <typename K>
int set_lds(K kern, size_t smem, const char *name) {
    if (smem > 160 * 1024) {
        skyemb_set_error("%s: sequence too long for the small-N kernel (%zu B LDS per wave)", name, smem);
        return 1;
    }
    if (smem > 65536) {
        hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) {
            skyemb_set_error("%s: hipFuncSetAttribute: %s", name, hipGetErrorString(e));
            return 2;
        }
    }
    return 0;
}

}  // namespace

// attention_mfma.hip: bf16, N <= 32, head dim 32 / 64 (the shapes of the MAE path); -1 = not handled
int skyemb_mha_mfma_try(bool bwd, const void *qkv, const void *dout, void *out, int dtype, int B, int N, int H, int hd, hipStream_t st);

extern "C" int skyemb_mha_fwd(const void *qkv, void *out, int dtype, int B, int N, int H, int hd, void *stream) {
    SKY_CHECK_ARG(B > 0 && N > 0 && H > 0 && hd > 0 && hd % 8 == 0, "skyemb_mha_fwd: bad shape (head dim must be a multiple of 8)");
    hipStream_t st = (hipStream_t)stream;
    if (sky_is_lp(dtype) && skyemb_mha_mfma_try(false, qkv, nullptr, out, dtype, B, N, H, hd, st) == 0) {
        SKY_LAUNCH_CHECK("skyemb_mha_fwd");
        return 0;
    }
    const Plan p = make_plan(N, hd, false);
    dim3 grid((B * H + p.waves - 1) / p.waves), block(64 * p.waves);
    int rc;
    if (dtype == SKYEMB_BF16) {
        if ((rc = set_lds(mha_fwd_kernel<bf16_t>, p.smem, "skyemb_mha_fwd"))) return rc;
        hipLaunchKernelGGL(mha_fwd_kernel<bf16_t>, grid, block, p.smem, st, (const bf16_t *)qkv, (bf16_t *)out, B, N, H, hd,
                           p.waves, p.per_wave_floats);
        } else if (dtype == SKYEMB_F16) {
        if ((rc = set_lds(mha_fwd_kernel<f16_t>, p.smem, "skyemb_mha_fwd"))) return rc;
        hipLaunchKernelGGL(mha_fwd_kernel<f16_t>, grid, block, p.smem, st, (const f16_t *)qkv, (f16_t *)out, B, N, H, hd,
                           p.waves, p.per_wave_floats);
    } else {
        if ((rc = set_lds(mha_fwd_kernel<float>, p.smem, "skyemb_mha_fwd"))) return rc;
        hipLaunchKernelGGL(mha_fwd_kernel<float>, grid, block, p.smem, st, (const float *)qkv, (float *)out, B, N, H, hd,
                           p.waves, p.per_wave_floats);
    }
    SKY_LAUNCH_CHECK("skyemb_mha_fwd");
    return 0;
}

extern "C" int skyemb_mha_bwd(const void *qkv, const void *dout, void *dqkv, int dtype, int B, int N, int H, int hd,
                              void *stream) {
    SKY_CHECK_ARG(B > 0 && N > 0 && H > 0 && hd > 0 && hd % 8 == 0, "skyemb_mha_bwd: bad shape (head dim must be a multiple of 8)");
    hipStream_t st = (hipStream_t)stream;
    if (sky_is_lp(dtype) && skyemb_mha_mfma_try(true, qkv, dout, dqkv, dtype, B, N, H, hd, st) == 0) {
        SKY_LAUNCH_CHECK("skyemb_mha_bwd");
        return 0;
    }
    const Plan p = make_plan(N, hd, true);
    dim3 grid((B * H + p.waves - 1) / p.waves), block(64 * p.waves);
    int rc;
    if (dtype == SKYEMB_BF16) {
        if ((rc = set_lds(mha_bwd_kernel<bf16_t>, p.smem, "skyemb_mha_bwd"))) return rc;
        hipLaunchKernelGGL(mha_bwd_kernel<bf16_t>, grid, block, p.smem, st, (const bf16_t *)qkv, (const bf16_t *)dout,
                           (bf16_t *)dqkv, B, N, H, hd, p.waves, p.per_wave_floats);
        } else if (dtype == SKYEMB_F16) {
        if ((rc = set_lds(mha_bwd_kernel<f16_t>, p.smem, "skyemb_mha_bwd"))) return rc;
        hipLaunchKernelGGL(mha_bwd_kernel<f16_t>, grid, block, p.smem, st, (const f16_t *)qkv, (const f16_t *)dout,
                           (f16_t *)dqkv, B, N, H, hd, p.waves, p.per_wave_floats);
    } else {
        if ((rc = set_lds(mha_bwd_kernel<float>, p.smem, "skyemb_mha_bwd"))) return rc;
        hipLaunchKernelGGL(mha_bwd_kernel<float>, grid, block, p.smem, st, (const float *)qkv, (const float *)dout,
                           (float *)dqkv, B, N, H, hd, p.waves, p.per_wave_floats);
    }
    SKY_LAUNCH_CHECK("skyemb_mha_bwd");
    return 0;
}
