// Multi-head attention core for tiny sequences (N = 5 / 17 / 65 / 66 tokens): one wavefront owns one
// (sample, head); q, k, v (and dO) are staged in LDS as fp32, softmax statistics stay in fp32, the
// backward recomputes the probabilities.  0.3 % of the model FLOPs: built for latency, not for MFMA.
//
// qkv layout is the reference's: [B, N, 3, H, hd]  (timm Attention: qkv(x).reshape(B,N,3,H,hd)).
#include "common.h"

namespace {

template <typename T>
__device__ __forceinline__ void load_head(const T *__restrict__ src, int64_t row_stride, float *dst, int N, int hd,
                                          int pitch, float scale, int lane) {
    // src points at [n=0][d=0] of this head; rows are row_stride elements apart
    for (int e = lane; e < N * hd; e += 64) {
        const int n = e / hd, d = e - n * hd;
        dst[n * pitch + d] = to_f32<T>(src[(int64_t)n * row_stride + d]) * scale;
    }
}

// probabilities P[N][N] (pitch N) from qs (pre-scaled q) and k
__device__ __forceinline__ void scores_softmax(const float *qs, const float *k, float *P, int N, int hd, int pitch,
                                               int lane) {
    for (int e = lane; e < N * N; e += 64) {
        const int i = e / N, j = e - i * N;
        float acc = 0.f;
        for (int d = 0; d < hd; ++d) acc = fmaf(qs[i * pitch + d], k[j * pitch + d], acc);
        P[e] = acc;
    }
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < N; i += 64) {
        float mx = -INFINITY;
        for (int j = 0; j < N; ++j) mx = fmaxf(mx, P[i * N + j]);
        float sum = 0.f;
        for (int j = 0; j < N; ++j) {
            const float ev = expf(P[i * N + j] - mx);
            P[i * N + j] = ev;
            sum += ev;
        }
        const float inv = 1.0f / sum;
        for (int j = 0; j < N; ++j) P[i * N + j] *= inv;
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
    const int pitch = hd + 1, D = H * hd;
    float *q = lds + (size_t)wave * per_wave_floats, *k = q + N * pitch, *v = k + N * pitch, *P = v + N * pitch;
    const T *base = qkv + (int64_t)b * N * 3 * D + h * hd;
    load_head<T>(base, 3 * D, q, N, hd, pitch, rsqrtf((float)hd), lane);
    load_head<T>(base + D, 3 * D, k, N, hd, pitch, 1.0f, lane);
    load_head<T>(base + 2 * D, 3 * D, v, N, hd, pitch, 1.0f, lane);
    __builtin_amdgcn_wave_barrier();
    scores_softmax(q, k, P, N, hd, pitch, lane);
    for (int e = lane; e < N * hd; e += 64) {
        const int i = e / hd, d = e - i * hd;
        float acc = 0.f;
        for (int j = 0; j < N; ++j) acc = fmaf(P[i * N + j], v[j * pitch + d], acc);
        out[((int64_t)b * N + i) * D + h * hd + d] = from_f32<T>(acc);
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
    const int pitch = hd + 1, D = H * hd;
    const float scale = rsqrtf((float)hd);
    float *q = lds + (size_t)wave * per_wave_floats, *k = q + N * pitch, *v = k + N * pitch, *dO = v + N * pitch;
    float *P = dO + N * pitch, *dS = P + N * N;
    const T *base = qkv + (int64_t)b * N * 3 * D + h * hd;
    load_head<T>(base, 3 * D, q, N, hd, pitch, scale, lane);
    load_head<T>(base + D, 3 * D, k, N, hd, pitch, 1.0f, lane);
    load_head<T>(base + 2 * D, 3 * D, v, N, hd, pitch, 1.0f, lane);
    load_head<T>(dout + (int64_t)b * N * D + h * hd, D, dO, N, hd, pitch, 1.0f, lane);
    __builtin_amdgcn_wave_barrier();
    scores_softmax(q, k, P, N, hd, pitch, lane);
    // dP = dO v^T
    for (int e = lane; e < N * N; e += 64) {
        const int i = e / N, j = e - i * N;
        float acc = 0.f;
        for (int d = 0; d < hd; ++d) acc = fmaf(dO[i * pitch + d], v[j * pitch + d], acc);
        dS[e] = acc;
    }
    __builtin_amdgcn_wave_barrier();
    // dS = P * (dP - rowsum(P * dP))
    for (int i = lane; i < N; i += 64) {
        float dot = 0.f;
        for (int j = 0; j < N; ++j) dot = fmaf(P[i * N + j], dS[i * N + j], dot);
        for (int j = 0; j < N; ++j) dS[i * N + j] = P[i * N + j] * (dS[i * N + j] - dot);
    }
    __builtin_amdgcn_wave_barrier();
    T *dbase = dqkv + (int64_t)b * N * 3 * D + h * hd;
    for (int e = lane; e < N * hd; e += 64) {
        const int n = e / hd, d = e - n * hd;
        float aq = 0.f, ak = 0.f, av = 0.f;
        for (int j = 0; j < N; ++j) {
            aq = fmaf(dS[n * N + j], k[j * pitch + d], aq);   // dq[n] = scale * sum_j dS[n][j] k[j]
            ak = fmaf(dS[j * N + n], q[j * pitch + d], ak);   // dk[n] = sum_i dS[i][n] (scale q[i])
            av = fmaf(P[j * N + n], dO[j * pitch + d], av);   // dv[n] = sum_i P[i][n] dO[i]
        }
        const int64_t o = (int64_t)n * 3 * D + d;
        dbase[o] = from_f32<T>(aq * scale);
        dbase[o + D] = from_f32<T>(ak);
        dbase[o + 2 * D] = from_f32<T>(av);
    }
}

struct Plan {
    int waves, per_wave_floats;
    size_t smem;
};

Plan make_plan(int N, int hd, bool bwd) {
    Plan p;
    const int pitch = hd + 1;
    p.per_wave_floats = (bwd ? 4 : 3) * N * pitch + (bwd ? 2 : 1) * N * N;
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

extern "C" int skyemb_mha_fwd(const void *qkv, void *out, int dtype, int B, int N, int H, int hd, void *stream) {
    SKY_CHECK_ARG(B > 0 && N > 0 && H > 0 && hd > 0, "skyemb_mha_fwd: bad shape");
    const Plan p = make_plan(N, hd, false);
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((B * H + p.waves - 1) / p.waves), block(64 * p.waves);
    int rc;
    if (dtype == SKYEMB_BF16) {
        if ((rc = set_lds(mha_fwd_kernel<bf16_t>, p.smem, "skyemb_mha_fwd"))) return rc;
        hipLaunchKernelGGL(mha_fwd_kernel<bf16_t>, grid, block, p.smem, st, (const bf16_t *)qkv, (bf16_t *)out, B, N, H, hd,
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
    SKY_CHECK_ARG(B > 0 && N > 0 && H > 0 && hd > 0, "skyemb_mha_bwd: bad shape");
    const Plan p = make_plan(N, hd, true);
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((B * H + p.waves - 1) / p.waves), block(64 * p.waves);
    int rc;
    if (dtype == SKYEMB_BF16) {
        if ((rc = set_lds(mha_bwd_kernel<bf16_t>, p.smem, "skyemb_mha_bwd"))) return rc;
        hipLaunchKernelGGL(mha_bwd_kernel<bf16_t>, grid, block, p.smem, st, (const bf16_t *)qkv, (const bf16_t *)dout,
                           (bf16_t *)dqkv, B, N, H, hd, p.waves, p.per_wave_floats);
    } else {
        if ((rc = set_lds(mha_bwd_kernel<float>, p.smem, "skyemb_mha_bwd"))) return rc;
        hipLaunchKernelGGL(mha_bwd_kernel<float>, grid, block, p.smem, st, (const float *)qkv, (const float *)dout,
                           (float *)dqkv, B, N, H, hd, p.waves, p.per_wave_floats);
    }
    SKY_LAUNCH_CHECK("skyemb_mha_bwd");
    return 0;
}
