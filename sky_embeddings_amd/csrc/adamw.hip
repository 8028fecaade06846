// Fused flat-buffer AdamW (torch.optim.AdamW single-tensor op order; utils/mim_vit.py:126-129,
// utils/pretrain_fns.py:36-41).  One launch updates every parameter: the model keeps all
// trainable tensors in ONE flat fp32 buffer laid out [decayed | not decayed], so the only
// per-element state is the position relative to n_decay.  HBM-bound: reads p,g,m,v, writes
// p,m,v (+ the bf16/fp32 shadow copy the GEMMs read, + optional gradient zeroing).
#include "common.h"
#include "adamw_math.h"

namespace {

template <typename T, bool LP, typename GT>
__global__ __launch_bounds__(256) void adamw_kernel(float *__restrict__ p, GT *__restrict__ g, float *__restrict__ m,
                                                    float *__restrict__ v, T *__restrict__ p_lp, int64_t n4,
                                                    int64_t n_decay, const float *__restrict__ hyper, float lr_arg,
                                                    float bc1_arg, float bc2_arg, float beta1, float beta2, float eps,
                                                    float wd, float grad_scale, int zero_grad) {
    // hyper (device) wins when given: kernel arguments are frozen inside a captured HIP graph
    const float lr = hyper ? hyper[0] : lr_arg, bc1 = hyper ? hyper[1] : bc1_arg, bc2 = hyper ? hyper[2] : bc2_arg;
    const SkyAdamScalars sc = sky_adam_scalars(lr, bc1, bc2, beta1, beta2, eps, wd, grad_scale);
    // two float4 groups per thread and iteration keep 8 loads in flight per lane (non-temporal loads / stores were tried:
    // no gain at 4.8 TB/s)
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i0 = (int64_t)blockIdx.x * 256 + threadIdx.x; i0 < n4; i0 += 2 * stride) {
        f32x4 pv[2], gv[2], mv[2], vv[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int64_t i = i0 + u * stride;
            if (i < n4) {
                pv[u] = *(const f32x4 *)(p + 4 * i);
                const float4 gl = load4<GT>(g + 4 * i);          // fp32 gradients, or the bf16 ones a bf16 all-reduce left
                gv[u] = (f32x4){gl.x, gl.y, gl.z, gl.w};
                mv[u] = *(const f32x4 *)(m + 4 * i);
                vv[u] = *(const f32x4 *)(v + 4 * i);
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int64_t i = i0 + u * stride;
            if (i >= n4) break;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                // (adamw_math.h: every rounding pinned -- a step applied range by range, or inside the weight-gradient GEMM's
                // epilogue, is bit-identical to one launch over the whole buffer)
                float pj = pv[u][j], mj = mv[u][j], vj = vv[u][j];
                sky_adamw_update(gv[u][j], pj, mj, vj, 4 * i + j < n_decay, sc);
                pv[u][j] = pj; mv[u][j] = mj; vv[u][j] = vj;
            }
            *(f32x4 *)(p + 4 * i) = pv[u];
            *(f32x4 *)(m + 4 * i) = mv[u];
            *(f32x4 *)(v + 4 * i) = vv[u];
            if (LP) store4<T>(p_lp + 4 * i, pv[u][0], pv[u][1], pv[u][2], pv[u][3]);
            if (zero_grad) store4<GT>(g + 4 * i, 0.f, 0.f, 0.f, 0.f);
        }
    }
}

__global__ void set_scalars_kernel(float *dst, float a, float b, float c, float d) {
    if (threadIdx.x == 0) { dst[0] = a; dst[1] = b; dst[2] = c; dst[3] = d; }
}

}  // namespace

// four floats into device memory by a kernel launch (values travel as kernel arguments): the step scalars of a graph-replayed
// optimiser step.  (A 16-byte host-to-device copy in front of every step put the copy engine's hand-over to the compute queue
// -- tens of microseconds -- on the step's critical path.)
extern "C" int skyemb_set_scalars(float *dst, float a, float b, float c, float d, void *stream) {
    SKY_CHECK_ARG(dst != nullptr, "skyemb_set_scalars: null destination");
    hipLaunchKernelGGL(set_scalars_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, dst, a, b, c, d);
    SKY_LAUNCH_CHECK("skyemb_set_scalars");
    return 0;
}

extern "C" int skyemb_adamw(float *p, void *g, float *m, float *v, void *p_lp, int dtype, int64_t n, int64_t n_decay,
                            const float *hyper, float lr, float bc1, float bc2, float beta1, float beta2, float eps,
                            float wd, float grad_scale, int zero_grad, int grad_dtype, void *stream) {
    SKY_CHECK_ARG(n > 0 && n % 4 == 0 && n_decay >= 0 && n_decay <= n, "skyemb_adamw: n must be a positive multiple of 4");
    SKY_CHECK_ARG(aligned16(p) && aligned16(m) && aligned16(v) && (((uintptr_t)g) & 7) == 0, "skyemb_adamw: unaligned buffers");
    SKY_CHECK_ARG(grad_dtype == SKYEMB_F32 || sky_is_lp(grad_dtype), "skyemb_adamw: bad grad_dtype %d", grad_dtype);
    SKY_CHECK_ARG(!sky_is_lp(grad_dtype) || !p_lp || grad_dtype == dtype || dtype == SKYEMB_F32,
                  "skyemb_adamw: a 16-bit gradient buffer and a 16-bit shadow must share their format");
    int64_t blocks = ceil_div64(n / 4, 256);
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((unsigned)blocks), block(256);
#define ADAMW_LAUNCH(T, LP, GT, lp_ptr)                                                                                        \
    hipLaunchKernelGGL((adamw_kernel<T, LP, GT>), grid, block, 0, st, p, (GT *)g, m, v, lp_ptr, n / 4, n_decay, hyper, lr, bc1, \
                       bc2, beta1, beta2, eps, wd, grad_scale, zero_grad)
    const bool g16 = grad_dtype == SKYEMB_BF16, gh = grad_dtype == SKYEMB_F16;
    if (!p_lp) {
        if (g16) ADAMW_LAUNCH(float, false, bf16_t, (float *)nullptr);
        else if (gh) ADAMW_LAUNCH(float, false, f16_t, (float *)nullptr);
        else ADAMW_LAUNCH(float, false, float, (float *)nullptr);
    } else if (dtype == SKYEMB_BF16) {
        if (g16) ADAMW_LAUNCH(bf16_t, true, bf16_t, (bf16_t *)p_lp);
        else ADAMW_LAUNCH(bf16_t, true, float, (bf16_t *)p_lp);
    } else if (dtype == SKYEMB_F16) {
        if (gh) ADAMW_LAUNCH(f16_t, true, f16_t, (f16_t *)p_lp);
        else ADAMW_LAUNCH(f16_t, true, float, (f16_t *)p_lp);
    } else {
        if (g16) ADAMW_LAUNCH(float, true, bf16_t, (float *)p_lp);
        else if (gh) ADAMW_LAUNCH(float, true, f16_t, (float *)p_lp);
        else ADAMW_LAUNCH(float, true, float, (float *)p_lp);
    }
#undef ADAMW_LAUNCH
    SKY_LAUNCH_CHECK("skyemb_adamw");
    return 0;
}
