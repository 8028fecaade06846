// One element of torch.optim.AdamW (single-tensor op order; utils/mim_vit.py:126-129, utils/pretrain_fns.py:36-41) with every
// rounding pinned (explicit fmaf, no compiler contraction): the update of an element must not depend on which kernel, launch or
// unroll slot it lands in -- the flat-buffer kernel (adamw.hip), its range-by-range use and the update fused into the
// weight-gradient GEMM's epilogue (gemm_pipe.hip) give the same bits.
#pragma once
#include <hip/hip_runtime.h>

struct SkyAdamScalars {
    float step_size;   // lr / (1 - beta1^t)
    float bc2_sqrt;    // sqrt(1 - beta2^t)
    float decay;       // 1 - lr * weight_decay
    float beta1, beta2, eps, grad_scale;
};

__device__ __forceinline__ SkyAdamScalars sky_adam_scalars(float lr, float bc1, float bc2, float beta1, float beta2, float eps, float wd,
                                                           float grad_scale) {
    SkyAdamScalars s;
    s.step_size = lr / bc1;
    s.bc2_sqrt = sqrtf(bc2);
    s.decay = 1.0f - lr * wd;
    s.beta1 = beta1; s.beta2 = beta2; s.eps = eps; s.grad_scale = grad_scale;
    return s;
}

__device__ __forceinline__ void sky_adamw_update(float g, float &p, float &m, float &v, bool decayed, const SkyAdamScalars &s) {
#pragma clang fp contract(off)
    const float gj = g * s.grad_scale;
    float pj = p;
    if (decayed) pj *= s.decay;                                         // p.mul_(1 - lr*wd)
    const float mj = fmaf(m, s.beta1, gj * (1.0f - s.beta1));           // exp_avg.lerp_(grad, 1-beta1)
    const float vj = fmaf(v, s.beta2, (gj * gj) * (1.0f - s.beta2));    // exp_avg_sq.mul_().addcmul_()
    const float denom = sqrtf(vj) / s.bc2_sqrt + s.eps;
    pj = fmaf(-s.step_size, mj / denom, pj);                            // p.addcdiv_(m, denom, -step_size)
    p = pj; m = mj; v = vj;
}
