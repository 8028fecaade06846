// The 16-bit MFMA kernels (gemm.hip, gemm_pipe.hip + gemm_pipe256.h, attention_mfma.hip) are compiled TWICE from one source:
// once with lp_t = bf16 (SKYEMB_BF16) and once, under -DSKY_F16, with lp_t = IEEE half (SKYEMB_F16).  The two formats share
// every layout, LDS image, fragment read and wait count; what differs is the MFMA opcode and the fp32 <-> 16-bit conversion.
// Entry points of the -DSKY_F16 objects carry the suffix _f16 (SKY_TWIN); the bf16 objects' entry points hand SKYEMB_F16
// calls over to them.  Kernels and host state live in anonymous namespaces, so the twins share nothing.
#pragma once
#include "common.h"

#ifdef SKY_F16
typedef f16_t lp_t;
typedef f16x8 lp8;
typedef f16x4 lp4;
#define SKY_LP_DTYPE SKYEMB_F16
#define SKY_TWIN(name) name##_f16
#define SKY_TWIN_VIS __attribute__((visibility("hidden")))   // the _f16 entry points are not part of the C ABI
#define sky_mfma_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0)
#define sky_mfma_32x32x16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
#else
typedef bf16_t lp_t;
typedef bf16x8 lp8;
typedef bf16x4 lp4;
#define SKY_LP_DTYPE SKYEMB_BF16
#define SKY_TWIN(name) name
#define SKY_TWIN_VIS
#define sky_mfma_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)
#define sky_mfma_32x32x16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)
#endif

// ds_read_b64_tr_b16 (the transposing LDS read) moves 16-bit words whatever they encode: one builtin for both formats
typedef __attribute__((ext_vector_type(4))) short sky_i16x4;
__device__ __forceinline__ lp4 sky_ds_read_tr16_b64(__attribute__((address_space(3))) lp4 *p) {
    return __builtin_bit_cast(lp4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) sky_i16x4 *)p));
}
