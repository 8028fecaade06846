// Shared device/host helpers for libskyemb (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/skyemb.h"

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef _Float16 f16_t;     // SKYEMB_F16: the second 16-bit operand format (11-bit significand, same MFMA rate as bf16)
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

#define SKYEMB_WAVE 64
// most four-wave blocks of a LayerNorm backward launch (skyemb_layernorm_bwd_blocks / sky_ln_bwd_blocks: every block the same
// number of rows); -DSKY_LN_BWD_CAP=... builds the experiment variants (tools/build_variant.sh)
#ifndef SKY_LN_BWD_CAP
#define SKY_LN_BWD_CAP 576
#endif

void skyemb_set_error(const char *fmt, ...);
// measurement aid (skyemb_debug_skip): compiled into libskyemb_measure.so only (-DSKYEMB_MEASURE, csrc/Makefile); the product
// library has no switch that turns launches into no-ops
#ifdef SKYEMB_MEASURE
int skyemb_skip_mask(void);   // api.cpp
#else
static inline int skyemb_skip_mask(void) { return 0; }
#endif
void skyemb_count_gemm(int slot);   // api.cpp: diagnostic launch counters, see skyemb_gemm_launch_counts

// Also drops any stale sticky HIP error left by other code in this thread (e.g. a device probe),
// so that SKY_LAUNCH_CHECK reports only this call's launches.  Every launching entry point starts
// with a SKY_CHECK_ARG.
#define SKY_CHECK_ARG(cond, ...)            \
    do {                                    \
        (void)hipGetLastError();            \
        if (!(cond)) {                      \
            skyemb_set_error(__VA_ARGS__);  \
            return 1;                       \
        }                                   \
    } while (0)

#define SKY_LAUNCH_CHECK(name)                                                        \
    do {                                                                              \
        hipError_t e_ = hipGetLastError();                                            \
        if (e_ != hipSuccess) {                                                       \
            skyemb_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));   \
            return 2;                                                                 \
        }                                                                             \
    } while (0)

template <typename T>
__device__ __forceinline__ float to_f32(T v);
template <>
__device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <>
__device__ __forceinline__ float to_f32<bf16_t>(bf16_t v) { return (float)v; }

template <>
__device__ __forceinline__ float to_f32<f16_t>(f16_t v) { return (float)v; }

template <typename T>
__device__ __forceinline__ T from_f32(float v);
template <>
__device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <>
__device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return (bf16_t)v; }  // v_cvt_pk_bf16_f32: RNE, NaN stays NaN

template <>
__device__ __forceinline__ f16_t from_f32<f16_t>(float v) { return (f16_t)v; }  // v_cvt_f16_f32: RNE, overflow -> inf

template <typename T>
__device__ __forceinline__ void store4(T *p, float a, float b, float c, float d);
template <>
__device__ __forceinline__ void store4<float>(float *p, float a, float b, float c, float d) {
    *(float4 *)p = make_float4(a, b, c, d);
}
template <>
__device__ __forceinline__ void store4<bf16_t>(bf16_t *p, float a, float b, float c, float d) {
    bf16x4 v;
    v[0] = (bf16_t)a; v[1] = (bf16_t)b; v[2] = (bf16_t)c; v[3] = (bf16_t)d;
    *(bf16x4 *)p = v;
}
template <>
__device__ __forceinline__ void store4<f16_t>(f16_t *p, float a, float b, float c, float d) {
    f16x4 v;
    v[0] = (f16_t)a; v[1] = (f16_t)b; v[2] = (f16_t)c; v[3] = (f16_t)d;
    *(f16x4 *)p = v;
}
template <typename T>
__device__ __forceinline__ float4 load4(const T *p);
template <>
__device__ __forceinline__ float4 load4<float>(const float *p) { return *(const float4 *)p; }
template <>
__device__ __forceinline__ float4 load4<bf16_t>(const bf16_t *p) {
    bf16x4 v = *(const bf16x4 *)p;
    return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
}

template <>
__device__ __forceinline__ float4 load4<f16_t>(const f16_t *p) {
    f16x4 v = *(const f16x4 *)p;
    return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
}
// is `dtype` one of the two 16-bit operand formats?
static inline bool sky_is_lp(int dtype) { return dtype == SKYEMB_BF16 || dtype == SKYEMB_F16; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline bool aligned16(const void *p) { return (((uintptr_t)p) & 15) == 0; }

// One step of an 8-bit radix select: hist[0..255] holds the bin counts of the keys still in play; finds the bin of the
// kth largest (largest b with sum_{c >= b} hist[c] >= kth; kth <= total) and leaves hist[256] = b, hist[257] = the rank
// still wanted inside that bin.  Run by the first wave of the workgroup (tid < 64), four bins per lane, suffix sums by
// shuffles -- a serial walk over the 256 bins by one thread dominated the select kernels.  Caller: __syncthreads() around it.
__device__ __forceinline__ void radix_pick_bin(int *hist, int kth, int tid) {
    if (tid >= 64) return;
    const int h0 = hist[4 * tid], h1 = hist[4 * tid + 1], h2 = hist[4 * tid + 2], h3 = hist[4 * tid + 3];
    int suf = h0 + h1 + h2 + h3;                       // -> inclusive suffix sum over lanes tid..63
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int up = __shfl_down(suf, o, 64);
        if (tid + o < 64) suf += up;
    }
    const unsigned long long reach = __ballot(suf >= kth);    // lanes 0..L reach kth (suf is non-increasing in the lane)
    const int L = reach ? 63 - __builtin_clzll(reach) : 0;
    if (tid == L) {
        int run = suf - (h0 + h1 + h2 + h3);                   // keys in the bins above this lane's four
        int b = 4 * tid + 3;
        const int hh[4] = {h0, h1, h2, h3};
#pragma unroll
        for (int j = 3; j > 0; --j) {
            if (run + hh[j] >= kth) break;
            run += hh[j];
            --b;
        }
        hist[256] = b;
        hist[257] = kth - run;
    }
}
