// LayerNorm backward of token rows, shared by the stand-alone kernel (layernorm.hip) and the side job that rides in the grouped
// weight-gradient launches (gemm_pipe.hip: side_ln_bwd_job).  One wave per row, four waves = one "block" of the partial-sum
// table: block `blk` of `nblk` takes rows blk * 4 + w, + nblk * 4, ... (w = wave in the block) and leaves the column sums of
// dy * xhat / dy over its rows in part[0 / 1][blk][D] -- the same rows, the same summation order and the same table whichever
// launch runs it, so the two forms give the same bits.  `red` = 4 x 256 floats of LDS owned by this block of four waves;
// every wave of the WORKGROUP must call (the partial sums pass through __syncthreads()).
#pragma once
#include "common.h"

// GAMMA_LDS: gamma is read from `gam_lds` ([D] floats in LDS, filled by the caller) where it is used instead of being held in
// NV float4 registers per lane (the side job: 12 registers that decide whether two 128 x 128 workgroups still share a CU).
template <typename TD, typename T, int NV, bool GAMMA_LDS = false>
__device__ __forceinline__ void sky_ln_bwd_rows(const TD *__restrict__ dy, const float *__restrict__ x, const float *__restrict__ gamma,
                                                const float *__restrict__ mean, const float *__restrict__ rstd, const float *g_in,
                                                float *g_out, T *__restrict__ g_lp, float *__restrict__ part, const int M, const int D,
                                                const int nblk, const int blk, const int wave, const int lane, float *red,
                                                const float *gam_lds = nullptr) {
    // every rounding pinned (no compiler contraction of a * b + c into an fma): the stand-alone kernel and the side job are two
    // instantiations of this text with different surroundings, and hipcc's default -ffp-contract=fast fused them differently at
    // D = 192 (gradients apart in the last bit); with contraction off both are the same sequence of multiplies and adds
#pragma clang fp contract(off)
    const int nv = D >> 2;
    float4 dgam[NV], dbet[NV], gam_r[GAMMA_LDS ? 1 : NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        dgam[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        dbet[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        const int c = lane + i * 64;
        if constexpr (!GAMMA_LDS) gam_r[i] = *(const float4 *)(gamma + 4 * (c < nv ? c : nv - 1));      // (columns past the row are never used)
    }
    auto gam_of = [&](int i) -> float4 {
        if constexpr (GAMMA_LDS) {
            const int c = lane + i * 64;
            return *(const float4 *)(gam_lds + 4 * (c < nv ? c : nv - 1));
        } else {
            return gam_r[i];
        }
    };
    const float *gsrc = g_in ? g_in : x;                      // (no incoming gradient: a harmless second read of x, selected away)
    const bool live = blk < nblk;                             // (a side workgroup's last block of four waves may be past the table)
    for (int row = blk * 4 + wave; live && row < M; row += nblk * 4) {
        const float mu = mean[row], rs = rstd[row];
        float4 xh[NV], dyv[NV], gi[NV];
        float s1 = 0.f, s2 = 0.f;
        // the row of x, of dy and of the incoming residual gradient: all requested together, unconditionally, at clamped columns
        // (loads under `if (c < nv)` were waited for one by one: 3 NV dependent memory round trips per row)
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane + i * 64, cc = c < nv ? c : nv - 1;
            xh[i] = *(const float4 *)(x + (int64_t)row * D + 4 * cc);
            dyv[i] = load4<TD>(dy + (int64_t)row * D + 4 * cc);
            gi[i] = *(const float4 *)(gsrc + (int64_t)row * D + 4 * cc);
        }
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane + i * 64;
            if (c < nv) {
                const float4 xv = xh[i];
                const float4 gm = gam_of(i);
                if (!g_in) gi[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                xh[i] = make_float4((xv.x - mu) * rs, (xv.y - mu) * rs, (xv.z - mu) * rs, (xv.w - mu) * rs);
                const float a0 = dyv[i].x * gm.x, a1 = dyv[i].y * gm.y, a2 = dyv[i].z * gm.z, a3 = dyv[i].w * gm.w;
                s1 += (a0 + a1) + (a2 + a3);
                s2 += (a0 * xh[i].x + a1 * xh[i].y) + (a2 * xh[i].z + a3 * xh[i].w);
                dgam[i].x += dyv[i].x * xh[i].x; dgam[i].y += dyv[i].y * xh[i].y;
                dgam[i].z += dyv[i].z * xh[i].z; dgam[i].w += dyv[i].w * xh[i].w;
                dbet[i].x += dyv[i].x; dbet[i].y += dyv[i].y; dbet[i].z += dyv[i].z; dbet[i].w += dyv[i].w;
            }
        }
        const float m1 = wave_sum(s1) / (float)D, m2 = wave_sum(s2) / (float)D;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane + i * 64;
            if (c < nv) {
                const float4 gm = gam_of(i);
                float4 o;
                o.x = rs * (dyv[i].x * gm.x - m1 - xh[i].x * m2);
                o.y = rs * (dyv[i].y * gm.y - m1 - xh[i].y * m2);
                o.z = rs * (dyv[i].z * gm.z - m1 - xh[i].z * m2);
                o.w = rs * (dyv[i].w * gm.w - m1 - xh[i].w * m2);
                const int64_t off = (int64_t)row * D + 4 * c;
                if (g_in) {
                    o.x += gi[i].x; o.y += gi[i].y; o.z += gi[i].z; o.w += gi[i].w;
                }
                *(float4 *)(g_out + off) = o;
                if (g_lp) store4<T>(g_lp + off, o.x, o.y, o.z, o.w);
            }
        }
    }
    // reduce the 4 waves' column partials through LDS, one float4 slot at a time
    float *pg = part + (int64_t)blk * D;
    float *pb = part + (int64_t)nblk * D + (int64_t)blk * D;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + i * 64;
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            const float4 val = pass == 0 ? dgam[i] : dbet[i];
            __syncthreads();
            *(float4 *)&red[wave * 256 + lane * 4] = val;
            __syncthreads();
            if (wave == 0 && c < nv && live) {
                float4 a = *(float4 *)&red[lane * 4];
#pragma unroll
                for (int w = 1; w < 4; ++w) {
                    const float4 b = *(float4 *)&red[w * 256 + lane * 4];
                    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
                }
                *(float4 *)((pass == 0 ? pg : pb) + 4 * c) = a;
            }
        }
    }
}
