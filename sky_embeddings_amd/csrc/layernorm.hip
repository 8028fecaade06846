// LayerNorm forward / backward (fp32 statistics, wavefront-shuffle reductions).
// One 64-lane wave per token row; a row of D <= 2048 floats lives in registers as float4s.
#include "common.h"
#include "ln_bwd_body.h"

namespace {

constexpr int MAXV = 8;  // float4 per lane -> D <= 2048 (kernels are instantiated per NV = ceil(D/256))

template <typename T, int NV>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float *__restrict__ x, const float *__restrict__ gamma,
                                                     const float *__restrict__ beta, T *__restrict__ y,
                                                     float *__restrict__ y32, float *__restrict__ mean_o,
                                                     float *__restrict__ rstd_o, int M, int D, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int nv = D >> 2;
    const float *xr = x + (int64_t)row * D;
    // Every load of the row (and of gamma / beta) is issued up front, unconditionally, at a clamped column: a load under
    // `if (c < nv)` makes the compiler branch around it and wait for it on the spot -- the kernel was six dependent memory round
    // trips (three for the row, three for gamma / beta) for a 768-wide row; now one.
    float4 v[NV], gm[NV], bt[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + i * 64, cc = c < nv ? c : nv - 1;
        v[i] = *(const float4 *)(xr + 4 * cc);
        gm[i] = *(const float4 *)(gamma + 4 * cc);
        bt[i] = *(const float4 *)(beta + 4 * cc);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
        if (lane + i * 64 < nv) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        if (lane + i * 64 < nv) {
            const float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
            q += (a * a + b * b) + (cc * cc + d * d);
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
    if (lane == 0) {
        mean_o[row] = mean;
        rstd_o[row] = rstd;
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + i * 64;
        if (c < nv) {
            const float4 g = gm[i], b = bt[i];
            const float o0 = (v[i].x - mean) * rstd * g.x + b.x, o1 = (v[i].y - mean) * rstd * g.y + b.y;
            const float o2 = (v[i].z - mean) * rstd * g.z + b.z, o3 = (v[i].w - mean) * rstd * g.w + b.w;
            if (y) store4<T>(y + (int64_t)row * D + 4 * c, o0, o1, o2, o3);
            if (y32) *(float4 *)(y32 + (int64_t)row * D + 4 * c) = make_float4(o0, o1, o2, o3);
        }
    }
}

// dy element type TD (T or float), low-precision copy type T.
template <typename TD, typename T, int NV>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const TD *__restrict__ dy, const float *__restrict__ x,
                                                     const float *__restrict__ gamma, const float *__restrict__ mean,
                                                     const float *__restrict__ rstd, const float *g_in, float *g_out,
                                                     T *__restrict__ g_lp, float *__restrict__ part, int M, int D,
                                                     int nblk) {
    __shared__ float red[4 * 256];
    // (the rows and the partial sums: ln_bwd_body.h -- shared with the side job of the grouped weight-gradient launches)
    sky_ln_bwd_rows<TD, T, NV>(dy, x, gamma, mean, rstd, g_in, g_out, g_lp, part, M, D, nblk, (int)blockIdx.x, (int)(threadIdx.x >> 6),
                               (int)(threadIdx.x & 63), red);
}

// second stage: out[which][d] = sum_b part[which][b][d]; block = 32 columns x 32 row groups, grid = (D/32, 2)
__global__ __launch_bounds__(1024) void ln_bwd_reduce_kernel(const float *__restrict__ part, float *__restrict__ dgamma,
                                                             float *__restrict__ dbeta, int nblk, int D) {
    __shared__ float red[32][33];
    const int col = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const int d = blockIdx.x * 32 + col;
    const float *src = part + (int64_t)blockIdx.y * nblk * D;
    float acc = 0.f;
    if (d < D)
#pragma unroll 8
        for (int b = rg; b < nblk; b += 32) acc += src[(int64_t)b * D + d];
    red[rg][col] = acc;
    __syncthreads();
    if (rg == 0 && d < D) {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 32; ++r) t += red[r][col];
        (blockIdx.y == 0 ? dgamma : dbeta)[d] = t;
    }
}

// every column reduce of a backward stage in one launch.  blocks[b] = {item, x | y << 16}: workgroup b takes columns [32 x, +32) of
// half y (0 dgamma, 1 dbeta) of items[item] -- a flat list, so that items of different widths (LayerNorm: dim; bias-gradient
// partial sums: up to 4 dim) share a launch without empty workgroups.
__global__ __launch_bounds__(1024) void ln_bwd_reduce_batch_kernel(const skyemb_ln_reduce_item *__restrict__ items,
                                                                   const int2 *__restrict__ blocks) {
    __shared__ float red[32][33];
    const int2 blk = blocks[blockIdx.x];
    const skyemb_ln_reduce_item it = items[blk.x];
    const int bx = blk.y & 0xffff, by = blk.y >> 16;
    const int col = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const int d = bx * 32 + col;
    const float *src = it.part + (int64_t)by * it.nblk * it.D;
    float acc = 0.f;
    if (d < it.D) {
#pragma unroll 8
        for (int b = rg; b < it.nblk; b += 32) acc += src[(int64_t)b * it.D + d];   // (loads in flight; same summation order)
    }
    red[rg][col] = acc;
    __syncthreads();
    if (rg == 0 && d < it.D) {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 32; ++r) t += red[r][col];
        (by == 0 ? it.dgamma : it.dbeta)[d] = t;
    }
}

}  // namespace

extern "C" int skyemb_layernorm_bwd_reduce_batch(const skyemb_ln_reduce_item *items, const int32_t *blocks, int n_blocks, void *stream) {
    SKY_CHECK_ARG(items && blocks && n_blocks > 0, "skyemb_layernorm_bwd_reduce_batch: bad arguments");
    hipLaunchKernelGGL(ln_bwd_reduce_batch_kernel, dim3((unsigned)n_blocks), dim3(1024), 0, (hipStream_t)stream, items, (const int2 *)blocks);
    SKY_LAUNCH_CHECK("skyemb_layernorm_bwd_reduce_batch");
    return 0;
}

extern "C" int skyemb_layernorm_fwd(const float *x, const float *gamma, const float *beta, void *y, float *y32,
                                    int dtype, float *mean, float *rstd, int M, int D, float eps, void *stream) {
    SKY_CHECK_ARG(M > 0 && D > 0 && D % 4 == 0 && D <= MAXV * 256, "skyemb_layernorm_fwd: bad shape M=%d D=%d", M, D);
    SKY_CHECK_ARG(aligned16(x) && aligned16(gamma) && aligned16(beta), "skyemb_layernorm_fwd: unaligned input");
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((M + 3) / 4), block(256);
#define LN_FWD(NV)                                                                                                   \
    if (dtype == SKYEMB_BF16)                                                                                        \
        hipLaunchKernelGGL((ln_fwd_kernel<bf16_t, NV>), grid, block, 0, st, x, gamma, beta, (bf16_t *)y, y32, mean,   \
                           rstd, M, D, eps);                                                                         \
    else if (dtype == SKYEMB_F16)                                                                                    \
        hipLaunchKernelGGL((ln_fwd_kernel<f16_t, NV>), grid, block, 0, st, x, gamma, beta, (f16_t *)y, y32, mean,     \
                           rstd, M, D, eps);                                                                         \
    else                                                                                                             \
        hipLaunchKernelGGL((ln_fwd_kernel<float, NV>), grid, block, 0, st, x, gamma, beta, (float *)y, y32, mean,     \
                           rstd, M, D, eps);
    switch ((D + 255) / 256) {
        case 1: LN_FWD(1) break;
        case 2: LN_FWD(2) break;
        case 3: LN_FWD(3) break;
        case 4: LN_FWD(4) break;
        case 5: LN_FWD(5) break;
        case 6: LN_FWD(6) break;
        case 7: LN_FWD(7) break;
        default: LN_FWD(8) break;
    }
#undef LN_FWD
    SKY_LAUNCH_CHECK("skyemb_layernorm_fwd");
    return 0;
}

extern "C" int skyemb_layernorm_bwd_blocks(int M) {
    // one row per wave and pass; about two waves per SIMD, and every wave the SAME number of rows (4352 decoder rows over 512
    // workgroups left a quarter of the waves with a third row: 544 workgroups x 2 rows; 8320 ViT-L rows: 520 x 4)
    int nb = (M + 3) / 4;
    if (nb < 1) nb = 1;
    const int rounds = (nb + SKY_LN_BWD_CAP - 1) / SKY_LN_BWD_CAP;
    return (nb + rounds - 1) / rounds;
}

extern "C" int skyemb_layernorm_bwd(const void *dy, int dy_is_f32, int dtype, const float *x, const float *gamma,
                                    const float *mean, const float *rstd, const float *g_in, float *g_out, void *g_lp,
                                    float *part, float *dgamma, float *dbeta, int M, int D, void *stream) {
    SKY_CHECK_ARG(M > 0 && D > 0 && D % 4 == 0 && D <= MAXV * 256, "skyemb_layernorm_bwd: bad shape M=%d D=%d", M, D);
    hipStream_t st = (hipStream_t)stream;
    const int nblk = skyemb_layernorm_bwd_blocks(M);
    dim3 grid(nblk), block(256);
#define LN_BWD(NV)                                                                                                   \
    if (dtype == SKYEMB_BF16) {                                                                                      \
        if (dy_is_f32)                                                                                               \
            hipLaunchKernelGGL((ln_bwd_kernel<float, bf16_t, NV>), grid, block, 0, st, (const float *)dy, x, gamma,   \
                               mean, rstd, g_in, g_out, (bf16_t *)g_lp, part, M, D, nblk);                           \
        else                                                                                                         \
            hipLaunchKernelGGL((ln_bwd_kernel<bf16_t, bf16_t, NV>), grid, block, 0, st, (const bf16_t *)dy, x, gamma, \
                               mean, rstd, g_in, g_out, (bf16_t *)g_lp, part, M, D, nblk);                           \
    } else if (dtype == SKYEMB_F16) {                                                                                \
        if (dy_is_f32)                                                                                               \
            hipLaunchKernelGGL((ln_bwd_kernel<float, f16_t, NV>), grid, block, 0, st, (const float *)dy, x, gamma,    \
                               mean, rstd, g_in, g_out, (f16_t *)g_lp, part, M, D, nblk);                            \
        else                                                                                                         \
            hipLaunchKernelGGL((ln_bwd_kernel<f16_t, f16_t, NV>), grid, block, 0, st, (const f16_t *)dy, x, gamma,    \
                               mean, rstd, g_in, g_out, (f16_t *)g_lp, part, M, D, nblk);                            \
    } else {                                                                                                         \
        hipLaunchKernelGGL((ln_bwd_kernel<float, float, NV>), grid, block, 0, st, (const float *)dy, x, gamma, mean,  \
                           rstd, g_in, g_out, (float *)g_lp, part, M, D, nblk);                                      \
    }
    switch ((D + 255) / 256) {
        case 1: LN_BWD(1) break;
        case 2: LN_BWD(2) break;
        case 3: LN_BWD(3) break;
        case 4: LN_BWD(4) break;
        case 5: LN_BWD(5) break;
        case 6: LN_BWD(6) break;
        case 7: LN_BWD(7) break;
        default: LN_BWD(8) break;
    }
#undef LN_BWD
    if (dgamma && dbeta)
        hipLaunchKernelGGL(ln_bwd_reduce_kernel, dim3((D + 31) / 32, 2), dim3(1024), 0, st, part, dgamma, dbeta, nblk, D);
    SKY_LAUNCH_CHECK("skyemb_layernorm_bwd");
    return 0;
}
