// Bank-streaming cosine top-k for Q <= 16 queries (the HBM-bound regime, e.g. the reference's one
// target vector): every wavefront streams its own slice of the bank straight from HBM into
// registers (16-byte loads, no LDS staging, no barriers), feeds v_mfma_f32_16x16x4_f32 and keeps
// private sorted top-k lists for the 16 queries in LDS.  Bit-identical to the tiled kernel and to
// oracle/topk_oracle.c: the fma chain runs over d = 0,1,2,... in order.
//
// Lane (n = lane&15, g = lane>>4) loads bank[row0+n][16c + 4g .. 4g+3]; the MFMA B operand of k-step m
// must hold bank[row0+n][16c + 4m + g], i.e. the 4x4 transpose of (lane group g) x (element s):
// two v_permlane32_swap (lanes +-32) and two v_permlane16_swap (lanes +-16) per float4.
#include "common.h"

namespace {

constexpr int UNROLL = 4;   // float4 loads per register set (2 sets: up to 8 KiB per wave in flight)

__device__ __forceinline__ float finish_score(float dot, float qn, float xn, float eps) {
    const float den = fmaf(qn, xn, eps);
    const float s = __fdiv_rn(dot, den);
    return s == s ? s : -INFINITY;
}

__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ void swap32(float &lo_half_src, float &hi_half_dst) {
    // lanes 32-63 of `hi_half_dst` <-> lanes 0-31 of `lo_half_src`
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(hi_half_dst), __float_as_uint(lo_half_src), false, false);
    hi_half_dst = __uint_as_float(r[0]);
    lo_half_src = __uint_as_float(r[1]);
}
__device__ __forceinline__ void swap16(float &even_src, float &odd_dst) {
    // odd 16-lane rows of `odd_dst` <-> even 16-lane rows of `even_src`
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(odd_dst), __float_as_uint(even_src), false, false);
    odd_dst = __uint_as_float(r[0]);
    even_src = __uint_as_float(r[1]);
}

// v[s] of lane group g  ->  v[g'] ... transpose so that afterwards v[m] (group g) == old v[g] of group m
__device__ __forceinline__ void transpose4(float4 &v) {
    swap32(v.z, v.x);   // upper half's x <-> lower half's z
    swap32(v.w, v.y);   // upper half's y <-> lower half's w
    swap16(v.y, v.x);   // odd rows' x <-> even rows' y
    swap16(v.w, v.z);   // odd rows' z <-> even rows' w
}

// A operand image in LDS: entry [c][l] (lane l = (q = l & 15, g = l >> 4)) holds tw[q][16c + 4m + g], m = 0..3.  Every load of
// a thread's (up to four) entries is issued before the first LDS write: a load that feeds an LDS store straight away is waited
// for on the spot, and the image of 16 x 768 queries was twelve dependent memory round trips per workgroup.
template <int WAVES>
__device__ __forceinline__ void build_imgA(float4 *imgA, const float *__restrict__ tw, int Q, int D, int nchunk, int tid) {
    const int total = nchunk * 64;
    for (int e0 = tid; e0 < total; e0 += 4 * WAVES * 64) {
        float4 a[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = e0 + u * WAVES * 64 < total ? e0 + u * WAVES * 64 : total - 1;
            const int c = e >> 6, l = e & 63, q = l & 15, g = l >> 4;
            const float *src = tw + (int64_t)(q < Q ? q : 0) * D + 16 * c + g;
            a[u] = make_float4(src[0], src[4], src[8], src[12]);
            if (q >= Q) a[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = e0 + u * WAVES * 64;
            if (e < total) imgA[e] = a[u];
        }
    }
}

// dot products of 16 queries (A image in LDS) with the 16 bank rows this wave's lanes point at (`src` = row + 4 g): the
// contract's fp32 fma chain over d = 0, 1, 2, ... on v_mfma_f32_16x16x4_f32.  Two register sets (nchunk % UNROLL == 0 is
// checked on the host): the next group's loads are in flight while the current group feeds the MFMAs.
__device__ __forceinline__ f32x4 stream_dot16(const float *__restrict__ src, const float4 *__restrict__ imgA, int nchunk, int lane) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    float4 b0[UNROLL], b1[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) b0[u] = *(const float4 *)(src + 16 * u);
    for (int c0 = 0; c0 < nchunk; c0 += 2 * UNROLL) {
        const bool more1 = c0 + UNROLL < nchunk, more2 = c0 + 2 * UNROLL < nchunk;
        if (more1) {
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) b1[u] = *(const float4 *)(src + 16 * (c0 + UNROLL + u));
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            transpose4(b0[u]);
            const float4 a = imgA[(c0 + u) * 64 + lane];
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b0[u].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b0[u].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b0[u].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b0[u].w, acc, 0, 0, 0);
        }
        if (more2) {
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) b0[u] = *(const float4 *)(src + 16 * (c0 + 2 * UNROLL + u));
        }
        if (more1) {
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                transpose4(b1[u]);
                const float4 a = imgA[(c0 + UNROLL + u) * 64 + lane];
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b1[u].x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b1[u].y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b1[u].z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b1[u].w, acc, 0, 0, 0);
            }
        }
    }
    return acc;
}

template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void cosine_topk_stream_kernel(const float *__restrict__ tw, const float *__restrict__ qn,
                                                                 const float *__restrict__ bank, const float *__restrict__ xn,
                                                                 int Q, int64_t N, int D, int k, float eps, int64_t idx_offset,
                                                                 int64_t rows_per_wave, float *__restrict__ part_s,
                                                                 int64_t *__restrict__ part_i, const float *__restrict__ thr0) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nchunk = D >> 4;
    float4 *imgA = (float4 *)lds;                                   // [nchunk][64]: A fragments of every k-step
    float *ls_all = lds + (size_t)nchunk * 64 * 4;                  // [WAVES][16][k]
    int *li_all = (int *)(ls_all + (size_t)WAVES * 16 * k);        // [WAVES][16][k]
    // A operand image: lane (q = l&15, g = l>>4), element m: tw[q][16c + 4m + g]
    build_imgA<WAVES>(imgA, tw, Q, D, nchunk, tid);
    __syncthreads();
    float *ls = ls_all + (size_t)wave * 16 * k;
    int *li = li_all + (size_t)wave * 16 * k;
    const int wid = blockIdx.x * WAVES + wave;
    const int64_t r_begin = (int64_t)wid * rows_per_wave;
    int64_t r_end = r_begin + rows_per_wave;
    if (r_end > N) r_end = N;
    const int n_lane = lane & 15, g = lane >> 4;
    // per-query list sizes / thresholds live in registers of ALL lanes (wave-uniform arrays of 16)
    int n_in[16];
    float thr[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) { n_in[q] = 0; thr[q] = (thr0 && q < Q) ? thr0[q] : -INFINITY; }
    float floor_thr[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) floor_thr[q] = thr[q];   // valid lower bound of the global k-th best (or -inf)
    float qn4[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) qn4[r] = (4 * g + r) < Q ? qn[4 * g + r] : 0.f;

    for (int64_t n0 = r_begin; n0 < r_end; n0 += 16) {
        int64_t row = n0 + n_lane;
        const bool row_ok = row < r_end;
        if (!row_ok) row = r_end - 1;                                // clamp: masked below
        const float *src = bank + row * D + 4 * g;
        const f32x4 acc = stream_dot16(src, imgA, nchunk, lane);
        // C/D: col = lane&15 -> bank row n0 + n_lane, row = 4g + r -> query
        const float xnv = xn[row];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int q_mine = 4 * g + r;
            const float s = (row_ok && q_mine < Q) ? finish_score(acc[r], qn4[r], xnv, eps) : -INFINITY;
            // candidates of the 4 queries {r, 4+r, 8+r, 12+r} (one per lane group), rows ascending within a group
            float my_thr = -INFINITY;
#pragma unroll
            for (int gg = 0; gg < 4; ++gg) my_thr = (g == gg) ? thr[4 * gg + r] : my_thr;
            unsigned long long m = __ballot(s > my_thr);
            while (m) {
                const int srcl = __builtin_ctzll(m);
                m &= m - 1;
                const float cv = __shfl(s, srcl, 64);
                const int q = 4 * (srcl >> 4) + r;
                float *lsq = ls + q * k;
                int *liq = li + q * k;
                // wave-uniform per-query state (static indexing through the unrolled select)
                int nq = 0;
                float tq = -INFINITY;
#pragma unroll
                for (int gg = 0; gg < 4; ++gg)
                    if (q == 4 * gg + r) { nq = n_in[4 * gg + r]; tq = thr[4 * gg + r]; }
                if (!(cv > tq)) continue;
                int pos = 0;
                for (int e = lane; e < nq; e += 64) pos += lsq[e] >= cv ? 1 : 0;
                pos = wave_sum_i(pos);
                const int new_n = nq < k ? nq + 1 : k;
                for (int e0 = ((new_n - 1) / 64) * 64; e0 >= 0; e0 -= 64) {
                    const int e = e0 + lane;
                    const bool mv = e >= pos && e < new_n - 1;
                    float sv = 0.f;
                    int iv = 0;
                    if (mv) { sv = lsq[e]; iv = liq[e]; }
                    __builtin_amdgcn_wave_barrier();
                    if (mv) { lsq[e + 1] = sv; liq[e + 1] = iv; }
                    __builtin_amdgcn_wave_barrier();
                }
                if (lane == 0) {
                    lsq[pos] = cv;
                    liq[pos] = (int)(n0 + (srcl & 15));
                }
                __builtin_amdgcn_wave_barrier();
                const float kth = new_n == k ? lsq[k - 1] : -INFINITY;
#pragma unroll
                for (int gg = 0; gg < 4; ++gg)
                    if (q == 4 * gg + r) {
                        n_in[4 * gg + r] = new_n;
                        thr[4 * gg + r] = new_n == k ? kth : floor_thr[4 * gg + r];
                    }
            }
        }
    }
    // write this wave's lists: part[q][wid][k]
    const int nlists = gridDim.x * WAVES;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        if (q >= Q) continue;
        const int64_t o = ((int64_t)q * nlists + wid) * k;
        // the list's entries and ONE terminator (-inf, -1): what follows the first negative index is unspecified (contract
        // of skyemb_cosine_topk; with a pruning floor a wave keeps a handful of rows, and 2048 lists x 16 queries x k padded
        // slots were 39 MB written here and read back by the merge)
        const int n_out = n_in[q] < k ? n_in[q] + 1 : k;
        for (int e = lane; e < n_out; e += 64) {
            const bool have = e < n_in[q];
            part_s[o + e] = have ? ls[q * k + e] : -INFINITY;
            part_i[o + e] = have ? idx_offset + (int64_t)li[q * k + e] : -1;
        }
    }
}

// Exact scores of a (small) set of rows for Q <= 16 queries, same arithmetic as the kernel above: one wave per 16 rows,
// rows straight from HBM into registers.  Serves the pruning floor's row sample (25,600 rows: 1,600 waves) -- the tiled
// score kernel took 50 us for it on 100 workgroups.
template <int WAVES, bool TILEMAX>
__global__ __launch_bounds__(WAVES * 64) void cosine_scores_stream_kernel(const float *__restrict__ tw, const float *__restrict__ qn,
                                                                          const float *__restrict__ bank, const float *__restrict__ xn,
                                                                          int Q, int64_t N, int D, float eps, float *__restrict__ scores) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nchunk = D >> 4;
    float4 *imgA = (float4 *)lds;
    build_imgA<WAVES>(imgA, tw, Q, D, nchunk, tid);
    __syncthreads();
    const int n_lane = lane & 15, g = lane >> 4;
    float qn4[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) qn4[r] = (4 * g + r) < Q ? qn[4 * g + r] : 0.f;
    const int64_t ntile = (N + 15) / 16;
    for (int64_t t = (int64_t)blockIdx.x * WAVES + wave; t < ntile; t += (int64_t)gridDim.x * WAVES) {
        const int64_t n0 = t * 16;
        int64_t row = n0 + n_lane;
        const bool row_ok = row < N;
        if (!row_ok) row = N - 1;
        const float *src = bank + row * D + 4 * g;
        const f32x4 acc = stream_dot16(src, imgA, nchunk, lane);
        const float xnv = xn[row];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int q = 4 * g + r;
            float sc = (row_ok && q < Q) ? finish_score(acc[r], qn4[r], xnv, eps) : -INFINITY;
            if (!TILEMAX) {
                if (row_ok && q < Q) scores[(int64_t)q * N + n0 + n_lane] = sc;
            } else {
                // maximum over the tile's 16 rows (the 16 lanes of this lane group): scores[q][tile]
#pragma unroll
                for (int o = 8; o > 0; o >>= 1) sc = fmaxf(sc, __shfl_xor(sc, o, 64));
                if (n_lane == 0 && q < Q) scores[(int64_t)q * ntile + t] = sc;
            }
        }
    }
}

}  // namespace

bool skyemb_topk_stream_applicable(int Q, int D, int k) { return Q <= 16 && k <= 128 && D % (16 * UNROLL) == 0 && D <= 1024; }

// 8 waves per workgroup when their private lists fit next to the A image in 160 KiB of LDS, else 4
static int stream_waves(int D, int k) {
    const size_t img = (size_t)(D >> 4) * 64 * 16;
    return img + (size_t)2 * 8 * 16 * k * 4 <= 160 * 1024 ? 8 : 4;
}

int skyemb_topk_stream_lists(int64_t N, int D, int k) {
    const int waves = stream_waves(D, k);
    int64_t blocks = 256;
    while (blocks > 1 && blocks * waves * 64 > N) blocks >>= 1;   // at least 64 rows per wave
    return (int)(blocks * waves);
}

bool skyemb_scores_stream_applicable(int Q, int64_t N, int D) { return Q <= 16 && D % (16 * UNROLL) == 0 && D <= 1024 && N <= (1 << 20); }

int skyemb_scores_stream_launch(const float *tw, const float *qn, const float *bank, const float *xn, int Q, int64_t N, int D,
                                float eps, float *scores, hipStream_t st) {
    const size_t smem = (size_t)(D >> 4) * 64 * 16;
    int64_t blocks = ceil_div64(ceil_div64(N, 16), 4);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL((cosine_scores_stream_kernel<4, false>), dim3((unsigned)blocks), dim3(256), smem, st, tw, qn, bank, xn, Q, N, D, eps, scores);
    SKY_LAUNCH_CHECK("skyemb_cosine_scores(stream)");
    return 0;
}

int skyemb_scores_stream_tilemax_launch(const float *tw, const float *qn, const float *bank, const float *xn, int Q, int64_t N, int D,
                                        float eps, float *tile_max, hipStream_t st) {
    const size_t smem = (size_t)(D >> 4) * 64 * 16;
    int64_t blocks = ceil_div64(ceil_div64(N, 16), 4);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL((cosine_scores_stream_kernel<4, true>), dim3((unsigned)blocks), dim3(256), smem, st, tw, qn, bank, xn, Q, N, D, eps, tile_max);
    SKY_LAUNCH_CHECK("skyemb_cosine_sample_floor(scores)");
    return 0;
}

int skyemb_topk_stream_launch(const float *tw, const float *qn, const float *bank, const float *xn, int Q, int64_t N, int D,
                              int k, float eps, int64_t idx_offset, int nlists, float *part_s, int64_t *part_i,
                              const float *thr0, hipStream_t st) {
    const int waves = stream_waves(D, k);
    const int blocks = nlists / waves;
    const size_t smem = sizeof(float) * ((size_t)(D >> 4) * 64 * 4 + 2 * (size_t)waves * 16 * k);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void *)cosine_topk_stream_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           160 * 1024);
        if (e == hipSuccess)
            e = hipFuncSetAttribute((const void *)cosine_topk_stream_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    160 * 1024);
        if (e != hipSuccess) {
            skyemb_set_error("skyemb_cosine_topk(stream): hipFuncSetAttribute: %s", hipGetErrorString(e));
            return 2;
        }
        attr_set = true;
    }
    int64_t rows_per_wave = ceil_div64(ceil_div64(N, nlists), 16) * 16;
    if (waves == 8)
        hipLaunchKernelGGL(cosine_topk_stream_kernel<8>, dim3(blocks), dim3(512), smem, st, tw, qn, bank, xn, Q, N, D, k, eps,
                           idx_offset, rows_per_wave, part_s, part_i, thr0);
    else
        hipLaunchKernelGGL(cosine_topk_stream_kernel<4>, dim3(blocks), dim3(256), smem, st, tw, qn, bank, xn, Q, N, D, k, eps,
                           idx_offset, rows_per_wave, part_s, part_i, thr0);
    SKY_LAUNCH_CHECK("skyemb_cosine_topk(stream)");
    return 0;
}
