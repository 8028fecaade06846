// Weighted-cosine scoring + exact top-k over an embedding bank (utils/similarity.py:18-35,98-102,
// 149-172), bit-exact against oracle/topk_oracle.c:
//   dot(q,n) is ONE fp32 fma chain over d = 0..D-1 -- v_mfma_f32_16x16x4_f32 is a k-ordered fmaf
//   chain on its C input, and consecutive MFMAs take consecutive 4-wide k groups;
//   score = dot / fmaf(qn, xn, eps) with IEEE sqrt / divide.
// Never materialises the Q x N score matrix: every block keeps, per query of its tile, a sorted
// top-k list in LDS for its bank chunk (threshold-filtered insertion); the per-chunk lists are
// merged by skyemb_topk_merge (also used after the RCCL all-gather of per-rank results).
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int BK = 16;           // k-tile (floats)
constexpr int PITCH = BK + 4;    // LDS row pitch of a staged operand tile

// ------------------------------------------------------------------------------------------------
// standardise: (x - mu) / (sigma + 1e-8)   (utils/similarity.py:101-102)
__global__ __launch_bounds__(256) void standardise_kernel(const float *__restrict__ x, const float *__restrict__ mu,
                                                          const float *__restrict__ sigma, float *__restrict__ out,
                                                          int64_t total4, int D4) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (int64_t)gridDim.x * 256) {
        const int d = (int)(i % D4) * 4;
        const float4 v = *(const float4 *)(x + 4 * i);
        const float4 m = *(const float4 *)(mu + d), s = *(const float4 *)(sigma + d);
        float4 o;
        o.x = __fdiv_rn(__fsub_rn(v.x, m.x), __fadd_rn(s.x, 1e-8f));
        o.y = __fdiv_rn(__fsub_rn(v.y, m.y), __fadd_rn(s.y, 1e-8f));
        o.z = __fdiv_rn(__fsub_rn(v.z, m.z), __fadd_rn(s.z, 1e-8f));
        o.w = __fdiv_rn(__fsub_rn(v.w, m.w), __fadd_rn(s.w, 1e-8f));
        *(float4 *)(out + 4 * i) = o;
    }
}

// ------------------------------------------------------------------------------------------------
// norms[n] = sqrt( chain_d fma(w[d]*x[n][d], x[n][d], acc) ), optional xw_out = w*x.
// Block = 256 rows; 32-wide d chunks are staged coalesced through LDS, each thread walks its own
// row in order (the chain is serial by definition).
__global__ __launch_bounds__(256) void wnorm_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                    float *__restrict__ norms, float *__restrict__ xw_out, int64_t N,
                                                    int D) {
    __shared__ float tile[256][33];
    const int tid = threadIdx.x;
    const int64_t n0 = (int64_t)blockIdx.x * 256;
    float acc = 0.f;
    for (int d0 = 0; d0 < D; d0 += 32) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int v = tid + i * 256, r = v >> 3, c = (v & 7) * 4;
            float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
            if (n0 + r < N && d0 + c < D) val = *(const float4 *)(x + (n0 + r) * D + d0 + c);
            tile[r][c] = val.x; tile[r][c + 1] = val.y; tile[r][c + 2] = val.z; tile[r][c + 3] = val.w;
        }
        __syncthreads();
        const int dn = (D - d0) < 32 ? (D - d0) : 32;
        for (int dd = 0; dd < dn; ++dd) {
            const float xv = tile[tid][dd];
            const float xw = w ? __fmul_rn(w[d0 + dd], xv) : xv;
            acc = fmaf(xw, xv, acc);
            if (xw_out) tile[tid][dd] = xw;
        }
        if (xw_out) {
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int v = tid + i * 256, r = v >> 3, c = (v & 7) * 4;
                if (n0 + r < N && d0 + c < D)
                    *(float4 *)(xw_out + (n0 + r) * D + d0 + c) =
                        make_float4(tile[r][c], tile[r][c + 1], tile[r][c + 2], tile[r][c + 3]);
            }
        }
    }
    if (n0 + tid < N) norms[n0 + tid] = __fsqrt_rn(acc);
}

// The same chain for a handful of rows (the queries of a search: N <= 16): the rows are staged in LDS by the whole
// workgroup (coalesced), then one lane per row walks its row -- the tiled kernel above spends 70 us on 16 rows (one workgroup, two barriers per
// 32 columns, a serial chain out of LDS).
template <bool HAS_W, bool HAS_OUT>
__global__ __launch_bounds__(256) void wnorm_rows_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                         float *__restrict__ norms, float *__restrict__ xw_out, int64_t N, int D) {
    extern __shared__ __attribute__((aligned(16))) float rows_lds[];    // [N][D + 4]: the rows, + w in the last row
    const int tid = threadIdx.x, P = D + 4;                              // pitch: lanes of different rows on different banks
    float *wl = rows_lds + (size_t)N * P;
    const int d4 = D >> 2;
    // (four loads in flight per thread before the first LDS write: left to itself the compiler waits for every load in
    // turn -- twelve dependent memory round trips for 16 rows of 768)
    const int total = (int)N * d4;
    for (int e0 = tid; e0 < total; e0 += 4 * 256) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = e0 + u * 256 < total ? e0 + u * 256 : total - 1;
            const int r = e / d4, c = e - r * d4;
            v[u] = *(const float4 *)(x + (int64_t)r * D + 4 * c);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = e0 + u * 256;
            if (e < total) {
                const int r = e / d4, c = e - r * d4;
                *(float4 *)(rows_lds + (size_t)r * P + 4 * c) = v[u];
            }
        }
    }
    if (HAS_W)
        for (int e = tid; e < d4; e += 256) *(float4 *)(wl + 4 * e) = *(const float4 *)(w + 4 * e);
    __syncthreads();
    if (tid < N) {
        const float *row = rows_lds + (size_t)tid * P;
        float acc = 0.f;
        // eight float4 groups of the row (and of w) are requested from LDS before the chain consumes them: group by group,
        // with the `w` / `xw_out` tests inside the loop, every group cost four LDS round trips (18 of the kernel's 21 us)
        for (int d0 = 0; d0 < D; d0 += 32) {
            float4 v[8], ww[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int d = d0 + 4 * u < D ? d0 + 4 * u : D - 4;
                v[u] = *(const float4 *)(row + d);
                if (HAS_W) ww[u] = *(const float4 *)(wl + d);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int d = d0 + 4 * u;
                if (d < D) {
                    const float xs[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
                    float o[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float wj = j == 0 ? ww[u].x : j == 1 ? ww[u].y : j == 2 ? ww[u].z : ww[u].w;
                        o[j] = HAS_W ? __fmul_rn(wj, xs[j]) : xs[j];
                        acc = fmaf(o[j], xs[j], acc);
                    }
                    // (straight to global memory: a store into the LDS rows would order every later LDS read behind it)
                    if (HAS_OUT) *(float4 *)(xw_out + (int64_t)tid * D + d) = make_float4(o[0], o[1], o[2], o[3]);
                }
            }
        }
        norms[tid] = __fsqrt_rn(acc);
    }
}

// ------------------------------------------------------------------------------------------------
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

template <int R, int NT>
__device__ __forceinline__ void load_rows(const float *__restrict__ X, int64_t r0, int64_t rows, int D, int k0, int tid,
                                          float4 (&reg)[(R * BK / 4 + NT - 1) / NT]) {
    constexpr int NV = (R * BK / 4 + NT - 1) / NT;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int v = tid + i * NT;
        const int r = v >> 2, k = (v & 3) * 4;
        const bool ok = (R * BK / 4 % NT == 0 || v < R * BK / 4) && r0 + r < rows && k0 + k < D;
        reg[i] = ok ? *(const float4 *)(X + (r0 + r) * D + k0 + k) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}
template <int R, int NT>
__device__ __forceinline__ void store_rows(float *s, int tid, const float4 (&reg)[(R * BK / 4 + NT - 1) / NT]) {
    constexpr int NV = (R * BK / 4 + NT - 1) / NT;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int v = tid + i * NT;
        if (R * BK / 4 % NT == 0 || v < R * BK / 4) *(float4 *)&s[(v >> 2) * PITCH + (v & 3) * 4] = reg[i];
    }
}

// QT queries x BT bank rows per tile; NW waves as WM x WN.  NW = 8 (two waves per SIMD: one wave's MFMAs cover the other's
// LDS / barrier waits) for the many-query tile 64 x 256; its score tile is exchanged in 128-column slabs to fit the LDS.
template <int QT, int BT, bool SCORES_ONLY, int NW>
__global__ __launch_bounds__(64 * NW) void cosine_topk_kernel(const float *__restrict__ tw, const float *__restrict__ qn,
                                                              const float *__restrict__ bank, const float *__restrict__ xn,
                                                              int Q, int64_t N, int D, int k, float eps, int64_t idx_offset,
                                                              int nchunks, int64_t rows_per_chunk, float *__restrict__ part_s,
                                                              int64_t *__restrict__ part_i, float *__restrict__ scores,
                                                              const float *__restrict__ thr0) {
    constexpr int NT = 64 * NW;
    constexpr int WM = QT >= 64 ? 2 : 1, WN = NW / WM;
    constexpr int TM = QT / WM / 16, TN = BT / WN / 16;
    constexpr int SLAB = NW > 4 ? 128 : BT, NSLAB = BT / SLAB;      // score-tile columns exchanged per epilogue pass
    constexpr int SCP = SLAB + 1;
    constexpr int QPW = QT / NW;                                     // queries whose lists a wave owns
    static_assert(BT / WN <= SLAB && SLAB % (BT / WN) == 0 && QT % NW == 0, "tile / wave layout");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *sA0 = lds, *sA1 = sA0 + QT * PITCH, *sB0 = sA1 + QT * PITCH, *sB1 = sB0 + BT * PITCH;
    float *sc = sB1 + BT * PITCH;            // [QT][SCP]
    float *ls = sc + QT * SCP;               // [QT][k]
    int *li = (int *)(ls + (SCORES_ONLY ? 0 : QT * k));  // [QT][k]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int qt = blockIdx.x / nchunks, chunk = blockIdx.x % nchunks;
    const int q0 = qt * QT;
    const int64_t c_begin = (int64_t)chunk * rows_per_chunk;
    int64_t c_end = c_begin + rows_per_chunk;
    if (c_end > N) c_end = N;
    const int KT = (D + BK - 1) / BK;

    int cnt[QPW];  // per-wave list sizes for its queries (static indexing below)
#pragma unroll
    for (int i = 0; i < QPW; ++i) cnt[i] = 0;

    for (int64_t nb = c_begin; nb < c_end; nb += BT) {
        f32x4 acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        float4 ra[(QT * BK / 4 + NT - 1) / NT], rb[(BT * BK / 4 + NT - 1) / NT];
        load_rows<QT, NT>(tw, q0, Q, D, 0, tid, ra);
        load_rows<BT, NT>(bank, nb, c_end, D, 0, tid, rb);
        __syncthreads();  // previous tile's epilogue is done with LDS
        store_rows<QT, NT>(sA0, tid, ra);
        store_rows<BT, NT>(sB0, tid, rb);
        __syncthreads();
        for (int kt = 0; kt < KT; ++kt) {
            const bool more = kt + 1 < KT;
            if (more) {
                load_rows<QT, NT>(tw, q0, Q, D, (kt + 1) * BK, tid, ra);
                load_rows<BT, NT>(bank, nb, c_end, D, (kt + 1) * BK, tid, rb);
            }
            const float *cA = (kt & 1) ? sA1 : sA0;
            const float *cB = (kt & 1) ? sB1 : sB0;
#pragma unroll
            for (int kk = 0; kk < BK / 4; ++kk) {
                float fa[TM], fb[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    fa[i] = cA[(wm * (QT / WM) + i * 16 + (lane & 15)) * PITCH + kk * 4 + (lane >> 4)];
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    fb[j] = cB[(wn * (BT / WN) + j * 16 + (lane & 15)) * PITCH + kk * 4 + (lane >> 4)];
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i], fb[j], acc[i][j], 0, 0, 0);
            }
            if (more) {
                store_rows<QT, NT>((kt & 1) ? sA0 : sA1, tid, ra);
                store_rows<BT, NT>((kt & 1) ? sB0 : sB1, tid, rb);
            }
            __syncthreads();
        }
        // ---- finish scores into the LDS score tile (C/D: col = lane&15 -> bank row, row -> query), SLAB columns per pass ----
#pragma unroll
        for (int slab = 0; slab < NSLAB; ++slab) {
            if (slab > 0) __syncthreads();          // the previous slab's scan is done with sc
            if ((wn * (BT / WN)) / SLAB == slab) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int qr = wm * (QT / WM) + i * 16 + 4 * (lane >> 4) + r;
                        const bool qok = q0 + qr < Q;
                        const float qnv = qok ? qn[q0 + qr] : 0.f;
#pragma unroll
                        for (int j = 0; j < TN; ++j) {
                            const int c = wn * (BT / WN) + j * 16 + (lane & 15);
                            const bool ok = qok && nb + c < c_end;
                            sc[qr * SCP + c - slab * SLAB] = ok ? finish_score(acc[i][j][r], qnv, xn[nb + c], eps) : -INFINITY;
                        }
                    }
            }
            __syncthreads();
            if (SCORES_ONLY) {
                for (int e = tid; e < QT * SLAB; e += NT) {
                    const int qr = e / SLAB, c = e % SLAB + slab * SLAB;
                    if (q0 + qr < Q && nb + c < c_end) scores[(int64_t)(q0 + qr) * N + nb + c] = sc[qr * SCP + c - slab * SLAB];
                }
            } else {
                // ---- per-query threshold-filtered sorted insertion; wave w owns queries w*QPW.. ----
#pragma unroll
                for (int qi = 0; qi < QPW; ++qi) {
                    const int qq = wave * QPW + qi;
                    if (q0 + qq >= Q) continue;
                    float *lsq = ls + qq * k;
                    int *liq = li + qq * k;
                    int n_in = cnt[qi];
                    const float floor_thr = thr0 ? thr0[q0 + qq] : -INFINITY;   // valid lower bound of the global k-th best
                    float thr = n_in == k ? lsq[k - 1] : floor_thr;
                    for (int t = 0; t < SLAB / 64; ++t) {
                        const float v = sc[qq * SCP + t * 64 + lane];
                        unsigned long long m = __ballot(v > thr);
                        while (m) {
                            const int src = __builtin_ctzll(m);
                            m &= m - 1;
                            const float cv = __shfl(v, src, 64);
                            if (!(cv > thr)) continue;  // threshold rose during this batch
                            int pos = 0;
                            for (int e = lane; e < n_in; e += 64) pos += lsq[e] >= cv ? 1 : 0;
                            pos = wave_sum_i(pos);
                            const int new_n = n_in < k ? n_in + 1 : k;
                            // shift [pos, new_n-1) down by one, highest 64-chunk first (read-then-write per chunk)
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
                                liq[pos] = (int)(nb + slab * SLAB + t * 64 + src);  // row index local to this bank shard
                            }
                            __builtin_amdgcn_wave_barrier();
                            n_in = new_n;
                            thr = n_in == k ? lsq[k - 1] : floor_thr;
                        }
                    }
                    cnt[qi] = n_in;
                }
            }
        }
    }
    if (!SCORES_ONLY) {
        __syncthreads();
#pragma unroll
        for (int qi = 0; qi < QPW; ++qi) {
            const int qq = wave * QPW + qi;
            if (q0 + qq >= Q) continue;
            const int64_t o = ((int64_t)(q0 + qq) * nchunks + chunk) * k;
            for (int e = lane; e < k; e += 64) {
                const bool have = e < cnt[qi];
                part_s[o + e] = have ? ls[qq * k + e] : -INFINITY;
                part_i[o + e] = have ? idx_offset + (int64_t)li[qq * k + e] : -1;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// merge: one wave per query, `nlists` sorted lists of length k -> best k by (score desc, idx asc).
// Heads live in LDS; each round every lane proposes the best head among its lists, a butterfly
// picks the winner, the owning lane advances that list.
__device__ __forceinline__ void tournament_merge(const float *__restrict__ in_s, const int64_t *__restrict__ in_i,
                                                 int nlists, int k, float *__restrict__ out_s,
                                                 int64_t *__restrict__ out_i, char *smem) {
    float *hs = (float *)smem;                       // [nlists] head score
    int64_t *hi = (int64_t *)(smem + ((nlists * 4 + 15) & ~15));  // [nlists] head idx
    int *hp = (int *)(hi + nlists);                  // [nlists] head position
    const int lane = threadIdx.x, q = blockIdx.x;
    const float *ps = in_s + (int64_t)q * nlists * k;
    const int64_t *pi = in_i + (int64_t)q * nlists * k;
    for (int l = lane; l < nlists; l += 64) {
        hs[l] = ps[(int64_t)l * k];
        hi[l] = pi[(int64_t)l * k];
        hp[l] = 0;
    }
    __builtin_amdgcn_wave_barrier();
    for (int r = 0; r < k; ++r) {
        float bs = -INFINITY;
        int64_t bi = INT64_MAX;
        int bl = -1;
        for (int l = lane; l < nlists; l += 64) {
            const float s = hs[l];
            const int64_t ix = hi[l];
            if (ix >= 0 && (s > bs || (s == bs && ix < bi))) { bs = s; bi = ix; bl = l; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float s2 = __shfl_xor(bs, o, 64);
            const int lo2 = __shfl_xor((int)(bi & 0xffffffffll), o, 64), hi2 = __shfl_xor((int)(bi >> 32), o, 64);
            const int l2 = __shfl_xor(bl, o, 64);
            const int64_t i2 = ((int64_t)hi2 << 32) | (unsigned int)lo2;
            if (l2 >= 0 && (bl < 0 || s2 > bs || (s2 == bs && i2 < bi))) { bs = s2; bi = i2; bl = l2; }
        }
        if (lane == 0) {
            out_s[(int64_t)q * k + r] = bl >= 0 ? bs : -INFINITY;
            out_i[(int64_t)q * k + r] = bl >= 0 ? bi : -1;
        }
        if (bl >= 0 && (bl & 63) == lane) {  // owner advances the winning list
            const int np = hp[bl] + 1;
            hp[bl] = np;
            if (np < k) { hs[bl] = ps[(int64_t)bl * k + np]; hi[bl] = pi[(int64_t)bl * k + np]; }
            else { hs[bl] = -INFINITY; hi[bl] = -1; }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

__global__ __launch_bounds__(64) void topk_merge_kernel(const float *__restrict__ in_s, const int64_t *__restrict__ in_i,
                                                        int nlists, int k, float *__restrict__ out_s,
                                                        int64_t *__restrict__ out_i) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    tournament_merge(in_s, in_i, nlists, k, out_s, out_i, smem);
}

// ------------------------------------------------------------------------------------------------
// merge of MANY lists (bank-streaming kernel: one short list per wavefront): the tournament above
// pays one dependent global load per output element, so instead the valid entries (few once a
// pruning floor is used) are gathered into LDS as 64-bit sort keys and bitonic-sorted by the block.
//   key = orderable(score) << 32 | ~idx   (descending key order == score desc, index asc)
// Falls back to the tournament when the entries do not fit (or an index needs more than 32 bits).
constexpr int MERGE_CAP = 16384;   // 128 KiB of LDS keys

__device__ __forceinline__ unsigned int orderable(float f) {
    const unsigned int u = __float_as_uint(f);
    return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float unorderable(unsigned int o) {
    return __uint_as_float(o ^ ((o >> 31) ? 0x80000000u : 0xFFFFFFFFu));
}

__global__ __launch_bounds__(1024) void topk_merge_sort_kernel(const float *__restrict__ in_s, const int64_t *__restrict__ in_i,
                                                               int nlists, int k, float *__restrict__ out_s,
                                                               int64_t *__restrict__ out_i) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long keys[];   // [MERGE_CAP]
    __shared__ int wsum[16];
    __shared__ int total_s, bad_s;
    const int tid = threadIdx.x, q = blockIdx.x;
    const float *ps = in_s + (int64_t)q * nlists * k;
    const int64_t *pi = in_i + (int64_t)q * nlists * k;
    if (tid == 0) bad_s = 0;
    __syncthreads();
    // valid-prefix length of each list: entries up to the first negative index (what follows it is unspecified -- the
    // bank-streaming kernel writes one terminator, not k - n padding slots).  With a pruning floor a list holds a few rows: the
    // first HEAD entries of a thread's lists are requested together (one memory round trip instead of one per entry; k >= 1,
    // clamped reads stay inside the list) and kept for the gather below
    // ... and a list that goes on past its head is walked eight entries per round trip (full lists -- no pruning floor: banks
    // under 8 x 256 x k rows -- were k dependent loads each)
    auto list_length = [&](int64_t base, int e) {
        while (e < k) {
            int64_t v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = pi[base + (e + u < k ? e + u : k - 1)];
            int f = 8;
#pragma unroll
            for (int u = 7; u >= 0; --u)
                if (e + u >= k || v[u] < 0) f = u;
            e += f;
            if (f < 8) break;
        }
        return e < k ? e : k;
    };
    constexpr int HEAD = 4, LPT = 2;                          // lists per thread handled with the register heads (nlists <= 2048)
    int64_t hix[LPT][HEAD];
    float hsc[LPT][HEAD];
    int hlen[LPT];
    const bool heads = nlists <= LPT * 1024;
    int mycount = 0;
    if (heads) {
#pragma unroll
        for (int j = 0; j < LPT; ++j) {
            const int l = tid + j * 1024;
            const int64_t base = (int64_t)(l < nlists ? l : 0) * k;
#pragma unroll
            for (int e = 0; e < HEAD; ++e) {
                hix[j][e] = pi[base + (e < k ? e : k - 1)];
                hsc[j][e] = ps[base + (e < k ? e : k - 1)];
            }
        }
#pragma unroll
        for (int j = 0; j < LPT; ++j) {
            const int l = tid + j * 1024;
            int e = 0;
#pragma unroll
            for (int h = 0; h < HEAD; ++h) e += (e == h && h < k && hix[j][h] >= 0) ? 1 : 0;
            if (l < nlists && e == HEAD) e = list_length((int64_t)l * k, e);
            hlen[j] = l < nlists ? e : 0;
            mycount += hlen[j];
        }
    } else {
        for (int l = tid; l < nlists; l += 1024) mycount += list_length((int64_t)l * k, 0);
    }
    // block exclusive scan of per-thread counts
    int incl = mycount;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o, 64);
        if ((tid & 63) >= o) incl += v;
    }
    if ((tid & 63) == 63) wsum[tid >> 6] = incl;
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int w = 0; w < 16; ++w) { const int t = wsum[w]; wsum[w] = run; run += t; }
        total_s = run;
    }
    __syncthreads();
    const int T = total_s;
    if (T > MERGE_CAP) {
        // more entries than the LDS image takes: the first wave merges this query's lists by the tournament (heads in the
        // image's space); it used to be a second launch behind every merge
        if (tid < 64) tournament_merge(in_s, in_i, nlists, k, out_s, out_i, (char *)keys);
        return;
    }
    int off = wsum[tid >> 6] + incl - mycount;
    if (heads) {
#pragma unroll
        for (int j = 0; j < LPT; ++j) {
            const int l = tid + j * 1024;
#pragma unroll
            for (int e = 0; e < HEAD; ++e)
                if (e < hlen[j]) {
                    if (hix[j][e] > 0xFFFFFFFFll) bad_s = 1;
                    keys[off++] = ((unsigned long long)orderable(hsc[j][e]) << 32) | (unsigned int)(~(unsigned int)hix[j][e]);
                }
            for (int e = HEAD; e < hlen[j]; ++e) {
                const int64_t ix = pi[(int64_t)l * k + e];
                if (ix > 0xFFFFFFFFll) bad_s = 1;
                keys[off++] = ((unsigned long long)orderable(ps[(int64_t)l * k + e]) << 32) | (unsigned int)(~(unsigned int)ix);
            }
        }
    } else {
        for (int l = tid; l < nlists; l += 1024) {
            for (int e = 0; e < k; ++e) {
                const int64_t ix = pi[(int64_t)l * k + e];
                if (ix < 0) break;
                if (ix > 0xFFFFFFFFll) bad_s = 1;
                keys[off++] = ((unsigned long long)orderable(ps[(int64_t)l * k + e]) << 32) | (unsigned int)(~(unsigned int)ix);
            }
        }
    }
    __syncthreads();
    if (bad_s) {
        __syncthreads();                                       // (every wave has read bad_s; wave 0 now reuses the image)
        if (tid < 64) tournament_merge(in_s, in_i, nlists, k, out_s, out_i, (char *)keys);
        return;
    }
    // Only the best k of the T gathered entries are wanted: select the k-th largest key first (keys are unique: 8 radix
    // passes of 8 bits over the 64-bit keys, histograms in LDS), keep the k keys >= it, and sort just those -- instead of a
    // bitonic sort of all T (thousands with a sample-derived floor: ~80 block-wide stages).
    int n = T;
    if (T > k) {                                              // (a bitonic sort of all T <= 4096 entries, 78 block-wide stages, was measured: 73 vs 31 us)
        __shared__ int hist[258];
        unsigned long long prefix = 0ull, mask = 0ull;
        int kth = k;
        for (int shift = 56; shift >= 0; shift -= 8) {
            for (int b = tid; b < 256; b += 1024) hist[b] = 0;
            __syncthreads();
            for (int e = tid; e < T; e += 1024) {
                const unsigned long long kx = keys[e];
                if ((kx & mask) == prefix) atomicAdd(&hist[(int)((kx >> shift) & 255ull)], 1);
            }
            __syncthreads();
            radix_pick_bin(hist, kth, tid);
            __syncthreads();
            prefix |= (unsigned long long)hist[256] << shift;
            kth = hist[257];
            mask |= 255ull << shift;
            __syncthreads();
        }
        // compaction of the k winners into the front of a second region (MERGE_CAP keys are followed by room for 512 more)
        unsigned long long *win = keys + MERGE_CAP;
        if (tid == 0) hist[0] = 0;
        __syncthreads();
        for (int e = tid; e < T; e += 1024)
            if (keys[e] >= prefix) win[atomicAdd(&hist[0], 1)] = keys[e];
        __syncthreads();
        for (int e = tid; e < k; e += 1024) keys[e] = win[e];
        n = k;
    }
    int n2 = 1;
    while (n2 < n) n2 <<= 1;
    if (n2 < 2) n2 = 2;
    __syncthreads();
    for (int e = n + tid; e < n2; e += 1024) keys[e] = 0ull;   // worst possible key
    __syncthreads();
    // bitonic sort, descending
    for (int k2 = 2; k2 <= n2; k2 <<= 1) {
        for (int j = k2 >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < n2; i += 1024) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const unsigned long long a = keys[i], b = keys[ixj];
                    const bool desc = (i & k2) == 0;
                    if (desc ? (a < b) : (a > b)) { keys[i] = b; keys[ixj] = a; }
                }
            }
            __syncthreads();
        }
    }
    for (int e = tid; e < k; e += 1024) {
        if (e < n) {
            const unsigned long long key = keys[e];
            out_s[(int64_t)q * k + e] = unorderable((unsigned int)(key >> 32));
            out_i[(int64_t)q * k + e] = (int64_t)(unsigned int)(~(unsigned int)key);
        } else {
            out_s[(int64_t)q * k + e] = -INFINITY;
            out_i[(int64_t)q * k + e] = -1;
        }
    }
}



// ------------------------------------------------------------------------------------------------
// pruning floor: per row of x [Q, S] the k-th largest value, returned one ulp lower (the top-k kernels keep rows STRICTLY
// above the floor).  One workgroup per row: 4 radix passes of 8 bits over orderable keys, histograms in LDS.
// Replaces torch.topk on the sample-score matrix (a dozen multi-block top-k / sort launches for a [Q, 25600] matrix).
__global__ __launch_bounds__(1024) void kth_floor_kernel(const float *__restrict__ x, int S, int kth, float *__restrict__ out) {
    __shared__ int hist[258];
    const int tid = threadIdx.x;
    const float *row = x + (int64_t)blockIdx.x * S;
    unsigned int prefix = 0, mask = 0;
    for (int shift = 24; shift >= 0; shift -= 8) {
        for (int b = tid; b < 256; b += 1024) hist[b] = 0;
        __syncthreads();
        for (int e = tid; e < S; e += 1024) {
            float v = row[e];
            if (!(v == v)) v = -INFINITY;                      // NaN scores rank as -inf (contract)
            const unsigned int u = __float_as_uint(v);
            const unsigned int key = u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
            if ((key & mask) == prefix) atomicAdd(&hist[(key >> shift) & 255], 1);
        }
        __syncthreads();
        radix_pick_bin(hist, kth, tid);
        __syncthreads();
        prefix |= (unsigned int)hist[256] << shift;
        kth = hist[257];
        mask |= 255u << shift;
        __syncthreads();
    }
    if (tid == 0) {
        const float v = __uint_as_float(prefix ^ ((prefix >> 31) ? 0x80000000u : 0xFFFFFFFFu));
        out[blockIdx.x] = nextafterf(v, -INFINITY);
    }
}

// The same result for S <= 32 K values per row without atomics: every thread keeps its (up to 32) keys in registers and the
// k-th largest key is found by bisection on the 32-bit key space -- per step one ballot + popcount per register and a
// 16-value sum through LDS.  (The radix select above spends its first pass adding 25,600 scores, which share a handful of
// sign / exponent bytes, into three or four LDS words: ~45 us for the [16, 25600] sample matrix of a search, against 0.5 ms
// for the whole bank pass it prunes.)
__device__ __forceinline__ int wave_total(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// largest t (searched over bits `top` .. 0 on top of the prefix `t`) with |{key[j] >= t}| >= kth over the NK keys each lane of
// ONE wave holds in registers: per bit NK compares + ballots (a wave-uniform count, no cross-lane exchange)
template <int NK>
__device__ __forceinline__ unsigned int wave_bisect(const unsigned int (&key)[NK], unsigned int t, int top, int kth) {
    for (int bit = top; bit >= 0; --bit) {
        const unsigned int cand = t | (1u << bit);
        int c = 0;
#pragma unroll
        for (int j = 0; j < NK; ++j) c += __popcll(__ballot(key[j] >= cand));
        if (c >= kth) t = cand;
    }
    return t;
}
__device__ __forceinline__ unsigned int score_key(float v) {
    if (!(v == v)) v = -INFINITY;                              // NaN scores rank as -inf (contract)
    const unsigned int u = __float_as_uint(v);
    return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);        // > 0 for every real score (a negative NaN would be 0)
}
__device__ __forceinline__ float key_floor(unsigned int t) {
    return nextafterf(__uint_as_float(t ^ ((t >> 31) ? 0x80000000u : 0xFFFFFFFFu)), -INFINITY);
}

// S <= 2048 values per row (e.g. the per-tile maxima of a search's row sample): two keys per thread of a 1024-thread
// workgroup, bisection on the key bits with one barrier per bit (two ballots per wave and sixteen LDS words per bit: ~4 us.
// One wave holding all 2048 keys in 32 registers was tried first: 32 compare -> scalar-popcount pairs per bit made it 22 us).
__global__ __launch_bounds__(1024) void kth_floor_wave_kernel(const float *__restrict__ x, int S, int kth, float *__restrict__ out) {
    __shared__ int part[2][16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float *row = x + (int64_t)blockIdx.x * S;
    const int e0 = tid, e1 = tid + 1024;                       // (clamped, unconditional loads: see the kernel below)
    const float v0 = row[e0 < S ? e0 : S - 1], v1 = row[e1 < S ? e1 : S - 1];
    const unsigned int k0 = e0 < S ? score_key(v0) : 0u, k1 = e1 < S ? score_key(v1) : 0u;
    unsigned int t = 0u;
    for (int bit = 31; bit >= 0; --bit) {
        const unsigned int cand = t | (1u << bit);
        const int c = __popcll(__ballot(k0 >= cand)) + __popcll(__ballot(k1 >= cand));
        const int buf = bit & 1;
        if (lane == 0) part[buf][wave] = c;
        __syncthreads();
        int tot = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) tot += part[buf][w];
        if (tot >= kth) t = cand;
    }
    if (tid == 0) out[blockIdx.x] = key_floor(t);
}

// 2048 < S <= 32 K: every thread of a 1024-thread workgroup keeps its (up to 32) keys in registers; a pivot from a sample
// (wave 0's first registers: the key of rank m there, m = three times the expected sample rank of the k-th largest + 8) is
// CHECKED by one count (kth <= |{key >= pivot}| <= CAP); those keys are pooled in LDS and wave 0 finishes on them in registers.
// A pivot that fails the check falls back to a workgroup-wide bisection, bit by bit, until the keys in play fit the pool.
// No atomics anywhere: the radix select further down adds 25,600 scores sharing a handful of sign / exponent bytes into
// three or four LDS words in its first pass.
__global__ __launch_bounds__(1024) void kth_floor_bisect_kernel(const float *__restrict__ x, int S, int kth, float *__restrict__ out) {
    constexpr int CAP = 1024, NSUB = 8;
    __shared__ int part[3][16];
    __shared__ unsigned int pool[CAP];
    __shared__ unsigned int pivot_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float *row = x + (int64_t)blockIdx.x * S;
    unsigned int key[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        // every load issued unconditionally (clamped index) and selected afterwards: a load under `if (e < S)` makes the compiler
        // branch around it and wait for each one in turn
        const int e = tid + i * 1024;
        const float v = row[e < S ? e : S - 1];
        key[i] = e < S ? score_key(v) : 0u;                    // padding: 0, never counted
    }
    // per-wave counts of the keys >= thr, their total and this wave's offset among them (buffer b of `part`)
    auto count_ge = [&](unsigned int thr, int b, int &total, int &before) {
        int c = 0;
#pragma unroll
        for (int i = 0; i < 32; ++i) c += __popcll(__ballot(key[i] >= thr && key[i] != 0u));
        if (lane == 0) part[b][wave] = c;
        __syncthreads();
        total = before = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) { const int p = part[b][w]; before += w < wave ? p : 0; total += p; }
    };
    if (wave == 0) {
        unsigned int sub[NSUB];
        int nsub = 0;
#pragma unroll
        for (int i = 0; i < NSUB; ++i) {
            sub[i] = key[i];
            const int left = S - i * 1024;
            nsub += left <= 0 ? 0 : (left < 64 ? left : 64);
        }
        const long long m = ((long long)kth * nsub * 3 + S - 1) / S + 8;
        unsigned int piv = 0u;
        if (m <= nsub) piv = wave_bisect<NSUB>(sub, 0u, 31, (int)m);
        if (lane == 0) pivot_s = piv;
    }
    __syncthreads();
    const unsigned int piv = pivot_s;
    int above, before;
    count_ge(piv, 2, above, before);
    unsigned int t = 0u, thr = piv;
    int bit = 31;
    if (piv == 0u || above < kth || above > CAP) {
        above = S;
        for (; bit >= 0 && above > CAP; --bit) {
            int tot, unused;
            count_ge(t | (1u << bit), bit & 1, tot, unused);
            if (tot >= kth) { t |= 1u << bit; above = tot; }   // (else |{key >= t}| is unchanged)
        }
        thr = t;
        __syncthreads();                                       // part[2]'s readers of the pivot count are long done
        if (bit >= 0) count_ge(thr, 2, above, before);
    }
    if (bit >= 0) {
        // pool the keys >= thr (kth <= `above` <= CAP): wave w writes behind the waves before it, lanes in ballot order
        int base = before;
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const bool in = key[i] >= thr && key[i] != 0u;
            const unsigned long long msk = __ballot(in);
            if (in) pool[(base + __popcll(msk & ((1ull << lane) - 1ull))) & (CAP - 1)] = key[i];
            base += __popcll(msk);
        }
        __syncthreads();
        if (wave == 0) {
            unsigned int pk[CAP / 64];
#pragma unroll
            for (int j = 0; j < CAP / 64; ++j) pk[j] = pool[lane + 64 * j];
#pragma unroll
            for (int j = 0; j < CAP / 64; ++j) pk[j] = lane + 64 * j < above ? pk[j] : 0u;
            t = wave_bisect<CAP / 64>(pk, t, bit, kth);
        }
    }
    if (tid == 0) out[blockIdx.x] = key_floor(t);
}

template <int QT, int BT, bool SO, int NW = 4>
int launch_topk(const float *tw, const float *qn, const float *bank, const float *xn, int Q, int64_t N, int D, int k,
                float eps, int64_t idx_offset, int nchunks, float *part_s, int64_t *part_i, float *scores, hipStream_t st,
                const char *name, const float *thr0 = nullptr) {
    constexpr int SLAB = NW > 4 ? 128 : BT;   // score-tile columns per epilogue pass (cosine_topk_kernel)
    const size_t smem = sizeof(float) * (2 * QT * PITCH + 2 * BT * PITCH + QT * (SLAB + 1) + (SO ? 0 : 2 * (size_t)QT * k));
    if (smem > 160 * 1024) {
        skyemb_set_error("%s: k=%d needs %zu B of LDS (max 160 KiB)", name, k, smem);
        return 1;
    }
    auto kern = cosine_topk_kernel<QT, BT, SO, NW>;
    if (smem > 65536) {
        hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) {
            skyemb_set_error("%s: hipFuncSetAttribute: %s", name, hipGetErrorString(e));
            return 2;
        }
    }
    int64_t rows_per_chunk = ceil_div64(ceil_div64(N, nchunks), BT) * BT;
    const int qtiles = (Q + QT - 1) / QT;
    hipLaunchKernelGGL(kern, dim3((unsigned)(qtiles * nchunks)), dim3(64 * NW), smem, st, tw, qn, bank, xn, Q, N, D, k, eps,
                       idx_offset, nchunks, rows_per_chunk, part_s, part_i, scores, thr0);
    hipError_t e_ = hipGetLastError();
    if (e_ != hipSuccess) {
        skyemb_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));
        return 2;
    }
    return 0;
}

inline bool small_tile(int Q, int k) { return Q <= 16 || k > 128; }
// many-query tile: 64 x 256 with 8 waves (two per SIMD) by default; SKYEMB_TOPK_NW=4 keeps the 64 x 128 four-wave tile
inline bool wide_tile() {
    static const bool on = []() { const char *e = getenv("SKYEMB_TOPK_NW"); return !(e && e[0] == '4'); }();
    return on;
}

}  // namespace

extern "C" int skyemb_standardise(const float *x, const float *mu, const float *sigma, float *out, int64_t N, int D,
                                  void *stream) {
    SKY_CHECK_ARG(N > 0 && D > 0 && D % 4 == 0, "skyemb_standardise: bad shape");
    const int64_t total4 = N * D / 4;
    int64_t blocks = ceil_div64(total4, 256);
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(standardise_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, mu, sigma, out,
                       total4, D / 4);
    SKY_LAUNCH_CHECK("skyemb_standardise");
    return 0;
}

extern "C" int skyemb_weighted_norms(const float *x, const float *w, float *norms, float *xw_out, int64_t N, int D,
                                     void *stream) {
    SKY_CHECK_ARG(N > 0 && D > 0 && D % 4 == 0, "skyemb_weighted_norms: bad shape");
    if (N <= 16 && (size_t)(N + 1) * (D + 4) * 4 <= 64 * 1024 && aligned16(x) && (!w || aligned16(w)) && (!xw_out || aligned16(xw_out))) {
        const size_t lds = (size_t)(N + 1) * (D + 4) * 4;
        hipStream_t st = (hipStream_t)stream;
        if (w && xw_out) hipLaunchKernelGGL((wnorm_rows_kernel<true, true>), dim3(1), dim3(256), lds, st, x, w, norms, xw_out, N, D);
        else if (w) hipLaunchKernelGGL((wnorm_rows_kernel<true, false>), dim3(1), dim3(256), lds, st, x, w, norms, xw_out, N, D);
        else if (xw_out) hipLaunchKernelGGL((wnorm_rows_kernel<false, true>), dim3(1), dim3(256), lds, st, x, w, norms, xw_out, N, D);
        else hipLaunchKernelGGL((wnorm_rows_kernel<false, false>), dim3(1), dim3(256), lds, st, x, w, norms, xw_out, N, D);
        SKY_LAUNCH_CHECK("skyemb_weighted_norms");
        return 0;
    }
    hipLaunchKernelGGL(wnorm_kernel, dim3((unsigned)ceil_div64(N, 256)), dim3(256), 0, (hipStream_t)stream, x, w, norms,
                       xw_out, N, D);
    SKY_LAUNCH_CHECK("skyemb_weighted_norms");
    return 0;
}

// topk_stream.hip: bank-streaming variant for Q <= 16
bool skyemb_topk_stream_applicable(int Q, int D, int k);
int skyemb_topk_stream_lists(int64_t N, int D, int k);
int skyemb_topk_stream_launch(const float *tw, const float *qn, const float *bank, const float *xn, int Q, int64_t N, int D,
                              int k, float eps, int64_t idx_offset, int nlists, float *part_s, int64_t *part_i,
                              const float *thr0, hipStream_t st);
bool skyemb_scores_stream_applicable(int Q, int64_t N, int D);
int skyemb_scores_stream_launch(const float *tw, const float *qn, const float *bank, const float *xn, int Q, int64_t N, int D,
                                float eps, float *scores, hipStream_t st);
static bool use_stream(int Q, int D, int k) {
    static const bool on = []() { const char *e = getenv("SKYEMB_TOPK_STREAM"); return !(e && e[0] == '0'); }();
    return on && skyemb_topk_stream_applicable(Q, D, k);
}

extern "C" int skyemb_cosine_topk_chunks(int64_t N, int Q, int D, int k) {
    if (use_stream(Q, D, k)) return skyemb_topk_stream_lists(N, D, k);
    const bool small = small_tile(Q, k);
    const int QT = small ? 16 : 64, BT = small ? 256 : (wide_tile() ? 256 : 128);
    const int qtiles = (Q + QT - 1) / QT;
    int64_t want = 1024 / qtiles;
    if (want < 1) want = 1;
    // keep chunks long enough that the top-k fill (k inserts) is amortised
    const int64_t max_by_rows = ceil_div64(N, (int64_t)BT * 8);
    if (want > max_by_rows) want = max_by_rows;
    if (want < 1) want = 1;
    return (int)want;
}

extern "C" int skyemb_cosine_topk(const float *tw, const float *qn, const float *bank, const float *xn, int Q, int64_t N,
                                  int D, int k, float eps, int64_t idx_offset, int nchunks, const float *thr0, float *part_s,
                                  int64_t *part_i, void *stream) {
    SKY_CHECK_ARG(Q > 0 && N > 0 && D > 0 && D % 4 == 0 && k > 0 && nchunks > 0, "skyemb_cosine_topk: bad shape");
    SKY_CHECK_ARG(N < (1ll << 31), "skyemb_cosine_topk: shard too large (N < 2^31 rows per call)");
    hipStream_t st = (hipStream_t)stream;
    if (use_stream(Q, D, k)) {
        SKY_CHECK_ARG(nchunks == skyemb_topk_stream_lists(N, D, k), "skyemb_cosine_topk: nchunks must come from skyemb_cosine_topk_chunks");
        return skyemb_topk_stream_launch(tw, qn, bank, xn, Q, N, D, k, eps, idx_offset, nchunks, part_s, part_i, thr0, st);
    }
    if (small_tile(Q, k))
        return launch_topk<16, 256, false>(tw, qn, bank, xn, Q, N, D, k, eps, idx_offset, nchunks, part_s, part_i, nullptr,
                                           st, "skyemb_cosine_topk", thr0);
    if (wide_tile())
        return launch_topk<64, 256, false, 8>(tw, qn, bank, xn, Q, N, D, k, eps, idx_offset, nchunks, part_s, part_i, nullptr,
                                              st, "skyemb_cosine_topk", thr0);
    return launch_topk<64, 128, false>(tw, qn, bank, xn, Q, N, D, k, eps, idx_offset, nchunks, part_s, part_i, nullptr, st,
                                       "skyemb_cosine_topk", thr0);
}

extern "C" int skyemb_cosine_scores(const float *tw, const float *qn, const float *bank, const float *xn, int Q, int64_t N,
                                    int D, float eps, float *scores, void *stream) {
    SKY_CHECK_ARG(Q > 0 && N > 0 && D > 0 && D % 4 == 0, "skyemb_cosine_scores: bad shape");
    hipStream_t st = (hipStream_t)stream;
    const int nchunks = (int)ceil_div64(N, 128 * 8) < 1 ? 1 : (int)ceil_div64(N, 128 * 8);
    if (use_stream(Q, D, 1) && skyemb_scores_stream_applicable(Q, N, D))
        return skyemb_scores_stream_launch(tw, qn, bank, xn, Q, N, D, eps, scores, st);
    if (Q <= 16)   // one 256-row tile per workgroup: a 25,600-row sample gives 100 workgroups (was 25: a quarter of the chip's CUs)
        return launch_topk<16, 256, true>(tw, qn, bank, xn, Q, N, D, 1, eps, 0, (int)ceil_div64(N, 256), nullptr,
                                          nullptr, scores, st, "skyemb_cosine_scores");
    return launch_topk<64, 128, true>(tw, qn, bank, xn, Q, N, D, 1, eps, 0, nchunks, nullptr, nullptr, scores, st,
                                      "skyemb_cosine_scores");
}


// The pruning floor of a small-Q search in two launches: exact scores of the row sample, reduced on the fly to the MAXIMUM of
// every 16-row tile (the k-th largest of those maxima is the score of at least k different rows, hence a lower bound of the k-th
// best score over any bank that contains the sample -- and with a few hundred tiles per wanted row almost the sample's own k-th
// best), then the one-wave selection over S / 16 values per query.  ws: Q * ceil(S / 16) floats.
int skyemb_scores_stream_tilemax_launch(const float *tw, const float *qn, const float *bank, const float *xn, int Q, int64_t N, int D,
                                        float eps, float *tile_max, hipStream_t st);
extern "C" int skyemb_cosine_sample_floor_applicable(int Q, int64_t S, int D, int k) {
    if (Q <= 0 || S <= 0 || D <= 0 || k < 1) return 0;
    const int64_t tiles = (S + 15) / 16;
    return (use_stream(Q, D, 1) && skyemb_scores_stream_applicable(Q, S, D) && tiles <= 2048 && k <= tiles) ? 1 : 0;
}

extern "C" int skyemb_cosine_sample_floor(const float *tw, const float *qn, const float *sample, const float *sample_norms, int Q,
                                          int64_t S, int D, int k, float eps, float *ws, float *floor_out, void *stream) {
    SKY_CHECK_ARG(tw && qn && sample && sample_norms && ws && floor_out && Q > 0 && S > 0 && D > 0 && k >= 1,
                  "skyemb_cosine_sample_floor: bad arguments");
    const int64_t tiles = (S + 15) / 16;
    SKY_CHECK_ARG(skyemb_cosine_sample_floor_applicable(Q, S, D, k),
                  "skyemb_cosine_sample_floor: needs Q <= 16, D %% 64 == 0, D <= 1024, k <= S / 16 <= 2048 and the streaming kernels "
                  "enabled (SKYEMB_TOPK_STREAM) (Q=%d D=%d S=%lld k=%d)", Q, D, (long long)S, k);
    // the streaming scorer reads 16 bytes per lane: rows must start on 16-byte boundaries (D % 64 == 0 keeps the pitch aligned)
    SKY_CHECK_ARG(((uintptr_t)sample & 15) == 0 && ((uintptr_t)tw & 15) == 0,
                  "skyemb_cosine_sample_floor: sample and tw must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const int rc = skyemb_scores_stream_tilemax_launch(tw, qn, sample, sample_norms, Q, S, D, eps, ws, st);
    if (rc != 0) return rc;
    hipLaunchKernelGGL(kth_floor_wave_kernel, dim3(Q), dim3(1024), 0, st, ws, (int)tiles, k, floor_out);
    SKY_LAUNCH_CHECK("skyemb_cosine_sample_floor");
    return 0;
}

extern "C" int skyemb_kth_largest_floor(const float *x, int Q, int S, int k, float *out, void *stream) {
    SKY_CHECK_ARG(x && out && Q > 0 && S > 0 && k >= 1 && k <= S, "skyemb_kth_largest_floor: bad arguments (Q=%d S=%d k=%d)", Q, S, k);
    if (S <= 2048) hipLaunchKernelGGL(kth_floor_wave_kernel, dim3(Q), dim3(1024), 0, (hipStream_t)stream, x, S, k, out);
    else if (S <= 32 * 1024) hipLaunchKernelGGL(kth_floor_bisect_kernel, dim3(Q), dim3(1024), 0, (hipStream_t)stream, x, S, k, out);
    else hipLaunchKernelGGL(kth_floor_kernel, dim3(Q), dim3(1024), 0, (hipStream_t)stream, x, S, k, out);
    SKY_LAUNCH_CHECK("skyemb_kth_largest_floor");
    return 0;
}

extern "C" int skyemb_topk_merge(const float *in_s, const int64_t *in_i, int Q, int nlists, int k, float *out_s,
                                 int64_t *out_i, void *ws, void *stream) {
    SKY_CHECK_ARG(Q > 0 && nlists > 0 && k > 0, "skyemb_topk_merge: bad shape");
    const size_t smem = ((nlists * 4 + 15) & ~15) + (size_t)nlists * 8 + (size_t)nlists * 4;
    SKY_CHECK_ARG(smem <= 65536, "skyemb_topk_merge: too many lists (%d)", nlists);
    if (nlists > 32 && ws != nullptr) {
        // many short lists (bank-streaming kernel): gather + block bitonic sort; per-query fallback to the tournament
        static bool attr_set = false;
        if (!attr_set) {
            hipError_t e = hipFuncSetAttribute((const void *)topk_merge_sort_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                               (MERGE_CAP + 512) * 8);
            if (e != hipSuccess) {
                skyemb_set_error("skyemb_topk_merge: hipFuncSetAttribute: %s", hipGetErrorString(e));
                return 2;
            }
            attr_set = true;
        }
        hipLaunchKernelGGL(topk_merge_sort_kernel, dim3(Q), dim3(1024), (size_t)(MERGE_CAP + 512) * 8, (hipStream_t)stream, in_s, in_i,
                           nlists, k, out_s, out_i);

        SKY_LAUNCH_CHECK("skyemb_topk_merge");
        return 0;
    }
    hipLaunchKernelGGL(topk_merge_kernel, dim3(Q), dim3(64), smem, (hipStream_t)stream, in_s, in_i, nlists, k, out_s, out_i);
    SKY_LAUNCH_CHECK("skyemb_topk_merge");
    return 0;
}
