// Attention core for short sequences (N <= 32 tokens) in bf16: ONE wavefront per 32-row tile of one head -- one sample, or
// 32 / N consecutive samples when N <= 16 (their token rows are consecutive rows of the qkv matrix; score entries pairing
// tokens of different samples are masked) -- every product on v_mfma_f32_32x32x16_bf16, operands loaded straight from HBM
// into MFMA fragments (16-byte row loads).  The operands that are contracted over the TOKEN index are the same rows
// transposed: the wave parks its row fragments in LDS and reads them back with ds_read_b64_tr_b16 (two per fragment).
//
// Orientation trick (cdna_hip_programming.md, "an accumulator tile as the next MFMA's operand"): a 32x32 result has its
// column on the lane and its rows in the 16 registers, so it is directly the operand of a product that sums over its
// ROW index; element e of lane half g of k-step s is row  pi(s,g,e) = 16 s + 8 (e>>2) + 4 g + (e&3).
//   forward : ST[j][i] = K Q^T  (softmax over registers + one lane^32 exchange)  ->  O^T = V^T . P^T
//   backward: both orientations of the scores and of dP are formed (4 cheap MFMAs instead of a transpose):
//             dQ^T = K^T . dS^T (sum over j),   dK^T = Q^T . dS,   dV^T = dO^T . P  (sum over i)
// qkv layout is the reference's [B, N, 3, H, hd] (timm Attention: qkv(x).reshape(B,N,3,H,hd)); softmax statistics,
// probabilities and dS are fp32, rounded to bf16 only as MFMA operands.
#include <mutex>
#include "lp_twin.h"
#include <stdlib.h>

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ __forceinline__ f32x16 mfma32(lp8 a, lp8 b, f32x16 c) {
    return sky_mfma_32x32x16(a, b, c);
}
__device__ __forceinline__ int pi_row(int s, int g, int e) { return 16 * s + 8 * (e >> 2) + 4 * g + (e & 3); }
__device__ __forceinline__ int acc_row(int reg, int g) { return (reg & 3) + 8 * (reg >> 2) + 4 * g; }

__device__ __forceinline__ lp8 zero8() {
    lp8 z;
#pragma unroll
    for (int e = 0; e < 8; ++e) z[e] = (lp_t)0.0f;
    return z;
}
// row fragment: lane (r, g) holds X[row r][16 s + 8 g + 0..7]  (the A and the B operand maps coincide)
__device__ __forceinline__ lp8 row_frag(const lp_t *base, int64_t row_stride, int r, int g, int s, int N) {
    return r < N ? *(const lp8 *)(base + (int64_t)r * row_stride + 16 * s + 8 * g) : zero8();
}
__device__ __forceinline__ lp8 pack_regs(const f32x16 &x, int s) {
    lp8 f;
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = (lp_t)x[8 * s + e];
    return f;
}

typedef __attribute__((address_space(3))) lp4 lds_lp4_t;
// a wave's [32 tokens][HD] operand parked in LDS, rows PITCH = HD + 8 elements apart (16 bytes of padding spread the four
// token rows of a transposing read over the banks)
template <int HD>
__device__ __forceinline__ void park_rows(lp_t *tile, const lp8 (&f)[HD / 16], int r, int g) {
#pragma unroll
    for (int s = 0; s < HD / 16; ++s) *(lp8 *)(tile + r * (HD + 8) + 16 * s + 8 * g) = f[s];
}
// transposed fragment out of a parked operand: lane (r, g) gets X[token pi(s, g, e)][32 blk + r], e = 0..7 -- two runs of four
// consecutive tokens, each one ds_read_b64_tr_b16 (16 lanes x 8 bytes = 4 token rows x 16 columns, transposed in flight)
template <int HD>
__device__ __forceinline__ lp8 tok_frag_lds(const lp_t *tile, int blk, int r, int g, int s) {
    const int i = r & 15, q = i >> 2, p = i & 3;
    const int col0 = 32 * blk + (r & 16) + 4 * p;
    const int t0 = 16 * s + 4 * g;
    const lp4 lo = sky_ds_read_tr16_b64((lds_lp4_t *)(tile + (t0 + q) * (HD + 8) + col0));
    const lp4 hi = sky_ds_read_tr16_b64((lds_lp4_t *)(tile + (t0 + 8 + q) * (HD + 8) + col0));
    lp8 f;
    f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
    f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
    return f;
}
// store the [d][token] result tile: lane (token r, g) owns d = 8 q + 4 g + 0..3 for q = 0..3
template <int HD>
__device__ __forceinline__ void store_tile(lp_t *dst_row, const f32x16 &t, int blk, int g, float scale) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int d0 = 32 * blk + 8 * q + 4 * g;
        if (d0 < HD) store4<lp_t>(dst_row + d0, t[4 * q] * scale, t[4 * q + 1] * scale, t[4 * q + 2] * scale, t[4 * q + 3] * scale);
    }
}

// Sequences of N <= 16 tokens are PACKED: a wave takes P = 32 / N consecutive samples of one head -- their token rows are
// consecutive rows of the qkv matrix -- as one 32-row tile and masks the score entries that pair tokens of different samples
// (the encoder's 5-token sequences: 6 samples = 30 of 32 rows per MFMA tile instead of 5, a sixth of the waves).
struct PackInfo {
    int b0, h, nrows, inv_n;   // first sample, head, valid rows of the tile, ceil(65536 / N)
};
__device__ __forceinline__ PackInfo pack_of(int head, int B, int N, int H) {
    const int P = N <= 16 ? 32 / N : 1;
    const int pack = head / H;
    PackInfo pi;
    pi.h = head - pack * H;
    pi.b0 = pack * P;
    const int ns = B - pi.b0 < P ? B - pi.b0 : P;
    pi.nrows = ns * N;
    pi.inv_n = (65536 + N - 1) / N;
    return pi;
}
__device__ __forceinline__ int seq_of(int row, const PackInfo &pi) { return (row * pi.inv_n) >> 16; }   // row / N, exact for row < 32

// WPB = waves (independent head tiles) per workgroup: 4 when there are enough tiles for every CU, 1 when there are not (the
// encoder's packed 5-token sequences at B = 256: 516 tiles -- 129 four-wave workgroups left half of the 256 CUs without work)
template <int HD, int WPB = 4>
__global__ __launch_bounds__(64 * WPB) void mha_fwd_mfma_kernel(const lp_t *__restrict__ qkv, lp_t *__restrict__ out, int B,
                                                           int N, int H, int nheads) {
    constexpr int KS = HD / 16, NB = (HD + 31) / 32;
    // transposed operands through LDS (see the file header)
    __shared__ __attribute__((aligned(16))) lp_t park[WPB][32 * (HD + 8)];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int head = blockIdx.x * WPB + wave;
    if (head >= nheads) return;
    const PackInfo pk = pack_of(head, B, N, H);
    const int h = pk.h, NR = pk.nrows;
    const int D = H * HD;
    const int64_t rs = 3 * (int64_t)D;
    const int r = lane & 31, g = lane >> 5;
    const lp_t *qb = qkv + (int64_t)pk.b0 * N * rs + h * HD, *kb = qb + D, *vb = qb + 2 * D;
    const int NS = NR > 16 ? 2 : 1;       // k-steps over the token index
    const int my_seq = r < NR ? seq_of(r, pk) : 0;

    lp8 qf[KS], kf[KS], vt[2][NB];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        qf[s] = row_frag(qb, rs, r, g, s, NR);
        kf[s] = row_frag(kb, rs, r, g, s, NR);
    }
    {
        lp8 vf[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) vf[s] = row_frag(vb, rs, r, g, s, NR);
        park_rows<HD>(park[wave], vf, r, g);
        __builtin_amdgcn_wave_barrier();
    }
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int blk = 0; blk < NB; ++blk) vt[s][blk] = s < NS ? tok_frag_lds<HD>(park[wave], blk, r, g, s) : zero8();

    f32x16 st;
#pragma unroll
    for (int e = 0; e < 16; ++e) st[e] = 0.f;
#pragma unroll
    for (int s = 0; s < KS; ++s) st = mfma32(kf[s], qf[s], st);      // ST[j][i] = sum_d K[j][d] Q[i][d]

    const float scale = rsqrtf((float)HD);
    float mx = -INFINITY;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int j = acc_row(e, g);
        st[e] = (j < NR && seq_of(j, pk) == my_seq) ? st[e] * scale : -INFINITY;
        mx = fmaxf(mx, st[e]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    float sum = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        st[e] = __expf(st[e] - mx);
        sum += st[e];
    }
    sum += __shfl_xor(sum, 32);
    const float inv = 1.0f / sum;
#pragma unroll
    for (int e = 0; e < 16; ++e) st[e] *= inv;

    lp8 pf[2] = {pack_regs(st, 0), pack_regs(st, 1)};
    lp_t *orow = out + ((int64_t)pk.b0 * N + r) * D + h * HD;
#pragma unroll
    for (int blk = 0; blk < NB; ++blk) {
        f32x16 ot;
#pragma unroll
        for (int e = 0; e < 16; ++e) ot[e] = 0.f;
        ot = mfma32(vt[0][blk], pf[0], ot);                           // O^T[d][i] = sum_j V[j][d] P[i][j]
        if (NS > 1) ot = mfma32(vt[1][blk], pf[1], ot);
        if (r < NR) store_tile<HD>(orow, ot, blk, g, 1.0f);
    }
}

template <int HD, int WPB = 4>
__global__ __launch_bounds__(64 * WPB) __attribute__((amdgpu_waves_per_eu(HD == 32 ? 4 : 2))) void mha_bwd_mfma_kernel(const lp_t *__restrict__ qkv, const lp_t *__restrict__ dout,
                                                           lp_t *__restrict__ dqkv, int B, int N, int H, int nheads) {
    constexpr int KS = HD / 16, NB = (HD + 31) / 32;
    __shared__ float stats[WPB][3][32];
    __shared__ __attribute__((aligned(16))) lp_t park[WPB][3][32 * (HD + 8)];      // K, Q, dO of each wave
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int head = blockIdx.x * WPB + wave;
    if (head >= nheads) return;
    const PackInfo pk = pack_of(head, B, N, H);        // packed short sequences: see mha_fwd_mfma_kernel
    const int h = pk.h, NR = pk.nrows;
    const int D = H * HD;
    const int64_t rs = 3 * (int64_t)D;
    const int r = lane & 31, g = lane >> 5;
    const lp_t *qb = qkv + (int64_t)pk.b0 * N * rs + h * HD, *kb = qb + D, *vb = qb + 2 * D;
    const lp_t *ob = dout + (int64_t)pk.b0 * N * D + h * HD;
    const int NS = NR > 16 ? 2 : 1;
    const float scale = rsqrtf((float)HD);
    const int my_seq = r < NR ? seq_of(r, pk) : 0;

    lp8 qf[KS], kf[KS], vf[KS], of[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        qf[s] = row_frag(qb, rs, r, g, s, NR);
        kf[s] = row_frag(kb, rs, r, g, s, NR);
        vf[s] = row_frag(vb, rs, r, g, s, NR);
        of[s] = row_frag(ob, D, r, g, s, NR);
    }
    park_rows<HD>(park[wave][0], kf, r, g);
    park_rows<HD>(park[wave][1], qf, r, g);
    park_rows<HD>(park[wave][2], of, r, g);
    f32x16 st, sn, dpt, dpn;     // scores / dP with (rows j, col i) and with (rows i, col j)
#pragma unroll
    for (int e = 0; e < 16; ++e) st[e] = sn[e] = dpt[e] = dpn[e] = 0.f;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        st = mfma32(kf[s], qf[s], st);
        sn = mfma32(qf[s], kf[s], sn);
        dpt = mfma32(vf[s], of[s], dpt);      // dP^T[j][i] = sum_d V[j][d] dO[i][d]
        dpn = mfma32(of[s], vf[s], dpn);
    }
    // ---- column-i orientation: softmax statistics, P^T, rowsum, dS^T
    bool same[16];                             // register row (token j, or i below) belongs to this lane's sample
    float mx = -INFINITY;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int j = acc_row(e, g);
        same[e] = j < NR && seq_of(j, pk) == my_seq;
        st[e] = same[e] ? st[e] * scale : -INFINITY;
        mx = fmaxf(mx, st[e]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    float sum = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        st[e] = __expf(st[e] - mx);
        sum += st[e];
    }
    sum += __shfl_xor(sum, 32);
    const float inv = 1.0f / sum;
    float rsum = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        st[e] *= inv;
        rsum = fmaf(st[e], dpt[e], rsum);
    }
    rsum += __shfl_xor(rsum, 32);
#pragma unroll
    for (int e = 0; e < 16; ++e) st[e] *= dpt[e] - rsum;             // dS^T[j][i]
    if (g == 0) {
        stats[wave][0][r] = mx;
        stats[wave][1][r] = inv;
        stats[wave][2][r] = rsum;
    }
    __builtin_amdgcn_wave_barrier();
    // ---- column-j orientation: P and dS with the token i in the registers
    f32x16 pn;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int i = acc_row(e, g);
        const float p = (r < NR && same[e]) ? __expf(sn[e] * scale - stats[wave][0][i]) * stats[wave][1][i] : 0.f;
        pn[e] = p;
        sn[e] = p * (dpn[e] - stats[wave][2][i]);                     // dS[i][j]
    }

    lp_t *dq = dqkv + ((int64_t)pk.b0 * N + r) * rs + h * HD, *dk = dq + D, *dv = dq + 2 * D;
    lp8 dst_f[2] = {pack_regs(st, 0), pack_regs(st, 1)};           // dS^T, k = j
    lp8 dsn_f[2] = {pack_regs(sn, 0), pack_regs(sn, 1)};           // dS,   k = i
    lp8 pn_f[2] = {pack_regs(pn, 0), pack_regs(pn, 1)};            // P,    k = i
#pragma unroll
    for (int blk = 0; blk < NB; ++blk) {
        f32x16 tq, tk, tv;
#pragma unroll
        for (int e = 0; e < 16; ++e) tq[e] = tk[e] = tv[e] = 0.f;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            if (s < NS) {
                const lp8 kt = tok_frag_lds<HD>(park[wave][0], blk, r, g, s);
                const lp8 qt = tok_frag_lds<HD>(park[wave][1], blk, r, g, s);
                const lp8 ot = tok_frag_lds<HD>(park[wave][2], blk, r, g, s);
                tq = mfma32(kt, dst_f[s], tq);   // dQ^T[d][i] = sum_j K[j][d] dS[i][j]
                tk = mfma32(qt, dsn_f[s], tk);   // dK^T[d][j] = sum_i Q[i][d] dS[i][j]
                tv = mfma32(ot, pn_f[s], tv);    // dV^T[d][j] = sum_i dO[i][d] P[i][j]
            }
        }
        if (r < NR) {
            store_tile<HD>(dq, tq, blk, g, scale);
            store_tile<HD>(dk, tk, blk, g, scale);
            store_tile<HD>(dv, tv, blk, g, 1.0f);
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------
// 32 < N <= 128 (SimMIM sequences: 65 / 66 tokens): one workgroup per (sample, head), wave s owns token strip
// [32 s, 32 s + 32) both as QUERY strip (softmax statistics, output / dQ) and as KEY strip (dK, dV).  Same orientation
// trick per 32x32 tile; the statistics of every query row travel through LDS between the two phases of backward.
// The head's K, V (forward) or Q, K, V, dO (backward) are staged ONCE in LDS with coalesced 16-byte loads (rows past N
// zero-filled, pitch HD + 8): row fragments come back as ds_read_b128, the token-contracted (transposed) ones as two
// ds_read_b64_tr_b16 -- the first version gathered those 2 bytes at a time from global memory, ~300 loads per lane in
// backward (mim_19: 193 us per layer backward, 49 us forward).
// ---------------------------------------------------------------------------------------------------------------
constexpr int MAX_NT = 4;

__device__ __forceinline__ lp8 row_frag_at(const lp_t *base, int64_t row_stride, int row, int g, int s, int N) {
    return row < N ? *(const lp8 *)(base + (int64_t)row * row_stride + 16 * s + 8 * g) : zero8();
}
__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int e = 0; e < 16; ++e) z[e] = 0.f;
    return z;
}
// [rows][HD] of one head -> LDS tile with pitch HD + 8; rows >= N become zeros
template <int HD>
__device__ __forceinline__ void stage_rows(lp_t *tile, const lp_t *base, int64_t row_stride, int N, int rows, int tid, int nthreads) {
    constexpr int PR = HD / 8;                            // 16-byte pieces per row
    for (int p = tid; p < rows * PR; p += nthreads) {
        const int row = p / PR, c = p - row * PR;
        *(lp8 *)(tile + row * (HD + 8) + 8 * c) = row < N ? *(const lp8 *)(base + (int64_t)row * row_stride + 8 * c) : zero8();
    }
}
// NTILES operands of one head at once: every 16-byte piece of every tile is REQUESTED before the first one is written to LDS.
// (stage_rows tile after tile, a loop of load -> write per piece, was 13 dependent global round trips per thread in the backward
// kernel: most of a workgroup's 20 us at mim_19's 65 tokens, whose matrix work is ~3 k cycles per wave.)
template <int HD, int NT, int NTILES>
__device__ __forceinline__ void stage_tiles(lp_t *const (&tiles)[NTILES], const lp_t *const (&bases)[NTILES],
                                            const int64_t (&strides)[NTILES], int N, int rows, int tid) {
    constexpr int PR = HD / 8, THREADS = 64 * NT;
    constexpr int MAXP = ((32 * NT + 4) * PR + THREADS - 1) / THREADS;       // pieces per thread and tile at the longest sequence
    lp8 v[NTILES][MAXP];
#pragma unroll
    for (int t = 0; t < NTILES; ++t)
#pragma unroll
        for (int i = 0; i < MAXP; ++i) {
            const int p = tid + i * THREADS, row = p / PR, c = p - row * PR;
            const int rc = row < N ? row : N - 1;                                // clamped: the request is unconditional
            v[t][i] = *(const lp8 *)(bases[t] + (int64_t)rc * strides[t] + 8 * c);
        }
#pragma unroll
    for (int t = 0; t < NTILES; ++t)
#pragma unroll
        for (int i = 0; i < MAXP; ++i) {
            const int p = tid + i * THREADS, row = p / PR, c = p - row * PR;
            if (p < rows * PR) *(lp8 *)(tiles[t] + row * (HD + 8) + 8 * c) = row < N ? v[t][i] : zero8();
        }
}
// The LDS tiles hold round_up(N, 4) + 4 rows: the data rows, zero-filled up to the 4-row group, and one all-zero group
// (rows zrow..zrow+3) that every read past the sequence is redirected to -- 72 rows instead of 96 for the 65-token
// sequences: with the register budget capped for three waves per SIMD (below) three backward workgroups share a CU.
struct StripRows {
    int n4, zrow;      // round_up(N, 4); first row of the zero group
};
template <int HD>
__device__ __forceinline__ lp8 row_frag_lds(const lp_t *tile, int row, int g, int s, const StripRows &sr) {
    row = row < sr.n4 ? row : sr.zrow;
    return *(const lp8 *)(tile + row * (HD + 8) + 16 * s + 8 * g);
}
// tok_frag_lds for tokens row0 + pi(s, g, e): the two 4-row groups it reads are redirected as wholes
template <int HD>
__device__ __forceinline__ lp8 tok_frag_strip(const lp_t *tile, int row0, int blk, int r, int g, int s, const StripRows &sr) {
    const int i = r & 15, q = i >> 2, p = i & 3;
    const int col0 = 32 * blk + (r & 16) + 4 * p;
    int g0 = row0 + 16 * s + 4 * g, g1 = g0 + 8;
    g0 = g0 < sr.n4 ? g0 : sr.zrow;
    g1 = g1 < sr.n4 ? g1 : sr.zrow;
    const lp4 lo = sky_ds_read_tr16_b64((lds_lp4_t *)(tile + (g0 + q) * (HD + 8) + col0));
    const lp4 hi = sky_ds_read_tr16_b64((lds_lp4_t *)(tile + (g1 + q) * (HD + 8) + col0));
    lp8 f;
    f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
    f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
    return f;
}

extern __shared__ __attribute__((aligned(16))) lp_t strip_lds[];

template <int HD, int NT>
__global__ __launch_bounds__(64 * NT) __attribute__((amdgpu_waves_per_eu(NT <= 3 ? 5 : 3))) void mha_fwd_strip_kernel(const lp_t *__restrict__ qkv, lp_t *__restrict__ out, int B,
                                                                int N, int H) {
    constexpr int KS = HD / 16, NB = (HD + 31) / 32, PITCH = HD + 8;
    const StripRows sr = {(N + 3) & ~3, (N + 3) & ~3};
    const int ROWS = sr.n4 + 4;
    lp_t *kt = strip_lds, *vt = strip_lds + ROWS * PITCH;
    const int lane = threadIdx.x & 63, strip = threadIdx.x >> 6;
    const int b = blockIdx.x / H, h = blockIdx.x - b * H;
    const int D = H * HD;
    const int64_t rs = 3 * (int64_t)D;
    const int r = lane & 31, g = lane >> 5;
    const lp_t *qb = qkv + (int64_t)b * N * rs + h * HD, *kb = qb + D, *vb = qb + 2 * D;
    const int my = 32 * strip + r;                        // this lane's query token
    const float scale = rsqrtf((float)HD);

    lp8 qf[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) qf[s] = row_frag_at(qb, rs, my, g, s, N);
    {
        lp_t *const tiles[2] = {kt, vt};
        const lp_t *const bases[2] = {kb, vb};
        const int64_t strides[2] = {rs, rs};
        stage_tiles<HD, NT, 2>(tiles, bases, strides, N, ROWS, threadIdx.x);
    }
    __syncthreads();
    f32x16 st[NT];
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        st[t] = zero16();
        if (32 * t < N) {
#pragma unroll
            for (int s = 0; s < KS; ++s) st[t] = mfma32(row_frag_lds<HD>(kt, 32 * t + r, g, s, sr), qf[s], st[t]);   // ST[j][i]
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            st[t][e] = 32 * t + acc_row(e, g) < N ? st[t][e] * scale : -INFINITY;
            mx = fmaxf(mx, st[t][e]);
        }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            st[t][e] = __expf(st[t][e] - mx);
            sum += st[t][e];
        }
    sum += __shfl_xor(sum, 32);
    const float inv = 1.0f / sum;
    lp_t *orow = out + ((int64_t)b * N + my) * D + h * HD;
#pragma unroll
    for (int blk = 0; blk < NB; ++blk) {
        f32x16 ot = zero16();
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int s = 0; s < 2; ++s)
                if (32 * t + 16 * s < N) {
                    lp8 pf;
#pragma unroll
                    for (int e = 0; e < 8; ++e) pf[e] = (lp_t)(st[t][8 * s + e] * inv);
                    ot = mfma32(tok_frag_strip<HD>(vt, 32 * t, blk, r, g, s, sr), pf, ot);   // O^T[d][i] += V[j][d] P[i][j]
                }
        if (my < N) store_tile<HD>(orow, ot, blk, g, 1.0f);
    }
}

template <int HD, int NT>
__global__ __launch_bounds__(64 * NT) __attribute__((amdgpu_waves_per_eu(NT <= 3 ? 3 : 2))) void mha_bwd_strip_kernel(const lp_t *__restrict__ qkv, const lp_t *__restrict__ dout,
                                                                lp_t *__restrict__ dqkv, int B, int N, int H) {
    constexpr int KS = HD / 16, NB = (HD + 31) / 32, PITCH = HD + 8;
    __shared__ float stats[3][32 * MAX_NT];               // per query token: softmax max, 1 / sum, rowsum(P dP)
    const StripRows sr = {(N + 3) & ~3, (N + 3) & ~3};
    const int ROWS = sr.n4 + 4;
    lp_t *qt = strip_lds, *kt = qt + ROWS * PITCH, *vt = kt + ROWS * PITCH, *dot = vt + ROWS * PITCH;
    const int lane = threadIdx.x & 63, strip = threadIdx.x >> 6;
    const int b = blockIdx.x / H, h = blockIdx.x - b * H;
    const int D = H * HD;
    const int64_t rs = 3 * (int64_t)D;
    const int r = lane & 31, g = lane >> 5;
    const lp_t *qb = qkv + (int64_t)b * N * rs + h * HD, *kb = qb + D, *vb = qb + 2 * D;
    const lp_t *ob = dout + (int64_t)b * N * D + h * HD;
    const int my = 32 * strip + r;
    const float scale = rsqrtf((float)HD);
    lp_t *dq = dqkv + ((int64_t)b * N + my) * rs + h * HD, *dk = dq + D, *dv = dq + 2 * D;

    {
        lp_t *const tiles[4] = {qt, kt, vt, dot};
        const lp_t *const bases[4] = {qb, kb, vb, ob};
        const int64_t strides[4] = {rs, rs, rs, (int64_t)D};
        stage_tiles<HD, NT, 4>(tiles, bases, strides, N, ROWS, threadIdx.x);
    }
    __syncthreads();

    // ---- phase A: my query strip against every key tile: statistics, dS^T, dQ
    {
        lp8 qf[KS], of[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            qf[s] = row_frag_lds<HD>(qt, my, g, s, sr);
            of[s] = row_frag_lds<HD>(dot, my, g, s, sr);
        }
        f32x16 st[NT], dpt[NT];
        float mx = -INFINITY;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            st[t] = zero16();
            dpt[t] = zero16();
            if (32 * t < N) {
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    st[t] = mfma32(row_frag_lds<HD>(kt, 32 * t + r, g, s, sr), qf[s], st[t]);
                    dpt[t] = mfma32(row_frag_lds<HD>(vt, 32 * t + r, g, s, sr), of[s], dpt[t]);   // dP^T[j][i]
                }
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                st[t][e] = 32 * t + acc_row(e, g) < N ? st[t][e] * scale : -INFINITY;
                mx = fmaxf(mx, st[t][e]);
            }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                st[t][e] = __expf(st[t][e] - mx);
                sum += st[t][e];
            }
        sum += __shfl_xor(sum, 32);
        const float inv = 1.0f / sum;
        float rsum = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                st[t][e] *= inv;
                rsum = fmaf(st[t][e], dpt[t][e], rsum);
            }
        rsum += __shfl_xor(rsum, 32);
        if (g == 0) {
            stats[0][my] = mx;
            stats[1][my] = inv;
            stats[2][my] = rsum;
        }
#pragma unroll
        for (int blk = 0; blk < NB; ++blk) {
            f32x16 tq = zero16();
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int s = 0; s < 2; ++s)
                    if (32 * t + 16 * s < N) {
                        lp8 df;
#pragma unroll
                        for (int e = 0; e < 8; ++e) df[e] = (lp_t)(st[t][8 * s + e] * (dpt[t][8 * s + e] - rsum));   // dS^T[j][i]
                        tq = mfma32(tok_frag_strip<HD>(kt, 32 * t, blk, r, g, s, sr), df, tq);   // dQ^T[d][i] += K[j][d] dS[i][j]
                    }
            if (my < N) store_tile<HD>(dq, tq, blk, g, scale);
        }
    }
    __syncthreads();
    // ---- phase B: my key strip against every query tile: P, dS (tokens i in the registers), dK, dV
    {
        lp8 kf[KS], vf[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            kf[s] = row_frag_lds<HD>(kt, my, g, s, sr);
            vf[s] = row_frag_lds<HD>(vt, my, g, s, sr);
        }
        f32x16 tk[NB], tv[NB];
#pragma unroll
        for (int blk = 0; blk < NB; ++blk) {
            tk[blk] = zero16();
            tv[blk] = zero16();
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (32 * t >= N) break;
            f32x16 sn = zero16(), dpn = zero16();
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                sn = mfma32(row_frag_lds<HD>(qt, 32 * t + r, g, s, sr), kf[s], sn);       // S[i][j]: rows i = 32 t + acc_row, col j = my
                dpn = mfma32(row_frag_lds<HD>(dot, 32 * t + r, g, s, sr), vf[s], dpn);    // dP[i][j]
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int i = 32 * t + acc_row(e, g);
                const bool ok = my < N && i < N;
                const float p = ok ? __expf(sn[e] * scale - stats[0][i]) * stats[1][i] : 0.f;
                dpn[e] = ok ? p * (dpn[e] - stats[2][i]) : 0.f;   // dS[i][j]
                sn[e] = p;
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                if (32 * t + 16 * s >= N) break;
                const lp8 pf = pack_regs(sn, s), df = pack_regs(dpn, s);
#pragma unroll
                for (int blk = 0; blk < NB; ++blk) {
                    tk[blk] = mfma32(tok_frag_strip<HD>(qt, 32 * t, blk, r, g, s, sr), df, tk[blk]);    // dK^T[d][j] += Q[i][d] dS[i][j]
                    tv[blk] = mfma32(tok_frag_strip<HD>(dot, 32 * t, blk, r, g, s, sr), pf, tv[blk]);   // dV^T[d][j] += dO[i][d] P[i][j]
                }
            }
        }
        if (my < N) {
#pragma unroll
            for (int blk = 0; blk < NB; ++blk) {
                store_tile<HD>(dk, tk[blk], blk, g, scale);
                store_tile<HD>(dv, tv[blk], blk, g, 1.0f);
            }
        }
    }
}

template <int HD, int NT>
int launch_strip(bool bwd, const lp_t *x, const lp_t *dout, lp_t *out, int B, int N, int H, hipStream_t st) {
    const dim3 grid(B * H), block(64 * NT);
    const int smem = (bwd ? 4 : 2) * (((N + 3) & ~3) + 4) * (HD + 8) * 2;
    if (smem > 65536) {
        // one flag per instantiation AND device: the dynamic-LDS limit is an attribute of the function per device
        static std::mutex attr_mutex;
        static bool attr_done[64] = {};
        int dev = 0;
        (void)hipGetDevice(&dev);
        std::lock_guard<std::mutex> lock(attr_mutex);
        if (!attr_done[dev & 63]) {
            hipError_t e = hipFuncSetAttribute((const void *)mha_bwd_strip_kernel<HD, NT>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                               4 * (32 * NT + 4) * (HD + 8) * 2);     // the longest sequence of this instance
            if (e != hipSuccess) {
                skyemb_set_error("skyemb_mha: hipFuncSetAttribute: %s", hipGetErrorString(e));
                return 2;
            }
            attr_done[dev & 63] = true;
        }
    }
    if (!bwd) hipLaunchKernelGGL((mha_fwd_strip_kernel<HD, NT>), grid, block, smem, st, x, out, B, N, H);
    else hipLaunchKernelGGL((mha_bwd_strip_kernel<HD, NT>), grid, block, smem, st, x, dout, out, B, N, H);
    return 0;
}

}  // namespace

// returns -1 when the shape is outside this kernel's subset (caller falls back to the LDS kernel of attention.hip)
#ifndef SKY_F16
int skyemb_mha_mfma_try_f16(bool bwd, const void *qkv, const void *dout, void *out, int dtype, int B, int N, int H, int hd, hipStream_t st);
#endif
int SKY_TWIN(skyemb_mha_mfma_try)(bool bwd, const void *qkv, const void *dout, void *out, int dtype, int B, int N, int H, int hd, hipStream_t st) {
#ifndef SKY_F16
    if (dtype == SKYEMB_F16) return skyemb_mha_mfma_try_f16(bwd, qkv, dout, out, dtype, B, N, H, hd, st);
#endif
    if (dtype != SKY_LP_DTYPE) return -1;
    if (N > 32 * MAX_NT || (hd != 32 && hd != 64)) return -1;
    static const bool off = []() { const char *e = getenv("SKYEMB_MHA_MFMA"); return e && e[0] == '0'; }();
    if (off) return -1;
    const lp_t *x = (const lp_t *)qkv;
    if (N > 32) {
        const int nt = (N + 31) / 32;
#define STRIP(HD_, NT_) launch_strip<HD_, NT_>(bwd, x, (const lp_t *)dout, (lp_t *)out, B, N, H, st)
        int rc;
        if (hd == 32) rc = nt == 2 ? STRIP(32, 2) : nt == 3 ? STRIP(32, 3) : STRIP(32, 4);
        else rc = nt == 2 ? STRIP(64, 2) : nt == 3 ? STRIP(64, 3) : STRIP(64, 4);
#undef STRIP
        return rc;
    }
    const int P = N <= 16 ? 32 / N : 1;                   // samples packed into one wave's 32-row tile
    const int nheads = ((B + P - 1) / P) * H;
    // fewer four-wave workgroups than twice the compute units: one tile per workgroup, so that the tiles spread over every CU
    static const int wpb_env = []() { const char *e = getenv("SKYEMB_MHA_WPB"); return e ? atoi(e) : 0; }();
    const bool one = wpb_env ? wpb_env == 1 : (nheads + 3) / 4 < 512;
    const dim3 grid(one ? nheads : (nheads + 3) / 4), block(one ? 64 : 256);
#define MHA_GO(KERN, HD_, ...)                                                                         \
    do {                                                                                               \
        if (one) hipLaunchKernelGGL((KERN<HD_, 1>), grid, block, 0, st, __VA_ARGS__);                  \
        else hipLaunchKernelGGL((KERN<HD_, 4>), grid, block, 0, st, __VA_ARGS__);                      \
    } while (0)
    if (!bwd) {
        if (hd == 32) MHA_GO(mha_fwd_mfma_kernel, 32, x, (lp_t *)out, B, N, H, nheads);
        else MHA_GO(mha_fwd_mfma_kernel, 64, x, (lp_t *)out, B, N, H, nheads);
    } else {
        if (hd == 32) MHA_GO(mha_bwd_mfma_kernel, 32, x, (const lp_t *)dout, (lp_t *)out, B, N, H, nheads);
        else MHA_GO(mha_bwd_mfma_kernel, 64, x, (const lp_t *)dout, (lp_t *)out, B, N, H, nheads);
    }
#undef MHA_GO
    return 0;
}
