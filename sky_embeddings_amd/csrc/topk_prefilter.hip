// Many-query weighted-cosine top-k in two stages (utils/similarity.py:18-35,149-172; contract: oracle/topk_oracle.c):
//
//   stage 1 (prefilter)  approximate dot products on the fp16 matrix cores with a PROVEN error bound,
//                        every (query, row) pair whose score could still reach the query's top-k is appended to
//                        the query's candidate list;
//   stage 2 (exact)      the few survivors are re-scored with the contract's fp32 fma chain and ordered
//                        (score desc, index asc): the result is bit-identical to the exact kernel's.
//
// The exact-fp32 kernel (topk.hip) is bound by the fp32 matrix rate (157 TFLOP/s: 15.4 TFLOP at Q = 10k over 1M x 768);
// fp16 operands run 16x faster per MFMA and halve the bank bytes.
//
// Arithmetic of stage 1.  Row i of the bank is stored as x^_i = fp16(x_i * 2^-e_i) (e_i puts the row maximum in
// [2^13, 2^14)); the weighted query tw_q = w * t_q as q' = tw_q * 2^-f_q split into qh = fp16(q'), ql = fp16(q' - qh).
//   dot^ = sum_d (qh_d + ql_d) * x^_d          two v_mfma_f32_16x16x32_f16 passes into one fp32 accumulator
// Products of two fp16 numbers are exact in fp32.  Against the contract's chain dot' = chain(tw, x) * 2^-(e+f):
//   |dot^ - dot'| <= eps_a * ||q'||_2 * ||x'||_2,
//   eps_a = 2^-11 (fp16 rounding of x', relative) + 2^-21 (residual of the query split and the cross term)
//         + 6.06 * D * 2^-24 (fp32 accumulation of 2D products in any order, priced at one ulp each in case the matrix
//           core truncates, + the chain's own D roundings)
//         + 2 * sqrt(D) * 2^-27 (fp16 subnormals of either operand, even if the matrix core flushes them to zero: absolute
//           errors <= 2^-14 per element, expressed through the row / query maxima >= 2^13).
// With den = fmaf(qn, xn, eps) > 0 the exact score e = dot' * 2^(e+f) / den therefore lies in
//   [L, U] = (dot^ -+ eps_a ||q'|| ||x'||) * 2^(e+f) / den   (widened by 2^-22 relative for the divisions).
// A threshold tau_q that is a lower bound of the query's true k-th best exact score makes "U >= tau_q" a necessary
// condition for membership in the top-k.  tau_q is the k-th largest L over the candidates found so far -- real rows, so
// always a valid bound -- and is refreshed between the phases of the pass (bank slices of 1/32, 3/32, 12/32, 16/32 of
// the tiles after a first slice that takes everything): the candidate lists stay at a few hundred entries per query.
// At the end the K' candidates with the largest U are re-scored exactly; the answer is accepted when its k-th best
// exact score beats the U of every candidate that was NOT re-scored, otherwise the query is flagged and the caller
// runs it through the exact kernel (never observed on embedding-like data; forced in the tests).
#include "common.h"
#include <math.h>
#include <stdlib.h>
#include <mutex>

namespace {

typedef _Float16 half_t;
typedef __attribute__((ext_vector_type(8))) _Float16 half8;
typedef __attribute__((address_space(1))) const void gvoid_t;
typedef __attribute__((address_space(3))) void lvoid_t;

constexpr int BT = 256, BK = 32, NW = 8, NT = NW * 64;   // bank rows per tile, k-step, waves
// queries per tile: 256 with one fp16 pass (hi), 128 with two (hi + lo) -- 16 KiB of query rows per k-step either way, but the
// 256 x 256 tile does twice the products per byte brought into LDS (the LDS-DMA rate, not the matrix cores, bounds the kernel)
constexpr int qtile(bool lo) { return lo ? 128 : 256; }
constexpr int QPAD = 256;                                // every per-query array is padded to this many rows
constexpr int B_BYTES = BT * BK * 2, Q_BYTES = 256 * BK * 2;
constexpr int stage_bytes(bool) { return Q_BYTES + B_BYTES; }            // query rows (hi, or hi + lo) + bank rows = 32 KiB
constexpr int pieces(bool) { return (Q_BYTES + B_BYTES) / 1024 / NW; }   // LDS-DMA instructions per wave per stage (one = 16 rows x 64 B)
constexpr int NSTAGE = 4, AHEAD = NSTAGE - 1;   // 96 KiB of LDS-DMA in flight per CU: one workgroup per CU has to cover the
                                                 // L2 latency alone (a 2 x 64 KiB ring ran at 23 GB/s per CU: 2.8 us per k-step);
                                                 // NSTAGE is a power of two: stage s lives in buffer s & (NSTAGE - 1)
constexpr int GQ = 5;                     // query tiles (hi + lo = 384 KiB each at D = 768) kept hot in an XCD's L2
constexpr int RESCORE_MAX = 256;          // K': candidates re-scored exactly per query
constexpr int SCAP = 2048;                // candidates of one item staged in LDS (8 bytes each)
constexpr int PF_GRID = 256;              // persistent workgroups of the stage-1 launch (one per CU)

__device__ __forceinline__ void glds16(const void *src, char *lds_wave_base) {
    __builtin_amdgcn_global_load_lds((gvoid_t *)src, (lvoid_t *)lds_wave_base, 16, 0, 0);
}

// lo: the query enters as hi + lo (see the header); without lo its own fp16 rounding (2^-11 relative, + the cross term) joins
// the row's
__host__ __device__ inline double eps_a_of(int D, bool lo) {
    return (lo ? 0x1p-11 + 0x1p-21 : 0x1p-10 + 0x1p-21) + 6.06 * D * 0x1p-24 + 2.0 * sqrt((double)D) * 0x1p-27;
}

// ---- preparation --------------------------------------------------------------------------------------------------
// one wave per row: scale exponent, fp16 image, ||x'||_2 (rounded up), scaled weighted norm.
// rowp[i] = {xn * 2^-e, ||x'||_2 (1 + D 2^-22), 2^-e, 0}; rows N .. rows_padded-1 (whole tiles of BT rows) hold a sentinel.
// The four components are the row's B operand of the candidate test's v_mfma_f32_16x16x4_f32 (prefilter_kernel, item epilogue):
// the last one multiplies the queries' 0 and must stay finite.
__global__ __launch_bounds__(256) void bank16_kernel(const float *__restrict__ bank, const float *__restrict__ xn, int64_t N, int D,
                                                      half_t *__restrict__ bank16, float4 *__restrict__ rowp, int64_t rows_padded) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= N) {
        // padding up to a whole bank tile: zeros, and ||x'|| = NaN makes the candidate test fail for every query
        if (row < rows_padded) {
            for (int d = lane * 4; d < D; d += 256) *(uint2 *)(bank16 + row * D + d) = make_uint2(0u, 0u);
            if (lane == 0) rowp[row] = make_float4(0.f, NAN, 0.f, 0.f);
        }
        return;
    }
    const float *x = bank + row * D;
    float mx = 0.f;
    for (int d = lane * 4; d < D; d += 256) {
        const float4 v = *(const float4 *)(x + d);
        mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
    mx = wave_max(mx);
    half_t *o = bank16 + row * D;
    int e = 0;
    if (mx > 0.f && mx < INFINITY) {
        (void)frexpf(mx, &e);            // mx = m * 2^e, m in [0.5, 1)
        e -= 14;                         // mx * 2^-e in [2^13, 2^14)
    }
    if (!(mx < INFINITY) || e < -126) {
        // a row with an infinite element cannot be bounded, and neither can one so small that its scale 2^-e would
        // overflow (largest |x| below ~2^-113: the fp16 image would be inf / NaN and the row silently lost): fp16 zeros
        // + ||x'|| = inf make every query keep it as a candidate (U = inf), so the exact stage decides
        for (int d = lane * 4; d < D; d += 256) *(uint2 *)(o + d) = make_uint2(0u, 0u);
        // (the weighted-norm slot is 0, not the row's -- possibly infinite -- norm: the test's fma chain must not meet inf - inf)
        if (lane == 0) rowp[row] = make_float4(0.f, INFINITY, 1.0f, 0.f);
        return;
    }
    const float s = ldexpf(1.0f, -e);
    float ss = 0.f;
    for (int d = lane * 4; d < D; d += 256) {
        const float4 v = *(const float4 *)(x + d);
        const float a = v.x * s, b = v.y * s, c = v.z * s, dd = v.w * s;
        ss = fmaf(a, a, fmaf(b, b, fmaf(c, c, fmaf(dd, dd, ss))));
        typedef __attribute__((ext_vector_type(4))) _Float16 half4;
        half4 h;
        h[0] = (half_t)a; h[1] = (half_t)b; h[2] = (half_t)c; h[3] = (half_t)dd;
        *(half4 *)(o + d) = h;
    }
    ss = wave_sum(ss);
    if (lane == 0) {
        const float nx = sqrtf(ss) * (1.0f + (float)D * 0x1p-22f);
        const float xnv = xn[row];
        rowp[row] = make_float4(xnv * s, nx, s, 0.f);
    }
}

// queries: tw[q] -> qh, ql (fp16, scaled by 2^-f); qbase[q] = {qn * 2^-f, eps_a * ||q'|| (1 + D 2^-22), 2^-f, qn}
__global__ __launch_bounds__(256) void query16_kernel(const float *__restrict__ tw, const float *__restrict__ qn, int Q, int D,
                                                       half_t *__restrict__ qh, half_t *__restrict__ ql, float4 *__restrict__ qbase,
                                                       float eps_a) {
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= Q) {
        // rows up to a whole query tile (the grid covers them): zeros; their test parameters let no pair pass
        if (q < (Q + QPAD - 1) / QPAD * QPAD)
            for (int d = lane * 4; d < D; d += 256) {
                *(uint2 *)(qh + (int64_t)q * D + d) = make_uint2(0u, 0u);
                *(uint2 *)(ql + (int64_t)q * D + d) = make_uint2(0u, 0u);
            }
        return;
    }
    const float *x = tw + (int64_t)q * D;
    float mx = 0.f;
    for (int d = lane * 4; d < D; d += 256) {
        const float4 v = *(const float4 *)(x + d);
        mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
    mx = wave_max(mx);
    int e = 0;
    if (mx > 0.f && mx < INFINITY) {
        (void)frexpf(mx, &e);
        e -= 14;
    }
    if (e < -126) e = -126;             // keep 2^-e finite; such a query (scale == 2^126) is flagged for the exact path by init_state_kernel
    const float s = ldexpf(1.0f, -e);
    float ss = 0.f;
    for (int d = lane * 4; d < D; d += 256) {
        const float4 v = *(const float4 *)(x + d);
        const float a[4] = {v.x * s, v.y * s, v.z * s, v.w * s};
        typedef __attribute__((ext_vector_type(4))) _Float16 half4;
        half4 h, l;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            ss = fmaf(a[j], a[j], ss);
            h[j] = (half_t)a[j];
            l[j] = (half_t)(a[j] - (float)h[j]);      // exact difference in fp32, then one fp16 rounding
        }
        *(half4 *)(qh + (int64_t)q * D + d) = h;
        *(half4 *)(ql + (int64_t)q * D + d) = l;
    }
    ss = wave_sum(ss);
    if (lane == 0) {
        const float nq = sqrtf(ss) * (1.0f + (float)D * 0x1p-22f);
        qbase[q] = make_float4(qn[q] * s, eps_a * nq, s, qn[q]);
    }
}

// ---- stage 1 ------------------------------------------------------------------------------------------------------
// Operand tile (rows x 32 k, fp16): 64-byte rows; 16-byte chunk c of row r is stored at chunk c ^ swz(r), the XOR applied
// to the per-lane SOURCE address (the LDS-DMA destination is wave base + lane * 16).  One instruction = 16 rows.
// swz makes the four 16-lane groups of a ds_read_b128 fragment read ({rows 0-3, 12-15 of chunk c, rows 4-11 of chunk c+1}
// and the like, MI355X_MICROARCH.md LDS table) hit 16 distinct 16-byte slots of the 256-byte bank row.
__device__ __forceinline__ int swz(int r) {
    const int g = (r >> 2) & 3;
    return (((g ^ (g >> 1)) & 1) << 1) | (g >> 1);          // g = 0,1,2,3 -> 0,2,3,1
}
// All three operand arrays are padded to whole tiles (query16_kernel / bank16_kernel), so a piece's address is a
// workgroup-uniform base (tile row, k-step: scalar registers) plus ONE per-lane 32-bit offset that never changes.  The
// instruction is written out (SGPR-base form, LDS destination in M0): left to the compiler, every piece's 64-bit vector
// address was hoisted out of the loop into registers the accumulators and fragments need (spills inside the k-loop).
// the lane index, recomputed where it is used (volatile asm: neither hoisted nor spilled).  In the k-loop's once-per-item paths
// a value kept alive from the kernel's entry gets spilled, and its reload -- a scratch load the compiler waits for with
// vmcnt(0) -- drains the LDS-DMA ring.
__device__ __forceinline__ int lane_now() {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}
__device__ __forceinline__ void glds16_sbase(const void *base_uniform, unsigned int lane_off, char *lds_wave_base) {
    const unsigned int dst = (unsigned int)(uintptr_t)(lvoid_t *)lds_wave_base;
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                 :
                 : "v"(lane_off), "s"(base_uniform), "s"(dst)
                 : "memory", "m0");
}
// fragment of 16 rows from row `rbase` (a multiple of 16: swz(rbase + l) == swz(l)): every fragment of a lane sits at
// rbase * 64 + ONE per-lane constant, frag_lane(lane) -- an immediate offset on a single address register
__device__ __forceinline__ int frag_lane(int lane) { return (lane & 15) * 64 + ((((lane >> 4) ^ swz(lane & 15))) << 4); }
__device__ __forceinline__ half8 frag(const char *sbase, int rbase, int lane_c) {
    return *(const half8 *)(sbase + rbase * 64 + lane_c);
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// The pair (q, i) is a candidate iff  dot^ + c * nx'_i >= a * xn'_i + b * se_i  with
//   a = tau * qn'   b = tau * eps * 2^-f   c = eps_a ||q'||     (a, b lowered and c raised by 2^-19 relative, see below).
// Round 4: the test is ONE matrix instruction per 16 x 16 block of pairs.  qpar[q] = {-a, c, -b, 0} is the query's row of an
// A operand, rowp[i] = {xn', nx', se, 0} the row's column of a B operand, and
//   t = v_mfma_f32_16x16x4_f32(qpar rows, rowp columns, C = dot^ block)  =  dot^ - a xn' + c nx' - b se   (fp32 fma chain)
// lands in scratch registers (the accumulators keep dot^ for the candidate lists); the pair passes iff t >= 0 (NaN fails), one
// compare + one add-with-carry per pair into the lane's bit mask.  Before, every pair cost two fmas, a multiply, a compare and
// the mask update on the vector ALU: 128 pairs per lane and item = 31 % of the pass.
// Roundings: the chain rounds four times, each by <= 2^-24 of a partial sum bounded by S = |dot^| + |a xn'| + |c nx'| + |b se|, and
// a, b carry one or two roundings of their own.  The slack 2^-19 (|a xn'| + |b se| + |c nx'|) =: 2^-19 T1 built into the constants
// covers them: if |dot^| <= 2 T1 the chain's error is <= 2^-22 * 3.01 T1 < 2^-19 T1; if |dot^| > 2 T1 the exact value has the
// sign of dot^ and magnitude > |dot^| / 2, far above the error.  So every pair that satisfies the inequality in exact arithmetic
// (the condition the proof of the file header needs) passes.
// Work items: bank tile t of [t0, t1) x query tile; bank tiles are dealt to the XCDs (t - t0) % 8 == blockIdx.x % 8, and an
// XCD walks its items group-of-GQ-query-tiles outermost, then bank tile, then query tile: the GQ query tiles stay in its
// L2 while its bank tiles stream through once per group.  The LDS ring never drains between items.
// MODE 0: first slice, every pair is stored (slot = row)   1: candidates appended to the per-query lists directly (returning
// global atomics: the short early phases, whose loose thresholds pass several per cent of the pairs)   2: candidates staged
// in LDS and flushed to the workgroup's own region (the long late phases)
template <int MODE, bool LO>
__global__ __launch_bounds__(NT) void prefilter_kernel(const half_t *__restrict__ qh, const half_t *__restrict__ ql,
                                                       const half_t *__restrict__ bank16, const float4 *__restrict__ rowp,
                                                       const float4 *__restrict__ qpar, int Q, int64_t N, int D, int t0, int t1,
                                                       int cap, int *__restrict__ cnt, int *__restrict__ cand_i, float *__restrict__ cand_d,
                                                       uint4 *__restrict__ wg_list, int *__restrict__ wg_count, int capw,
                                                       int *__restrict__ overflow
#if defined(PF_STAMP) || defined(PF_CLOCK)
                                                       , unsigned long long *__restrict__ dbg
#endif
                                                       ) {
    constexpr bool TAKE_ALL = MODE == 0, STAGED = MODE == 2;
    constexpr int QT = qtile(LO), MI = QT / 2 / 16;        // queries per tile; 16-row query fragments per wave (the wave's half of the tile)
    constexpr int STAGE = stage_bytes(LO), NI = pieces(LO), A_BYTES = QT * BK * 2, B_OFF = Q_BYTES;
    extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef PF_STAMP
    unsigned int seg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tprev;
    auto stamp = [&](int which) {
        unsigned long long t;
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
        __builtin_amdgcn_sched_barrier(0);
        if (which >= 0) seg[which] += (unsigned int)(t - tprev);
        tprev = t;
    };
#define STAMP(w) stamp(w)
#else
#define STAMP(w)
#endif
    float4 *spar = (float4 *)(smem + NSTAGE * STAGE);        // [QT] test parameters of the item's queries
    float4 *srow = spar + QT;                                 // [BT] constants of the item's bank rows
    uint2 *stg = (uint2 *)(srow + BT);                        // [SCAP] candidates of the item being finished: {q_local << 8 | row_local, dot^}
    int *stg_n = (int *)(stg + SCAP);                         // their number (may exceed SCAP: the excess was flagged, not stored)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;                 // 2 x 4 waves: QT / 2 queries x 64 rows each
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3, nlocal = gridDim.x >> 3;
    const int T = t1 - t0, Tx = xcd < T ? (T - xcd + 7) / 8 : 0;
    const int nqt = (Q + QT - 1) / QT;
    const int full = nqt / GQ, last = nqt % GQ;
    const int items = Tx * nqt;                               // < 2^31: a shard has < 2^31 rows
    const int KT = D / BK;
    // this workgroup's items: local, local + nlocal, ...
    const int my_items = items > local ? (items - local + nlocal - 1) / nlocal : 0;
    if (my_items == 0) {
        if (MODE == 2 && threadIdx.x == 0) wg_count[blockIdx.x] = 0;
        return;
    }

    auto item_of = [&](int n, int &qt, int &t) {            // n-th item of this workgroup -> (query tile, bank tile)
        const int idx = local + n * nlocal;
        const int per_full = GQ * Tx;
        int g, within, tt;
        if (idx < per_full * full) {
            g = idx / per_full;
            const int rem = idx - g * per_full;
            tt = rem / GQ;
            within = rem - tt * GQ;
        } else {
            g = full;
            const int rem = idx - per_full * full;
            tt = rem / last;
            within = rem - tt * last;
        }
        qt = __builtin_amdgcn_readfirstlane(g * GQ + within);
        t = __builtin_amdgcn_readfirstlane(t0 + tt * 8 + xcd);
    };
    // ---- issue cursor: the stage being requested runs AHEAD k-steps in front of the one being read, across item boundaries.
    // Each wave owns four 1 KiB pieces of a stage: rows [16 wave, +16) of the hi and of the lo query image, rows
    // [32 wave, +32) of the bank tile.  Their addresses are three running scalar pointers (64 bytes further per k-step,
    // re-based when the cursor enters a new item) plus one per-lane offset that never changes: a k-step's address work is a
    // handful of scalar adds (recomputing bases from the tile numbers cost ~80 scalar instructions per k-step and wave).
    static_assert(BT / 16 / NW == 2 && Q_BYTES / 1024 / NW == 2, "piece ownership below: two query and two bank pieces per wave");
    const unsigned int lane_off = (unsigned int)((lane >> 2) * D * 2 + (((lane & 3) ^ swz(lane >> 2)) << 4));
    int n_iss = 0, kt_iss = 0;
    int qt_cur, t_cur, qt_nxt = 0, t_nxt = 0;                // the item being multiplied / the one the issue cursor has entered
    item_of(0, qt_cur, t_cur);
    const char *ph, *pl, *pb;
    auto rebase = [&](int qt, int t) {
        const int64_t qoff = ((int64_t)qt * QT + wave * (QT / NW)) * D * 2;   // LO: 16 rows of hi and of lo; else 32 rows of hi
        ph = (const char *)qh + qoff;
        pl = (const char *)ql + qoff;
        pb = (const char *)bank16 + ((int64_t)t * BT + wave * 32) * D * 2;
    };
    rebase(qt_cur, t_cur);
    const int bank_piece2 = 16 * D * 2;                      // second bank piece: 16 rows further
    // half 0: the two query pieces; half 1: the two bank pieces, then the cursor moves on
    auto issue_half = [&](int buf, int half) {
        char *st = smem + buf * STAGE;
        if (half == 0) {
            glds16_sbase(ph, lane_off, st + wave * (QT / NW) * 64);
            if (LO) glds16_sbase(pl, lane_off, st + A_BYTES + wave * 1024);
            else glds16_sbase(ph + bank_piece2, lane_off, st + wave * 2048 + 1024);      // rows 16..31 of the wave's 32
        } else {
            glds16_sbase(pb, lane_off, st + B_OFF + wave * 2048);
            glds16_sbase(pb + bank_piece2, lane_off, st + B_OFF + wave * 2048 + 1024);
            ph += BK * 2; pl += BK * 2; pb += BK * 2;
            if (++kt_iss == KT) {
                kt_iss = 0;
                if (++n_iss < my_items) {
                    item_of(n_iss, qt_nxt, t_nxt);
                    rebase(qt_nxt, t_nxt);
                }
            }
        }
    };

    f32x4 acc[MI][4];
    half8 fh[MI], fl[LO ? MI : 1], fb[4];
    const bool late = wave >= 4;                             // wave-uniform (SGPR)
    const int steps = my_items * KT;
#pragma unroll
    for (int p = 0; p < AHEAD; ++p)
        if (p < steps) {
            issue_half(p, 0);
            issue_half(p, 1);
        }

    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    };
    const int flane = frag_lane(lane);
    auto read_frags = [&](int b) {
#ifdef PF_NOREAD
        return;
#endif
        // one address register per operand: stage base + the wave's first row + the lane's constant; the rest are immediates
        const char *sa = smem + b * STAGE + wm * (QT / 2) * 64 + flane, *sb = smem + b * STAGE + B_OFF + wn * 64 * 64 + flane;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            fh[i] = frag(sa, i * 16, 0);
            if (LO) fl[i] = frag(sa + A_BYTES, i * 16, 0);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[j] = frag(sb, j * 16, 0);
    };
    // the four LDS-DMA instructions of the next stage cost ~100+ issue cycles each: they go BETWEEN the MFMAs (whose execution
    // covers them), not in front of them
    // two of the wave's four LDS-DMA pieces of a k-step are requested between the MFMAs (whose execution covers their
    // issue cost), the other two in the wave's read phase
    auto multiply = [&](bool issue, int ibuf, int half) {
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            if (i == 1) {
                __builtin_amdgcn_sched_barrier(0);
#ifndef PF_NODMA
                if (issue) issue_half(ibuf, half);
#endif
                __builtin_amdgcn_sched_barrier(0);
            }
#ifndef PF_NOMFMA
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fh[i], fb[j], acc[i][j], 0, 0, 0);
                if (LO) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fl[i], fb[j], acc[i][j], 0, 0, 0);
            }
#else
            acc[i][0][0] += (float)fh[i][0] + (float)fb[i & 3][2];
#endif
        }
    };
    auto item_epilogue = [&](int qt, int t) {
        // ---- epilogue of the item: per 16x16 block a lane holds queries 4*(lane>>4)+r (r = 0..3) x bank row (lane & 15).
        // All 16 MI tests of the lane are branch-free (bit masks mr[r]); only waves that found something enter the append path.
        // (the lane coordinates go through an empty asm: everything addressed from them is then computed HERE, once per item,
        // instead of being hoisted out of the k-loop into registers that the accumulators and fragments need)
        const int ln = lane_now();
        const int l16 = ln & 15, l4 = ln >> 4;
        // B operand of the test: component (lane >> 4) of the constants of bank row (lane & 15) of each 16-row block
        float tb[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) tb[j] = ((const float *)srow)[(wn * 64 + j * 16 + l16) * 4 + l4];
        // mr[r]: the lane's tests of query 4 * (lane >> 4) + r, one bit per (block i, row fragment j): bit (MI - 1 - i) * 4 + j
        unsigned int mr[4] = {0u, 0u, 0u, 0u};
#ifdef PF_NOTEST   // experiment build: the accumulators are consumed, no pair is tested (what the 128 tests per lane and item cost)
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) asm volatile("" ::"v"(acc[i][j]));
#else
        // A operands: component (lane >> 4) of the parameters of query (lane & 15) of each block
        float ta[MI];
#pragma unroll
        for (int i = 0; i < MI; ++i) ta[i] = ((const float *)spar)[(wm * (QT / 2) + i * 16 + l16) * 4 + l4];
        // Software pipeline, written out: one group = the matrix instruction of fragment j of block i + 1, then the eight vector
        // instructions that fold fragment j of block i into the four masks (mr[r] = 2 mr[r] + (t >= 0): compare into a scalar
        // pair, add-with-carry; NaN compares false).  An 8-pass matrix instruction occupies its pipe for the 32 cycles the eight
        // vector instructions take to issue, so both run all the time; the four masks are four independent chains (one mask
        // register for all sixteen pairs of a block was a chain of 32 dependent instructions: 375 cycles per block, stamped).
        // The compiler does not see the matrix instruction inside the asm: the distance from it to the first read of its result
        // (three more groups = 27 instructions) is kept by construction.
        f32x4 tst[2][4];
#pragma unroll
        for (int j = 3; j >= 0; --j) tst[0][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(ta[0], tb[j], acc[0][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
#pragma unroll
        for (int i = 0; i < MI; ++i) {
#pragma unroll
            for (int j = 3; j >= 0; --j) {
                unsigned long long c0, c1, c2, c3;
                if (i + 1 < MI)
                    asm volatile("v_mfma_f32_16x16x4_f32 %0, %9, %10, %11\n\t"
                                 "v_cmp_le_f32_e64 %5, 0, %12\n\tv_cmp_le_f32_e64 %6, 0, %13\n\t"
                                 "v_cmp_le_f32_e64 %7, 0, %14\n\tv_cmp_le_f32_e64 %8, 0, %15\n\t"
                                 "v_addc_co_u32_e64 %1, %5, %1, %1, %5\n\tv_addc_co_u32_e64 %2, %6, %2, %2, %6\n\t"
                                 "v_addc_co_u32_e64 %3, %7, %3, %3, %7\n\tv_addc_co_u32_e64 %4, %8, %4, %4, %8"
                                 : "=&v"(tst[(i + 1) & 1][j]), "+v"(mr[0]), "+v"(mr[1]), "+v"(mr[2]), "+v"(mr[3]), "=&s"(c0), "=&s"(c1),
                                   "=&s"(c2), "=&s"(c3)
                                 : "v"(ta[(i + 1) % MI]), "v"(tb[j]), "v"(acc[(i + 1) % MI][j]), "v"(tst[i & 1][j][0]), "v"(tst[i & 1][j][1]),
                                   "v"(tst[i & 1][j][2]), "v"(tst[i & 1][j][3]));
                else
                    asm volatile("v_cmp_le_f32_e64 %4, 0, %8\n\tv_cmp_le_f32_e64 %5, 0, %9\n\t"
                                 "v_cmp_le_f32_e64 %6, 0, %10\n\tv_cmp_le_f32_e64 %7, 0, %11\n\t"
                                 "v_addc_co_u32_e64 %0, %4, %0, %0, %4\n\tv_addc_co_u32_e64 %1, %5, %1, %1, %5\n\t"
                                 "v_addc_co_u32_e64 %2, %6, %2, %2, %6\n\tv_addc_co_u32_e64 %3, %7, %3, %3, %7"
                                 : "+v"(mr[0]), "+v"(mr[1]), "+v"(mr[2]), "+v"(mr[3]), "=&s"(c0), "=&s"(c1), "=&s"(c2), "=&s"(c3)
                                 : "v"(tst[i & 1][j][0]), "v"(tst[i & 1][j][1]), "v"(tst[i & 1][j][2]), "v"(tst[i & 1][j][3]));
            }
        }
#endif
        STAMP(5);
#if defined(PF_NOMFMA) || defined(PF_NOREAD) || defined(PF_NODMA) || defined(PF_NOAPPEND)
#pragma unroll
        for (int r = 0; r < 4; ++r) {                // experiment builds: the tests are computed, nothing is appended
            asm volatile("" ::"v"(mr[r]));
            mr[r] = 0;
        }
#endif
        auto passed = [&](int i, int r, int j) -> bool { return (mr[r] >> ((MI - 1 - i) * 4 + j)) & 1u; };
        // Appends are rare (a few per wave and item): the walk over the lane's 128 bits is pruned by WAVE-UNIFORM branches (a
        // ballot per block, then per query row of a flagged block); per-lane branches only below those.  (Thirty-two per-lane
        // branches per item, taken or not, cost 2.7 k cycles per item and wave group, stamped.)
        const unsigned int anyr = mr[0] | mr[1] | mr[2] | mr[3];
        auto block_flagged = [&](int i) -> bool { return __builtin_amdgcn_ballot_w64(((anyr >> ((MI - 1 - i) * 4)) & 15u) != 0u) != 0ull; };
        auto row_flagged = [&](int i, int r) -> bool { return __builtin_amdgcn_ballot_w64(((mr[r] >> ((MI - 1 - i) * 4)) & 15u) != 0u) != 0ull; };
        if (TAKE_ALL) {
            // first slice: every (query, row) pair of the slice is a candidate, slot = row within the slice (no counters);
            // pairs that fail even the open test (NaN) are stored as -inf and dropped by the select step
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int q = qt * QT + wm * (QT / 2) + i * 16 + 4 * l4 + r;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int64_t row = (int64_t)t * BT + wn * 64 + j * 16 + l16;
                        if (q < Q && row < N) {
                            const int64_t o = (int64_t)q * cap + (row - (int64_t)t0 * BT);
                            cand_i[o] = (int)row;
                            cand_d[o] = passed(i, r, j) ? acc[i][j][r] : -INFINITY;
                        }
                    }
                }
        } else if (!STAGED) {
#pragma unroll
            for (int i = 0; i < MI; ++i)
                if (block_flagged(i)) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (row_flagged(i, r)) {
                            const int q = qt * QT + wm * (QT / 2) + i * 16 + 4 * l4 + r;
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                if (passed(i, r, j)) {
                                    const int pos = atomicAdd(cnt + q, 1);
                                    if (pos < cap) {
                                        cand_i[(int64_t)q * cap + pos] = t * BT + wn * 64 + j * 16 + l16;
                                        cand_d[(int64_t)q * cap + pos] = acc[i][j][r];
                                    }
                                }
                        }
                }
        } else {
            // Candidates go to a list in LDS (slot from an LDS counter: no global atomic, no wait on the vector-memory
            // counter that the LDS-DMA ring lives on -- a returning global atomic per candidate cost ~750 cycles per k-step).
            // The workgroup flushes the list to its OWN region of wg_list with plain stores (flush_item); bucket_kernel sorts
            // the regions into the per-query lists afterwards.
            const int kbase = ((wm * (QT / 2) + 4 * l4) << 8) | (wn * 64 + l16);
#pragma unroll
            for (int i = 0; i < MI; ++i)
                if (block_flagged(i)) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (row_flagged(i, r)) {
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                if (passed(i, r, j)) {
                                    const int key = kbase + (((i * 16 + r) << 8) | (j * 16));      // q_local << 8 | row_local
                                    const int slot = atomicAdd(stg_n, 1);
                                    if (slot < SCAP) stg[slot] = make_uint2((unsigned int)key, __float_as_uint(acc[i][j][r]));
                                    else overflow[qt * QT + (key >> 8)] = 1;     // more than the staging list holds: the query is re-run exactly
                                }
                        }
                }
        }
        STAMP(6);
    };
    // the staged candidates of item (qt, t) -> this workgroup's region of wg_list (16 bytes each: query, row, dot^)
    int wpos = 0;
    auto flush_item = [&](int qt, int t) {
        const int tid = wave * 64 + lane_now();
        int n = *stg_n;      // (a plain LDS read: a volatile one became a FLAT load, whose vmcnt(0) drained the LDS-DMA ring once per item)
        n = __builtin_amdgcn_readfirstlane(n < SCAP ? n : SCAP);
        uint4 *dst = wg_list + (int64_t)blockIdx.x * capw;
        for (int e = tid; e < n; e += NT) {
            const uint2 c = stg[e];
            const int q = qt * QT + (int)(c.x >> 8), row = t * BT + (int)(c.x & 255u);
#ifdef PF_NOFLUSHSTORE
            asm volatile("" ::"v"(q), "v"(row), "v"(c.y));
            continue;
#endif
            if (wpos + e < capw) dst[wpos + e] = make_uint4((unsigned int)q, (unsigned int)row, c.y, 0u);
            else overflow[q] = 1;                                          // region full: the query is re-run exactly
        }
        wpos = wpos + n < capw ? wpos + n : capw;
    };
    // the k-step of its item that this wave multiplies next / that the current step reads
    int kt_c = 0, kt_s = 0, qt_done = 0, t_done = 0;
    zero_acc();
    if (tid == 0) *stg_n = 0;                                 // (the first append is many barriers away)
    auto multiply_stage = [&](bool issue, int ibuf, int half) {
        multiply(issue, ibuf, half);
        if (++kt_c == KT) {
            STAMP(4);
            item_epilogue(qt_cur, t_cur);
            zero_acc();
            STAMP(7);
            kt_c = 0;
            qt_done = qt_cur;         // (flushed after the next P1, when both groups have finished the item)
            t_done = t_cur;
            qt_cur = qt_nxt;          // (the issue cursor entered the next item AHEAD steps ago, and enters the one after it only later: KT > AHEAD)
            t_cur = t_nxt;
        }
    };
    auto landed = [&](int s) {
        // stage s has landed (this wave's pieces); up to AHEAD - 1 younger stages stay in flight.  (The LDS-DMA of the per-item
        // constants is younger still: it can only make this wait stricter, never weaker.)
        const int rem = steps - 1 - s;
        if (rem >= 2) wait_vmcnt<2 * NI>();
        else if (rem == 1) wait_vmcnt<NI>();
        else wait_vmcnt<0>();
    };
    auto item_constants = [&](int s_now) {
        // after P1 of an item's first step: test parameters of its 128 queries (2 KiB) and constants of its 256 rows (4 KiB) go
        // to LDS by the same DMA path (by then every wave's current item is this one).  The late group's epilogue of the
        // PREVIOUS item (which reads spar / srow) ran between P0 and P1.  Three steps on they are older than everything a
        // counted wait leaves in flight, so they have landed before the item's epilogue (KT >= 4).  Both arrays are padded
        // to whole tiles with entries no pair can pass (init_state_kernel / bank16_kernel).  (Written-out DMA like the stages:
        // the compiler, seeing an LDS-DMA it cannot tell apart from the ring's, would drain the whole ring with vmcnt(0) in
        // front of the epilogue's reads of spar / srow.)
        if (kt_s == 0) {
            constexpr int QW = QT * 16 / 1024;           // waves that fetch the queries' parameters (1 KiB each); four more fetch the rows'
            const unsigned int l16b = (unsigned int)lane_now() * 16u;
            if (wave < QW) glds16_sbase((const char *)(qpar + (int64_t)qt_cur * QT) + wave * 1024, l16b, (char *)spar + wave * 1024);
            else if (wave < QW + 4) glds16_sbase((const char *)(rowp + (int64_t)t_cur * BT) + (wave - QW) * 1024, l16b, (char *)srow + (wave - QW) * 1024);
            // ... and the item finished one step ago (early group) / in this step's first phase (late group) is flushed
            if (STAGED && s_now > 0) flush_item(qt_done, t_done);
        } else if (kt_s == 1) {
            if (STAGED && tid == 0) *stg_n = 0;         // a barrier after the flush's reads, two before the next item's first append
        }
        kt_s = kt_s + 1 < KT ? kt_s + 1 : 0;
    };
    STAMP(-1);
#ifdef PF_CLOCK
    // -DPF_CLOCK (a diagnostic build of its own, no per-segment stamps): the clock the chip holds inside THIS kernel's loop =
    // delta s_memtime / delta s_memrealtime x 100 MHz, stamped once around the whole pass (MI355X_MICROARCH.md, DVFS item 6)
    const unsigned long long pfc_t0 = __builtin_amdgcn_s_memtime(), pfc_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    if (!late) {
        for (int s = 0; s < steps; ++s) {
            const bool issue = s + AHEAD < steps;
            const int ibuf = (s + AHEAD) & (NSTAGE - 1);
            landed(s);
            STAMP(0);
            __builtin_amdgcn_s_barrier();                                  // P0(s)
            STAMP(1);
            read_frags(s & (NSTAGE - 1));
            if (issue) {
                issue_half(ibuf, 0);
                issue_half(ibuf, 1);
            }
            STAMP(2);
            __builtin_amdgcn_s_barrier();                                  // P1(s)
            STAMP(3);
            item_constants(s);
            multiply_stage(false, 0, 0);
            STAMP(4);
        }
    } else {
        for (int s = 0; s < steps; ++s) {
            const bool issue = s + AHEAD < steps;
            const int ibuf = (s + AHEAD) & (NSTAGE - 1);
            landed(s);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");             // the fragment reads of stage s-1 are out of its buffer
            STAMP(0);
            __builtin_amdgcn_s_barrier();                                  // P0(s)
            STAMP(1);
            if (s > 0) multiply_stage(false, 0, 0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");             // staged candidates of a finished item are in LDS
            STAMP(4);
            __builtin_amdgcn_s_barrier();                                  // P1(s)
            STAMP(3);
            item_constants(s);
            read_frags(s & (NSTAGE - 1));
            if (issue) {
                issue_half(ibuf, 0);
                issue_half(ibuf, 1);
            }
            STAMP(2);
        }
        multiply_stage(false, 0, 0);                                       // the last stage (+ its item epilogue)
    }
    if (STAGED) {
        __syncthreads();                                                    // both groups are done with the last item
        flush_item(qt_done, t_done);
        if (tid == 0) wg_count[blockIdx.x] = wpos;
    }
#ifdef PF_CLOCK
    {
        const unsigned long long pfc_t1 = __builtin_amdgcn_s_memtime(), pfc_r1 = __builtin_amdgcn_s_memrealtime();
        if (tid == 0) {
            atomicAdd(dbg + 20, pfc_t1 - pfc_t0);
            atomicAdd(dbg + 21, pfc_r1 - pfc_r0);
            atomicAdd(dbg + 22, (unsigned long long)steps);
            atomicAdd(dbg + 23, 1ull);
        }
    }
#endif
#ifdef PF_STAMP
    if (lane == 0)
        for (int i = 0; i < 8; ++i) atomicAdd(dbg + (late ? 8 : 0) + i, (unsigned long long)seg[i]);
    if (lane == 0 && wave == 0) atomicAdd(dbg + 16, (unsigned long long)steps);
    if (lane == 0 && wave == 0) atomicAdd(dbg + 17, (unsigned long long)my_items);
#endif
}

// ---- the workgroups' candidate regions -> the per-query lists (slot from the query's counter; a count past `cap` marks the
// query as overflowed in select_kernel).  grid = (chunks, PF_GRID)
__global__ __launch_bounds__(256) void bucket_kernel(const uint4 *__restrict__ wg_list, const int *__restrict__ wg_count, int capw,
                                                     int cap, int *__restrict__ cnt, int *__restrict__ cand_i,
                                                     float *__restrict__ cand_d) {
    const int n = wg_count[blockIdx.y];
    const uint4 *src = wg_list + (int64_t)blockIdx.y * capw;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < n; e += gridDim.x * 256) {
        const uint4 c = src[e];
        const int q = (int)c.x;
        const int pos = atomicAdd(cnt + q, 1);
        if (pos < cap) {
            cand_i[(int64_t)q * cap + pos] = (int)c.y;
            cand_d[(int64_t)q * cap + pos] = __uint_as_float(c.z);
        }
    }
}

// the query's row of the candidate test's A operand (see prefilter_kernel): {-a, c (raised), -b, 0}
__device__ __forceinline__ float4 test_row(float a, float b, float c) { return make_float4(-a, c * (1.0f + 0x1p-19f), -b, 0.f); }

// ---- between phases: new threshold, compaction; last phase: pick the candidates to re-score ---------------------------------
__device__ __forceinline__ unsigned int orderable(float f) {
    const unsigned int u = __float_as_uint(f);
    return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float unorderable(unsigned int o) { return __uint_as_float(o ^ ((o >> 31) ? 0x80000000u : 0xFFFFFFFFu)); }

// key of the kth largest (1-based, kth <= n) among keys[0..n) (LDS); 4 radix passes of 8 bits.  All threads return it.
__device__ unsigned int radix_kth_largest(const unsigned int *keys, int n, int kth, int *hist, int tid, int nthreads) {
    unsigned int prefix = 0, mask = 0;
    for (int shift = 24; shift >= 0; shift -= 8) {
        for (int b = tid; b < 256; b += nthreads) hist[b] = 0;
        __syncthreads();
        for (int e = tid; e < n; e += nthreads) {
            const unsigned int kx = keys[e];
            if ((kx & mask) == prefix) atomicAdd(&hist[(kx >> shift) & 255], 1);
        }
        __syncthreads();
        radix_pick_bin(hist, kth, tid);
        __syncthreads();
        const int b = hist[256];
        kth = hist[257];
        prefix |= (unsigned int)b << shift;
        mask |= 255u << shift;
        __syncthreads();
    }
    return prefix;
}

// one workgroup (256 threads) per query.  state[q] = {tau (current threshold), overflow flag as float, -, -}
// final == 0: tau <- max(tau, k-th largest L); keep the candidates with U >= tau; write the test parameters of the next phase.
// final == 1: sel_i[q][0..nsel) <- the (up to) RESCORE_MAX candidates with the largest U; bound[q] = largest U NOT selected
//             (-inf when every candidate is selected).
__global__ __launch_bounds__(256) void select_kernel(int Q, int k, int cap, float eps, const float4 *__restrict__ qbase,
                                                     const float4 *__restrict__ rowp, const float *__restrict__ xn, int *__restrict__ cnt,
                                                     int *__restrict__ cand_i, float *__restrict__ cand_d,
                                                     float4 *__restrict__ qpar, float *__restrict__ tau_q, int *__restrict__ overflow,
                                                     int final, int *__restrict__ sel_i, int *__restrict__ nsel,
                                                     float *__restrict__ bound) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned int *kL = (unsigned int *)smem;                  // [cap] orderable(L)
    unsigned int *kU = kL + cap;                              // [cap] orderable(U)
    int *hist = (int *)(kU + cap);                            // [258]
    int *scan = hist + 260;                                   // [8]
    const int tid = threadIdx.x, q = blockIdx.x;
    const int raw = cnt[q];
    if (raw > cap && tid == 0) overflow[q] = 1;
    const int n = raw < cap ? raw : cap;
    const float4 qb = qbase[q];                               // {qn', eps_a ||q'||, 2^-f, qn}
    int *ci = cand_i + (int64_t)q * cap;
    float *cd = cand_d + (int64_t)q * cap;
    const float inv_sf = 1.0f / qb.z;                         // 2^f (exact)
    for (int e = tid; e < n; e += 256) {
        const float4 rp = rowp[ci[e]];                        // {xn', ||x'||, 2^-e, 0}
        const float den = fmaf(qb.w, xn[ci[e]], eps);
        const float sc = inv_sf * (1.0f / rp.z);              // 2^(e+f), exact
        const float err = qb.y * rp.y;
        const float hi = (cd[e] + err) * sc, lo = (cd[e] - err) * sc;
        float U = hi / den, L = lo / den;
        U = U >= 0.f ? U * (1.0f + 0x1p-21f) : U * (1.0f - 0x1p-21f);
        L = L >= 0.f ? L * (1.0f - 0x1p-21f) : L * (1.0f + 0x1p-21f);
        if (!(U == U)) U = INFINITY;                          // never drop what cannot be bounded
        if (!(L == L)) L = -INFINITY;
        kL[e] = orderable(L);
        kU[e] = orderable(U);
    }
    __syncthreads();
    float tau = tau_q[q];
    if (!final) {
        if (n >= k) {
            const float kth = unorderable(radix_kth_largest(kL, n, k, hist, tid, 256));
            tau = fmaxf(tau, kth);
        }
        // compaction: survivors keep their relative order (deterministic lists)
        const unsigned int tk = orderable(tau);
        int kept = 0;
        for (int base = 0; base < n; base += 256) {
            const int e = base + tid;
            const bool keep = e < n && kU[e] >= tk;
            int idx_v = 0;
            float d_v = 0.f;
            if (keep) { idx_v = ci[e]; d_v = cd[e]; }
            const unsigned long long m = __ballot(keep);
            if ((tid & 63) == 0) scan[tid >> 6] = __builtin_popcountll(m);
            __syncthreads();
            int off = kept;
            for (int w = 0; w < (tid >> 6); ++w) off += scan[w];
            const int tot = scan[0] + scan[1] + scan[2] + scan[3];
            off += __builtin_popcountll(m & ((1ull << (tid & 63)) - 1));
            __syncthreads();                                   // reads of this batch precede writes (off <= e)
            if (keep) { ci[off] = idx_v; cd[off] = d_v; }
            kept += tot;
            __syncthreads();
        }
        if (tid == 0) {
            cnt[q] = kept;
            tau_q[q] = tau;
            // test parameters of the next phase (see prefilter_kernel); tau = -inf -> everything passes
            const float t = tau > -3.0e38f ? tau : -3.0e38f;
            float a = t * qb.x, b = t * eps * qb.z;
            a -= fabsf(a) * 0x1p-19f;
            b -= fabsf(b) * 0x1p-19f;
            if (!(a > -3.0e38f)) a = -3.0e38f;
            qpar[q] = test_row(a, b, qb.y);
        }
        return;
    }
    // ---- final.  Only candidates whose interval reaches the k-th largest LOWER bound L_k can be among the k best: at least k
    // candidates have an exact score >= L_k (each score is >= its own L), so the k-th best exact score is >= L_k, and whoever
    // has U < L_k is out.  When those needed fit the re-score quota (the rule on embedding-like data: k plus a few dozen), only
    // they are re-scored and the answer is certified by construction (bound = -inf) -- half the exact stage's HBM traffic.
    int *so = sel_i + (int64_t)q * RESCORE_MAX;
    if (n >= k && n > 0) {
        const unsigned int lk = radix_kth_largest(kL, n, k, hist, tid, 256);
        if (unorderable(lk) > -INFINITY) {            // (k finite lower bounds exist; U = +inf / NaN-turned-inf rows always qualify)
            if (tid == 0) hist[0] = 0;
            __syncthreads();
            int mine = 0;
            for (int e = tid; e < n; e += 256) mine += kU[e] >= lk ? 1 : 0;
            if (mine) atomicAdd(&hist[0], mine);
            __syncthreads();
            const int need = hist[0];
            __syncthreads();
            if (need <= RESCORE_MAX) {
                if (tid == 0) hist[0] = 0;
                __syncthreads();
                for (int e = tid; e < n; e += 256)
                    if (kU[e] >= lk) so[atomicAdd(&hist[0], 1)] = ci[e];
                if (tid == 0) { nsel[q] = need; bound[q] = -INFINITY; }
                return;
            }
        }
    }
    // ---- otherwise: the RESCORE_MAX largest U
    if (n <= RESCORE_MAX) {
        for (int e = tid; e < n; e += 256) so[e] = ci[e];
        if (tid == 0) { nsel[q] = n; bound[q] = -INFINITY; }
        return;
    }
    const unsigned int uk = radix_kth_largest(kU, n, RESCORE_MAX, hist, tid, 256);   // RESCORE_MAX-th largest U
    // strictly larger first, then ties of uk in list order until the quota is full; bound = uk if anything with U <= uk is left
    if (tid == 0) { hist[0] = 0; hist[1] = 0; }
    __syncthreads();
    for (int e = tid; e < n; e += 256)
        if (kU[e] > uk) so[atomicAdd(&hist[0], 1)] = ci[e];
    __syncthreads();
    const int above = hist[0];
    for (int e = tid; e < n; e += 256)
        if (kU[e] == uk) {
            const int p = atomicAdd(&hist[1], 1);
            if (above + p < RESCORE_MAX) so[above + p] = ci[e];
        }
    __syncthreads();
    if (tid == 0) {
        nsel[q] = RESCORE_MAX;
        bound[q] = unorderable(uk);                            // every candidate left out has U <= uk
    }
}

// ---- stage 2: exact scores of the selected candidates, final order, acceptance test ------------------------------------------
__device__ __forceinline__ float finish_score(float dot, float qn, float xn, float eps) {
    const float den = fmaf(qn, xn, eps);
    const float s = __fdiv_rn(dot, den);
    return s == s ? s : -INFINITY;
}

// one workgroup (256 threads) per query; thread c owns candidate c: the contract's fma chain over d = 0..D-1
__global__ __launch_bounds__(256) void rescore_kernel(const float *__restrict__ tw, const float *__restrict__ qn,
                                                      const float *__restrict__ bank, const float *__restrict__ xn, int Q, int D,
                                                      int k, float eps, int64_t idx_offset, const int *__restrict__ sel_i,
                                                      const int *__restrict__ nsel, const float *__restrict__ bound,
                                                      const int *__restrict__ overflow, float *__restrict__ out_s,
                                                      int64_t *__restrict__ out_i, int *__restrict__ redo) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *sq = (float *)smem;                                // [D] weighted query
    float *ss = sq + D;                                       // [RESCORE_MAX] exact scores
    int *si = (int *)(ss + RESCORE_MAX);                      // [RESCORE_MAX] rows
    const int tid = threadIdx.x, q = blockIdx.x;
    const int n = nsel[q];
    for (int d = tid; d < D; d += 256) sq[d] = tw[(int64_t)q * D + d];
    __syncthreads();
    float sc = -INFINITY;
    int row = 0x7fffffff;
    if (tid < n) {
        row = sel_i[(int64_t)q * RESCORE_MAX + tid];
        const float *x = bank + (int64_t)row * D;
        float acc = 0.f;
        for (int d = 0; d < D; d += 4) {      // explicit fmaf only: nothing here for the compiler to contract
            const float4 v = *(const float4 *)(x + d);
            acc = fmaf(sq[d], v.x, acc);
            acc = fmaf(sq[d + 1], v.y, acc);
            acc = fmaf(sq[d + 2], v.z, acc);
            acc = fmaf(sq[d + 3], v.w, acc);
        }
        sc = finish_score(acc, qn[q], xn[row], eps);
    }
    ss[tid] = sc;
    si[tid] = row;
    __syncthreads();
    // rank by counting: (score desc, row asc)
    int rank = 0;
    if (tid < n) {
        for (int e = 0; e < n; ++e) {
            const float s2 = ss[e];
            const int r2 = si[e];
            rank += (s2 > sc || (s2 == sc && r2 < row)) ? 1 : 0;
        }
        if (rank < k) {
            out_s[(int64_t)q * k + rank] = sc;
            out_i[(int64_t)q * k + rank] = idx_offset + row;
        }
        if (rank == k - 1) {
            // accepted iff the k-th best exact score beats (strictly) the upper bound of everything that was not re-scored
            redo[q] = (overflow[q] || !(sc > bound[q])) ? 1 : 0;
        }
    }
    if (n < k) {                                              // fewer candidates than k (tiny shard, NaN rows): the exact kernel decides
        if (tid == 0) redo[q] = 1;
        for (int e = n + tid; e < k; e += 256) {
            out_s[(int64_t)q * k + e] = -INFINITY;
            out_i[(int64_t)q * k + e] = -1;
        }
    }
}

__global__ void init_state_kernel(int Q, int Q_padded, const float *__restrict__ thr0, const float4 *__restrict__ qbase, float eps,
                                  float4 *__restrict__ qpar, float *__restrict__ tau_q, int *__restrict__ cnt,
                                  int *__restrict__ overflow, int first_rows) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= Q) {
        if (q < Q_padded) qpar[q] = make_float4(NAN, NAN, NAN, NAN);   // padding of the last query tile: t = NaN, no pair passes
        return;
    }
    const float tau = thr0 ? thr0[q] : -INFINITY;
    tau_q[q] = tau;
    cnt[q] = thr0 ? 0 : first_rows;          // without a floor the first slice is taken whole: slot = row (prefilter_kernel<true>)
    const float4 qb = qbase[q];
    // a query whose scale hit the clamp (largest |t w| below ~2^-112) has an fp16 image outside the error bound's
    // assumptions: answered by the exact kernel (redo), never silently by this path
    overflow[q] = qb.z >= 0x1p126f ? 1 : 0;
    const float t = tau > -3.0e38f ? tau : -3.0e38f;
    float a = t * qb.x, b = t * eps * qb.z;
    a -= fabsf(a) * 0x1p-19f;
    b -= fabsf(b) * 0x1p-19f;
    if (!(a > -3.0e38f)) a = -3.0e38f;
    qpar[q] = test_row(a, b, qb.y);
}

struct Workspace {
    half_t *qh, *ql;
    float4 *qbase, *qpar;
    float *tau, *bound, *cand_d;
    int *cnt, *overflow, *cand_i, *sel_i, *nsel, *wg_count;
    uint4 *wg_list;
    int capw;
};
inline int64_t align256(int64_t v) { return (v + 255) / 256 * 256; }
int64_t carve(char *base, int Q, int D, int cap, Workspace *w) {
    int64_t off = 0;
    auto take = [&](int64_t bytes) {
        char *p = base ? base + off : nullptr;
        off += align256(bytes);
        return p;
    };
    char *p;
    const int64_t Qp = (Q + QPAD - 1) / QPAD * QPAD;          // the fp16 query images are padded to whole tiles
    p = take(Qp * D * 2); if (w) w->qh = (half_t *)p;
    p = take(Qp * D * 2); if (w) w->ql = (half_t *)p;
    p = take((int64_t)Q * 16); if (w) w->qbase = (float4 *)p;
    p = take(Qp * 16); if (w) w->qpar = (float4 *)p;
    p = take((int64_t)Q * 4); if (w) w->tau = (float *)p;
    p = take((int64_t)Q * 4); if (w) w->bound = (float *)p;
    p = take((int64_t)Q * 4); if (w) w->cnt = (int *)p;
    p = take((int64_t)Q * 4); if (w) w->overflow = (int *)p;
    p = take((int64_t)Q * 4); if (w) w->nsel = (int *)p;
    p = take((int64_t)Q * RESCORE_MAX * 4); if (w) w->sel_i = (int *)p;
    p = take((int64_t)Q * cap * 4); if (w) w->cand_i = (int *)p;
    p = take((int64_t)Q * cap * 4); if (w) w->cand_d = (float *)p;
    // stage 1 writes each workgroup's candidates to a region of its own: a few hundred candidates per query and phase in all,
    // dealt evenly over the workgroups; a full region flags the queries it had to drop
    const int capw = Q * 4 > 8192 ? Q * 4 : 8192;
    p = take((int64_t)PF_GRID * capw * 16); if (w) { w->wg_list = (uint4 *)p; w->capw = capw; }
    p = take((int64_t)PF_GRID * 4); if (w) w->wg_count = (int *)p;
    return off;
}

}  // namespace

extern "C" int skyemb_topk_prefilter_applicable(int Q, int64_t N, int D, int k) {
    // the fp16 pass pays from a few query tiles on; the re-score list must leave room above k
    return Q >= 17 && D % BK == 0 && D >= 4 * BK && D <= 4096 && k >= 1 && k <= RESCORE_MAX - 64 && N >= 8 * BT && N < (1ll << 31);
}
static int skyemb_topk_prefilter_cap(int k) { return k <= 128 ? 4096 : 8192; }

extern "C" int64_t skyemb_bank16_bytes(int64_t N, int D) { return ceil_div64(N, BT) * BT * D * 2; }   // whole tiles of BT rows

extern "C" int64_t skyemb_bank16_rowp_rows(int64_t N) { return ceil_div64(N, BT) * BT; }

extern "C" int skyemb_bank16_prepare(const float *bank, const float *xn, int64_t N, int D, void *bank16, float *rowp, void *stream) {
    SKY_CHECK_ARG(bank && xn && bank16 && rowp && N > 0 && D > 0 && D % 4 == 0, "skyemb_bank16_prepare: bad arguments");
    const int64_t padded = skyemb_bank16_rowp_rows(N);
    hipLaunchKernelGGL(bank16_kernel, dim3((unsigned)ceil_div64(padded, 4)), dim3(256), 0, (hipStream_t)stream, bank, xn, N, D,
                       (half_t *)bank16, (float4 *)rowp, padded);
    SKY_LAUNCH_CHECK("skyemb_bank16_prepare");
    return 0;
}

extern "C" int64_t skyemb_topk_prefilter_ws_bytes(int Q, int D, int k) { return carve(nullptr, Q, D, skyemb_topk_prefilter_cap(k), nullptr); }

extern "C" int skyemb_cosine_topk_prefiltered(const float *tw, const float *qn, int Q, const float *bank, const float *xn,
                                              const void *bank16, const float *rowp, int64_t N, int D, int k, float eps,
                                              int64_t idx_offset, const float *thr0, void *ws, int64_t ws_bytes, float *out_s,
                                              int64_t *out_i, int *redo, void *stream) {
    SKY_CHECK_ARG(tw && qn && bank && xn && bank16 && rowp && ws && out_s && out_i && redo, "skyemb_cosine_topk_prefiltered: null argument");
    SKY_CHECK_ARG(skyemb_topk_prefilter_applicable(Q, N, D, k), "skyemb_cosine_topk_prefiltered: outside the prefiltered subset "
                  "(Q >= 17, D %% 32 == 0, k <= %d, N >= %d)", RESCORE_MAX - 64, 8 * BT);
    const int cap = skyemb_topk_prefilter_cap(k);
    SKY_CHECK_ARG(ws_bytes >= skyemb_topk_prefilter_ws_bytes(Q, D, k), "skyemb_cosine_topk_prefiltered: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    Workspace w;
    carve((char *)ws, Q, D, cap, &w);
    // one fp16 pass (hi only) by default: half the matrix work, 3/4 of the LDS-DMA bytes, intervals ~1.6x wider -- on
    // embedding-like data a few dozen more candidates per query.  SKYEMB_PREFILTER_LO=1 keeps the hi + lo passes.
    // (read per call on purpose: a getenv is a sub-microsecond scan next to a multi-millisecond search, and the tests switch it)
    const char *lo_env = getenv("SKYEMB_PREFILTER_LO");
    const bool use_lo = lo_env && lo_env[0] == '1';
    const float eps_a = (float)(eps_a_of(D, use_lo) * (1.0 + 1e-6));
    hipLaunchKernelGGL(query16_kernel, dim3((unsigned)(((Q + QPAD - 1) / QPAD * QPAD + 3) / 4)), dim3(256), 0, st, tw, qn, Q, D, w.qh, w.ql,
                       w.qbase, eps_a);
    const int T = (int)ceil_div64(N, BT);
    // phases: [0, first) is taken whole unless a floor came in; then slices ending at 1/128, 1/32, 1/8, 1/2 and all of the tiles
    int first = (int)(cap / 2 / BT);                           // rows of the take-everything slice <= cap / 2
    if (first < 1) first = 1;
    if (first > T) first = T;
    const int64_t first_rows = (int64_t)first * BT < N ? (int64_t)first * BT : N;
    const int Q_padded = (Q + QPAD - 1) / QPAD * QPAD;
    hipLaunchKernelGGL(init_state_kernel, dim3((unsigned)((Q_padded + 255) / 256)), dim3(256), 0, st, Q, Q_padded, thr0, w.qbase, eps,
                       w.qpar, w.tau, w.cnt, w.overflow, (int)first_rows);
    // the dynamic-LDS limit is a per-DEVICE attribute of the function: one flag per device, set under a mutex
    static std::mutex attr_mutex;
    static bool attr_done[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    const int smem1 = NSTAGE * stage_bytes(use_lo) + (qtile(use_lo) + BT) * 16 + SCAP * 8 + 16;
    constexpr int smem_max = NSTAGE * stage_bytes(true) + (256 + BT) * 16 + SCAP * 8 + 16;
    std::lock_guard<std::mutex> attr_lock(attr_mutex);
    bool &attr_set = attr_done[dev & 63];
    if (!attr_set) {
        hipError_t e = hipSuccess;
#define PF_ATTR(M, L) if (e == hipSuccess) e = hipFuncSetAttribute((const void *)prefilter_kernel<M, L>, hipFuncAttributeMaxDynamicSharedMemorySize, smem_max)
        PF_ATTR(0, false); PF_ATTR(1, false); PF_ATTR(2, false); PF_ATTR(0, true); PF_ATTR(1, true); PF_ATTR(2, true);
#undef PF_ATTR
        if (e == hipSuccess)
            e = hipFuncSetAttribute((const void *)select_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 8192 * 4 + 272 * 4);
        if (e != hipSuccess) {
            skyemb_set_error("skyemb_cosine_topk_prefiltered: hipFuncSetAttribute: %s", hipGetErrorString(e));
            return 2;
        }
        attr_set = true;
    }
    // (the slice right after the first one meets a threshold that still passes ~5 % of the pairs -- k of the ~2000 rows seen --
    // so it is kept short, to ~96 k rows or 1/128 of the tiles, and appends directly; from then on the threshold passes ~1 %
    // or less and a workgroup's candidates fit its staging list and region)
    int direct_end = (int)ceil_div64((int64_t)96 * k, BT);
    if (direct_end < T / 128) direct_end = T / 128;
    if (direct_end < first + 1) direct_end = first + 1;
    int ends[6] = {first, direct_end, T / 32, T / 8, T / 2, T};
    int t0 = 0;
    bool thresholded_once = thr0 != nullptr;                   // a caller's floor comes from 256 k sampled rows: tight enough
    const size_t smem_sel = (size_t)2 * cap * 4 + 272 * 4;
    for (int p = 0; p < 6; ++p) {
        int t1 = ends[p];
        if (p == 0 && thr0) continue;                          // a valid floor is as good as the first slice
        if (t1 <= t0) continue;
        if (t1 > T) t1 = T;
#if defined(PF_STAMP) || defined(PF_CLOCK)
        static unsigned long long *dbg = nullptr;
        if (!dbg) { hipMalloc(&dbg, 32 * 8); }
        hipMemsetAsync(dbg, 0, 32 * 8, st);
#define PF_DBG , dbg
#else
#define PF_DBG
#endif
#define PF_ARGS w.qh, w.ql, (const half_t *)bank16, (const float4 *)rowp, w.qpar, Q, N, D, t0, t1, cap, w.cnt, w.cand_i, w.cand_d, \
                w.wg_list, w.wg_count, w.capw, w.overflow PF_DBG
#define PF_LAUNCH(M)                                                                                          \
    do {                                                                                                      \
        if (use_lo) hipLaunchKernelGGL((prefilter_kernel<M, true>), dim3(PF_GRID), dim3(NT), smem1, st, PF_ARGS); \
        else hipLaunchKernelGGL((prefilter_kernel<M, false>), dim3(PF_GRID), dim3(NT), smem1, st, PF_ARGS);     \
    } while (0)
        if (p == 0) {
            PF_LAUNCH(0);
        } else if (!thresholded_once) {
            PF_LAUNCH(1);
            thresholded_once = true;
        } else {
            PF_LAUNCH(2);
            hipLaunchKernelGGL(bucket_kernel, dim3(8, PF_GRID), dim3(256), 0, st, (const uint4 *)w.wg_list, (const int *)w.wg_count,
                               w.capw, cap, w.cnt, w.cand_i, w.cand_d);
        }
#undef PF_LAUNCH
#undef PF_ARGS
#ifdef PF_CLOCK
        {
            unsigned long long h[32];
            hipStreamSynchronize(st);
            hipMemcpy(h, dbg, sizeof(h), hipMemcpyDeviceToHost);
            if (h[23])
                fprintf(stderr, "[clock] phase %d (%s): in-kernel clock %.3f GHz over %.2f ms, %.0f shader cycles per k-step (256 MFMAs = 64 per SIMD x 16 cycles = 1024 at the "
                                "matrix pipe's rate), steps/wg %.0f\n",
                        p, p == 0 ? "take-all" : p == 1 ? "direct append" : "staged", (double)h[20] / (double)h[21] * 0.1, (double)h[21] / h[23] * 1e-5,
                        (double)h[20] / (double)h[22], (double)h[22] / h[23]);
        }
#endif
#ifdef PF_STAMP
        {
            unsigned long long h[32];
            hipStreamSynchronize(st);
            hipMemcpy(h, dbg, sizeof(h), hipMemcpyDeviceToHost);
            const double n = (double)h[16] * 4.0;     // steps summed over workgroups x 4 waves per group
            if (p >= 2) {
                int hc[PF_GRID];
                hipMemcpy(hc, w.wg_count, sizeof(hc), hipMemcpyDeviceToHost);
                long tot = 0; int mx = 0;
                for (int i = 0; i < PF_GRID; ++i) { tot += hc[i]; if (hc[i] > mx) mx = hc[i]; }
                fprintf(stderr, "[stamp] phase %d staged candidates: total %ld (%.1f per query), max per workgroup %d of %d\n", p, tot, (double)tot / Q, mx, w.capw);
            }
            fprintf(stderr, "[stamp] phase %d steps/wg %.0f | early: wait %.0f P0 %.0f read+issue %.0f P1 %.0f mult %.0f | late: wait %.0f P0 %.0f read+issue %.0f P1 %.0f mult %.0f (cycles of s_memtime per k-step)\n",
                    p, (double)h[16] / 256.0, h[0] / n, h[1] / n, h[2] / n, h[3] / n, h[4] / n, h[8] / n, h[9] / n, h[10] / n, h[11] / n, h[12] / n);
            const double ni = (double)h[17] * 4.0;    // items summed over workgroups x 4 waves per group
            fprintf(stderr, "[stamp] phase %d items/wg %.0f | item epilogue, cycles per item: early tests %.0f appends %.0f zero %.0f | late tests %.0f appends %.0f zero %.0f\n",
                    p, (double)h[17] / 256.0, h[5] / ni, h[6] / ni, h[7] / ni, h[13] / ni, h[14] / ni, h[15] / ni);
        }
#endif
        const int final = t1 == T;
        hipLaunchKernelGGL(select_kernel, dim3((unsigned)Q), dim3(256), smem_sel, st, Q, k, cap, eps, w.qbase, (const float4 *)rowp, xn,
                           w.cnt, w.cand_i, w.cand_d, w.qpar, w.tau, w.overflow, final, w.sel_i, w.nsel, w.bound);
        t0 = t1;
    }
    hipLaunchKernelGGL(rescore_kernel, dim3((unsigned)Q), dim3(256), (size_t)D * 4 + RESCORE_MAX * 8, st, tw, qn, bank, xn, Q, D, k, eps,
                       idx_offset, w.sel_i, w.nsel, w.bound, w.overflow, out_s, out_i, redo);
    SKY_LAUNCH_CHECK("skyemb_cosine_topk_prefiltered");
    return 0;
}
