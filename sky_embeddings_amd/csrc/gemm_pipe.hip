// Pipelined bf16 MFMA GEMM: LDS-DMA (global_load_lds_dwordx4) staging, 3-stage ring, counted
// vmcnt + raw s_barrier (one barrier per k-tile), XOR-swizzled LDS images, vectorised epilogue.
//
// Same contract as gemm.hip (skyemb_gemm_args) for the fast-path subset:
//   bf16, K % 64 == 0, N % 4 == 0, M >= 8 (k-contiguous rows) -- everything the ViT layers need.
// Operand tiles (BK = 64):
//   KC operand  [rows][64 k]  : 128-byte rows; 16-byte chunk c of row r is stored at chunk c ^ (r & 7)
//                               (the swizzle is applied to the per-lane SOURCE address: the LDS-DMA
//                               destination is always wave-base + lane*16); fragments by ds_read_b128.
//   RC operand  [64 k][rows]  : rows*2-byte k-rows; chunk ch of k-row k is stored at chunk
//                               ch ^ rc_swz(k); fragments by ds_read_b64_tr_b16 (transposed read).
// The MFMA is issued with operands swapped (D = B_frag x A_frag) so that every lane owns 4
// consecutive output COLUMNS of one row: 16-byte fp32 / 8-byte bf16 stores, float4 bias loads.
#include "lp_twin.h"
#include "adamw_math.h"
#include "ln_bwd_body.h"
#include <stdlib.h>
#include <string.h>
#include <mutex>
#include <type_traits>

namespace {

typedef __attribute__((address_space(1))) const void gvoid_t;
typedef __attribute__((address_space(3))) void lvoid_t;
typedef __attribute__((address_space(3))) lp4 lds_lp4_t;

constexpr int BK = 64;

// -DGEMM_STAMP (tools/ubench/gemm_lab.hip, LAB_STAMP): per-workgroup timeline of a launch.  Wave 0 keeps s_memtime stamps
// in scalar registers and writes them out after its last store has been acknowledged: nothing is added to the vector-memory
// queue the k-loop's counted waits rely on.  Slot layout per workgroup (16 x u64): 0 entry, 1 first LDS-DMA issued,
// 2 first stage landed (after the barrier), 3 k-loop done, 4 tile staged in LDS, 5 last store issued, 6 stores acknowledged,
// 8 / 9 s_memrealtime (100 MHz, chip-wide) at entry / exit, 10 XCC id.  The product build compiles none of it.
#ifdef GEMM_STAMP
__device__ unsigned long long *g_gemm_stamp = nullptr;
#define GSTAMP(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#else
#define GSTAMP(var)
#endif

__device__ __forceinline__ void glds16(const void *src, char *lds_wave_base) {
    __builtin_amdgcn_global_load_lds((gvoid_t *)src, (lvoid_t *)lds_wave_base, 16, 0, 0);
}

// loads through an explicitly GLOBAL pointer: a struct member like `const float *bias` is a generic pointer to the compiler, which
// then emits flat_load -- out of order with respect to global loads, so every wait behind one is a full `vmcnt(0) lgkmcnt(0)`
typedef __attribute__((address_space(1))) const f32x4 gf32x4_t;
typedef __attribute__((address_space(1))) const lp8 gbf16x8_t;
typedef __attribute__((address_space(1))) const int gint_t;
__device__ __forceinline__ float4 gload4(const float *p) {
    const f32x4 v = *(gf32x4_t *)p;
    return make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ lp8 gload8h(const lp_t *p) { return *(gbf16x8_t *)p; }
__device__ __forceinline__ int gloadi(const int *p) { return *(gint_t *)p; }

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// exact-erf GELU for bf16 outputs.  erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, three orders below a bf16
// ulp; the fp32 parity path in gemm.hip keeps erff): one v_exp + one v_rcp instead of erff's ~40 instructions, and
// exp(-x^2/2) serves both the erf tail and the Gaussian density of dGELU.  The epilogue of the [tokens, 4*dim] launches
// was ALU-bound on erff/expf before (dec.fc2 dgrad 36 -> 2x the plain GEMM).
__device__ __forceinline__ void gelu_parts(float x, float &cdf, float &gauss) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    gauss = __expf(-0.5f * x * x);                       // = exp(-z^2)
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float tail = 0.5f * p * t * gauss;             // 0.5 * erfc(z)
    cdf = x >= 0.f ? 1.0f - tail : tail;
}
__device__ __forceinline__ float gelu_f(float x) {
    float cdf, gauss;
    gelu_parts(x, cdf, gauss);
    return x * cdf;
}
__device__ __forceinline__ float dgelu_f(float x) {
    float cdf, gauss;
    gelu_parts(x, cdf, gauss);
    return fmaf(x * 0.39894228040143267794f, gauss, cdf);
}

__device__ __forceinline__ void store8(lp_t *p, const float (&v)[8]) {
    lp8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (lp_t)v[e];
    *(lp8 *)p = o;
}

// One 4-byte LDS-DMA read per lane from a workgroup-uniform base + a 32-bit lane offset into the 256 bytes at `lds_wave_base`: the
// prefetch hint of skyemb_gemm_args (a wave touches 64 lines of 128 bytes = 8 KiB per instruction; the data is never used -- it
// lands where the wave's own first operand piece of stage 0 lands AFTER it, the wave's requests completing in order).
__device__ __forceinline__ void glds4_sbase(const void *base_uniform, unsigned int lane_off, char *lds_wave_base) {
    const unsigned int dst = (unsigned int)(uintptr_t)(__attribute__((address_space(3))) void *)lds_wave_base;
    asm volatile("s_nop 4\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1"
                 :
                 : "v"(lane_off), "s"(base_uniform), "s"(dst)
                 : "memory", "m0");
}

// ---- stage issue -----------------------------------------------------------------------------
// KC operand: R rows x 64 k.  One wave-instruction = 8 rows x 128 B.
// Instructions of one operand tile per wave: R / 8 of them over NW waves, rounded up.  When they do not divide (144-row tiles,
// six-wave workgroups) the waves past the end repeat the LAST instruction -- the same bytes to the same LDS address -- so that
// every wave issues the same number and the counted vmcnt waits stay uniform.
template <int R, int NW>
constexpr int issue_per_wave() { return (R / 8 + NW - 1) / NW; }
template <int R, int NW>
__device__ __forceinline__ void issue_kc(const lp_t *__restrict__ X, int64_t ld, int r0, int rows, int k0, char *sbase,
                                         int wave, int lane) {
    constexpr int PER_WAVE = issue_per_wave<R, NW>();
    constexpr bool EXACT = (R / 8) % NW == 0;
    const int rr = lane >> 3, cs = (lane & 7) ^ rr;   // source chunk for LDS chunk lane&7 of row rr (rows8 % 8 == 0)
#pragma unroll
    for (int j = 0; j < PER_WAVE; ++j) {
        int idx = wave * PER_WAVE + j;
        if (!EXACT) idx = idx < R / 8 ? idx : R / 8 - 1;
        const int rows8 = idx * 8;
        int gr = r0 + rows8 + rr;
        gr = gr < rows ? gr : rows - 1;            // rows past the edge feed outputs that are never stored
        glds16(X + (int64_t)gr * ld + k0 + cs * 8, sbase + rows8 * 128);
    }
}
// XOR applied to the 16-byte chunk index of k-row k in an RC image [64 k][R rows]: makes the 8 k-rows
// that one half-wave of a ds_read_b64_tr_b16 touches land on disjoint bank ranges (checked
// exhaustively with the bank rules of MI355X_MICROARCH.md: conflict-free for R = 64, 128 and 256 -- a 512-byte k-row
// is two 256-byte bank rows, the XOR stays inside each).
template <int R>
__device__ __forceinline__ int rc_swz(int k) {
    return R >= 128 ? (((k & 3) << 1) | (((k >> 3) & 1) << 3)) : ((((k >> 1) & 1) << 1) | (((k >> 3) & 1) << 2));
}
// RC operand: 64 k-rows x R rows (R*2 bytes per k-row).  One wave-instruction = 1 KiB = 512/R k-rows.
template <int R, int NW>
__device__ __forceinline__ void issue_rc(const lp_t *__restrict__ X, int64_t ld, int r0, int rows, int k0, char *sbase,
                                         int wave, int lane) {
    constexpr int CH = R / 8;                 // 16-byte chunks per k-row
    constexpr int KROWS = 64 / CH;            // k-rows per wave-instruction (4 for R=128, 8 for R=64)
    constexpr int PER_WAVE = issue_per_wave<R, NW>(); // instructions per wave (64 / KROWS = R / 8 in all)
    constexpr bool EXACT = (R / 8) % NW == 0;
    static_assert(R == 64 || R == 128 || R == 256, "row-contiguous operand tiles: 64, 128 or 256 rows");
    const int kr = lane / CH, ch = lane % CH;
#pragma unroll
    for (int j = 0; j < PER_WAVE; ++j) {
        int idx = wave * PER_WAVE + j;
        if (!EXACT) idx = idx < R / 8 ? idx : R / 8 - 1;
        const int kbase = idx * KROWS;
        const int k = kbase + kr;
        const int sch = ch ^ rc_swz<R>(k);                          // source chunk for LDS chunk `ch`
        int gr = r0 + sch * 8;
        gr = gr + 8 <= rows ? gr : rows - 8;   // clamp whole chunk (rows % 8 == 0)
        glds16(X + (int64_t)(k0 + k) * ld + gr, sbase + kbase * (R * 2));
    }
}

// byte offset of 16-byte chunk `ch` of k-row `k` in an RC stage image
template <int R>
__device__ __forceinline__ int rc_off(int k, int ch) {
    return k * (R * 2) + ((ch ^ rc_swz<R>(k)) << 4);
}

// ---- fragment reads --------------------------------------------------------------------------
__device__ __forceinline__ lp8 frag_kc(const char *sbase, int rbase, int kk, int lane) {
    const int r = rbase + (lane & 15);
    const int c = (4 * kk + (lane >> 4)) ^ (lane & 7);
    return *(const lp8 *)(sbase + r * 128 + (c << 4));
}
template <int R>
__device__ __forceinline__ lp8 frag_rc(const char *sbase, int rbase, int kk, int lane) {
    const int i = lane & 15, q = i >> 2, p = i & 3;
    const int kb = kk * 32 + 8 * (lane >> 4);
    const int ch = (rbase >> 3) + (p >> 1);
    const char *a0 = sbase + rc_off<R>(kb + q, ch) + 8 * (p & 1);
    const char *a1 = sbase + rc_off<R>(kb + 4 + q, ch) + 8 * (p & 1);
    lp4 lo = sky_ds_read_tr16_b64((lds_lp4_t *)a0);
    lp4 hi = sky_ds_read_tr16_b64((lds_lp4_t *)a1);
    lp8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
}

// The same fragment reads as inline asm, for every kernel with a row-contiguous operand.  The compiler cannot tell that a
// ds_read_b64_tr_b16 (an intrinsic without alias information) does not touch the stages the LDS-DMA is still filling, and
// puts an `s_waitcnt vmcnt(0)` in front of it: every k-step then waits for ALL loads in flight, the ring degenerates to
// load -> wait -> compute (measured: 0.61 instead of 0.29 us per k-step on [1280 x 768] launches, +2-3 us on every data- and
// weight-gradient launch).  As asm the reads are invisible to that pass; the LDS counter is then waited on by hand
// (lds_wait), with every read of a k-step in asm so that the count is exact.
__device__ __forceinline__ uint32_t lds_addr(const char *p) { return (uint32_t)(uintptr_t)(lvoid_t *)p; }
__device__ __forceinline__ lp8 frag_kc_asm(const char *sbase, int rbase, int kk, int lane) {
    const int r = rbase + (lane & 15);
    const int c = (4 * kk + (lane >> 4)) ^ (lane & 7);
    lp8 v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(lds_addr(sbase + r * 128 + (c << 4))) : "memory");
    return v;
}
template <int R>
__device__ __forceinline__ lp8 frag_rc_asm(const char *sbase, int rbase, int kk, int lane) {
    const int i = lane & 15, q = i >> 2, p = i & 3;
    const int kb = kk * 32 + 8 * (lane >> 4);
    const int ch = (rbase >> 3) + (p >> 1);
    lp4 lo, hi;
#ifdef SKY_RC_AS_B64      // experiment build (wrong products): plain 8-byte reads at the same addresses -- same k-step time
    asm volatile("ds_read_b64 %0, %1" : "=v"(lo) : "v"(lds_addr(sbase + rc_off<R>(kb + q, ch) + 8 * (p & 1))) : "memory");
    asm volatile("ds_read_b64 %0, %1" : "=v"(hi) : "v"(lds_addr(sbase + rc_off<R>(kb + 4 + q, ch) + 8 * (p & 1))) : "memory");
#else
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(lds_addr(sbase + rc_off<R>(kb + q, ch) + 8 * (p & 1))) : "memory");
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(hi) : "v"(lds_addr(sbase + rc_off<R>(kb + 4 + q, ch) + 8 * (p & 1))) : "memory");
#endif
    lp8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
}
// at most N asm LDS reads still outstanding (they return in order); the fragments are then passed through lds_use so that
// nothing consuming them can be scheduled above the wait
template <int N>
__device__ __forceinline__ void lds_wait() {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N < 15 ? N : 15) : "memory");
}
__device__ __forceinline__ void lds_use(lp8 &f) { asm volatile("" : "+v"(f)); }

// wait until at most `rem` younger stages (NI LDS-DMA instructions each) are still in flight
template <int NI, int MAXREM>
__device__ __forceinline__ void wait_stages(int rem) {
    if constexpr (MAXREM == 0) {
        wait_vmcnt<0>();
    } else {
        if (rem >= MAXREM) wait_vmcnt<MAXREM * NI>();
        else wait_stages<NI, MAXREM - 1>(rem);
    }
}

// NSTAGE-deep LDS ring: NSTAGE-1 k-tiles are in flight ahead of the one being consumed.
// WM x WN waves per workgroup, each owning a (BM/WM) x (BN/WN) block of the tile: 2x2 on 64x64 (32x32 per wave),
// 4x2 on 128x64, 2x2 / 4x2 on 128x128 (64x64 / 32x64 per wave), 4x2 on 256x128 (64x64 per wave).  A wave tile of
// 64x64 reads 8 fragments for 16 MFMAs (32x32: 4 for 4), i.e. half the LDS read traffic per FLOP.
// One workgroup's work: tile `tb` of `ntiles` (XCD-aware order), k-range `split` of `S`.
// WK = 2 ("two wave groups over k"): a ring stage holds TWO consecutive k-tiles and the workgroup has 2 x WM x WN waves; group wk
// multiplies k-tile wk of every stage into its own accumulators, the two partial tiles are added in LDS on the way to the
// epilogue (fixed order: even k-tiles + odd k-tiles).  Why: with ONE workgroup per CU (launches of <= 256 tiles: [1280 x 768]
// outputs) a k-step is one wave's serial chain -- barrier, LDS-DMA issue, fragment reads, 8 MFMAs: 0.29 us whatever the ring
// depth (K-sweeps with 3 / 4 / 6 / 8 stages all gave 0.29; three co-resident workgroups reach 0.14 per k-step and tile) --
// and a second wave per SIMD running the same chain on the other half of the stage overlaps it.
// ADAM (grouped weight-gradient launches planned with skyemb_gemm_group_plan_adamw): the output tile is not stored as a gradient;
// the epilogue applies the AdamW step to the parameters the tile belongs to (`ad`: the flat buffers; see include/skyemb.h).
// BMS = row STRIDE of the tiles (default BM).  BMS < BM: consecutive tiles overlap by BM - BMS rows, which the later tile computes
// and the earlier one does not store -- a 144-row image (nine 16-row MFMA fragments) stepping by 136 rows cuts the decoder's
// 4352 = 32 x 136 token rows into exactly 32 row blocks: [4352 x 512] outputs are then 256 tiles of 136 x 64, one per CU, where
// 64-row tiles gave 544 (three on 32 CUs, two on the rest: the launch ran at the pace of the CUs with three).
template <int BM, int BN, bool A_KC, bool B_KC, int NSTAGE, int WM, int WN, int WK = 1, bool ADAM = false, int BMS = BM>
__device__ __forceinline__ void gemm_pipe_body(const skyemb_gemm_args &g, const int tb, const int ntiles, const int split,
                                               const int S, char *smem, const skyemb_adamw_desc *ad = nullptr, const bool linear = false) {
    constexpr int NWG = WM * WN, NW = NWG * WK;         // waves per k-group / per workgroup
    constexpr int SM = BM / WM, SN = BN / WN;           // rows / columns of the tile owned by one wave
    constexpr int TM = SM / 16, TN = SN / 16;
    constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, SUB = A_BYTES + B_BYTES, STAGE = SUB * WK;
    constexpr int NI = (issue_per_wave<BM, NW>() + issue_per_wave<BN, NW>()) * WK;   // LDS-DMA instructions per wave per stage
    static_assert(BM % 8 == 0 && BN % 8 == 0 && SM % 16 == 0 && SN % 16 == 0 && SM * WM == BM && SN * WN == BN, "tile / wave grid mismatch");
    static_assert(WK == 1 || WK == 2, "one or two k-groups");
    static_assert(BMS <= BM && BMS % 2 == 0 && (BMS == BM || A_KC), "overlapping row blocks: k-contiguous A only (any even row stride: rows are addressed one by one)");

    GSTAMP(st_entry);
#ifdef GEMM_STAMP
    const unsigned long long st_real0 = __builtin_amdgcn_s_memrealtime();
#endif
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wk = WK == 1 ? 0 : wave / NWG, wave_g = WK == 1 ? wave : wave % NWG;
    const int wm = wave_g / WN, wn = wave_g % WN;
    // (index arithmetic is unsigned 32-bit on purpose: every workgroup runs it before its first load, and the signed / 64-bit
    // divisions it used to contain were ~300 scalar instructions of software division)
    const unsigned int tiles_n = ((unsigned int)g.N + BN - 1) / BN;
    unsigned int wg;
    {   // XCD-aware tile order (see gemm.hip)
        // (re-mapping the workgroups that share a CU onto horizontally adjacent tiles was measured: no L1 reuse, no gain)
        const unsigned int nwg = (unsigned int)ntiles, xcd = (unsigned int)tb & 7u, local = (unsigned int)tb >> 3;
        const unsigned int q = nwg >> 3, r = nwg & 7u;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
        if (linear) wg = (unsigned int)tb;               // (grouped launches: the group kernel has laid the tiles out per XCD)
    }
    // An XCD's chunk of consecutive tiles walks ONE operand in full and a slice of the other (the 8 L2s are not coherent:
    // each fetches what its tiles touch).  Row-major tile order makes that full operand B, column-major A: let the SMALLER
    // operand be the one every L2 fetches whole (PMC, round 2: 12.1 GB of L2-side fetches + writes per step against 4.7 GB
    // of single-copy bytes, mostly the weight matrices fetched by all eight L2s).
#ifndef SKY_TILE_ROWMAJOR
    const unsigned int tiles_m = ((unsigned int)g.M + BMS - 1) / BMS;
    const bool colmajor = g.N > g.M;
    const unsigned int div = colmajor ? tiles_m : tiles_n, quo = wg / div, rem = wg - quo * div;     // one division
    const int tile_m = (int)(colmajor ? rem : quo), tile_n = (int)(colmajor ? quo : rem);
#else
    const unsigned int quo = wg / tiles_n;
    const int tile_m = (int)quo, tile_n = (int)(wg - quo * tiles_n);
#endif
    const int m0 = tile_m * BMS, n0 = tile_n * BN;
    const lp_t *A = (const lp_t *)g.A;
    const lp_t *B = (const lp_t *)g.B;
    const int KT_all = g.K / (BK * WK);                 // ring stages (WK k-tiles each; the host checks divisibility)
    int kt_begin = 0, KT = KT_all;
    if (S > 1) {   // K / 64 * 8 < 2^32
        kt_begin = (int)((unsigned int)KT_all * (unsigned int)split / (unsigned int)S);
        KT = (int)((unsigned int)KT_all * (unsigned int)(split + 1) / (unsigned int)S) - kt_begin;
    }

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    auto issue = [&](int kt, int buf) {
#pragma unroll
        for (int j = 0; j < WK; ++j) {                   // every wave takes its share of every k-tile of the stage
            char *sa = smem + buf * STAGE + j * SUB, *sb = sa + A_BYTES;
            const int k0 = ((kt_begin + kt) * WK + j) * BK;
            if constexpr (A_KC) issue_kc<BM, NW>(A, g.lda, m0, g.M, k0, sa, wave, lane);
            else issue_rc<BM, NW>(A, g.lda, m0, g.M, k0, sa, wave, lane);
            if constexpr (B_KC) issue_kc<BN, NW>(B, g.ldb, n0, g.N, k0, sb, wave, lane);
            else issue_rc<BN, NW>(B, g.ldb, n0, g.N, k0, sb, wave, lane);
        }
    };

    // bias gradient = column sums of the RC A tile, taken by the waves of the first tile column with one extra MFMA
    // per A fragment against a fragment of ones (exact products, fp32 accumulation in k order)
    // (The first column's tiles are the stragglers of every weight-gradient launch: 0.52 -> 0.44 us per k-step without the sums on
    // the 64x64 tile, 0.66 -> 0.60 on 128x128 -- tools/ubench/gemm_lab, LAB_NOCOLSUM.  Spreading the sums over all tile columns as
    // partial sums was built in round 4 and returned nothing in the step -- the extra reduce items cost what was gained --
    // and was removed in round 5: profiles/HISTORY.md, round 4 item 2.)
    const bool do_colsum = !A_KC && wn == 0 && g.colsum_a != nullptr && tile_n == 0;
    f32x4 cacc[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) cacc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    lp8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (lp_t)1.0f;

    constexpr int AHEAD = NSTAGE - 1;
    // The prefetch hint (skyemb.h): this workgroup's share of the lines a later launch will read, requested BEFORE its own first
    // operand loads -- the oldest requests of every wave, so the counted waits below cover them and they cost no wait of their own:
    // they return with the first stage (both are HBM / memory-side-cache misses issued at the same moment).
    // Where the 4 bytes per lane land: in the LDS piece THIS wave writes its own first operand piece of stage 0 to, right after.  A
    // wave's vector-memory loads return in issue order (that order is what the counted vmcnt waits of this loop are built on: vmcnt
    // retires loads oldest first), so the wave's operand piece overwrites its hint; where several waves share a piece (tiles whose
    // piece count does not divide over the waves: the surplus waves repeat the LAST piece), each of them writes that piece's correct
    // bytes after its own hint, and nobody reads the stage before every wave's waits and the barrier: whichever write lands last, it
    // is the operand.  (test_gemm_prefetch_hint_changes_no_result: every ring tile, both the exact and the surplus-wave cases.)
    if (g.prefetch != nullptr && g.prefetch_wgs == 0 && split == 0 && KT > 0) {
        constexpr int PA = issue_per_wave<BM, NW>();
        const int piece = wave * PA < BM / 8 ? wave * PA : BM / 8 - 1;          // the wave's first A piece of stage 0, k-group 0
        const unsigned int nchunk = (unsigned int)((g.prefetch_bytes + 8191) >> 13);
        const unsigned int slot = (unsigned int)tb * NW + (unsigned int)wave, nslot = (unsigned int)ntiles * NW;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const unsigned int c = slot + (unsigned int)j * nslot;               // wave-uniform
            if (c < nchunk) {
                const long long left = g.prefetch_bytes - ((long long)c << 13);
                const unsigned int last = (unsigned int)(left < 8192 ? left : 8192) - 4u;
                const unsigned int off = (unsigned int)lane * 128u;
                glds4_sbase((const char *)g.prefetch + ((size_t)c << 13), off < last ? off : last, smem + piece * 1024);
            }
        }
    }
#pragma unroll
    for (int p = 0; p < AHEAD; ++p)
        if (p < KT) issue(p, p);
    GSTAMP(st_issued);
#ifdef GEMM_STAMP
    unsigned long long st_landed = 0;
#endif
    int buf = 0;
    for (int kt = 0; kt < KT; ++kt) {
        wait_stages<NI, AHEAD - 1>(KT - 1 - kt);   // stage kt has landed; up to AHEAD-1 younger ones stay in flight
        __builtin_amdgcn_s_barrier();
#ifdef GEMM_STAMP
        if (kt == 0) st_landed = __builtin_amdgcn_s_memtime();
#endif
        if (kt + AHEAD < KT) issue(kt + AHEAD, buf >= 1 ? buf - 1 : NSTAGE - 1);   // (kt+AHEAD) % NSTAGE == (buf-1) mod NSTAGE
        const char *sa = smem + buf * STAGE + wk * SUB, *sb = sa + A_BYTES;
#ifdef SKY_NOMATH   // experiment build: LDS-DMA stream + barriers only
        buf = buf + 1 < NSTAGE ? buf + 1 : 0;
        continue;
#endif
        if constexpr (A_KC && B_KC) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                lp8 fa[TM], fb[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[i] = frag_kc(sa, wm * SM + i * 16, kk, lane);
#pragma unroll
                for (int j = 0; j < TN; ++j) fb[j] = frag_kc(sb, wn * SN + j * 16, kk, lane);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = sky_mfma_16x16x32(fb[j], fa[i], acc[i][j]);  // D[n][m]
            }
        } else {
            // both halves of the k-step are requested up front; the first half's MFMAs run while the second half lands
            constexpr int READS = TM * (A_KC ? 1 : 2) + TN * (B_KC ? 1 : 2);     // asm LDS reads per half
            lp8 fa[2][TM], fb[2][TN];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    fa[kk][i] = A_KC ? frag_kc_asm(sa, wm * SM + i * 16, kk, lane) : frag_rc_asm<BM>(sa, wm * SM + i * 16, kk, lane);
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    fb[kk][j] = B_KC ? frag_kc_asm(sb, wn * SN + j * 16, kk, lane) : frag_rc_asm<BN>(sb, wn * SN + j * 16, kk, lane);
            }
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                if (kk == 0) {
                    lds_wait<READS>();
                } else {
                    __builtin_amdgcn_sched_barrier(0);      // keep the first half's MFMAs above the second wait
                    lds_wait<0>();
                }
#pragma unroll
                for (int i = 0; i < TM; ++i) lds_use(fa[kk][i]);
#pragma unroll
                for (int j = 0; j < TN; ++j) lds_use(fb[kk][j]);
#ifndef SKY_NOMFMA     // experiment build: load pipeline + fragment reads, no matrix instruction
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = sky_mfma_16x16x32(fb[kk][j], fa[kk][i], acc[i][j]);  // D[n][m]
#endif
                if (!A_KC && do_colsum) {
#pragma unroll
                    for (int i = 0; i < TM; ++i)
                        cacc[i] = sky_mfma_16x16x32(ones, fa[kk][i], cacc[i]);
                }
            }
        }
        buf = buf + 1 < NSTAGE ? buf + 1 : 0;
    }
    GSTAMP(st_kloop);
    // ---- epilogue ------------------------------------------------------------------------------------------------
    // The accumulators leave the MFMA with one ROW per lane (16 rows per wave-instruction): stored as they stand, a
    // wave-instruction scatters 64 pieces of 8-16 bytes over 16 rows, and the launch ends in a store phase at ~1.3 TB/s
    // (measured: the K -> 0 intercept of the [4352, 2048] bf16 launch was 13 us for 17.8 MB).  So the tile is transposed
    // through LDS (the ring is free now) and written in pieces of 8 consecutive columns per lane, consecutive lanes on
    // consecutive pieces of a row: every wave-instruction writes whole 128-byte lines.  The fused inputs (bias, table
    // rows, residual, dGELU operand) are read in the same lane order.
    float *cs_out = S > 1 ? (float *)g.ws + (int64_t)S * g.M * g.N + (int64_t)split * g.M : g.colsum_a;
    constexpr int PITCH = BN * 4 + 16;                   // bytes per tile row in LDS (+16: conflict-free b128 writes)
    float *cs_lds = (float *)(smem + BM * PITCH);        // (two k-groups) the odd k-tiles' column sums, BM floats behind the image
    // The fused inputs of this thread's pieces (8 consecutive columns of a row each) are requested HERE, all of them, before
    // the tile goes through LDS: the row maps first (dst_row / tab_row), then bias, table rows, residual and the dGELU operand
    // at clamped coordinates, with no use in between.  Loaded where they were used -- under `if (m < M)`, one input after the
    // other -- the compiler waited for every load on the spot: two to five dependent memory round trips per piece (1.4-2.6 us
    // of a 6-10 us workgroup, profiles/r03_gemm_timeline.json).
    constexpr int PPR = BN / 8, PIECES = BM * PPR, NP = (PIECES + NW * 64 - 1) / (NW * 64);   // pieces per row / per tile / per thread
    constexpr bool PIECES_EXACT = PIECES % (NW * 64) == 0;   // (else the last round of pieces is short: clamped requests, skipped stores)
    // ... in chunks of at most two pieces per thread (the four pieces of a 128x128 tile at once cost 112 registers: past the
    // 128 that let two such workgroups share a CU -- measured: mim_19 28.7 -> 32.5 ms)
    constexpr int CH = (ADAM || NP < 2 || NP % 2) ? 1 : 2, NCH = NP / CH;   // (ADAM: p, m, v of a piece are 24 registers more)
    static_assert(NP % CH == 0, "pieces per thread must split into chunks");
    const bool fused = S == 1;                           // split-K: raw partial tiles, splitk_reduce_kernel applies the epilogue
    int orow[CH], trow[CH];
    float4 e_bias[CH][2], e_tab[CH][2], e_res[CH][2];
    lp8 e_aux[CH];
    float4 e_p[CH][2], e_m[CH][2], e_v[CH][2];           // (ADAM) the parameters and moments the piece updates
    const int64_t ad_off = ADAM ? (int64_t)(g.out_f32 - ad->g_base) : 0;   // element offset of this problem in the flat buffers
    const lp_t *aux = (const lp_t *)g.aux;
    // (two copies of the request block: without row maps -- every launch of a transformer block -- no load feeds an address,
    // so nothing is waited for before the tile is staged; with them the residual / table rows wait for the maps only)
    auto request_inputs = [&](int j0, auto with_maps) {
        constexpr bool MAPS = decltype(with_maps)::value;
#pragma unroll
        for (int jj = 0; jj < CH; ++jj) {
            int p = tid + (j0 + jj) * NW * 64;
            if (!PIECES_EXACT) p = p < PIECES ? p : PIECES - 1;
            const int m = m0 + p / PPR;
            const int mc = m < g.M ? m : g.M - 1;
            orow[jj] = (MAPS && g.dst_row) ? gloadi(g.dst_row + mc) : mc;
            trow[jj] = (MAPS && g.tab_row) ? gloadi(g.tab_row + mc) : 0;
        }
#pragma unroll
        for (int jj = 0; jj < CH; ++jj) {
            int p = tid + (j0 + jj) * NW * 64;
            if (!PIECES_EXACT) p = p < PIECES ? p : PIECES - 1;
            const int m = m0 + p / PPR, n = n0 + (p % PPR) * 8;
            const int mc = m < g.M ? m : g.M - 1, nc = n < g.N ? n : g.N - 8;
            if (g.bias) {
                e_bias[jj][0] = gload4(g.bias + nc);
                e_bias[jj][1] = gload4(g.bias + nc + 4);
            }
            if (g.act == SKYEMB_ACT_DGELU) e_aux[jj] = gload8h(aux + (int64_t)mc * g.ldaux + nc);
        }
#pragma unroll
        for (int jj = 0; jj < CH; ++jj) {
            int p = tid + (j0 + jj) * NW * 64;
            if (!PIECES_EXACT) p = p < PIECES ? p : PIECES - 1;
            const int n = n0 + (p % PPR) * 8;
            const int nc = n < g.N ? n : g.N - 8;
            const int oc = orow[jj] < 0 ? 0 : orow[jj];
            if (MAPS && g.table) {
                e_tab[jj][0] = gload4(g.table + (int64_t)trow[jj] * g.ldt + nc);
                e_tab[jj][1] = gload4(g.table + (int64_t)trow[jj] * g.ldt + nc + 4);
            }
            if (g.resid) {
                e_res[jj][0] = gload4(g.resid + (int64_t)oc * g.ldr + nc);
                e_res[jj][1] = gload4(g.resid + (int64_t)oc * g.ldr + nc + 4);
            }
            if (ADAM) {
                const int64_t o = ad_off + (int64_t)oc * g.ldo32 + nc;
                e_p[jj][0] = gload4(ad->p + o); e_p[jj][1] = gload4(ad->p + o + 4);
                e_m[jj][0] = gload4(ad->m + o); e_m[jj][1] = gload4(ad->m + o + 4);
                e_v[jj][0] = gload4(ad->v + o); e_v[jj][1] = gload4(ad->v + o + 4);
            }
        }
    };
    const bool maps = g.dst_row || g.tab_row;
    __builtin_amdgcn_s_barrier();                        // every wave is done with the fragments of the last stage
    if (WK == 2) {                                       // partial tile of the odd k-tiles first, the even ones are added to it
        if (wk == 1) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int r = wm * SM + i * 16 + (lane & 15), c = wn * SN + j * 16 + 4 * (lane >> 4);
                    *(f32x4 *)(smem + r * PITCH + c * 4) = acc[i][j];
                }
            if (!A_KC && do_colsum && lane < 16) {
#pragma unroll
                for (int i = 0; i < TM; ++i) cs_lds[wm * SM + i * 16 + lane] = cacc[i][0];
            }
        }
        __syncthreads();
    }
    if (!A_KC && do_colsum && lane < 16 && wk == 0) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = m0 + wm * SM + i * 16 + lane;
            float v = cacc[i][0];
            if (WK == 2) v += cs_lds[wm * SM + i * 16 + lane];
            if (m < g.M) cs_out[m] = v;
        }
    }
    if (wk == 0) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int r = wm * SM + i * 16 + (lane & 15), c = wn * SN + j * 16 + 4 * (lane >> 4);
                f32x4 v = acc[i][j];
                if (WK == 2) v += *(const f32x4 *)(smem + r * PITCH + c * 4);
                *(f32x4 *)(smem + r * PITCH + c * 4) = v * g.alpha;
            }
    }
    // (requested once the accumulators have left the registers: hoisted above the staging, the inputs of a 128x128 tile's four
    // pieces per thread pushed the kernel past 128 registers -- one workgroup per CU instead of two, mim_19 28.7 -> 32.5 ms)
    if (fused) {
        if (maps) request_inputs(0, std::true_type{});
        else request_inputs(0, std::false_type{});
    }
    __syncthreads();
    GSTAMP(st_staged);
    lp_t *out = (lp_t *)g.out;
    lp_t *out2 = (lp_t *)g.out2;
    float *slab = S > 1 ? (float *)g.ws + (int64_t)split * g.M * g.N : nullptr;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        if (ch > 0 && fused) {
            if (maps) request_inputs(ch * CH, std::true_type{});
            else request_inputs(ch * CH, std::false_type{});
        }
#pragma unroll
        for (int jj = 0; jj < CH; ++jj) {
            const int p = tid + (ch * CH + jj) * NW * 64;
            if (!PIECES_EXACT && p >= PIECES) continue;
            const int r = p / PPR, c = (p % PPR) * 8;
            const int m = m0 + r, n = n0 + c;
            if (m >= g.M || n >= g.N || (BMS < BM && r >= BMS)) continue;
            const float4 lo = *(const float4 *)(smem + r * PITCH + c * 4), hi = *(const float4 *)(smem + r * PITCH + c * 4 + 16);
            float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#ifdef SKY_NOSTORE   // experiment build: everything but the global traffic of the epilogue
            asm volatile("" ::"v"(v[0]), "v"(v[7]));
            continue;
#endif
            if (!fused) {   // split-K: raw partial tile to the workspace; splitk_reduce_kernel finishes (full epilogue there)
                *(float4 *)(slab + (int64_t)m * g.N + n) = lo;
                *(float4 *)(slab + (int64_t)m * g.N + n + 4) = hi;
                continue;
            }
            const int orw = orow[jj];
            if (orw < 0) continue;
            auto add8 = [&](const float4 &a, const float4 &b) {
                v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w; v[4] += b.x; v[5] += b.y; v[6] += b.z; v[7] += b.w;
            };
            if (g.bias) add8(e_bias[jj][0], e_bias[jj][1]);
            if (g.table) add8(e_tab[jj][0], e_tab[jj][1]);
            if (g.resid) add8(e_res[jj][0], e_res[jj][1]);
            if (g.act == SKYEMB_ACT_GELU) {
                if (out2) store8(out2 + (int64_t)orw * g.ldo2 + n, v);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = gelu_f(v[e]);
            } else if (g.act == SKYEMB_ACT_DGELU) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] *= dgelu_f((float)e_aux[jj][e]);
            }
            if (ADAM) {
                // v[] is the gradient of 8 consecutive parameters: the optimiser step, here (adamw_math.h)
                const float lr = ad->hyper[0], bc1 = ad->hyper[1], bc2 = ad->hyper[2];
                const SkyAdamScalars sc = sky_adam_scalars(lr, bc1, bc2, ad->beta1, ad->beta2, ad->eps, ad->weight_decay, ad->grad_scale);
                const int64_t o = ad_off + (int64_t)orw * g.ldo32 + n;
                float pp[8] = {e_p[jj][0].x, e_p[jj][0].y, e_p[jj][0].z, e_p[jj][0].w, e_p[jj][1].x, e_p[jj][1].y, e_p[jj][1].z, e_p[jj][1].w};
                float mm[8] = {e_m[jj][0].x, e_m[jj][0].y, e_m[jj][0].z, e_m[jj][0].w, e_m[jj][1].x, e_m[jj][1].y, e_m[jj][1].z, e_m[jj][1].w};
                float vv[8] = {e_v[jj][0].x, e_v[jj][0].y, e_v[jj][0].z, e_v[jj][0].w, e_v[jj][1].x, e_v[jj][1].y, e_v[jj][1].z, e_v[jj][1].w};
#pragma unroll
                for (int e = 0; e < 8; ++e) sky_adamw_update(v[e], pp[e], mm[e], vv[e], o + e < ad->n_decay, sc);
                *(float4 *)(ad->p + o) = make_float4(pp[0], pp[1], pp[2], pp[3]);
                *(float4 *)(ad->p + o + 4) = make_float4(pp[4], pp[5], pp[6], pp[7]);
                *(float4 *)(ad->m + o) = make_float4(mm[0], mm[1], mm[2], mm[3]);
                *(float4 *)(ad->m + o + 4) = make_float4(mm[4], mm[5], mm[6], mm[7]);
                *(float4 *)(ad->v + o) = make_float4(vv[0], vv[1], vv[2], vv[3]);
                *(float4 *)(ad->v + o + 4) = make_float4(vv[4], vv[5], vv[6], vv[7]);
                store8((lp_t *)ad->p_lp + o, pp);
                continue;
            }
            if (g.out_f32) {
                *(float4 *)(g.out_f32 + (int64_t)orw * g.ldo32 + n) = make_float4(v[0], v[1], v[2], v[3]);
                *(float4 *)(g.out_f32 + (int64_t)orw * g.ldo32 + n + 4) = make_float4(v[4], v[5], v[6], v[7]);
            }
            if (out) store8(out + (int64_t)orw * g.ldo + n, v);
        }
    }
#ifdef GEMM_STAMP
    {
        const unsigned long long st_stored = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long st_acked = __builtin_amdgcn_s_memtime();
        const unsigned long long st_real1 = __builtin_amdgcn_s_memrealtime();
        if (tid == 0 && g_gemm_stamp) {
            unsigned long long *o = g_gemm_stamp + (size_t)blockIdx.x * 16;
            o[0] = st_entry; o[1] = st_issued; o[2] = st_landed; o[3] = st_kloop; o[4] = st_staged; o[5] = st_stored; o[6] = st_acked;
            o[8] = st_real0; o[9] = st_real1;
            o[10] = (unsigned long long)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xf);   // HW_REG_XCC_ID, bits 3:0
        }
    }
#endif
}

// row stride of a tile shape (gemm_pipe_body, BMS): the 144-row image steps by 136 rows with 64 columns (the ViT-B decoder's
// 4352 = 32 x 136 token rows) and by 130 rows with 256 columns (ViT-L: 8320 = 64 x 130 token rows -- [8320 x 1024] outputs are
// exactly 256 tiles, [8320 x 3072] 768, [8320 x 4096] 1024: whole rounds of a 256-CU device, no row tail)
constexpr int tile_stride_m(int bm, int bn) { return bm == 144 ? (bn == 256 ? 130 : 136) : bm; }
constexpr bool pow2_rows(int r) { return r == 64 || r == 128 || r == 256; }

// The prefetch hint carried by WORKGROUPS OF ITS OWN (skyemb_gemm_args.prefetch_wgs of them, behind the launch's tiles: launch_n
// adds them where the tiles leave workgroup slots of the device free).  Carried by the tiles' own waves -- gemm_pipe_body -- the
// hint's lines, which come from HBM, sit in front of the wave's first operand stage in its in-order request queue: 0.8-1.7 us per
// launch once the operands themselves are cache hits (tools/ubench/hint_cost_probe.py).  Here nobody waits for them but the wave that
// asked: chunk c of 8 KiB = one instruction (a 4-byte LDS-DMA read per 128-byte line into the wave's 256 bytes of scratch LDS).
template <int NW>
__device__ __forceinline__ void prefetch_job(const skyemb_gemm_args &g, const unsigned int wg, const unsigned int nwg, char *smem) {
    const int lane = threadIdx.x & 63;
    const unsigned int wave = (unsigned int)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned int nchunk = (unsigned int)((g.prefetch_bytes + 8191) >> 13);
    for (unsigned int c = wg * NW + wave; c < nchunk; c += nwg * NW) {
        const long long left = g.prefetch_bytes - ((long long)c << 13);
        const unsigned int last = (unsigned int)(left < 8192 ? left : 8192) - 4u;
        const unsigned int off = (unsigned int)lane * 128u;
        glds4_sbase((const char *)g.prefetch + ((size_t)c << 13), off < last ? off : last, smem + wave * 256);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the LDS must not be handed to another workgroup with requests in flight)
}

template <int BM, int BN, bool A_KC, bool B_KC, int NSTAGE, int WM, int WN, int WK = 1>
__global__ __launch_bounds__(WM * WN * WK * 64) void gemm_pipe_kernel(const skyemb_gemm_args g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int S = g.split_k > 1 ? g.split_k : 1;           // split-K factor (host-resolved)
    const unsigned int tile_wgs = gridDim.x - (unsigned int)g.prefetch_wgs;
    if (blockIdx.x >= tile_wgs) return prefetch_job<WM * WN * WK>(g, blockIdx.x - tile_wgs, (unsigned int)g.prefetch_wgs, smem);
    unsigned int ntiles = tile_wgs, split = 0;
    if (S > 1) {
        ntiles = tile_wgs / (unsigned int)S;
        split = blockIdx.x / ntiles;
    }
    gemm_pipe_body<BM, BN, A_KC, B_KC, NSTAGE, WM, WN, WK, false, tile_stride_m(BM, BN)>(g, (int)(blockIdx.x - split * ntiles), (int)ntiles,
                                                                                      (int)split, S, smem);
}

// Grouped launch: several independent problems in ONE grid (the four weight-gradient GEMMs of a transformer block,
// optionally with the data-gradient GEMM they share an operand with: together they fill the chip, so none of them needs
// split-K or its reduce launch, and a small dgrad no longer leaves most CUs idle).  One tile shape per launch; the
// problems may be of different operand-layout classes (CLASSES = bit mask of the classes compiled into this instance:
// 1 KC.KC, 2 KC.RC, 4 RC.RC, 8 RC.KC).
// blob = [int32 n, total_blocks, 6 x pad, start[0..n] (multiples of 8), ...pad to 256 B][n x skyemb_gemm_args]
constexpr int GROUP_HEADER_BYTES = 256, GROUP_MAX = 32;
constexpr int GROUP_ADAMW_OFFSET = 176;                  // skyemb_adamw_desc (80 bytes) ends the 256-byte header
static_assert(sizeof(skyemb_adamw_desc) == 80, "the blob header reserves 80 bytes for the fused-AdamW descriptor");

// SIDE job of a grouped weight-gradient launch (skyemb_gemm_group_plan_side_adamw): the workgroups behind the launch's tiles
// (header words 2 / 3: first side workgroup, their count) stream the AdamW step of ANOTHER slice of the flat buffers -- [lo, hi),
// header words 4-7: the tensors whose gradients the PREVIOUS grouped launch of the backward pass stored -- while the tile
// workgroups run their k-loops.  Why: a grouped launch rarely fills the chip evenly (440 tiles on 512 slots at ViT-B, 384 on
// 512 in the decoder, 192 tiles of 256 x 256 on 256 CUs at ViT-L), its k-loops leave HBM idle, and an optimiser pass that waits
// for every tile's own k-loop (the epilogue form: ADAM) cannot overlap anything; side workgroups have the highest workgroup
// numbers, so they are dispatched into whatever slots the tiles leave free and, as tiles finish, into theirs.  One thread = 8
// consecutive elements per turn (two float4 of p, g, m, v in; p, m, v and the bf16 shadow out), U turns in flight.
template <int THREADS, int U>
__device__ __forceinline__ void side_adamw_job(const char *__restrict__ blob, const int wg, const int nwg) {
    const skyemb_adamw_desc *ad = (const skyemb_adamw_desc *)(blob + GROUP_ADAMW_OFFSET);
    const long long *range = (const long long *)(blob + 16);
    const int64_t lo = range[0], hi = range[1];
    const float lr = ad->hyper[0], bc1 = ad->hyper[1], bc2 = ad->hyper[2];
    const SkyAdamScalars sc = sky_adam_scalars(lr, bc1, bc2, ad->beta1, ad->beta2, ad->eps, ad->weight_decay, ad->grad_scale);
    const int64_t n8 = (hi - lo) >> 3, stride = (int64_t)nwg * THREADS, n_decay = ad->n_decay;
    float *__restrict__ P = ad->p, *__restrict__ M = ad->m, *__restrict__ V = ad->v;
    const float *__restrict__ G = ad->g_base;
    lp_t *__restrict__ PL = (lp_t *)ad->p_lp;
    for (int64_t i0 = (int64_t)wg * THREADS + threadIdx.x; i0 < n8; i0 += U * stride) {
        float4 p4[U][2], g4[U][2], m4[U][2], v4[U][2];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            int64_t i = i0 + u * stride;
            i = i < n8 ? i : n8 - 1;                      // clamped request, skipped store
            const int64_t o = lo + 8 * i;
            p4[u][0] = gload4(P + o); p4[u][1] = gload4(P + o + 4);
            g4[u][0] = gload4(G + o); g4[u][1] = gload4(G + o + 4);
            m4[u][0] = gload4(M + o); m4[u][1] = gload4(M + o + 4);
            v4[u][0] = gload4(V + o); v4[u][1] = gload4(V + o + 4);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = i0 + u * stride;
            if (i >= n8) break;
            const int64_t o = lo + 8 * i;
            float pp[8] = {p4[u][0].x, p4[u][0].y, p4[u][0].z, p4[u][0].w, p4[u][1].x, p4[u][1].y, p4[u][1].z, p4[u][1].w};
            float gg[8] = {g4[u][0].x, g4[u][0].y, g4[u][0].z, g4[u][0].w, g4[u][1].x, g4[u][1].y, g4[u][1].z, g4[u][1].w};
            float mm[8] = {m4[u][0].x, m4[u][0].y, m4[u][0].z, m4[u][0].w, m4[u][1].x, m4[u][1].y, m4[u][1].z, m4[u][1].w};
            float vv[8] = {v4[u][0].x, v4[u][0].y, v4[u][0].z, v4[u][0].w, v4[u][1].x, v4[u][1].y, v4[u][1].z, v4[u][1].w};
#pragma unroll
            for (int e = 0; e < 8; ++e) sky_adamw_update(gg[e], pp[e], mm[e], vv[e], o + e < n_decay, sc);
            *(float4 *)(P + o) = make_float4(pp[0], pp[1], pp[2], pp[3]);
            *(float4 *)(P + o + 4) = make_float4(pp[4], pp[5], pp[6], pp[7]);
            *(float4 *)(M + o) = make_float4(mm[0], mm[1], mm[2], mm[3]);
            *(float4 *)(M + o + 4) = make_float4(mm[4], mm[5], mm[6], mm[7]);
            *(float4 *)(V + o) = make_float4(vv[0], vv[1], vv[2], vv[3]);
            *(float4 *)(V + o + 4) = make_float4(vv[4], vv[5], vv[6], vv[7]);
            store8(PL + o, pp);
        }
    }
}

// The other side job (skyemb_gemm_group_attach_ln_bwd): a LayerNorm backward whose rows do not depend on this launch's tiles -- the
// block's norm1, which needs the qkv data gradient only -- taken by side workgroups (header word 41 = their count, word 42 = byte
// offset of the skyemb_ln_bwd_side record in the blob; they come before the optimiser's).  A workgroup of W waves is W / 4 blocks of
// the stand-alone kernel's partial-sum table (ln_bwd_body.h): same rows, same sums, same bits.
constexpr int GROUP_LN_COUNT_WORD = 41, GROUP_LN_OFFSET_WORD = 42;
__host__ __device__ inline int sky_ln_bwd_blocks(int M) {   // == skyemb_layernorm_bwd_blocks (layernorm.hip)
    int nb = (M + 3) / 4;
    if (nb < 1) nb = 1;
    const int rounds = (nb + SKY_LN_BWD_CAP - 1) / SKY_LN_BWD_CAP;
    return (nb + rounds - 1) / rounds;
}
// MAXNV = widest row in 256-column units the instance carries: 4 (D <= 1024) in the 256 x 256 kernel, whose waves own 236 registers
// anyway; 3 (D <= 768) in the ring-tile kernels -- a row of 1024 held in registers took them from 110 to 168 registers and the
// 128 x 128 tile from two workgroups per CU to one.
template <int THREADS, int MAXNV>
__device__ __forceinline__ void side_ln_bwd_job(const char *__restrict__ blob, const int wg, char *smem) {
    static_assert(THREADS % 256 == 0, "LayerNorm side workgroups are whole blocks of four waves");
    const skyemb_ln_bwd_side ln = *(const skyemb_ln_bwd_side *)(blob + ((const int *)blob)[GROUP_LN_OFFSET_WORD]);
    const int wave = (int)(threadIdx.x >> 6), lane = (int)(threadIdx.x & 63);
    const int nblk = sky_ln_bwd_blocks(ln.M);
    const int blk = wg * (THREADS / 256) + (wave >> 2);
    float *red = (float *)smem + (wave >> 2) * 1024;
    // gamma once into LDS behind the reduce slots (read from there row by row: ln_bwd_body.h, GAMMA_LDS)
    float *gam_lds = (float *)smem + (THREADS / 256) * 1024;
    for (int c = (int)threadIdx.x; c < ln.D; c += THREADS) gam_lds[c] = ln.gamma[c];
    __syncthreads();
    const lp_t *dy = (const lp_t *)ln.dy;
    lp_t *g_lp = (lp_t *)ln.g_lp;
    switch ((ln.D + 255) / 256) {                           // (workgroup-uniform)
        case 1: return sky_ln_bwd_rows<lp_t, lp_t, 1, true>(dy, ln.x, ln.gamma, ln.mean, ln.rstd, ln.g_in, ln.g_out, g_lp, ln.part, ln.M, ln.D, nblk, blk, wave & 3, lane, red, gam_lds);
        case 2: return sky_ln_bwd_rows<lp_t, lp_t, 2, true>(dy, ln.x, ln.gamma, ln.mean, ln.rstd, ln.g_in, ln.g_out, g_lp, ln.part, ln.M, ln.D, nblk, blk, wave & 3, lane, red, gam_lds);
        case 3: return sky_ln_bwd_rows<lp_t, lp_t, 3, true>(dy, ln.x, ln.gamma, ln.mean, ln.rstd, ln.g_in, ln.g_out, g_lp, ln.part, ln.M, ln.D, nblk, blk, wave & 3, lane, red, gam_lds);
        default:
            if constexpr (MAXNV >= 4)
                return sky_ln_bwd_rows<lp_t, lp_t, 4, true>(dy, ln.x, ln.gamma, ln.mean, ln.rstd, ln.g_in, ln.g_out, g_lp, ln.part, ln.M, ln.D, nblk, blk, wave & 3, lane, red, gam_lds);
    }
}
// which side job a workgroup behind the tiles runs: LayerNorm rows first, then the optimiser's slice
template <int THREADS, int U, int MAXNV>
__device__ __forceinline__ void side_job(const char *__restrict__ blob, char *smem) {
    const int *hdr = (const int *)blob;
    int s = (int)blockIdx.x - hdr[2];
    __builtin_amdgcn_s_setprio(0);
    const int n_ln = hdr[GROUP_LN_COUNT_WORD];
    if (s < n_ln) return side_ln_bwd_job<THREADS, MAXNV>(blob, s, smem);
    return side_adamw_job<THREADS, U>(blob, s - n_ln, hdr[3]);
}

__device__ __forceinline__ int starts_total(const int *hdr, int n) { return hdr[8 + n]; }   // padded tile count of the whole group

template <int BM, int BN, int NSTAGE, int WM, int WN, int CLASSES, int WK = 1, bool ADAM = false, bool SIDE = false>
__global__ __launch_bounds__(WM * WN * WK * 64) void gemm_pipe_group_kernel(const char *__restrict__ blob) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if constexpr (SIDE) {
        if ((int)blockIdx.x >= ((const int *)blob)[2]) return side_job<WM * WN * WK * 64, 2, 3>(blob, smem);
    }
    // the tile prefix of every problem in one scalar request (the walk `while (blockIdx.x >= hdr[9 + p]) ++p` was one dependent
    // scalar load per problem in front of every workgroup's first operand load); unused slots hold 0 and never match
    const int *hdr = (const int *)blob;
    const int n = hdr[0];
    int starts[GROUP_MAX + 1];
#pragma unroll
    for (int i = 0; i <= GROUP_MAX; ++i) starts[i] = hdr[8 + i];
    // Tile order across the GROUP (round 6, as the 256 x 256 group since round 5; header word 1 bit 30 = on): workgroup b runs on XCD
    // b & 7, and XCD x takes the x-th EIGHTH of the concatenated tile list -- consecutive tiles of mostly ONE problem -- instead of an
    // eighth of EVERY problem.  Config A encoder block (456 tiles of 128 x 128, five problems): an XCD's 57 tiles touch 16 operand
    // panels of 327 KB instead of 36, i.e. 42 instead of 94 MB enter the eight L2s per launch.
    const bool by_xcd = (hdr[1] >> 30) & 1;
    int gt = blockIdx.x;
    if (by_xcd) gt = (gt & 7) * (starts_total(hdr, n) >> 3) + (gt >> 3);
    int p = 0, first = 0;
#pragma unroll
    for (int i = 1; i < GROUP_MAX; ++i)
        if (i < n && gt >= starts[i]) { p = i; first = starts[i]; }
    const skyemb_gemm_args g = ((const skyemb_gemm_args *)(blob + GROUP_HEADER_BYTES))[p];
    const int tb = gt - first;
    const int ntiles = ((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN);
    if (tb >= ntiles) return;                             // padding up to the next multiple of 8 (keeps tb & 7 == XCD)
    const bool a = g.a_layout == SKYEMB_KC, b = g.b_layout == SKYEMB_KC;   // workgroup-uniform
    if constexpr (CLASSES & 1) if (a && b) return gemm_pipe_body<BM, BN, true, true, NSTAGE, WM, WN, WK>(g, tb, ntiles, 0, 1, smem, nullptr, by_xcd);
    if constexpr (CLASSES & 2) if (a && !b) return gemm_pipe_body<BM, BN, true, false, NSTAGE, WM, WN, WK>(g, tb, ntiles, 0, 1, smem, nullptr, by_xcd);
    if constexpr (SIDE) static_assert(CLASSES == 4, "side optimiser jobs ride in weight-gradient (RC.RC) groups only");
    if constexpr (ADAM) {
        static_assert(CLASSES == 4, "the fused optimiser step exists for weight-gradient (RC.RC) groups only");
        return gemm_pipe_body<BM, BN, false, false, NSTAGE, WM, WN, WK, true>(g, tb, ntiles, 0, 1, smem,
                                                                            (const skyemb_adamw_desc *)(blob + GROUP_ADAMW_OFFSET), by_xcd);
    }
    if constexpr (CLASSES & 4) if (!a && !b) return gemm_pipe_body<BM, BN, false, false, NSTAGE, WM, WN, WK>(g, tb, ntiles, 0, 1, smem, nullptr, by_xcd);
    if constexpr (CLASSES & 8) if (!a && b) return gemm_pipe_body<BM, BN, false, true, NSTAGE, WM, WN, WK>(g, tb, ntiles, 0, 1, smem, nullptr, by_xcd);
}

// second launch of a split-K GEMM: v = sum_s slab[s][m][n] (fixed order, alpha already applied), then
// the full epilogue of the contract; colsum[m] = sum_s cs[s][m].
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const skyemb_gemm_args g, int S) {
    const float *ws = (const float *)g.ws;
    const int M = g.M, N = g.N;
    const int64_t mn4 = (int64_t)M * N / 4;
    lp_t *out = (lp_t *)g.out;
    lp_t *out2 = (lp_t *)g.out2;
    const lp_t *aux = (const lp_t *)g.aux;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < mn4; i += (int64_t)gridDim.x * 256) {
        float4 a = *(const float4 *)(ws + 4 * i);
        for (int s = 1; s < S; ++s) {
            const float4 b = *(const float4 *)(ws + (int64_t)s * M * N + 4 * i);
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        }
        const int m = (int)((4 * i) / N), n = (int)((4 * i) % N);
        const int orow = g.dst_row ? g.dst_row[m] : m;
        if (orow < 0) continue;
        float v[4] = {a.x, a.y, a.z, a.w};
        if (g.bias) {
            const float4 t = *(const float4 *)(g.bias + n);
            v[0] += t.x; v[1] += t.y; v[2] += t.z; v[3] += t.w;
        }
        if (g.table) {
            const float4 t = *(const float4 *)(g.table + (int64_t)g.tab_row[m] * g.ldt + n);
            v[0] += t.x; v[1] += t.y; v[2] += t.z; v[3] += t.w;
        }
        if (g.resid) {
            const float4 t = *(const float4 *)(g.resid + (int64_t)orow * g.ldr + n);
            v[0] += t.x; v[1] += t.y; v[2] += t.z; v[3] += t.w;
        }
        if (g.act == SKYEMB_ACT_GELU) {
            if (out2) store4<lp_t>(out2 + (int64_t)orow * g.ldo2 + n, v[0], v[1], v[2], v[3]);
            v[0] = gelu_f(v[0]); v[1] = gelu_f(v[1]); v[2] = gelu_f(v[2]); v[3] = gelu_f(v[3]);
        } else if (g.act == SKYEMB_ACT_DGELU) {
            const float4 t = load4<lp_t>(aux + (int64_t)m * g.ldaux + n);
            v[0] *= dgelu_f(t.x); v[1] *= dgelu_f(t.y); v[2] *= dgelu_f(t.z); v[3] *= dgelu_f(t.w);
        }
        if (g.out_f32) *(float4 *)(g.out_f32 + (int64_t)orow * g.ldo32 + n) = make_float4(v[0], v[1], v[2], v[3]);
        if (out) store4<lp_t>(out + (int64_t)orow * g.ldo + n, v[0], v[1], v[2], v[3]);
    }
    if (g.colsum_a && g.a_layout == SKYEMB_RC && blockIdx.x == 0) {
        const float *cs = ws + (int64_t)S * M * N;
        for (int m = threadIdx.x; m < M; m += 256) {
            float t = cs[m];
            for (int s = 1; s < S; ++s) t += cs[(int64_t)s * M + m];
            g.colsum_a[m] = t;
        }
    }
}

template <int BM, int BN, bool A_KC, bool B_KC, int NSTAGE, int WM, int WN, int WK>
int launch_n(const skyemb_gemm_args &g, hipStream_t st) {
    constexpr size_t ring = (size_t)NSTAGE * (BM + BN) * BK * 2 * WK, image = (size_t)BM * (BN * 4 + 16) + (WK > 1 ? BM * 4 : 0);   // k-loop ring / epilogue tile (+ column sums)
    constexpr size_t smem = ring > image ? ring : image;
    auto kern = gemm_pipe_kernel<BM, BN, A_KC, B_KC, NSTAGE, WM, WN, WK>;
    // the dynamic-LDS limit is an attribute of the function PER DEVICE
    static std::mutex attr_mutex;
    static bool attr_done[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    {
        std::lock_guard<std::mutex> lock(attr_mutex);
        if (!attr_done[dev & 63]) {
            hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
            if (e != hipSuccess) {
                skyemb_set_error("skyemb_gemm(pipe): hipFuncSetAttribute(%zu B LDS): %s", smem, hipGetErrorString(e));
                return 2;
            }
            attr_done[dev & 63] = true;
        }
    }
    const int64_t tiles = ceil_div64(g.M, tile_stride_m(BM, BN)) * ceil_div64(g.N, BN);
    const int S = g.split_k > 1 ? g.split_k : 1;
    // the prefetch hint goes to workgroups of its own where the tiles leave slots of the device free (prefetch_job), else it stays
    // with the tiles' waves (gemm_pipe_body): slots = resident workgroups per CU of this instance x CUs
    skyemb_gemm_args gl = g;
    gl.prefetch_wgs = 0;
    if (g.prefetch != nullptr && g.prefetch_bytes >= 4) {
        static int slots[64] = {};
        static const bool hint_wgs_on = []() { const char *e = getenv("SKYEMB_PREFETCH_WGS"); return !(e && e[0] == '0'); }();
        {
            std::lock_guard<std::mutex> lock(attr_mutex);
            if (slots[dev & 63] == 0) {
                int per_cu = 0, cus = 0;
                if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)kern, WM * WN * WK * 64, smem) != hipSuccess) per_cu = 1;
                if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
                slots[dev & 63] = (per_cu > 0 ? per_cu : 1) * cus;
            }
        }
        const int64_t free_slots = slots[dev & 63] - tiles * S;
        if (hint_wgs_on && free_slots >= 8) gl.prefetch_wgs = (int)(free_slots < 32 ? free_slots : 32);
    } else {
        gl.prefetch = nullptr;
        gl.prefetch_bytes = 0;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)(tiles * S + gl.prefetch_wgs)), dim3(WM * WN * WK * 64), smem, st, gl);
    skyemb_count_gemm(SKYEMB_GEMM_COUNT_PIPE);
    if (S > 1) {
        int64_t blocks = ceil_div64((int64_t)g.M * g.N / 4, 256);
        if (blocks > 1024) blocks = 1024;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, st, g, S);
        skyemb_count_gemm(SKYEMB_GEMM_COUNT_SPLITK);
    }
    SKY_LAUNCH_CHECK("skyemb_gemm(pipe)");
    return 0;
}

template <int BM, int BN, int NSTAGE, int WM, int WN, int WK>
int dispatch(const skyemb_gemm_args &g, hipStream_t st) {
    const bool a = g.a_layout == SKYEMB_KC, b = g.b_layout == SKYEMB_KC;
    if constexpr (WK == 2) {
        if ((g.K / BK) % 2 != 0) {                          // a ring stage holds two k-tiles
            skyemb_set_error("skyemb_gemm(pipe): the two-k-group tiles need K %% 128 == 0");
            return 1;
        }
    }
    // row-contiguous operand tiles exist with 64 / 128 / 256 rows (one LDS-DMA instruction = a whole number of k-rows)
    if (a && b) return launch_n<BM, BN, true, true, NSTAGE, WM, WN, WK>(g, st);
    if constexpr (pow2_rows(BN)) {
        if (a && !b) return launch_n<BM, BN, true, false, NSTAGE, WM, WN, WK>(g, st);
    }
    if constexpr (pow2_rows(BM) && pow2_rows(BN)) {
        if (!a && !b) return launch_n<BM, BN, false, false, NSTAGE, WM, WN, WK>(g, st);
    }
    if constexpr (pow2_rows(BM)) {
        if (!a && b) return launch_n<BM, BN, false, true, NSTAGE, WM, WN, WK>(g, st);
    }
    skyemb_set_error("skyemb_gemm(pipe): the %d x %d tile is not built for this operand layout", BM, BN);
    return 1;
}

// Launch shapes.  code = variant * 1,000,000 + BM * 1000 + BN; X(variant, BM, BN, NSTAGE, WM, WN, WK).
// LDS per workgroup = NSTAGE * (BM + BN) * 128 B, which fixes the workgroups per CU (160 KiB): 64x64 x3 = 48 KB -> 3, x2 = 32 KB -> 5
// (code 6064064, tuned table);
// 128x64 x3 = 72 KB -> 2, x2 = 48 KB -> 3 (code 6128064: the decoder's [4352 x 2048] launches, tuned table);
// 128x128 x3 = 96 KB -> 1, x2 = 64 KB -> 2; 256x128 x3 = 144 KB -> 1.  The 128x128 tile ships with the
// 2-stage ring: two resident workgroups (16 waves) cover each other's barriers, which a third stage did not (ViT-L shapes,
// tools/ubench/gemm_lab with LAB_VITL=1: [8320 x 3072 x 1024] 93 us on 128x64, 86 us on 128x128 x3, 74 us on 128x128 x2).
#define SKY_GEMM_PRODUCT_VARIANTS(X) \
    X(0, 64, 64, 3, 2, 2, 1)         \
    X(0, 128, 64, 3, 4, 2, 1)        \
    X(0, 128, 128, 2, 4, 2, 1)       \
    X(2, 256, 128, 3, 4, 4, 1)       \
    X(6, 128, 64, 2, 4, 2, 1)        \
    X(6, 64, 64, 2, 2, 2, 1)         \
    X(9, 64, 64, 3, 2, 2, 2)         \
    X(9, 128, 128, 2, 4, 2, 2)       \
    X(9, 144, 64, 3, 3, 2, 2)        \
    X(13, 144, 256, 3, 3, 4, 1)
#ifdef SKY_GEMM_LAB   // experiment builds (tools/ubench/gemm_lab.hip): every shape under study
#define SKY_GEMM_VARIANTS(X) SKY_GEMM_PRODUCT_VARIANTS(X) SKY_GEMM_LAB_VARIANTS(X)
#else
#define SKY_GEMM_VARIANTS(X) SKY_GEMM_PRODUCT_VARIANTS(X)
#endif

int dispatch_code(int code, const skyemb_gemm_args &g, hipStream_t st) {
#define X(V, BM_, BN_, NS, WM_, WN_, WK_) \
    if (code == V * 1000000 + BM_ * 1000 + BN_) return dispatch<BM_, BN_, NS, WM_, WN_, WK_>(g, st);
    SKY_GEMM_VARIANTS(X)
#undef X
    skyemb_set_error("skyemb_gemm(pipe): unknown tile code %d", code);
    return 1;
}
void tile_dims(int code, int &bm, int &bn) {   // bm = the row STRIDE of the tiles (136 for the 144-row image)
    bn = code % 1000;
    bm = tile_stride_m((code % 1000000) / 1000, bn);
}
int canonical_tile(int tile) {   // legacy codes of the round-1 ABI
    return tile == 64 ? 64064 : tile == 128 ? 128128 : tile == 12864 ? 128064 : tile;
}

#include "gemm_pipe256.h"

struct TunedGemm {
    int M, N, K, a_kc, b_kc, tile, split;
};
const TunedGemm kTuned[] = {
#include "gemm_tuned.h"
    {0, 0, 0, 0, 0, 0, 0}};

}  // namespace

// returns -1 when the problem is outside the fast-path subset (caller falls back to gemm.hip)
int SKY_TWIN(skyemb_gemm_pipe_try)(const skyemb_gemm_args &g_in, hipStream_t st) {
    skyemb_gemm_args g = g_in;
    if (g.dtype != SKY_LP_DTYPE || g.K % BK != 0 || g.N % 8 != 0) return -1;
    // alignment of the vectorised epilogue operands
    if ((g.ldo32 % 4) || (g.ldo % 8) || (g.ldo2 % 8) || (g.ldr % 4) || (g.ldt % 4) || (g.ldaux % 8)) return -1;
    if (!aligned16(g.out) || !aligned16(g.out2) || !aligned16(g.aux) || !aligned16(g.out_f32) || !aligned16(g.resid) ||
        !aligned16(g.bias) || !aligned16(g.table))
        return -1;
    if (g.a_layout == SKYEMB_KC ? g.M < 1 : (g.M % 8 != 0 || g.M < 8)) return -1;
    if (g.b_layout == SKYEMB_KC ? g.N < 1 : (g.N % 8 != 0 || g.N < 8)) return -1;
    int tile = canonical_tile(g.tile);
    int want_split = g.split_k;
    if (tile == 0 && want_split == 0) {
        // launch shapes tuned on hardware (tools/gemm_tune.py); anything else goes through the heuristic below
        static const bool use_table = []() { const char *e = getenv("SKYEMB_GEMM_TUNED"); return !(e && e[0] == '0'); }();
        if (use_table)
            for (const TunedGemm *t = kTuned; t->M; ++t)
                if (t->M == g.M && t->N == g.N && t->K == g.K && t->a_kc == (g.a_layout == SKYEMB_KC) &&
                    t->b_kc == (g.b_layout == SKYEMB_KC) && (t->split == 1 || g.ws)) {
                    tile = canonical_tile(t->tile);
                    want_split = t->split;
                    break;
                }
    }
    if (tile == 0) {
        static const int env_tile = []() { const char *e = getenv("SKYEMB_GEMM_TILE"); return e ? atoi(e) : 0; }();   // experiments
        tile = canonical_tile(env_tile);
    }
    // measured: the 64x64 tile (3 workgroups per CU) is the best all-round choice at ViT-B sizes; launches with several
    // full rounds of 128x64 tiles (ViT-L token counts) gain ~5 % from the larger tile's lower L2->LDS traffic
    // (from 4 rounds on with long k-loops: ViT-L data gradients and fc2, 12-19 % in tools/gemm_tune.py --model mim19)
    const int64_t t12864 = ceil_div64(g.M, 128) * ceil_div64(g.N, 64), t128 = ceil_div64(g.M, 128) * ceil_div64(g.N, 128);
    // ViT-L token counts, long k-loops or the widest outputs: 256x128 with 16 waves, one workgroup per CU (fc2 forward
    // [8320 x 1024 x 4096]: 91 -> 84 us, fc1 forward [8320 x 4096 x 1024]: 128 -> 120 us; the [8320 x 3072 x 1024] launch is
    // better off with two 128x128 workgroups per CU overlapping each other's prologue and epilogue: 73 vs 80 us)
    // Two or more rounds of 256 x 256 tiles, one per CU (gemm_pipe256.h: half the L2 -> LDS bytes per FLOP of the 128 x 128 tile,
    // matrix pipe and load pipe side by side on every SIMD).  Taken when the tile count is within a sixth of whole rounds of 256 --
    // a few tiles over are cut off as a row tail below, a few under leave CUs idle for one round -- and the epilogue is one the
    // kernel has (no row maps, column sums or split-K).
    static const bool t256_on = []() { const char *e = getenv("SKYEMB_GEMM_256"); return !(e && e[0] == '0'); }();
    if (tile == 0 && t256_on && g_in.split_k <= 1 && gemm256_applicable(g)) {
        const int64_t R = ceil_div64(g.M, 256), C = ceil_div64(g.N, 256), T = R * C;
        const int64_t over = T % 256;                     // tiles beyond whole rounds
        const bool tail_ok = over > 0 && over <= C && (R - 1) * C % 256 == 0 &&              // the last row block alone is the excess
                             (g.ws || ((g.K / BK) % 2 == 0 && g.K <= 2048 && ceil_div64(g.M - (R - 1) * 256, 64) * ceil_div64(g.N, 64) <= 256));
        if (T >= 512 && g.N % 256 == 0 && (over == 0 || over >= 214 || tail_ok)) tile = 256256;
    }
    if (tile == 0 && g.a_layout == SKYEMB_KC && ((t128 >= 512 && g.K >= 2048) || (t128 >= 2048 && g.N >= 4096))) tile = 2256128;
    if (tile == 0 && (t128 >= 1024 || (t128 >= 512 && g.K >= 1024))) tile = 128128;   // >= 2 rounds of 2 workgroups per CU
    if (tile == 0 && (t12864 >= 2048 || (t12864 >= 1024 && g.K >= 1024))) tile = 128064;
    if (tile == 0) tile = 64064;
    // At most one 64x64 workgroup per CU: two k-groups of waves in the workgroup instead of an idle second wave slot per
    // SIMD (tools/ubench/gemm_lab: [1280 x 768] outputs 0.32 -> 0.20 us per k-step; K = 3072: 17.8 -> 14.3 us unsplit, where
    // the best one-group launch was a 2-way split-K plus its reduce launch).
    static const bool wk2_on = []() { const char *e = getenv("SKYEMB_GEMM_WK2"); return !(e && e[0] == '0'); }();
    if (wk2_on && tile == 64064 && g.tile == 0 && g_in.split_k <= 1 && g.a_layout == SKYEMB_KC && !g.colsum_a && (g.K / BK) % 2 == 0 &&
        g.K / BK >= 4 && ceil_div64(g.M, 64) * ceil_div64(g.N, 64) <= 256) {
        tile = 9064064;
        want_split = 1;
    }
    int bm, bn;
    tile_dims(tile, bm, bn);
    // split-K (deterministic slabs + a reduce launch that applies the epilogue) for launches with too few tiles to
    // fill the chip.  The reduce launch costs ~5 us, so a split must leave >= 10-12 k-steps per workgroup.
    int S = 1;
    if (g.ws && want_split != 1) {
        const int64_t tiles = ceil_div64(g.M, bm) * ceil_div64(g.N, bn);
        const int KT = g.K / BK;
        const int min_steps = (g.a_layout == SKYEMB_RC && g.b_layout == SKYEMB_RC) ? 10 : 12;
        S = want_split > 1 ? want_split : (int)(768 / tiles);
        if (S > 8) S = 8;
        if (want_split <= 1 && S > KT / min_steps) S = KT / min_steps;
        if (S > KT) S = KT;
        while (S > 1 && (int64_t)S * ((int64_t)g.M * g.N + g.M) * 4 > g.ws_bytes) --S;
        if (S < 1) S = 1;
    }
    g.split_k = S;
    // Row tail: with two 128x128 workgroups per CU a launch has 512 slots per round, and a few tiles more than whole rounds
    // (ViT-L: 65 row blocks of 128 token rows -> 520 / 1560 / 2080 tiles) cost a round of their own on 8-32 CUs while the
    // rest of the chip idles.  The row blocks that fill whole rounds go as one launch; the remaining rows as a second,
    // finely split launch (64x64 tiles, split-K) that is over in a fraction of a round.
    static const bool tail_on = []() { const char *e = getenv("SKYEMB_GEMM_TAIL"); return !(e && e[0] == '0'); }();
    if (tile == 256256 && (g_in.split_k > 1 || !(gemm256_applicable(g) || gemm256_wgrad_applicable(g)))) {
        skyemb_set_error("skyemb_gemm(256x256): the problem is outside this tile's subset (k-contiguous A, or a weight gradient of whole tiles; plain epilogue, K >= 128, no split-K)");
        return 1;
    }
    if (tile == 256256) S = g.split_k = 1;
    // (without a split-K workspace the tail must fit the one-launch two-k-group form)
    const bool tail_wk2 = wk2_on && (g.K / BK) % 2 == 0 && g.K <= 2048;
    if (tail_on && (tile == 128128 || tile == 2256128 || tile == 256256) && S == 1 && g.a_layout == SKYEMB_KC && (g.ws || tail_wk2) && !g.dst_row &&
        !g.tab_row && !g.colsum_a) {
        const int64_t bm_t = tile == 128128 ? 128 : 256, slots = tile == 128128 ? 512 : 256;   // 256x128, 256x256: one workgroup per CU
        const int64_t R = ceil_div64(g.M, bm_t), C = ceil_div64(g.N, tile == 256256 ? 256 : 128);
        const int64_t full = (R * C / slots) * slots;                 // tiles in whole rounds
        const int64_t Rm = full / C;                                  // row blocks of the main launch
        const int64_t tail_tiles = (R - Rm) * C;
        // (measured on the ViT-L shapes: 118 -> 92 us at one round + 8 tiles; nothing gained at three or four rounds)
        const int64_t t64_tail = ceil_div64(g.M - Rm * bm_t, 64) * ceil_div64(g.N, 64);
        if (Rm >= 1 && Rm < R && tail_tiles <= 64 && full - Rm * C < C && (full <= 2 * slots || tile == 2256128 || tile == 256256) &&
            (g.ws || t64_tail <= 256)) {
            const int64_t r0 = Rm * bm_t;
            skyemb_gemm_args gm = g, gt = g;
            gm.M = (int)r0;
            gt.M = g.M - (int)r0;
            gt.A = (const char *)g.A + r0 * g.lda * 2;
            if (g.out) gt.out = (char *)g.out + r0 * g.ldo * 2;
            if (g.out2) gt.out2 = (char *)g.out2 + r0 * g.ldo2 * 2;
            if (g.out_f32) gt.out_f32 = g.out_f32 + r0 * g.ldo32;
            if (g.resid) gt.resid = g.resid + r0 * g.ldr;
            if (g.aux) gt.aux = (const char *)g.aux + r0 * g.ldaux * 2;
            const int64_t t64 = ceil_div64(gt.M, 64) * ceil_div64(gt.N, 64);
            int St = (int)(768 / t64);
            if (St > 8) St = 8;
            if (St > g.K / BK / 4) St = g.K / BK / 4;
            while (St > 1 && (int64_t)St * ((int64_t)gt.M * gt.N + gt.M) * 4 > g.ws_bytes) --St;
            if (St < 1 || !g.ws) St = 1;
            gt.tile = 64064;
            gt.split_k = St;
            gt.prefetch = nullptr;                          // (the main launch carries the hint)
            gt.prefetch_bytes = 0;
            // short k-loops: the tail as ONE launch of the two-k-group tile (at most one workgroup per CU, no slabs, no reduce
            // launch -- mim_19 spent 145 reduce launches per step on its row tails)
            if (wk2_on && (g.K / BK) % 2 == 0 && g.K <= 2048 && t64 <= 256) {
                gt.tile = 9064064;
                gt.split_k = 1;
            }
            const int rc = tile == 256256 ? gemm256_launch(gm, st) : dispatch_code(tile, gm, st);
            if (rc != 0) return rc;
            return SKY_TWIN(skyemb_gemm_pipe_try)(gt, st);
        }
    }
    if (tile == 256256) return gemm256_launch(g, st);
    return dispatch_code(tile, g, st);
}

// ---- grouped launch (see gemm_pipe_group_kernel) -------------------------------------------------------
// [header 256 B][n problems][skyemb_ln_bwd_side, 128 B reserved]
constexpr int GROUP_TAIL_BYTES = 128;
static_assert(sizeof(skyemb_ln_bwd_side) <= GROUP_TAIL_BYTES, "the blob's tail holds the LayerNorm side job's record");
// (the plans are host logic common to both 16-bit formats: built once, in the bf16 object; only the launch has a twin)
#ifndef SKY_F16
extern "C" __attribute__((visibility("hidden"))) int skyemb_gemm_group_launch_f16(const void *blob_dev, const skyemb_gemm_group_info *info, void *stream);
extern "C" int64_t skyemb_gemm_group_blob_bytes(int n) { return GROUP_HEADER_BYTES + (int64_t)n * sizeof(skyemb_gemm_args) + GROUP_TAIL_BYTES; }

static int class_bit(const skyemb_gemm_args &g) {
    const bool a = g.a_layout == SKYEMB_KC, b = g.b_layout == SKYEMB_KC;
    return a ? (b ? 1 : 2) : (b ? 8 : 4);
}

extern "C" int skyemb_gemm_group_plan(const skyemb_gemm_args *args, int n, int tile, void *blob_host, int64_t blob_bytes,
                                      skyemb_gemm_group_info *info) {
    SKY_CHECK_ARG(args && blob_host && info && n >= 1 && n <= GROUP_MAX, "skyemb_gemm_group_plan: 1..%d problems", GROUP_MAX);
    SKY_CHECK_ARG(blob_bytes >= skyemb_gemm_group_blob_bytes(n), "skyemb_gemm_group_plan: blob too small");
    tile = canonical_tile(tile);
    if (tile == 0) {
        // default: 128x64 tiles once they still give every CU two workgroups' worth of tiles, else 64x64
        int64_t t12864 = 0;
        for (int i = 0; i < n; ++i) t12864 += ceil_div64(args[i].M, 128) * ceil_div64(args[i].N, 64);
        tile = t12864 >= 320 ? 128064 : 64064;
        // ViT-L weight gradients (>= 2.5 rounds of 128x128 tiles, K = thousands of token rows): 128x128 (mim_19: 35.8 -> 35.2 ms/step)
        int64_t t128 = 0;
        for (int i = 0; i < n; ++i) t128 += ceil_div64(args[i].M, 128) * ceil_div64(args[i].N, 128);
        if (t128 >= 400) tile = 128128;
        // at most one 128x128 workgroup per CU (the decoder's four weight gradients: 192 tiles, K = 4352 token rows): two k-groups
        static const int wk2 = []() { const char *e = getenv("SKYEMB_GROUP_WK2"); return e ? atoi(e) : 0; }();
        if (wk2 && t128 <= 256) {
            bool even = true;
            for (int i = 0; i < n; ++i) even = even && (args[i].K / BK) % 2 == 0 && args[i].K >= 4 * BK;
            if (even) tile = 9128128;
        }
        // ViT-L weight gradients as ONE round of 256x256 tiles (mim_19: 192 tiles over 8320 token rows: 1.85 us per k-tile and
        // workgroup against 4 x 0.82 on the 128x128 tile)
        static const bool g256 = []() { const char *e = getenv("SKYEMB_GROUP_256"); return !(e && e[0] == '0'); }();
        if (g256) {
            int64_t t256 = 0;
            bool ok = true;
            for (int i = 0; i < n; ++i) {
                ok = ok && gemm256_wgrad_applicable(args[i]) && args[i].K >= 2048 && !args[i].bias && !args[i].resid &&
                     args[i].act == SKYEMB_ACT_NONE && !args[i].out2;
                t256 += (args[i].M / 256) * (args[i].N / 256);
            }
            if (ok && t256 >= 160 && t256 <= 256) tile = 256256;
        }
        static const int env_tile = []() { const char *e = getenv("SKYEMB_GROUP_TILE"); return e ? atoi(e) : 0; }();   // experiments
        if (env_tile) tile = canonical_tile(env_tile);
    }
    SKY_CHECK_ARG(tile == 64064 || tile == 128064 || tile == 128128 || tile == 9128128 || tile == 256256,
                  "skyemb_gemm_group_plan: tile %d is not built for grouped launches", tile);
    if (tile == 256256)
        for (int i = 0; i < n; ++i)
            SKY_CHECK_ARG(gemm256_wgrad_applicable(args[i]) && !args[i].bias && !args[i].resid && args[i].act == SKYEMB_ACT_NONE && !args[i].out2,
                          "skyemb_gemm_group_plan: the 256x256 tile takes weight gradients of whole tiles only (problem %d)", i);
    if (tile == 9128128)
        for (int i = 0; i < n; ++i)
            SKY_CHECK_ARG((args[i].K / BK) % 2 == 0, "skyemb_gemm_group_plan: the two-k-group tile needs K %% 128 == 0 (problem %d)", i);
    int bm, bn;
    tile_dims(tile, bm, bn);
    int *hdr = (int *)blob_host;
    memset(blob_host, 0, GROUP_HEADER_BYTES);
    skyemb_gemm_args *out = (skyemb_gemm_args *)((char *)blob_host + GROUP_HEADER_BYTES);
    int start = 0, mask = 0;
    for (int i = 0; i < n; ++i) {
        skyemb_gemm_args g = args[i];
        const bool ok = sky_is_lp(g.dtype) && g.dtype == args[0].dtype && g.K % BK == 0 && g.K >= BK && g.N % 8 == 0 && !(g.ldo32 % 4) && !(g.ldo % 8) &&
                        !(g.ldo2 % 8) && !(g.ldr % 4) && !(g.ldt % 4) && !(g.ldaux % 8) &&
                        (g.a_layout == SKYEMB_KC ? g.M >= 1 : (g.M % 8 == 0 && g.M >= 8)) &&
                        (g.b_layout == SKYEMB_KC ? g.N >= 1 : (g.N % 8 == 0 && g.N >= 8)) && aligned16(g.A) && aligned16(g.B) &&
                        aligned16(g.out) && aligned16(g.out2) && aligned16(g.aux) && aligned16(g.out_f32) && aligned16(g.resid) &&
                        aligned16(g.bias) && aligned16(g.table) && (g.out || g.out_f32);
        if (!ok) {
            skyemb_set_error("skyemb_gemm_group_plan: problem %d is outside the pipelined 16-bit subset (or mixes formats)", i);
            return -1;
        }
        g.split_k = 1;
        g.tile = tile;
        g.prefetch_wgs = 0;                                // (a grouped launch's hint stays with its tiles' waves)
        out[i] = g;
        mask |= class_bit(g);
        hdr[8 + i] = start;
        const int64_t tiles = ceil_div64(g.M, bm) * ceil_div64(g.N, bn);
        start += (int)((tiles + 7) / 8 * 8);
    }
    // instances are built for: one class alone, and data-gradient (KC.RC) + weight-gradient (RC.RC) together
    if (!(mask == 1 || mask == 2 || mask == 4 || mask == 6)) {
        skyemb_set_error("skyemb_gemm_group_plan: operand-layout mix %d is not built (single class, or KC.RC with RC.RC)", mask);
        return -1;
    }
    hdr[8 + n] = start;
    hdr[0] = n;
    hdr[1] = start;
    {
        // tile order across the group: an XCD takes consecutive tiles of the concatenated list.  Default: the 256 x 256 groups
        // (round 5: their L2 traffic 2.6x -> 1.35x of the operands).  For the ring-tile groups the same order was built in round 6
        // and measured (tools/group_dec_time.py, rotating operand sets): config-A decoder group 65.5 -> 66.8 us plain, 78.1 -> 80.8
        // with the optimiser side job; encoder group 37.8 -> 35.1 us plain but 69.1 -> 70.4 with the optimiser in its epilogue (the
        // form the step runs); in the step 4.735 -> 4.743 ms: not shipped, SKYEMB_GROUP_XCD_ORDER=1 turns it on, =0 turns both off.
        // (read per plan, not once per process: bench.py builds both orders in one process for its interleaved A/B)
        const char *e = getenv("SKYEMB_GROUP_XCD_ORDER");
        const bool off = e && e[0] == '0', all = e && e[0] == '1';
        if (!off && (tile == 256256 || all)) hdr[1] |= 1 << 30;
    }
    info->total_blocks = start;
    info->tile = tile;
    info->class_mask = mask;
    info->reserved = args[0].dtype == SKYEMB_F16 ? 4 : 0;     // bit 2: the problems are SKYEMB_F16 (the launch takes the _f16 twin)
    return 0;
}

extern "C" int skyemb_gemm_group_plan_adamw(const skyemb_gemm_args *args, int n, int tile, const skyemb_adamw_desc *adamw, void *blob_host,
                                            int64_t blob_bytes, skyemb_gemm_group_info *info) {
    SKY_CHECK_ARG(adamw && adamw->g_base && adamw->p && adamw->m && adamw->v && adamw->p_lp && adamw->hyper,
                  "skyemb_gemm_group_plan_adamw: incomplete descriptor");
    const int rc = skyemb_gemm_group_plan(args, n, tile, blob_host, blob_bytes, info);
    if (rc != 0) return rc;
    if (info->class_mask != 4 || !(info->tile == 64064 || info->tile == 128064 || info->tile == 128128 || info->tile == 256256)) {
        skyemb_set_error("skyemb_gemm_group_plan_adamw: weight-gradient (RC.RC) problems on the 64x64 / 128x64 / 128x128 / 256x256 tiles only");
        return -1;
    }
    for (int i = 0; i < n; ++i) {
        // a whole number of 8-element pieces per output row, inside the flat buffers, no fused extras besides the bias gradient
        const skyemb_gemm_args &g = args[i];
        if (!g.out_f32 || g.out || g.out2 || g.bias || g.table || g.resid || g.dst_row || g.act != SKYEMB_ACT_NONE || g.alpha != 1.0f ||
            g.out_f32 < adamw->g_base || ((g.out_f32 - adamw->g_base) % 8) != 0 || (g.ldo32 ? g.ldo32 : g.N) % 8 != 0) {
            skyemb_set_error("skyemb_gemm_group_plan_adamw: problem %d is not a plain weight gradient into the flat buffer", i);
            return -1;
        }
    }
    skyemb_adamw_desc d = *adamw;
    d.enabled = 1;
    memcpy((char *)blob_host + GROUP_ADAMW_OFFSET, &d, sizeof d);
    info->reserved = (info->reserved & 4) | 1;
    return 0;
}

extern "C" int skyemb_gemm_group_plan_side_adamw(const skyemb_gemm_args *args, int n, int tile, const skyemb_adamw_desc *adamw, int own_step,
                                                 int64_t side_lo, int64_t side_hi, int side_blocks, void *blob_host, int64_t blob_bytes,
                                                 skyemb_gemm_group_info *info) {
    SKY_CHECK_ARG(adamw && adamw->g_base && adamw->p && adamw->m && adamw->v && adamw->p_lp && adamw->hyper,
                  "skyemb_gemm_group_plan_side_adamw: incomplete descriptor");
    SKY_CHECK_ARG(side_lo >= 0 && side_hi >= side_lo && side_lo % 8 == 0 && side_hi % 8 == 0 && side_blocks >= 0 && side_blocks <= 65536 &&
                      (side_hi == side_lo) == (side_blocks == 0),
                  "skyemb_gemm_group_plan_side_adamw: side range [%lld, %lld) must be whole 8-element pieces with >= 1 workgroup (or empty with none)",
                  (long long)side_lo, (long long)side_hi);
    int rc = own_step ? skyemb_gemm_group_plan_adamw(args, n, tile, adamw, blob_host, blob_bytes, info)
                      : skyemb_gemm_group_plan(args, n, tile, blob_host, blob_bytes, info);
    if (rc != 0) return rc;
    if (info->class_mask != 4 || !(info->tile == 64064 || info->tile == 128064 || info->tile == 128128 || info->tile == 256256)) {
        skyemb_set_error("skyemb_gemm_group_plan_side_adamw: weight-gradient (RC.RC) problems on the 64x64 / 128x64 / 128x128 / 256x256 tiles only");
        return -1;
    }
    skyemb_adamw_desc d = *adamw;
    d.enabled = own_step ? 1 : 0;
    memcpy((char *)blob_host + GROUP_ADAMW_OFFSET, &d, sizeof d);
    int *hdr = (int *)blob_host;
    hdr[2] = info->total_blocks;                          // first side workgroup
    hdr[3] = side_blocks;
    const long long range[2] = {(long long)side_lo, (long long)side_hi};
    memcpy(hdr + 4, range, sizeof range);
    info->total_blocks += side_blocks;
    hdr[1] = (hdr[1] & (1 << 30)) | info->total_blocks;
    info->reserved = (info->reserved & 4) | (own_step ? 1 : 0) | 2;
    return 0;
}

extern "C" int skyemb_gemm_group_attach_ln_bwd(void *blob_host, int64_t blob_bytes, skyemb_gemm_group_info *info, const skyemb_ln_bwd_side *ln) {
    SKY_CHECK_ARG(blob_host && info && ln && info->total_blocks > 0, "skyemb_gemm_group_attach_ln_bwd: bad arguments");
    int *hdr = (int *)blob_host;
    const int n = hdr[0];
    SKY_CHECK_ARG(n >= 1 && n <= GROUP_MAX && blob_bytes >= skyemb_gemm_group_blob_bytes(n), "skyemb_gemm_group_attach_ln_bwd: not a planned blob");
    SKY_CHECK_ARG(hdr[GROUP_LN_COUNT_WORD] == 0, "skyemb_gemm_group_attach_ln_bwd: the launch already carries a LayerNorm");
    if (info->class_mask != 4 || !(info->tile == 64064 || info->tile == 128064 || info->tile == 128128 || info->tile == 256256)) {
        skyemb_set_error("skyemb_gemm_group_attach_ln_bwd: weight-gradient (RC.RC) groups on the 64x64 / 128x64 / 128x128 / 256x256 tiles only");
        return -1;
    }
    SKY_CHECK_ARG(ln->dy && ln->x && ln->gamma && ln->mean && ln->rstd && ln->g_out && ln->part && ln->M > 0 && ln->D > 0 && ln->D % 4 == 0 &&
                      aligned16(ln->dy) && aligned16(ln->x) && aligned16(ln->gamma) && aligned16(ln->g_out) && aligned16(ln->g_in) && aligned16(ln->g_lp) &&
                      aligned16(ln->part),
                  "skyemb_gemm_group_attach_ln_bwd: bad LayerNorm record (M=%d D=%d; 16-byte aligned rows)", ln->M, ln->D);
    if (ln->D > (info->tile == 256256 ? 1024 : 768)) {       // (not an error: the caller launches skyemb_layernorm_bwd itself)
        skyemb_set_error("skyemb_gemm_group_attach_ln_bwd: rows of %d columns are wider than this tile's instance carries (1024 on 256 x 256 tiles, 768 on the others)", ln->D);
        return -1;
    }
    const int per_wg = info->tile == 64064 ? 1 : 2;          // four-wave blocks of the partial-sum table per side workgroup (256 / 512 threads)
    const int n_ln = (sky_ln_bwd_blocks(ln->M) + per_wg - 1) / per_wg;
    const int off = GROUP_HEADER_BYTES + n * (int)sizeof(skyemb_gemm_args);
    memcpy((char *)blob_host + off, ln, sizeof *ln);
    if (!(info->reserved & 2)) {                             // a launch without an optimiser side job: the side workgroups start behind its tiles
        hdr[2] = info->total_blocks;
        hdr[3] = 0;
    }
    hdr[GROUP_LN_COUNT_WORD] = n_ln;
    hdr[GROUP_LN_OFFSET_WORD] = off;
    info->total_blocks += n_ln;
    hdr[1] = (hdr[1] & (1 << 30)) | info->total_blocks;
    info->reserved |= 2;
    return 0;
}
#endif   // !SKY_F16

template <int BM, int BN, int NSTAGE, int WM, int WN, int CLASSES, int WK = 1, bool ADAM = false, bool SIDE = false>
static int group_launch_n(const void *blob_dev, int total_blocks, hipStream_t st) {
    constexpr size_t ring = (size_t)NSTAGE * (BM + BN) * BK * 2 * WK, image = (size_t)BM * (BN * 4 + 16) + (WK > 1 ? BM * 4 : 0);
    constexpr size_t smem = ring > image ? ring : image;
    auto kern = gemm_pipe_group_kernel<BM, BN, NSTAGE, WM, WN, CLASSES, WK, ADAM, SIDE>;
    static std::mutex attr_mutex;
    static bool attr_done[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    {
        std::lock_guard<std::mutex> lock(attr_mutex);
        if (!attr_done[dev & 63]) {
            hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
            if (e != hipSuccess) {
                skyemb_set_error("skyemb_gemm_group_launch: hipFuncSetAttribute(%zu B LDS): %s", smem, hipGetErrorString(e));
                return 2;
            }
            attr_done[dev & 63] = true;
        }
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)total_blocks), dim3(WM * WN * WK * 64), smem, st, (const char *)blob_dev);
    skyemb_count_gemm(SKYEMB_GEMM_COUNT_GROUP);
    SKY_LAUNCH_CHECK("skyemb_gemm_group_launch");
    return 0;
}
template <int BM, int BN, int NSTAGE, int WM, int WN, int WK = 1>
static int group_launch_classes(const void *blob_dev, int total_blocks, int mask, hipStream_t st) {
    switch (mask) {
        case 1: return group_launch_n<BM, BN, NSTAGE, WM, WN, 1, WK>(blob_dev, total_blocks, st);
        case 2: return group_launch_n<BM, BN, NSTAGE, WM, WN, 2, WK>(blob_dev, total_blocks, st);
        case 4: return group_launch_n<BM, BN, NSTAGE, WM, WN, 4, WK>(blob_dev, total_blocks, st);
        case 6: if constexpr (WK == 1) return group_launch_n<BM, BN, NSTAGE, WM, WN, 6, WK>(blob_dev, total_blocks, st); else break;
    }
    skyemb_set_error("skyemb_gemm_group_launch: class mask %d not built", mask);
    return 1;
}

extern "C" SKY_TWIN_VIS int SKY_TWIN(skyemb_gemm_group_launch)(const void *blob_dev, const skyemb_gemm_group_info *info, void *stream) {
    SKY_CHECK_ARG(blob_dev && info && info->total_blocks > 0, "skyemb_gemm_group_launch: bad arguments");
#ifndef SKY_F16
    if (info->reserved & 4) return skyemb_gemm_group_launch_f16(blob_dev, info, stream);
#endif
    if (skyemb_skip_mask() & 1) return 0;
    hipStream_t st = (hipStream_t)stream;
    const int mode = info->reserved & 3;
    if (mode == 2 || mode == 3) {          // side optimiser job (plan_side_adamw), own tiles stored (2) or stepped (3)
        const bool own = mode == 3;
        switch (info->tile) {
            case 64064: return own ? group_launch_n<64, 64, 3, 2, 2, 4, 1, true, true>(blob_dev, info->total_blocks, st)
                                   : group_launch_n<64, 64, 3, 2, 2, 4, 1, false, true>(blob_dev, info->total_blocks, st);
            case 128064: return own ? group_launch_n<128, 64, 3, 4, 2, 4, 1, true, true>(blob_dev, info->total_blocks, st)
                                    : group_launch_n<128, 64, 3, 4, 2, 4, 1, false, true>(blob_dev, info->total_blocks, st);
            case 128128: return own ? group_launch_n<128, 128, 2, 4, 2, 4, 1, true, true>(blob_dev, info->total_blocks, st)
                                    : group_launch_n<128, 128, 2, 4, 2, 4, 1, false, true>(blob_dev, info->total_blocks, st);
            case 256256: return own ? gemm256_group_launch<true, true>(blob_dev, info->total_blocks, st)
                                    : gemm256_group_launch<false, true>(blob_dev, info->total_blocks, st);
        }
        skyemb_set_error("skyemb_gemm_group_launch: tile %d not built with a side optimiser job", info->tile);
        return 1;
    }
    if (mode == 1) {                                 // optimiser step fused into the epilogue (plan_adamw: class 4 only)
        switch (info->tile) {
            case 64064: return group_launch_n<64, 64, 3, 2, 2, 4, 1, true>(blob_dev, info->total_blocks, st);
            case 128064: return group_launch_n<128, 64, 3, 4, 2, 4, 1, true>(blob_dev, info->total_blocks, st);
            case 128128: return group_launch_n<128, 128, 2, 4, 2, 4, 1, true>(blob_dev, info->total_blocks, st);
            case 256256: return gemm256_group_launch<true>(blob_dev, info->total_blocks, st);
        }
        skyemb_set_error("skyemb_gemm_group_launch: tile %d not built with the fused optimiser step", info->tile);
        return 1;
    }
    switch (info->tile) {
        case 64064: return group_launch_classes<64, 64, 3, 2, 2>(blob_dev, info->total_blocks, info->class_mask, st);
        case 128064: return group_launch_classes<128, 64, 3, 4, 2>(blob_dev, info->total_blocks, info->class_mask, st);
        case 128128: return group_launch_classes<128, 128, 2, 4, 2>(blob_dev, info->total_blocks, info->class_mask, st);
        case 9128128: return group_launch_classes<128, 128, 2, 4, 2, 2>(blob_dev, info->total_blocks, info->class_mask, st);
        case 256256: return gemm256_group_launch<false>(blob_dev, info->total_blocks, st);
    }
    skyemb_set_error("skyemb_gemm_group_launch: tile %d not built", info->tile);
    return 1;
}
