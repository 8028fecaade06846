// MFMA GEMM with fused epilogues for the ViT linear layers (fwd, dgrad, wgrad).
//
// C[M,N] = alpha * A[M,K] * B[N,K]^T (+bias +table rows +fp32 residual, GELU / dGELU), where each
// operand may be stored k-contiguous (KC) or row-contiguous (RC):
//   fwd   y  = x  * W^T      A = x  (KC)   B = W  (KC)
//   dgrad dx = dy * W        A = dy (KC)   B = W  (RC)   (B(n,k) = W[k][n])
//   wgrad dW = dy^T * x      A = dy (RC)   B = x  (RC)   (contraction over tokens)
// bf16 mode: v_mfma_f32_16x16x32_bf16, fp32 accumulate; RC operands are staged [k][rows] in LDS
// and their fragments come from ds_read_b64_tr_b16 (hardware transposed read) so no transposed
// copies of activations or weights are ever materialised.
// f32 mode (parity): v_mfma_f32_16x16x4_f32, exact fp32 fma chains.
//
// Block = 256 threads = 4 waves (2x2), tile BM x BN in {128x128, 64x64}, BK = 64 (bf16) / 16 (f32),
// register-staged global->LDS double buffer (one barrier per k-tile).
#include <mutex>
#include "lp_twin.h"
#include <stdlib.h>

namespace {

template <typename T>
struct Cfg;
template <>
struct Cfg<lp_t> {
    static constexpr int BK = 64, VEC = 8, KSTEP = 32, PAD_KC = 8, PAD_RC = 16;
    typedef lp8 Frag;
};
template <>
struct Cfg<float> {
    static constexpr int BK = 16, VEC = 4, KSTEP = 4, PAD_KC = 4, PAD_RC = 16;
    typedef float Frag;
};

template <typename T, bool KC, int R>
struct TileShape {
    static constexpr int PITCH = KC ? (Cfg<T>::BK + Cfg<T>::PAD_KC) : (R + Cfg<T>::PAD_RC);
    static constexpr int ELEMS = KC ? R * PITCH : Cfg<T>::BK * PITCH;
    static constexpr int NVEC = R * Cfg<T>::BK / Cfg<T>::VEC / 256;  // 16-byte vectors per thread
};

typedef __attribute__((address_space(3))) lp4 lds_lp4_t;

// ---- fragment reads ------------------------------------------------------------------
// rbase: first tile row of this 16-row MFMA operand block (multiple of 16); lane supplies row lane&15
template <typename T, bool KC, int R>
__device__ __forceinline__ typename Cfg<T>::Frag read_frag(const T *s, int rbase, int kk, int lane);

template <>
__device__ __forceinline__ lp8 read_frag<lp_t, true, 128>(const lp_t *s, int rbase, int kk, int lane) {
    return *(const lp8 *)&s[(rbase + (lane & 15)) * TileShape<lp_t, true, 128>::PITCH + kk * 32 + 8 * (lane >> 4)];
}
template <>
__device__ __forceinline__ lp8 read_frag<lp_t, true, 64>(const lp_t *s, int rbase, int kk, int lane) {
    return *(const lp8 *)&s[(rbase + (lane & 15)) * TileShape<lp_t, true, 64>::PITCH + kk * 32 + 8 * (lane >> 4)];
}
template <int R>
__device__ __forceinline__ lp8 read_frag_rc_lp(const lp_t *s, int rbase, int kk, int lane) {
    constexpr int P = TileShape<lp_t, false, R>::PITCH;
    const int i = lane & 15, q = i >> 2, p = i & 3;
    const int kb = kk * 32 + 8 * (lane >> 4);
    // block = 4 k-rows x 16 matrix rows; lane 4q+p supplies &[kb+q][rbase + 4p]; lane i receives
    // matrix row rbase+i for k = kb..kb+3 (cdna_hip_programming.md T10)
    const lp_t *a0 = &s[(kb + q) * P + rbase + 4 * p];
    const lp_t *a1 = a0 + 4 * P;
    lp4 lo = sky_ds_read_tr16_b64((lds_lp4_t *)a0);
    lp4 hi = sky_ds_read_tr16_b64((lds_lp4_t *)a1);
    lp8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
}
template <>
__device__ __forceinline__ lp8 read_frag<lp_t, false, 128>(const lp_t *s, int rbase, int kk, int lane) {
    return read_frag_rc_lp<128>(s, rbase, kk, lane);
}
template <>
__device__ __forceinline__ lp8 read_frag<lp_t, false, 64>(const lp_t *s, int rbase, int kk, int lane) {
    return read_frag_rc_lp<64>(s, rbase, kk, lane);
}
template <>
__device__ __forceinline__ float read_frag<float, true, 128>(const float *s, int rbase, int kk, int lane) {
    return s[(rbase + (lane & 15)) * TileShape<float, true, 128>::PITCH + kk * 4 + (lane >> 4)];
}
template <>
__device__ __forceinline__ float read_frag<float, true, 64>(const float *s, int rbase, int kk, int lane) {
    return s[(rbase + (lane & 15)) * TileShape<float, true, 64>::PITCH + kk * 4 + (lane >> 4)];
}
template <>
__device__ __forceinline__ float read_frag<float, false, 128>(const float *s, int rbase, int kk, int lane) {
    return s[(kk * 4 + (lane >> 4)) * TileShape<float, false, 128>::PITCH + rbase + (lane & 15)];
}
template <>
__device__ __forceinline__ float read_frag<float, false, 64>(const float *s, int rbase, int kk, int lane) {
    return s[(kk * 4 + (lane >> 4)) * TileShape<float, false, 64>::PITCH + rbase + (lane & 15)];
}

__device__ __forceinline__ f32x4 mfma(lp8 a, lp8 b, f32x4 c) {
    return sky_mfma_16x16x32(a, b, c);
}
__device__ __forceinline__ f32x4 mfma(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// ---- global -> register -> LDS staging ---------------------------------------------------
template <typename T, bool KC, int R>
__device__ __forceinline__ void load_tile(const T *__restrict__ X, int64_t ld, int r0, int rows, int k0, int K, int tid,
                                          uint4 (&reg)[TileShape<T, KC, R>::NVEC]) {
    constexpr int VEC = Cfg<T>::VEC, BK = Cfg<T>::BK, NV = TileShape<T, KC, R>::NVEC;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int v = tid + i * 256;
        int r, k;
        if (KC) {
            r = v / (BK / VEC);
            k = (v % (BK / VEC)) * VEC;
        } else {
            k = v / (R / VEC);
            r = (v % (R / VEC)) * VEC;
        }
        const int gr = r0 + r, gk = k0 + k;
        const bool ok = gr < rows && gk < K;
        const T *ptr = KC ? (X + (int64_t)gr * ld + gk) : (X + (int64_t)gk * ld + gr);
        reg[i] = ok ? *(const uint4 *)ptr : make_uint4(0, 0, 0, 0);
    }
}

template <typename T, bool KC, int R>
__device__ __forceinline__ void store_tile(T *s, int tid, const uint4 (&reg)[TileShape<T, KC, R>::NVEC]) {
    constexpr int VEC = Cfg<T>::VEC, BK = Cfg<T>::BK, NV = TileShape<T, KC, R>::NVEC, P = TileShape<T, KC, R>::PITCH;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int v = tid + i * 256;
        int off;
        if (KC) {
            off = (v / (BK / VEC)) * P + (v % (BK / VEC)) * VEC;
        } else {
            off = (v / (R / VEC)) * P + (v % (R / VEC)) * VEC;
        }
        *(uint4 *)&s[off] = reg[i];
    }
}

__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float dgelu_f(float x) {
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * expf(-0.5f * x * x);
    return cdf + x * pdf;
}

template <typename T, int BM, int BN, bool A_KC, bool B_KC>
__global__ __launch_bounds__(256) void gemm_kernel(const skyemb_gemm_args g) {
    using C = Cfg<T>;
    using SA = TileShape<T, A_KC, BM>;
    using SB = TileShape<T, B_KC, BN>;
    constexpr int BK = C::BK;
    constexpr int TM = BM / 32, TN = BN / 32;  // 16x16 MFMA tiles per wave (2x2 waves)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T *sA0 = (T *)smem;
    T *sA1 = sA0 + SA::ELEMS;
    T *sB0 = sA1 + SA::ELEMS;
    T *sB1 = sB0 + SB::ELEMS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_n = (g.N + BN - 1) / BN;
    // XCD-aware tile order (speed only): workgroups are dealt round-robin over the 8 XCDs, each with
    // a private 4 MiB L2.  Give every XCD a CONTIGUOUS run of tiles (a few tile-rows x all tile-
    // columns) so that its A rows stay L2-resident while it sweeps B.  Bijective for any grid size.
    int wg;
    {
        const int nwg = gridDim.x, xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
        const int q = nwg >> 3, r = nwg & 7;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
    }
    const int m0 = (wg / tiles_n) * BM, n0 = (wg % tiles_n) * BN;
    const T *A = (const T *)g.A;
    const T *B = (const T *)g.B;
    const int KT = (g.K + BK - 1) / BK;

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // fused column sum of an RC A operand (bias gradient of the wgrad launch): only the first
    // column of tiles does it, thread t owns logical row m0 + t, summing the staged [k][rows] tile
    const bool do_colsum = !A_KC && g.colsum_a != nullptr && (wg % tiles_n) == 0 && tid < BM;
    float csum = 0.f;

    uint4 ra[SA::NVEC], rb[SB::NVEC];
    load_tile<T, A_KC, BM>(A, g.lda, m0, g.M, 0, g.K, tid, ra);
    load_tile<T, B_KC, BN>(B, g.ldb, n0, g.N, 0, g.K, tid, rb);
    store_tile<T, A_KC, BM>(sA0, tid, ra);
    store_tile<T, B_KC, BN>(sB0, tid, rb);
    __syncthreads();

    for (int kt = 0; kt < KT; ++kt) {
        const bool more = kt + 1 < KT;
        if (more) {
            load_tile<T, A_KC, BM>(A, g.lda, m0, g.M, (kt + 1) * BK, g.K, tid, ra);
            load_tile<T, B_KC, BN>(B, g.ldb, n0, g.N, (kt + 1) * BK, g.K, tid, rb);
        }
        const T *cA = (kt & 1) ? sA1 : sA0;
        const T *cB = (kt & 1) ? sB1 : sB0;
        if (!A_KC && do_colsum) {
#pragma unroll 8
            for (int k = 0; k < BK; ++k) csum += to_f32<T>(cA[k * SA::PITCH + tid]);
        }
#pragma unroll
        for (int kk = 0; kk < BK / C::KSTEP; ++kk) {
            typename C::Frag fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i] = read_frag<T, A_KC, BM>(cA, wm * (BM / 2) + i * 16, kk, lane);
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[j] = read_frag<T, B_KC, BN>(cB, wn * (BN / 2) + j * 16, kk, lane);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = mfma(fa[i], fb[j], acc[i][j]);
        }
        if (more) {
            store_tile<T, A_KC, BM>((kt & 1) ? sA0 : sA1, tid, ra);
            store_tile<T, B_KC, BN>((kt & 1) ? sB0 : sB1, tid, rb);
        }
        __syncthreads();
    }

    if (!A_KC && do_colsum && m0 + tid < g.M) g.colsum_a[m0 + tid] = csum;
    // ---- epilogue: C/D layout col = lane&15, row = 4*(lane>>4) + r -------------------------
    T *out = (T *)g.out;
    T *out2 = (T *)g.out2;
    const T *aux = (const T *)g.aux;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + wm * (BM / 2) + i * 16 + 4 * (lane >> 4) + r;
            if (m >= g.M) continue;
            const int orow = g.dst_row ? g.dst_row[m] : m;
            if (orow < 0) continue;
            const float *trow = g.table ? g.table + (int64_t)g.tab_row[m] * g.ldt : nullptr;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + wn * (BN / 2) + j * 16 + (lane & 15);
                if (n >= g.N) continue;
                float v = acc[i][j][r] * g.alpha;
                if (g.bias) v += g.bias[n];
                if (trow) v += trow[n];
                if (g.resid) v += g.resid[(int64_t)orow * g.ldr + n];
                if (g.act == SKYEMB_ACT_GELU) {
                    if (out2) out2[(int64_t)orow * g.ldo2 + n] = from_f32<T>(v);
                    v = gelu_f(v);
                } else if (g.act == SKYEMB_ACT_DGELU) {
                    v *= dgelu_f(to_f32<T>(aux[(int64_t)m * g.ldaux + n]));
                }
                if (g.out_f32) g.out_f32[(int64_t)orow * g.ldo32 + n] = v;
                if (out) out[(int64_t)orow * g.ldo + n] = from_f32<T>(v);
            }
        }
    }
}

template <typename T, int BM, int BN, bool A_KC, bool B_KC>
int launch(const skyemb_gemm_args &g, hipStream_t st) {
    constexpr size_t smem =
        2 * (size_t)(TileShape<T, A_KC, BM>::ELEMS + TileShape<T, B_KC, BN>::ELEMS) * sizeof(T);
    auto kern = gemm_kernel<T, BM, BN, A_KC, B_KC>;
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
                skyemb_set_error("skyemb_gemm: hipFuncSetAttribute(%zu B LDS): %s", smem, hipGetErrorString(e));
                return 2;
            }
            attr_done[dev & 63] = true;
        }
    }
    const int64_t tiles = ceil_div64(g.M, BM) * ceil_div64(g.N, BN);
    hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(256), smem, st, g);
    skyemb_count_gemm(SKYEMB_GEMM_COUNT_FALLBACK);
    SKY_LAUNCH_CHECK("skyemb_gemm");
    return 0;
}

template <typename T, int BT>
int dispatch_layout(const skyemb_gemm_args &g, hipStream_t st) {
    if (g.a_layout == SKYEMB_KC && g.b_layout == SKYEMB_KC) return launch<T, BT, BT, true, true>(g, st);
    if (g.a_layout == SKYEMB_KC && g.b_layout == SKYEMB_RC) return launch<T, BT, BT, true, false>(g, st);
    if (g.a_layout == SKYEMB_RC && g.b_layout == SKYEMB_RC) return launch<T, BT, BT, false, false>(g, st);
    if (g.a_layout == SKYEMB_RC && g.b_layout == SKYEMB_KC) return launch<T, BT, BT, false, true>(g, st);
    skyemb_set_error("skyemb_gemm: bad layouts %d/%d", g.a_layout, g.b_layout);
    return 1;
}

}  // namespace

int SKY_TWIN(skyemb_gemm_pipe_try)(const skyemb_gemm_args &g, hipStream_t st);  // gemm_pipe.hip (-1: not applicable)
#ifndef SKY_F16
extern "C" __attribute__((visibility("hidden"))) int skyemb_gemm_f16(const skyemb_gemm_args *args, void *stream);   // this file, compiled with -DSKY_F16 (lp_twin.h)
#endif

extern "C" SKY_TWIN_VIS int SKY_TWIN(skyemb_gemm)(const skyemb_gemm_args *args, void *stream) {
    SKY_CHECK_ARG(args != nullptr, "skyemb_gemm: null args");
    const skyemb_gemm_args &g = *args;
#ifndef SKY_F16
    if (g.dtype == SKYEMB_F16) return skyemb_gemm_f16(args, stream);
#endif
    SKY_CHECK_ARG(g.M > 0 && g.N > 0 && g.K > 0, "skyemb_gemm: empty problem M=%d N=%d K=%d", g.M, g.N, g.K);
    SKY_CHECK_ARG(g.dtype == SKY_LP_DTYPE || g.dtype == SKYEMB_F32, "skyemb_gemm: bad dtype %d", g.dtype);
    const int vec = g.dtype == SKY_LP_DTYPE ? 8 : 4;
    const int a_cont = g.a_layout == SKYEMB_KC ? g.K : g.M;
    const int b_cont = g.b_layout == SKYEMB_KC ? g.K : g.N;
    SKY_CHECK_ARG(a_cont % vec == 0 && b_cont % vec == 0 && g.lda % vec == 0 && g.ldb % vec == 0,
                  "skyemb_gemm: contiguous extents/ld must be multiples of %d (A %d ld %lld, B %d ld %lld)", vec, a_cont,
                  (long long)g.lda, b_cont, (long long)g.ldb);
    SKY_CHECK_ARG(aligned16(g.A) && aligned16(g.B), "skyemb_gemm: A/B must be 16-byte aligned");
    SKY_CHECK_ARG(g.out || g.out_f32, "skyemb_gemm: no output");
    SKY_CHECK_ARG(g.act != SKYEMB_ACT_DGELU || g.aux, "skyemb_gemm: ACT_DGELU needs aux");
    SKY_CHECK_ARG(!g.table || g.tab_row, "skyemb_gemm: table without tab_row");
    SKY_CHECK_ARG(!g.colsum_a || g.a_layout == SKYEMB_RC, "skyemb_gemm: colsum_a needs an RC A operand");
    hipStream_t st = (hipStream_t)stream;
    if (skyemb_skip_mask() & 1) return 0;
    static const bool use_pipe = []() { const char *e = getenv("SKYEMB_GEMM_PIPE"); return !(e && e[0] == '0'); }();
    if (use_pipe) {
        const int rc = SKY_TWIN(skyemb_gemm_pipe_try)(g, st);
        if (rc >= 0) return rc;
    }
    int tile = g.tile;
    if (tile == 0) tile = (ceil_div64(g.M, 128) * ceil_div64(g.N, 128) >= 200) ? 128 : 64;
    if (g.dtype == SKY_LP_DTYPE) {
        return tile == 128 ? dispatch_layout<lp_t, 128>(g, st) : dispatch_layout<lp_t, 64>(g, st);
    }
    return tile == 128 ? dispatch_layout<float, 128>(g, st) : dispatch_layout<float, 64>(g, st);
}
