// Lab (not a product path): the GEMM kernel shape the round-5 review asked for, measured beside the library's 256 x 256 kernel.
//   one wave per SIMD (4 waves per workgroup, up to 512 registers each: 256 accumulator registers = a 64 x 256 strip of the 256 x 256
//   tile per wave), the A operand global -> registers directly in MFMA layout (no LDS), the B operand global -> registers -> LDS -> fragments,
//   both prefetched one k-tile ahead; v_mfma_f32_16x16x32_f16.
// C[M, N] = A[M, K] B[N, K]^T, fp16 in, fp32 accumulate, fp16 out; M % 64 == 0 is NOT required (rows clamped, stores predicated),
// N % 256 == 0, K % 64 == 0.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/gemm_bigreg.hip -o /tmp/gemm_bigreg -L sky_embeddings_amd -l:libskyemb.so -Iinclude
//   LD_LIBRARY_PATH=sky_embeddings_amd /tmp/gemm_bigreg [M N K]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include "skyemb.h"

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int BM = 256, BN = 256, BK = 64, PITCH = BK * 2 + 16;     // bytes per B row in LDS
constexpr int LDS_BUF = BN * PITCH;
// (inline asm: the accumulators are PINNED to the accumulator half of the register file, the operands to the other half -- left to
// the compiler's allocator the operands landed in AGPRs, parts of the accumulators in VGPRs, and 300 registers were spilled)
__device__ __forceinline__ void mfma(f4 &c, const h8 &a, const h8 &b) {
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}

template <int KT_UNROLL = 6>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm_bigreg(const _Float16 *__restrict__ A,
                                                                                                const _Float16 *__restrict__ B,
                                                                                                _Float16 *__restrict__ C, int M, int N, int K,
                                                                                                int lda, int ldb, int ldc) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_n = N / BN, tiles_m = (M + BM - 1) / BM, ntiles = tiles_n * tiles_m;
    // XCD b & 7 takes a contiguous eighth of the tiles (row-major: consecutive tiles share the A rows)
    int tile = blockIdx.x;
    {
        const int per = (ntiles + 7) / 8;
        const int t = (tile & 7) * per + (tile >> 3);
        tile = t < ntiles && (ntiles % 8 == 0) ? t : tile;
    }
    const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
    f4 acc[4][16];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};
    const _Float16 *ap[4];
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
        int row = m0 + 64 * wave + 16 * rb + (lane & 15);
        row = row < M ? row : M - 1;
        ap[rb] = A + (size_t)row * lda + 8 * (lane >> 4);
    }
    const _Float16 *bp = B + (size_t)(n0 + (tid >> 3)) * ldb + 8 * (tid & 7);
    h8 a[2][4][2], bst[8];
    const int KT = K / BK;
    auto loadA = [&](auto buf, int kt) {
#pragma unroll
        for (int rb = 0; rb < 4; ++rb)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) a[buf][rb][ks] = *(const h8 *)(ap[rb] + (size_t)kt * BK + 32 * ks);
    };
    auto loadB = [&](int kt) {
#pragma unroll
        for (int j = 0; j < 8; ++j) bst[j] = *(const h8 *)(bp + (size_t)32 * j * ldb + (size_t)kt * BK);
    };
    auto writeB = [&](auto buf) {
        char *base = smem + buf * LDS_BUF + (tid >> 3) * PITCH + (tid & 7) * 16;
#pragma unroll
        for (int j = 0; j < 8; ++j) *(h8 *)(base + 32 * j * PITCH) = bst[j];
    };
    // a k-tile in four quarters of four column blocks: the NEXT quarter's eight fragment reads go out ahead of this quarter's 32 MFMAs
    // (hard scheduling fences: left alone, the compiler sinks every read to just before its first use)
    auto compute = [&](auto abuf, auto lbuf) {
        const char *base = smem + lbuf * LDS_BUF + (lane & 15) * PITCH + (lane >> 4) * 16;
        h8 bf[2][4][2];
        auto read_q = [&](auto set, int q) {
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) bf[set][c][ks] = *(const h8 *)(base + (4 * q + c) * 16 * PITCH + 64 * ks);
        };
        read_q(std::integral_constant<int, 0>{}, 0);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (q + 1 < 4) {
                if (q & 1) read_q(std::integral_constant<int, 0>{}, q + 1);
                else read_q(std::integral_constant<int, 1>{}, q + 1);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int rb = 0; rb < 4; ++rb)
                        mfma(acc[rb][4 * q + c], a[abuf][rb][ks], bf[q & 1][c][ks]);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // prologue
    loadA(std::integral_constant<int, 0>{}, 0);
    loadB(0);
    writeB(std::integral_constant<int, 0>{});
    __syncthreads();
    auto step = [&](auto I, int kt) {
        constexpr int i = decltype(I)::value;
        if (kt + 1 < KT) {
            loadA(std::integral_constant<int, (i + 1) % 2>{}, kt + 1);
            loadB(kt + 1);
        }
        compute(std::integral_constant<int, i % 2>{}, std::integral_constant<int, i % 2>{});
        if (kt + 1 < KT) writeB(std::integral_constant<int, (i + 1) % 2>{});
        __syncthreads();
    };
#pragma unroll 1
    for (int kt = 0; kt < KT; kt += 2) {
        step(std::integral_constant<int, 0>{}, kt);
        if (kt + 1 < KT) step(std::integral_constant<int, 1>{}, kt + 1);
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");      // (the last MFMAs have left the pipe before the accumulators are read)
    // epilogue: the wave's 64 x 256 strip through LDS (its own 33 KB slab), rows leave in 16-byte pieces
    constexpr int EP = 256 * 2 + 16;
    char *slab = smem + wave * 64 * EP;
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
        for (int cb = 0; cb < 16; ++cb)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                *(_Float16 *)(slab + (16 * rb + 4 * (lane >> 4) + i) * EP + (16 * cb + (lane & 15)) * 2) = (_Float16)acc[rb][cb][i];
    __builtin_amdgcn_s_waitcnt(0xc07f);     // lgkmcnt(0): the wave reads back only its own slab
#pragma unroll
    for (int r = 0; r < 64; r += 2) {
        const int row = r + (lane >> 5), piece = lane & 31;
        const int grow = m0 + 64 * wave + row;
        const h8 v = *(const h8 *)(slab + row * EP + piece * 16);
        if (grow < M) *(h8 *)(C + (size_t)grow * ldc + n0 + piece * 8) = v;
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

int main(int argc, char **argv) {
    const int M = argc > 3 ? atoi(argv[1]) : 8192, N = argc > 3 ? atoi(argv[2]) : 4096, K = argc > 3 ? atoi(argv[3]) : 1024;
    if (N % 256 || K % 64) { printf("N %% 256 == 0 and K %% 64 == 0\n"); return 1; }
    std::vector<_Float16> hA((size_t)M * K), hB((size_t)N * K);
    srand(1);
    for (auto &v : hA) v = (_Float16)((rand() % 2001 - 1000) / 1000.0f);
    for (auto &v : hB) v = (_Float16)((rand() % 2001 - 1000) / 4000.0f);
    _Float16 *dA, *dB, *dC, *dC2;
    CK(hipMalloc(&dA, hA.size() * 2)); CK(hipMalloc(&dB, hB.size() * 2));
    CK(hipMalloc(&dC, (size_t)M * N * 2)); CK(hipMalloc(&dC2, (size_t)M * N * 2));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, hB.data(), hB.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemset(dC, 0, (size_t)M * N * 2)); CK(hipMemset(dC2, 0, (size_t)M * N * 2));
    const int tiles = ((M + 255) / 256) * (N / 256);
    const int smem = 4 * 64 * (256 * 2 + 16);      // the epilogue's slabs (132 KB) cover the two B buffers (72 KB)
    CK(hipFuncSetAttribute((const void *)gemm_bigreg<6>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    auto run_lab = [&]() { hipLaunchKernelGGL(gemm_bigreg<6>, dim3(tiles), dim3(256), smem, st, dA, dB, dC, M, N, K, K, K, N); };
    skyemb_gemm_args g = {};
    g.A = dA; g.B = dB; g.lda = K; g.ldb = K; g.a_layout = SKYEMB_KC; g.b_layout = SKYEMB_KC; g.M = M; g.N = N; g.K = K; g.dtype = SKYEMB_F16;
    g.alpha = 1.0f; g.out = dC2; g.ldo = N;
    auto run_lib = [&](int tile) { g.tile = tile; if (skyemb_gemm(&g, st) != 0) { printf("skyemb_gemm: %s\n", skyemb_last_error()); exit(1); } };
    run_lab(); run_lib(0);
    CK(hipStreamSynchronize(st));
    std::vector<_Float16> c1((size_t)M * N), c2((size_t)M * N);
    CK(hipMemcpy(c1.data(), dC, c1.size() * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(c2.data(), dC2, c2.size() * 2, hipMemcpyDeviceToHost));
    double maxd = 0, maxv = 0;
    for (size_t i = 0; i < c1.size(); i += 7) { maxd = fmax(maxd, fabs((double)c1[i] - (double)c2[i])); maxv = fmax(maxv, fabs((double)c2[i])); }
    // a few entries against a double-precision dot product
    double maxe = 0;
    for (int s = 0; s < 64; ++s) {
        const int i = (int)((size_t)rand() % M), j = rand() % N;
        double ref = 0;
        for (int k = 0; k < K; ++k) ref += (double)hA[(size_t)i * K + k] * (double)hB[(size_t)j * K + k];
        maxe = fmax(maxe, fabs(ref - (double)c1[(size_t)i * N + j]));
    }
    printf("[%d x %d x %d] lab vs library: max |diff| %.4g (max |C| %.3g); lab vs fp64 on 64 entries: %.4g\n", M, N, K, maxd, maxv, maxe);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time = [&](auto f, const char *name) {
        for (int i = 0; i < 5; ++i) f();
        CK(hipStreamSynchronize(st));
        float best = 1e9f, sum = 0;
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < 20; ++i) f();
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            best = fminf(best, ms / 20); sum += ms / 20;
        }
        printf("  %-44s %8.1f us (best of 5 x 20; mean %.1f)  %7.0f TFLOP/s\n", name, best * 1e3, sum / 5 * 1e3, 2.0 * M * N * K / (best * 1e-3) / 1e12);
    };
    time(run_lab, "lab: one wave per SIMD, 64 x 256 per wave");
    time([&]() { run_lib(0); }, "library, plan's choice");
    time([&]() { run_lib(256256); }, "library, 256 x 256 persistent kernel");
    return 0;
}
