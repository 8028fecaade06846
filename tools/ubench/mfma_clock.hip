// The denominator of the many-query search's roofline: what a BARE fp16 16x16x32 MFMA loop delivers on
// this device -- operands in registers, no memory traffic inside the loop -- on random and on all-zero operands, at one and at two
// waves per SIMD, with the clock the chip actually holds inside the loop (delta s_memtime / delta s_memrealtime x 100 MHz,
// MI355X_MICROARCH.md "DVFS give-back" item 6; stamped once around the loop, median over workgroups).
// build: hipcc --offload-arch=gfx950 -O3 mfma_clock.hip -o mfma_clock ; run: ./mfma_clock [seconds of warm-up, default 2]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
typedef __attribute__((ext_vector_type(8))) _Float16 h8;
typedef __attribute__((ext_vector_type(4))) float f4;

template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void loop_kernel(const h8 *__restrict__ src, float *__restrict__ sink, unsigned long long *__restrict__ stamps, int iters) {
    const int lane = threadIdx.x & 63;
    h8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a[i] = src[(blockIdx.x * 8 + i) * 64 + lane];
        b[i] = src[(blockIdx.x * 8 + 4 + i) * 64 + lane];
    }
    f4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f4){0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)   // (inline asm: through the builtin the compiler shuffled accumulators between VGPRs and AGPRs inside the loop)
                asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[i][j]) : "v"(a[i]), "v"(b[j]));
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (s == 12345.678f) sink[0] = s;                       // (keeps the accumulators live)
    if (threadIdx.x == 0) {
        stamps[2 * blockIdx.x] = t1 - t0;
        stamps[2 * blockIdx.x + 1] = r1 - r0;
    }
}

template <int WAVES>
void run(const char *what, const h8 *src, float *sink, unsigned long long *stamps, int ncu, double warm_s) {
    const int iters = 20000;                                 // 320 k MFMAs per wave
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    // warm-up: back-to-back launches for `warm_s` seconds (the clock settles under load)
    hipEventRecord(e0);
    float ms = 0.f;
    do {
        for (int k = 0; k < 8; ++k) hipLaunchKernelGGL(loop_kernel<WAVES>, dim3(ncu), dim3(64 * WAVES), 0, 0, src, sink, stamps, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    } while (ms < warm_s * 1e3);
    hipEventRecord(e0);
    const int reps = 8;
    for (int k = 0; k < reps; ++k) hipLaunchKernelGGL(loop_kernel<WAVES>, dim3(ncu), dim3(64 * WAVES), 0, 0, src, sink, stamps, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(2 * ncu);
    hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> clk(ncu), cyc(ncu);
    for (int i = 0; i < ncu; ++i) {
        clk[i] = (double)h[2 * i] / (double)h[2 * i + 1] * 0.1;              // GHz (s_memrealtime ticks at 100 MHz)
        cyc[i] = (double)h[2 * i] / ((double)iters * 16.0 * (WAVES / 4.0));  // shader cycles per MFMA and SIMD
    }
    std::sort(clk.begin(), clk.end());
    std::sort(cyc.begin(), cyc.end());
    const double flops = 2.0 * 16 * 16 * 32 * 16.0 * iters * WAVES * ncu * reps;
    printf("%-34s %d wave(s)/SIMD: %7.1f TFLOP/s wall | in-loop clock %.3f GHz (min %.3f max %.3f) | %.2f cycles per MFMA and SIMD\n", what,
           WAVES / 4, flops / (ms * 1e-3) / 1e12, clk[ncu / 2], clk.front(), clk.back(), cyc[ncu / 2]);
}

int main(int argc, char **argv) {
    const double warm = argc > 1 ? atof(argv[1]) : 2.0;
    int dev = 0, ncu = 256;
    hipGetDevice(&dev);
    hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
    const size_t n = (size_t)ncu * 8 * 64;
    std::vector<_Float16> host(n * 8);
    srand(1);
    for (auto &v : host) v = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 2.0f);
    h8 *rnd, *zero;
    float *sink;
    unsigned long long *stamps;
    hipMalloc(&rnd, n * 16);
    hipMalloc(&zero, n * 16);
    hipMalloc(&sink, 64);
    hipMalloc(&stamps, (size_t)ncu * 16);
    hipMemcpy(rnd, host.data(), n * 16, hipMemcpyHostToDevice);
    hipMemset(zero, 0, n * 16);
    printf("# bare v_mfma_f32_16x16x32_f16 loop, operands in registers, %d CUs; dense fp16 peak by the data sheet: 2.5 PFLOP/s at 2.4 GHz = 16 cycles per MFMA and SIMD\n", ncu);
    run<4>("random operands", rnd, sink, stamps, ncu, warm);
    run<8>("random operands", rnd, sink, stamps, ncu, warm);
    run<4>("all-zero operands", zero, sink, stamps, ncu, warm);
    run<8>("all-zero operands", zero, sink, stamps, ncu, warm);
    return 0;
}
