// Microbenchmark: per-CU L2 -> LDS (LDS-DMA) and L2 -> VGPR streaming rates on gfx950.
// build: hipcc --offload-arch=gfx950 -O3 glds_bw.hip -o glds_bw ; run: ./glds_bw
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((address_space(1))) const void gvoid_t;
typedef __attribute__((address_space(3))) void lvoid_t;
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// each wave: iters x (P LDS-DMA pieces of 1 KiB, rows of 128 B at stride `ld` bytes like a KC GEMM tile)
template <int P, bool BARRIER, bool TO_LDS>
__global__ __launch_bounds__(256) void stream_kernel(const char *__restrict__ src, size_t region, int ld, int iters, float *out) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t base = (size_t)blockIdx.x * 65536;
    float4 acc = make_float4(0, 0, 0, 0);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int p = 0; p < P; ++p) {
            // piece origin wraps inside the region; the per-lane part (8 rows x 128 B at stride ld) stays within the 1 MiB tail margin
            const size_t piece = (base + ((size_t)(it * P + p) * 4 + wave) * 8 * ld) % region;
            const char *a = src + piece + (size_t)(lane >> 3) * ld + (lane & 7) * 16;
            if (TO_LDS) __builtin_amdgcn_global_load_lds((gvoid_t *)a, (lvoid_t *)(lds + (wave * P + p) * 1024), 16, 0, 0);
            else { float4 v = *(const float4 *)a; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
        }
        if (TO_LDS) wait_vmcnt<0>();
        if (BARRIER) __builtin_amdgcn_s_barrier();
    }
    // keep results alive without ever storing in practice (iters is never negative); out is a valid buffer
    if (iters < 0) out[threadIdx.x] = TO_LDS ? ((float *)lds)[threadIdx.x] : (acc.x + acc.y + acc.z + acc.w);
}

template <int P, bool B, bool L>
void run(const char *src, size_t region, int blocks_per_cu, int ld, const char *name, float *out) {
    const int iters = 256 / P, grid = 256 * blocks_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((stream_kernel<P, B, L>), dim3(grid), dim3(256), 4 * P * 1024, 0, src, region, ld, iters, out);
    hipEventRecord(e0);
    const int reps = 5;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((stream_kernel<P, B, L>), dim3(grid), dim3(256), 4 * P * 1024, 0, src, region, ld, iters, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    const double bytes = (double)grid * 4 * iters * P * 1024;
    printf("%-28s P=%d blocks/CU=%d ld=%5d : %7.1f us  %6.2f TB/s  %6.1f GB/s/CU\n", name, P, blocks_per_cu, ld, ms * 1e3, bytes / ms / 1e9, bytes / ms / 1e6 / 256);
}

int main() {
    const size_t region = 8u << 20;  // 8 MiB source: L2 (4 MiB per XCD) / MALL resident
    char *src; hipMalloc(&src, region + (1 << 20)); hipMemset(src, 1, region + (1 << 20));
    float *out; hipMalloc(&out, 4096);
    for (int ld : {128, 1024, 1536}) {
        for (int b : {1, 2, 3, 5}) {
            run<2, true, true>(src, region, b, ld, "glds+barrier", out);
            run<4, true, true>(src, region, b, ld, "glds+barrier", out);
            run<4, false, true>(src, region, b, ld, "glds no barrier", out);
            run<8, false, true>(src, region, b, ld, "glds no barrier", out);
            run<4, false, false>(src, region, b, ld, "global_load->VGPR", out);
            run<8, false, false>(src, region, b, ld, "global_load->VGPR", out);
        }
    }
    return 0;
}
