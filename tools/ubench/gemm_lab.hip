// GEMM lab: times launch shapes (tile code x split-K) of the pipelined bf16 GEMM on the layer shapes of one MAE ViT-B
// step, standalone (no torch), and checks every variant against a naive fp32 kernel.
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DSKY_GEMM_LAB tools/ubench/gemm_lab.hip -o tools/ubench/gemm_lab
//   run:   tools/ubench/gemm_lab [codes...]      (default: every compiled variant)
//   LAB_STAMP=1 (binary built with -DGEMM_STAMP): per-launch timeline of the product's launch shapes as JSON
//   (profiles/r03_gemm_timeline.json): marker kernel -> GEMM -> marker kernel, s_memtime / s_memrealtime stamps per workgroup.
// Timing: each (shape, config) is launched over a rotation of 6 operand sets (the weights of six different layers: cold
// in L2, as in the training step, where a layer's weights were last read a whole step earlier) between two HIP events.
#define SKY_GEMM_LAB_VARIANTS(X) \
    X(1, 64, 64, 4, 2, 2, 1)        \
    X(7, 64, 64, 6, 2, 2, 1)        \
    X(8, 64, 64, 8, 2, 2, 1)        \
    X(0, 64, 128, 3, 2, 4, 1)       \
    X(1, 64, 128, 3, 2, 2, 1)       \
    X(1, 128, 64, 3, 2, 2, 1)       \
    X(1, 128, 128, 3, 2, 2, 1)      \
    X(2, 128, 128, 2, 2, 2, 1)      \
    X(3, 128, 128, 4, 2, 2, 1)      \
    X(4, 128, 128, 4, 4, 2, 1)      \
    X(5, 128, 128, 3, 4, 2, 1)      \
    X(0, 256, 128, 3, 4, 2, 1)      \
    X(1, 256, 128, 2, 4, 2, 1)      \
    X(0, 128, 256, 3, 2, 4, 1)      \
    X(1, 128, 256, 2, 2, 4, 1)   \
    X(10, 64, 64, 2, 2, 2, 2)    \
    X(11, 64, 64, 4, 2, 2, 2)    \
    X(9, 128, 128, 2, 4, 2, 2)   \
    X(9, 144, 64, 3, 3, 2, 2)    \
    X(10, 144, 64, 2, 3, 2, 2)   \
    X(9, 144, 128, 2, 3, 2, 2)   \
    X(1, 144, 64, 2, 3, 2, 1)    \
    X(2, 144, 64, 4, 3, 2, 1)    \
    X(14, 144, 256, 2, 3, 4, 1)
#include "../../sky_embeddings_amd/csrc/gemm_pipe.hip"

#include <stdarg.h>
#include <algorithm>
#include <string>
#include <vector>
#include <tuple>

static char g_err[512];
int skyemb_skip_mask(void) { return 0; }
void skyemb_count_gemm(int) {}
void skyemb_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

#define CK(x)                                                                              \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) {                                                            \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            exit(1);                                                                       \
        }                                                                                  \
    } while (0)

__global__ void fill_bf16(bf16_t *p, size_t n, unsigned seed, float scale) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u + seed;
        x ^= x >> 15; x *= 2246822519u; x ^= x >> 13; x *= 3266489917u; x ^= x >> 16;
        p[i] = (bf16_t)(((int)(x & 0xffff) - 32768) * (scale / 32768.f));
    }
}
// naive reference: v[m,n] = sum_k A(m,k) B(n,k) in fp32 (+ bias + resid), gelu optional
__global__ void ref_gemm(const bf16_t *A, const bf16_t *B, int M, int N, int K, int64_t lda, int64_t ldb, int a_kc, int b_kc,
                         const float *bias, const float *resid, float *out) {
    const int n = blockIdx.x * 16 + (threadIdx.x & 15), m = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (m >= M || n >= N) return;
    float acc = 0.f;
    for (int k = 0; k < K; ++k) {
        const float a = (float)(a_kc ? A[(int64_t)m * lda + k] : A[(int64_t)k * lda + m]);
        const float b = (float)(b_kc ? B[(int64_t)n * ldb + k] : B[(int64_t)k * ldb + n]);
        acc = fmaf(a, b, acc);
    }
    if (bias) acc += bias[n];
    if (resid) acc += resid[(int64_t)m * N + n];
    out[(int64_t)m * N + n] = acc;
}

// one wave: s_memrealtime (100 MHz, chip-wide) when it starts and when it ends -- brackets the stamped launch on the stream
__global__ void marker_kernel(unsigned long long *o) {
    const unsigned long long a = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) {
        o[0] = a;
        o[1] = __builtin_amdgcn_s_memrealtime();
    }
}

__global__ void empty_kernel(float *p) {
    extern __shared__ char sm[];
    if (p == nullptr) sm[threadIdx.x] = 1;
}

struct Shape {
    const char *name;
    int M, N, K, a_kc, b_kc;   // GEMM view: out[M,N], contraction K
    int count;                 // launches per step
    int epi;                   // 0 bias->bf16, 1 bias+resid->f32, 2 bias+gelu (2 bf16 outs), 3 dgelu, 4 wgrad (f32 + colsum)
};

int main(int argc, char **argv) {
    const int B = 256, Me = B * 5, Md = B * 17;
    std::vector<Shape> shapes;
    const bool vitl = getenv("LAB_VITL") != nullptr;      // mim_19: SimMIM ViT-L/16 on 128x128 cutouts, 128 x 65 token rows
    for (int dec = 0; dec < (vitl ? 1 : 2); ++dec) {
        const int M = vitl ? 128 * 65 : dec ? Md : Me, D = vitl ? 1024 : dec ? 512 : 768, cnt = vitl ? 24 : dec ? 8 : 12;
        const char *t = dec ? "dec" : "enc";
        static char names[64][32];
        static int ni = 0;
        auto nm = [&](const char *l, const char *k) { snprintf(names[ni], 32, "%s.%s.%s", t, l, k); return names[ni++]; };
        shapes.push_back({nm("qkv", "fwd"), M, 3 * D, D, 1, 1, cnt, 0});
        shapes.push_back({nm("proj", "fwd"), M, D, D, 1, 1, cnt, 1});
        shapes.push_back({nm("fc1", "fwd"), M, 4 * D, D, 1, 1, cnt, 2});
        shapes.push_back({nm("fc2", "fwd"), M, D, 4 * D, 1, 1, cnt, 1});
        shapes.push_back({nm("qkv", "dgrad"), M, D, 3 * D, 1, 0, cnt, 0});
        shapes.push_back({nm("proj", "dgrad"), M, D, D, 1, 0, cnt, 0});
        shapes.push_back({nm("fc1", "dgrad"), M, D, 4 * D, 1, 0, cnt, 0});
        shapes.push_back({nm("fc2", "dgrad"), M, 4 * D, D, 1, 0, cnt, 3});
        shapes.push_back({nm("qkv", "wgrad"), 3 * D, D, M, 0, 0, cnt, 4});
        shapes.push_back({nm("proj", "wgrad"), D, D, M, 0, 0, cnt, 4});
        shapes.push_back({nm("fc1", "wgrad"), 4 * D, D, M, 0, 0, cnt, 4});
        shapes.push_back({nm("fc2", "wgrad"), D, 4 * D, M, 0, 0, cnt, 4});
    }
    std::vector<int> codes;
    for (int i = 1; i < argc; ++i) codes.push_back(atoi(argv[i]));
    if (codes.empty()) {
#define X(V, BM_, BN_, NS, WM_, WN_, WK_) codes.push_back(V * 1000000 + BM_ * 1000 + BN_);
        SKY_GEMM_VARIANTS(X)
#undef X
    }
    const int ROT = 6;
    size_t max_a = (size_t)Md * 3072, max_b = (size_t)3072 * 4352, max_o = (size_t)Md * 2048;
    if (vitl && getenv("LAB_WSWEEP")) { max_a = max_b = (size_t)4096 * 8320; max_o = (size_t)4096 * 1024; }
    for (const Shape &s : shapes) {
        max_a = std::max(max_a, (size_t)s.M * s.K);
        max_b = std::max(max_b, (size_t)s.N * s.K);
        max_o = std::max(max_o, (size_t)s.M * s.N);
    }
    bf16_t *A[ROT], *Bm[ROT], *aux;
    float *o32, *ref, *bias, *resid, *colsum, *ws;
    bf16_t *o16, *o16b;
    for (int r = 0; r < ROT; ++r) {
        CK(hipMalloc(&A[r], max_a * 2));
        CK(hipMalloc(&Bm[r], max_b * 2));
        fill_bf16<<<1024, 256>>>(A[r], max_a, 17 + r, 1.0f);
        fill_bf16<<<1024, 256>>>(Bm[r], max_b, 91 + r, 0.05f);
    }
    CK(hipMalloc(&aux, max_o * 2));
    fill_bf16<<<1024, 256>>>(aux, max_o, 5, 1.0f);
    CK(hipMalloc(&o32, max_o * 4)); CK(hipMalloc(&ref, max_o * 4)); CK(hipMalloc(&resid, max_o * 4));
    CK(hipMalloc(&o16, max_o * 2)); CK(hipMalloc(&o16b, max_o * 2));
    CK(hipMalloc(&bias, 8192 * 4)); CK(hipMalloc(&colsum, 8192 * 4));
    const size_t ws_bytes = 256u << 20;
    CK(hipMalloc(&ws, ws_bytes));
    CK(hipMemset(bias, 0, 8192 * 4)); CK(hipMemset(resid, 0, max_o * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> h_out(max_o), h_ref(max_o);

    auto make_args = [&](const Shape &s, int r, int code, int split) {
        skyemb_gemm_args g;
        memset(&g, 0, sizeof g);
        g.A = A[r]; g.B = Bm[r];
        g.a_layout = s.a_kc ? SKYEMB_KC : SKYEMB_RC; g.b_layout = s.b_kc ? SKYEMB_KC : SKYEMB_RC;
        g.lda = s.a_kc ? s.K : s.M; g.ldb = s.b_kc ? s.K : s.N;
        g.M = s.M; g.N = s.N; g.K = s.K; g.dtype = SKYEMB_BF16; g.alpha = 1.f;
        g.tile = code; g.split_k = split; g.ws = ws; g.ws_bytes = (int64_t)ws_bytes;
        switch (s.epi) {
            case 0: g.bias = bias; g.out = o16; g.ldo = s.N; break;
            case 1: g.bias = bias; g.resid = resid; g.ldr = s.N; g.out_f32 = o32; g.ldo32 = s.N; break;
            case 2: g.bias = bias; g.act = SKYEMB_ACT_GELU; g.out = o16; g.ldo = s.N; g.out2 = o16b; g.ldo2 = s.N; break;
            case 3: g.act = SKYEMB_ACT_DGELU; g.aux = aux; g.ldaux = s.N; g.out = o16; g.ldo = s.N; break;
            default: g.out_f32 = o32; g.ldo32 = s.N; g.colsum_a = getenv("LAB_NOCOLSUM") ? nullptr : colsum; break;
        }
        return g;
    };

    // ---- correctness of every variant on two shapes per layout class (plain f32 output, no activation)
    int bad = 0;
    for (int code : codes) {
        for (const Shape &s0 : shapes) {
            if (strcmp(s0.name, "enc.proj.fwd") && strcmp(s0.name, "enc.qkv.dgrad") && strcmp(s0.name, "enc.proj.wgrad") &&
                strcmp(s0.name, "enc.fc2.dgrad"))
                continue;
            Shape s = s0;
            s.epi = 1;
            if (code / 1000000 >= 9 && !s.a_kc) continue;                                       // two-k-group tiles: k-contiguous A only
            if (code % 1000000 == 256256 && !s.a_kc && (s.b_kc || s.M % 256 || s.N % 256)) continue;   // 256x256: k-contiguous A, or a weight gradient of whole tiles
            if ((code % 1000000) / 1000 == 144 && !s.a_kc) continue;                           // 144-row image (136-row stride): k-contiguous A only
            if (code % 1000 == 192 && !s.b_kc) continue;                                       // 192 columns: k-contiguous B only
            skyemb_gemm_args g = make_args(s, 0, code, 1);
            g.resid = nullptr; g.bias = nullptr;
            CK(hipMemset(o32, 0xff, (size_t)s.M * s.N * 4));
            if (skyemb_gemm_pipe_try(g, 0) != 0) { printf("code %d shape %s: launch refused: %s\n", code, s.name, g_err); ++bad; continue; }
            ref_gemm<<<dim3((s.N + 15) / 16, (s.M + 15) / 16), 256>>>(A[0], Bm[0], s.M, s.N, s.K, g.lda, g.ldb, s.a_kc, s.b_kc, nullptr,
                                                                       nullptr, ref);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(h_out.data(), o32, (size_t)s.M * s.N * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(h_ref.data(), ref, (size_t)s.M * s.N * 4, hipMemcpyDeviceToHost));
            double maxe = 0, maxr = 0;
            for (size_t i = 0; i < (size_t)s.M * s.N; ++i) {
                maxe = std::max(maxe, (double)fabsf(h_out[i] - h_ref[i]));
                maxr = std::max(maxr, (double)fabsf(h_ref[i]));
            }
            if (!(maxe <= 2e-3 * maxr)) { printf("code %d shape %s: MISMATCH max err %g (max |ref| %g)\n", code, s.name, maxe, maxr); ++bad; }
        }
    }
    if (getenv("LAB_NOCHECK")) bad = 0;
    printf("correctness: %s\n", bad ? "FAILED" : "all variants match the naive reference");
    if (bad) return 1;

    // ---- timing
    auto time_us = [&](const Shape &s, int code, int split) -> float {
        const int iters = 30;
        for (int i = 0; i < ROT; ++i) {
            skyemb_gemm_args g = make_args(s, i % ROT, code, split);
            if (skyemb_gemm_pipe_try(g, 0) != 0) return -1.f;
        }
        CK(hipEventRecord(e0));
        for (int i = 0; i < iters; ++i) {
            skyemb_gemm_args g = make_args(s, i % ROT, code, split);
            skyemb_gemm_pipe_try(g, 0);
        }
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        return ms * 1e3f / iters;
    };

#ifdef GEMM_STAMP
    if (getenv("LAB_STAMP256")) {
        // the 256 x 256 kernel's phases: cycles per phase and wave in its four parts (READ part, first barrier, MFMA part, second
        // barrier), waves 0-3 = the early group, 4-7 the late one; workgroups 0-7
        unsigned long long *d_st;
        CK(hipMalloc(&d_st, 64 * 8 * 8));
        CK(hipMemcpyToSymbol(HIP_SYMBOL(g_gemm_stamp), &d_st, sizeof(d_st)));
        for (int bkc = 1; bkc >= 0; --bkc) {
            Shape s{"stamp256", 8192, 4096, 1024, 1, bkc, 1, 0};
            CK(hipMemset(d_st, 0, 64 * 8 * 8));
            for (int it = 0; it < 4; ++it) {
                skyemb_gemm_args g = make_args(s, it % ROT, 256256, 1);
                if (skyemb_gemm_pipe_try(g, 0) != 0) { fprintf(stderr, "refused: %s\n", g_err); return 1; }
            }
            CK(hipDeviceSynchronize());
            std::vector<unsigned long long> h(64 * 8);
            CK(hipMemcpy(h.data(), d_st, 64 * 8 * 8, hipMemcpyDeviceToHost));
            for (int w = 0; w < 8; ++w) {
                double a[4] = {0, 0, 0, 0}, n = 0;
                for (int b = 0; b < 8; ++b) {
                    for (int i = 0; i < 4; ++i) a[i] += (double)h[(b * 8 + w) * 8 + i];
                    n += (double)h[(b * 8 + w) * 8 + 4];
                }
                double pro = 0, epi = 0, nt = 0;
                for (int b = 0; b < 8; ++b) { pro += (double)h[(b * 8 + w) * 8 + 5]; epi += (double)h[(b * 8 + w) * 8 + 6]; nt += (double)h[(b * 8 + w) * 8 + 7]; }
                printf("B %s wave %d: per phase  read %.0f  barrier1 %.0f  mfma %.0f  barrier2 %.0f  = %.0f cycles | per tile: prologue %.0f  k-loop %.0f  epilogue %.0f cycles\n", bkc ? "kc" : "rc", w,
                       a[0] / n, a[1] / n, a[2] / n, a[3] / n, (a[0] + a[1] + a[2] + a[3]) / n, pro / nt, (a[0] + a[1] + a[2] + a[3]) / nt, epi / nt);
            }
        }
        return 0;
    }
    if (getenv("LAB_STAMP")) {
        // timeline of the launches that dominate the step (product tile choice: code 0 = tuned table / heuristic), cold
        // operands (rotation over six sets, as in the step)
        const int MAXWG = 4096;
        unsigned long long *d_st, *d_mk;
        CK(hipMalloc(&d_st, (size_t)MAXWG * 16 * 8));
        CK(hipMalloc(&d_mk, 4 * 8));
        CK(hipMemcpyToSymbol(HIP_SYMBOL(g_gemm_stamp), &d_st, sizeof(d_st)));
        std::vector<unsigned long long> h((size_t)MAXWG * 16), hm(4);
        struct Want { const char *what; int M, N, K, a_kc, b_kc, epi; };
        const Want wants[] = {{"enc.proj.fwd [1280 x 768 x 768] bias + fp32 residual", 1280, 768, 768, 1, 1, 1},
                              {"enc.qkv.fwd [1280 x 2304 x 768] bias -> bf16", 1280, 2304, 768, 1, 1, 0},
                              {"dec.fc1.fwd [4352 x 2048 x 512] bias + GELU (two bf16 outputs)", 4352, 2048, 512, 1, 1, 2},
                              {"enc.fc1.dgrad [1280 x 768 x 3072] data gradient (row-contiguous weights)", 1280, 768, 3072, 1, 0, 0},
                              {"enc.fc2.dgrad [1280 x 3072 x 768] data gradient + dGELU", 1280, 3072, 768, 1, 0, 3},
                              {"dec.qkv.fwd [4352 x 1536 x 512] bias -> bf16", 4352, 1536, 512, 1, 1, 0}};
        printf("{\n \"source\": \"tools/ubench/gemm_lab built with -DGEMM_STAMP -DSKY_GEMM_LAB, LAB_STAMP=1: marker kernel -> launch -> marker kernel on one stream, 24 repetitions over 6 operand sets; medians over repetitions of per-launch figures, medians over workgroups of per-phase figures; times in us (s_memrealtime 100 MHz for spans and gaps, s_memtime shader cycles / measured clock for phases)\",\n \"launches\": [\n");
        bool first_out = true;
        for (const Want &wn : wants) {
            Shape s{"stamp", wn.M, wn.N, wn.K, wn.a_kc, wn.b_kc, 1, wn.epi};
            std::vector<double> span, gap_in, gap_out, spread, clk, ph[6], evt, xcd_skew;
            int grid = 0, nsplit = 1;
            for (int it = 0; it < 26; ++it) {
                skyemb_gemm_args g = make_args(s, it % ROT, 0, 0);
                CK(hipMemsetAsync(d_st, 0, (size_t)MAXWG * 16 * 8, 0));
                marker_kernel<<<1, 64>>>(d_mk);
                CK(hipEventRecord(e0));
                if (skyemb_gemm_pipe_try(g, 0) != 0) { fprintf(stderr, "refused: %s\n", g_err); return 1; }
                CK(hipEventRecord(e1));
                marker_kernel<<<1, 64>>>(d_mk + 2);
                CK(hipDeviceSynchronize());
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                CK(hipMemcpy(h.data(), d_st, (size_t)MAXWG * 16 * 8, hipMemcpyDeviceToHost));
                CK(hipMemcpy(hm.data(), d_mk, 32, hipMemcpyDeviceToHost));
                if (it < 2) continue;                         // warm-up
                int n = 0;
                while (n < MAXWG && h[(size_t)n * 16 + 8]) ++n;
                grid = n;
                unsigned long long lo = ~0ull, hi = 0, lo_max = 0;
                std::vector<double> p[6], c;
                for (int w = 0; w < n; ++w) {
                    const unsigned long long *o = &h[(size_t)w * 16];
                    lo = std::min(lo, o[8]); lo_max = std::max(lo_max, o[8]); hi = std::max(hi, o[9]);
                    const double cyc_per_us = (double)(o[6] - o[0]) / ((double)(o[9] - o[8]) / 100.0);
                    c.push_back(cyc_per_us);
                    for (int k = 0; k < 6; ++k) p[k].push_back((double)(o[k + 1] - o[k]));
                }
                auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
                const double cm = med(c);
                for (int k = 0; k < 6; ++k) ph[k].push_back(med(p[k]) / cm);
                clk.push_back(cm / 1e3);
                span.push_back((double)(hi - lo) / 100.0);
                gap_in.push_back((double)((long long)lo - (long long)hm[1]) / 100.0);
                gap_out.push_back((double)((long long)hm[2] - (long long)hi) / 100.0);
                spread.push_back((double)(lo_max - lo) / 100.0);
                evt.push_back(ms * 1e3);
            }
            auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
            printf("%s  {\"launch\": \"%s\", \"workgroups\": %d, \"hip_event_us\": %.2f, \"marker_to_first_workgroup_us\": %.2f, "
                   "\"first_to_last_workgroup_start_us\": %.2f, \"first_start_to_last_end_us\": %.2f, \"last_end_to_next_kernel_us\": %.2f, "
                   "\"clock_ghz\": %.3f, \"workgroup_phases_us\": {\"entry_to_ring_issued\": %.2f, \"first_stage_landed\": %.2f, "
                   "\"k_loop\": %.2f, \"tile_to_lds\": %.2f, \"epilogue_loads_math_stores_issued\": %.2f, \"stores_acknowledged\": %.2f}}",
                   first_out ? "" : ",\n", wn.what, grid, med(evt), med(gap_in), med(spread), med(span), med(gap_out), med(clk), med(ph[0]), med(ph[1]),
                   med(ph[2]), med(ph[3]), med(ph[4]), med(ph[5]));
            first_out = false;
            (void)nsplit;
        }
        printf("\n ]\n}\n");
        return 0;
    }
#endif
    if (getenv("LAB_EMPTY")) {
        for (auto cfg : {std::tuple<int, int, int>{960, 256, 49152}, {2176, 256, 49152}, {480, 512, 73728}, {240, 512, 98304}, {544, 512, 98304}}) {
            const int grid = std::get<0>(cfg), thr = std::get<1>(cfg), lds = std::get<2>(cfg);
            hipFuncSetAttribute((const void *)empty_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            for (int i = 0; i < 5; ++i) empty_kernel<<<grid, thr, lds>>>(o32);
            CK(hipEventRecord(e0));
            for (int i = 0; i < 50; ++i) empty_kernel<<<grid, thr, lds>>>(o32);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            printf("empty kernel grid %d x %d threads, %d B LDS: %.2f us per launch\n", grid, thr, lds, ms * 1e3f / 50);
        }
    }
    if (getenv("LAB_GROUPS")) {
        // grouped launches of one transformer block's backward: the four dW alone (round-1 schedule), and each dW with
        // the dX that shares its dY, per tile shape
        char *blob_dev;
        CK(hipMalloc(&blob_dev, 65536));
        const size_t slot = (size_t)3072 * 768;
        float *o32g;
        CK(hipMalloc(&o32g, 4 * slot * 4));
        std::vector<char> blob_host(65536);
        auto group_us = [&](std::vector<skyemb_gemm_args> probs, int tile) -> float {
            skyemb_gemm_group_info info;
            if (skyemb_gemm_group_plan(probs.data(), (int)probs.size(), tile, blob_host.data(), 65536, &info) != 0) return -1.f;
            CK(hipMemcpy(blob_dev, blob_host.data(), 65536, hipMemcpyHostToDevice));
            for (int i = 0; i < 3; ++i) if (skyemb_gemm_group_launch(blob_dev, &info, 0) != 0) return -1.f;
            CK(hipEventRecord(e0));
            for (int i = 0; i < 20; ++i) skyemb_gemm_group_launch(blob_dev, &info, 0);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            return ms * 1e3f / 20;
        };
        for (int dec = 0; dec < 2; ++dec) {
            const int base = dec * 12;
            auto prob = [&](int idx, int r) {
                skyemb_gemm_args g = make_args(shapes[base + idx], r, 0, 1);
                // distinct outputs per problem so that nothing aliases
                if (g.out_f32) g.out_f32 = o32g + (size_t)(r % 4) * slot;   // every weight gradient of a block fits one slot
                return g;
            };
            // indices: 4 qkv.dgrad 5 proj.dgrad 6 fc1.dgrad 7 fc2.dgrad 8 qkv.wgrad 9 proj.wgrad 10 fc1.wgrad 11 fc2.wgrad
            for (int tile : {64064, 128064, 128128}) {
                const float w4 = group_us({prob(8, 0), prob(9, 1), prob(10, 2), prob(11, 3)}, tile);
                float singles = 0, pairs = 0;
                for (int l = 0; l < 4; ++l) {
                    singles += time_us(shapes[base + 4 + l], 0, 0);
                    pairs += group_us({prob(4 + l, 0), prob(8 + l, 1)}, tile);
                }
                // shifted pairing: dX(fc2) alone, dX(fc1)+dW(fc2), dX(proj)+dW(fc1), dX(qkv)+dW(proj), dW(qkv) rides with the next block's dX(fc2)
                const float shifted = group_us({prob(7, 0), prob(8, 1)}, tile) + group_us({prob(6, 0), prob(11, 1)}, tile) +
                                      group_us({prob(5, 0), prob(10, 1)}, tile) + group_us({prob(4, 0), prob(9, 1)}, tile);
                printf("%s block bwd, tile %6d: 4 dW grouped %.1f us + 4 dX single (product) %.1f = %.1f | 4 x (dX+dW) %.1f | shifted pairs %.1f\n",
                       dec ? "dec" : "enc", tile, w4, singles, w4 + singles, pairs, shifted);
            }
        }
        return 0;
    }
    if (getenv("LAB_WSWEEP")) {
        // weight-gradient launches (both operands row-contiguous, contraction over token rows): time vs the token count
        for (int code : codes) {
            if (code / 1000000 >= 9) continue;
            int bm, bn;
            tile_dims(code, bm, bn);
            std::vector<std::pair<int, int>> wmn = {{2048, 512}, {512, 2048}, {1536, 512}, {3072, 768}, {768, 3072}, {2304, 768}};
            std::vector<int> wk = {256, 1088, 2176, 4352};
            if (vitl) {       // mim_19: the four weight gradients of a ViT-L block over 8320 token rows
                wmn = {{3072, 1024}, {1024, 1024}, {4096, 1024}, {1024, 4096}};
                wk = {1024, 2176, 4352, 8320};
            }
            for (auto mn : wmn) {
                if (code % 1000000 == 256256 && (mn.first % 256 || mn.second % 256)) continue;
                const int64_t tiles = ceil_div64(mn.first, bm) * ceil_div64(mn.second, bn);
                printf("wsweep code %7d M %d N %d (%4ld tiles):", code, mn.first, mn.second, (long)tiles);
                float t1 = 0, t2 = 0;
                for (int K : wk) {
                    if (vitl && K == 4352) { Shape s{"w", mn.first, mn.second, K, 0, 0, 1, 4}; t1 = time_us(s, code, 1); printf("  K%d:%.1f", K, t1); continue; }
                    if (vitl && K == 8320) { Shape s{"w", mn.first, mn.second, K, 0, 0, 1, 4}; t2 = time_us(s, code, 1); printf("  K%d:%.1f", K, t2);
                        printf("  | %.2f us/k-step, %.0f TF/s at K = 8320\n", (t2 - t1) / 62.0, 2.0 * mn.first * mn.second * 8320 / t2 / 1e6); continue; }
                    Shape s{"w", mn.first, mn.second, K, 0, 0, 1, 4};
                    if ((size_t)mn.first * K > max_a || (size_t)mn.second * K > max_b || (size_t)mn.first * mn.second > max_o) continue;
                    const float t = time_us(s, code, 1);
                    printf("  K%d:%.1f", K, t);
                    if (K == 2176) t1 = t;
                    if (K == 4352) t2 = t;
                }
                if (vitl) continue;
                const double per_step = (t2 - t1) / 34.0;
                printf("  | %.2f us/k-step, %.0f TF/s at K = 4352\n", per_step, 2.0 * mn.first * mn.second * 4352 / t2 / 1e6);
            }
        }
        return 0;
    }
    if (getenv("LAB_KSWEEP")) {
        // time vs K at fixed M, N: the intercept is the launch + prologue + epilogue cost, the slope the k-loop
        const int epi = atoi(getenv("LAB_KSWEEP"));
        const int bkc = getenv("LAB_BKC") ? atoi(getenv("LAB_BKC")) : 1;      // 0: B operand row-contiguous (the dgrad layout)
        for (int code : codes) {
            int bm, bn;
            tile_dims(code, bm, bn);
            std::vector<std::pair<int, int>> mns = {{1280, 3072}, {1280, 768}, {4352, 2048}, {4352, 512}};
            if (vitl) mns = {{8320, 4096}, {8320, 3072}, {8320, 1024}, {8192, 4096}, {8192, 1024}};
            for (auto mn : mns) {
                const int64_t tiles = ceil_div64(mn.first, bm) * ceil_div64(mn.second, bn);
                printf("ksweep code %7d epi %d M %d N %d (%4ld tiles):", code, epi, mn.first, mn.second, (long)tiles);
                float t768 = 0, t1536 = 0;
                for (int K : {64, 256, 768, 1536}) {
                    if ((size_t)mn.first * K > max_a || (size_t)mn.second * K > max_b) continue;
                    Shape s{"k", mn.first, mn.second, K, 1, bkc, 1, epi};
                    const float t = time_us(s, code, 1);
                    printf("  K%d:%.1f", K, t);
                    if (K == 768) t768 = t;
                    if (K == 1536) t1536 = t;
                }
                const double per_step = (t1536 - t768) / 12.0;
                printf("  | %.2f us/k-step = %.1f TB/s L2->LDS\n", per_step, tiles * (bm + bn) * 128.0 / per_step / 1e6);
            }
        }
        return 0;
    }
    double tot_base = 0, tot_best = 0;
    const char *only = getenv("LAB_ONLY");              // e.g. LAB_ONLY=dec. : the decoder's shapes only
    for (const Shape &s : shapes) {
        if (only && !strstr(s.name, only)) continue;
        struct R { int code, split; float us; };
        std::vector<R> res;
        const float base = time_us(s, 0, 0);    // current product choice (tuned table / heuristic)
        for (int code : codes) {
            int bm, bn;
            tile_dims(code, bm, bn);
            const int64_t tiles = ceil_div64(s.M, bm) * ceil_div64(s.N, bn);
            for (int split : {1, 2, 3, 4, 6, 8}) {
                if (split > 1 && (tiles * split > 1100 || s.K / 64 / split < 4)) continue;
                const float us = time_us(s, code, split);
                if (us > 0) res.push_back({code, split, us});
            }
        }
        std::sort(res.begin(), res.end(), [](const R &a, const R &b) { return a.us < b.us; });
        const double flop = 2.0 * s.M * s.N * s.K;
        printf("%-16s M%5d N%5d K%5d  product %6.1f us (%4.0f TF/s) |", s.name, s.M, s.N, s.K, base, flop / base / 1e6);
        for (size_t i = 0; i < res.size() && i < 6; ++i) printf("  %d/s%d:%.1f", res[i].code, res[i].split, res[i].us);
        printf("\n");
        fflush(stdout);
        tot_base += base * s.count;
        tot_best += std::min(base, res.empty() ? base : res[0].us) * s.count;
    }
    printf("per step (block layers only): product %.3f ms, best-of %.3f ms\n", tot_base / 1e3, tot_best / 1e3);
    return 0;
}
