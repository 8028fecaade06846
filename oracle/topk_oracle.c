/*
 * CPU oracle for the build-level batched weighted-cosine top-k contract.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): used by tests/, by
 * __graft_entry__.smoke() and by bench.py's cpu_baseline leg as the checker /
 * timed baseline; never linked or called by the product path.
 *
 * Restates utils/similarity.py:149-172 (weighted_cosine_similarity) for a
 * batch of Q target vectors against an N-row bank, followed by the
 * "keep the best n_save" of utils/similarity.py:18-35 expressed as an exact
 * top-k with a deterministic order.  The reference leaves the fp32 summation
 * order to torch; this oracle FIXES it so that the HIP path can be bit-exact:
 *
 *   tw[d]  = w[d] * t[q][d]                       (one fp32 rounding)
 *   dot    = fma-chain over d = 0..D-1, acc0 = 0: acc = fmaf(tw[d], x[n][d], acc)
 *   qn2    = fma-chain: acc = fmaf(tw[d], t[q][d], acc)
 *   xn2    = fma-chain: acc = fmaf(w[d]*x[n][d], x[n][d], acc)
 *   score  = dot / fmaf(sqrtf(qn2), sqrtf(xn2), eps)   (fused multiply-add, IEEE sqrt and divide)
 *   NaN scores compare as -inf.
 *   order  = score descending, then index ascending (reference argsort is not
 *            stable: utils/similarity.py:24).
 *
 * The chain order d = 0,1,2,... is exactly what v_mfma_f32_16x16x4_f32 does
 * when consecutive MFMAs take consecutive 4-wide k groups (each MFMA is a
 * k-ordered fmaf chain on its C input).
 *
 * Build: make -C oracle   (gcc -O2 -mfma -ffp-contract=off -fopenmp)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ROWS 8 /* independent chains in flight to hide fma latency */

int skyemb_oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

static void set_threads(int threads) {
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#else
    (void)threads;
#endif
}

static inline float chain_dot(const float *a, const float *b, int64_t D) {
    float acc = 0.0f;
    for (int64_t d = 0; d < D; ++d) acc = fmaf(a[d], b[d], acc);
    return acc;
}

static inline float chain_wnorm2(const float *w, const float *x, int64_t D) {
    float acc = 0.0f;
    for (int64_t d = 0; d < D; ++d) {
        float xw = w[d] * x[d];
        acc = fmaf(xw, x[d], acc);
    }
    return acc;
}

static inline float finish(float dot, float qn, float xn, float eps) {
    float den = fmaf(qn, xn, eps); /* one rounding: what the GPU compiler emits for qn*xn+eps anyway */
    float s = dot / den;
    if (!(s == s)) s = -INFINITY;
    return s;
}

/* scores for query block [q0,q1) x all N rows into out[(q-q0)*N + n] */
static void score_block(const float *tw, const float *qn, const float *X, const float *w, int64_t q0, int64_t q1,
                        int64_t N, int64_t D, float eps, float *out) {
#pragma omp parallel for schedule(static)
    for (int64_t nb = 0; nb < N; nb += ROWS) {
        int64_t nr = (N - nb) < ROWS ? (N - nb) : ROWS;
        float xn[ROWS];
        for (int64_t r = 0; r < nr; ++r) xn[r] = sqrtf(chain_wnorm2(w, X + (nb + r) * D, D));
        for (int64_t q = q0; q < q1; ++q) {
            const float *a = tw + q * D;
            float acc[ROWS];
            for (int r = 0; r < ROWS; ++r) acc[r] = 0.0f;
            if (nr == ROWS) {
                const float *x0 = X + nb * D;
                for (int64_t d = 0; d < D; ++d) {
                    float av = a[d];
                    for (int r = 0; r < ROWS; ++r) acc[r] = fmaf(av, x0[r * D + d], acc[r]);
                }
            } else {
                for (int64_t r = 0; r < nr; ++r) acc[r] = chain_dot(a, X + (nb + r) * D, D);
            }
            for (int64_t r = 0; r < nr; ++r) out[(q - q0) * N + nb + r] = finish(acc[r], qn[q], xn[r], eps);
        }
    }
}

static void prep_queries(const float *T, const float *w, int64_t Q, int64_t D, float *tw, float *qn) {
    for (int64_t q = 0; q < Q; ++q) {
        for (int64_t d = 0; d < D; ++d) tw[q * D + d] = w[d] * T[q * D + d];
        qn[q] = sqrtf(chain_dot(tw + q * D, T + q * D, D));
    }
}

int skyemb_oracle_cosine_scores(const float *T, const float *X, const float *w, int64_t Q, int64_t N, int64_t D,
                                float eps, float *scores, int threads) {
    set_threads(threads);
    float *tw = (float *)malloc(sizeof(float) * Q * D), *qn = (float *)malloc(sizeof(float) * Q);
    if (!tw || !qn) return 1;
    prep_queries(T, w, Q, D, tw, qn);
    score_block(tw, qn, X, w, 0, Q, N, D, eps, scores);
    free(tw);
    free(qn);
    return 0;
}

/* better(a,b): a precedes b in (score desc, idx asc) */
static inline int better(float sa, int64_t ia, float sb, int64_t ib) { return sa > sb || (sa == sb && ia < ib); }

static void select_topk(const float *s, int64_t N, int64_t k, float *os, int64_t *oi) {
    int64_t cnt = 0;
    for (int64_t n = 0; n < N; ++n) {
        float v = s[n];
        if (cnt == k && !better(v, n, os[k - 1], oi[k - 1])) continue;
        int64_t pos = cnt < k ? cnt : k - 1;
        while (pos > 0 && better(v, n, os[pos - 1], oi[pos - 1])) {
            os[pos] = os[pos - 1];
            oi[pos] = oi[pos - 1];
            --pos;
        }
        os[pos] = v;
        oi[pos] = n;
        if (cnt < k) ++cnt;
    }
    for (int64_t j = cnt; j < k; ++j) {
        os[j] = -INFINITY;
        oi[j] = -1;
    }
}

int skyemb_oracle_cosine_topk(const float *T, const float *X, const float *w, int64_t Q, int64_t N, int64_t D,
                              int64_t k, float eps, float *out_s, int64_t *out_i, int threads) {
    set_threads(threads);
    const int64_t QB = 32;
    float *tw = (float *)malloc(sizeof(float) * Q * D), *qn = (float *)malloc(sizeof(float) * Q);
    float *buf = (float *)malloc(sizeof(float) * QB * N);
    if (!tw || !qn || !buf) return 1;
    prep_queries(T, w, Q, D, tw, qn);
    for (int64_t q0 = 0; q0 < Q; q0 += QB) {
        int64_t q1 = q0 + QB < Q ? q0 + QB : Q;
        score_block(tw, qn, X, w, q0, q1, N, D, eps, buf);
#pragma omp parallel for schedule(dynamic, 1)
        for (int64_t q = q0; q < q1; ++q) select_topk(buf + (q - q0) * N, N, k, out_s + q * k, out_i + q * k);
    }
    free(tw);
    free(qn);
    free(buf);
    return 0;
}

/* utils/similarity.py:101-102: (x - mean_feats) / (std_feats + 1e-8) */
int skyemb_oracle_standardise(const float *X, const float *mu, const float *sigma, int64_t N, int64_t D, float *out,
                              int threads) {
    set_threads(threads);
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < N; ++n)
        for (int64_t d = 0; d < D; ++d) {
            float den = sigma[d] + 1e-8f;
            out[n * D + d] = (X[n * D + d] - mu[d]) / den;
        }
    return 0;
}

/* ---- diagnostics (used once to pin the MFMA accumulation order; kept for the parity tests) ----
 * variants of a D-long dot product, D % 4 == 0:
 *  0: sequential fma chain d = 0..D-1 (the contract)
 *  1: per group of 4, reversed order inside the group
 *  2: per group of 4: t = fma(a1,b1, a0*b0); u = fma(a3,b3, a2*b2); acc = acc + (t + u)
 *  3: per group of 4: exact (double) sum of the 4 products added to acc with one rounding
 *  4: unfused: acc = acc + (float)(a*b) sequential
 */
int skyemb_oracle_dot_variants(const float *a, const float *b, int64_t D, float *out) {
    float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f, v4 = 0.f;
    for (int64_t d = 0; d < D; ++d) v0 = fmaf(a[d], b[d], v0);
    for (int64_t g = 0; g < D; g += 4) {
        for (int j = 3; j >= 0; --j) v1 = fmaf(a[g + j], b[g + j], v1);
        float t = fmaf(a[g + 1], b[g + 1], a[g] * b[g]);
        float u = fmaf(a[g + 3], b[g + 3], a[g + 2] * b[g + 2]);
        v2 = v2 + (t + u);
        double s = (double)a[g] * b[g] + (double)a[g + 1] * b[g + 1] + (double)a[g + 2] * b[g + 2] +
                   (double)a[g + 3] * b[g + 3];
        v3 = (float)((double)v3 + s);
    }
    for (int64_t d = 0; d < D; ++d) {
        float p = a[d] * b[d];
        v4 = v4 + p;
    }
    out[0] = v0; out[1] = v1; out[2] = v2; out[3] = v3; out[4] = v4;
    return 0;
}

int skyemb_oracle_wnorms(const float *X, const float *w, int64_t N, int64_t D, float *out) {
    for (int64_t n = 0; n < N; ++n) out[n] = sqrtf(chain_wnorm2(w, X + n * D, D));
    return 0;
}
