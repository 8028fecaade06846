// Sorted top-k list maintenance shared by the cosine top-k kernels.
//
// A list of capacity k (k <= 512) lives in LDS as lsq[k] (scores, descending) + liq[k] (row ids);
// ONE wavefront owns it.  Candidates arrive in ascending row order, so among equal scores the
// earlier (lower index) entry stays in front: order = (score desc, index asc).
#pragma once
#include "common.h"

// Insert (cv, cidx) into the wave-owned sorted list; returns the new size.  Caller guarantees
// cv > (n_in == k ? lsq[k-1] : -inf).  ~2 LDS reads + 2 writes per lane-slot, no loops over k.
__device__ __forceinline__ int topk_list_insert(float *lsq, int *liq, int n_in, int k, float cv, int cidx, int lane) {
    const int new_n = n_in < k ? n_in + 1 : k;
    // entries >= cv form a prefix; pos = its length
    int pos = 0;
    for (int e0 = 0; e0 < n_in; e0 += 64) {
        const int e = e0 + lane;
        const bool ge = e < n_in && lsq[e] >= cv;
        pos += __builtin_popcountll(__ballot(ge));
    }
    // shift [pos, new_n-1) down by one, highest 64-slot first so that reads precede overwrites
    for (int e0 = ((new_n - 1) >> 6) << 6; e0 >= 0; e0 -= 64) {
        const int e = e0 + lane;
        if (e0 + 64 <= pos) break;                       // nothing at or after pos in this and lower slots
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
        liq[pos] = cidx;
    }
    __builtin_amdgcn_wave_barrier();
    return new_n;
}
