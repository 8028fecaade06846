// 256 x 256 tile, eight waves, one workgroup per CU: the launches of ViT-L-sized layers whose output has at least ~2 such tiles
// per CU ([8320 x 4096 x 1024], [8320 x 3072 x 1024]: configs/mim_19.ini).  Included by gemm_pipe.hip inside its anonymous namespace
// (shares the LDS-DMA issue / fragment-read helpers and the epilogue arithmetic).
//
// Why another loop structure.  The ring kernels above keep one barrier per k-tile and let co-resident workgroups (or a second
// wave group over k) cover each other's waits; per byte brought into LDS a 128 x 128 tile does 64 MFMA-columns of work, a
// 256 x 256 tile 128: the ViT-L launches sat at 650-800 TFLOP/s on the L2 -> LDS fill rate.  Here a wave owns 128 x 64 of the
// tile (32 accumulator fragments = 128 registers), the k-tile (64 wide) is consumed in TWO PHASES of 32 MFMAs (one 64-row half of
// the wave's block each), and the two waves that share a SIMD run half a phase apart: while one multiplies, the other reads
// the fragments of its next phase and issues the next LDS-DMA pieces -- matrix pipe beside LDS / memory pipe on every SIMD,
// all the time (MI355X_MICROARCH.md, "Two waves per SIMD").
//
// LDS: a ring of ten 16 KiB HALF-TILES (128 rows x 64 k of A or of B) = all 160 KiB.  Half-tile h = 4 t + e of k-tile t, e in
// {0: B rows 0-127, 1: B rows 128-255, 2: A rows 0-127, 3: A rows 128-255}, lives in slot h mod 10.  Every wave issues two of
// a half-tile's sixteen 1 KiB pieces.  Phase p = 2 t + q requests half-tiles 2 p + 6 and 2 p + 7 in its READ part: six to seven
// half-tiles (1.5+ k-tiles) run ahead.
//
// Synchronisation (slots = the intervals between consecutive workgroup barriers; wave group G0 = waves 0-3 runs the READ part
// of phase p in slot 2p and its MFMA part in slot 2p + 1, G1 = waves 4-7 one slot later; a wave reads its B fragments and the
// first half of its A rows in phase 2t, the second half in phase 2t + 1):
//   RAW  phase 2t + 1 ends its READ part with a counted wait that leaves the two youngest half-tiles in flight (4t + 8, 4t + 9,
//        requested in that phase) and thereby retires every piece of k-tile t + 1 this wave issued; G0 has waited by slot
//        4t + 2, G1 by 4t + 3; the first read of k-tile t + 1 is G0's in slot 4t + 4 -- a barrier interval after the LAST wait.
//   WAR  slot of half-tile h = 4t + e is re-filled by half-tile h + 10, requested in phase (h + 4) / 2 (rounded down), by G0 in
//        slot 4t + 4 (e = 0, 1) or 4t + 6 (e = 2, 3).  The last reads of k-tile t: B halves in phase 2t, G1's issued in slot
//        4t + 1 and complete (lgkmcnt, at the head of its MFMA part) in slot 4t + 2; A rows 0-127 by G0 in phase 2t + 1, complete
//        in slot 4t + 3; A rows 128-255 by G1, complete in slot 4t + 4.  Every re-fill is behind a barrier that follows the
//        completion of the reads it overwrites (two or more barriers, one for A rows 128-255).
// Fragments are read by ds_read_b128 from the XOR-swizzled k-contiguous images (conflict-free, see issue_kc / frag_kc); a
// row-contiguous B (data gradients) uses the transposing reads of the ring kernels on two [64 k][128 n] images.
// Epilogue: the fp32 tile goes through LDS in two halves of 128 rows (133 KB each) and leaves in 16 / 32-byte row pieces with
// bias, fp32 residual, GELU (+ pre-activation) or dGELU fused, like the ring kernels'.

constexpr int G256_HT = 16384, G256_RING = 10, G256_AHEAD = 6;

// LDS-DMA of 16 bytes per lane from a WORKGROUP-UNIFORM base (scalar registers) + one 32-bit per-lane offset, destination = the
// wave's 1 KiB piece at `lds_wave_base` (M0).  Written out: through the builtin every piece's 64-bit per-lane address lived in
// vector registers across the k-loop (eight pointers = 16 registers the accumulators and fragments need: spills, reloaded by
// scratch loads INSIDE the loop).  The compiler does not count these requests: the loop's vmcnt waits are written by hand.
__device__ __forceinline__ void glds16_sbase(const void *base_uniform, unsigned int lane_off, char *lds_wave_base) {
    const unsigned int dst = (unsigned int)(uintptr_t)(lvoid_t *)lds_wave_base;
    asm volatile("s_nop 4\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                 :
                 : "v"(lane_off), "s"(base_uniform), "s"(dst)
                 : "memory", "m0");
}

// the lane index, recomputed where it is called (an asm the compiler cannot hoist or merge): everything addressed from it is then
// computed per tile / per epilogue instead of living in registers across the k-loop, whose accumulators and fragments need them
__device__ __forceinline__ int lane_fresh() {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}

// One 256 x 256 tile.  A_KC = false: the weight-gradient layout (both operands row-contiguous, contraction over token rows;
// M, N multiples of 256).  ADAM: the tile is the gradient of a block of parameters in the flat buffers of `ad` and the epilogue is
// their AdamW step (as in gemm_pipe_body); colsum_a (row-contiguous A): the bias gradient, as there.
template <bool A_KC, bool B_KC, bool ADAM>
__device__ __forceinline__ void gemm256_tile(const skyemb_gemm_args &g, const unsigned int tile, const unsigned int ntiles, char *smem,
                                             const skyemb_adamw_desc *ad = nullptr, const bool linear = false) {
    static_assert(A_KC || !B_KC, "a row-contiguous A comes with a row-contiguous B (weight gradients)");
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const unsigned int tiles_n = ((unsigned int)g.N + 255u) / 256u, tiles_m = ((unsigned int)g.M + 255u) / 256u;
    const lp_t *A = (const lp_t *)g.A + 0;
    const lp_t *B = (const lp_t *)g.B + 0;
    const int KT = g.K / BK, H = 4 * KT;
    const bool colmajor = g.N > g.M;
  {
    unsigned int wg;
    {   // XCD-aware bijective tile order (gemm_pipe_body): consecutive tiles of an XCD share operand panels in its L2
        const unsigned int nwg = ntiles, xcd = tile & 7u, local = tile >> 3;
        const unsigned int q = nwg >> 3, r = nwg & 7u;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
        if (linear) wg = tile;                            // (grouped launches: the caller has already laid the tiles out per XCD)
    }
    const unsigned int div = colmajor ? tiles_m : tiles_n, quo = wg / div, rem = wg - quo * div;
    const int m0 = (int)(colmajor ? rem : quo) * 256, n0 = (int)(colmajor ? quo : rem) * 256;
    const int lane = lane_fresh();
#ifdef GEMM_STAMP
    const unsigned long long t_tile0 = __builtin_amdgcn_s_memtime();
#endif

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ---- half-tile issue: e = the half-tile's kind (compile time in the loop: (q + 3) & 3), k0 its k offset, slot its ring slot.
    // A wave's two pieces of a half-tile = rows [16 wave, +16) of a k-contiguous image (8 rows x 128 B per piece, 16-byte chunk c
    // of row r at chunk c ^ (r & 7): the XOR is on the source address) or k-rows [8 wave, +8) of a row-contiguous one (4 k-rows
    // x 256 B per piece, chunk XOR rc_swz<128>).  Rows past the edge: the 8-row group is moved up to the last whole group (its
    // outputs are never stored), so the base stays uniform.
    const unsigned int a_off = A_KC ? (unsigned int)((lane >> 3) * (int)g.lda * 2 + (((lane & 7) ^ (lane >> 3)) << 4))
                                    : (unsigned int)((lane >> 4) * (int)g.lda * 2 + (((lane & 15) ^ rc_swz<128>(wave * 8 + (lane >> 4))) << 4));
    const unsigned int b_off = B_KC ? (unsigned int)((lane >> 3) * (int)g.ldb * 2 + (((lane & 7) ^ (lane >> 3)) << 4))
                                    : (unsigned int)((lane >> 4) * (int)g.ldb * 2 + (((lane & 15) ^ rc_swz<128>(wave * 8 + (lane >> 4))) << 4));
    auto issue_piece = [&](int e, int k0, int slot, int j) {
        {
            const int idx = wave * 2 + j;
            char *dst = smem + slot * G256_HT + idx * 1024;
            if (e >= 2) {
                if constexpr (A_KC) {
                    int row = m0 + (e - 2) * 128 + idx * 8;
                    row = row + 8 <= g.M ? row : g.M - 8;
                    glds16_sbase(A + (int64_t)row * g.lda + k0, a_off, dst);
                } else {
                    glds16_sbase(A + (int64_t)(k0 + idx * 4) * g.lda + m0 + (e - 2) * 128, a_off, dst);
                }
            } else if constexpr (B_KC) {
                int row = n0 + e * 128 + idx * 8;
                row = row + 8 <= g.N ? row : g.N - 8;
                glds16_sbase(B + (int64_t)row * g.ldb + k0, b_off, dst);
            } else {
                glds16_sbase(B + (int64_t)(k0 + idx * 4) * g.ldb + n0 + e * 128, b_off, dst);
            }
        }
    };
    auto next = [](int s, int by) { s += by; return s >= G256_RING ? s - G256_RING : s; };

    // prologue: half-tiles 0 .. 6, then k-tile 0 (half-tiles 0-3) must have landed
    int iss = 0, iss_slot = 0;                            // next half-tile to issue / its slot
#pragma unroll
    for (int h = 0; h < G256_AHEAD; ++h) {
        if (h < H) {
            issue_piece(h & 3, (h >> 2) * BK, iss_slot, 0);
            issue_piece(h & 3, (h >> 2) * BK, iss_slot, 1);
        }
        ++iss;
        iss_slot = next(iss_slot, 1);
    }
    wait_vmcnt<2 * (G256_AHEAD - 4)>();                   // (K >= 128: half-tiles 0-5 all exist; 4 and 5 stay in flight)
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();            // the second wave group runs half a phase behind
#ifdef GEMM_STAMP
    const unsigned long long t_loop0 = __builtin_amdgcn_s_memtime();
#endif

#ifdef GEMM_STAMP
    // per-wave cycle counts of the four parts of a phase: READ part, wait at the first barrier, MFMA part, wait at the second
    unsigned int seg[4] = {0, 0, 0, 0};
    unsigned long long tprev = __builtin_amdgcn_s_memtime();
    auto stamp = [&](int which) {
        __builtin_amdgcn_sched_barrier(0);
        unsigned long long t_;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");
        __builtin_amdgcn_sched_barrier(0);
        seg[which] += (unsigned int)(t_ - tprev);
        tprev = t_;
    };
#define G256_STAMP(w) stamp(w)
#else
#define G256_STAMP(w)
#endif
    lp8 fa[4][2], fb[2][2][2];                         // A: one 64-row half of the wave's block; B: both 32-column halves
    int rs = 0;                                           // slot of half-tile 4 t
    const int a_rows = 0, b_rows = (wc & 1) * 64;         // first row of the wave's block inside its A / B half-tile
    (void)a_rows;
    auto read_a = [&](int sub) {
        const char *sa = smem + next(rs, 2 + wr) * G256_HT;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int ii = 0; ii < 4; ++ii)
                if constexpr (!A_KC) fa[ii][kk] = frag_rc_asm<128>(sa, sub * 64 + ii * 16, kk, lane);
                else if constexpr (B_KC) fa[ii][kk] = frag_kc(sa, sub * 64 + ii * 16, kk, lane);
                else fa[ii][kk] = frag_kc_asm(sa, sub * 64 + ii * 16, kk, lane);
    };
    auto read_b = [&](int sub) {
        const char *sb = smem + next(rs, wc >> 1) * G256_HT;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                if constexpr (B_KC) fb[sub][jj][kk] = frag_kc(sb, b_rows + sub * 32 + jj * 16, kk, lane);
                else fb[sub][jj][kk] = frag_rc_asm<128>(sb, b_rows + sub * 32 + jj * 16, kk, lane);
            }
    };
    // 32 MFMAs: one 64-row half of the wave's block against all four of its column fragments, both k halves of the k-tile.
    // (LDS-DMA pieces were tried BETWEEN the MFMAs: an LDS-DMA costs ~100 cycles of the wave's issue wherever it stands --
    // stamped: a 16-MFMA part 307 cycles without, 515 with two of them -- so they stay in the READ part, beside the partner's
    // MFMAs.  And FOUR phases of 16 MFMAs per k-tile were measured first: READ + 2 pieces = 435 cycles beside MFMA = 306, two
    // barriers of ~75 each per phase, 1025 cycles per phase = 4100 per k-tile; two phases balance 16 / 8 reads + 4 pieces
    // against 32 MFMAs and halve the barriers.)
    auto mfma_half = [&](int ih) {
#ifndef G256_NOPRIO
        __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int ii = 0; ii < 4; ++ii)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[ih * 4 + ii][j] =
                        sky_mfma_16x16x32(fb[j >> 1][j & 1][kk], fa[ii][kk], acc[ih * 4 + ii][j]);  // D[n][m]
#ifndef G256_NOPRIO
        __builtin_amdgcn_s_setprio(0);
#endif
    };
    // Bias gradient (row-contiguous A only) = column sums of the A tile over k: waves of the first tile column.  ONE accumulator
    // collects the wave's eight
    // 16-row blocks: block b's fragment is multiplied by an operand that is 1 in row b and 0 elsewhere, so D[b][m] += sum_k A[m][k]
    // (4 + 4 registers; a ones-operand per block, as in the ring kernels, would be eight accumulators = 32 registers this kernel
    // does not have).
    const int tile_n = n0 >> 8;
    const bool do_colsum = !A_KC && (wc == 0) && g.colsum_a != nullptr && tile_n == 0;
    f32x4 cacc = (f32x4){0.f, 0.f, 0.f, 0.f};
    // one phase: READ part (fragments of this phase, two half-tiles requested, the counted wait in the last phase of a k-tile),
    // barrier, MFMA part, barrier
    auto phase = [&](auto Q, int t) {
        constexpr int q = decltype(Q)::value;
        if constexpr (q == 0) {
            read_b(0);
            read_b(1);
            __builtin_amdgcn_sched_barrier(0);            // (the B fragments first)
            read_a(0);
        } else {
            read_a(1);
        }
#ifndef G256_NODMA
#pragma unroll
        for (int u = 0; u < 2; ++u) {                     // half-tiles 2 p + 6, 2 p + 7: kinds 2, 3 (A) in the first phase of a k-tile, 0, 1 (B) in the second
            if (iss < H) {
                issue_piece((2 * q + 2 + u) & 3, (iss >> 2) * BK, iss_slot, 0);
                issue_piece((2 * q + 2 + u) & 3, (iss >> 2) * BK, iss_slot, 1);
            }
            ++iss;
            iss_slot = next(iss_slot, 1);
        }
#else
        iss += 2;
        iss_slot = next(iss_slot, 2);
#endif
        if constexpr (q == 1) {
            if (t + 2 < KT) wait_vmcnt<4>();              // k-tile t + 1 has landed; the two half-tiles behind it stay in flight
            else if (t + 1 < KT) wait_vmcnt<0>();
        }
        G256_STAMP(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        G256_STAMP(1);
        if constexpr (!B_KC) {                            // asm fragment reads: counted by hand (see frag_rc_asm), waited for HERE
            lds_wait<0>();
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int ii = 0; ii < 4; ++ii) lds_use(fa[ii][kk]);
            if constexpr (q == 0)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int j = 0; j < 4; ++j) lds_use(fb[j >> 1][j & 1][kk]);
        }
        mfma_half(q);
        if constexpr (!A_KC) {
            if (do_colsum) {
                const int row = lane_fresh() & 15;
#pragma unroll
                for (int ii = 0; ii < 4; ++ii) {
                    lp8 sel;
#pragma unroll
                    for (int e = 0; e < 8; ++e) sel[e] = row == q * 4 + ii ? (lp_t)1.0f : (lp_t)0.0f;
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk) cacc = sky_mfma_16x16x32(sel, fa[ii][kk], cacc);
                }
            }
        }
        G256_STAMP(2);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        G256_STAMP(3);
    };
    for (int t = 0; t < KT; ++t) {
        phase(std::integral_constant<int, 0>{}, t);
        phase(std::integral_constant<int, 1>{}, t);
        rs = next(rs, 4);
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();            // (the first group is a barrier ahead: both have now left the last MFMA part)
#ifdef GEMM_STAMP
    const unsigned long long t_loop_end = __builtin_amdgcn_s_memtime();
#endif

    // ---- epilogue: two halves of 128 rows through LDS (the ring is free), rows leave in pieces of 8 columns per lane
    constexpr int PITCH = 256 * 4 + 16;
    const int lane_e = lane_fresh(), tid = wave * 64 + lane_e;
    if constexpr (!A_KC) {
        if (do_colsum && lane_e < 32) {                   // D[b][m]: lane (m = lane & 15, rows 4 (lane >> 4) + r) -> blocks 0-3 / 4-7
            float *cs_out = g.colsum_a;
#pragma unroll
            for (int r = 0; r < 4; ++r) cs_out[m0 + wr * 128 + (4 * (lane_e >> 4) + r) * 16 + (lane_e & 15)] = cacc[r];
        }
    }
    const int64_t ad_off = ADAM ? (int64_t)(g.out_f32 - ad->g_base) : 0;
    lp_t *out = (lp_t *)g.out;
    lp_t *out2 = (lp_t *)g.out2;
    const lp_t *aux = (const lp_t *)g.aux;
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {
        if (wr == half) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int r = i * 16 + (lane_e & 15), c = wc * 64 + j * 16 + 4 * (lane_e >> 4);
                    *(f32x4 *)(smem + r * PITCH + c * 4) = acc[i][j] * g.alpha;
                }
        }
        __syncthreads();
#pragma unroll 1
        for (int jj = 0; jj < 8; jj += 2) {               // 128 rows x 32 pieces over 512 threads: 8 each, two at a time
            float4 e_bias[2][2], e_res[2][2];
            lp8 e_aux[2];
            float4 e_p[2][2], e_m[2][2], e_v[2][2];       // (ADAM) the parameters and moments the piece updates
#pragma unroll
            for (int u = 0; u < 2; ++u) {                 // every fused input of the two pieces requested up front, clamped
                const int p = tid + (jj + u) * 512, m = m0 + half * 128 + (p >> 5), n = n0 + (p & 31) * 8;
                const int mc = m < g.M ? m : g.M - 1, nc = n < g.N ? n : g.N - 8;
                if constexpr (ADAM) {
                    const int64_t o = ad_off + (int64_t)mc * g.ldo32 + nc;
                    e_p[u][0] = gload4(ad->p + o); e_p[u][1] = gload4(ad->p + o + 4);
                    e_m[u][0] = gload4(ad->m + o); e_m[u][1] = gload4(ad->m + o + 4);
                    e_v[u][0] = gload4(ad->v + o); e_v[u][1] = gload4(ad->v + o + 4);
                    continue;
                }
                if (g.bias) { e_bias[u][0] = gload4(g.bias + nc); e_bias[u][1] = gload4(g.bias + nc + 4); }
                if (g.resid) { e_res[u][0] = gload4(g.resid + (int64_t)mc * g.ldr + nc); e_res[u][1] = gload4(g.resid + (int64_t)mc * g.ldr + nc + 4); }
                if (g.act == SKYEMB_ACT_DGELU) e_aux[u] = gload8h(aux + (int64_t)mc * g.ldaux + nc);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int p = tid + (jj + u) * 512, r = p >> 5, c = (p & 31) * 8;
                const int m = m0 + half * 128 + r, n = n0 + c;
                if (m >= g.M || n >= g.N) continue;
                const float4 lo = *(const float4 *)(smem + r * PITCH + c * 4), hi = *(const float4 *)(smem + r * PITCH + c * 4 + 16);
                float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                if constexpr (ADAM) {
                    // v[] is the gradient of 8 consecutive parameters: the optimiser step, here (adamw_math.h; as in gemm_pipe_body)
                    const float lr = ad->hyper[0], bc1 = ad->hyper[1], bc2 = ad->hyper[2];
                    const SkyAdamScalars sc = sky_adam_scalars(lr, bc1, bc2, ad->beta1, ad->beta2, ad->eps, ad->weight_decay, ad->grad_scale);
                    const int64_t o = ad_off + (int64_t)m * g.ldo32 + n;
                    float pp[8] = {e_p[u][0].x, e_p[u][0].y, e_p[u][0].z, e_p[u][0].w, e_p[u][1].x, e_p[u][1].y, e_p[u][1].z, e_p[u][1].w};
                    float mm[8] = {e_m[u][0].x, e_m[u][0].y, e_m[u][0].z, e_m[u][0].w, e_m[u][1].x, e_m[u][1].y, e_m[u][1].z, e_m[u][1].w};
                    float vv[8] = {e_v[u][0].x, e_v[u][0].y, e_v[u][0].z, e_v[u][0].w, e_v[u][1].x, e_v[u][1].y, e_v[u][1].z, e_v[u][1].w};
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
                auto add8 = [&](const float4 &a, const float4 &b) {
                    v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w; v[4] += b.x; v[5] += b.y; v[6] += b.z; v[7] += b.w;
                };
                if (g.bias) add8(e_bias[u][0], e_bias[u][1]);
                if (g.resid) add8(e_res[u][0], e_res[u][1]);
                if (g.act == SKYEMB_ACT_GELU) {
                    if (out2) store8(out2 + (int64_t)m * g.ldo2 + n, v);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = gelu_f(v[e]);
                } else if (g.act == SKYEMB_ACT_DGELU) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] *= dgelu_f((float)e_aux[u][e]);
                }
                if (g.out_f32) {
                    *(float4 *)(g.out_f32 + (int64_t)m * g.ldo32 + n) = make_float4(v[0], v[1], v[2], v[3]);
                    *(float4 *)(g.out_f32 + (int64_t)m * g.ldo32 + n + 4) = make_float4(v[4], v[5], v[6], v[7]);
                }
                if (out) store8(out + (int64_t)m * g.ldo + n, v);
            }
        }
        __syncthreads();
    }
#ifdef GEMM_STAMP
    {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t_end = __builtin_amdgcn_s_memtime();
        if (g_gemm_stamp && lane_fresh() == 0 && blockIdx.x < 8) {
            unsigned long long *o = g_gemm_stamp + (size_t)(blockIdx.x * 8 + wave) * 8;
            for (int i = 0; i < 4; ++i) atomicAdd(o + i, (unsigned long long)seg[i]);
            atomicAdd(o + 4, (unsigned long long)(2 * KT));
            atomicAdd(o + 5, (unsigned long long)(t_loop0 - t_tile0));      // prologue: tile start -> first phase
            atomicAdd(o + 6, (unsigned long long)(t_end - t_loop_end));     // epilogue incl. store acknowledgement
            atomicAdd(o + 7, 1ull);
        }
    }
#endif
  }
}

// Persistent workgroups (one per CU): tile = blockIdx.x, + gridDim.x, ...  The epilogue's global stores are not waited for: they
// drain under the next tile's prologue and k-loop.
template <bool B_KC>
__global__ __launch_bounds__(512) void gemm256_kernel(const skyemb_gemm_args g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned int ntiles = (((unsigned int)g.N + 255u) / 256u) * (((unsigned int)g.M + 255u) / 256u);
#pragma unroll 1
    for (unsigned int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) gemm256_tile<true, B_KC, false>(g, tile, ntiles, smem);
}

// a single weight gradient: one tile per workgroup
__global__ __launch_bounds__(512) void gemm256_wgrad_kernel(const skyemb_gemm_args g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    gemm256_tile<false, false, false>(g, blockIdx.x, gridDim.x, smem);
}

// Grouped weight gradients (the blob of gemm_pipe_group_kernel below: header with the tile prefix of every problem, then the
// problems; ADAM: the fused-AdamW descriptor in the header): one tile per workgroup.
// (Round 4 also built tiles SHARED between an owner and a helper workgroup for launches with idle compute units -- mim_19: 192
// tiles on 256 CUs -- and measured them a third slower, 413 against 303 us per launch: the launch is bound by what its tiles miss in
// the L2s, and owners and helpers on different k-ranges share fewer panels.  Removed in round 5; profiles/HISTORY.md keeps the numbers.)
// SIDE: workgroups behind the tiles stream another slice's AdamW step (side_adamw_job, gemm_pipe.hip): at ViT-L a block's four weight
// gradients are 192 tiles for 256 compute units -- 64 units idle for the whole launch, and HBM idle under everybody's k-loops.
template <bool ADAM, bool SIDE = false>
__global__ __launch_bounds__(512) void gemm256_group_kernel(const char *__restrict__ blob) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int *hdr = (const int *)blob;
    if constexpr (SIDE) {
        if ((int)blockIdx.x >= hdr[2]) return side_job<512, 4, 4>(blob, smem);
    }
    const int n = hdr[0];
    // Tile order across the GROUP (round 5; header word 1 bit 30 = on): workgroup b runs on XCD b & 7, and XCD x takes the x-th
    // EIGHTH of the concatenated tile list -- 24 consecutive tiles of (mostly) one problem: a 6 x 4 block sharing 10 operand panels
    // per k-tile -- instead of an eighth of EVERY problem (8 + 8 + 6 + 2 tiles of the four weight gradients of a ViT-L block: 21
    // panels per k-tile for the same 48 panel loads; PMC round 4: 62 % L2 hits, 2.6 x the operand bytes from beyond the L2).
    const bool by_xcd = (hdr[1] >> 30) & 1;
    int gt = blockIdx.x;
    if (by_xcd) gt = (gt & 7) * (hdr[8 + n] >> 3) + (gt >> 3);
    int p = 0, first = 0;                                 // (the tile prefix is walked in memory: n <= 32 scalar loads per tile)
#pragma unroll 1
    for (int i = 1; i < n; ++i) {
        const int s_i = hdr[8 + i];
        if (gt >= s_i) { p = i; first = s_i; }
    }
    const skyemb_gemm_args g = ((const skyemb_gemm_args *)(blob + GROUP_HEADER_BYTES))[p];
    const unsigned int tb = gt - first;
    const unsigned int ntiles = ((unsigned int)g.M / 256u) * ((unsigned int)g.N / 256u);
    if (tb >= ntiles) return;                             // padding up to the next multiple of 8 (keeps tb & 7 == XCD)
    gemm256_tile<false, false, ADAM>(g, tb, ntiles, smem, ADAM ? (const skyemb_adamw_desc *)(blob + GROUP_ADAMW_OFFSET) : nullptr, by_xcd);
}

// the weight-gradient variant: both operands row-contiguous, whole tiles, K a multiple of 64 with at least two k-tiles
bool gemm256_wgrad_applicable(const skyemb_gemm_args &g) {
    return g.a_layout == SKYEMB_RC && g.b_layout == SKYEMB_RC && !g.dst_row && !g.tab_row && !g.table && g.split_k <= 1 &&
           g.K % BK == 0 && g.K >= 2 * BK && g.M % 256 == 0 && g.N % 256 == 0 && g.M >= 256 && g.N >= 256 &&
           g.lda * 2 * 4 < (1ll << 31) && g.ldb * 2 * 4 < (1ll << 31);
}

template <bool ADAM, bool SIDE = false>
int gemm256_group_launch(const void *blob_dev, int total_blocks, hipStream_t st) {
    constexpr int smem = G256_RING * G256_HT;
    auto kern = gemm256_group_kernel<ADAM, SIDE>;
    static std::mutex attr_mutex;
    static bool attr_done[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    {
        std::lock_guard<std::mutex> lock(attr_mutex);
        if (!attr_done[dev & 63]) {
            hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
            if (e != hipSuccess) {
                skyemb_set_error("skyemb_gemm_group_launch(256x256): hipFuncSetAttribute(%d B LDS): %s", smem, hipGetErrorString(e));
                return 2;
            }
            attr_done[dev & 63] = true;
        }
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)total_blocks), dim3(512), smem, st, (const char *)blob_dev);
    skyemb_count_gemm(SKYEMB_GEMM_COUNT_GROUP256);
    SKY_LAUNCH_CHECK("skyemb_gemm_group_launch(256x256)");
    return 0;
}

// what the 256 x 256 kernel takes: k-contiguous A, no row maps / column sums / split-K, K a multiple of 64 with at least two k-tiles
bool gemm256_applicable(const skyemb_gemm_args &g) {
    return g.a_layout == SKYEMB_KC && !g.dst_row && !g.tab_row && !g.table && !g.colsum_a && g.split_k <= 1 &&
           g.K % BK == 0 && g.K >= 2 * BK && g.M % 8 == 0 && g.M >= 8 && g.N % 8 == 0 && (g.b_layout == SKYEMB_KC || g.N % 128 == 0) &&
           g.lda * 2 * 8 < (1ll << 31) && g.ldb * 2 * 8 < (1ll << 31);
}

template <bool B_KC>
int gemm256_launch_n(const skyemb_gemm_args &g, hipStream_t st) {
    constexpr int smem = G256_RING * G256_HT;             // all 160 KiB
    auto kern = gemm256_kernel<B_KC>;
    static std::mutex attr_mutex;
    static bool attr_done[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    {
        std::lock_guard<std::mutex> lock(attr_mutex);
        if (!attr_done[dev & 63]) {
            hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
            if (e != hipSuccess) {
                skyemb_set_error("skyemb_gemm(256x256): hipFuncSetAttribute(%d B LDS): %s", smem, hipGetErrorString(e));
                return 2;
            }
            attr_done[dev & 63] = true;
        }
    }
    const int64_t tiles = ceil_div64(g.M, 256) * ceil_div64(g.N, 256);
    static const int ncu = []() {
        int dev_ = 0, n = 256;
        if (hipGetDevice(&dev_) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev_);
        return n > 0 ? n : 256;
    }();
    static const bool persist = []() { const char *e = getenv("SKYEMB_GEMM_256_PERSIST"); return !(e && e[0] == '0'); }();
    hipLaunchKernelGGL(kern, dim3((unsigned)(tiles < ncu || !persist ? tiles : ncu)), dim3(512), smem, st, g);
    skyemb_count_gemm(SKYEMB_GEMM_COUNT_256);
    SKY_LAUNCH_CHECK("skyemb_gemm(256x256)");
    return 0;
}

int gemm256_wgrad_launch(const skyemb_gemm_args &g, hipStream_t st) {
    constexpr int smem = G256_RING * G256_HT;
    static std::mutex attr_mutex;
    static bool attr_done[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    {
        std::lock_guard<std::mutex> lock(attr_mutex);
        if (!attr_done[dev & 63]) {
            hipError_t e = hipFuncSetAttribute((const void *)gemm256_wgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
            if (e != hipSuccess) {
                skyemb_set_error("skyemb_gemm(256x256, weight gradient): hipFuncSetAttribute(%d B LDS): %s", smem, hipGetErrorString(e));
                return 2;
            }
            attr_done[dev & 63] = true;
        }
    }
    hipLaunchKernelGGL(gemm256_wgrad_kernel, dim3((unsigned)((g.M / 256) * (g.N / 256))), dim3(512), smem, st, g);
    skyemb_count_gemm(SKYEMB_GEMM_COUNT_256);
    SKY_LAUNCH_CHECK("skyemb_gemm(256x256, weight gradient)");
    return 0;
}

int gemm256_launch(const skyemb_gemm_args &g, hipStream_t st) {
    if (gemm256_wgrad_applicable(g)) return gemm256_wgrad_launch(g, st);
    if (!gemm256_applicable(g)) {
        skyemb_set_error("skyemb_gemm(256x256): the problem is outside this tile's subset (k-contiguous A, plain epilogue, K >= 128)");
        return 1;
    }
    return g.b_layout == SKYEMB_KC ? gemm256_launch_n<true>(g, st) : gemm256_launch_n<false>(g, st);
}
