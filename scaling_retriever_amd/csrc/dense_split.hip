// Dense scoring in split-bf16 arithmetic: fp32-class inner products on the bf16 MFMA pipe.
//
// Every fp32 operand is split into bf16 planes, x = p0 + p1 (+ p2), each plane the bf16 rounding of what
// the previous ones left over (the subtractions are exact in fp32):
//   bf16x3:  q . d ~= q0.d0 + q0.d1 + q1.d0                        (dropped terms O(2^-17 |q||d|))
//   bf16x6:  q . d ~= q0.d0 + q0.d1 + q1.d0 + q1.d1 + q0.d2 + q2.d0  (dropped terms O(2^-25): three planes carry
//            the full 24-bit fp32 significand, so the result is in the error class of an fp32 dot product)
// Products of bf16 pairs are exact in fp32 and v_mfma_f32_16x16x32_bf16 accumulates in fp32.  The plane pairs are
// accumulated smallest first.  3 (6) bf16 MFMAs at the 2.5 PF bf16 rate replace one fp32 MFMA at 157 TF.
//
// Implementation: ONE GEMM with an n_pairs x longer k loop.  k-tile kt in [0, n_pairs * H/64): pair = kt / (H/64);
// the doc operand streams D plane pair_d[pair], the query operand Q plane pair_q[pair].  Tile 256 docs x 256
// queries, 8 waves, LDS-DMA staging with source-side swizzle and the 4-phase fragment pipeline of
// csrc/gemm_bf16.hip.  Docs sit on the accumulator registers and queries on the lanes, so the tau filter is
// lane-local and survivors are appended as 64-bit keys exactly as in dense_score.hip.
#include "dense_split.h"

// Timing-only diagnostics (SR_SPLIT_DIAG: wrong results by design; SR_SPLIT_STAMPS: per-tile s_memrealtime stamps) exist only in a
// diagnostic build (make EXTRA=-DSR_DIAG_BUILD, tools/micro): in the product library the tests below are compile-time constants
// and the kernel carries no trace of them.
#ifdef SR_DIAG_BUILD
#define SR_SPLIT_DIAG_BIT(a, bit) ((a).diag & (bit))
#define SR_SPLIT_STAMPS_PTR(a) ((a).stamps)
#else
#define SR_SPLIT_DIAG_BIT(a, bit) 0
#define SR_SPLIT_STAMPS_PTR(a) (static_cast<unsigned long long*>(nullptr))
#endif

typedef __bf16 mfma_bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void* lds_void_ptr;
typedef __attribute__((address_space(1))) const void* gbl_void_ptr;

// ---- fp32 -> bf16 planes ---------------------------------------------------------------------------
__global__ void split_bf16_kernel(const float* __restrict__ src, unsigned short* __restrict__ p0, unsigned short* __restrict__ p1,
                                  unsigned short* __restrict__ p2, int64_t n4) {
    // grid-stride: a launch may not carry 2^32 or more threads (HIP), and 8 841 823 x 2048 / 4 elements is 4.5e9
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const f32x4 v = reinterpret_cast<const f32x4*>(src)[i];
        bf16x4 a, b, c;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const unsigned short h = f32_to_bf16(v[e]);
            const float r1 = v[e] - bf16_to_f32(h);          // exact
            const unsigned short m = f32_to_bf16(r1);
            const float r2 = r1 - bf16_to_f32(m);            // exact
            a[e] = (short)h;
            b[e] = (short)m;
            c[e] = (short)f32_to_bf16(r2);
        }
        reinterpret_cast<bf16x4*>(p0)[i] = a;
        if (p1) reinterpret_cast<bf16x4*>(p1)[i] = b;
        if (p2) reinterpret_cast<bf16x4*>(p2)[i] = c;
    }
}

int launch_split_bf16(const float* src, unsigned short* p0, unsigned short* p1, unsigned short* p2, int64_t n_elems, hipStream_t s) {
    const int64_t n4 = n_elems / 4;
    if (n4 == 0) return SR_OK;
    int64_t blocks = ceil_div64(n4, 256);
    if (blocks > (1 << 22)) blocks = 1 << 22;          // 2^30 threads per launch, the rest by the grid stride
    hipLaunchKernelGGL(split_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, s, src, p0, p1, p2, n4);
    SR_CHECK_LAUNCH();
    return SR_OK;
}

#define SP_BN 256   // docs per workgroup
#define SP_BM 256   // queries per workgroup
typedef _Float16 mfma_f16x8 __attribute__((ext_vector_type(8)));
template <bool F16>
__device__ __forceinline__ f32x4 split_mma(const mfma_bf16x8& w, const mfma_bf16x8& a, const f32x4& c) {
    if constexpr (F16)
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(mfma_f16x8, w), __builtin_bit_cast(mfma_f16x8, a), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, a, c, 0, 0, 0);
}

// ---- epilogue: lane = query (frow + 16 j), registers = docs (16 i + 4 fg + r): tau filter, survivors as 64-bit keys ----
// UB (the certified filter's pass): the accumulators are turned into upper bounds of the exact score first, in the scaled
// domain, U' = acc + A'[q] x[j] + B'[q] y[j] ((x, y) of the tile's 256 documents staged in LDS by the tile prologue), and
// compared with tau[q] * sq * sd (a power of two: exact); a survivor's key carries U = U' / (sq sd).
template <bool UB, int NB, int MB>
__device__ __forceinline__ void split_epilogue(const DenseSplitArgs& a, f32x4 (&acc)[NB][MB], int64_t row0, int q0, int wn, int wm,
                                               int frow, int fg, const float* xy_s, const float* qa_s, const float* tau_s, float gx, float gy) {
#pragma clang fp contract(off)
    // the workgroup is persistent: without this hipcc hoists the 32 per-register row indices, id offsets and slots of this
    // epilogue out of the tile loop and carries (spills) them through the k-loop
    asm volatile("" : "+v"(fg), "+v"(frow));
    const int64_t left = a.row_end - row0;
    const int rows_valid = left < SP_BN ? (int)left : SP_BN;
    const uint32_t gid0 = a.id_base + (uint32_t)row0 * a.id_stride;
    if (SR_SPLIT_DIAG_BIT(a, 2)) return;                   // timing only (SR_SPLIT_DIAG): no epilogue at all
    f32x4 qa[MB];
    // UB, block test: a block of 4 accumulators is first held against tq - e_max, e_max the error term with the largest x and y of
    // the wave's 128 documents (gx, gy: one scalar load per tile).  e(q, j) <= e_max, so a block without an accumulator at or above
    // that line has no survivor and its 4 bounds are never formed; the blocks that pass (a handful per wave and tile) get their
    // bounds, the test against tau and their keys exactly as before.  Two fmas and an LDS read per pair had been the bulk of the
    // epilogue's instructions.
    const bool pre = UB && a.dxy_gmax != nullptr;
    if constexpr (UB) {
#pragma unroll
        for (int j = 0; j < MB; ++j) qa[j] = *reinterpret_cast<const f32x4*>(qa_s + (wm * MB * 16 + j * 16 + frow) * 4);
        if (!pre && !SR_SPLIT_DIAG_BIT(a, 1))             // timing only (SR_SPLIT_DIAG): plane product without the error term
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const float* p = xy_s + 2 * (wn * NB * 16 + i * 16 + fg * 4);
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(p), v1 = *reinterpret_cast<const f32x4*>(p + 4);
            const float x[4] = {v0[0], v0[2], v1[0], v1[2]}, y[4] = {v0[1], v0[3], v1[1], v1[3]};
#pragma unroll
            for (int j = 0; j < MB; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    acc[i][j][r] = __builtin_fmaf(qa[j][0], x[r], __builtin_fmaf(qa[j][1], y[r], acc[i][j][r]));
        }
    }
    // Survivors.  In-kernel stamps showed this epilogue at 10 us of a 60 us tile - not waiting for memory, but executing: a
    // wave runs the per-element `if (score >= tau) store` code of a query (128 exec-masked micro-branches) whenever ANY of its
    // 64 lanes keeps something, which is nearly always, although a wave keeps only a handful of documents per tile.  So: per
    // (query j, block of 4 rows) one lane-local maximum and a WAVE-uniform skip; the few blocks that hold a survivor store it
    // straight into the lane's own segment of the query's candidate buffer (common.h TopkWS: no atomic, no second pass), the
    // count goes to seg_cnt.  A (lane, query) pair with more than SR_SEG_P survivors (the first launches of a search, before
    // tau has risen) or a launch without segments takes the counted path behind one atomic reservation.
    if (SR_SPLIT_DIAG_BIT(a, 8)) {                         // timing only: the error term alone
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < NB; ++i)
#pragma unroll
            for (int j = 0; j < MB; ++j) t += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
        if (t == 1.2345e-30f) a.cand_count[0] = 1;
        return;
    }
    const int seg_idx = (int)((row0 - a.row_begin) / SP_BN) * SR_SEG_PROD + wn * 4 + fg;
    const bool seg_ok = a.seg_cnt != nullptr && seg_idx < a.seg_n;
    const bool full = rows_valid == SP_BN;
    bool slow[MB];
    float tq[MB], out_scale[MB];
#pragma unroll
    for (int j = 0; j < MB; ++j) {
        const int q = q0 + wm * MB * 16 + j * 16 + frow;
        tq[j] = q < a.nq ? tau_s[wm * MB * 16 + j * 16 + frow] : INFINITY;          // a query beyond the batch keeps nothing
        out_scale[j] = 1.f;
        if constexpr (UB) {
            tq[j] = tq[j] * (qa[j][2] * a.sd);           // scaled-domain threshold (-inf stays -inf)
            out_scale[j] = qa[j][3] * a.isd;
        }
        uint64_t* seg_dst = a.cand_keys + (int64_t)q * a.cand_cap + a.seg_off + (int64_t)seg_idx * SR_SEG_P;
        // the line of the block test.  Sound in fp32: a pair is kept when fl(A' x + fl(B' y + acc)) >= tq, which implies
        // acc >= tq - e - 2.0001 u (|acc| + e), u = 2^-24; |acc| <= |q0||d0| (1 + H u) <= e (1 / sigma + 1) wherever the pair is near the
        // line (dense_filter.hip: x carries sigma |d'|), so 2.0001 u |acc| < 2^-11 e; the line below sits at
        // tq - e_max (1 + 2^-7) - 2^-22 |tq|, which also covers its own three roundings (<= 3 u (|tq| + 1.01 e_max)).
        // tq = -inf (no threshold yet): the line is -inf and every block passes; a query that cannot be filtered (A' = inf) passes too.
        float line = tq[j];
        if constexpr (UB)
            if (pre) line = (tq[j] - (qa[j][0] * gx + qa[j][1] * gy) * 1.0078125f) - fabsf(tq[j]) * 2.384185791015625e-07f;
        int n = 0;
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int lr0 = wn * NB * 16 + i * 16 + fg * 4;
            float m4 = fmaxf(fmaxf(acc[i][j][0], acc[i][j][1]), fmaxf(acc[i][j][2], acc[i][j][3]));
            if (!full) {
                m4 = -INFINITY;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (lr0 + r < rows_valid) m4 = fmaxf(m4, acc[i][j][r]);
            }
            if (__ballot(m4 >= line) == 0) continue;                   // wave-uniform: nobody keeps anything of this block
            if constexpr (UB)
                if (pre) {                                             // the block's 4 bounds, in place (the counted path below reads them)
                    const float* p = xy_s + 2 * lr0;
                    const f32x4 v0 = *reinterpret_cast<const f32x4*>(p), v1 = *reinterpret_cast<const f32x4*>(p + 4);
                    const float x[4] = {v0[0], v0[2], v1[0], v1[2]}, y[4] = {v0[1], v0[3], v1[1], v1[3]};
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        acc[i][j][r] = __builtin_fmaf(qa[j][0], x[r], __builtin_fmaf(qa[j][1], y[r], acc[i][j][r]));
                }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float sc = acc[i][j][r];
                if (lr0 + r < rows_valid && sc >= tq[j]) {
                    if (seg_ok && n < SR_SEG_P) seg_dst[n] = sr_make_key(UB ? sc * out_scale[j] : sc, gid0 + (uint32_t)(lr0 + r) * a.id_stride);
                    ++n;
                }
            }
        }
        slow[j] = n > 0 && (!seg_ok || n > SR_SEG_P);
        if (n > 0 && !slow[j]) a.seg_cnt[(int64_t)q * a.seg_n + seg_idx] = (unsigned char)n;
    }
    if (SR_SPLIT_DIAG_BIT(a, 4)) return;                  // timing only: the counted path skipped
    bool any_slow = false;
#pragma unroll
    for (int j = 0; j < MB; ++j) any_slow = any_slow || slow[j];
    if (__ballot(any_slow) == 0) return;
#pragma unroll
    for (int j = 0; j < MB; ++j) {
        if (__ballot(slow[j]) == 0) continue;
        const int q = q0 + wm * MB * 16 + j * 16 + frow;
        int cnt = 0;
        if (slow[j]) {
#pragma unroll
            for (int i = 0; i < NB; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int lr = wn * NB * 16 + i * 16 + fg * 4 + r;
                    cnt += (lr < rows_valid && acc[i][j][r] >= tq[j]) ? 1 : 0;
                }
        }
        if (cnt == 0) continue;
        int p = atomicAdd(&a.cand_count[q], cnt);
        uint64_t* dst = a.cand_keys + (int64_t)q * a.cand_cap;
        const int64_t limit = a.seg_off > 0 ? a.seg_off : a.cand_cap;          // the atomically appended candidates stay in the head
#pragma unroll
        for (int i = 0; i < NB; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int lr = wn * NB * 16 + i * 16 + fg * 4 + r;
                const float sc = acc[i][j][r];
                if (lr < rows_valid && sc >= tq[j]) {
                    if (p < limit) dst[p] = sr_make_key(UB ? sc * out_scale[j] : sc, gid0 + (uint32_t)lr * a.id_stride);
                    ++p;
                }
            }
    }
}

// Tile slot -> (doc tile, query tile).  Workgroups are dealt to the 8 XCDs round-robin by their linear id, every XCD has its
// own 4 MB L2, and 256 workgroups are resident at a time (one per CU), 32 per XCD.  xcd_order: the 32 tiles an XCD works on
// together form a block of 8 doc tiles x 4 query tiles (12 operand tiles behind 32 output tiles - in query-tile-fastest
// linear order they touch ~9 doc tiles and most of the query tiles), blocks walked query-block fastest so that a doc tile is
// fetched from HBM once and the query planes stay in the Infinity Cache.  Without it: query tile fastest.  Workgroups are
// persistent: workgroup b takes slots b, b + G, b + 2G, ... (G a multiple of 8, so all its slots belong to its XCD).
struct SplitGrid { int qt, dt, bq, bd, nbq, total; };
static inline SplitGrid split_grid(int64_t rows, int nq, int xcd_order) {
    SplitGrid g;
    g.qt = (int)ceil_div64(nq, SP_BM);
    g.dt = (int)ceil_div64(rows, SP_BN);
    g.bq = !xcd_order ? 1 : (g.qt % 4 == 0 ? 4 : g.qt % 2 == 0 ? 2 : 1);
    g.bd = 32 / g.bq;
    g.nbq = g.qt / g.bq;
    g.total = !xcd_order ? g.qt * g.dt : (int)(ceil_div64((int64_t)g.nbq * ceil_div64(g.dt, g.bd) * 32, 256) * 256);
    return g;
}
__device__ __forceinline__ bool split_tile_of(const DenseSplitArgs& a, int lin, int& d_tile, int& q_tile) {
    if (!a.xcd_order) {
        q_tile = lin % a.grid_qt;
        d_tile = lin / a.grid_qt;
        return true;
    }
    const int s = lin >> 3;
    const int c = (s >> 5) * 256 + (lin & 7) * 32 + (s & 31);      // XCD x of a round: compact indices [256 r + 32 x, + 32)
    const int blk = c >> 5, w2 = c & 31;
    const int bd_i = blk / a.grid_nbq, bq_i = blk - bd_i * a.grid_nbq;
    d_tile = bd_i * a.grid_bd + w2 / a.grid_bq;
    q_tile = bq_i * a.grid_bq + w2 % a.grid_bq;
    return d_tile < a.grid_dt;
}

// ---- scoring kernel ----------------------------------------------------------------------------------
// Persistent workgroups: the LDS-DMA of the NEXT tile's first two k-steps is issued during the last two k-steps of the
// current one (the stages they free), so a tile's prologue latency and most of its epilogue hide behind the neighbour
// tile's transfers; at K = 2048 a tile lives only 32 k-steps, and an exposed prologue + epilogue per tile was ~10 % of it.
template <bool UB>
__global__ __launch_bounds__(512, 2) void dense_split_kernel(DenseSplitArgs a) {
    constexpr int NB = 8, MB = 4, WAVES_M = 4, HB = NB / 2;   // wave tile: 128 docs x 64 queries
    constexpr int W_BYTES = SP_BN * 128, STAGE_BYTES = (SP_BN + SP_BM) * 128;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* xy_s = reinterpret_cast<float*>(smem + 2 * STAGE_BYTES);       // UB: (x, y) of the tile's 256 documents
    float* qa_s = xy_s + 2 * SP_BN;                                        // UB: (A', B', sq, 1 / sq) of the tile's 256 queries
    float* tau_s = qa_s + 4 * SP_BM;                                       // tau of the tile's 256 queries
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int G = (int)gridDim.x;
    const int H = a.H;

    const int wave_s = __builtin_amdgcn_readfirstlane(wave);      // scalar: the LDS-DMA destinations (m0) need no vector arithmetic
    const int wn = wave_s / WAVES_M, wm = wave_s % WAVES_M;       // scalar as well: the epilogue's row / query bases, the group of gx, gy
    uint32_t doff[4], qoff[4];     // BYTE offsets inside the tile's row block: the block's base is wave-uniform, so a piece's address is
                                   // scalar base + 32-bit lane offset (the saddr form of global_load_lds: no 64-bit vector add per piece)
    int64_t st_dbase = 0, st_qbase = 0;
    // k-tiles are staged strictly in order (0, 1, 2, ...), so the plane pair of the NEXT tile is tracked incrementally:
    // the plane pointers change once per H / 64 tiles.  (Looking them up per tile - a division, then two dependent
    // scalar loads from the kernel arguments - sat right behind the k-step barrier, in front of the LDS-DMA issue.)
    int st_pair = 0, st_k0 = 0;
    const unsigned short* st_d = a.D[a.pair_d[0]];
    const unsigned short* st_q = a.Q[a.pair_q[0]];
    int64_t st_row0 = 0;          // the tile `stage` is streaming
    int st_q0 = 0;
    auto next_slot = [&](int t) {
        int d_t, q_t;
        while (t < a.grid_total && !split_tile_of(a, t, d_t, q_t)) t += G;
        return t;
    };
    auto set_tile = [&](int lin) {
        int d_tile, q_tile;
        (void)split_tile_of(a, lin, d_tile, q_tile);
        st_row0 = a.row_begin + (int64_t)d_tile * SP_BN;
        st_q0 = q_tile * SP_BM;
        st_dbase = st_row0 * H;
        st_qbase = (int64_t)st_q0 * H;
        const int64_t dleft = a.row_end - 1 - st_row0;
        const int dmax = dleft < SP_BN - 1 ? (int)dleft : SP_BN - 1, qmax = a.nq - 1 - st_q0 < SP_BM - 1 ? a.nq - 1 - st_q0 : SP_BM - 1;
        // the workgroup is persistent and the k-loop leaves no register free: whatever is derived from the lane index alone is invariant in
        // the tile loop, gets hoisted above it and is then SPILLED through it - the tile prologue had become a chain of scratch loads,
        // each waited for on its own.  So the lane index is made opaque wherever a tile's set-up needs it: recomputing costs 2-3 VALU.
        int lane_o = lane;
        asm volatile("" : "+v"(lane_o));
        const int srow = lane_o >> 3;
        const int schunk = (lane_o & 7) ^ (srow & 7);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = (wave_s * 4 + i) * 8 + srow;
            doff[i] = (uint32_t)((r < dmax ? r : dmax) * H + schunk * 8) * 2u;
            qoff[i] = (uint32_t)((r < qmax ? r : qmax) * H + schunk * 8) * 2u;
        }
        st_pair = 0; st_k0 = 0;
        st_d = a.D[a.pair_d[0]];
        st_q = a.Q[a.pair_q[0]];
    };
    auto stage_pieces = [&](int st) {   // the 8 pieces of the next k-tile in order; no branch: the steady-state k-step stays ONE scheduling region
        unsigned char* wbase = smem + st * STAGE_BYTES + (wave_s * 4) * 1024;
        unsigned char* abase = smem + st * STAGE_BYTES + W_BYTES + (wave_s * 4) * 1024;
        const unsigned char* dsrc = reinterpret_cast<const unsigned char*>(st_d + (st_dbase + st_k0));
        const unsigned char* qsrc = reinterpret_cast<const unsigned char*>(st_q + (st_qbase + st_k0));
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((gbl_void_ptr)(dsrc + doff[i]), (lds_void_ptr)(wbase + i * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((gbl_void_ptr)(qsrc + qoff[i]), (lds_void_ptr)(abase + i * 1024), 16, 0, 0);
        st_k0 += 64;
    };
    auto stage_wrap = [&]() {           // the end of a plane: on to the next plane pair
        if (st_k0 == H) {
            st_k0 = 0;
            ++st_pair;
            if (st_pair < a.n_pairs) {
                st_d = a.D[a.pair_d[st_pair]];
                st_q = a.Q[a.pair_q[st_pair]];
            }
        }
    };
    auto stage = [&](int st) {
        stage_pieces(st);
        stage_wrap();
    };

    const int frow = lane & 15, fg = lane >> 4;
    const int nkt = a.n_pairs * (H / 64);      // >= 2 (checked at launch)
    f32x4 acc[NB][MB];
    mfma_bf16x8 wx[HB], wy[HB], a0[MB], a1[MB];
    auto load_w = [&](int st, int kk, int h, mfma_bf16x8 (&wf)[HB]) {
        const unsigned char* wt = smem + st * STAGE_BYTES;
        const int pos = ((4 * kk + fg) ^ (frow & 7)) * 16;
#pragma unroll
        for (int i = 0; i < HB; ++i)
            wf[i] = *reinterpret_cast<const mfma_bf16x8*>(wt + (wn * NB * 16 + (h * HB + i) * 16 + frow) * 128 + pos);
    };
    auto load_a = [&](int st, int kk, mfma_bf16x8 (&af)[MB]) {
        const unsigned char* at = smem + st * STAGE_BYTES + W_BYTES;
        const int pos = ((4 * kk + fg) ^ (frow & 7)) * 16;
#pragma unroll
        for (int j = 0; j < MB; ++j)
            af[j] = *reinterpret_cast<const mfma_bf16x8*>(at + (wm * MB * 16 + j * 16 + frow) * 128 + pos);
    };
#define SR_MFMA_HALF(HH, WF, AF)                                                                              \
    _Pragma("unroll") for (int i = 0; i < HB; ++i)                                                            \
        _Pragma("unroll") for (int j = 0; j < MB; ++j)                                                        \
            acc[(HH) * HB + i][j] = split_mma<UB>(WF[i], AF[j], acc[(HH) * HB + i][j]);

    int tile = next_slot((int)blockIdx.x);
    if (tile >= a.grid_total) return;
    set_tile(tile);
    stage(0);
    stage(1);
    int buf = 0;
    bool carried = false;       // the previous tile's last k-step already read this tile's first fragments (and its barriers vouched for k-step 0)
    for (;;) {
        const int64_t row0 = st_row0;           // this tile (set_tile moves st_* on to the next one inside the k-loop)
        const int q0 = st_q0;
        const int tile_next = next_slot(tile + G);
        const bool has_next = tile_next < a.grid_total;
        float gx = 0.f, gy = 0.f;           // UB: the largest x and y among the wave's 128 documents (a scalar load, under the k-loop)
        if constexpr (UB)
            if (a.dxy_gmax) {
                const float* gm = a.dxy_gmax + 2 * ((row0 >> 7) + wn);
                gx = gm[0];
                gy = gm[1];
            }
#pragma unroll
        for (int i = 0; i < NB; ++i)
#pragma unroll
            for (int j = 0; j < MB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        // k-steps 0 and 1 of this tile are on their way (issued above, or under the previous tile's last two k-steps).  The barrier ends
        // the previous tile's epilogue reads of xy_s / qa_s / tau_s before they are refilled below.  A carried tile needs no more than
        // that: k-step 0 landed before the previous tile's last barrier, k-step 1 is drained by this tile's first k-step barrier, and
        // the epilogue's stores of candidates stay in flight (a bare s_barrier: __syncthreads() would wait for every one of them)
        if (carried) {
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        const unsigned long long st0 = SR_SPLIT_STAMPS_PTR(a) ? __builtin_amdgcn_s_memrealtime() : 0;
        int lane_t = lane;            // opaque: see set_tile
        asm volatile("" : "+v"(lane_t));
        if (wave_s >= 4) {    // tau of the tile's queries: one 4-byte LDS-DMA piece per lane of waves 4-7 (no wait in the epilogue)
            int q = q0 + (wave_s - 4) * 64 + lane_t;
            q = q < a.nq ? q : a.nq - 1;
            __builtin_amdgcn_global_load_lds((gbl_void_ptr)(a.tau + q), (lds_void_ptr)(tau_s + (wave_s - 4) * 64), 4, 0, 0);
        }
        if constexpr (UB) {
            // the tile's per-document (x, y): 512 floats, 64 per wave, one 4-byte LDS-DMA piece per lane, and its per-query
            // constants: 256 x 16 bytes, one 16-byte piece per lane of waves 0-3; they land under the k-loop (every k-step
            // drains vmcnt before its barrier) and cost no register there
            int64_t r = row0 + wave_s * 32 + (lane_t >> 1);
            r = r < a.row_end ? r : a.row_end - 1;
            __builtin_amdgcn_global_load_lds((gbl_void_ptr)(a.dxy + r * 2 + (lane_t & 1)), (lds_void_ptr)(xy_s + wave_s * 64), 4, 0, 0);
            if (wave_s < 4) {
                int q = q0 + wave_s * 64 + lane_t;
                q = q < a.nq ? q : a.nq - 1;
                __builtin_amdgcn_global_load_lds((gbl_void_ptr)(a.qa + (int64_t)q * 4), (lds_void_ptr)(qa_s + wave_s * 256), 16, 0, 0);
            }
        }
        if (!carried) {
            load_w(buf, 0, 0, wx);
            load_a(buf, 0, a0);
        }
        int kt = 0;
        // steady state with the issue order pinned (see gemm_bf16.hip): reads and LDS-DMA pieces dealt out one per MFMA
#define SR_SGB(MASK, N, ID) __builtin_amdgcn_sched_group_barrier(MASK, N, ID)
        // The plane-pair switch of the staged stream used to be a branch inside stage(): it split the k-step's last phase into blocks,
        // and that phase's 16 MFMAs were issued in a clump BEHIND the 8 LDS-DMA pieces and their address arithmetic instead of
        // between them.  The steady state now runs in stretches that stay inside one plane (the whole tile for the filter's pass).
        // With a next tile the same k-step body also runs this tile's LAST TWO k-steps: the stage they free takes the next tile's k-steps
        // 0 and 1, and the very last one reads the next tile's first fragments (`carried`).  The plain two-k-step tail below (reads in
        // clumps, a drain of every outstanding store at the next tile's top) is left to a workgroup's last tile.
        const int n_sched = has_next ? nkt : nkt - 2;
        bool switched = false;
        while (kt < n_sched) {
        int to_go = nkt - 2 - kt;
        if (to_go <= 0) {                   // the staged stream has reached the next tile
            if (!switched) { set_tile(tile_next); switched = true; }
            to_go = nkt - kt;
        }
        const int in_plane = (H - st_k0) >> 6;
        const int stretch = in_plane < to_go ? in_plane : to_go;          // >= 1: st_k0 < H after every stage_wrap()
#pragma unroll 1
        for (int it = 0; it < stretch; ++it, ++kt) {
            load_w(buf, 0, 1, wy);
            load_a(buf, 1, a1);
            SR_MFMA_HALF(0, wx, a0)
#pragma unroll
            for (int i = 0; i < HB + MB; ++i) { SR_SGB(0x008, 1, 0); SR_SGB(0x100, 1, 0); }
            SR_SGB(0x008, HB * MB - HB - MB, 0);
            __builtin_amdgcn_sched_barrier(0);
            load_w(buf, 1, 0, wx);
            SR_MFMA_HALF(1, wy, a0)
#pragma unroll
            for (int i = 0; i < HB; ++i) { SR_SGB(0x008, 1, 1); SR_SGB(0x100, 1, 1); }
            SR_SGB(0x008, HB * MB - HB, 1);
            __builtin_amdgcn_sched_barrier(0);
            load_w(buf, 1, 1, wy);
            SR_MFMA_HALF(0, wx, a1)
#pragma unroll
            for (int i = 0; i < HB; ++i) { SR_SGB(0x008, 1, 2); SR_SGB(0x100, 1, 2); }
            SR_SGB(0x008, HB * MB - HB, 2);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            load_w(buf ^ 1, 0, 0, wx);
            load_a(buf ^ 1, 0, a0);
            stage_pieces(buf);
            SR_MFMA_HALF(1, wy, a1)
#pragma unroll
            for (int i = 0; i < HB + MB; ++i) { SR_SGB(0x008, 1, 3); SR_SGB(0x100, 1, 3); }
#pragma unroll
            for (int i = 0; i < 8; ++i) { SR_SGB(0x008, 1, 3); SR_SGB(0x010, 1, 3); }
            __builtin_amdgcn_sched_barrier(0);
            buf ^= 1;
        }
        stage_wrap();
        }
#undef SR_SGB
        for (; kt < nkt; ++kt) {     // the last two k-steps of the workgroup's last tile: 4 phases per k-step, see gemm_bf16.hip
            load_w(buf, 0, 1, wy);
            SR_MFMA_HALF(0, wx, a0)
            load_w(buf, 1, 0, wx);
            load_a(buf, 1, a1);
            SR_MFMA_HALF(1, wy, a0)
            load_w(buf, 1, 1, wy);
            SR_MFMA_HALF(0, wx, a1)
            // The LDS-DMA of k-step kt + 1 was issued in the PREVIOUS iteration: hipcc does not see it as pending here and
            // emits no vmcnt wait for this barrier, so drain it by hand (every wave its own pieces, then the barrier).
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (kt + 1 < nkt) {             // (a workgroup's last tile: nothing left to stage)
                load_w(buf ^ 1, 0, 0, wx);
                load_a(buf ^ 1, 0, a0);
            }
            SR_MFMA_HALF(1, wy, a1)
            buf ^= 1;
        }
        const unsigned long long st1 = SR_SPLIT_STAMPS_PTR(a) ? __builtin_amdgcn_s_memrealtime() : 0;
        split_epilogue<UB, NB, MB>(a, acc, row0, q0, wn, wm, frow, fg, xy_s, qa_s, tau_s, gx, gy);
        if (SR_SPLIT_STAMPS_PTR(a)) {          // dev switch SR_SPLIT_STAMPS: 10 ns ticks per tile of wave 0: k-loop, epilogue issue, wait at the next tile's top
            const unsigned long long st2 = __builtin_amdgcn_s_memrealtime();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned long long st3 = __builtin_amdgcn_s_memrealtime();
            if (tid == 0) {
                atomicAdd(&a.stamps[0], st1 - st0);
                atomicAdd(&a.stamps[1], st2 - st1);
                atomicAdd(&a.stamps[2], st3 - st2);
                atomicAdd(&a.stamps[3], 1ull);
            }
        }
        if (!has_next) break;
        tile = tile_next;
        carried = true;
    }
#undef SR_MFMA_HALF
}

int launch_dense_split(const DenseSplitArgs& a, hipStream_t s) {
    const int64_t rows = a.row_end - a.row_begin;
    if (rows <= 0) return SR_OK;
    SR_REQUIRE(a.H % 64 == 0, "dense_split: dim %d must be a multiple of 64", a.H);
    SR_REQUIRE(a.n_pairs >= 1 && a.n_pairs <= 6 && a.n_pairs * (a.H / 64) >= 2, "dense_split: bad plane-pair count %d", a.n_pairs);
    SR_REQUIRE(!a.upper_bound || (a.n_pairs == 1 && a.dxy && a.qa), "dense_split: the upper-bound pass is one plane product");
    constexpr size_t lds = 2 * (size_t)(SP_BN + SP_BM) * 128 + SP_BN * 2 * sizeof(float) + SP_BM * 4 * sizeof(float) + SP_BM * sizeof(float);
    static DeviceOnce attr_once;
    bool* attr_slot = attr_once.pending();
    if (attr_slot) {
        SR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&dense_split_kernel<false>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        SR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&dense_split_kernel<true>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        *attr_slot = true;
    }
    DenseSplitArgs b = a;
    if (const char* e = sr_dev_getenv("SR_SPLIT_BLOCKTEST")) if (atoi(e) == 0) b.dxy_gmax = nullptr;     // A/B switch: every bound formed
    SR_REQUIRE(!b.dxy_gmax || a.row_begin % SP_BN == 0, "dense_split: the block test needs launches that start on a tile boundary");
    b.xcd_order = 1;
    if (const char* e = sr_dev_getenv("SR_SPLIT_XCD")) b.xcd_order = atoi(e);     // A/B switch
    b.diag = 0;
#ifdef SR_DIAG_BUILD        // wrong results by design: read only by a diagnostic build (make EXTRA=-DSR_DIAG_BUILD), never by the product .so
    if (const char* e = sr_dev_getenv("SR_SPLIT_DIAG")) b.diag = atoi(e);         // timing only
#endif
    b.stamps = nullptr;
#ifdef SR_DIAG_BUILD
    if (sr_dev_getenv("SR_SPLIT_STAMPS")) {
        static unsigned long long* d_st = nullptr;
        static int calls = 0;
        if (!d_st) { SR_CHECK_HIP(hipMalloc((void**)&d_st, 32)); SR_CHECK_HIP(hipMemset(d_st, 0, 32)); }
        if (++calls % 272 == 0) {
            unsigned long long h[4];
            SR_CHECK_HIP(hipMemcpy(h, d_st, 32, hipMemcpyDeviceToHost));
            if (h[3]) fprintf(stderr, "[split stamps] tiles %llu: k-loop %.2f us, epilogue issue %.2f us, drain wait %.2f us\n", h[3],
                              h[0] * 0.01 / h[3], h[1] * 0.01 / h[3], h[2] * 0.01 / h[3]);
            SR_CHECK_HIP(hipMemset(d_st, 0, 32));
        }
        b.stamps = d_st;
    }
#endif
    const SplitGrid sg = split_grid(rows, a.nq, b.xcd_order);
    b.grid_qt = sg.qt; b.grid_dt = sg.dt; b.grid_bq = sg.bq; b.grid_bd = sg.bd; b.grid_nbq = sg.nbq; b.grid_total = sg.total;
    // one persistent workgroup per CU (130 KB of LDS each); SR_SPLIT_PERSIST=0: one workgroup per tile slot (A/B)
    int wgs = sr_cu_count();
    wgs -= wgs % 8;
    if (wgs < 8) wgs = 8;
    if (const char* e = sr_dev_getenv("SR_SPLIT_PERSIST")) if (atoi(e) == 0) wgs = sg.total;
    const dim3 grid((unsigned)(sg.total < wgs ? sg.total : wgs));
    if (a.upper_bound) hipLaunchKernelGGL(dense_split_kernel<true>, grid, dim3(512), lds, s, b);
    else hipLaunchKernelGGL(dense_split_kernel<false>, grid, dim3(512), lds, s, b);
    SR_CHECK_LAUNCH();
    return SR_OK;
}
