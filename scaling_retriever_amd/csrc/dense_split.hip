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

// ---- scoring kernel ----------------------------------------------------------------------------------
#ifndef SP_ORDER
#define SP_ORDER 1
#endif
#define SP_BN 256   // docs per workgroup
#define SP_BM 256   // queries per workgroup
__global__ __launch_bounds__(512, 2) void dense_split_kernel(DenseSplitArgs a) {
    constexpr int NB = 8, MB = 4, WAVES_M = 4, HB = NB / 2;   // wave tile: 128 docs x 64 queries
    constexpr int W_BYTES = SP_BN * 128, STAGE_BYTES = (SP_BN + SP_BM) * 128;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave / WAVES_M, wm = wave % WAVES_M;
    // Workgroup order.  SP_ORDER 1 (default): query tile fastest - the ~256 resident workgroups are all the query tiles of
    // a handful of doc tiles, so a doc tile (3 MB of plane rows) is fetched from HBM once and then shared through L2 /
    // Infinity Cache by its 28 query tiles, and the query planes (84 MB at 6 980 queries) stay in the Infinity Cache.
    // SP_ORDER 0: doc tile fastest - every resident workgroup streams a doc tile of its own, each of which is read again
    // by every other query tile later (28 x 268 MB per launch).
#if SP_ORDER
    const int64_t row0 = a.row_begin + (int64_t)blockIdx.y * SP_BN;
    const int q0 = blockIdx.x * SP_BM;
#else
    const int64_t row0 = a.row_begin + (int64_t)blockIdx.x * SP_BN;
    const int q0 = blockIdx.y * SP_BM;
#endif
    const int H = a.H;

    const int srow = lane >> 3;
    const int schunk = (lane & 7) ^ (srow & 7);
    int64_t doff[4], qoff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int64_t rn = row0 + (wave * 4 + i) * 8 + srow;
        rn = rn < a.row_end ? rn : a.row_end - 1;
        doff[i] = rn * H + schunk * 8;
        int rq = q0 + (wave * 4 + i) * 8 + srow;
        rq = rq < a.nq ? rq : a.nq - 1;
        qoff[i] = (int64_t)rq * H + schunk * 8;
    }
    const int nk = H / 64;
    // k-tiles are staged strictly in order (0, 1, 2, ...), so the plane pair of the NEXT tile is tracked incrementally:
    // the plane pointers change once per H / 64 tiles.  (Looking them up per tile - a division, then two dependent
    // scalar loads from the kernel arguments - sat right behind the k-step barrier, in front of the LDS-DMA issue.)
    int st_pair = 0, st_k0 = 0;
    const unsigned short* st_d = a.D[a.pair_d[0]];
    const unsigned short* st_q = a.Q[a.pair_q[0]];
    auto stage = [&](int st, int /*kt: the next tile in order*/) {
        unsigned char* wbase = smem + st * STAGE_BYTES + (wave * 4) * 1024;
        unsigned char* abase = smem + st * STAGE_BYTES + W_BYTES + (wave * 4) * 1024;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((gbl_void_ptr)(st_d + doff[i] + st_k0), (lds_void_ptr)(wbase + i * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((gbl_void_ptr)(st_q + qoff[i] + st_k0), (lds_void_ptr)(abase + i * 1024), 16, 0, 0);
        st_k0 += 64;
        if (st_k0 == H) {
            st_k0 = 0;
            ++st_pair;
            if (st_pair < a.n_pairs) {
                st_d = a.D[a.pair_d[st_pair]];
                st_q = a.Q[a.pair_q[st_pair]];
            }
        }
    };

    f32x4 acc[NB][MB];
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int j = 0; j < MB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15, fg = lane >> 4;
    const int nkt = a.n_pairs * nk;      // >= 2 (checked at launch)
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
            acc[(HH) * HB + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(WF[i], AF[j], acc[(HH) * HB + i][j], 0, 0, 0);

    stage(0, 0);
    stage(1, 1);
    __syncthreads();
    int buf = 0;
    load_w(buf, 0, 0, wx);
    load_a(buf, 0, a0);
    int kt = 0;
    // steady state with the issue order pinned (see gemm_bf16.hip): reads and LDS-DMA pieces dealt out one per MFMA
#define SR_SGB(MASK, N, ID) __builtin_amdgcn_sched_group_barrier(MASK, N, ID)
    for (; kt + 2 < nkt; ++kt) {
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
        stage(buf, kt + 2);
        SR_MFMA_HALF(1, wy, a1)
#pragma unroll
        for (int i = 0; i < HB + MB; ++i) { SR_SGB(0x008, 1, 3); SR_SGB(0x100, 1, 3); }
#pragma unroll
        for (int i = 0; i < 8; ++i) { SR_SGB(0x008, 1, 3); SR_SGB(0x010, 1, 3); }
        __builtin_amdgcn_sched_barrier(0);
        buf ^= 1;
    }
#undef SR_SGB
    for (; kt < nkt; ++kt) {     // last two k-steps: 4 phases per k-step, see gemm_bf16.hip
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
        if (kt + 2 < nkt) stage(buf, kt + 2);
        if (kt + 1 < nkt) {
            load_w(buf ^ 1, 0, 0, wx);
            load_a(buf ^ 1, 0, a0);
        }
        SR_MFMA_HALF(1, wy, a1)
        buf ^= 1;
    }
#undef SR_MFMA_HALF

    // ---- epilogue: lane = query (frow + 16 j), registers = docs (16 i + 4 fg + r) --------------------
    const int64_t left = a.row_end - row0;
    const int rows_valid = left < SP_BN ? (int)left : SP_BN;
    const uint32_t gid0 = a.id_base + (uint32_t)row0 * a.id_stride;
#pragma unroll
    for (int j = 0; j < MB; ++j) {
        const int q = q0 + wm * MB * 16 + j * 16 + frow;
        if (q >= a.nq) continue;
        const float tq = a.tau[q];
        int cnt = 0;
#pragma unroll
        for (int i = 0; i < NB; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int lr = wn * NB * 16 + i * 16 + fg * 4 + r;
                cnt += (lr < rows_valid && acc[i][j][r] >= tq) ? 1 : 0;
            }
        if (cnt == 0) continue;
        int pos = atomicAdd(&a.cand_count[q], cnt);
        uint64_t* dst = a.cand_keys + (int64_t)q * a.cand_cap;
#pragma unroll
        for (int i = 0; i < NB; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int lr = wn * NB * 16 + i * 16 + fg * 4 + r;
                const float sc = acc[i][j][r];
                if (lr < rows_valid && sc >= tq) {
                    if (pos < a.cand_cap) dst[pos] = sr_make_key(sc, gid0 + (uint32_t)lr * a.id_stride);
                    ++pos;
                }
            }
    }
}

int launch_dense_split(const DenseSplitArgs& a, hipStream_t s) {
    const int64_t rows = a.row_end - a.row_begin;
    if (rows <= 0) return SR_OK;
    SR_REQUIRE(a.H % 64 == 0, "dense_split: dim %d must be a multiple of 64", a.H);
    SR_REQUIRE(a.n_pairs >= 1 && a.n_pairs <= 6 && a.n_pairs * (a.H / 64) >= 2, "dense_split: bad plane-pair count %d", a.n_pairs);
    constexpr size_t lds = 2 * (size_t)(SP_BN + SP_BM) * 128;
    static DeviceOnce attr_once;
    bool* attr_slot = attr_once.pending();
    if (attr_slot) {
        SR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&dense_split_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        *attr_slot = true;
    }
#if SP_ORDER
    const dim3 grid((unsigned)ceil_div64(a.nq, SP_BM), (unsigned)ceil_div64(rows, SP_BN));
#else
    const dim3 grid((unsigned)ceil_div64(rows, SP_BN), (unsigned)ceil_div64(a.nq, SP_BM));
#endif
    SR_REQUIRE(grid.y <= 65535, "dense_split: launch of %lld rows x %d queries exceeds the grid", (long long)rows, a.nq);
    hipLaunchKernelGGL(dense_split_kernel, grid, dim3(512), lds, s, a);
    SR_CHECK_LAUNCH();
    return SR_OK;
}
