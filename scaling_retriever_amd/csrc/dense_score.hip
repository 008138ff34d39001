// Dense brute-force scoring Q . D^T with fused top-k filtering (gfx950).
//
// Replaces faiss.IndexFlatIP.search as used by DenseFlatIndexer.search_knn
// (scaling_retriever/indexer.py:191-217).  Exact fp32: products are accumulated
// by v_mfma_f32_32x32x2_f32 (f32 in / f32 acc), which is bit-for-bit a k-ordered
// fmaf chain.  Per 8-wide k group s the chain visits k = 8s+j (lane half 0) then
// 8s+4+j (half 1) for j = 0..3 - oracle/scoring.py::mfma_korder restates it.
//
// Layout: D row-major [N, H] fp32 in HBM (72.4 GB at N = 8 841 823, H = 2048),
// Q row-major [nq, H].  Docs are the MFMA A rows, queries the B columns, so in
// the accumulator a lane owns ONE query column (lane & 31) and 16 doc rows per
// 32x32 block: the tau compare is lane-local.  Scores never go to HBM; survivors
// (score >= tau[q]) are appended as 64-bit keys to the per-query candidate buffer
// (topk.hip).  A doc chunk = one launch; tau is raised between chunks.
#include "common.h"
#include "dense_stream.h"
#include "dense_split.h"
#include "dense_filter.h"
#include <mutex>
#include <stdlib.h>
#include <vector>

struct DenseArgs {
    const float* D;       // segment base
    const float* Q;
    int64_t row_begin;    // first doc row of this launch (within the segment)
    int64_t row_end;      // one past the last doc row of this launch
    int H;
    int nq;
    const float* tau;
    uint64_t* cand_keys;
    int* cand_count;
    int64_t cand_cap;
    uint32_t id_base, id_stride;
};

// Tile = (32*WM*WAVES_M docs) x (32*WN*WAVES_N queries), 4 or 8 waves, BK = 16.
// 8 waves = 2 per SIMD: while one wave sits at the k-step barrier or waits for its LDS fragments,
// its SIMD partner keeps the (64-cycle) fp32 MFMA pipe busy.
template <int WAVES_M, int WAVES_N, int WM, int WN, int BK = 16, int OCC = (WAVES_M * WAVES_N) / 4>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N, OCC) void dense_score_kernel(DenseArgs a) {
    static_assert(WAVES_M * WAVES_N == 4 || WAVES_M * WAVES_N == 8 || WAVES_M * WAVES_N == 16, "4, 8 or 16 waves per workgroup");
    constexpr int NT = 64 * WAVES_M * WAVES_N;
    constexpr int TM = 32 * WM * WAVES_M, TN = 32 * WN * WAVES_N, LDK = BK + 4, KC = BK / 4;
    constexpr int A_CHUNKS = TM * (BK / 4), B_CHUNKS = TN * (BK / 4);
    constexpr int A_PER_T = (A_CHUNKS + NT - 1) / NT, B_PER_T = (B_CHUNKS + NT - 1) / NT;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                  // [2][TM][LDK]
    float* Bs = smem + 2 * TM * LDK;   // [2][TN][LDK]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int64_t row0 = a.row_begin + (int64_t)blockIdx.x * TM;
    const int q0 = blockIdx.y * TN;
    const int H = a.H;

    f32x4 ra[A_PER_T], rb[B_PER_T];
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < A_PER_T; ++i) {
            const int c = tid + i * NT;
            const int r = c / KC, kc = c % KC;
            const int64_t row = row0 + r;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (c < A_CHUNKS && row < a.row_end) v = *reinterpret_cast<const f32x4*>(a.D + row * H + k0 + kc * 4);
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < B_PER_T; ++i) {
            const int c = tid + i * NT;
            const int r = c / KC, kc = c % KC;
            const int q = q0 + r;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (c < B_CHUNKS && q < a.nq) v = *reinterpret_cast<const f32x4*>(a.Q + (int64_t)q * H + k0 + kc * 4);
            rb[i] = v;
        }
    };
    auto sstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < A_PER_T; ++i) {
            const int c = tid + i * NT;
            if (c < A_CHUNKS) *reinterpret_cast<f32x4*>(&As[(buf * TM + (c / KC)) * LDK + (c % KC) * 4]) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < B_PER_T; ++i) {
            const int c = tid + i * NT;
            if (c < B_CHUNKS) *reinterpret_cast<f32x4*>(&Bs[(buf * TN + (c / KC)) * LDK + (c % KC) * 4]) = rb[i];
        }
    };

    f32x16 acc[WM][WN];
#pragma unroll
    for (int m = 0; m < WM; ++m)
#pragma unroll
        for (int n = 0; n < WN; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    const int nk = H / BK;
    gload(0);
    sstore(0);
    __syncthreads();
    const int arow = wm * WM * 32 + (lane & 31);
    const int brow = wn * WN * 32 + (lane & 31);
    const int koff = 4 * (lane >> 5);
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) gload((kt + 1) * BK);
#pragma unroll
        for (int s = 0; s < BK / 8; ++s) {
            f32x4 af[WM], bf[WN];
#pragma unroll
            for (int m = 0; m < WM; ++m)
                af[m] = *reinterpret_cast<const f32x4*>(&As[(buf * TM + arow + m * 32) * LDK + 8 * s + koff]);
#pragma unroll
            for (int n = 0; n < WN; ++n)
                bf[n] = *reinterpret_cast<const f32x4*>(&Bs[(buf * TN + brow + n * 32) * LDK + 8 * s + koff]);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int m = 0; m < WM; ++m)
#pragma unroll
                    for (int n = 0; n < WN; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[m][j], bf[n][j], acc[m][n], 0, 0, 0);
        }
        if (kt + 1 < nk) sstore(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: lane-local tau filter, append survivors -------------------
    const int half = lane >> 5;
    const int64_t left = a.row_end - row0;
    const int rows_valid = left < TM ? (int)left : TM;            // doc rows of this tile that exist
    const int lr0 = wm * WM * 32 + 4 * half;                      // local row of register 0 in block m = 0
    const uint32_t gid0 = a.id_base + (uint32_t)row0 * a.id_stride;
#pragma unroll
    for (int n = 0; n < WN; ++n) {
        const int q = q0 + wn * WN * 32 + n * 32 + (lane & 31);
        if (q >= a.nq) continue;
        const float tq = a.tau[q];
        int cnt = 0;
#pragma unroll
        for (int m = 0; m < WM; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int lr = lr0 + m * 32 + (r & 3) + 8 * (r >> 2);
                cnt += (lr < rows_valid && acc[m][n][r] >= tq) ? 1 : 0;
            }
        if (cnt == 0) continue;
        int pos = atomicAdd(&a.cand_count[q], cnt);
        uint64_t* dst = a.cand_keys + (int64_t)q * a.cand_cap;
#pragma unroll
        for (int m = 0; m < WM; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int lr = lr0 + m * 32 + (r & 3) + 8 * (r >> 2);
                const float sc = acc[m][n][r];
                if (lr < rows_valid && sc >= tq) {
                    if (pos < a.cand_cap) dst[pos] = sr_make_key(sc, gid0 + (uint32_t)lr * a.id_stride);
                    ++pos;
                }
            }
    }
}

// ---- the same tile (256 docs x 256 queries, 8 waves, wave tile 128 x 64), software-pipelined ---------------------
// In dense_score_kernel every k-step ends  MFMAs -> ds_write -> barrier -> ds_read -> MFMAs: both waves of a SIMD belong
// to the same workgroup, reach the barrier together and then wait for their first fragments together, so the MFMA pipe
// drains once per k-step (rocprofv3 PMC: waves parked 14 % of their cycles, MFMA busy 84 %).  Here three LDS stages make
// the hand-offs a whole k-step old: during step kt the registers fetched two steps ahead are written to stage (kt + 2) % 3
// (last read in step kt - 1), and the first fragments of step kt + 1 are read - from a stage made visible by the
// PREVIOUS barrier - under the last MFMAs of step kt.  One barrier per k-step remains, with MFMAs issued right up to
// it and right after it.  Same k order per accumulator as dense_score_kernel: bit-identical scores.
// WN = 32-query blocks per wave: 2 -> 256 docs x 256 queries (wave tile 128 x 64), 1 -> 256 x 128 for 65-128 queries.
template <int WN>
__global__ __launch_bounds__(512, 2) void dense_score_pipe_kernel(DenseArgs a) {
    constexpr int WAVES_N = 4, WM = 4, BK = 16;
    constexpr int NT = 512, TM = 256, TN = 32 * WN * WAVES_N, LDK = BK + 4, KC = BK / 4, NSTAGE = 3;
    constexpr int PER_T = TM * KC / NT;                 // 16-B chunks per thread of the doc tile (= 2)
    constexpr int PER_TB = TN * KC / NT;                // ... of the query tile (2 or 1)
    constexpr int HALF_MFMAS = 4 * WM * WN;             // MFMAs per half k-step and wave
    constexpr int MPO = WN == 2 ? 2 : 1;                // MFMAs between two memory operations
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                       // [NSTAGE][TM][LDK]
    float* Bs = smem + NSTAGE * TM * LDK;   // [NSTAGE][TN][LDK]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int64_t row0 = a.row_begin + (int64_t)blockIdx.x * TM;
    const int q0 = blockIdx.y * TN;
    const int H = a.H;

    // staging: thread t moves chunks t and t + 512 of each operand tile (row = chunk / 4, k-chunk = chunk % 4);
    // rows past the end are clamped (their scores are never emitted)
    const float* asrc[PER_T];
    const float* bsrc[PER_TB];
    int soff[PER_T];
#pragma unroll
    for (int i = 0; i < PER_T; ++i) {
        // chunk c -> (row r, k-chunk kc): the 8 lanes of a ds_write_b128 group take rows r and r + 4, whose 80-byte pitch
        // puts them 16 banks apart (rows r and r + 1 would overlap: 2-way conflicts on every stage write); a row is still
        // fetched as one 64-byte piece by 4 neighbouring lanes
        const int c = tid + i * NT, kc = c & 3, grp = c >> 3;
        const int r = (grp >> 2) * 8 + (grp & 3) + 4 * ((c >> 2) & 1);
        int64_t row = row0 + r;
        row = row < a.row_end ? row : a.row_end - 1;
        asrc[i] = a.D + row * H + kc * 4;
        soff[i] = r * LDK + kc * 4;
        if (i < PER_TB) {
            int q = q0 + r;
            q = q < a.nq ? q : a.nq - 1;
            bsrc[i] = a.Q + (int64_t)q * H + kc * 4;
        }
    }
    f32x4 ra[PER_T], rb[PER_TB];
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < PER_T; ++i) ra[i] = *reinterpret_cast<const f32x4*>(asrc[i] + k0);
#pragma unroll
        for (int i = 0; i < PER_TB; ++i) rb[i] = *reinterpret_cast<const f32x4*>(bsrc[i] + k0);
    };
    auto sstore = [&](int st) {
#pragma unroll
        for (int i = 0; i < PER_T; ++i) *reinterpret_cast<f32x4*>(&As[st * TM * LDK + soff[i]]) = ra[i];
#pragma unroll
        for (int i = 0; i < PER_TB; ++i) *reinterpret_cast<f32x4*>(&Bs[st * TN * LDK + soff[i]]) = rb[i];
    };

    f32x16 acc[WM][WN];
#pragma unroll
    for (int m = 0; m < WM; ++m)
#pragma unroll
        for (int n = 0; n < WN; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    const int aoff = (wm * WM * 32 + (lane & 31)) * LDK + 4 * (lane >> 5);
    const int boff = (wn * WN * 32 + (lane & 31)) * LDK + 4 * (lane >> 5);
    auto frag = [&](int st, int sub, f32x4 (&af)[WM], f32x4 (&bf)[WN]) {
#pragma unroll
        for (int m = 0; m < WM; ++m) af[m] = *reinterpret_cast<const f32x4*>(&As[st * TM * LDK + aoff + m * 32 * LDK + 8 * sub]);
#pragma unroll
        for (int n = 0; n < WN; ++n) bf[n] = *reinterpret_cast<const f32x4*>(&Bs[st * TN * LDK + boff + n * 32 * LDK + 8 * sub]);
    };
#define SR_DENSE_MFMAS(AF, BF)                                                                                   \
    _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                                \
        _Pragma("unroll") for (int m = 0; m < WM; ++m)                                                           \
            _Pragma("unroll") for (int n = 0; n < WN; ++n)                                                       \
                acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(AF[m][j], BF[n][j], acc[m][n], 0, 0, 0);

    const int nk = H / BK;
    {   // prologue: the first three k-steps' loads are all in flight before the first stage is written (one memory
        // latency instead of three)
        f32x4 pa[2][PER_T], pb[2][PER_TB];
#pragma unroll
        for (int st0 = 0; st0 < 2; ++st0) {
            const int k0 = (st0 < nk ? st0 : 0) * BK;
#pragma unroll
            for (int i = 0; i < PER_T; ++i) pa[st0][i] = *reinterpret_cast<const f32x4*>(asrc[i] + k0);
#pragma unroll
            for (int i = 0; i < PER_TB; ++i) pb[st0][i] = *reinterpret_cast<const f32x4*>(bsrc[i] + k0);
        }
        gload(nk > 2 ? 2 * BK : 0);
#pragma unroll
        for (int st0 = 0; st0 < 2; ++st0) {
#pragma unroll
            for (int i = 0; i < PER_T; ++i) *reinterpret_cast<f32x4*>(&As[st0 * TM * LDK + soff[i]]) = pa[st0][i];
#pragma unroll
            for (int i = 0; i < PER_TB; ++i) *reinterpret_cast<f32x4*>(&Bs[st0 * TN * LDK + soff[i]]) = pb[st0][i];
        }
    }
    __syncthreads();
    f32x4 a0[WM], b0[WN], a1[WM], b1[WN];
    frag(0, 0, a0, b0);
    int st = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const int st1 = st == NSTAGE - 1 ? 0 : st + 1;
        const int st2 = st1 == NSTAGE - 1 ? 0 : st1 + 1;
        // branch-free body (a branch makes hipcc wait for ALL outstanding LDS reads at the join): past the last k-steps
        // the extra stage writes / fragment reads / re-loads of the last k-block touch valid memory and are never used
        // Issue order inside a half (sched_group_barrier): the half's MFMAs start at once and its LDS / global operations are
        // dealt out between them, two MFMAs (128 cycles of pipe time) apart, so nothing queues up in front of the MFMA pipe
        // after the barrier.  First half: fragment reads of the second half, stage write of k-step kt + 2, global loads of
        // k-step kt + 3 (a whole k-step ahead of the write that consumes them).  Second half: first fragments of k-step kt + 1.
        frag(st, 1, a1, b1);
        sstore(st2);
        gload(kt + 3 < nk ? (kt + 3) * BK : H - BK);
        SR_DENSE_MFMAS(a0, b0)
#pragma unroll
        for (int i = 0; i < WM + WN; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, MPO, 0);  // MFMAs
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);    // 1 DS read
        }
#pragma unroll
        for (int i = 0; i < PER_T + PER_TB; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, MPO, 0);
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);    // 1 DS write
        }
#pragma unroll
        for (int i = 0; i < PER_T + PER_TB; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, MPO, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);    // 1 VMEM read
        }
        __builtin_amdgcn_sched_group_barrier(0x008, HALF_MFMAS - MPO * (WM + WN + 2 * (PER_T + PER_TB)), 0);
        __builtin_amdgcn_sched_barrier(0);
        frag(st1, 0, a0, b0);
        SR_DENSE_MFMAS(a1, b1)
#pragma unroll
        for (int i = 0; i < WM + WN; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, MPO, 1);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 1);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, HALF_MFMAS - MPO * (WM + WN), 1);
        __builtin_amdgcn_sched_barrier(0);      // the barrier (and its lgkmcnt(0)) after the MFMAs that cover the reads
        __syncthreads();
        st = st1;
    }
#undef SR_DENSE_MFMAS

    // ---- epilogue: lane-local tau filter, append survivors (as dense_score_kernel) -------------------
    const int half = lane >> 5;
    const int64_t left = a.row_end - row0;
    const int rows_valid = left < TM ? (int)left : TM;
    const int lr0 = wm * WM * 32 + 4 * half;
    const uint32_t gid0 = a.id_base + (uint32_t)row0 * a.id_stride;
#pragma unroll
    for (int n = 0; n < WN; ++n) {
        const int q = q0 + wn * WN * 32 + n * 32 + (lane & 31);
        if (q >= a.nq) continue;
        const float tq = a.tau[q];
        int cnt = 0;
#pragma unroll
        for (int m = 0; m < WM; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int lr = lr0 + m * 32 + (r & 3) + 8 * (r >> 2);
                cnt += (lr < rows_valid && acc[m][n][r] >= tq) ? 1 : 0;
            }
        if (cnt == 0) continue;
        int pos = atomicAdd(&a.cand_count[q], cnt);
        uint64_t* dst = a.cand_keys + (int64_t)q * a.cand_cap;
#pragma unroll
        for (int m = 0; m < WM; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int lr = lr0 + m * 32 + (r & 3) + 8 * (r >> 2);
                const float sc = acc[m][n][r];
                if (lr < rows_valid && sc >= tq) {
                    if (pos < a.cand_cap) dst[pos] = sr_make_key(sc, gid0 + (uint32_t)lr * a.id_stride);
                    ++pos;
                }
            }
    }
}

template <int WN>
static int launch_dense_pipe(const DenseArgs& a, int64_t rows, hipStream_t s) {
    constexpr int TN = 128 * WN;
    constexpr size_t lds = sizeof(float) * 3 * (256 + TN) * (16 + 4);
    static DeviceOnce attr_once;
    bool* attr_slot = attr_once.pending();
    if (attr_slot) {
        SR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&dense_score_pipe_kernel<WN>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        *attr_slot = true;
    }
    dim3 grid((unsigned)ceil_div64(rows, 256), (unsigned)ceil_div64(a.nq, TN));
    hipLaunchKernelGGL(dense_score_pipe_kernel<WN>, grid, dim3(512), lds, s, a);
    SR_CHECK_LAUNCH();
    return SR_OK;
}

// --------------------------------------------------------------------- host ---
struct DenseSegment {
    const float* rows;
    int64_t n;
    int64_t id_base, id_stride;
    unsigned short* pl[3] = {nullptr, nullptr, nullptr};   // library-owned bf16 planes (bf16x3 / bf16x6 score modes only)
    int n_planes = 0;
    // certified filter (SR_PRECISION_FP32_FILTERED, dense_filter.hip): fp16 plane of the rows scaled by fsd (a power of two)
    // and the per-document error terms (x, y); f_state: 0 = not built, 1 = ready, -1 = cannot be filtered (non-finite or
    // out-of-range values, no memory)
    unsigned short* fpl = nullptr;
    float* fxy = nullptr;               // [n, 2], then [ceil(n / 128), 2]: the maxima of x and of y over every group of 128 documents
    float fsd = 1.f, fisd = 1.f;
    float fx_max = 0.f, fy_max = 0.f;   // the largest x and y of the segment (scaled domain): the second threshold's e_max (dense_filter.h)
    int f_state = 0;
};

struct sr_dense_index {
    int dim = 0;
    std::vector<DenseSegment> segs;
    int64_t ntotal = 0;
    int64_t ws_limit = 4ll << 30;
    int precision = SR_PRECISION_FP32;
    bool batch_invariant = false;          // sr_dense_index_set_batch_invariant: batches <= 64 run the tiled kernels (one k order)
    unsigned short* qpl[3] = {nullptr, nullptr, nullptr};   // query planes for the split precisions
    int64_t q_cap = 0;
    TopkWS ws;
    StreamOrder order;
    LaunchProfile prof;
    std::mutex mu;
    // SR_PRECISION_FP32_FILTERED (dense_filter.hip)
    TopkWS wsf;                       // the 16-bit passes (upper-bound pass, bf16x3 / bf16x6): segmented candidate slots - a workspace of
                                      // its own, so an index that alternates pass kinds does not free and re-allocate GBs per search
    TopkWS ws2;                       // exact top-k over the re-scored candidates
    TopkWS ws3;                       // exact re-do of the queries without a certificate
    float* qa = nullptr;              // [fq_cap, 4] per query (A', B', sq, 1 / sq), then 3 x [fq_cap]: slack, tau2, tau_eff (dense_filter.h)
    float* a_scores = nullptr;        // [fq_cap, fkp] the kp largest upper bounds U, sorted
    int64_t* a_ids = nullptr;
    int* flags = nullptr;             // [2 fq_cap + 2]: per-query flags, then the xmin scratch of the re-score
    int64_t fq_cap = 0;
    int fkp = 0;
    unsigned int* f_scratch = nullptr;    // [2]: absmax bits, bad flag (segment preparation)
    // queries the certificate could not be given for are re-done by the exact kernel, those alone
    float* redo_q = nullptr; int64_t* redo_idx = nullptr; float* redo_scores = nullptr; int64_t* redo_ids = nullptr;
    int64_t redo_cap = 0; int redo_k = 0;
    int64_t n_filtered = 0, n_fallback = 0;   // searches answered by the filter alone / with queries (or all) redone by the exact kernel
    int64_t nq_certified = 0, nq_redone = 0;  // queries answered by the filter / re-done by the exact kernel
    int64_t pend_nq = 0; int pend_k = 0; const float* pend_q = nullptr;   // sr_dense_search_begin ran for this batch
};

// bf16 planes of every segment a score mode needs (the certified filter keeps its own fp16 plane, filter_prepare_segment)
static int planes_of(int precision) {
    return precision == SR_PRECISION_BF16X6 ? 3 : (precision == SR_PRECISION_BF16X3 ? 2 : 0);
}
#define SR_PASS_FILTER 101   // dense_search_pass: the filter's upper-bound pass (one fp16 plane product + the per-pair error term)

// fp16 plane + per-document error terms of one segment.  Never fails the caller: a segment that cannot be filtered (values
// that are not finite or beyond 2^55, no device memory for the plane) is marked, and the index then answers with the exact
// kernel - the filter is an accelerator of the exact search, not a precondition.
static int filter_prepare_segment(sr_dense_index* idx, DenseSegment& seg) {
    if (seg.f_state != 0) return SR_OK;
    seg.f_state = -1;
    if (idx->dim % 64 != 0) return SR_OK;
    if (!idx->f_scratch && hipMalloc((void**)&idx->f_scratch, 8) != hipSuccess) { (void)hipGetLastError(); idx->f_scratch = nullptr; return SR_OK; }
    unsigned int h[2] = {0, 0};
    SR_CHECK_HIP(hipMemsetAsync(idx->f_scratch, 0, 8, nullptr));
    SR_TRY(launch_filter_absmax(seg.rows, seg.n, idx->dim, idx->f_scratch, nullptr));
    SR_CHECK_HIP(hipMemcpy(h, idx->f_scratch, 8, hipMemcpyDeviceToHost));
    float absmax;
    memcpy(&absmax, &h[0], 4);
    if (h[0] >= 0x7f800000u || !sr_filter_scale_of(absmax, &seg.fsd, &seg.fisd)) return SR_OK;
    const size_t bytes = (size_t)seg.n * (size_t)idx->dim * 2;
    if (hipMalloc((void**)&seg.fpl, bytes) != hipSuccess || hipMalloc((void**)&seg.fxy, (size_t)(seg.n + ceil_div64(seg.n, 128)) * 8) != hipSuccess) {
        (void)hipGetLastError();
        if (seg.fpl) (void)hipFree(seg.fpl);
        seg.fpl = nullptr; seg.fxy = nullptr;
        return SR_OK;
    }
    SR_TRY(launch_filter_plane(seg.rows, seg.n, idx->dim, seg.fsd, sr_filter_sigma(idx->dim), seg.fpl, seg.fxy,
                               reinterpret_cast<int*>(idx->f_scratch) + 1, nullptr));
    SR_TRY(launch_filter_group_max(seg.fxy, seg.n, seg.fxy + 2 * seg.n, nullptr));
    SR_CHECK_HIP(hipMemcpy(h, idx->f_scratch, 8, hipMemcpyDeviceToHost));
    {
        std::vector<float> gm((size_t)ceil_div64(seg.n, 128) * 2);
        SR_CHECK_HIP(hipMemcpy(gm.data(), seg.fxy + 2 * seg.n, gm.size() * 4, hipMemcpyDeviceToHost));
        seg.fx_max = seg.fy_max = 0.f;
        for (size_t g = 0; g < gm.size(); g += 2) {
            seg.fx_max = gm[g] > seg.fx_max ? gm[g] : seg.fx_max;
            seg.fy_max = gm[g + 1] > seg.fy_max ? gm[g + 1] : seg.fy_max;
        }
    }
    if (h[1] != 0) {                          // a non-finite error term: the segment is not filterable, its plane is of no use
        (void)hipFree(seg.fpl); (void)hipFree(seg.fxy);
        seg.fpl = nullptr; seg.fxy = nullptr;
        return SR_OK;
    }
    seg.f_state = 1;
    return SR_OK;
}

static int split_segment(sr_dense_index* idx, DenseSegment& seg, int want) {
    if (seg.n_planes >= want) return SR_OK;
    const size_t bytes = (size_t)seg.n * (size_t)idx->dim * 2;
    for (int p = seg.n_planes; p < want; ++p) {
        if (hipMalloc((void**)&seg.pl[p], bytes) != hipSuccess) {
            (void)hipGetLastError();
            seg.pl[p] = nullptr;
            for (int r = seg.n_planes; r < p; ++r) { (void)hipFree(seg.pl[r]); seg.pl[r] = nullptr; }
            sr_set_error("split precision needs %zu more bytes of device memory per plane of this segment", bytes);
            return SR_ERR_NOMEM;
        }
    }
    // (re)compute all planes: cheap next to one search, and keeps the planes consistent
    SR_TRY(launch_split_bf16(seg.rows, seg.pl[0], want >= 2 ? seg.pl[1] : nullptr, want == 3 ? seg.pl[2] : nullptr, seg.n * (int64_t)idx->dim, nullptr));
    SR_CHECK_HIP(hipStreamSynchronize(nullptr));
    seg.n_planes = want;
    return SR_OK;
}

template <int WAVES_M, int WAVES_N, int WM, int WN, int BK = 16, int OCC = (WAVES_M * WAVES_N) / 4>
static int launch_dense(const DenseArgs& a, int64_t rows, hipStream_t s) {
    constexpr int TM = 32 * WM * WAVES_M, TN = 32 * WN * WAVES_N;
    constexpr size_t lds = sizeof(float) * 2 * (TM + TN) * (BK + 4);
    static DeviceOnce attr_once;
    bool* attr_slot = attr_once.pending();
    if (attr_slot) {
        SR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&dense_score_kernel<WAVES_M, WAVES_N, WM, WN, BK, OCC>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        *attr_slot = true;
    }
    dim3 grid((unsigned)ceil_div64(rows, TM), (unsigned)ceil_div64(a.nq, TN));
    hipLaunchKernelGGL((dense_score_kernel<WAVES_M, WAVES_N, WM, WN, BK, OCC>), grid, dim3(64 * WAVES_M * WAVES_N), lds, s, a);
    SR_CHECK_LAUNCH();
    return SR_OK;
}

extern "C" int sr_dense_index_create(sr_dense_index** out, int dim) {
    SR_REQUIRE(out, "sr_dense_index_create: null out");
    SR_REQUIRE(dim > 0 && dim % 16 == 0, "sr_dense_index_create: dim=%d must be a positive multiple of 16", dim);
    *out = new sr_dense_index();
    (*out)->dim = dim;
    return SR_OK;
}

extern "C" int sr_dense_index_add(sr_dense_index* idx, const float* d_rows, int64_t n_rows, int64_t id_base,
                                  int64_t id_stride) {
    SR_REQUIRE(idx, "sr_dense_index_add: null index");
    SR_REQUIRE(n_rows >= 0 && id_stride >= 1 && id_base >= 0, "sr_dense_index_add: bad sizes");
    if (n_rows == 0) return SR_OK;
    SR_REQUIRE(d_rows, "sr_dense_index_add: null rows");
    SR_REQUIRE(((uintptr_t)d_rows & 15) == 0, "sr_dense_index_add: rows must be 16-byte aligned");
    SR_REQUIRE(id_base + (n_rows - 1) * id_stride < 0xffffffffll, "sr_dense_index_add: global doc index exceeds 32 bits");
    std::lock_guard<std::mutex> lock(idx->mu);
    DenseSegment seg;
    seg.rows = d_rows; seg.n = n_rows; seg.id_base = id_base; seg.id_stride = id_stride;
    if (planes_of(idx->precision)) SR_TRY(split_segment(idx, seg, planes_of(idx->precision)));
    if (idx->precision == SR_PRECISION_FP32_FILTERED) SR_TRY(filter_prepare_segment(idx, seg));
    idx->segs.push_back(seg);
    idx->ntotal += n_rows;
    return SR_OK;
}

extern "C" int64_t sr_dense_index_ntotal(const sr_dense_index* idx) { return idx ? idx->ntotal : -1; }

extern "C" int sr_dense_index_set_workspace_limit(sr_dense_index* idx, int64_t bytes) {
    SR_REQUIRE(idx && bytes >= (1 << 20), "sr_dense_index_set_workspace_limit: bad argument");
    idx->ws_limit = bytes;
    return SR_OK;
}

extern "C" int sr_dense_index_set_batch_invariant(sr_dense_index* idx, int on) {
    SR_REQUIRE(idx, "sr_dense_index_set_batch_invariant: null index");
    std::lock_guard<std::mutex> lock(idx->mu);
    idx->batch_invariant = on != 0;
    return SR_OK;
}

extern "C" int sr_dense_index_destroy(sr_dense_index* idx) {
    if (!idx) return SR_OK;
    idx->ws.release();
    idx->order.release();
    for (DenseSegment& seg : idx->segs) {
        for (int p = 0; p < 3; ++p)
            if (seg.pl[p]) (void)hipFree(seg.pl[p]);
        if (seg.fpl) (void)hipFree(seg.fpl);
        if (seg.fxy) (void)hipFree(seg.fxy);
    }
    for (int p = 0; p < 3; ++p)
        if (idx->qpl[p]) (void)hipFree(idx->qpl[p]);
    idx->wsf.release();
    idx->ws2.release();
    idx->ws3.release();
    if (idx->qa) (void)hipFree(idx->qa);
    if (idx->f_scratch) (void)hipFree(idx->f_scratch);
    if (idx->redo_q) (void)hipFree(idx->redo_q);
    if (idx->redo_idx) (void)hipFree(idx->redo_idx);
    if (idx->redo_scores) (void)hipFree(idx->redo_scores);
    if (idx->redo_ids) (void)hipFree(idx->redo_ids);
    if (idx->a_scores) (void)hipFree(idx->a_scores);
    if (idx->a_ids) (void)hipFree(idx->a_ids);
    if (idx->flags) (void)hipFree(idx->flags);
    delete idx;
    return SR_OK;
}

extern "C" int sr_dense_index_set_precision(sr_dense_index* idx, int mode) {
    SR_REQUIRE(idx, "sr_dense_index_set_precision: null index");
    SR_REQUIRE(mode == SR_PRECISION_FP32 || mode == SR_PRECISION_BF16X3 || mode == SR_PRECISION_BF16X6 ||
                   mode == SR_PRECISION_FP32_FILTERED,
               "sr_dense_index_set_precision: unknown mode %d", mode);
    std::lock_guard<std::mutex> lock(idx->mu);
    if (planes_of(mode)) {
        SR_REQUIRE(idx->dim % 64 == 0, "split precisions need dim %% 64 == 0 (dim = %d)", idx->dim);
        for (DenseSegment& seg : idx->segs) SR_TRY(split_segment(idx, seg, planes_of(mode)));
    }
    if (mode == SR_PRECISION_FP32_FILTERED)
        for (DenseSegment& seg : idx->segs) SR_TRY(filter_prepare_segment(idx, seg));
    idx->precision = mode;
    return SR_OK;
}

// one pass in the given arithmetic (SR_PRECISION_FP32 | _BF16X3 | _BF16X6); caller holds idx->mu
static int dense_search_pass(sr_dense_index* idx, const float* d_queries, int64_t nq, int k, float* d_out_scores,
                             int64_t* d_out_ids, int precision, hipStream_t s, bool force_tiled = false, int k_inner = 0) {
    const bool pass16 = (planes_of(precision) || precision == SR_PASS_FILTER) && nq > 64;
    // the re-do of a few queries and the 16-bit passes keep their own (differently shaped) workspaces
    TopkWS& ws = force_tiled ? idx->ws3 : (pass16 ? idx->wsf : idx->ws);

    // tile config by query count; chunk = docs per launch (= candidate capacity per query)
    int cfg;
    int TN;
    const char* env_variant = sr_dev_getenv("SR_DENSE_VARIANT");   // A/B switch, read per call: 5 = pipelined kernel (default), 1 = plain double buffer
    const int variant = env_variant ? atoi(env_variant) : 5;
    if (nq > 128) { cfg = 0; TN = (variant == 4) ? 128 : 256; }
    else if (nq > 64) { cfg = 1; TN = 128; }
    else if (nq > 32) { cfg = 2; TN = 64; }
    else { cfg = 3; TN = 32; }
    const int TM = 256;
    const int64_t qtiles = ceil_div64(nq, TN);
    // Workgroups per launch = doc tiles x query tiles: a multiple of the 256 CUs (one workgroup per CU:
    // no partially filled last round), at least 2048.
    int64_t g = 256, t = qtiles;
    while (t) { const int64_t r = g % t; g = t; t = r; }     // g = gcd(256, qtiles)
    const int64_t unit = 256 / g;
    const char* env_wgs = sr_dev_getenv("SR_DENSE_LAUNCH_WGS");      // A/B switch: workgroups per launch (default 2048)
    // the 16-bit passes take twice the docs per launch (28 rounds of tiles instead of 14 at 6 980 queries): half the launch ramps and
    // compactions, 0.7606 -> 0.7417 ms per 14 rounds, search -2 % (same results; tools/split_ab.py SR_DENSE_LAUNCH_WGS=...)
    const int64_t launch_wgs = env_wgs ? atoll(env_wgs) : (pass16 ? 7168 : 2048);
    int64_t chunk = TM * unit * ceil_div64(launch_wgs, unit * qtiles);
    int64_t max_cap = idx->ws_limit / (8 * nq);
    max_cap = (max_cap / TM) * TM;
    if (max_cap < TM) max_cap = TM;
    if (chunk > max_cap) chunk = max_cap;
    if (pass16) {
        // scores on the 16-bit MFMA pipe (dense_split.hip); same chunking and top-k machinery
        const int np = precision == SR_PASS_FILTER ? 0 : planes_of(precision);
        if (idx->q_cap < nq) {
            for (int p = 0; p < 3; ++p) {
                if (idx->qpl[p]) (void)hipFree(idx->qpl[p]);
                idx->qpl[p] = nullptr;
            }
            idx->q_cap = 0;
            for (int p = 0; p < 3; ++p) SR_CHECK_HIP(hipMalloc((void**)&idx->qpl[p], (size_t)nq * idx->dim * 2));
            idx->q_cap = nq;
        }
        // the filter's second threshold (dense_filter.h): k_inner = the k of the search, `k` here = kp candidates
        bool two = np == 0 && k_inner > 0 && k_inner < k;
        if (const char* e = sr_dev_getenv("SR_FILTER_TAU2")) two = two && atoi(e) != 0;                  // A/B switch
        // with the second threshold the running set is cut back (and both thresholds refreshed) once it holds k + 1 024 keys instead of 2 k:
        // fresher thresholds save more in the pass than the extra selects cost (search 229.3 -> 224.4 ms at the MSMARCO shape)
        int select_over = two ? k + 1024 : 2 * k;
        if (const char* e = sr_dev_getenv("SR_FILTER_SELECT_OVER")) select_over = atoi(e);              // A/B switch
        float* f_slack = idx->qa + 4 * idx->fq_cap, *f_tau2 = f_slack + idx->fq_cap, *f_tau_eff = f_tau2 + idx->fq_cap;
        if (np == 0) SR_TRY(launch_filter_queries(d_queries, nq, idx->dim, idx->qpl[0], idx->qa, s));     // idx->qa: dense_search_filtered
        else SR_TRY(launch_split_bf16(d_queries, idx->qpl[0], idx->qpl[1], idx->qpl[2], nq * (int64_t)idx->dim, s));
        if (two) {
            FilterSegMax fm;
            fm.count = (int)idx->segs.size();
            for (int i = 0; i < fm.count; ++i) { fm.x[i] = idx->segs[i].fx_max; fm.y[i] = idx->segs[i].fy_max; fm.isd[i] = idx->segs[i].fisd; }
            SR_TRY(launch_filter_slack(idx->qa, nq, fm, f_slack, f_tau2, s));
        }
        // segmented candidate slots: one segment per (256-doc tile of a launch, producer lane group), see common.h
        bool use_seg = true;
        if (const char* e = sr_dev_getenv("SR_SPLIT_SEG")) use_seg = atoi(e) != 0;       // A/B switch: 0 = atomic appends only
        if (use_seg) SR_TRY(ws.ensure_segments(nq, k, chunk, (int)(chunk / TM) * SR_SEG_PROD));
        else SR_TRY(ws.ensure(nq, k, chunk));
        SR_TRY(topk_reset(ws, nq, s));
        int64_t step = ceil_div64((int64_t)k + 1024, TM) * TM;   // short first launches, see below
        if (step < TM * ceil_div64(256, qtiles)) step = TM * ceil_div64(256, qtiles);
        if (step > chunk) step = chunk;
        bool first_launch = true;
        for (const DenseSegment& seg : idx->segs) {
            for (int64_t r0 = 0; r0 < seg.n;) {
                const int64_t r1 = r0 + step < seg.n ? r0 + step : seg.n;
                const int64_t r0_next = r1;
                step = step * 2 < chunk ? step * 2 : chunk;
                DenseSplitArgs a{};
                for (int p = 0; p < 3; ++p) { a.D[p] = seg.pl[p]; a.Q[p] = idx->qpl[p]; }
                if (np == 0) {            // the certified filter: one fp16 plane product + the per-pair error term
                    a.n_pairs = 1;
                    a.pair_d[0] = 0; a.pair_q[0] = 0;
                    a.D[0] = seg.fpl;
                    a.upper_bound = 1;
                    a.dxy = seg.fxy; a.qa = idx->qa; a.sd = seg.fsd; a.isd = seg.fisd;
                    a.dxy_gmax = seg.fxy + 2 * seg.n;
                } else if (np == 2) {     // (d plane, q plane), smallest products first
                    a.n_pairs = 3;
                    const int pd[3] = {1, 0, 0}, pq[3] = {0, 1, 0};
                    for (int i = 0; i < 3; ++i) { a.pair_d[i] = pd[i]; a.pair_q[i] = pq[i]; }
                } else {
                    a.n_pairs = 6;
                    const int pd[6] = {2, 0, 1, 1, 0, 0}, pq[6] = {0, 2, 1, 0, 1, 0};
                    for (int i = 0; i < 6; ++i) { a.pair_d[i] = pd[i]; a.pair_q[i] = pq[i]; }
                }
                a.row_begin = r0; a.row_end = r1; a.H = idx->dim; a.nq = (int)nq;
                a.tau = two ? f_tau_eff : ws.tau; a.cand_keys = ws.cand_keys; a.cand_count = ws.cand_count;
                a.cand_cap = ws.cand_cap; a.id_base = (uint32_t)seg.id_base; a.id_stride = (uint32_t)seg.id_stride;
                a.seg_cnt = ws.seg_n > 0 ? ws.seg_cnt : nullptr; a.seg_n = ws.seg_n; a.seg_off = ws.seg_off;
                if (two && first_launch) SR_TRY(launch_filter_tau(ws.tau, f_tau2, f_slack, f_tau_eff, nq, s));    // both -inf
                first_launch = false;
                idx->prof.begin(s);
                SR_TRY(launch_dense_split(a, s));
                idx->prof.end(s, 2.0 * (double)nq * (double)(r1 - r0) * idx->dim, (double)(r1 - r0) * idx->dim * 4.0);
                if (two) {
                    SR_TRY(topk_compact2(ws, nq, k, k_inner, f_tau2, select_over, s));
                    SR_TRY(launch_filter_tau(ws.tau, f_tau2, f_slack, f_tau_eff, nq, s));
                } else {
                    SR_TRY(topk_compact(ws, nq, k, s));
                }
                r0 = r0_next;
            }
        }
        SR_TRY(topk_finalize(ws, nq, k, -3.402823466e38f, d_out_scores, d_out_ids, nullptr, s));
        return SR_OK;
    }
    const bool use_stream = variant != 9 && !force_tiled && !idx->batch_invariant && dense_stream_supports((int)nq, idx->dim);
    if (use_stream) {
        // HBM-bound regime: D straight to registers, chunks grow geometrically (64 Ki docs, x2 per launch)
        int64_t cap = idx->ws_limit / (8 * nq);
        if (cap > (1ll << 22)) cap = 1ll << 22;
        cap = (cap / 128) * 128;
        if (cap < 128) cap = 128;
        SR_TRY(ws.ensure(nq, k, cap));
        SR_TRY(topk_reset(ws, nq, s));
        int64_t step = 65536 < cap ? 65536 : cap;
        for (const DenseSegment& seg : idx->segs) {
            for (int64_t r0 = 0; r0 < seg.n;) {
                const int64_t r1 = r0 + step < seg.n ? r0 + step : seg.n;
                DenseStreamArgs a;
                a.D = seg.rows; a.Q = d_queries; a.row_begin = r0; a.row_end = r1; a.H = idx->dim; a.nq = (int)nq;
                a.tau = ws.tau; a.cand_keys = ws.cand_keys; a.cand_count = ws.cand_count;
                a.cand_cap = ws.cand_cap; a.id_base = (uint32_t)seg.id_base; a.id_stride = (uint32_t)seg.id_stride;
                idx->prof.begin(s);
                SR_TRY(launch_dense_stream(a, s));
                idx->prof.end(s, 2.0 * (double)nq * (double)(r1 - r0) * idx->dim, (double)(r1 - r0) * idx->dim * 4.0);
                SR_TRY(topk_compact(ws, nq, k, s));
                r0 = r1;
                step = step * 2 < cap ? step * 2 : cap;
            }
        }
        SR_TRY(topk_finalize(ws, nq, k, -3.402823466e38f, d_out_scores, d_out_ids, nullptr, s));
        return SR_OK;
    }
    SR_TRY(ws.ensure(nq, k, chunk));
    SR_TRY(topk_reset(ws, nq, s));

    // The first launches see no threshold yet (every doc is a candidate until k have been seen), so they are kept short
    // and doubled - 2048, 4096, ... docs - until the regular chunk: each then appends about k survivors per query
    // instead of a whole chunk's worth for the first one (32 768 keys per query, 1.8 GB at 6980 queries).
    // ... but never shorter than one workgroup per CU
    int64_t step = ceil_div64((int64_t)k + 1024, TM) * TM;
    if (step < TM * ceil_div64(256, qtiles)) step = TM * ceil_div64(256, qtiles);
    if (step > chunk) step = chunk;
    for (const DenseSegment& seg : idx->segs) {
        for (int64_t r0 = 0; r0 < seg.n;) {
            const int64_t r1 = r0 + step < seg.n ? r0 + step : seg.n;
            const int64_t r0_next = r1;
            step = step * 2 < chunk ? step * 2 : chunk;
            DenseArgs a;
            a.D = seg.rows;
            a.Q = d_queries;
            a.row_begin = r0;
            a.row_end = r1;
            a.H = idx->dim;
            a.nq = (int)nq;
            a.tau = ws.tau;
            a.cand_keys = ws.cand_keys;
            a.cand_count = ws.cand_count;
            a.cand_cap = ws.cand_cap;
            a.id_base = (uint32_t)seg.id_base;
            a.id_stride = (uint32_t)seg.id_stride;
            idx->prof.begin(s);
            switch (cfg) {
                case 0:
                    if (variant == 0) SR_TRY((launch_dense<2, 2, 4, 4>(a, r1 - r0, s)));
                    else if (variant == 1) SR_TRY((launch_dense<2, 4, 4, 2>(a, r1 - r0, s)));
                    else if (variant == 2) SR_TRY((launch_dense<2, 4, 4, 2, 32>(a, r1 - r0, s)));
                    else if (variant == 3) SR_TRY((launch_dense<4, 4, 2, 2>(a, r1 - r0, s)));
                    else if (variant == 5) SR_TRY(launch_dense_pipe<2>(a, r1 - r0, s));
                    else SR_TRY((launch_dense<4, 2, 2, 2, 16, 4>(a, r1 - r0, s)));   // 256 x 128 tile, 2 workgroups per CU
                    break;
                case 1:
                    if (variant == 5) SR_TRY(launch_dense_pipe<1>(a, r1 - r0, s));
                    else SR_TRY((launch_dense<2, 2, 4, 2>(a, r1 - r0, s)));
                    break;
                case 2: SR_TRY((launch_dense<4, 1, 2, 2>(a, r1 - r0, s))); break;
                default: SR_TRY((launch_dense<4, 1, 2, 1>(a, r1 - r0, s))); break;
            }
            idx->prof.end(s, 2.0 * (double)nq * (double)(r1 - r0) * idx->dim, (double)(r1 - r0) * idx->dim * 4.0);
            SR_TRY(topk_compact(ws, nq, k, s));
            r0 = r0_next;
        }
    }
    SR_TRY(topk_finalize(ws, nq, k, -3.402823466e38f, d_out_scores, d_out_ids, nullptr, s));
    return SR_OK;
}

// SR_PRECISION_FP32_FILTERED: exact results at 16-bit MFMA speed (dense_filter.hip).  Returns SR_OK with *done = false when
// the batch has to go through the exact kernel as a whole (filter not applicable to this index / batch).
static int filter_segs_of(sr_dense_index* idx, FilterSegs& fs) {
    fs.count = (int)idx->segs.size();
    for (int i = 0; i < fs.count; ++i) {
        fs.rows[i] = idx->segs[i].rows; fs.n[i] = idx->segs[i].n;
        fs.xy[i] = idx->segs[i].fxy; fs.isd[i] = idx->segs[i].fisd;
        fs.id_base[i] = (uint32_t)idx->segs[i].id_base; fs.id_stride[i] = (uint32_t)idx->segs[i].id_stride;
    }
    return SR_OK;
}

// First half: the kp documents with the largest upper bounds U per query (idx->a_scores / a_ids / qa).  *done = false: the
// filter does not apply to this index / batch.
static int dense_filtered_candidates(sr_dense_index* idx, const float* d_queries, int64_t nq, int k, hipStream_t s, bool* done) {
    *done = false;
    int kp = 3 * k > k + 2048 ? 3 * k : k + 2048;            // candidates per query: k = 1000 -> 3072
    if (const char* e = sr_dev_getenv("SR_FILTER_KP")) kp = atoi(e);
    if (kp > SR_MAX_TOPK) kp = SR_MAX_TOPK;
    if (nq <= 64 || kp < k + 64 || idx->dim % 64 != 0 || idx->dim < 128 || (int)idx->segs.size() > SR_FILTER_MAX_SEGS) return SR_OK;
    for (DenseSegment& seg : idx->segs) {
        if (seg.f_state == 0) SR_TRY(filter_prepare_segment(idx, seg));
        if (seg.f_state != 1) return SR_OK;                   // not filterable / no room for the plane: exact kernel
    }
    if (idx->fq_cap < nq || idx->fkp != kp) {
        auto F = [](void* p) { if (p) (void)hipFree(p); };
        F(idx->qa); F(idx->a_scores); F(idx->a_ids); F(idx->flags);
        idx->qa = nullptr; idx->a_scores = nullptr; idx->a_ids = nullptr; idx->flags = nullptr;
        idx->fq_cap = 0;
        if (hipMalloc((void**)&idx->qa, (size_t)nq * 28) != hipSuccess || hipMalloc((void**)&idx->a_scores, (size_t)nq * kp * 4) != hipSuccess ||
            hipMalloc((void**)&idx->a_ids, (size_t)nq * kp * 8) != hipSuccess || hipMalloc((void**)&idx->flags, (size_t)(2 * nq + 2) * 4) != hipSuccess) {
            (void)hipGetLastError();
            return SR_OK;                                     // no room for the candidate lists: exact kernel
        }
        idx->fq_cap = nq;
        idx->fkp = kp;
    }
    SR_CHECK_HIP(hipMemsetAsync(idx->flags, 0, (size_t)nq * 4, s));
    // 1. the kp documents with the largest upper bounds U (the query planes and constants are made by the pass)
    SR_TRY(dense_search_pass(idx, d_queries, nq, kp, idx->a_scores, idx->a_ids, SR_PASS_FILTER, s, false, k));
    *done = true;
    return SR_OK;
}

// Second half: exact re-score of the candidates that can still be in the top-k, certificate, exact re-do of the queries without
// one.  d_thr (nullable, doc-sharded search): per query a value proven not to exceed the GLOBAL k-th exact score.
static int dense_filtered_finish(sr_dense_index* idx, const float* d_queries, int64_t nq, int k, const float* d_thr, float* d_out_scores,
                                 int64_t* d_out_ids, hipStream_t s) {
    const int kp = idx->fkp;
    FilterSegs fs;
    SR_TRY(filter_segs_of(idx, fs));
    // 2. exact scores of the candidates that can still be in the top-k -> exact top-k
    SR_TRY(idx->ws2.ensure(nq, k, kp));
    SR_TRY(topk_reset(idx->ws2, nq, s));
    SR_TRY(launch_filter_rescore(fs, d_queries, idx->a_scores, idx->a_ids, idx->qa, nq, k, kp, idx->dim, idx->ws2.cand_keys,
                                 idx->ws2.cand_count, idx->ws2.cand_cap, idx->flags, reinterpret_cast<unsigned int*>(idx->flags) + nq + 1, d_thr, s));
    SR_TRY(topk_compact(idx->ws2, nq, k, s));
    SR_TRY(topk_finalize(idx->ws2, nq, k, -3.402823466e38f, d_out_scores, d_out_ids, nullptr, s));
    // 3. certificate against the k-th exact score
    SR_TRY(launch_filter_certify(idx->a_scores, d_out_scores, idx->qa, nq, k, kp, idx->flags, d_thr, s));
    // the queries that were not certified are re-done by the exact kernel - those alone (one small D2H per search)
    std::vector<int> h((size_t)nq);
    SR_CHECK_HIP(hipMemcpyAsync(h.data(), idx->flags, (size_t)nq * 4, hipMemcpyDeviceToHost, s));
    SR_CHECK_HIP(hipStreamSynchronize(s));
    std::vector<int64_t> redo;
    for (int64_t q = 0; q < nq; ++q)
        if (h[(size_t)q] != 0) redo.push_back(q);
    const int64_t nf = (int64_t)redo.size();
    idx->nq_certified += nq - nf;
    idx->nq_redone += nf;
    if (nf == 0) { ++idx->n_filtered; return SR_OK; }
    ++idx->n_fallback;
    if (nf * 2 > nq) {                                         // most of the batch: redo all of it in place
        return dense_search_pass(idx, d_queries, nq, k, d_out_scores, d_out_ids, SR_PRECISION_FP32, s);
    }
    if (idx->redo_cap < nf || idx->redo_k != k) {
        auto F = [](void* p) { if (p) (void)hipFree(p); };
        F(idx->redo_q); F(idx->redo_idx); F(idx->redo_scores); F(idx->redo_ids);
        idx->redo_q = nullptr; idx->redo_idx = nullptr; idx->redo_scores = nullptr; idx->redo_ids = nullptr;
        idx->redo_cap = 0;
        const int64_t cap = nf < 64 ? 64 : nf;
        SR_CHECK_HIP(hipMalloc((void**)&idx->redo_q, (size_t)cap * idx->dim * 4));
        SR_CHECK_HIP(hipMalloc((void**)&idx->redo_idx, (size_t)cap * 8));
        SR_CHECK_HIP(hipMalloc((void**)&idx->redo_scores, (size_t)cap * k * 4));
        SR_CHECK_HIP(hipMalloc((void**)&idx->redo_ids, (size_t)cap * k * 8));
        idx->redo_cap = cap;
        idx->redo_k = k;
    }
    SR_CHECK_HIP(hipMemcpyAsync(idx->redo_idx, redo.data(), (size_t)nf * 8, hipMemcpyHostToDevice, s));
    SR_TRY(launch_filter_gather_rows(d_queries, idx->redo_idx, nf, idx->dim, idx->redo_q, false, s));
    // the tiled kernels whatever the count: one k order for the whole batch (the streaming kernel accumulates in another)
    SR_TRY(dense_search_pass(idx, idx->redo_q, nf, k, idx->redo_scores, idx->redo_ids, SR_PRECISION_FP32, s, true));
    SR_TRY(launch_filter_gather_rows(idx->redo_scores, idx->redo_idx, nf, k, d_out_scores, true, s));
    SR_TRY(launch_filter_gather_rows(idx->redo_ids, idx->redo_idx, nf, 2 * (int64_t)k, d_out_ids, true, s));
    SR_CHECK_HIP(hipStreamSynchronize(s));                    // `redo` (pageable host memory) is the source of an async copy
    return SR_OK;
}

static int dense_search_filtered(sr_dense_index* idx, const float* d_queries, int64_t nq, int k, float* d_out_scores,
                                 int64_t* d_out_ids, hipStream_t s, bool* done) {
    SR_TRY(dense_filtered_candidates(idx, d_queries, nq, k, s, done));
    if (!*done) return SR_OK;
    return dense_filtered_finish(idx, d_queries, nq, k, nullptr, d_out_scores, d_out_ids, s);
}

extern "C" int sr_dense_search(sr_dense_index* idx, const float* d_queries, int64_t nq, int k, float* d_out_scores,
                               int64_t* d_out_ids, sr_stream stream) {
    SR_REQUIRE(idx, "sr_dense_search: null index");
    SR_REQUIRE(nq >= 0 && nq < (1ll << 30), "sr_dense_search: bad nq=%lld", (long long)nq);
    SR_REQUIRE(k >= 1 && k <= SR_MAX_TOPK, "sr_dense_search: k=%d outside [1, %d]", k, SR_MAX_TOPK);
    if (nq == 0) return SR_OK;
    SR_REQUIRE(d_queries && d_out_scores && d_out_ids, "sr_dense_search: null pointer");
    SR_REQUIRE(((uintptr_t)d_queries & 15) == 0, "sr_dense_search: queries must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    std::lock_guard<std::mutex> lock(idx->mu);
    StreamOrder::Scope in_order(idx->order, s);
    if (idx->precision == SR_PRECISION_FP32_FILTERED) {
        bool done = false;
        SR_TRY(dense_search_filtered(idx, d_queries, nq, k, d_out_scores, d_out_ids, s, &done));
        if (done) return SR_OK;
        if (nq > 64) { ++idx->n_fallback; idx->nq_redone += nq; }
        return dense_search_pass(idx, d_queries, nq, k, d_out_scores, d_out_ids, SR_PRECISION_FP32, s);
    }
    return dense_search_pass(idx, d_queries, nq, k, d_out_scores, d_out_ids, idx->precision, s);
}

// Doc-sharded search in two halves (distributed.py ShardedDenseRetriever): every rank runs _begin on its shard, the ranks take
// the minimum of d_lower over the shards (nq floats, one small all-reduce), every rank runs _finish with it.  A shard then
// re-scores only the candidates that can reach the GLOBAL top-k - about k / W of them instead of k - and returns those (the
// rest of its [nq, k] output is padding, id -1): the merge of the shards' outputs is the global top-k, bit for bit what one
// index over all documents returns.  `share` = the number of shards W.  When the certified filter does not apply (exact
// precision mode, <= 64 queries, no room for the plane) d_lower is -inf and _finish is a plain sr_dense_search.  The pair must
// not be interleaved with other searches on the same handle.
extern "C" int sr_dense_search_begin(sr_dense_index* idx, const float* d_queries, int64_t nq, int k, int share, float* d_lower,
                                     sr_stream stream) {
    SR_REQUIRE(idx && d_lower, "sr_dense_search_begin: null argument");
    SR_REQUIRE(nq >= 0 && nq < (1ll << 30) && k >= 1 && k <= SR_MAX_TOPK && share >= 1, "sr_dense_search_begin: bad argument");
    if (nq == 0) return SR_OK;
    SR_REQUIRE(d_queries && ((uintptr_t)d_queries & 15) == 0, "sr_dense_search_begin: queries must be non-null and 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    std::lock_guard<std::mutex> lock(idx->mu);
    StreamOrder::Scope in_order(idx->order, s);
    idx->pend_nq = 0;
    bool done = false;
    if (idx->precision == SR_PRECISION_FP32_FILTERED) SR_TRY(dense_filtered_candidates(idx, d_queries, nq, k, s, &done));
    if (!done) {
        std::vector<float> h((size_t)nq, -INFINITY);
        SR_CHECK_HIP(hipMemcpyAsync(d_lower, h.data(), (size_t)nq * 4, hipMemcpyHostToDevice, s));
        SR_CHECK_HIP(hipStreamSynchronize(s));
        return SR_OK;
    }
    FilterSegs fs;
    SR_TRY(filter_segs_of(idx, fs));
    int j = (k + share - 1) / share;
    if (j > idx->fkp) j = idx->fkp;
    SR_TRY(launch_filter_lower_bound(fs, d_queries, idx->a_scores, idx->a_ids, idx->qa, nq, idx->fkp, j, idx->dim, idx->flags,
                                     reinterpret_cast<unsigned int*>(idx->flags) + nq + 1, d_lower, s));
    idx->pend_nq = nq; idx->pend_k = k; idx->pend_q = d_queries;
    return SR_OK;
}

extern "C" int sr_dense_search_finish(sr_dense_index* idx, const float* d_queries, int64_t nq, int k, const float* d_threshold,
                                      float* d_out_scores, int64_t* d_out_ids, sr_stream stream) {
    SR_REQUIRE(idx, "sr_dense_search_finish: null index");
    if (nq == 0) return SR_OK;
    SR_REQUIRE(d_queries && d_out_scores && d_out_ids, "sr_dense_search_finish: null pointer");
    bool pending;
    {
        std::lock_guard<std::mutex> lock(idx->mu);
        pending = idx->pend_nq == nq && idx->pend_k == k && idx->pend_q == d_queries && nq > 0;
        idx->pend_nq = 0;
        if (pending) {
            hipStream_t s = (hipStream_t)stream;
            StreamOrder::Scope in_order(idx->order, s);
            return dense_filtered_finish(idx, d_queries, nq, k, d_threshold, d_out_scores, d_out_ids, s);
        }
    }
    return sr_dense_search(idx, d_queries, nq, k, d_out_scores, d_out_ids, stream);
}

extern "C" int sr_dense_index_filter_stats(sr_dense_index* idx, int64_t* n_filtered, int64_t* n_fallback) {
    SR_REQUIRE(idx && n_filtered && n_fallback, "sr_dense_index_filter_stats: null argument");
    std::lock_guard<std::mutex> lock(idx->mu);
    *n_filtered = idx->n_filtered;
    *n_fallback = idx->n_fallback;
    return SR_OK;
}

extern "C" int sr_dense_index_filter_query_stats(sr_dense_index* idx, int64_t* n_certified, int64_t* n_redone) {
    SR_REQUIRE(idx && n_certified && n_redone, "sr_dense_index_filter_query_stats: null argument");
    std::lock_guard<std::mutex> lock(idx->mu);
    *n_certified = idx->nq_certified;
    *n_redone = idx->nq_redone;
    return SR_OK;
}

extern "C" int sr_dense_index_profile(sr_dense_index* idx, int enable) {
    SR_REQUIRE(idx, "sr_dense_index_profile: null index");
    std::lock_guard<std::mutex> lock(idx->mu);
    idx->prof.enabled = enable != 0;
    return SR_OK;
}

extern "C" int sr_dense_index_profile_read(sr_dense_index* idx, int64_t* n_launches, double* total_ms, double* total_flop,
                                           double* total_d_bytes) {
    SR_REQUIRE(idx && n_launches && total_ms && total_flop && total_d_bytes, "sr_dense_index_profile_read: null argument");
    std::lock_guard<std::mutex> lock(idx->mu);
    *total_flop = idx->prof.flop;
    *total_d_bytes = idx->prof.bytes;
    *n_launches = idx->prof.read(total_ms);
    idx->prof.flop = idx->prof.bytes = 0;
    return SR_OK;
}
