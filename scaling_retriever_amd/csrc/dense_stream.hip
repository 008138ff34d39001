// Dense scoring for SMALL query batches (nq <= 64): the HBM-bound regime of north_star.
//
// D streams from HBM exactly once, straight into registers in the MFMA A-operand layout
// (no LDS round trip for D: each lane's global_load_dwordx4 is one 16-B piece of a doc row
// and feeds 4 k-steps of v_mfma_f32_16x16x4_f32).  The query slab (<= 64 queries x 256 k
// fp32, <= 66 KB) sits in LDS and is shared by the workgroup's 4 waves x 2 doc blocks.
// Algorithmic bytes = rows * H * 4 per launch; arithmetic is exact fp32 (k-ordered fmaf
// chain): per 16-wide k group s the chain visits, for jj = 0..3, k = 16s + 4g + jj for
// g = 0..3  (oracle/scoring.py::mfma_korder16).
#include "common.h"
#include "dense_stream.h"

#define DS_KS 256                 // k per query slab
#define DS_LDQ (DS_KS + 4)        // padded LDS row (floats)
#define DS_R 2                    // 16-doc blocks per wave
#define DS_TM (4 * DS_R * 16)     // docs per workgroup = 128
#define DS_PF 8                   // 16-k groups per prefetch step (8 x float4 per lane per doc block)

template <int NQB>  // number of 16-query blocks (1 or 2)
__global__ __launch_bounds__(256) void dense_stream_kernel(DenseStreamArgs a) {
    extern __shared__ __attribute__((aligned(16))) float Qs[];   // [NQB * 16][DS_LDQ]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, g = lane >> 4;
    const int H = a.H;
    const int64_t row0 = a.row_begin + (int64_t)blockIdx.x * DS_TM + wave * (DS_R * 16);

    const float* drow[DS_R];
#pragma unroll
    for (int r = 0; r < DS_R; ++r) {
        int64_t row = row0 + r * 16 + li;
        row = row < a.row_end ? row : a.row_end - 1;   // clamp: results of padded rows are never emitted
        drow[r] = a.D + row * H + 4 * g;
    }
    f32x4 acc[DS_R][NQB];
#pragma unroll
    for (int r = 0; r < DS_R; ++r)
#pragma unroll
        for (int b = 0; b < NQB; ++b) acc[r][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    f32x4 nxt[DS_R][DS_PF], cur[DS_R][DS_PF];
    auto gload = [&](int k0) {
#pragma unroll
        for (int r = 0; r < DS_R; ++r)
#pragma unroll
            for (int s = 0; s < DS_PF; ++s) nxt[r][s] = *reinterpret_cast<const f32x4*>(drow[r] + k0 + 16 * s);
    };
    const int nsteps = H / (16 * DS_PF);          // prefetch steps of 128 k
    constexpr int STEPS_PER_SLAB = DS_KS / (16 * DS_PF);
    gload(0);
    for (int step = 0; step < nsteps; ++step) {
        if (step % STEPS_PER_SLAB == 0) {
            // stage the next query slab: Q[q][slab*KS .. +KS) -> Qs[q][0..KS)
            __syncthreads();
            const int k0 = step * 16 * DS_PF;
            for (int c = tid; c < NQB * 16 * (DS_KS / 4); c += 256) {
                const int q = c / (DS_KS / 4), kc = c % (DS_KS / 4);
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (q < a.nq) v = *reinterpret_cast<const f32x4*>(a.Q + (int64_t)q * H + k0 + kc * 4);
                *reinterpret_cast<f32x4*>(&Qs[q * DS_LDQ + kc * 4]) = v;
            }
            __syncthreads();
        }
#pragma unroll
        for (int r = 0; r < DS_R; ++r)
#pragma unroll
            for (int s = 0; s < DS_PF; ++s) cur[r][s] = nxt[r][s];
        if (step + 1 < nsteps) gload((step + 1) * 16 * DS_PF);
        const int kslab = (step % STEPS_PER_SLAB) * 16 * DS_PF;
#pragma unroll
        for (int s = 0; s < DS_PF; ++s) {
            f32x4 qf[NQB];
#pragma unroll
            for (int b = 0; b < NQB; ++b)
                qf[b] = *reinterpret_cast<const f32x4*>(&Qs[(b * 16 + li) * DS_LDQ + kslab + 16 * s + 4 * g]);
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                for (int r = 0; r < DS_R; ++r)
#pragma unroll
                    for (int b = 0; b < NQB; ++b)
                        acc[r][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[r][s][jj], qf[b][jj], acc[r][b], 0, 0, 0);
        }
    }

    // ---- epilogue: accumulator lane = query (lane & 15), registers = docs 4*(lane >> 4) + x ----
    const int64_t left = a.row_end - row0;
#pragma unroll
    for (int b = 0; b < NQB; ++b) {
        const int q = b * 16 + li;
        if (q >= a.nq) continue;
        const float tq = a.tau[q];
        int cnt = 0;
#pragma unroll
        for (int r = 0; r < DS_R; ++r)
#pragma unroll
            for (int x = 0; x < 4; ++x) cnt += ((int64_t)(r * 16 + 4 * g + x) < left && acc[r][b][x] >= tq) ? 1 : 0;
        if (cnt == 0) continue;
        int pos = atomicAdd(&a.cand_count[q], cnt);
        uint64_t* dst = a.cand_keys + (int64_t)q * a.cand_cap;
#pragma unroll
        for (int r = 0; r < DS_R; ++r)
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                const int lr = r * 16 + 4 * g + x;
                const float sc = acc[r][b][x];
                if ((int64_t)lr < left && sc >= tq) {
                    if (pos < a.cand_cap) dst[pos] = sr_make_key(sc, a.id_base + (uint32_t)(row0 + lr) * a.id_stride);
                    ++pos;
                }
            }
    }
}

int launch_dense_stream(const DenseStreamArgs& a, hipStream_t s) {
    const int64_t rows = a.row_end - a.row_begin;
    if (rows <= 0) return SR_OK;
    SR_REQUIRE(a.nq >= 1 && a.nq <= 64, "dense_stream: nq=%d outside [1, 64]", a.nq);
    SR_REQUIRE(a.H % (16 * DS_PF) == 0 && a.H % DS_KS == 0, "dense_stream: dim %d must be a multiple of %d", a.H, DS_KS);
    const dim3 grid((unsigned)ceil_div64(rows, DS_TM));
    const int nqb = (a.nq + 15) / 16;
    const size_t lds = (size_t)nqb * 16 * DS_LDQ * sizeof(float);
    static DeviceOnce attr_once;
    bool* attr_slot = attr_once.pending();
    if (attr_slot) {
        SR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&dense_stream_kernel<3>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 16 * DS_LDQ * 4));
        SR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&dense_stream_kernel<4>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 16 * DS_LDQ * 4));
        *attr_slot = true;
    }
    switch (nqb) {
        case 1: hipLaunchKernelGGL(dense_stream_kernel<1>, grid, dim3(256), lds, s, a); break;
        case 2: hipLaunchKernelGGL(dense_stream_kernel<2>, grid, dim3(256), lds, s, a); break;
        case 3: hipLaunchKernelGGL(dense_stream_kernel<3>, grid, dim3(256), lds, s, a); break;
        default: hipLaunchKernelGGL(dense_stream_kernel<4>, grid, dim3(256), lds, s, a); break;
    }
    SR_CHECK_LAUNCH();
    return SR_OK;
}

bool dense_stream_supports(int nq, int H) { return nq >= 1 && nq <= 64 && H % DS_KS == 0; }
