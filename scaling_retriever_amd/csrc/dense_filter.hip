// Exact dense top-k at bf16 speed: a certified filter in front of an exact re-score (gfx950).
//
// sr_dense_search must return what faiss.IndexFlatIP.search returns (scaling_retriever/indexer.py:210-214): the k largest
// fp32 inner products.  dense_score_pipe_kernel computes every one of the nq x N products on the fp32 MFMA pipe (157 TF
// peak).  The bf16 MFMA pipe is 16x faster, and the score S_a = (q0 + q1) . d0 - two bf16 planes of the query against ONE
// bf16 plane of the document, 2 plane products (dense_split.hip) - differs from the fp32 chain S_x by a PROVABLY bounded
// amount, so it can decide which few documents need the exact arithmetic at all:
//
//   1. approximate pass: the kp = 3k best documents by S_a per query (dense_split_kernel + the fused top-k, unchanged);
//   2. exact pass: S_x for those kp candidates only - the same fp32 fmaf chain, in the same k order, as
//      dense_score_pipe_kernel (oracle: scoring.mfma_korder) - then the top-k by (S_x desc, doc index asc);
//   3. certificate: with E = c |q| max|d| >= |S_a - S_x| for every document, a document that is NOT a candidate has
//      S_x <= S_a + E <= a_kp + E (a_kp = the kp-th best S_a).  If a_kp + E is strictly below the k-th best exact score
//      found among the candidates, no outsider can enter the top-k or tie with its last member: the result IS the exact
//      top-k.
// The result is bit-identical to the exact kernel's (tests/test_dense_filtered_gpu.py, and every bench.py run on the whole
// problem).  A query whose certificate fails (more than kp - k documents inside the margin: near-duplicates, a zero query)
// or for which any candidate's |S_a - S_x| exceeds E (the bound is checked on every pair that is re-scored) makes
// sr_dense_search redo the batch with the exact kernel - correctness never rests on the filter.
//
// The bound.  B = sum |q_i||d_i| <= |q||d|;  u = 2^-24, gamma_n = n u / (1 - n u).
//   fp32 chain of H fmaf:                          |S_x - q.d| <= gamma_H B
//   planes: x0 = bf16(x): |x - x0| <= 2^-9 |x|;  x1 = bf16(x - x0): |x - x0 - x1| <= 2^-18 |x|
//     q.d - (q0 + q1).d0 = q.(d - d0) + (q - q0 - q1).d0:      <= (2^-9 + 2^-18 (1 + 2^-9)) B
//   2H bf16 products (exact in fp32) summed in fp32 by the MFMA, 32 per instruction, modelled as no better than a plain
//     fp32 summation of 2H terms:                                <= gamma_2H * 1.004 B
//   c(H) = 2^-9 * 1.004 + 1.25 * (3 H 2^-24) + 1e-5   (2.43e-3 at H = 2048, 2.89e-3 at 4096) covers the sum with room to
//   spare; the measured maximum is ~1e-3 (the 2^-9 plane truncation dominates).
//   One product, S_a = q0 . d0:   q.d - q0.d0 = q.(d - d0) + (q - q0).d0 <= (2^-9 + 2^-9 (1 + 2^-9)) B, H products summed:
//   c1(H) = 2^-8 * 1.004 + the same summation terms (4.38e-3 at H = 2048).  Half the MFMA work of the two-product pass; it
//   needs twice the gap between the k-th and the kp-th score, which sr_dense_search tries first (dense_score.hip).
#include "dense_filter.h"
#include <math.h>

double sr_filter_c(int H, int products) {
    return ldexp(1.0, products == 1 ? -8 : -9) * 1.004 + 1.25 * (3.0 * (double)H * ldexp(1.0, -24)) + 1.0e-5;
}

__global__ __launch_bounds__(256) void row_norm2_max_kernel(const float* __restrict__ rows, int64_t n, int H, float* __restrict__ d_max2) {
    const int lane = threadIdx.x & 63;
    float mx = 0.f;
    for (int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < n; r += (int64_t)gridDim.x * 4) {   // grid-stride (< 2^32 threads)
        const float* p = rows + r * H;
        float ss = 0.f;
        for (int i = lane * 4; i < H; i += 256) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(p + i);
            ss += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
        }
        for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off);
        // NaN / inf rows make every bound meaningless: record +inf so that nothing is ever certified
        if (!(ss < INFINITY)) ss = INFINITY;
        mx = fmaxf(mx, ss);
    }
    if (lane == 0 && mx > 0.f) atomicMax(reinterpret_cast<unsigned int*>(d_max2), __float_as_uint(mx * 1.0001f));   // summation slack
}

int launch_row_norm2_max(const float* rows, int64_t n, int H, float* d_max2, hipStream_t s) {
    if (n == 0) return SR_OK;
    int64_t blocks = ceil_div64(n, 4);
    if (blocks > (1 << 20)) blocks = 1 << 20;
    hipLaunchKernelGGL(row_norm2_max_kernel, dim3((unsigned)blocks), dim3(256), 0, s, rows, n, H, d_max2);
    SR_CHECK_LAUNCH();
    return SR_OK;
}

__global__ __launch_bounds__(256) void query_norm_kernel(const float* __restrict__ Q, int64_t nq, int H, float* __restrict__ qnorm) {
    const int lane = threadIdx.x & 63;
    const int64_t q = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= nq) return;
    float ss = 0.f;
    for (int i = lane * 4; i < H; i += 256) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(Q + q * H + i);
        ss += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    }
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off);
    if (lane == 0) qnorm[q] = (ss < INFINITY) ? sqrtf(ss * 1.0001f) : INFINITY;
}

int launch_query_norms(const float* Q, int64_t nq, int H, float* qnorm, hipStream_t s) {
    hipLaunchKernelGGL(query_norm_kernel, dim3((unsigned)ceil_div64(nq, 4)), dim3(256), 0, s, Q, nq, H, qnorm);
    SR_CHECK_LAUNCH();
    return SR_OK;
}

__global__ void filter_certify_kernel(const float* __restrict__ a_scores, const float* __restrict__ x_scores,
                                      const float* __restrict__ qnorm, const float* __restrict__ d_max2, int64_t nq, int k, int kp,
                                      double c, int* __restrict__ flags) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    const double E = c * (double)qnorm[q] * sqrt((double)*d_max2);
    const double akp = (double)a_scores[q * kp + (kp - 1)];   // kp-th best approximate score; -FLT_MAX pad when fewer documents exist
    const double xk = (double)x_scores[q * k + (k - 1)];      // k-th best EXACT score among the candidates (pad when fewer than k)
    // fewer than kp documents: every document was re-scored.  Otherwise every outsider's exact score is <= akp + E, which
    // must stay strictly below the k-th exact score.  NaN / inf anywhere -> not certified.
    const bool all_docs = akp <= -3.0e38;
    const bool ok = all_docs || (E < INFINITY && xk > -3.0e38 && akp + E < xk);
    if (!ok) atomicOr(&flags[q], 1);
}

int launch_filter_certify(const float* a_scores, const float* x_scores, const float* qnorm, const float* d_max2, int64_t nq, int k,
                          int kp, double c, int* flags, hipStream_t s) {
    hipLaunchKernelGGL(filter_certify_kernel, dim3((unsigned)ceil_div64(nq, 256)), dim3(256), 0, s, a_scores, x_scores, qnorm, d_max2,
                       nq, k, kp, c, flags);
    SR_CHECK_LAUNCH();
    return SR_OK;
}

// One wave per (query, 64 candidates).  The candidates' rows are gathered 64 columns at a time through LDS (coalesced
// 256-byte pieces of each row; lane = candidate reads its row conflict-free from the padded tile), the query chunk is a
// broadcast read, and every lane runs the fp32 fmaf chain of ITS candidate in dense_score_pipe_kernel's k order: per
// group of 8 columns, k = 8s + j then 8s + 4 + j for j = 0..3.
#define RS_KC 64
__global__ __launch_bounds__(64) void filter_rescore_kernel(FilterSegs segs, const float* __restrict__ Q, const float* __restrict__ a_scores,
                                                            const int64_t* __restrict__ a_ids, const float* __restrict__ qnorm,
                                                            const float* __restrict__ d_max2, int k, int kp, int H, double c,
                                                            uint64_t* __restrict__ cand_keys, int* __restrict__ cand_count,
                                                            int64_t cand_cap, int* __restrict__ flags, int j_begin,
                                                            unsigned int* __restrict__ xmin) {
#pragma clang fp contract(off)
    __shared__ float tile[64][RS_KC + 1];
    __shared__ float qs[RS_KC];
    __shared__ const float* rowp[64];
    const int lane = threadIdx.x;
    const int64_t q = blockIdx.x;
    const int j0 = j_begin + blockIdx.y * 64;
    const int j = j0 + lane;
    const int j_end = j_begin == 0 ? (k < kp ? k : kp) : kp;        // stage 1: the k best approximate scores; stage 2: the rest
    if (j0 >= j_end) return;
    // A candidate whose approximate score lies more than 2E below the k-th best approximate score cannot be in the exact
    // top-k: its exact score is < a_k - E, and the k candidates with approximate scores >= a_k all have exact scores >= a_k - E.
    // The approximate list is sorted, so what has to be re-scored is a prefix of it.
    const double E = c * (double)qnorm[q] * sqrt((double)*d_max2);
    double need = (double)a_scores[q * kp + (k < kp ? k : kp) - 1] - 2.0 * E;
    // stage 2 knows better: the smallest EXACT score among the k candidates of stage 1 is a lower bound of the exact k-th
    // score, and a candidate with S_a + E strictly below it cannot reach the top-k
    if (j_begin > 0) {
        const double lo = (double)sr_ord2f(xmin[q]) - E;
        need = lo > need ? lo : need;
    }
    if ((double)a_scores[q * kp + j0] < need) return;           // the whole wave
    int64_t gid = j < j_end ? a_ids[q * kp + j] : -1;
    if (gid >= 0 && (double)a_scores[q * kp + j] < need) gid = -1;
    const float* row = nullptr;
    if (gid >= 0) {
        for (int sgi = 0; sgi < segs.count; ++sgi) {
            const int64_t off = gid - (int64_t)segs.id_base[sgi];
            if (off >= 0 && off % segs.id_stride[sgi] == 0 && off / segs.id_stride[sgi] < segs.n[sgi]) {
                row = segs.rows[sgi] + (off / segs.id_stride[sgi]) * (int64_t)H;
                break;
            }
        }
    }
    rowp[lane] = row;
    __syncthreads();
    float acc = 0.f;
    // 64 rows x 256 B per chunk: 16 lanes per row, 4 rows per instruction.  All 16 loads of a chunk are issued together
    // (interleaved with the LDS stores each one was waited for before the next went out: one memory latency per 4 rows),
    // and the loads of chunk k + 1 fly while chunk k goes through LDS and the fmaf chain.
    const float* myp[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) myp[i] = rowp[i * 4 + (lane >> 4)];
    auto fetch = [&](int k0, f32x4 (&v)[16], float& qv) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (myp[i]) v[i] = *reinterpret_cast<const f32x4*>(myp[i] + k0 + (lane & 15) * 4);
        }
        qv = Q[q * H + k0 + lane];
    };
    auto chunk = [&](const f32x4 (&v)[16], float qv) {
        __syncthreads();                              // the previous chunk is consumed
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            float* t = &tile[i * 4 + (lane >> 4)][(lane & 15) * 4];
            t[0] = v[i][0]; t[1] = v[i][1]; t[2] = v[i][2]; t[3] = v[i][3];
        }
        qs[lane] = qv;
        __syncthreads();
#pragma unroll
        for (int s8 = 0; s8 < RS_KC; s8 += 8)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                acc = __builtin_fmaf(qs[s8 + jj], tile[lane][s8 + jj], acc);
                acc = __builtin_fmaf(qs[s8 + 4 + jj], tile[lane][s8 + 4 + jj], acc);
            }
    };
    f32x4 va[16], vb[16];
    float qa, qb;
    fetch(0, va, qa);
    for (int k0 = 0; k0 < H; k0 += 2 * RS_KC) {       // H is a multiple of RS_KC; an odd chunk count ends in the first half
        const bool has_b = k0 + RS_KC < H;
        if (has_b) fetch(k0 + RS_KC, vb, qb);
        chunk(va, qa);
        if (!has_b) break;
        if (k0 + 2 * RS_KC < H) fetch(k0 + 2 * RS_KC, va, qa);
        chunk(vb, qb);
    }
    if (row) {
        // the bound, checked on every pair that is re-scored
        if (!(fabs((double)acc - (double)a_scores[q * kp + j]) <= E)) atomicOr(&flags[q], 2);
        const int pos = atomicAdd(&cand_count[q], 1);
        if (pos < cand_cap) cand_keys[q * cand_cap + pos] = sr_make_key(acc, (uint32_t)gid);
        if (j_begin == 0) atomicMin(&xmin[q], sr_f2ord(acc));
    }
}

int launch_filter_rescore(const FilterSegs& segs, const float* Q, const float* a_scores, const int64_t* a_ids, const float* qnorm,
                          const float* d_max2, int64_t nq, int k, int kp, int H, double c, uint64_t* cand_keys, int* cand_count,
                          int64_t cand_cap, int* flags, unsigned int* xmin, hipStream_t s) {
    SR_REQUIRE(H % RS_KC == 0, "filter(rescore): dim %d must be a multiple of %d", H, RS_KC);
    SR_REQUIRE(nq <= 0x7fffffff && ceil_div64(kp, 64) <= 65535, "filter(rescore): grid too large");
    // stage 1: the k best candidates by approximate score (and the smallest exact score among them, xmin); stage 2: the rest,
    // pruned against xmin.  When stage 1 saw fewer than k documents there is nothing left for stage 2.
    SR_CHECK_HIP(hipMemsetAsync(xmin, 0xff, (size_t)nq * 4, s));
    const int k1 = k < kp ? k : kp;
    hipLaunchKernelGGL(filter_rescore_kernel, dim3((unsigned)nq, (unsigned)ceil_div64(k1, 64)), dim3(64), 0, s, segs, Q, a_scores, a_ids,
                       qnorm, d_max2, k, kp, H, c, cand_keys, cand_count, cand_cap, flags, 0, xmin);
    SR_CHECK_LAUNCH();
    if (kp > k1) {
        hipLaunchKernelGGL(filter_rescore_kernel, dim3((unsigned)nq, (unsigned)ceil_div64(kp - k1, 64)), dim3(64), 0, s, segs, Q, a_scores,
                           a_ids, qnorm, d_max2, k, kp, H, c, cand_keys, cand_count, cand_cap, flags, k1, xmin);
        SR_CHECK_LAUNCH();
    }
    return SR_OK;
}
