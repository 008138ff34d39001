// Exact top-k through a certified filter (dense_filter.hip): pieces used by sr_dense_search in SR_PRECISION_FP32_FILTERED.
#pragma once
#include "common.h"

#define SR_FILTER_MAX_SEGS 64
struct FilterSegs {                 // where a global doc index lives: row = (gid - id_base) / id_stride of segment s
    const float* rows[SR_FILTER_MAX_SEGS];
    int64_t n[SR_FILTER_MAX_SEGS];
    uint32_t id_base[SR_FILTER_MAX_SEGS], id_stride[SR_FILTER_MAX_SEGS];
    int count;
};

// |S_a - S_x| <= sr_filter_c(H, products) * |q| * |d| for the filter's score S_a = q0 . d0 (products = 1) or
// (q0 + q1) . d0 (products = 2) against the fp32 fmaf chain S_x
double sr_filter_c(int H, int products);
// *d_max2 = max(*d_max2, max over rows of |row|^2)   (non-negative floats order as their bit patterns)
int launch_row_norm2_max(const float* rows, int64_t n, int H, float* d_max2, hipStream_t s);
// qnorm[q] = |Q[q]|
int launch_query_norms(const float* Q, int64_t nq, int H, float* qnorm, hipStream_t s);
// Candidates = the kp best approximate scores per query, sorted descending (a_scores / a_ids [nq, kp], pads id < 0).
// launch_filter_rescore: exact score of every candidate that can still be in the top-k (approximate score within 2E of
// the k-th best approximate score, and - for the candidates beyond the k best - within E of the smallest exact score of
// those k, xmin [nq] scratch; the others are provably out) = fp32 fmaf chain in the k order of dense_score_pipe_kernel, key =
// (score desc, doc index asc) into cand_keys [nq, cand_cap]; flags[q] |= 2 if a candidate's two scores differ by more than
// the bound.  launch_filter_certify (after the exact top-k x_scores [nq, k] is known): flags[q] |= 1 unless every document
// outside the candidates is provably below the k-th exact score.  flags must be zeroed first; non-zero -> the caller falls
// back to the exact kernel.
int launch_filter_certify(const float* a_scores, const float* x_scores, const float* qnorm, const float* d_max2, int64_t nq, int k,
                          int kp, double c, int* flags, hipStream_t s);
int launch_filter_rescore(const FilterSegs& segs, const float* Q, const float* a_scores, const int64_t* a_ids, const float* qnorm,
                          const float* d_max2, int64_t nq, int k, int kp, int H, double c, uint64_t* cand_keys, int* cand_count,
                          int64_t cand_cap, int* flags, unsigned int* xmin, hipStream_t s);
