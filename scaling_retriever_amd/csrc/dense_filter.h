// Exact top-k through a certified filter (dense_filter.hip): pieces used by sr_dense_search in SR_PRECISION_FP32_FILTERED.
#pragma once
#include "common.h"

#define SR_FILTER_MAX_SEGS 64
#define SR_FILTER_MAX_SHIFT 40     // the power-of-two scales of segments and queries are 2^t with |t| <= 40
struct FilterSegs {                 // where a global doc index lives: row = (gid - id_base) / id_stride of segment s
    const float* rows[SR_FILTER_MAX_SEGS];
    const float* xy[SR_FILTER_MAX_SEGS];      // [n, 2] per-document error terms of the segment's fp16 plane (scaled domain)
    float isd[SR_FILTER_MAX_SEGS];            // inverse of the segment's scale
    int64_t n[SR_FILTER_MAX_SEGS];
    uint32_t id_base[SR_FILTER_MAX_SEGS], id_stride[SR_FILTER_MAX_SEGS];
    int count;
};

// sigma(H): the share of |q||d| the two fp32 summations (the exact fmaf chain and the MFMA's accumulation of the plane
// product) can differ by, with slack for the epilogue's own roundings
double sr_filter_sigma(int H);
// *d_absmax_bits = max(*d_absmax_bits, bits of max |x| over the rows); a NaN or an infinity shows up as bits >= 0x7f800000
int launch_filter_absmax(const float* rows, int64_t n, int H, unsigned int* d_absmax_bits, hipStream_t s);
// the power-of-two scale that puts absmax into [2^14, 2^15) (clamped to 2^+-SR_FILTER_MAX_SHIFT); false when absmax is not
// finite or the clamp would let a scaled value overflow fp16
bool sr_filter_scale_of(float absmax, float* scale, float* inv_scale);
// fp16 plane of a segment: plane[r, i] = fp16(rows[r, i] * sd), and per document, in the scaled domain,
//   xy[r] = ((|rows_r sd - plane_r| + sigma |rows_r sd|) * 1.001, |plane_r| * 1.001);  *d_bad |= 1 on any non-finite value
int launch_filter_plane(const float* rows, int64_t n, int H, float sd, double sigma, unsigned short* plane, float* xy, int* d_bad,
                        hipStream_t s);
// gmax[g] = (max x, max y) over the documents [128 g, 128 g + 128) of xy [n, 2]: what the upper-bound pass tests a whole block of
// accumulators against before it forms a single per-pair bound (dense_split.hip split_epilogue)
int launch_filter_group_max(const float* xy, int64_t n, float* gmax, hipStream_t s);
// The second threshold of the upper-bound pass.  The k documents with the largest upper bounds seen so far have exact scores
// S >= U - 2 e >= U_(k) - 2 e_max(q), so a document whose U lies below  t2 = U_(k) - 2 e_max(q)  cannot enter the top-k or tie with its
// last member - whatever the kp-th largest U is.  e_max(q) = max over the segments of (A' X_max + B' Y_max) / (sq sd), the error term
// with the largest x and y of the segment.  On data whose scores are not bunched at the k-th one this is a far tighter filter than the
// kp-th largest U (kp = 3 k keeps room for the certificate), and every pair it drops is one the epilogue does not turn into a key.
//   slack[q] = 2 e_max(q) (1 + 2^-9) (rounded up by construction);  tau2[q] := -inf
struct FilterSegMax { float x[SR_FILTER_MAX_SEGS], y[SR_FILTER_MAX_SEGS], isd[SR_FILTER_MAX_SEGS]; int count; };
int launch_filter_slack(const float* qa, int64_t nq, const FilterSegMax& m, float* slack, float* tau2, hipStream_t s);
//   tau_eff[q] = max(tau[q], tau2[q] - slack[q] - 2^-22 |tau2[q]|)      (tau: the kp-th largest U so far, tau2: the k-th, both -inf at first)
int launch_filter_tau(const float* tau, const float* tau2, const float* slack, float* tau_eff, int64_t nq, hipStream_t s);
// fp16 plane of the queries, each scaled by its own power of two sq, and qa[q] = (A', B', sq, 1 / sq):
//   A' = |q sq| * 1.001, B' = |q sq - plane_q| * 1.001; a query that cannot be filtered (non-finite, out of the scale range)
//   gets A' = +inf and is re-done by the exact kernel
int launch_filter_queries(const float* Q, int64_t nq, int H, unsigned short* plane, float* qa, hipStream_t s);
// Candidates = the kp best upper bounds U per query, sorted descending (u_scores / u_ids [nq, kp], pads id < 0).
// launch_filter_rescore: exact score (fp32 fmaf chain in the k order of dense_score_pipe_kernel) of the first min(k, kp)
// candidates, then of every later candidate whose upper bound reaches the smallest exact score among those (xmin [nq]
// scratch) - the others are provably outside the top-k; key = (score desc, doc index asc) into cand_keys [nq, cand_cap];
// flags[q] |= 2 if a candidate's exact score lies outside [U - 2 e, U] (the bound itself, checked on every re-scored pair).
// launch_filter_certify (after the exact top-k x_scores [nq, k] is known): flags[q] |= 1 unless every document outside the
// candidates is provably below the k-th exact score (U_kp < x_k), or there are no outsiders.  flags must be zeroed first;
// a flagged query is re-done by the exact kernel.
int launch_filter_rescore(const FilterSegs& segs, const float* Q, const float* u_scores, const int64_t* u_ids, const float* qa,
                          int64_t nq, int k, int kp, int H, uint64_t* cand_keys, int* cand_count, int64_t cand_cap, int* flags,
                          unsigned int* xmin, const float* thr, hipStream_t s);
int launch_filter_certify(const float* u_scores, const float* x_scores, const float* qa, int64_t nq, int k, int kp, int* flags,
                          const float* thr, hipStream_t s);
// doc-sharded search (sr_dense_search_begin): lower[q] = the smallest EXACT score among the j candidates with the largest upper
// bounds (re-scored by the rescore kernel's fmaf chain): at least j documents of this index reach lower[q], with no appeal to the
// error model; -inf when there are fewer than j documents or a re-scored pair violates [U - 2e, U] (flags[q] |= 2).
// thr (above, nullable): [nq] values not above the GLOBAL k-th exact score
int launch_filter_lower_bound(const FilterSegs& segs, const float* Q, const float* u_scores, const int64_t* u_ids, const float* qa,
                              int64_t nq, int kp, int j, int H, int* flags, unsigned int* xmin, float* lower, hipStream_t s);
// rows of src [*, width] picked by idx [n] -> dst [n, width] (gather), or dst rows idx[i] <- src row i (scatter); 4-byte elements
int launch_filter_gather_rows(const void* src, const int64_t* idx, int64_t n, int64_t width, void* dst, bool scatter, hipStream_t s);
