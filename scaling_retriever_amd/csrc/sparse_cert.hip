// Certified two-stage inverted-index scoring (gfx950): the sparse twin of the dense filter (dense_split / dense_filter).
//
// Reference semantics: SparseRetrieval.numba_score_float + select_topk (scaling_retriever/indexer.py:315-344): per query a
// term-serial fp32 chain  score[doc] += q_t * v  (unfused, ascending query-term order), docs with score > threshold, k best.
// The exact kernels (sparse_score.hip) evaluate that chain for EVERY (query, doc, matching term): 1.1e12 lane operations on the
// columns of the heavy terms per MSMARCO-Dev pass.  For a non-negative index and non-negative queries (always the case behind
// the encoder's relu + log1p head, llm_encoder.py:186-196) this file does the bulk of that work on the matrix pipe instead and
// keeps the results identical:
//
//  stage 1, cert_score_kernel (approximate, with a proven error bound).  Workgroup = (32 queries, persistent over doc tiles of
//    1 024 docs), 16 waves in two roles:
//      matrix waves (8): scores of the T heaviest terms as an fp16 MFMA product (v_mfma_f32_32x32x16_f16, fp32 accumulate) of
//        the tile's [1 024 docs x T] operand - stored in HBM in MFMA fragment order, read once per 32 queries with coalesced
//        16-byte loads, no LDS staging - and the query block's [T x 32] operand held in registers;
//      scatter waves (8): the postings of the other ("rare") query terms inside the tile - lane j = rare term j of the query,
//        runs found through a per-term table of tile boundaries, postings packed to 4 bytes (doc in tile | fp16 value) - are
//        added as 16-bit FIXED-POINT numbers into a [32 queries x 1 024 docs] tile of LDS with integer atomics (ds_add_u32 on
//        one half of a word: exact, order-independent, deterministic);
//    then the matrix waves convert their accumulators to the same fixed point (v_cvt_pknorm_u16_f32), add the LDS tile and compare
//    with the query's running threshold; survivors go to the fused top-k machinery (topk.hip) under the 16-bit key.
//    Per query the scale maps [0, sum_t q_t * max_t(v)] onto [0, 0.98 * 65 535], so a key cannot overflow its half word.
//  stage 2, certificate + exact re-score.  With true_fix = 65 535 * s_q * (real-arithmetic score) the key obeys
//        true_fix (1 - dd) - 1.2  <=  key  <=  true_fix (1 + dd) + 1.2 + 1.01 n_rare          (dd: fp16 rounding of both operands)
//    and the reference's fp32 chain is within (n_terms + 2) 2^-24 of the real-arithmetic score (all terms >= 0).  The k-th largest
//    key therefore gives a LOWER bound on the k-th largest reference score, and every doc whose UPPER bound stays below it is
//    out.  The top (k + 1 024) keys are kept; if the band of keys that cannot be excluded lies inside them the query is
//    certified, its candidates (k + a few hundred) are re-scored by cert_rescore_kernel with the reference's exact chain from
//    a doc-major forward index, and the exact top-k of those is the result - bit for bit what the exact kernels return.
//    Anything else - a query outside the preconditions (negative / unordered / too many terms), a band that does not fit,
//    fewer than k docs with a non-zero key, an overflowing candidate buffer - is flagged and served by the exact kernels.
#include "sparse_index.h"
#include <algorithm>
#include <math.h>
#include <vector>

#define SC_DT 1024                    // docs per tile
#define SC_QB 32                      // queries per workgroup (one MFMA N block)
#define SC_MB (SC_DT / 32)            // 32-doc M blocks per tile
#define SC_PITCH_W (SC_DT / 2 + 2)    // 32-bit words per query row of the LDS tile (two 16-bit slots per word; + 2: bank spread)
#define SC_MAXR 64                    // rare terms per group: one lane of a scatter wave each
#define SC_MAXG 4                     // groups per query: the first runs through the staged, flattened walk, the others (queries with more than 64
                                      // rare terms) through a plain per-lane walk in the same step
#define SC_MAXRT (SC_MAXR * SC_MAXG)  // rare terms per query the scatter role can add
#define SC_BAND 1024                  // keys kept beyond k
#define SC_CAND_CAP 16384             // candidate slots per query and launch
#define SC_MAXQT 256                  // terms of a fast-path query
#define SC_TMAX 128                   // dense terms (MFMA K), at most: 8 k-steps of B fragments in registers, 8 KB of them in LDS
#define SC_FWD_MAX 1024               // postings per doc the forward-index sort handles
#ifndef SC_DIAG
#define SC_DIAG 0                     // timing-only variants (tools/micro/cert_diag.sh, wrong results): 1 no posting work, 2 no MFMA work, 4 no table lookups, 8 no LDS adds, 16 plain LDS read-modify-write, 32 no MFMA (operand loads stay), 64 no operand loads (MFMAs stay)
#endif

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x2_u __attribute__((ext_vector_type(2), aligned(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4_u __attribute__((ext_vector_type(4), aligned(4)));

struct SparseCert {
    int T = 0, KS = 0;                // dense terms (multiple of 16) and MFMA k-steps
    int n_tiles = 0;
    float vscale = 1.f;               // power of two applied to the values before the fp16 rounding
    int32_t* dslot = nullptr;         // [V] dense slot of a term, -1 = rare
    float* vmax = nullptr;            // [V] largest value of a posting list
    _Float16* d16 = nullptr;          // [n_tiles][SC_MB][KS][64][8]: MFMA A fragments (row = doc, k = dense slot); a matrix wave streams 4 KS KB per tile
    uint32_t* P = nullptr;            // [nnz] packed postings: (doc % SC_DT / 2) * 4 | doc % 2  (byte offset of the doc's LDS word | its half) | fp16(v * vscale) << 16
    uint32_t* S = nullptr;            // [V][n_tiles + 1]: first posting (absolute index) of term t with doc >= tile * SC_DT
    uint32_t* E = nullptr;            // [V][e_stride]: the run of term t inside tile i in one word: 0 = empty, a packed posting | 2 = its only
                                      // posting, 0xffff0000 | length = two or more postings (in P from the running start on)
    int e_stride = 0;                 // n_tiles rounded up to 4, + 4: 16-byte rows, a 16-byte read never leaves its row
    int64_t* fwd_indptr = nullptr;    // doc-major forward index, terms ascending inside a doc
    uint64_t* fwd_tv = nullptr;       // [nnz] term << 32 | value bits: one 16-byte load per lane brings two postings of a row
    // per-call plan (grown on demand)
    int64_t nq_cap = 0;
    _Float16* bfrag = nullptr;        // [nq_pad / 32][KS][64][8]: MFMA B fragments (k = dense slot, col = query)
    int32_t* rare_term = nullptr;     // [nq_pad][SC_MAXRT], -1 = none
    float* rare_w = nullptr;          // [nq_pad][SC_MAXRT]: q_t * s_q * 65535 / vscale
    float* cq = nullptr;              // [nq_pad]: MFMA sum -> [0, 1]
    float* sq = nullptr;              // [nq_pad]: score -> [0, 0.98]
    int32_t* n_rare = nullptr;        // [nq_pad]
    int32_t* n_qt = nullptr;          // [nq_pad]
    int32_t* n_drop = nullptr;        // [nq_pad] rare terms left out of stage 1 (weight below fp16's normal range)
    float* tau2 = nullptr;            // [nq_pad] k-th best key so far (topk_compact2), 0 until a select has run: the band below it is the filter threshold
    uint8_t* elig = nullptr;          // [nq_pad]
    uint8_t* overflow = nullptr;      // [nq_pad]
    int32_t* m_count = nullptr;       // [nq_pad] candidates to re-score
    int* d_n_uncert = nullptr;
    uint8_t* d_uncert = nullptr;      // [uncert_cap] flags of a search's batch (kept: a hipMalloc / hipFree pair per search synchronises the device)
    int64_t uncert_cap = 0;
    int64_t ap_cap = 0;               // approximate top lists [nq][k_eff]
    int64_t* ap_ids = nullptr;
    TopkWS ws;
    // statistics (sr_sparse_index_cert_stats)
    int64_t n_calls = 0, n_queries = 0, n_uncert = 0, n_rescored = 0;
    uint16_t* dump = nullptr;         // debug: [nq_pad][n_tiles * SC_DT] keys of the last search (sr_sparse_index_cert_debug)
    int64_t dump_nq = 0;
    bool want_dump = false;
    unsigned long long* d_stamps = nullptr;   // dev switch SR_CERT_STAMPS
};

// ------------------------------------------------------------------------------------------------------- build ---
// per term: largest value; flags |= 1 for a negative or non-finite value
__global__ __launch_bounds__(256) void cert_term_stats_kernel(const int64_t* __restrict__ indptr, const float* __restrict__ vals,
                                                              float* __restrict__ vmax, int* __restrict__ flags) {
    __shared__ float red[4];
    const int64_t t = blockIdx.x;
    const int64_t b = indptr[t], e = indptr[t + 1];
    float mx = 0.f;
    int bad = 0;
    for (int64_t p = b + threadIdx.x; p < e; p += 256) {
        const float v = vals[p];
        if (!(v >= 0.f) || !(v < 3.0e38f)) bad = 1;
        mx = fmaxf(mx, v);
    }
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    if (bad) atomicOr(flags, 1);
    __syncthreads();
    if (threadIdx.x == 0) vmax[t] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

__global__ void cert_fill_d16_kernel(const int64_t* __restrict__ indptr, const int32_t* __restrict__ doc_ids,
                                     const float* __restrict__ vals, const int32_t* __restrict__ slot_term, int KS, float vscale,
                                     _Float16* __restrict__ d16) {
    const int sl = blockIdx.y;
    const int64_t t = slot_term[sl];
    const int64_t b = indptr[t], e = indptr[t + 1];
    const int s = sl >> 4, kk = sl & 15;
    for (int64_t p = b + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < e; p += (int64_t)gridDim.x * blockDim.x) {
        const int64_t doc = doc_ids[p];
        const int64_t tile = doc / SC_DT;
        const int dl = (int)(doc - tile * SC_DT);
        const int mb = dl >> 5, r = dl & 31;
        const int lane = r + 32 * (kk >> 3);
        d16[((((tile * SC_MB + mb) * KS + s) * 64 + lane) << 3) + (kk & 7)] = (_Float16)(vals[p] * vscale);
    }
}

__global__ void cert_pack_postings_kernel(const int32_t* __restrict__ doc_ids, const float* __restrict__ vals, int64_t nnz, float vscale,
                                          uint32_t* __restrict__ P) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nnz) return;
    const _Float16 h = (_Float16)(vals[p] * vscale);
    const uint32_t dl = (uint32_t)doc_ids[p] & (SC_DT - 1);
    P[p] = ((dl >> 1) << 2) | (dl & 1u) | ((uint32_t)__builtin_bit_cast(unsigned short, h) << 16);
}

// S[t][i] = indptr[t] + #{postings of t with doc < i * SC_DT}, i = 0 .. n_tiles.  The thread of posting p (doc d, predecessor d')
// fills the tiles in (tile(d'), tile(d)]; the thread "behind the last posting" fills the rest.
__global__ __launch_bounds__(256) void cert_s_table_kernel(const int64_t* __restrict__ indptr, const int32_t* __restrict__ doc_ids, int n_tiles,
                                                           uint32_t* __restrict__ S) {
    const int64_t t = blockIdx.x;
    const int64_t b = indptr[t], e = indptr[t + 1];
    uint32_t* row = S + t * (int64_t)(n_tiles + 1);
    for (int64_t p = b + threadIdx.x; p <= e; p += 256) {
        const int prev = p > b ? (int)(doc_ids[p - 1] / SC_DT) : -1;
        const int cur = p < e ? (int)(doc_ids[p] / SC_DT) : n_tiles;
        for (int i = prev + 1; i <= cur; ++i) row[i] = (uint32_t)p;
    }
}

// E[t][i] from the table of starts and the packed postings (see SparseCert::E)
__global__ __launch_bounds__(256) void cert_e_table_kernel(const uint32_t* __restrict__ S, const uint32_t* __restrict__ P, int n_tiles, int e_stride,
                                                           uint32_t* __restrict__ E) {
    const int64_t t = blockIdx.x;
    const uint32_t* srow = S + t * (int64_t)(n_tiles + 1);
    uint32_t* erow = E + t * (int64_t)e_stride;
    for (int i = threadIdx.x; i < e_stride; i += 256) {
        uint32_t e = 0u;
        if (i < n_tiles) {
            const uint32_t b = srow[i], n = srow[i + 1] - b;
            if (n == 1u) e = P[b] | 2u;
            else if (n >= 2u) e = 0xffff0000u | (n < 0xffffu ? n : 0xffffu);
        }
        erow[i] = e;
    }
}

__global__ void cert_fwd_count_kernel(const int32_t* __restrict__ doc_ids, int64_t nnz, int32_t* __restrict__ cnt) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p < nnz) atomicAdd(&cnt[doc_ids[p]], 1);
}
__global__ void cert_i32_to_i64_kernel(const int32_t* __restrict__ in, int64_t n, int64_t* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i];
}
__global__ __launch_bounds__(256) void cert_fwd_fill_kernel(const int64_t* __restrict__ indptr, const int32_t* __restrict__ doc_ids,
                                                            const float* __restrict__ vals, const int64_t* __restrict__ fwd_indptr,
                                                            int32_t* __restrict__ cursor, uint64_t* __restrict__ fwd_tv) {
    const int64_t t = blockIdx.y;
    const int64_t b = indptr[t], e = indptr[t + 1];
    for (int64_t p = b + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < e; p += (int64_t)gridDim.x * blockDim.x) {
        const int32_t d = doc_ids[p];
        const int64_t pos = fwd_indptr[d] + atomicAdd(&cursor[d], 1);
        fwd_tv[pos] = ((uint64_t)(uint32_t)t << 32) | (uint64_t)__float_as_uint(vals[p]);
    }
}
// one wave per doc: bitonic sort of its (term, value) pairs by term in LDS; flags |= 2 for a doc with more than SC_FWD_MAX postings
__global__ __launch_bounds__(256) void cert_fwd_sort_kernel(const int64_t* __restrict__ fwd_indptr, int64_t n_docs, uint64_t* __restrict__ fwd_tv,
                                                            int* __restrict__ flags) {
    __shared__ uint64_t keys_all[4][SC_FWD_MAX];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t d = (int64_t)blockIdx.x * 4 + wave;
    if (d >= n_docs) return;
    const int64_t b = fwd_indptr[d];
    const int n = (int)(fwd_indptr[d + 1] - b);
    if (n <= 1) return;
    if (n > SC_FWD_MAX) {
        if (lane == 0) atomicOr(flags, 2);
        return;
    }
    uint64_t* keys = keys_all[wave];
    int P = 64;
    while (P < n) P <<= 1;
    for (int i = lane; i < P; i += 64)
        keys[i] = i < n ? fwd_tv[b + i] : ~0ull;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    for (int size = 2; size <= P; size <<= 1)
        for (int j = size >> 1; j > 0; j >>= 1) {
            for (int i = lane; i < P; i += 64) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const bool up = (i & size) == 0;
                    const uint64_t x = keys[i], y = keys[ixj];
                    if (up ? (x > y) : (x < y)) { keys[i] = y; keys[ixj] = x; }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
    for (int i = lane; i < n; i += 64) fwd_tv[b + i] = keys[i];
}

void sparse_cert_destroy(SparseCert* c) {
    if (!c) return;
    if (c->d_stamps) {          // diagnostic: mean cycles per tile step of the sampled waves
        unsigned long long h[80] = {0};
        if (hipMemcpy(h, c->d_stamps, sizeof(h), hipMemcpyDeviceToHost) == hipSuccess && h[0]) {
            const double n = (double)h[0];
            fprintf(stderr, "[cert stamps] %llu sampled tile steps of %.0f cycles: last matrix wave at the barrier after %.0f, last scatter wave after %.0f (scatter last in %.0f %% of the steps); first blocks of wave 1 with a candidate: %.0f %%\n",
                    h[0], (double)h[1] / n, (double)h[2] / n, (double)h[3] / n, 100.0 * (double)h[4] / n, 100.0 * (double)h[5] / n);
            fprintf(stderr, "[cert stamps]   matrix waves  : barrier arrival");
            for (int w = 0; w < 8; ++w) fprintf(stderr, " %.0f", (double)h[16 + w] / n);
            fprintf(stderr, " | end of the last MFMA chain");
            for (int w = 0; w < 8; ++w) fprintf(stderr, " %.0f", (double)h[32 + w] / n);
            fprintf(stderr, " | keys of the first block done");
            for (int w = 0; w < 8; ++w) fprintf(stderr, " %.0f", (double)h[48 + w] / n);
            fprintf(stderr, " | end of the first filter");
            for (int w = 0; w < 8; ++w) fprintf(stderr, " %.0f", (double)h[64 + w] / n);
            fprintf(stderr, "\n[cert stamps]   scatter waves : barrier arrival");
            for (int w = 8; w < 16; ++w) fprintf(stderr, " %.0f", (double)h[16 + w] / n);
            fprintf(stderr, " | end of the adds");
            for (int w = 8; w < 16; ++w) fprintf(stderr, " %.0f", (double)h[32 + w] / n);
            fprintf(stderr, "\n");
        }
        (void)hipFree(c->d_stamps);
    }
    void* ptrs[] = {c->dslot, c->vmax, c->d16, c->P, c->S, c->E, c->fwd_indptr, c->fwd_tv, c->bfrag, c->rare_term, c->rare_w,
                    c->cq, c->sq, c->n_rare, c->n_qt, c->n_drop, c->tau2, c->elig, c->overflow, c->m_count, c->d_n_uncert, c->d_uncert, c->ap_ids, c->dump};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    c->ws.release();
    delete c;
}

int sparse_cert_build(sr_sparse_index* idx, hipStream_t s) {
    // dev switch SR_SPARSE_CERT: 0 = never, 1 = whenever the index qualifies structurally (tests on small indexes);
    // default: collections of at least 64 tiles
    int mode = -1;
    if (const char* e = sr_dev_getenv("SR_SPARSE_CERT")) mode = atoi(e);
    // public switches (include/sr_hip.h): SR_SPARSE_SCORER=exact keeps the index without the scorer's side structures (they take up to half
    // of the free device memory: ~22 bytes per posting + 8 bytes per (term, doc tile)); SR_LOG=1 says on stderr what was built or why not
    const char* scorer = getenv("SR_SPARSE_SCORER");
    const bool log = getenv("SR_LOG") && atoi(getenv("SR_LOG")) != 0;
    if (scorer && strcmp(scorer, "exact") == 0) mode = 0;
    if (mode == 0 || (mode < 0 && idx->n_docs < 64 * SC_DT)) {
        if (log) fprintf(stderr, "[sr_hip] sparse index of %lld docs: exact kernels only (%s)\n", (long long)idx->n_docs,
                         mode == 0 ? "certified scorer switched off" : "fewer than 65 536 docs");
        return SR_OK;
    }
    std::vector<int64_t> h_indptr((size_t)idx->n_terms + 1);
    SR_CHECK_HIP(hipMemcpyAsync(h_indptr.data(), idx->indptr, sizeof(int64_t) * h_indptr.size(), hipMemcpyDeviceToHost, s));
    SR_CHECK_HIP(hipStreamSynchronize(s));
    const int64_t nnz = h_indptr.back() - h_indptr.front();
    if (nnz <= 0 || nnz >= 0xffffffffll || h_indptr.front() != 0) return SR_OK;
    const int64_t V = idx->n_terms, N = idx->n_docs;
    SparseCert* c = new SparseCert();
    c->n_tiles = (int)ceil_div64(N, SC_DT);
    int* d_flags = nullptr;
    int32_t *d_terms = nullptr, *d_cnt = nullptr;
    int64_t* d_cnt64 = nullptr;
    bool ok = false;        // false: the index does not qualify (or no memory): not an error
    int rc = SR_OK;
    do {
        if (hipMalloc((void**)&c->vmax, sizeof(float) * (size_t)V) != hipSuccess || hipMalloc((void**)&d_flags, sizeof(int)) != hipSuccess) break;
        if (hipMemsetAsync(d_flags, 0, sizeof(int), s) != hipSuccess) { rc = SR_ERR_HIP; break; }
        hipLaunchKernelGGL(cert_term_stats_kernel, dim3((unsigned)V), dim3(256), 0, s, idx->indptr, idx->vals, c->vmax, d_flags);
        std::vector<float> h_vmax((size_t)V);
        int h_flags = 0;
        if (hipMemcpyAsync(h_vmax.data(), c->vmax, sizeof(float) * (size_t)V, hipMemcpyDeviceToHost, s) != hipSuccess ||
            hipMemcpyAsync(&h_flags, d_flags, sizeof(int), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) {
            rc = SR_ERR_HIP;
            break;
        }
        if (h_flags & 1) break;                   // a negative / non-finite value: exact kernels only
        float vmax_all = 0.f;
        for (float v : h_vmax) vmax_all = std::max(vmax_all, v);
        if (!(vmax_all > 0.f)) break;
        {   // largest value lands in [2^13, 2^14): far from fp16's subnormals and from its maximum
            int ex;
            (void)frexpf(vmax_all, &ex);          // vmax_all = m * 2^ex, m in [0.5, 1)
            c->vscale = ldexpf(1.0f, 14 - ex);
        }
        // dense terms: the longest lists, those present in at least 1 / 1024 of the docs, at most T_max, a multiple of 16
        int t_max = 128;
        if (const char* e = sr_dev_getenv("SR_SPARSE_CERT_T")) t_max = atoi(e);
        t_max = std::max(16, std::min(SC_TMAX, t_max / 16 * 16));
        std::vector<std::pair<int64_t, int32_t>> heavy;
        for (int64_t t = 0; t < V; ++t) {
            const int64_t len = h_indptr[(size_t)t + 1] - h_indptr[(size_t)t];
            if (len > 0 && len * 1024 >= N) heavy.emplace_back(-len, (int32_t)t);
        }
        std::sort(heavy.begin(), heavy.end());
        int n_heavy = (int)std::min<size_t>(heavy.size(), (size_t)t_max);
        c->T = std::max(16, (n_heavy + 15) / 16 * 16);
        c->KS = c->T / 16;
        if (c->KS == 3) { c->KS = 4; c->T = 64; }                         // instantiated k-step counts: 1, 2, 4 .. 8
        std::vector<int32_t> h_slot((size_t)V, -1), h_terms((size_t)std::max(1, n_heavy));
        for (int i = 0; i < n_heavy; ++i) {
            h_slot[(size_t)heavy[(size_t)i].second] = i;
            h_terms[(size_t)i] = heavy[(size_t)i].second;
        }
        const size_t d16_halves = (size_t)c->n_tiles * (size_t)c->T * SC_DT;
        const size_t s_words = (size_t)V * (size_t)(c->n_tiles + 1);
        c->e_stride = (c->n_tiles + 3) / 4 * 4 + 4;
        const size_t e_words = (size_t)V * (size_t)c->e_stride;
        if (e_words >= 0xffffffffull) break;      // the kernel addresses table E with 32-bit word offsets
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { rc = SR_ERR_HIP; break; }
        const size_t need = d16_halves * 2 + (size_t)nnz * 12 + (s_words + e_words) * 4 + (size_t)N * 16 + (64u << 20);
        if (need > free_b / 2) {                  // side structures may take at most half of what is free
            if (log) fprintf(stderr, "[sr_hip] sparse index: the certified scorer needs %.2f GB, more than half of the %.2f GB free: exact kernels only\n",
                             (double)need / 1e9, (double)free_b / 1e9);
            break;
        }
        if (log) fprintf(stderr, "[sr_hip] sparse index of %lld docs, %lld postings: certified scorer with %d heavy terms, %.2f GB of side structures "
                                 "(SR_SPARSE_SCORER=exact to do without)\n", (long long)N, (long long)nnz, c->T, (double)need / 1e9);
        if (hipMalloc((void**)&c->dslot, sizeof(int32_t) * (size_t)V) != hipSuccess || hipMalloc((void**)&c->d16, d16_halves * 2) != hipSuccess ||
            hipMalloc((void**)&c->P, sizeof(uint32_t) * ((size_t)nnz + 4)) != hipSuccess || hipMalloc((void**)&c->S, s_words * 4) != hipSuccess ||
            hipMalloc((void**)&c->E, e_words * 4) != hipSuccess ||
            hipMalloc((void**)&c->fwd_indptr, sizeof(int64_t) * (size_t)(N + 1)) != hipSuccess ||
            hipMalloc((void**)&c->fwd_tv, sizeof(uint64_t) * ((size_t)nnz + 2)) != hipSuccess ||
            hipMalloc((void**)&d_terms, sizeof(int32_t) * h_terms.size()) != hipSuccess ||
            hipMalloc((void**)&d_cnt, sizeof(int32_t) * (size_t)N) != hipSuccess || hipMalloc((void**)&d_cnt64, sizeof(int64_t) * (size_t)N) != hipSuccess) {
            (void)hipGetLastError();
            break;
        }
        rc = SR_ERR_HIP;
        if (hipMemcpyAsync(c->dslot, h_slot.data(), sizeof(int32_t) * (size_t)V, hipMemcpyHostToDevice, s) != hipSuccess) break;
        if (hipMemcpyAsync(d_terms, h_terms.data(), sizeof(int32_t) * h_terms.size(), hipMemcpyHostToDevice, s) != hipSuccess) break;
        if (hipMemsetAsync(c->d16, 0, d16_halves * 2, s) != hipSuccess) break;
        if (hipMemsetAsync(d_cnt, 0, sizeof(int32_t) * (size_t)N, s) != hipSuccess) break;
        if (n_heavy > 0)
            hipLaunchKernelGGL(cert_fill_d16_kernel, dim3(256, (unsigned)n_heavy), dim3(256), 0, s, idx->indptr, idx->doc_ids, idx->vals, d_terms,
                               c->KS, c->vscale, c->d16);
        hipLaunchKernelGGL(cert_pack_postings_kernel, dim3((unsigned)ceil_div64(nnz, 256)), dim3(256), 0, s, idx->doc_ids, idx->vals, nnz,
                           c->vscale, c->P);
        hipLaunchKernelGGL(cert_s_table_kernel, dim3((unsigned)V), dim3(256), 0, s, idx->indptr, idx->doc_ids, c->n_tiles, c->S);
        if (hipMemsetAsync(c->P + nnz, 0, 4 * sizeof(uint32_t), s) != hipSuccess) break;       // an 8-byte read of the last posting stays inside
        hipLaunchKernelGGL(cert_e_table_kernel, dim3((unsigned)V), dim3(256), 0, s, c->S, c->P, c->n_tiles, c->e_stride, c->E);
        hipLaunchKernelGGL(cert_fwd_count_kernel, dim3((unsigned)ceil_div64(nnz, 256)), dim3(256), 0, s, idx->doc_ids, nnz, d_cnt);
        hipLaunchKernelGGL(cert_i32_to_i64_kernel, dim3((unsigned)ceil_div64(N, 256)), dim3(256), 0, s, d_cnt, N, d_cnt64);
        if (hipGetLastError() != hipSuccess) break;
        if (sr_device_exclusive_scan_i64(d_cnt64, c->fwd_indptr, N, s) != SR_OK) break;
        if (hipMemsetAsync(d_cnt, 0, sizeof(int32_t) * (size_t)N, s) != hipSuccess) break;
        hipLaunchKernelGGL(cert_fwd_fill_kernel, dim3(64, (unsigned)V), dim3(256), 0, s, idx->indptr, idx->doc_ids, idx->vals, c->fwd_indptr, d_cnt,
                           c->fwd_tv);
        hipLaunchKernelGGL(cert_fwd_sort_kernel, dim3((unsigned)ceil_div64(N, 4)), dim3(256), 0, s, c->fwd_indptr, N, c->fwd_tv, d_flags);
        if (hipGetLastError() != hipSuccess) break;
        if (hipMemcpyAsync(&h_flags, d_flags, sizeof(int), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) break;
        rc = SR_OK;
        if (h_flags & 2) break;                   // a doc with more postings than the forward sort handles
        ok = true;
    } while (0);
    if (d_flags) (void)hipFree(d_flags);
    if (d_terms) (void)hipFree(d_terms);
    if (d_cnt) (void)hipFree(d_cnt);
    if (d_cnt64) (void)hipFree(d_cnt64);
    if (rc != SR_OK) {
        sr_set_error("sr_sparse_index_create: building the certified scorer's structures failed: %s", hipGetErrorString(hipGetLastError()));
        sparse_cert_destroy(c);
        return rc;
    }
    if (!ok) {
        (void)hipGetLastError();
        sparse_cert_destroy(c);
        return SR_OK;
    }
    idx->cert = c;
    return SR_OK;
}

// -------------------------------------------------------------------------------------------------------- plan ---
// One wave per query: preconditions, scale, MFMA B fragment entries of its dense terms, rare term list.
struct CertPlanArgs {
    const int64_t* q_indptr;
    const int32_t* q_cols;
    const float* q_vals;
    int64_t nq, n_terms;
    const int64_t* indptr;
    const int32_t* dslot;
    const float* vmax;
    int KS;
    float vscale;
    _Float16* bfrag;
    int32_t* rare_term;
    float* rare_w;
    float* cq;
    float* sq;
    int32_t* n_rare;
    int32_t* n_qt;
    int32_t* n_drop;
    uint8_t* elig;
    uint8_t* overflow;
    int* max_rare;           // [0] largest rare-term count among the queries on the fast path (sizes the certificate's band), [1] how many queries are on it
};

__global__ __launch_bounds__(256) void cert_plan_kernel(CertPlanArgs a) {
    const int lane = threadIdx.x & 63;
    const int64_t q = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nq_pad = (a.nq + SC_QB - 1) / SC_QB * SC_QB;
    if (q >= nq_pad) return;
    // defaults: a query outside the fast path contributes nothing to stage 1
#pragma unroll
    for (int g = 0; g < SC_MAXG; ++g) {
        a.rare_term[q * SC_MAXRT + g * SC_MAXR + lane] = -1;
        a.rare_w[q * SC_MAXRT + g * SC_MAXR + lane] = 0.f;
    }
    if (lane == 0) { a.cq[q] = 0.f; a.sq[q] = 0.f; a.n_rare[q] = 0; a.n_qt[q] = 0; a.n_drop[q] = 0; a.elig[q] = 0; a.overflow[q] = 0; }
    if (q >= a.nq) return;
    const int64_t tb = a.q_indptr[q], te = a.q_indptr[q + 1];
    const int n = (int)(te - tb);
    if (n <= 0 || n > SC_MAXQT) return;
    bool ok = true;
    int term[SC_MAXQT / 64], slot[SC_MAXQT / 64];
    float val[SC_MAXQT / 64], vmx[SC_MAXQT / 64];
    float tot = 0.f, dmax = 0.f, dmin = 3.0e38f;
    int nr = 0, nd = 0;
#pragma unroll
    for (int c = 0; c < SC_MAXQT / 64; ++c) {
        const int i = c * 64 + lane;
        term[c] = -1; slot[c] = -1; val[c] = 0.f; vmx[c] = 0.f;
        if (i < n) {
            const int t = a.q_cols[tb + i];
            const float v = a.q_vals[tb + i];
            if (i > 0 && a.q_cols[tb + i - 1] >= t) ok = false;          // the union order must be the query's own (strictly ascending)
            if (!(v >= 0.f) || !(v < 3.0e38f)) ok = false;
            if (t >= 0 && (int64_t)t < a.n_terms && v > 0.f && a.indptr[t + 1] > a.indptr[t]) {
                term[c] = t; val[c] = v; vmx[c] = a.vmax[t]; slot[c] = a.dslot[t];
                tot += v * vmx[c];
                if (slot[c] >= 0) { dmax = fmaxf(dmax, v); dmin = fminf(dmin, v); ++nd; } else ++nr;
            }
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        tot += __shfl_xor(tot, off);
        dmax = fmaxf(dmax, __shfl_xor(dmax, off));
        dmin = fminf(dmin, __shfl_xor(dmin, off));
        nr += __shfl_xor(nr, off);
        nd += __shfl_xor(nd, off);
    }
    ok = __ballot(!ok) == 0;
    if (!ok || nr > SC_MAXRT || !(tot > 0.f) || !(tot < 1.0e30f)) return;
    const float sq = 0.98f / (tot * 1.0001f);               // every real-arithmetic score * sq <= 0.98
    // dense operand: q_t * sq * 2^e with the largest one in [2^13, 2^14); all of them normal fp16 numbers
    float two_e = 1.f;
    if (nd > 0) {
        int ex;
        (void)frexpf(dmax * sq, &ex);
        two_e = ldexpf(1.0f, 14 - ex);
        if (!(dmin * sq * two_e >= 6.2e-5f)) return;       // a term 2^27 below the largest: exact kernels
    }
    const float cqv = 1.0f / (two_e * a.vscale);
    // flushed / subnormal fp16 values of the doc side cost at most 65535 * cq * nd * 2^14 * 2^-14 key units: must stay below 0.1
    if (!(65535.0f * cqv * (float)nd <= 0.1f)) return;
    const int64_t qb = q / SC_QB;
    const int qn = (int)(q % SC_QB);
    // A rare term's weight is kept as fp16 by the scatter waves.  One below fp16's normal range is LEFT OUT of stage 1 (its postings
    // are worth less than 32768 * 2^-14 = 2 key units each; the certificate widens by 2.03 per such term), one above it: exact kernels.
    int rbase = 0, n_drop = 0;
    bool w_bad = false;
#pragma unroll
    for (int c = 0; c < SC_MAXQT / 64; ++c) {
        bool is_r = term[c] >= 0 && slot[c] < 0;
        const float wr = is_r ? (float)(_Float16)(val[c] * sq * 65535.0f / a.vscale) : 1.f;
        if (is_r && !(wr < 6.0e4f)) w_bad = true;
        const bool dropped = is_r && wr < 6.2e-5f;
        n_drop += __popcll(__ballot(dropped));
        is_r = is_r && !dropped;
        const uint64_t m = __ballot(is_r);
        if (term[c] >= 0 && slot[c] >= 0) {
            const int s = slot[c] >> 4, kk = slot[c] & 15;
            a.bfrag[(((qb * a.KS + s) * 64 + qn + 32 * (kk >> 3)) << 3) + (kk & 7)] = (_Float16)(val[c] * sq * two_e);
        }
        if (is_r) {
            const int j = rbase + __popcll(m & ((1ull << lane) - 1ull));
            a.rare_term[q * SC_MAXRT + j] = term[c];
            a.rare_w[q * SC_MAXRT + j] = wr;
        }
        rbase += __popcll(m);
    }
    if (__ballot(w_bad)) return;            // elig stays 0; what stage 1 computes for the query is discarded
    nr = rbase;
    if (lane == 0) a.n_drop[q] = n_drop;
    if (lane == 0) { a.cq[q] = cqv; a.sq[q] = sq; a.n_rare[q] = nr; a.n_qt[q] = n; a.elig[q] = 1; }
    if (lane == 0 && nr > 64) atomicMax(a.max_rare, nr);
    if (lane == 0) atomicAdd(a.max_rare + 1, 1);         // queries on the fast path
}

// ------------------------------------------------------------------------------------------------- score kernel ---
// The cut on stage-1 keys that follows from K = the k-th best key (of all docs, or of the docs seen so far): a doc whose key lies under it
// cannot be among the k best reference scores (the bound of DESIGN.md 4.6).  Monotone in K, so the cut from the k-th best key SO FAR
// is a valid filter threshold while the scan runs, and it is what cert_select_kernel applies to the final list.  < 1: no cut.
__device__ __forceinline__ double cert_cut_from_kth(double Kk, int T, int n_rare, int n_qt, int n_drop) {
    // relative error of the MFMA part: two fp16 roundings (2^-11 each, and their product), fp32 accumulation over T terms, the
    // scaling multiply; gamma: the reference's fp32 chain against real arithmetic
    const double dd = 2.0 * 4.8828125e-4 + 2.4e-7 + (double)T * 2.4e-7 + 1.0e-5;
    const double gamma = ((double)n_qt + 2.0) * 6.0e-8 * 1.01;
    const double lb = (Kk - 1.2 - 1.01 * (double)n_rare) / (1.0 + dd);           // lower bound of the k-th true_fix
    // a doc's true_fix is at most (key + 1.2) / (1 - dd) + 2.03 per rare term that stage 1 left out
    return floor((lb * (1.0 - gamma) / (1.0 + gamma) - 2.03 * (double)n_drop) * (1.0 - dd) - 1.2) - 1.0;
}

struct CertArgs {
    const f16x8* d16;
    const uint32_t* P;
    const uint32_t* S;
    int s_stride;            // n_tiles + 1
    const uint32_t* E;
    int e_stride;
    const f16x8* bfrag;
    const int32_t* rare_term;
    const float* rare_w;
    const float* cq;
    const float* tau;        // (k + band)-th best key so far, -inf until that many are held
    const float* tau2;       // k-th best key so far (0: not known yet); with n_rare / n_qt / n_drop / T it gives the band cut, usually the tighter threshold
    const int32_t* n_rare;
    const int32_t* n_qt;
    const int32_t* n_drop;
    int T;
    int64_t nq;
    uint64_t* cand_keys;
    int* cand_count;
    int64_t cand_cap;
    uint8_t* overflow;
    int tile_begin, n_tiles_launch, tiles_per_wg, n_qblocks;
    uint16_t* dump;          // debug: [nq_pad][dump_stride] keys
    int stamps_from;              // SR_CERT_STAMPS=<first tile>: only launches that begin at this tile or later are sampled
    unsigned long long* stamps;   // dev switch SR_CERT_STAMPS: [8] cycle sums per phase of sampled waves
    int64_t dump_stride;
};

// low half of a packed posting: byte offset of the doc's word inside a query row of the LDS tile | doc parity (SC_POST_LO)
__device__ __forceinline__ void cert_add_posting(uint32_t* row, uint32_t p, float w) {
    const float v = (float)__builtin_bit_cast(_Float16, (unsigned short)(p >> 16));
    const uint32_t c = (uint32_t)(v * w + 1.0f);                 // in (x, x + 1]: rounded up, never below the real contribution
    uint32_t* const wp = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(row) + (p & 0xfffcu));
    if (SC_DIAG & 8) { if (c == 0xffffffffu) *wp = c; }          // timing only: no LDS traffic
    else if (SC_DIAG & 16) *wp = *wp + (c << ((p & 1u) * 16u));  // timing only: plain read-modify-write
    else atomicAdd(wp, c << ((p & 1u) * 16u));
}

// wave-wide inclusive scans on the DPP row shifts / row broadcasts of gfx9 (6 VALU instructions, no LDS)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int sc_dpp(int src) { return __builtin_amdgcn_update_dpp(0, src, CTRL, ROW_MASK, 0xf, false); }
__device__ __forceinline__ int sc_scan_add(int v) {
    v += sc_dpp<0x111, 0xf>(v);      // row_shr:1
    v += sc_dpp<0x112, 0xf>(v);      // row_shr:2
    v += sc_dpp<0x114, 0xf>(v);      // row_shr:4
    v += sc_dpp<0x118, 0xf>(v);      // row_shr:8
    v += sc_dpp<0x142, 0xa>(v);      // row_bcast:15 into rows 1 and 3
    v += sc_dpp<0x143, 0xc>(v);      // row_bcast:31 into rows 2 and 3
    return v;
}
__device__ __forceinline__ int sc_scan_max(int v) {             // values >= 0
    v = max(v, sc_dpp<0x111, 0xf>(v));
    v = max(v, sc_dpp<0x112, 0xf>(v));
    v = max(v, sc_dpp<0x114, 0xf>(v));
    v = max(v, sc_dpp<0x118, 0xf>(v));
    v = max(v, sc_dpp<0x142, 0xa>(v));
    v = max(v, sc_dpp<0x143, 0xc>(v));
    return v;
}

#ifndef SC_STAMPS
#define SC_STAMPS 0                   // 1: the per-role cycle stamps (dev switch SR_CERT_STAMPS) are compiled in; they cost registers, so the product build leaves them out
#endif
#ifndef SC_DP
#define SC_DP 8                       // A fragments in flight per matrix wave
#endif
#define SC_ITERS 8                    // 64-lane steps (FOUR postings per lane) of a scatter wave's 4 items whose loads are issued a tile ahead, in registers
#define SC_QUADS (SC_ITERS * 64)      // quads staged per wave and tile; also the 16-bit entries of a wave's mark buffer
#define SC_WAVE_LDS (SC_QUADS * 2 + 256 * 4 + 256 * 2)   // bytes per scatter wave: marks, delta table, weight table
#define SC_SLOT_WORDS (SC_QB * SC_PITCH_W)
#define SC_BREGION 8192               // bytes: the query block's B fragments until they sit in registers (1 KB per k-step)
#define SC_TAIL_LDS (8 * SC_WAVE_LDS)  // the scatter waves' buffers

// The quads of a scatter wave's 4 items beyond the SC_QUADS staged ones: further windows of SC_QUADS through the same flat walk, loads
// and adds back to back (one memory round trip per 2 048 postings).  Not inlined: the kernel's register allocation is sized for the
// staged path, this one saves and restores around itself and runs once in about a hundred wave-steps at the MSMARCO shape.
__device__ __noinline__ void cert_overflow_windows(unsigned short* mark, const uint32_t* tab_delta, const _Float16* tab_w, const uint32_t* P,
                                                   uint32_t* wrow, uint32_t e0, uint32_t e1, uint32_t e2, uint32_t e3, int Ltot) {
    const int lane = threadIdx.x & 63;
    const uint32_t ev[4] = {e0, e1, e2, e3};
    int nq4[4], bq4[4], lnv[4];
    int base = 0;
#pragma unroll
    for (int qi = 0; qi < 4; ++qi) {
        lnv[qi] = (ev[qi] >> 16) == 0xffffu ? (int)(ev[qi] & 0xffffu) : 0;
        nq4[qi] = (lnv[qi] + 3) >> 2;
        const int es = sc_scan_add(nq4[qi]);
        bq4[qi] = base + es - nq4[qi];
        base += __builtin_amdgcn_readlane(es, 63);
    }
    for (int win = SC_QUADS; win < Ltot; win += SC_QUADS) {
#pragma unroll
        for (int it = 0; it < SC_ITERS; ++it) mark[it * 64 + lane] = 0;
        int carry = 0;                                           // id + 1 of the run that straddles the window's first quad
#pragma unroll
        for (int qi = 0; qi < 4; ++qi) {
            const int first = bq4[qi] - win, last = first + nq4[qi] - 1;
            const int id1 = qi * 64 + lane + 1;
            if (nq4[qi] > 0) {
                if (first >= 0 && first < SC_QUADS) mark[first] = (unsigned short)id1;
                if (last >= 0 && last < SC_QUADS) {
                    if (last == first) mark[last] = (unsigned short)(id1 | ((lnv[qi] & 3) << 9));
                    else mark[last] = (unsigned short)((lnv[qi] & 3) << 9);
                }
            }
            const uint64_t strad = __ballot(nq4[qi] > 0 && first < 0 && last >= 0);
            if (strad) carry = qi * 64 + __builtin_ctzll(strad) + 1;
        }
        const int nwin = Ltot - win < SC_QUADS ? Ltot - win : SC_QUADS;
        int mk[SC_ITERS], idw[SC_ITERS];
#pragma unroll
        for (int it = 0; it < SC_ITERS; ++it) mk[it] = (int)mark[it * 64 + lane];
#pragma unroll
        for (int it = 0; it < SC_ITERS; ++it) {
            const int v = sc_scan_max(max(mk[it] & 0x1ff, carry));
            carry = __builtin_amdgcn_readlane(v, 63);
            idw[it] = (v - 1) & 255;
        }
        uint32_t dlw[SC_ITERS];
#pragma unroll
        for (int it = 0; it < SC_ITERS; ++it) dlw[it] = tab_delta[idw[it]];
        uint4 pw[SC_ITERS];
#pragma unroll
        for (int it = 0; it < SC_ITERS; ++it) {                  // the window's loads go out together
            const uint32_t pi = it * 64 + lane < nwin ? dlw[it] + 4u * (uint32_t)(win + it * 64 + lane) : 0u;
            const u32x4_u v4 = *reinterpret_cast<const u32x4_u*>(P + pi);
            pw[it] = make_uint4(v4.x, v4.y, v4.z, v4.w);
        }
#pragma unroll
        for (int it = 0; it < SC_ITERS; ++it)
            if (it * 64 + lane < nwin) {
                uint32_t* row = wrow + (idw[it] >> 6) * SC_PITCH_W;
                const float wv = (float)tab_w[idw[it]];
                const int tail = mk[it] >> 9;
                cert_add_posting(row, pw[it].x, wv);
                if (tail != 1) cert_add_posting(row, pw[it].y, wv);
                if (tail == 0 || tail == 3) cert_add_posting(row, pw[it].z, wv);
                if (tail == 0) cert_add_posting(row, pw[it].w, wv);
            }
    }
}

// The rare terms of a query beyond its first 64 (group g >= 1, lane j = rare term 64 g + j), for the wave's 4 queries and one tile: the term,
// its E word and, for a run, its start in table S are looked up on the spot and a lane walks its run posting by posting.  Not inlined, like
// the overflow path above: the kernel's registers are sized for the staged walk.
__device__ __noinline__ void cert_extra_groups(const int32_t* rare_term, const float* rare_w, const uint32_t* E, const uint32_t* S, const uint32_t* P,
                                               int e_stride, int s_stride, int64_t q0, int ng0, int ng1, int ng2, int ng3, int tile, uint32_t* wrow) {
    const int lane = threadIdx.x & 63;
    const int ngq[4] = {ng0, ng1, ng2, ng3};
    constexpr int XG = SC_MAXG - 1;
    // every look-up level for all (query, group) pairs of the wave at once - 12 independent loads in flight per level instead of 12 dependent
    // chains of three (a step of a block of 256-term queries took 7 x a plain step when the chains ran one after the other)
    int32_t t[4][XG];
    uint32_t e[4][XG], p0[4][XG];
    float w[4][XG];
#pragma unroll
    for (int qi = 0; qi < 4; ++qi)
#pragma unroll
        for (int g = 0; g < XG; ++g)
            t[qi][g] = (g + 1 < ngq[qi] && !(SC_DIAG & 1)) ? rare_term[(q0 + qi) * SC_MAXRT + (g + 1) * SC_MAXR + lane] : -1;
#pragma unroll
    for (int qi = 0; qi < 4; ++qi)
#pragma unroll
        for (int g = 0; g < XG; ++g)
            e[qi][g] = t[qi][g] >= 0 ? E[(uint32_t)t[qi][g] * (uint32_t)e_stride + (uint32_t)tile] : 0u;
#pragma unroll
    for (int qi = 0; qi < 4; ++qi)
#pragma unroll
        for (int g = 0; g < XG; ++g) {
            w[qi][g] = e[qi][g] != 0u ? (float)(_Float16)rare_w[(q0 + qi) * SC_MAXRT + (g + 1) * SC_MAXR + lane] : 0.f;
            p0[qi][g] = (e[qi][g] >> 16) == 0xffffu ? S[(int64_t)t[qi][g] * s_stride + tile] : 0u;
        }
#pragma unroll
    for (int qi = 0; qi < 4; ++qi) {
        uint32_t* const row = wrow + qi * SC_PITCH_W;
#pragma unroll
        for (int g = 0; g < XG; ++g) {
            if (g + 1 >= ngq[qi]) continue;                      // wave-uniform
            const uint32_t ev = e[qi][g];
            if (ev != 0u && (ev >> 16) != 0xffffu) cert_add_posting(row, ev, w[qi][g]);      // the term's only posting in this tile
            const uint32_t len = (ev >> 16) == 0xffffu ? (ev & 0xffffu) : 0u;
            uint32_t longest = len;                              // the walk runs as long as the wave's longest run, four postings per round trip
            for (int off = 32; off > 0; off >>= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)longest, off); longest = o > longest ? o : longest; }
            for (uint32_t u = 0; u < longest; u += 4) {
                uint32_t pv[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) pv[r] = u + r < len ? P[p0[qi][g] + u + r] : 0u;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (u + r < len) cert_add_posting(row, pv[r], w[qi][g]);
            }
        }
    }
}

#if SC_STAMPS
#define SC_STAMP(ph) do { if (st_wg && lane == 0) t_ph[(ph) * 16 + wave] = (uint32_t)__builtin_readcyclecounter(); } while (0)
#else
#define SC_STAMP(ph) do { } while (0)
#endif

constexpr int sc_ring_depth(int nl) {
    int d = nl < SC_DP ? nl : SC_DP;
    while (nl % d) --d;
    return d;
}

template <int KS>
__global__ __launch_bounds__(1024) void cert_score_kernel(CertArgs a) {
    extern __shared__ uint32_t slots[];                          // 2 x [SC_QB][SC_PITCH_W] | B fragments | mark buffers
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // workgroups of one doc chunk share blockIdx.x % 8, i.e. one XCD under round-robin placement: the chunk's tiles, postings
    // and table rows are pulled into ONE L2 (speed only, never correctness)
    const int L = (int)blockIdx.x;
    const int xcd = L & 7, r = L >> 3;
    const int qb = r % a.n_qblocks, cg = r / a.n_qblocks;
    const int chunk = cg * 8 + xcd;
    const int n_chunks = (a.n_tiles_launch + a.tiles_per_wg - 1) / a.tiles_per_wg;
    if (chunk >= n_chunks) return;
    const int tile0 = a.tile_begin + chunk * a.tiles_per_wg;
    const int tile_end = a.tile_begin + a.n_tiles_launch;
    const int tile1 = tile0 + a.tiles_per_wg < tile_end ? tile0 + a.tiles_per_wg : tile_end;
    for (int i = tid; i < 2 * SC_SLOT_WORDS; i += 1024) slots[i] = 0u;
    // the query block's MFMA B fragments (k-step s: 64 lanes x 8 halves), read back per k-step: they would cost 4 KS VGPRs
    f16x8* const bl = reinterpret_cast<f16x8*>(slots + 2 * SC_SLOT_WORDS);
    if (tid < KS * 64) bl[tid] = a.bfrag[(int64_t)qb * KS * 64 + tid];
    // dev switch SR_CERT_STAMPS (compiled in with -DSC_STAMPS=1): every wave of a sampled workgroup leaves the time of its phase ends in LDS (one lane, one
    // 32-bit write each); matrix wave 0 adds them up in global memory after the step's barrier - nothing is carried in registers across the loop
    const bool st_wg = SC_STAMPS && a.stamps != nullptr && (blockIdx.x & 63) == 0 && a.tile_begin >= a.stamps_from;
    volatile uint32_t* const t_ph = reinterpret_cast<volatile uint32_t*>(reinterpret_cast<char*>(bl) + SC_BREGION + SC_TAIL_LDS);   // [4 phases][16 waves]
    __syncthreads();
    // The two roles work on DIFFERENT tiles: in step t the scatter waves add the rare postings of tile t into LDS tile t & 1 while
    // the matrix waves multiply tile t - 1, add LDS tile (t - 1) & 1 to it, filter and clear it.  One barrier per step.

    if (wave < 8) {
        // ---------------- matrix waves: M blocks 4 wave .. 4 wave + 3 of every tile, one after the other ----------------
        const int qn = lane & 31, h = lane >> 5;
        const int64_t q = (int64_t)qb * SC_QB + qn;
        const float cq = a.cq[q];
        const f32x2 cq2 = {cq, cq};
        float tq = q < a.nq ? a.tau[q] : INFINITY;
        if (q < a.nq && a.tau2[q] > 0.f) {
            const double cb = cert_cut_from_kth((double)a.tau2[q], a.T, a.n_rare[q], a.n_qt[q], a.n_drop[q]);
            if (cb >= 1.0 && (float)cb > tq) tq = (float)cb;
        }
        int cut = 1;                                             // keys of 0 are never candidates
        if (tq > 1.f) cut = tq >= 65536.f ? 65536 : (int)ceilf(tq);
        const uint32_t cutm1 = (uint32_t)(cut - 1);
        const uint32_t cutm1x2 = cutm1 | (cutm1 << 16);
        // The A fragments of a tile are a sequence of NL = 4 KS loads (block, k-step); DP of them are in flight
        // in a ring of registers at any time, across tile boundaries too - the matrix pipe never waits for a fresh round trip.
        constexpr int NL = 4 * KS;
        constexpr int DP = sc_ring_depth(NL);                    // the ring runs across tile boundaries: its depth divides NL
        // fragment i = (block, k-step) of the wave's tile slice lies i KB behind the slice's start: wave-uniform base (SGPRs) + 16 lane + immediate.
        // (Measured and dropped, both at the same time per tile step: a per-workgroup rotation of the fragment order, against 32 workgroups
        // of an XCD reading the same lines of a tile at the same time; an L2 warm-up of the tile three steps ahead by LDS-DMA loads.)
        const int wave_u = __builtin_amdgcn_readfirstlane(wave);
        auto a_ptr = [&](int tile, int i) -> const f16x8* {
            const char* base = reinterpret_cast<const char*>(a.d16) + (((int64_t)tile * SC_MB + wave_u * 4) * KS) * 1024;
            return reinterpret_cast<const f16x8*>(base + i * 1024 + lane * 16);
        };
        f16x8 af[DP];
#pragma unroll
        for (int i = 0; i < DP; ++i) af[i] = (SC_DIAG & 2) ? f16x8{} : *a_ptr(tile0, i);
        // The query block's B fragments stay in registers (4 KS VGPRs): read from LDS in front of every MFMA they cost an LDS round trip
        // per MFMA behind the scatter waves' atomics (5 000 of a tile step's 9 400 cycles, measured with the MFMAs and loads switched off)
        f16x8 bq[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) bq[ks] = bl[ks * 64 + lane];
        __syncthreads();                                         // step tile0: the scatter waves fill LDS tile tile0 & 1
        uint32_t st_0 = st_wg ? (uint32_t)__builtin_readcyclecounter() : 0u;      // start of the step (matrix wave 0's barrier exit)
        for (int tile = tile0; tile < tile1; ++tile) {
            const int tnext = tile + 1 < tile1 ? tile + 1 : tile;    // past the end: re-reads this tile (no branch around the loads)
            uint32_t* const buf = slots + (tile & 1) * SC_SLOT_WORDS;
#pragma unroll
            for (int mbi = 0; mbi < 4; ++mbi) {
                const int mb = mbi;
                f32x16 acc;
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
                for (int ksi = 0; ksi < KS; ++ksi) {
                    const int i = mbi * KS + ksi;
                    if (!(SC_DIAG & 2)) {
                        __builtin_amdgcn_sched_barrier(0);
                        if (!(SC_DIAG & 32)) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i % DP], bq[ksi], acc, 0, 0, 0);
                        else acc[0] += (float)af[i % DP][0];     // timing only: the load is still waited for
                        __builtin_amdgcn_sched_barrier(0);       // the ring's order is the point: hipcc otherwise re-packs the loads into one register and waits for each
                        if (!(SC_DIAG & 64)) af[i % DP] = i + DP < NL ? *a_ptr(tile, i + DP) : *a_ptr(tnext, i + DP - NL);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                if (mbi == 3) SC_STAMP(1);                       // end of the step's last MFMA chain
                // accumulators -> 16-bit fixed point, two docs per word: register pair (2 g, 2 g + 1) = rows 8 (g / 2) + 4 h + 2 (g % 2) + {0, 1};
                // + the LDS tile's fixed-point sums of the rare terms
                uint2 sv[4];
#pragma unroll
                for (int G = 0; G < 4; ++G) {
                    const int dl = (wave * 4 + mb) * 32 + 8 * G + 4 * h;         // 4 consecutive docs: one 8-byte read
                    uint32_t* sp = buf + qn * SC_PITCH_W + (dl >> 1);
                    sv[G] = *reinterpret_cast<const uint2*>(sp);
                    *reinterpret_cast<uint2*>(sp) = make_uint2(0u, 0u);
                }
                uint32_t key[8];
                uint32_t any = 0;
#pragma unroll
                for (int G = 0; G < 4; ++G) {
                    const f32x2 a0 = f32x2{acc[4 * G], acc[4 * G + 1]} * cq2, a1 = f32x2{acc[4 * G + 2], acc[4 * G + 3]} * cq2;   // v_pk_mul_f32
                    const u16x2 k0 = __builtin_amdgcn_cvt_pknorm_u16(a0[0], a0[1]) + __builtin_bit_cast(u16x2, sv[G].x);
                    const u16x2 k1 = __builtin_amdgcn_cvt_pknorm_u16(a1[0], a1[1]) + __builtin_bit_cast(u16x2, sv[G].y);
                    key[2 * G] = __builtin_bit_cast(uint32_t, k0);
                    key[2 * G + 1] = __builtin_bit_cast(uint32_t, k1);
                    any |= __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(k0, __builtin_bit_cast(u16x2, cutm1x2)));
                    any |= __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(k1, __builtin_bit_cast(u16x2, cutm1x2)));
                }
                if (a.dump) {
#pragma unroll
                    for (int g = 0; g < 8; ++g) {
                        const int64_t doc = (int64_t)tile * SC_DT + (wave * 4 + mb) * 32 + 8 * (g >> 1) + 4 * h + 2 * (g & 1);
                        *reinterpret_cast<uint32_t*>(a.dump + q * a.dump_stride + doc) = key[g];
                    }
                }
                if (mbi == 0) {
                    SC_STAMP(2);                                 // first block: keys done (LDS sums read and cleared), before the candidates
                    if (SC_STAMPS && st_wg && wave == 1 && __builtin_amdgcn_ballot_w64(any != 0) != 0 && lane == 0) atomicAdd(&a.stamps[5], 1ull);
                }
                if (SC_DIAG & 128) any = 0;                      // timing only: no candidates
                if (any != 0) {                                  // one block in two at the MSMARCO shape once the threshold has risen (46 % in the launches from tile 2 000 on)
                    int cnt = 0;
#pragma unroll
                    for (int g = 0; g < 8; ++g) cnt += ((key[g] & 0xffffu) > cutm1 ? 1 : 0) + ((key[g] >> 16) > cutm1 ? 1 : 0);
                    int pos = atomicAdd(&a.cand_count[q], cnt);
                    if ((int64_t)pos + cnt > a.cand_cap) a.overflow[q] = 1;
                    uint64_t* dst = a.cand_keys + q * a.cand_cap;
#pragma unroll
                    for (int g = 0; g < 8; ++g) {
                        const uint32_t doc = (uint32_t)tile * SC_DT + (uint32_t)((wave * 4 + mb) * 32 + 8 * (g >> 1) + 4 * h + 2 * (g & 1));
                        const uint32_t lo = key[g] & 0xffffu, hi = key[g] >> 16;
                        if (lo > cutm1) { if (pos < a.cand_cap) dst[pos] = sr_make_key((float)lo, doc); ++pos; }
                        if (hi > cutm1) { if (pos < a.cand_cap) dst[pos] = sr_make_key((float)hi, doc + 1u); ++pos; }
                    }
                }
                if (mbi == 0) SC_STAMP(3);                       // end of the first block's filter
            }
            SC_STAMP(0);
            __syncthreads();                                     // step tile + 1
            if (st_wg && wave == 0) {
                const uint32_t now = (uint32_t)__builtin_readcyclecounter();
                uint32_t d = t_ph[lane] - st_0;            // lane 16 p + w: phase p of wave w, cycles since the step began
                if ((int32_t)d < 0) d = 0u;                                   // a workgroup's last step: the scatter waves have no tile left (their figures come out low by 1 / steps per workgroup)
                atomicAdd(&a.stamps[16 + lane], (unsigned long long)d);
                uint32_t mx = lane < 16 ? d : 0u;                             // barrier arrival of the slowest wave per role
                for (int o = 1; o < 8; o <<= 1) { const uint32_t oth = (uint32_t)__shfl_xor((int)mx, o); mx = oth > mx ? oth : mx; }
                const uint32_t mm = (uint32_t)__shfl((int)mx, 0), ms = (uint32_t)__shfl((int)mx, 8);
                if (lane == 0) {
                    atomicAdd(&a.stamps[0], 1ull); atomicAdd(&a.stamps[1], (unsigned long long)(now - st_0));
                    atomicAdd(&a.stamps[2], (unsigned long long)mm); atomicAdd(&a.stamps[3], (unsigned long long)ms); atomicAdd(&a.stamps[4], ms > mm ? 1ull : 0ull);
                }
                st_0 = now;
            }
        }
    } else {
        // ---------------- scatter waves: queries 4 sw .. 4 sw + 3 of the block, lane j = rare term j ----------------
        // What a (query, tile) item adds: per rare term the run of its postings inside the tile.  Table E gives the run in ONE word per
        // (term, tile) - 16 bytes = 4 consecutive tiles per load, a cache line = 16 tiles: nothing, the single posting itself, or a
        // length (postings in P from the term's running start on; the start at the chunk's first tile comes from table S, once).
        //   single postings: lane j adds its own, one predicated LDS add per item;
        //   longer runs are walked FLATTENED, four postings (one 16-byte load) per lane: quad f of the item belongs to the run r with
        //   b_r <= f < b_r + nq_r (nq = quads of a run, b = exclusive prefix); r comes from a 16-bit array in LDS holding r + 1 at
        //   position b_r (bits 8..9 at a run's last quad: its postings if fewer than four) and a wave-wide running maximum - every lane
        //   busy whatever the run lengths are.
        // An item's loads are issued ONE TILE AHEAD (gathers over 4.5 GB take 2-3 us under load) and UNCONDITIONALLY (idle lanes read
        // entry 0): the number of vector-memory instructions per stage is a constant, so the compiler can wait for an item's loads with
        // a counted vmcnt that leaves everything issued after them in flight.
        const int sw = wave - 8;
        unsigned short* const mark = reinterpret_cast<unsigned short*>(reinterpret_cast<char*>(bl) + SC_BREGION + sw * SC_WAVE_LDS);
        int32_t term[4];
        uint32_t cur[4];            // running start (index into P) of every rare term's next run, at the tile being staged
#pragma unroll
        for (int qi = 0; qi < 4; ++qi) {
            const int64_t q = (int64_t)qb * SC_QB + sw * 4 + qi;
            term[qi] = a.rare_term[q * SC_MAXRT + lane];
            cur[qi] = term[qi] >= 0 ? a.S[(int64_t)term[qi] * a.s_stride + tile0] : 0u;
        }
        // the E words of tile pair g (tiles 2 g, 2 g + 1); pairs behind the chunk read the row's zero padding or a later chunk's words
        // (never used: the per-tile guard below)
        // row of the lane's term in table E as a 32-bit word offset (V * e_stride < 2^32 is checked at build time): wave-uniform base in
        // SGPRs + one VGPR offset per load, no 64-bit vector address held (or spilled) across the tile loop
        uint32_t erow[4];
#pragma unroll
        for (int qi = 0; qi < 4; ++qi) erow[qi] = (term[qi] >= 0 && !(SC_DIAG & 4)) ? (uint32_t)term[qi] * (uint32_t)a.e_stride : 0u;
        auto load_group = [&](int g, uint2 (&e2)[4]) {
#pragma unroll
            for (int qi = 0; qi < 4; ++qi) e2[qi] = *reinterpret_cast<const uint2*>(a.E + (erow[qi] + 2u * (uint32_t)g));
        };
        auto multi_len = [&](uint32_t e) -> int { return (e >> 16) == 0xffffu ? (int)(e & 0xffffu) : 0; };
        auto run_len = [&](uint32_t e) -> uint32_t { return (e >> 16) == 0xffffu ? (e & 0xffffu) : (e != 0u ? 1u : 0u); };
        auto add_quad = [&](uint32_t* row, const uint4& p4, int tail, float wv) {
            cert_add_posting(row, p4.x, wv);
            if (tail != 1) cert_add_posting(row, p4.y, wv);
            if (tail == 0 || tail == 3) cert_add_posting(row, p4.z, wv);
            if (tail == 0) cert_add_posting(row, p4.w, wv);
        };
        // E words: ecur = the pair of tiles holding the tile that is staged next; the pair behind it is loaded right after its second
        // tile was staged, a whole step (the adds of 4 items) before it is needed
        uint2 ecur[4];
        load_group(tile0 >> 1, ecur);
        if (tile0 & 1) {
#pragma unroll
            for (int qi = 0; qi < 4; ++qi) ecur[qi].x = ecur[qi].y;
        }
        // The quads of the wave's 4 items form ONE flat sequence (item 0's runs, then item 1's, ...): SC_QUADS = 64 SC_ITERS quads are
        // staged per step whatever their split over the items.  Per item the count varies a lot (mean ~70, one query in ten above 128
        // at the MSMARCO shape) and a step waits for its slowest wave, so a per-item capacity met its overflow path in almost every
        // step; the sum over 4 items exceeds 512 in about one wave-step in a hundred.
        // Run (qi, lane) has the id 64 qi + lane; its delta (posting index of quad f = 4 f + delta) is looked up in an LDS table by id,
        // its weight (fp16) in a second one written once.
        uint32_t* const tab_delta = reinterpret_cast<uint32_t*>(mark + SC_QUADS);
        _Float16* const tab_w = reinterpret_cast<_Float16*>(tab_delta + 256);
#pragma unroll
        for (int qi = 0; qi < 4; ++qi)
            tab_w[qi * 64 + lane] = (_Float16)a.rare_w[((int64_t)qb * SC_QB + sw * 4 + qi) * SC_MAXRT + lane];
        uint4 pp[SC_ITERS];
        uint32_t rid8[SC_ITERS / 4];      // run id of the lane's quad per step, 8 bits each
        uint32_t tail2;                   // tail code per step, 2 bits each
        int Ltot;
        uint32_t e_cons[4];
        auto stage_all = [&](int tile) {
            int nq4[4], bq4[4], Lq[4];
#pragma unroll
            for (int qi = 0; qi < 4; ++qi) {
                const uint32_t e = (tile < tile1 && term[qi] >= 0 && !(SC_DIAG & 1)) ? ecur[qi].x : 0u;
                e_cons[qi] = e;
                nq4[qi] = (multi_len(e) + 3) >> 2;
            }
#pragma unroll
            for (int qi = 0; qi < 4; ++qi) {
                const int es = sc_scan_add(nq4[qi]);
                Lq[qi] = __builtin_amdgcn_readlane(es, 63);
                bq4[qi] = es - nq4[qi];
            }
            int base = 0;
#pragma unroll
            for (int qi = 0; qi < 4; ++qi) {
                bq4[qi] += base;                                 // position of the lane's run in the wave's flat sequence
                base += Lq[qi];
                tab_delta[qi * 64 + lane] = cur[qi] - 4u * (uint32_t)bq4[qi];
            }
            Ltot = base;
#pragma unroll
            for (int it = 0; it < SC_ITERS; ++it) mark[it * 64 + lane] = 0;
#pragma unroll
            for (int qi = 0; qi < 4; ++qi) {
                const int ln = multi_len(e_cons[qi]);
                const int last = bq4[qi] + nq4[qi] - 1;
                const int id1 = qi * 64 + lane + 1;
                if (nq4[qi] == 1) {
                    if (bq4[qi] < SC_QUADS) mark[bq4[qi]] = (unsigned short)(id1 | ((ln & 3) << 9));
                } else if (nq4[qi] > 1) {
                    if (bq4[qi] < SC_QUADS) mark[bq4[qi]] = (unsigned short)id1;
                    if (last < SC_QUADS) mark[last] = (unsigned short)((ln & 3) << 9);
                }
            }
            int rid[SC_ITERS];
#pragma unroll
            for (int it = 0; it < SC_ITERS; ++it) rid[it] = (int)mark[it * 64 + lane];
            int carry = 0;
            tail2 = 0u;
#pragma unroll
            for (int it = 0; it < SC_ITERS / 4; ++it) rid8[it] = 0u;
            // running maximum over the 512 positions: eight INDEPENDENT in-wave scans (the compiler interleaves their DPP steps), the carry
            // from one 64-lane step into the next applied afterwards (max-scan(max(x, c)) = max(max-scan(x), c) for a wave-uniform c)
            int sm[SC_ITERS];
#pragma unroll
            for (int it = 0; it < SC_ITERS; ++it) {
                tail2 |= (uint32_t)(rid[it] >> 9) << (2 * it);
                sm[it] = sc_scan_max(rid[it] & 0x1ff);
            }
#pragma unroll
            for (int it = 0; it < SC_ITERS; ++it) {
                const int v = max(sm[it], carry);
                carry = max(carry, __builtin_amdgcn_readlane(sm[it], 63));
                rid[it] = (v - 1) & 255;
                rid8[it / 4] |= (uint32_t)rid[it] << (8 * (it % 4));
            }
            uint32_t dl[SC_ITERS];
#pragma unroll
            for (int it = 0; it < SC_ITERS; ++it) dl[it] = tab_delta[rid[it]];
#pragma unroll
            for (int it = 0; it < SC_ITERS; ++it) {
                const uint32_t pi = it * 64 + lane < Ltot ? dl[it] + 4u * (uint32_t)(it * 64 + lane) : 0u;
                const u32x4_u v4 = *reinterpret_cast<const u32x4_u*>(a.P + pi);
                pp[it] = make_uint4(v4.x, v4.y, v4.z, v4.w);
            }
#pragma unroll
            for (int qi = 0; qi < 4; ++qi) cur[qi] += run_len(e_cons[qi]);
        };
        // Adds what was staged a step ago.  cur is the start of the NEXT tile's run at this point: an item's run began run_len earlier.
        auto consume_all = [&](uint32_t* buf) {
            uint32_t* const wrow = buf + sw * 4 * SC_PITCH_W;
            _Float16 wv[SC_ITERS];
#pragma unroll
            for (int it = 0; it < SC_ITERS; ++it) wv[it] = tab_w[(rid8[it / 4] >> (8 * (it % 4))) & 255u];
#pragma unroll
            for (int qi = 0; qi < 4; ++qi) {
                const uint32_t e = e_cons[qi];
                if (e != 0u && (e >> 16) != 0xffffu) cert_add_posting(wrow + qi * SC_PITCH_W, e, (float)tab_w[qi * 64 + lane]);      // the lane's single posting
            }
#pragma unroll
            for (int it = 0; it < SC_ITERS; ++it)
                if (it * 64 + lane < Ltot) {
                    const uint32_t id = (rid8[it / 4] >> (8 * (it % 4))) & 255u;
                    add_quad(wrow + (id >> 6) * SC_PITCH_W, pp[it], (int)((tail2 >> (2 * it)) & 3u), (float)wv[it]);
                }
            if (Ltot > SC_QUADS)                                 // wave-uniform, rare at the shape this kernel is sized for
                cert_overflow_windows(mark, tab_delta, tab_w, a.P, wrow, e_cons[0], e_cons[1], e_cons[2], e_cons[3], Ltot);
        };
        auto advance_queue = [&](int tile) {
            if (tile & 1) {                                      // wave-uniform: the pair of tiles tile + 1, tile + 2
                load_group((tile >> 1) + 1, ecur);
            } else {
#pragma unroll
                for (int qi = 0; qi < 4; ++qi) ecur[qi].x = ecur[qi].y;
            }
        };
        // Queries with more than 64 rare terms: the terms beyond the first group (lane j of group g = rare term 64 g + j) are added in the
        // same step by a plain walk - the term, its E word and, for a run, its start in table S are looked up on the spot and a lane walks
        // its run posting by posting.  No prefetch, no flattening: a step of such a block takes a few memory round trips longer, which
        // only queries of 65-256 rare terms pay (they were handed to the exact kernels before, at a quarter of this speed).  Integer adds
        // commute: the keys do not depend on which path added a posting.
        int ngq[4], ng_max = 1;
#pragma unroll
        for (int qi = 0; qi < 4; ++qi) {
            ngq[qi] = __builtin_amdgcn_readfirstlane((a.n_rare[(int64_t)qb * SC_QB + sw * 4 + qi] + SC_MAXR - 1) / SC_MAXR);
            ng_max = ngq[qi] > ng_max ? ngq[qi] : ng_max;
        }
        stage_all(tile0);
        advance_queue(tile0);
        for (int tile = tile0; tile < tile1; ++tile) {
            consume_all(slots + (tile & 1) * SC_SLOT_WORDS);     // the loads were issued a step ago
            if (ng_max > 1)                                      // wave-uniform, off at the MSMARCO shape
                cert_extra_groups(a.rare_term, a.rare_w, a.E, a.S, a.P, a.e_stride, a.s_stride, (int64_t)qb * SC_QB + sw * 4, ngq[0], ngq[1], ngq[2], ngq[3],
                                  tile, slots + (tile & 1) * SC_SLOT_WORDS + sw * 4 * SC_PITCH_W);
            SC_STAMP(1);                                         // end of the adds
            stage_all(tile + 1);
            advance_queue(tile + 1);
            SC_STAMP(0);
            __syncthreads();                                     // step tile + 1
        }
        __syncthreads();                                         // the matrix waves' last step
    }
}

// ------------------------------------------------------------------------------------------- certificate kernel ---
// One workgroup per query over its UNSORTED running set (up to 2 (k + band) keys: the best k + band at the last select and everything
// appended since): the k-th best key by two 256-bin histogram passes over the 16-bit key values, the cut that follows from it, and the
// docs whose key reaches the cut, in any order - the re-score does not care, and sorting 8 192 keys per query for this took 0.8 ms.
// The set holds every doc whose key lies above max(tau, cut(tau2)): tau = the (k + band)-th best key at the last select (keys under
// it were cut away then), cut(tau2) <= the final cut (monotone).  So it is complete down to the final cut iff tau < cut.
struct CertSelectArgs {
    const uint64_t* run_keys;
    const int* run_count;
    const float* tau;
    int64_t nq;
    int k, k_eff, T;
    const uint8_t* elig;
    const uint8_t* overflow;
    const int32_t* n_rare;
    const int32_t* n_qt;
    const int32_t* n_drop;
    int64_t* ap_ids;          // [nq][2 k_eff] candidates' doc indices
    int32_t* m_count;
    uint8_t* uncert;
    int* n_uncert;
};
#define SC_SEL_R 32          // keys per thread: 2 (k + band) <= 2 SR_MAX_TOPK = 8 192
__global__ __launch_bounds__(256) void cert_select_kernel(CertSelectArgs a) {
    __shared__ int hist[256];
    __shared__ int scan[4];
    __shared__ int ctrl[4];
    const int64_t q = blockIdx.x;
    const int tid = threadIdx.x;
    const int cap = 2 * a.k_eff;
    const int n = a.run_count[q] < cap ? a.run_count[q] : cap;
    const uint64_t* run = a.run_keys + q * cap;
    bool good = a.elig[q] && !a.overflow[q] && n >= a.k;       // fewer than k docs with a non-zero key: the exact kernels decide
    if (!good) {
        if (tid == 0) { a.m_count[q] = 0; a.uncert[q] = 1; atomicAdd(a.n_uncert, 1); }
        return;
    }
    uint32_t v[SC_SEL_R];                                       // key values, integers in [1, 65 535]
#pragma unroll
    for (int j = 0; j < SC_SEL_R; ++j) {
        const int i = tid + 256 * j;
        v[j] = i < n ? (uint32_t)sr_key_score(run[i]) : 0u;
    }
    // rank `rem` from the top among the values whose high byte is `hi` (pass 1: all values, by high byte)
    auto select_byte = [&](int pass, uint32_t hi, int rem, int& bin, int& rem_out) {
        hist[tid] = 0;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < SC_SEL_R; ++j)
            if (tid + 256 * j < n && (pass == 0 || (v[j] >> 8) == hi)) atomicAdd(&hist[pass == 0 ? (v[j] >> 8) : (v[j] & 255u)], 1);
        __syncthreads();
        const int c = hist[tid];
        int sfx = c;                                             // sum over bins >= tid
        for (int off = 1; off < 64; off <<= 1) {
            const int o = __shfl_down(sfx, off);
            if ((tid & 63) + off < 64) sfx += o;
        }
        if ((tid & 63) == 0) scan[tid >> 6] = sfx;
        __syncthreads();
        for (int w = (tid >> 6) + 1; w < 4; ++w) sfx += scan[w];
        const int above = sfx - c;
        if (sfx >= rem && above < rem) { ctrl[0] = tid; ctrl[1] = rem - above; }
        __syncthreads();
        bin = ctrl[0];
        rem_out = ctrl[1];
        __syncthreads();
    };
    int b_hi, b_lo, rem;
    select_byte(0, 0u, a.k, b_hi, rem);
    select_byte(1, (uint32_t)b_hi, rem, b_lo, rem);
    const double cutd = cert_cut_from_kth((double)(b_hi * 256 + b_lo), a.T, a.n_rare[q], a.n_qt[q], a.n_drop[q]);
    good = cutd >= 1.0 && a.tau[q] < (float)cutd;
    if (tid == 0) ctrl[2] = 0;
    __syncthreads();
    if (good) {
        const uint32_t cut = (uint32_t)cutd;
#pragma unroll
        for (int j = 0; j < SC_SEL_R; ++j) {
            const int i = tid + 256 * j;
            if (i < n && v[j] >= cut) a.ap_ids[q * cap + atomicAdd(&ctrl[2], 1)] = (int64_t)(uint32_t)(~(uint32_t)run[i]);
        }
    }
    __syncthreads();
    if (tid == 0) {
        a.m_count[q] = good ? ctrl[2] : 0;
        a.uncert[q] = good ? 0 : 1;
        if (!good) atomicAdd(a.n_uncert, 1);
        else atomicAdd(a.n_uncert + 1, (ctrl[2] + 15) >> 4);       // statistics
    }
}

// ---------------------------------------------------------------------------------------------- exact re-score ---
// One workgroup per certified query, one wave per candidate: the doc's forward row (terms ascending) is intersected with the
// query's terms (ascending) and the products are added in that order, unfused - the reference's chain (indexer.py:324-340).
struct CertRescoreArgs {
    const int64_t* q_indptr;
    const int32_t* q_cols;
    const float* q_vals;
    const int64_t* fwd_indptr;
    const uint64_t* fwd_tv;
    const int64_t* ap_ids;
    const int32_t* m_count;
    int ap_stride;
    float threshold;
    uint32_t id_base, id_stride;
    uint64_t* cand_keys;
    int* cand_count;
    int64_t cand_cap;
};
__global__ __launch_bounds__(256) void cert_rescore_kernel(CertRescoreArgs a) {
#pragma clang fp contract(off)
    // the query's terms in an open-addressed table (1 024 slots for at most 256 terms, linear probing): a doc term finds its partner in
    // ~1.1 LDS reads; a binary search over the sorted terms took 8 dependent ones per doc term and was most of this kernel's time
    __shared__ int32_t hk[1024];
    __shared__ float hv[1024];
    __shared__ int n_kept;
    // queries of at most 64 terms (a row then has at most 64 matching postings): a row's products are compacted into LDS in term order
    // and 16 rows are summed side by side, lane r adding up row r - the ordered sum costs a lane-parallel loop per 16 rows instead of
    // a scalar loop iteration per match
    __shared__ float pl[4][16][65];
    __shared__ int pcnt[4][16];
    __shared__ uint32_t pdoc[4][16];
    const int64_t q = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = a.m_count[q];
    if (m == 0) return;                       // cand_count[q] stays 0 (topk_reset)
    const int64_t tb = a.q_indptr[q];
    const int nqt = (int)(a.q_indptr[q + 1] - tb);
    for (int i = tid; i < 1024; i += 256) hk[i] = -1;
    if (tid == 0) n_kept = 0;
    __syncthreads();
    for (int i = tid; i < nqt; i += 256) {                       // nqt <= SC_MAXQT, terms distinct (cert_plan_kernel's eligibility)
        const int32_t t = a.q_cols[tb + i];
        if (t < 0) continue;                                     // matches no doc term (and -1 marks an empty slot)
        uint32_t h = ((uint32_t)t * 2654435761u) >> 22;
        while (atomicCAS(&hk[h], -1, t) != -1) h = (h + 1u) & 1023u;
        hv[h] = a.q_vals[tb + i];
    }
    __syncthreads();
    uint64_t* dst = a.cand_keys + q * a.cand_cap;
    const bool lists = nqt <= 64 && !(SC_DIAG & 2048);       // bit 2048: the scalar loop for every query (A/B)
    // One wave per candidate.  What the kernel's time is made of (timing-only builds, 7.7 M candidates per pass): the row gathers alone
    // 1.5 ms (5.4 TB/s); the intersection alone, every load served from cache, 3.4 ms - and of that nearly all is the ORDERED sum: the
    // products must be added one after the other in term order (the reference's fp32 chain), a scalar loop iteration per match (15-20
    // per top candidate) with 63 lanes idle.  Tried and dropped: one LANE per candidate (64 chains side by side, but 64 rows per load
    // instruction: 7.7 ms).  Kept: doc indices and row bounds of 64 candidates gathered at once (lane j = candidate j of the batch); the
    // first 256 postings of the next TWO candidates' rows in flight, 16 bytes = two postings per lane, no load and no register touched
    // at issue time on the common path (the compiler then waits with a counted vmcnt); the query's terms in a hash table instead of a
    // binary search; a branch-free body of the ordered sum.
    auto rl64 = [&](int64_t x, int j) -> int64_t {
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)x, j), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(x >> 32), j);
        return (int64_t)(((uint64_t)hi << 32) | lo);
    };
    for (int c0 = wave; c0 < m; c0 += 256) {
        const int cj = c0 + 4 * lane;                 // the wave's candidates: wave, wave + 4, ...
        int64_t b_l = 0, e_l = 0;
        uint32_t doc_l = 0u;
        if (cj < m) {
            const int64_t d = a.ap_ids[q * a.ap_stride + cj];
            doc_l = (uint32_t)d;
            b_l = a.fwd_indptr[d];
            e_l = a.fwd_indptr[d + 1];
        }
        const int nb = (m - c0 + 3) / 4 < 64 ? (m - c0 + 3) / 4 : 64;
        // a chunk = 128 postings of a row, two per lane; idle lanes read the table's start (the load is unconditional), and what lies beyond
        // the row is masked where the chunk is USED - touching the registers here would wait for the load on the spot
        auto load_chunk = [&](int64_t p0, int64_t e) -> u32x4_u {
            const int64_t p = p0 + 2 * lane;
            if (SC_DIAG & 1024) return *reinterpret_cast<const u32x4_u*>(a.fwd_tv + 2 * lane);      // timing only: every row is the table's start (cache hits)
            return *reinterpret_cast<const u32x4_u*>(a.fwd_tv + (p < e ? p : 0));
        };
        u32x4_u rr[2][2];                             // [row in flight][chunk]: (value bits, term) x 2 per lane
        auto fetch = [&](int j, u32x4_u (&r)[2]) {
            const int jj = j < nb ? j : nb - 1;       // past the batch: re-reads its last row
            const int64_t b = rl64(b_l, jj), e = rl64(e_l, jj);
            r[0] = load_chunk(b, e);
            r[1] = load_chunk(b + 128, e);
        };
        auto add_chunk = [&](const u32x4_u& r, int64_t p0, int64_t e, float& s) {
            const int64_t p = p0 + 2 * lane;
            const uint32_t t0 = r.y, t1 = r.w;
            bool d0 = !(p < e), d1 = !(p + 1 < e), m0 = false, m1 = false;
            uint32_t h0 = (t0 * 2654435761u) >> 22, h1 = (t1 * 2654435761u) >> 22;
            while (!(d0 && d1)) {                      // both postings of a lane probe in one loop: their LDS reads go out together
                const int32_t k0 = hk[h0], k1 = hk[h1];
                if (!d0) {
                    if (k0 == (int32_t)t0) { m0 = true; d0 = true; }
                    else if (k0 == -1) d0 = true;
                    else h0 = (h0 + 1u) & 1023u;
                }
                if (!d1) {
                    if (k1 == (int32_t)t1) { m1 = true; d1 = true; }
                    else if (k1 == -1) d1 = true;
                    else h1 = (h1 + 1u) & 1023u;
                }
            }
            const float w0 = hv[h0], w1 = hv[h1];
            // a posting without a partner contributes +0.0f: s + 0.0f == s bit for bit (s is a sum of non-negative products, never -0)
            const float prod0 = m0 ? w0 * __uint_as_float(r.x) : 0.f, prod1 = m1 ? w1 * __uint_as_float(r.z) : 0.f;
            uint64_t mm = __ballot(m0 || m1);
            while (mm) {                               // ascending terms: lane by lane, a lane's first posting before its second
                const int i = __builtin_ctzll(mm);
                mm &= mm - 1;
                s = s + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(prod0), i));
                s = s + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(prod1), i));
            }
        };
        // the same probe, the products written to their place in the row's list instead of being added
        auto list_chunk = [&](const u32x4_u& r, int64_t p0, int64_t e, int row, int& base) {
            const int64_t p = p0 + 2 * lane;
            const uint32_t t0 = r.y, t1 = r.w;
            bool d0 = !(p < e), d1 = !(p + 1 < e), m0 = false, m1 = false;
            uint32_t h0 = (t0 * 2654435761u) >> 22, h1 = (t1 * 2654435761u) >> 22;
            while (!(d0 && d1)) {
                const int32_t k0 = hk[h0], k1 = hk[h1];
                if (!d0) {
                    if (k0 == (int32_t)t0) { m0 = true; d0 = true; }
                    else if (k0 == -1) d0 = true;
                    else h0 = (h0 + 1u) & 1023u;
                }
                if (!d1) {
                    if (k1 == (int32_t)t1) { m1 = true; d1 = true; }
                    else if (k1 == -1) d1 = true;
                    else h1 = (h1 + 1u) & 1023u;
                }
            }
            const uint64_t b0 = __ballot(m0), b1 = __ballot(m1);
            if ((b0 | b1) == 0ull) return;
            const float w0 = hv[h0], w1 = hv[h1];
            const int below = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(b0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b0, 0u)) +
                              (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(b1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b1, 0u));
            if (m0) pl[wave][row][base + below] = w0 * __uint_as_float(r.x);
            if (m1) pl[wave][row][base + below + (m0 ? 1 : 0)] = w1 * __uint_as_float(r.z);
            base += __popcll(b0) + __popcll(b1);
        };
        auto list_row = [&](int j, const u32x4_u (&cr)[2]) {
            const int64_t b = rl64(b_l, j), e = rl64(e_l, j);
            const int row = j & 15;
            int base = 0;
            list_chunk(cr[0], b, e, row, base);
            if (e - b > 128) list_chunk(cr[1], b + 128, e, row, base);
            for (int64_t p0 = b + 256; p0 < e; p0 += 128) {
                const u32x4_u r = load_chunk(p0, e);
                list_chunk(r, p0, e, row, base);
            }
            if (lane == 0) { pcnt[wave][row] = base; pdoc[wave][row] = (uint32_t)__builtin_amdgcn_readlane((int)doc_l, j); }
        };
        auto sum_rows = [&](int rows) {                 // lane r: row r's products, one after the other
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            if (lane < rows) {
                const int n = pcnt[wave][lane];
                float s = 0.f;
                for (int i = 0; i < n; ++i) s = s + pl[wave][lane][i];
                if (s > a.threshold) {
                    const int pos = atomicAdd(&n_kept, 1);
                    dst[pos] = sr_make_key(s, a.id_base + pdoc[wave][lane] * a.id_stride);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        };
        auto intersect = [&](int j, const u32x4_u (&cr)[2]) {
            const int64_t b = rl64(b_l, j), e = rl64(e_l, j);
            const uint32_t doc = (uint32_t)__builtin_amdgcn_readlane((int)doc_l, j);
            float s = 0.f;
            add_chunk(cr[0], b, e, s);
            if (e - b > 128) add_chunk(cr[1], b + 128, e, s);    // wave-uniform, no load inside
            for (int64_t p0 = b + 256; p0 < e; p0 += 128) {      // rows beyond 256 postings: their own, rare loop
                const u32x4_u r = load_chunk(p0, e);
                add_chunk(r, p0, e, s);
            }
            if (lane == 0 && s > a.threshold) {
                const int pos = atomicAdd(&n_kept, 1);
                dst[pos] = sr_make_key(s, a.id_base + doc * a.id_stride);
            }
        };
        fetch(0, rr[0]);
        fetch(1, rr[1]);
        for (int j = 0; j < nb; j += 2) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const u32x4_u cr[2] = {rr[h][0], rr[h][1]};
                fetch(j + h + 2, rr[h]);               // unconditional (clamped): a constant number of loads per step
                if (j + h < nb && !(SC_DIAG & 512)) {     // wave-uniform (bit 512, timing only: rows loaded, nothing intersected)
                    if (lists) {
                        list_row(j + h, cr);
                        if (((j + h) & 15) == 15 || j + h == nb - 1) sum_rows(((j + h) & 15) + 1);
                    } else intersect(j + h, cr);
                }
                if ((SC_DIAG & 512) && cr[0].x == 0x12345u && cr[1].y == 0x54321u) n_kept = 1;       // keeps the loads alive
            }
        }
    }
    __syncthreads();
    if (tid == 0) a.cand_count[q] = n_kept;
}

// ------------------------------------------------------------------------------------------------------ driver ---
template <typename T>
static bool cert_realloc(T*& p, size_t n) {
    if (p) (void)hipFree(p);
    p = nullptr;
    if (hipMalloc((void**)&p, sizeof(T) * n) != hipSuccess) {
        (void)hipGetLastError();
        p = nullptr;
        return false;
    }
    return true;
}

// the per-call buffers of a batch of nq_pad queries; false = out of device memory: everything per-call is released (the index-side
// structures stay) and the caller serves the batch with the exact kernels
static bool cert_ensure_call_buffers(SparseCert* c, int64_t nq_pad, int k_eff) {      // k_eff = 0: the plan's buffers only (the band is chosen after the plan)
    bool ok = true;
    if (nq_pad > c->nq_cap) {
        c->nq_cap = 0;
        ok = ok && cert_realloc(c->bfrag, (size_t)nq_pad * (size_t)c->T);
        ok = ok && cert_realloc(c->rare_term, (size_t)nq_pad * SC_MAXRT);
        ok = ok && cert_realloc(c->rare_w, (size_t)nq_pad * SC_MAXRT);
        ok = ok && cert_realloc(c->cq, (size_t)nq_pad);
        ok = ok && cert_realloc(c->sq, (size_t)nq_pad);
        ok = ok && cert_realloc(c->n_rare, (size_t)nq_pad);
        ok = ok && cert_realloc(c->n_qt, (size_t)nq_pad);
        ok = ok && cert_realloc(c->n_drop, (size_t)nq_pad);
        ok = ok && cert_realloc(c->tau2, (size_t)nq_pad);
        ok = ok && cert_realloc(c->elig, (size_t)nq_pad);
        ok = ok && cert_realloc(c->overflow, (size_t)nq_pad);
        ok = ok && cert_realloc(c->m_count, (size_t)nq_pad);
        if (ok) c->nq_cap = nq_pad;
    }
    if (ok && k_eff > 0 && nq_pad * k_eff > c->ap_cap) {
        c->ap_cap = 0;
        ok = cert_realloc(c->ap_ids, (size_t)nq_pad * (size_t)(2 * k_eff));
        if (ok) c->ap_cap = nq_pad * k_eff;
    }
    if (ok && !c->d_n_uncert && hipMalloc((void**)&c->d_n_uncert, 4 * sizeof(int)) != hipSuccess) {     // [0] uncertified queries, [1] candidates re-scored (in units of 16), [2] largest rare-term count
        (void)hipGetLastError();
        c->d_n_uncert = nullptr;
        ok = false;
    }
    if (ok && k_eff > 0 && c->ws.ensure(nq_pad, k_eff, SC_CAND_CAP) != SR_OK) ok = false;      // releases itself on failure
    if (!ok) {
        void** ptrs[] = {(void**)&c->bfrag, (void**)&c->rare_term, (void**)&c->rare_w, (void**)&c->cq, (void**)&c->sq, (void**)&c->n_rare, (void**)&c->n_qt,
                         (void**)&c->n_drop, (void**)&c->tau2, (void**)&c->elig, (void**)&c->overflow, (void**)&c->m_count, (void**)&c->ap_ids};
        for (void** pp : ptrs) {
            if (*pp) (void)hipFree(*pp);
            *pp = nullptr;
        }
        c->nq_cap = 0;
        c->ap_cap = 0;
        c->ws.release();
    }
    return ok;
}

// a sub-batch of `ns` handed-back queries went through the scorer a second time: count them once (queries given; handed to the exact kernels
// = what the second pass handed back)
void sparse_cert_count_retry(SparseCert* c, int64_t ns) {
    c->n_queries -= ns;
    c->n_uncert -= ns;
    --c->n_calls;
}

uint8_t* sparse_cert_uncert_buffer(SparseCert* c, int64_t nq) {
    if (nq > c->uncert_cap) {
        c->uncert_cap = 0;
        if (!cert_realloc(c->d_uncert, (size_t)nq)) return nullptr;
        c->uncert_cap = nq;
    }
    return c->d_uncert;
}

template <int KS>
static int cert_launch_score(const CertArgs& a, unsigned grid, hipStream_t s) {
    static DeviceOnce lds_set;
    static_assert(1024 * KS <= SC_BREGION, "B fragments");
    const int lds = (int)(sizeof(uint32_t) * 2 * SC_SLOT_WORDS + SC_BREGION + SC_TAIL_LDS + (a.stamps ? 256 : 0));
    if (bool* slot = lds_set.pending()) {
        SR_CHECK_HIP(hipFuncSetAttribute((const void*)cert_score_kernel<KS>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        *slot = true;
    }
    hipLaunchKernelGGL(cert_score_kernel<KS>, dim3(grid), dim3(1024), lds, s, a);
    SR_CHECK_LAUNCH();
    return SR_OK;
}

int sparse_cert_search(sr_sparse_index* idx, const int64_t* d_q_indptr, const int32_t* d_q_cols, const float* d_q_vals, int64_t nq,
                       int k, float threshold, int64_t id_base, int64_t id_stride, float* d_out_scores, int64_t* d_out_ids,
                       int32_t* d_out_counts, uint8_t* d_uncert, int64_t* n_uncert, bool* no_memory, int band_keys, int* band_used, hipStream_t s) {
    SparseCert* c = idx->cert;
    const int64_t nq_pad = ceil_div64(nq, SC_QB) * SC_QB;
    const int n_qblocks = (int)(nq_pad / SC_QB);
    *no_memory = false;
    if (!cert_ensure_call_buffers(c, nq_pad, 0)) {
        *no_memory = true;
        return SR_OK;
    }
    const int64_t dump_stride = (int64_t)c->n_tiles * SC_DT;
    if (c->want_dump && c->dump_nq < nq_pad) {
        c->dump_nq = 0;
        if (!cert_realloc(c->dump, (size_t)nq_pad * (size_t)dump_stride)) {
            sr_set_error("sparse_cert_search: no device memory for the debug key dump");
            return SR_ERR_NOMEM;
        }
        c->dump_nq = nq_pad;
    }

    // plan
    SR_CHECK_HIP(hipMemsetAsync(c->bfrag, 0, sizeof(_Float16) * (size_t)nq_pad * (size_t)c->T, s));
    SR_CHECK_HIP(hipMemsetAsync(c->d_n_uncert, 0, 4 * sizeof(int), s));
    CertPlanArgs pa;
    pa.q_indptr = d_q_indptr; pa.q_cols = d_q_cols; pa.q_vals = d_q_vals; pa.nq = nq; pa.n_terms = idx->n_terms;
    pa.indptr = idx->indptr; pa.dslot = c->dslot; pa.vmax = c->vmax; pa.KS = c->KS; pa.vscale = c->vscale;
    pa.bfrag = c->bfrag; pa.rare_term = c->rare_term; pa.rare_w = c->rare_w; pa.cq = c->cq; pa.sq = c->sq; pa.n_rare = c->n_rare; pa.n_qt = c->n_qt; pa.n_drop = c->n_drop;
    pa.elig = c->elig; pa.overflow = c->overflow; pa.max_rare = c->d_n_uncert + 2;
    hipLaunchKernelGGL(cert_plan_kernel, dim3((unsigned)ceil_div64(nq_pad, 4)), dim3(256), 0, s, pa);
    SR_CHECK_LAUNCH();
    // The band: how many keys beyond k the running set keeps.  The certificate needs every key that the 16-bit arithmetic cannot tell from
    // the k-th inside it, and each rare term of a query widens that stretch by one key unit (the bound of DESIGN.md 4.6): 1 024 keys hold
    // it up to ~100 rare terms (L0_q = 128 at the MSMARCO shape: all certified), beyond that the band grows - at 145 rare terms (L0_q = 256)
    // 2 048 keys certify 93 % of the queries and 3 072 all of them, at 12 100 instead of 4 900 queries/s through the exact kernels.  The
    // short queries of a default batch keep the band that is fastest for them (140 000 q/s at 1 024, 130 000 at 3 072).
    int band = band_keys;
    int h_plan[2] = {0, 0};                                  // largest rare-term count, queries on the fast path
    SR_CHECK_HIP(hipMemcpyAsync(h_plan, c->d_n_uncert + 2, 2 * sizeof(int), hipMemcpyDeviceToHost, s));
    SR_CHECK_HIP(hipStreamSynchronize(s));
    if (h_plan[1] == 0) {                                    // nothing for the scorer in this batch (negative / unordered / too long queries): no pass over the collection
        SR_CHECK_HIP(hipMemsetAsync(d_uncert, 1, (size_t)nq, s));
        SR_CHECK_HIP(hipStreamSynchronize(s));
        if (band_used) *band_used = SR_MAX_TOPK;             // no retry either
        *n_uncert = nq;
        ++c->n_calls;
        c->n_queries += nq;
        c->n_uncert += nq;
        return SR_OK;
    }
    if (band <= 0) {
        band = h_plan[0] > 160 ? 3072 : (h_plan[0] > 96 ? 2048 : SC_BAND);
        if (const char* e = sr_dev_getenv("SR_SPARSE_CERT_BANDKEYS")) band = atoi(e);        // A/B switch: keys kept beyond k
    }
    if (band < 64) band = 64;
    if (k + band > SR_MAX_TOPK) band = SR_MAX_TOPK - k;
    const int k_eff = k + band;
    if (band_used) *band_used = band;
    if (const char* e = getenv("SR_LOG")) if (atoi(e) >= 2)
        fprintf(stderr, "[sr_hip] certified sparse search: %lld queries, %d on the fast path, largest rare-term count %d, band %d keys%s\n", (long long)nq, h_plan[1], h_plan[0],
                band, band_keys > 0 ? " (second pass)" : "");
    if (!cert_ensure_call_buffers(c, nq_pad, k_eff)) {
        *no_memory = true;
        return SR_OK;
    }

    // stage 1: doc tiles in launches that grow geometrically (the threshold tightens early), at most 512 tiles each
    SR_TRY(topk_reset(c->ws, nq_pad, s));
    CertArgs a;
    a.d16 = reinterpret_cast<const f16x8*>(c->d16);
    a.P = c->P; a.S = c->S; a.s_stride = c->n_tiles + 1; a.E = c->E; a.e_stride = c->e_stride;
    a.bfrag = reinterpret_cast<const f16x8*>(c->bfrag);
    a.rare_term = c->rare_term; a.rare_w = c->rare_w; a.cq = c->cq; a.tau = c->ws.tau; a.nq = nq;
    a.tau2 = c->tau2; a.n_rare = c->n_rare; a.n_qt = c->n_qt; a.n_drop = c->n_drop; a.T = c->T;
    SR_CHECK_HIP(hipMemsetAsync(c->tau2, 0, sizeof(float) * (size_t)nq_pad, s));
    a.cand_keys = c->ws.cand_keys; a.cand_count = c->ws.cand_count; a.cand_cap = c->ws.cand_cap; a.overflow = c->overflow;
    a.n_qblocks = n_qblocks;
    a.dump = c->want_dump ? c->dump : nullptr;
    if (SC_STAMPS && sr_dev_getenv("SR_CERT_STAMPS") && !c->d_stamps) {
        SR_CHECK_HIP(hipMalloc((void**)&c->d_stamps, 80 * 8));
        SR_CHECK_HIP(hipMemsetAsync(c->d_stamps, 0, 80 * 8, s));
    }
    a.stamps = c->d_stamps;
    a.stamps_from = 0;
    if (SC_STAMPS) { if (const char* e = sr_dev_getenv("SR_CERT_STAMPS")) a.stamps_from = atoi(e); }
    a.dump_stride = dump_stride;
    int64_t step = k_eff / SC_DT + 1;      // the first launch covers just over k + band docs (no threshold exists before that many keys are held), then doubling
    if (const char* e = sr_dev_getenv("SR_SPARSE_CERT_STEP0")) step = std::max(1, atoi(e));
    bool band_filter = true;               // dev switch SR_SPARSE_CERT_BAND=0: filter with the (k + band)-th best key only
    if (const char* e = sr_dev_getenv("SR_SPARSE_CERT_BAND")) band_filter = atoi(e) != 0;
    int sel_over = 2 * k_eff;
    if (const char* e = sr_dev_getenv("SR_SPARSE_CERT_SELOVER")) sel_over = (int)(atof(e) * k_eff);
    // while the launches still double, every compaction selects: a launch then appends k + band candidates per query whatever its size, and a
    // threshold one launch old lets twice as many through (-0.9 ms per pass at the MSMARCO shape; dev switch SR_SPARSE_CERT_SELEARLY=0)
    bool sel_early = true;
    if (const char* e = sr_dev_getenv("SR_SPARSE_CERT_SELEARLY")) sel_early = atoi(e) != 0;
    int64_t tpw_max = 32;                  // tiles a workgroup walks: amortises its prologue; the chunk's operand (256 KB per tile at T = 128) should stay in its XCD's L2
    if (const char* e = sr_dev_getenv("SR_SPARSE_CERT_TPW")) tpw_max = std::max(1, atoi(e));
    for (int64_t t0 = 0; t0 < c->n_tiles;) {
        int64_t nt = step < 512 ? step : 512;
        if (t0 + nt > c->n_tiles) nt = c->n_tiles - t0;
        a.tile_begin = (int)t0;
        a.n_tiles_launch = (int)nt;
        int64_t tpw = nt * n_qblocks / 1024;           // enough workgroups to fill the chip, then longer walks per workgroup
        tpw = tpw < 1 ? 1 : (tpw > tpw_max ? tpw_max : tpw);
        a.tiles_per_wg = (int)tpw;
        const int64_t n_chunks = ceil_div64(nt, tpw);
        const unsigned grid = (unsigned)(ceil_div64(n_chunks, 8) * 8 * n_qblocks);
        idx->prof.begin(s);
        int rc;
        switch (c->KS) {
            case 1: rc = cert_launch_score<1>(a, grid, s); break;
            case 2: rc = cert_launch_score<2>(a, grid, s); break;
            case 4: rc = cert_launch_score<4>(a, grid, s); break;
            case 5: rc = cert_launch_score<5>(a, grid, s); break;
            case 6: rc = cert_launch_score<6>(a, grid, s); break;
            case 7: rc = cert_launch_score<7>(a, grid, s); break;
            case 8: rc = cert_launch_score<8>(a, grid, s); break;
            default: sr_set_error("sparse_cert_search: %d k-steps are not instantiated", c->KS); rc = SR_ERR_INVALID; break;
        }
        idx->prof.end(s, 0, 0);
        SR_TRY(rc);
        SR_TRY(topk_compact2(c->ws, nq_pad, k_eff, band_filter ? k : 0, c->tau2, (sel_early && nt < 512) ? k_eff : sel_over, s));
        t0 += nt;
        step *= 2;
    }

    // stage 2: certificate, exact re-score of the candidates, exact top-k of those
    CertSelectArgs sa;
    sa.run_keys = c->ws.run_keys; sa.run_count = c->ws.run_count; sa.tau = c->ws.tau; sa.ap_ids = c->ap_ids;
    sa.nq = nq; sa.k = k; sa.k_eff = k_eff; sa.T = c->T;
    sa.elig = c->elig; sa.overflow = c->overflow; sa.n_rare = c->n_rare; sa.n_qt = c->n_qt; sa.n_drop = c->n_drop; sa.m_count = c->m_count;
    sa.uncert = d_uncert; sa.n_uncert = c->d_n_uncert;
    hipLaunchKernelGGL(cert_select_kernel, dim3((unsigned)nq), dim3(256), 0, s, sa);
    SR_CHECK_LAUNCH();
    SR_TRY(topk_reset(c->ws, nq_pad, s));
    CertRescoreArgs ra;
    ra.q_indptr = d_q_indptr; ra.q_cols = d_q_cols; ra.q_vals = d_q_vals;
    ra.fwd_indptr = c->fwd_indptr; ra.fwd_tv = c->fwd_tv;
    ra.ap_ids = c->ap_ids; ra.m_count = c->m_count; ra.ap_stride = 2 * k_eff; ra.threshold = threshold;
    ra.id_base = (uint32_t)id_base; ra.id_stride = (uint32_t)id_stride;
    ra.cand_keys = c->ws.cand_keys; ra.cand_count = c->ws.cand_count; ra.cand_cap = c->ws.cand_cap;
    hipLaunchKernelGGL(cert_rescore_kernel, dim3((unsigned)nq), dim3(256), 0, s, ra);
    SR_CHECK_LAUNCH();
    SR_TRY(topk_compact(c->ws, nq, k, s));
    SR_TRY(topk_finalize(c->ws, nq, k, 0.f, d_out_scores, d_out_ids, d_out_counts, s));
    int h_un2[2] = {0, 0};
    SR_CHECK_HIP(hipMemcpyAsync(h_un2, c->d_n_uncert, 2 * sizeof(int), hipMemcpyDeviceToHost, s));
    SR_CHECK_HIP(hipStreamSynchronize(s));
    const int h_un = h_un2[0];
    c->n_rescored += 16ll * h_un2[1];
    *n_uncert = h_un;
    ++c->n_calls;
    c->n_queries += nq;
    c->n_uncert += h_un;
    return SR_OK;
}

// Which path served the searches so far.  out[0] = 1 if the certified scorer exists for this index, [1] dense terms (MFMA K),
// [2] searches it ran, [3] queries it was given, [4] queries it handed to the exact kernels (uncertified), [5] doc tiles, [6] candidates
// re-scored, [7] query batches served by the exact kernels because the scorer's per-call buffers did not fit in device memory
extern "C" int sr_sparse_index_cert_stats(sr_sparse_index* idx, int64_t* out8) {
    SR_REQUIRE(idx && out8, "sr_sparse_index_cert_stats: null argument");
    std::lock_guard<std::mutex> lock(idx->mu);
    for (int i = 0; i < 8; ++i) out8[i] = 0;
    if (SparseCert* c = idx->cert) {
        out8[0] = 1; out8[1] = c->T; out8[2] = c->n_calls; out8[3] = c->n_queries; out8[4] = c->n_uncert; out8[5] = c->n_tiles; out8[6] = c->n_rescored;
    }
    out8[7] = idx->n_cert_no_memory;
    return SR_OK;
}

// Debug / test hook: after enable = 1 every search keeps the stage-1 keys of all (query, doc) pairs; enable = 2 copies the keys of
// the last search to h_keys [nq_pad][n_tiles * 1024] (uint16) and returns the per-query constants the bound needs:
// h_consts [nq_pad][5] = {cq (0: the query is outside the fast path), s_q, rare terms in stage 1, query terms, rare terms left out}.  *vscale / *T: the index-side constants.
extern "C" int sr_sparse_index_cert_debug(sr_sparse_index* idx, int enable, uint16_t* h_keys, int64_t keys_capacity, float* h_consts,
                                          int64_t nq_pad, float* vscale, int32_t* T) {
    SR_REQUIRE(idx, "sr_sparse_index_cert_debug: null index");
    std::lock_guard<std::mutex> lock(idx->mu);
    SparseCert* c = idx->cert;
    SR_REQUIRE(c, "sr_sparse_index_cert_debug: this index has no certified scorer");
    if (enable != 2) {
        c->want_dump = enable != 0;
        return SR_OK;
    }
    SR_REQUIRE(c->dump && h_keys && h_consts && nq_pad <= c->dump_nq, "sr_sparse_index_cert_debug: nothing recorded");
    const int64_t stride = (int64_t)c->n_tiles * SC_DT;
    SR_REQUIRE(keys_capacity >= nq_pad * stride, "sr_sparse_index_cert_debug: key buffer too small");
    SR_CHECK_HIP(hipDeviceSynchronize());
    SR_CHECK_HIP(hipMemcpy(h_keys, c->dump, sizeof(uint16_t) * (size_t)(nq_pad * stride), hipMemcpyDeviceToHost));
    std::vector<float> cqv((size_t)nq_pad), sqv((size_t)nq_pad);
    std::vector<int32_t> nr((size_t)nq_pad), nt((size_t)nq_pad), nd((size_t)nq_pad);
    SR_CHECK_HIP(hipMemcpy(cqv.data(), c->cq, sizeof(float) * (size_t)nq_pad, hipMemcpyDeviceToHost));
    SR_CHECK_HIP(hipMemcpy(sqv.data(), c->sq, sizeof(float) * (size_t)nq_pad, hipMemcpyDeviceToHost));
    SR_CHECK_HIP(hipMemcpy(nr.data(), c->n_rare, sizeof(int32_t) * (size_t)nq_pad, hipMemcpyDeviceToHost));
    SR_CHECK_HIP(hipMemcpy(nt.data(), c->n_qt, sizeof(int32_t) * (size_t)nq_pad, hipMemcpyDeviceToHost));
    SR_CHECK_HIP(hipMemcpy(nd.data(), c->n_drop, sizeof(int32_t) * (size_t)nq_pad, hipMemcpyDeviceToHost));
    std::vector<float> t2((size_t)nq_pad);
    SR_CHECK_HIP(hipMemcpy(t2.data(), c->tau2, sizeof(float) * (size_t)nq_pad, hipMemcpyDeviceToHost));
    for (int64_t q = 0; q < nq_pad; ++q) {
        h_consts[6 * q] = cqv[(size_t)q];
        h_consts[6 * q + 1] = sqv[(size_t)q];
        h_consts[6 * q + 2] = (float)nr[(size_t)q];
        h_consts[6 * q + 3] = (float)nt[(size_t)q];
        h_consts[6 * q + 4] = (float)nd[(size_t)q];
        h_consts[6 * q + 5] = t2[(size_t)q];
    }
    if (vscale) *vscale = c->vscale;
    if (T) *T = c->T;
    return SR_OK;
}
