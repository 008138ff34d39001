/* ORACLE (test infrastructure, not product code): plain-C restatement of the
 * reference's CPU scoring stage.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library.
 *
 * Follows:
 *   sparse  /root/reference/scaling_retriever/indexer.py:324-344 (numba_score_float)
 *           /root/reference/scaling_retriever/indexer.py:315-322 (select_topk)
 *           threading shape of indexer.py:458-459 (4 query-level workers x
 *           parallel posting loop)
 *   dense   faiss-cpu==1.8.0 IndexFlatIP.search [3P, not under /root/reference;
 *           call sites indexer.py:196,203,211]: exact fp32 inner products,
 *           k best per query, descending.
 * Parity status: sparse PINNED by tests/golden/sparse_score.npz (outputs of the
 * reference's own function bodies); dense "parity unpinned" (faiss absent) -
 * anchored on brute-force numpy.
 *
 * Build: see oracle/Makefile  (gcc -O3 -fopenmp -shared -fPIC)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ---- sparse: numba_score_float, term-serial, non-fused multiply-add ------ */
/* scores must hold N floats; returns number of docs with score > threshold and
 * writes their ascending indices to out_idx (int64) and NEGATED scores to
 * out_neg, exactly what the reference returns. inner_threads parallelises the
 * posting loop like numba.prange (doc ids are unique inside one posting list). */
int64_t oracle_sparse_score(const int64_t* indptr, const int32_t* doc_ids, const float* vals,
                            const int32_t* q_cols, const float* q_vals, int n_terms,
                            float threshold, int64_t N, float* scores,
                            int64_t* out_idx, float* out_neg, int inner_threads) {
    memset(scores, 0, (size_t)N * sizeof(float));
    for (int t = 0; t < n_terms; ++t) {
        const int64_t b = indptr[q_cols[t]], e = indptr[q_cols[t] + 1];
        const float q = q_vals[t];
#pragma omp parallel for num_threads(inner_threads) if (inner_threads > 1 && e - b > 4096)
        for (int64_t j = b; j < e; ++j) {
            scores[doc_ids[j]] += q * vals[j];   /* built with -ffp-contract=off: mul and add stay unfused, as in numba (no fastmath) */
        }
    }
    int64_t m = 0;
    for (int64_t i = 0; i < N; ++i)
        if (scores[i] > threshold) { out_idx[m] = i; out_neg[m] = -scores[i]; ++m; }
    return m;
}

typedef struct { float s; int64_t i; } cand_t;
/* order: higher score first, ties by lower index */
static int cand_better(const cand_t* a, const cand_t* b) {
    return (a->s > b->s) || (a->s == b->s && a->i < b->i);
}
static int cand_cmp_desc(const void* pa, const void* pb) {
    const cand_t* a = (const cand_t*)pa; const cand_t* b = (const cand_t*)pb;
    if (cand_better(a, b)) return -1;
    if (cand_better(b, a)) return 1;
    return 0;
}
/* min-heap on "better" (root = worst kept) */
static void heap_sift(cand_t* h, int n, int i) {
    for (;;) {
        int l = 2 * i + 1, r = l + 1, w = i;
        if (l < n && cand_better(&h[w], &h[l])) w = l;
        if (r < n && cand_better(&h[w], &h[r])) w = r;
        if (w == i) return;
        cand_t t = h[i]; h[i] = h[w]; h[w] = t; i = w;
    }
}
static int topk_push(cand_t* h, int n, int k, cand_t c) {
    if (n < k) {
        h[n] = c; int i = n;
        while (i > 0) { int p = (i - 1) / 2; if (cand_better(&h[p], &h[i])) { cand_t t = h[p]; h[p] = h[i]; h[i] = t; i = p; } else break; }
        return n + 1;
    }
    if (cand_better(&c, &h[0])) { h[0] = c; heap_sift(h, n, 0); }
    return n;
}

/* select_topk canonicalised: the k best of (idx, -neg) by (score desc, idx asc),
 * returned sorted.  The reference returns the same SET in arbitrary order
 * (np.argpartition) when there are no ties at the cut. Returns count. */
int64_t oracle_select_topk(const int64_t* idx, const float* neg, int64_t m, int k,
                           int64_t* out_idx, float* out_score) {
    cand_t* h = (cand_t*)malloc(sizeof(cand_t) * (size_t)(k > 0 ? k : 1));
    int n = 0;
    for (int64_t j = 0; j < m; ++j) { cand_t c = { -neg[j], idx[j] }; n = topk_push(h, n, k, c); }
    qsort(h, (size_t)n, sizeof(cand_t), cand_cmp_desc);
    for (int j = 0; j < n; ++j) { out_idx[j] = h[j].i; out_score[j] = h[j].s; }
    free(h);
    return n;
}

/* Whole retrieve loop of indexer.py:405-474 for a batch of queries:
 * q_threads query-level workers (reference: 4), each scoring with
 * inner_threads posting-loop threads.  out_idx/out_score: [Nq, k], padded with
 * -1 / 0; out_count: [Nq]. */
void oracle_sparse_retrieve(const int64_t* indptr, const int32_t* doc_ids, const float* vals,
                            const int64_t* q_indptr, const int32_t* q_cols, const float* q_vals,
                            int64_t Nq, int k, float threshold, int64_t N,
                            int64_t* out_idx, float* out_score, int64_t* out_count,
                            int q_threads, int inner_threads) {
#ifdef _OPENMP
    omp_set_max_active_levels(2);
#endif
#pragma omp parallel num_threads(q_threads)
    {
        float* scores = (float*)malloc(sizeof(float) * (size_t)N);
        int64_t* fi = (int64_t*)malloc(sizeof(int64_t) * (size_t)N);
        float* fn = (float*)malloc(sizeof(float) * (size_t)N);
#pragma omp for schedule(dynamic, 1)
        for (int64_t q = 0; q < Nq; ++q) {
            const int64_t b = q_indptr[q];
            const int nt = (int)(q_indptr[q + 1] - b);
            int64_t m = oracle_sparse_score(indptr, doc_ids, vals, q_cols + b, q_vals + b, nt,
                                            threshold, N, scores, fi, fn, inner_threads);
            int64_t c = oracle_select_topk(fi, fn, m, k, out_idx + q * k, out_score + q * k);
            for (int64_t j = c; j < k; ++j) { out_idx[q * k + j] = -1; out_score[q * k + j] = 0.f; }
            out_count[q] = c;
        }
        free(scores); free(fi); free(fn);
    }
}

/* ---- dense: exact inner product, k-ordered fp32 FMA chain ----------------- */
/* Bit-level model of the gfx950 f32 MFMA accumulation used by the HIP kernel:
 * acc = fmaf(q[k], d[k], acc) over k in the order given by `korder` (H entries),
 * starting from 0.  With korder = identity this is a plain sequential dot. */
void oracle_dense_scores_fma(const float* Q, const float* D, int64_t Nq, int64_t N, int H,
                             const int32_t* korder, float* out /* [Nq, N] */) {
#pragma omp parallel for schedule(static)
    for (int64_t d = 0; d < N; ++d) {
        const float* dr = D + d * (int64_t)H;
        for (int64_t q = 0; q < Nq; ++q) {
            const float* qr = Q + q * (int64_t)H;
            float acc = 0.f;
            for (int i = 0; i < H; ++i) { int kk = korder ? korder[i] : i; acc = fmaf(dr[kk], qr[kk], acc); }
            out[q * N + d] = acc;
        }
    }
}

/* Top-k of a dense score matrix by (score desc, doc index asc); rows padded
 * with (-FLT_MAX, -1) when k > N, as faiss does. */
void oracle_topk_rows(const float* S, int64_t Nq, int64_t N, int k, float* out_score, int64_t* out_idx) {
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t q = 0; q < Nq; ++q) {
        cand_t* h = (cand_t*)malloc(sizeof(cand_t) * (size_t)k);
        int n = 0;
        for (int64_t d = 0; d < N; ++d) { cand_t c = { S[q * N + d], d }; n = topk_push(h, n, k, c); }
        qsort(h, (size_t)n, sizeof(cand_t), cand_cmp_desc);
        for (int j = 0; j < n; ++j) { out_score[q * k + j] = h[j].s; out_idx[q * k + j] = h[j].i; }
        for (int j = n; j < k; ++j) { out_score[q * k + j] = -3.402823466e38f; out_idx[q * k + j] = -1; }
        free(h);
    }
}


/* ---- dense: faiss IndexFlatIP.search restated as the library runs it (faiss-cpu 1.8.0, IndexFlat.cpp ->
 * knn_inner_product -> exhaustive_inner_product_blas): sgemm over (query block, database block) pairs, then one heap per
 * query fed from the block's score rows (HeapBlockResultHandler: a score enters when it beats the heap's worst).  The sgemm
 * is the host BLAS (called from oracle/scoring.py); this is the heap side, one OpenMP thread per query row.
 * heaps: [Nq, k] cand_t min-heaps on (score, then higher doc index = worse), heap_n: [Nq] fill counts. */
void oracle_heap_block(const float* S, int64_t Nq, int64_t nb, int64_t ld, int64_t base, int k, void* heaps, int32_t* heap_n, int threads) {
    cand_t* H = (cand_t*)heaps;
#pragma omp parallel for schedule(static) num_threads(threads > 0 ? threads : 1)
    for (int64_t q = 0; q < Nq; ++q) {
        cand_t* h = H + q * (int64_t)k;
        int n = heap_n[q];
        const float* row = S + q * ld;
        for (int64_t j = 0; j < nb; ++j) {
            if (n == k && !(row[j] > h[0].s || (row[j] == h[0].s && base + j < h[0].i))) continue;
            cand_t c = { row[j], base + j };
            n = topk_push(h, n, k, c);
        }
        heap_n[q] = n;
    }
}

void oracle_heap_finish(void* heaps, const int32_t* heap_n, int64_t Nq, int k, float* out_score, int64_t* out_idx) {
    cand_t* H = (cand_t*)heaps;
#pragma omp parallel for schedule(static)
    for (int64_t q = 0; q < Nq; ++q) {
        cand_t* h = H + q * (int64_t)k;
        const int n = heap_n[q];
        qsort(h, (size_t)n, sizeof(cand_t), cand_cmp_desc);
        for (int j = 0; j < n; ++j) { out_score[q * k + j] = h[j].s; out_idx[q * k + j] = h[j].i; }
        for (int j = n; j < k; ++j) { out_score[q * k + j] = -3.402823466e38f; out_idx[q * k + j] = -1; }
    }
}
