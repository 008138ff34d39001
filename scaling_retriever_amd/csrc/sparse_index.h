// State of a sparse (inverted) index handle, shared by the exact scorer (sparse_score.hip), the certified MFMA scorer
// (sparse_cert.hip) and the on-device index build (sparse_build.hip).
#pragma once
#include "common.h"
#include <mutex>

struct SparseCert;      // sparse_cert.hip

struct sr_sparse_index {
    const int64_t* indptr = nullptr;
    const int32_t* doc_ids = nullptr;
    const float* vals = nullptr;
    int64_t n_terms = 0, n_docs = 0;
    int n_tiles = 0;
    int32_t* skip = nullptr;
    // heavy terms as dense columns (query-block kernel)
    int n_dense = 0;
    int64_t dense_stride = 0;
    float* dense = nullptr;
    int32_t* dense_slot = nullptr;
    // per-call plan of the query blocks
    int64_t plan_cap = 0, plan_blocks_cap = 0;
    int32_t* plan_term = nullptr;
    float* plan_w = nullptr;
    int32_t* plan_n = nullptr;
    uint8_t* plan_ok = nullptr;
    int64_t* plan_off = nullptr;
    unsigned long long* d_stamps = nullptr;   // dev switch SR_SPARSE_STAMPS
    unsigned long long* d_counters = nullptr; // sr_sparse_index_work_counters
    int* seg_cnt = nullptr;                   // wave-owned candidate regions of the query-block kernel: [q_batch][seg_cap]
    int64_t seg_q_cap = 0; int seg_cap = 0;
    bool count_work = false;
    int32_t* plan_perm = nullptr;     // every batch's queries in block order
    uint8_t* q_done = nullptr;
    int64_t plan_q_cap = 0;
    int* plan_bad = nullptr;
    int64_t n_block_calls = 0, n_fallback_calls = 0;
    int64_t ws_limit = 4ll << 30;
    TopkWS ws;
    StreamOrder order;
    LaunchProfile prof;
    unsigned long long* d_postings = nullptr;  // device counter of postings touched (profiling only)
    // certified two-stage scorer (null: the index or the device does not qualify; every search then runs the exact kernels)
    SparseCert* cert = nullptr;
    int64_t n_cert_no_memory = 0;       // query batches served by the exact kernels because the certified scorer's buffers did not fit
    int64_t n_cert_retries = 0;         // sub-batches of handed-back queries sent through the scorer again with the widest band
    std::mutex mu;
};

// ---- certified two-stage scorer (sparse_cert.hip) ----
// Builds the side structures (fp16 MFMA tiles of the heavy terms, packed postings, per-tile run table, doc-major forward index).
// SR_OK with idx->cert == nullptr when the index does not qualify (a negative or non-finite value, too few docs, no memory).
int sparse_cert_build(sr_sparse_index* idx, hipStream_t s);
void sparse_cert_destroy(SparseCert* c);
// Scores every query; d_uncert[q] = 1 marks the queries whose result rows were NOT written and must be served by the exact
// kernels (query outside the fast path's preconditions, or its candidate set could not be certified).  *n_uncert = their number
// (the call synchronises the stream once to read it).  *no_memory = true (with SR_OK): the per-call buffers of this batch did not fit
// in device memory; they were released, nothing was computed, and the caller serves the batch with the exact kernels.
int sparse_cert_search(sr_sparse_index* idx, const int64_t* d_q_indptr, const int32_t* d_q_cols, const float* d_q_vals, int64_t nq,
                       int k, float threshold, int64_t id_base, int64_t id_stride, float* d_out_scores, int64_t* d_out_ids,
                       int32_t* d_out_counts, uint8_t* d_uncert, int64_t* n_uncert, bool* no_memory, int band_keys, int* band_used, hipStream_t s);
// band_keys: keys the running set keeps beyond k (the certificate's room); 0 = chosen from the batch's largest rare-term count (1 024 /
// 2 048 / 3 072, capped at SR_MAX_TOPK - k); *band_used returns it.
void sparse_cert_count_retry(SparseCert* c, int64_t ns);
// [nq] flag bytes kept with the scorer (grown on demand); nullptr = out of device memory
uint8_t* sparse_cert_uncert_buffer(SparseCert* c, int64_t nq);
// queries per call of sparse_cert_search: its per-query workspace is ~200 KB (candidate slots of a launch, running set, approximate lists)
#define SR_CERT_QUERY_BATCH 8192

// ---- device-wide exclusive scan of int64 counts (sparse_build.hip): out[i] = sum_{j < i} in[j], out[n] = total; in == out allowed
int sr_device_exclusive_scan_i64(const int64_t* d_in, int64_t* d_out, int64_t n, hipStream_t s);
