/* sr_hip.h - C ABI of libsr_hip.so, the MI355X (gfx950) encode + retrieval hot path
 * of scaling-retriever.
 *
 * Every entry point replaces one piece of the reference's Python hot path; the
 * reference interface it stands in for is cited as file:line into
 * HansiZeng/scaling-retriever.  Plain C types only: opaque handles, device
 * pointers (what torch's tensor.data_ptr() returns), sizes, and a hipStream_t
 * passed as void*.  All functions return 0 on success, non-zero on error
 * (sr_last_error() then holds a message for the calling thread).  The library
 * never frees caller memory; output buffers are caller-allocated.
 *
 * Threading: the reference calls its scorer from 4 Python threads with the GIL
 * released (scaling_retriever/indexer.py:325,459), so search entry points are
 * re-entrant: each handle owns its workspace and serialises concurrent calls on
 * an internal mutex (the GPU runs a whole query batch per call; threads are a
 * CPU artefact of the reference) and, when those calls arrive on different
 * streams, chains their device work through an event so that the workspace is
 * never used by two calls at once.  Different handles never share mutable state.
 */
#ifndef SR_HIP_H
#define SR_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define SR_OK 0
#define SR_ERR_INVALID 1
#define SR_ERR_HIP 2
#define SR_ERR_NOMEM 3
#define SR_ERR_UNSUPPORTED 4

#define SR_DTYPE_F32 0
#define SR_DTYPE_BF16 1

typedef void* sr_stream; /* hipStream_t; NULL = default stream */

const char* sr_last_error(void);
int sr_version(void);
/* Largest k supported by the fused top-k (LDS sort width). */
int sr_max_topk(void);

/* ------------------------------------------------------------------ dense ---
 * Replaces DenseFlatIndexer.init_index / index_data / search_knn
 * (scaling_retriever/indexer.py:191-217) over faiss.IndexFlatIP:
 * exact fp32 inner products, k best per query, sorted descending; ties broken
 * by ascending doc index; rows padded with (-FLT_MAX, -1) when ntotal < k.   */
typedef struct sr_dense_index sr_dense_index;

int sr_dense_index_create(sr_dense_index** out, int dim);
/* Registers a segment of `n_rows` fp32 row-major [n_rows, dim] embeddings that
 * already live in device memory (non-owning view; caller keeps it alive).
 * Global doc index of local row r is id_base + r * id_stride (id_stride = W and
 * id_base = rank reproduces the reference's g_row = row*W + rank sharding,
 * indexer.py:262).  Global indices must fit in 32 bits.                      */
int sr_dense_index_add(sr_dense_index* idx, const float* d_rows, int64_t n_rows,
                       int64_t id_base, int64_t id_stride);
int64_t sr_dense_index_ntotal(const sr_dense_index* idx);
/* d_queries: fp32 [nq, dim] on device.  d_out_scores fp32 [nq, k],
 * d_out_ids int64 [nq, k] (global doc indices).                              */
int sr_dense_search(sr_dense_index* idx, const float* d_queries, int64_t nq, int k,
                    float* d_out_scores, int64_t* d_out_ids, sr_stream stream);
/* Arithmetic of the score kernel for query batches > 64:
 *   SR_PRECISION_FP32   (default) exact fp32 MFMA, bit-for-bit a k-ordered fmaf chain;
 *   SR_PRECISION_BF16X3 fp32 operands split into bf16 hi + lo, q.d ~= qh.dh + qh.dl + ql.dh on the
 *                       bf16 MFMA pipe with fp32 accumulation: ~1e-7 relative to the fp32 result
 *                       (the size of an fp32 summation-order change), ~3x faster.  Keeps a
 *                       library-owned bf16 copy of every segment (same bytes as the fp32 rows). */
#define SR_PRECISION_FP32 0
#define SR_PRECISION_BF16X3 1
/*   SR_PRECISION_BF16X6 three bf16 planes per operand (the full 24-bit significand), six products:
 *                       the error class of an fp32 dot product (measured vs float64 like the exact
 *                       path), ~1.7x faster than fp32 MFMA; keeps three bf16 planes per segment.  */
#define SR_PRECISION_BF16X6 2
/*   SR_PRECISION_FP32_FILTERED  the SAME results as SR_PRECISION_FP32, bit for bit, several times faster for batches
 *                       > 64 queries: one fp16 plane product (fp16 rounding of the query x fp16 rounding of the document,
 *                       both scaled by powers of two) plus a per-pair error term e(q, j) built from the ACTUAL rounding
 *                       residuals |d - d0|, |q - q0| (Cauchy-Schwarz) gives an upper bound U >= the exact fp32 score of
 *                       every pair; the 3k documents with the largest U are the candidates, those that can still reach
 *                       the top-k are re-scored with the exact fp32 fmaf chain, and the certificate U_3k < (k-th exact
 *                       score found) proves that the result is the exact top-k.  A query without a certificate (more
 *                       than 2k near-ties at the cut, a zero or non-finite query) is re-done by the exact kernel, that
 *                       query alone.  Keeps one fp16 plane + 8 bytes per document (half the bytes of the fp32 rows);
 *                       without room for it, or for data that is not finite, the exact kernel is used.            */
#define SR_PRECISION_FP32_FILTERED 3
int sr_dense_index_set_precision(sr_dense_index* idx, int mode);
/* Doc-sharded search in two halves (replaces nothing in the reference, which scores on ONE process: eval_dense.py:191; this
 * is the multi-GPU row of SURVEY.md 8e).  Every rank calls _begin on its shard: d_lower [nq] receives, per query, a value that
 * at least ceil(k / share) documents of THIS shard reach exactly (share = number of shards).  The minimum of d_lower over the
 * shards (one small all-reduce) is therefore not above the global k-th exact score; every rank passes it to _finish as
 * d_threshold and re-scores only the candidates that can reach the GLOBAL top-k (about k / share of them instead of k).  The
 * shard's [nq, k] output then holds those (padding: score -FLT_MAX, id -1); sr_topk_merge of the shards' outputs is the global
 * top-k, bit for bit what one index over all documents returns.  When the certified filter does not apply, d_lower is -inf and
 * _finish is a plain sr_dense_search (d_threshold may be null).  Do not interleave other searches on the handle between the two. */
int sr_dense_search_begin(sr_dense_index* idx, const float* d_queries, int64_t nq, int k, int share,
                          float* d_lower, sr_stream stream);
int sr_dense_search_finish(sr_dense_index* idx, const float* d_queries, int64_t nq, int k, const float* d_threshold,
                           float* d_out_scores, int64_t* d_out_ids, sr_stream stream);
/* searches of more than 64 queries answered by the filter alone / with some (or all) queries re-done by the exact kernel */
int sr_dense_index_filter_stats(sr_dense_index* idx, int64_t* n_filtered, int64_t* n_fallback);
/* the same per query: queries certified by the filter / re-done by the exact kernel so far */
int sr_dense_index_filter_query_stats(sr_dense_index* idx, int64_t* n_certified, int64_t* n_redone);
/* Workspace ceiling in bytes for candidate buffers (default 4 GiB). */
int sr_dense_index_set_workspace_limit(sr_dense_index* idx, int64_t bytes);
/* on != 0: batches of <= 64 queries run the tiled kernels instead of the streaming one, so that EVERY batch size accumulates in
 * one k order: a query's ids and fp32 scores are then the same bits alone, in a batch of 8 and in a batch of 6 980 (cost: a pass
 * of <= 32 queries reads D at ~4.4 TB/s instead of ~5.9).  Off (default): batches <= 64 and larger ones are each exact fp32 chains
 * but differ in the last bit - as faiss's IndexFlatIP.search does between its small-batch loop and its sgemm path (the switch at
 * 20 queries that /root/reference/scaling_retriever/indexer.py:211 inherits).                                                     */
int sr_dense_index_set_batch_invariant(sr_dense_index* idx, int on);
int sr_dense_index_destroy(sr_dense_index* idx);
/* Measurement hook: while enabled, every launch of the score kernel is bracketed by HIP
 * events on the search stream.  _read synchronises those events and returns the number
 * of launches, their summed duration (ms) and the algorithmic work they covered
 * (2*nq*rows*dim FLOP, rows*dim*4 bytes of D), then clears the log.              */
int sr_dense_index_profile(sr_dense_index* idx, int enable);
int sr_dense_index_profile_read(sr_dense_index* idx, int64_t* n_launches, double* total_ms,
                                double* total_flop, double* total_d_bytes);

/* ----------------------------------------------------------------- sparse ---
 * Replaces SparseRetrieval.numba_score_float + select_topk
 * (scaling_retriever/indexer.py:315-344): per query, term-serial
 * scores[doc] += q_t * v (unfused fp32 multiply-add, ascending query-term
 * order), keep docs with score > threshold, k best; sorted descending, ties by
 * ascending doc index.  The index is the IndexDictOfArray posting lists
 * (scaling_retriever/utils/inverted_index.py:15-105) laid out as CSR by term. */
typedef struct sr_sparse_index sr_sparse_index;

/* d_indptr int64 [n_terms+1], d_doc_ids int32 [nnz], d_vals fp32 [nnz], all on
 * device (non-owning views).  Inside each posting list doc ids must be strictly
 * ascending (checked).  n_docs = IndexDictOfArray.nb_docs().                  */
int sr_sparse_index_create(sr_sparse_index** out, const int64_t* d_indptr,
                           const int32_t* d_doc_ids, const float* d_vals,
                           int64_t n_terms, int64_t n_docs, sr_stream stream);
/* Queries as CSR: d_q_indptr int64 [nq+1] (non-decreasing), d_q_cols int32, d_q_vals fp32 (term
 * order inside a query = accumulation order).  A term id outside [0, n_terms) is an
 * empty posting list, as in the reference's vocabulary-filled dict (indexer.py:364-370).  Outputs [nq, k] padded with
 * (0, -1); d_out_counts int32 [nq] = number of valid entries per row.
 * Global doc index = id_base + doc * id_stride.                              */
int sr_sparse_search(sr_sparse_index* idx, const int64_t* d_q_indptr, const int32_t* d_q_cols,
                     const float* d_q_vals, int64_t nq, int k, float threshold,
                     int64_t id_base, int64_t id_stride,
                     float* d_out_scores, int64_t* d_out_ids, int32_t* d_out_counts,
                     sr_stream stream);
int sr_sparse_index_set_workspace_limit(sr_sparse_index* idx, int64_t bytes);
int sr_sparse_index_destroy(sr_sparse_index* idx);
/* Measurement hook as for the dense index; algorithmic bytes = 8 B per posting of the
 * query terms that falls in the launched doc tiles (computed on the device).       */
/* Work counters of the query-block kernel (measurement hook): while enabled, every workgroup adds what it loads and applies
 * to six device counters; each call returns them (out6 may be null) and resets them.  out6[0] dense columns loaded (one =
 * 4096 floats), [1] (column, query) applications (one = 4096 unfused multiply-adds), [2] / [3] postings loaded by the one-step /
 * grouped scatter runs (8 bytes and one LDS read-modify-write each), [4] plan entries fetched, [5] (query block, doc tile)
 * workgroups.  bench.py prices the kernel's VALU and L2 bounds from these.                                                  */
int sr_sparse_index_work_counters(sr_sparse_index* idx, int enable, uint64_t* out6);
int sr_sparse_index_profile(sr_sparse_index* idx, int enable);
int sr_sparse_index_profile_read(sr_sparse_index* idx, int64_t* n_launches, double* total_ms,
                                 double* total_posting_bytes);
/* Which scoring path served the searches so far (both return the same bits): n_dense_terms = heavy
 * terms (present in at least a quarter of the docs) that also have a dense column; n_block_calls =
 * sr_sparse_search calls that ran the 4-queries-per-workgroup kernel; n_fallback_calls = those of them
 * in which at least one block of 4 queries went through the per-query kernel because a query listed
 * its terms in non-ascending order (the accumulation order the reference follows,
 * indexer.py:315-328, is the query's own).                                        */
int sr_sparse_index_block_stats(sr_sparse_index* idx, int64_t* n_dense_terms, int64_t* n_block_calls,
                                int64_t* n_fallback_calls);

/* The certified two-stage scorer (csrc/sparse_cert.hip).  For an index without negative values sr_sparse_index_create also
 * builds: fp16 MFMA tiles of the heaviest terms, 4-byte packed postings, a per-term table of doc-tile boundaries and a
 * doc-major forward index.  sr_sparse_search then scores every (query, doc) approximately with a proven error bound (matrix
 * pipe for the heavy terms, 16-bit fixed-point LDS atomics for the others), keeps the k + 1024 best keys, certifies that the
 * true top-k lies among them, re-scores those candidates with the reference's exact fp32 chain
 * (scaling_retriever/indexer.py:324-340) from the forward index and returns their exact top-k: the same bits as the exact
 * kernels.  Queries it cannot certify (a negative or unordered query, more than 256 rare terms, a band of undecided keys
 * wider than 1024, fewer than k docs with a non-zero key) are re-done by the exact kernels inside the same call.
 * out8: [0] 1 if this index has the scorer, [1] heavy terms on the matrix pipe, [2] searches it ran, [3] queries it was
 * given, [4] queries re-done by the exact kernels, [5] doc tiles of 1024, [6] candidates whose exact chain the certified
 * path ran (rounded up to 16 per query), [7] query batches (of up to 8 192 queries: the scorer's per-call workspace is ~200 KB per
 * query) served by the exact kernels because that workspace did not fit in device memory.
 * Environment (read at sr_sparse_index_create): SR_SPARSE_SCORER=exact builds the index without the scorer (its side structures
 * take ~22 bytes per posting + 8 bytes per (term, 1 024-doc tile), at most half of the free device memory: 25 GB at the MS MARCO
 * shape); SR_LOG=1 prints one line per index on stderr saying what was built, with how many bytes, or why not.
 * Limits of the fast path, beyond which a query is served by the exact kernels (same results, about a quarter of the speed):
 * k <= SR_MAX_TOPK - 1024 = 3 072 (the band of extra keys: 1 024, or 2 048 / 3 072 for batches whose queries bring more than 96 / 160
 * terms outside the index's 128 heaviest ones, capped at SR_MAX_TOPK - k), <= 256 query terms (rare terms beyond the first 64 take a
 * slower walk inside the same kernel), terms strictly ascending, values >= 0, at least 8 (k + 1 024) documents in the collection.                            */
int sr_sparse_index_cert_stats(sr_sparse_index* idx, int64_t* out8);
/* Test hook for the error bound: enable = 1 / 0 switches the recording of the stage-1 keys of every (query, doc) pair on /
 * off; enable = 2 copies the last search's keys to h_keys uint16 [nq_pad][n_tiles * 1024] (nq_pad = nq rounded up to 32)
 * and the per-query constants to h_consts fp32 [nq_pad][6] = {c_q (0: query outside the fast path), s_q, rare terms in
 * stage 1, query terms, rare terms left out of stage 1 (weight below fp16's normal range), the k-th best key the scan's
 * last top-k select saw (0: none ran) - the cut below it is the scan's filter threshold}; *vscale, *T = the index-side
 * constants.  With true_fix = 65535 * s_q * (real-arithmetic score) every key obeys
 * true_fix (1 - dd) - 1.2 - 2.03 * left out <= key <= true_fix (1 + dd) + 1.2 + 1.01 * rare terms.                          */
int sr_sparse_index_cert_debug(sr_sparse_index* idx, int enable, uint16_t* h_keys, int64_t keys_capacity,
                               float* h_consts, int64_t nq_pad, float* vscale, int32_t* T);

/* On-device index build: doc-major postings -> CSR by term.  Replaces the per-posting Python append of
 * IndexDictOfArray.add_batch_document (scaling_retriever/utils/inverted_index.py:67-76) and the per-term concatenation of
 * merge_indexes (:108-170) behind SparseIndexer.index (scaling_retriever/indexer.py:239-308).
 * d_rows int32 [nnz] = global doc row of every posting, d_cols int32 [nnz] = term in [0, n_terms), d_vals fp32 [nnz], in
 * insertion order.  A stable radix sort by term (this library's kernels) keeps the insertion order inside a term - the
 * reference's posting order; sort_docs = 1 additionally orders every posting list by ascending doc row (needs n_docs >
 * every row), which is what sr_sparse_index_create requires of a merged multi-rank index.  Outputs: d_indptr int64
 * [n_terms + 1], d_out_rows int32 [nnz], d_out_vals fp32 [nnz] (must not alias the inputs).  Synchronises the stream.
 * Errors: SR_ERR_INVALID for a term outside [0, n_terms), a negative row, or (sort_docs) a row >= n_docs - reported when the
 * call returns; the outputs then hold the postings in no defined order (the digits are masked: never a write outside the
 * arrays).  SR_ERR_NOMEM when the ping-pong buffers (4 bytes x nnz x 4, x 6 with three or more passes) cannot be allocated.  */
/* Term of every posting of a CSR-by-term index: d_out_terms int32 [nnz][p] = t for d_indptr[t] <= p < d_indptr[t + 1]
 * (the per-term arrays of IndexDictOfArray, inverted_index.py:22-55, flattened back to triples for a re-sort).            */
int sr_sparse_csr_expand_terms(const int64_t* d_indptr, int64_t n_terms, int64_t nnz, int32_t* d_out_terms,
                               sr_stream stream);
int sr_sparse_csr_build(const int32_t* d_rows, const int32_t* d_cols, const float* d_vals, int64_t nnz,
                        int64_t n_terms, int64_t n_docs, int sort_docs, int64_t* d_indptr,
                        int32_t* d_out_rows, float* d_out_vals, sr_stream stream);

/* ------------------------------------------------------------ top-k merge ---
 * The one exchange step of doc-sharded retrieval: merge `n_lists` per-shard
 * top-k lists (after the RCCL gather) into the global top-k per query.
 * d_scores fp32 [n_lists, nq, k], d_ids int64 [n_lists, nq, k] (ids < 0 = pad).
 * Outputs as sr_dense_search (pad_score fills unused slots).                  */
int sr_topk_merge(const float* d_scores, const int64_t* d_ids, int n_lists, int64_t nq, int k,
                  float pad_score, float* d_out_scores, int64_t* d_out_ids, sr_stream stream);

/* ---------------------------------------------------------------- encoder ---
 * Replaces LlamaBiDense / LlamaBiSparse .encode / .query_encode / .doc_encode
 * (scaling_retriever/modeling/llm_encoder.py:66-70,186-196,424-443) and the
 * LlamaBiModel forward they call (modeling/bidirectional_llama.py:67-188 over
 * transformers' LlamaModel).                                                  */
typedef struct sr_model sr_model;

typedef struct {
    int32_t vocab_size, hidden_size, intermediate_size, num_layers;
    int32_t num_heads, num_kv_heads, head_dim;
    float rms_norm_eps;
    float rope_theta;
    int32_t rope_llama3;              /* 0 = default rope, 1 = "llama3" scaling */
    float rope_factor, rope_low_freq_factor, rope_high_freq_factor;
    int32_t rope_original_max_pos;
    int32_t tie_word_embeddings;      /* lm_head shares embed_tokens */
    int32_t has_lm_head;              /* 0: LlamaBiModel (dense), 1: LlamaBiForMNTP (sparse) */
    int32_t max_batch_tokens;         /* workspace sizing: max packed tokens per encode call */
    int32_t max_batch_seqs;
    int32_t fp32_planes;              /* fp32 regime (sr_encode_*_fp32): 0 = not available; 16 = every weight also kept as two
                                         fp16 planes of power-of-two scaled rows (22 significand bits, 3 plane products
                                         per GEMM: truncation below the fp32 accumulation's own rounding); 3 = three bf16
                                         planes (24 bits, 6 products); 2 = two bf16 planes (3 products, ~2^-17)        */
} sr_model_config;

int sr_model_create(sr_model** out, const sr_model_config* cfg);
/* Copies one checkpoint tensor (HF Llama naming, e.g.
 * "model.layers.3.self_attn.q_proj.weight") from device memory into the
 * model's internal layout.  dtype = SR_DTYPE_F32 or SR_DTYPE_BF16.           */
int sr_model_set_weight(sr_model* m, const char* name, const void* d_ptr, int dtype,
                        int64_t rows, int64_t cols, sr_stream stream);
/* Verifies all tensors were provided. */
int sr_model_finalize(sr_model* m);
/* d_input_ids / d_attention_mask: int64 [B, L] on device (the tokenizer
 * collator's output, data_collator.py:177-190).  d_out: fp32 [B, hidden].    */
int sr_encode_dense(sr_model* m, const int64_t* d_input_ids, const int64_t* d_attention_mask,
                    int32_t B, int32_t L, float* d_out, sr_stream stream);
/* d_out: fp32 [B, vocab]. */
int sr_encode_sparse(sr_model* m, const int64_t* d_input_ids, const int64_t* d_attention_mask,
                     int32_t B, int32_t L, float* d_out, sr_stream stream);
/* The two calls above are the reference's torch.autocast(bf16) regime (documents: indexer.py:46-52,
 * :255-256; sparse queries: :390-391): bf16 GEMM inputs, fp32 accumulation, fp32 everything else.
 * The _fp32 variants are its NO-autocast regime - dense queries (eval_dense.py:94-106, no autocast at
 * :101-102) and examples/quick_start.py: every nn.Linear is an fp32 GEMM and SDPA runs on fp32
 * operands.  Same arguments and outputs; needs a model created with fp32_planes > 0
 * (SR_ERR_INVALID otherwise).                                                */
int sr_encode_dense_fp32(sr_model* m, const int64_t* d_input_ids, const int64_t* d_attention_mask,
                         int32_t B, int32_t L, float* d_out, sr_stream stream);
int sr_encode_sparse_fp32(sr_model* m, const int64_t* d_input_ids, const int64_t* d_attention_mask,
                          int32_t B, int32_t L, float* d_out, sr_stream stream);
/* Both heads from ONE backbone pass - the hybrid model HybridIndexer / HybridRetriever drive
 * (`batch_sparse_reps, batch_dense_reps = self.model.encode(**inputs)`, indexer.py:764, :939):
 * d_out_sparse fp32 [B, vocab], d_out_dense fp32 [B, hidden]; fp32 = 0 autocast regime, 1 fp32 regime. */
int sr_encode_both(sr_model* m, const int64_t* d_input_ids, const int64_t* d_attention_mask,
                   int32_t B, int32_t L, int32_t fp32, float* d_out_sparse, float* d_out_dense,
                   sr_stream stream);
/* Rows of SEVERAL collator batches in one call.  The reference's drivers hand the encoder eval_batch_size rows at a time
 * (DenseRetriever.generate_query_vecs, eval_dense.py:94-106; SparseRetrieval._generate_query_vecs, indexer.py:382-403; default 128
 * queries = ~1 100 tokens, far too few rows to fill 256-row GEMM tiles).  The host side lays the batches into ONE [B, L] matrix
 * (L = the widest batch; a narrower batch gets extra left padding, mask 0) and passes d_row_shift int32 [B]: how many columns row b
 * moved right.  Positions (RoPE, the dense head's `[-length:]` pooling) are counted from there, i.e. every row keeps the
 * position_ids it had in its own batch, and its output is bit-identical to encoding that batch alone.  d_row_shift may be NULL
 * (no shifts).  mode: 0 dense head, 1 sparse head, 2 both; fp32: 0 autocast regime, 1 fp32 regime.  d_out_sparse fp32 [B, vocab]
 * (modes 1, 2), d_out_dense fp32 [B, hidden] (modes 0, 2).                                                                      */
int sr_encode_rows(sr_model* m, const int64_t* d_input_ids, const int64_t* d_attention_mask, int32_t B, int32_t L,
                   const int32_t* d_row_shift, int32_t mode, int32_t fp32, float* d_out_sparse, float* d_out_dense,
                   sr_stream stream);
/* Debug/test hook: last_hidden_state (after the final norm) of the packed
 * tokens of the last encode call, fp32 [n_tokens, hidden]; returns n_tokens
 * through *n_tokens.                                                         */
int sr_model_last_hidden(sr_model* m, float* d_out, int64_t capacity_rows, int64_t* n_tokens, sr_stream stream);
int sr_model_destroy(sr_model* m);

/* peft merge_and_unload for one Linear (llm_encoder.py:116-122,502-508):
 * W[out,in] += scale * B[out,r] @ A[r,in], fp32 in place, scale = alpha / r. */
int sr_lora_merge(float* d_W, const float* d_A, const float* d_B, int64_t out_features,
                  int64_t in_features, int32_t r, float scale, sr_stream stream);

/* nonzero -> (row, col, val) compaction of sparse reps [B, V]
 * (indexer.py:259-260 torch.nonzero + gather; _generate_query_vecs :393-399).
 * d_row_ptr int64 [B+1] receives CSR row offsets, d_cols int32 / d_vals fp32
 * [capacity] the entries (cols ascending inside a row, like torch.nonzero).
 * *h_nnz gets the total (call synchronises the stream). Returns SR_ERR_NOMEM
 * if capacity is too small (then *h_nnz holds the needed size).               */
int sr_sparse_compact(const float* d_reps, int64_t B, int64_t V, int64_t* d_row_ptr,
                      int32_t* d_cols, float* d_vals, int64_t capacity, int64_t* h_nnz,
                      sr_stream stream);

/* run.json of the retrieval drivers, written straight from the result arrays (HOST function: every pointer is host memory).
 * Replaces the per-hit Python loops + json.dump of eval_dense.py:225-241 (`qid_to_rankdata[str(qid)][str(docid)] = float(score)`)
 * and SparseRetrieval.retrieve, indexer.py:405-474,530-540 (`res[str(qid)][str(doc_ids[id_])] = float(sc)`; `json.dump(res)`):
 * the file holds byte for byte what Python's json.dump writes for that nested dict - {"qid": {"docid": score, ...}, ...},
 * ", " / ": " separators, entries in row order, scores as float.__repr__ of the fp32 value widened to double.
 * h_scores fp32 [nq, k], h_idx int64 [nq, k] = positions in the document id table (negative = padding, skipped), h_counts int32
 * [nq] or NULL = hits per row (NULL: k).  A query without a hit gets no entry, as in the reference.  Keys: decimal int64
 * (h_qid_i64 [nq] / h_doc_i64 [n_docs]) or, when the *_i64 pointer is NULL, JSON string bodies already escaped by the caller:
 * bytes + offsets [n + 1], or - offsets NULL - fixed-width NUL-padded entries of *_width bytes (a numpy 'S' array of ASCII ids
 * that need no escaping).  Keys are assumed distinct (the caller checks; a dict would merge duplicates).
 * n_threads <= 0: all hardware threads.  *bytes_written (nullable) = file size.                                             */
int sr_write_run_json(const char* path, int64_t nq, int64_t k, const float* h_scores, const int64_t* h_idx, const int32_t* h_counts,
                      const int64_t* h_qid_i64, const char* h_qid_bytes, const int64_t* h_qid_off, int64_t qid_width,
                      const int64_t* h_doc_i64, const char* h_doc_bytes, const int64_t* h_doc_off, int64_t doc_width,
                      int64_t n_docs, int32_t n_threads, int64_t* bytes_written);
/* The same file in pieces, so that the host writes piece c while the GPU searches piece c + 1 (eval_dense.py:225-241 writes after the whole
 * search): part 1 = first piece (creates the file, leaves it open-ended), 2 = a middle piece, 3 = the last piece (closes the object).
 * The finished file holds the bytes of one sr_write_run_json call over the concatenated pieces.  *bytes_written = size so far.      */
int sr_write_run_json_part(const char* path, int32_t part, int64_t nq, int64_t k, const float* h_scores, const int64_t* h_idx,
                           const int32_t* h_counts, const int64_t* h_qid_i64, const char* h_qid_bytes, const int64_t* h_qid_off,
                           int64_t qid_width, const int64_t* h_doc_i64, const char* h_doc_bytes, const int64_t* h_doc_off,
                           int64_t doc_width, int64_t n_docs, int32_t n_threads, int64_t* bytes_written);
/* Test hook: rounds of sr_write_run_json that were copied through the shared file mapping (rounds of >= 8 MB with more than
 * one thread) since the library was loaded; the other rounds are written with pwrite.                                       */
int64_t sr_run_writer_mapped_rounds(void);

/* ----------------------------------------------------- building blocks ---
 * The two MFMA kernels of the encoder, exported for per-kernel parity tests and
 * profiling (they are what sr_encode_* launches per layer).
 * sr_gemm_bf16: y = A[M,K] @ W[N,K]^T, bf16 inputs, fp32 accumulate.
 *   epilogue 0: C bf16 [M,N];  1: C fp32 [M,N] += y;  2: SwiGLU, W rows interleaved
 *   gate/up in 16-row blocks, C bf16 [M,N/2];  3: per-sequence max over token rows,
 *   C fp32 [n_seq,N] (pre-zeroed), d_seq_of int32 [M];  4: C fp32 [M,N].
 * sr_attention_varlen: bidirectional GQA attention over packed sequences with the
 *   RoPE rotation fused; d_qkv bf16 [T,(nh+2nkv)*hd], d_out bf16 [T,nh*hd],
 *   d_cu_seqlens int32 [B+1], d_pos int32 [T], d_key_valid uint8 [T],
 *   d_rope_cos/sin fp32 [max_pos, hd/2]; pass NULL for both when d_qkv is already
 *   rotated (sr_gemm_qkv_rope output) - that is the form sr_encode_* uses.      */
int sr_gemm_bf16(const void* d_A, const void* d_W, int32_t M, int32_t N, int32_t K, int32_t epilogue,
                 void* d_C, const int32_t* d_seq_of, sr_stream stream);
/* QKV projection with the RoPE rotation fused into the epilogue (fp32, HF rotate_half layout):
 * C bf16 [M,N]; features [0, n_rope) = q heads then k heads are rotated with the angle of
 * d_pos[m], features [n_rope, N) (v heads) are stored as is.                      */
/* The GEMM of the encoder's fp32 regime on fp16 planes (fp32_planes = 16): d_A [M, K] / d_W [N, K] fp16 plane segments of rows
 * scaled by powers of two, d_a_scale [M] / d_w_scale [N] the inverse scales; C fp32 [M, N] += (A W^T) a_scale[m] w_scale[n]. */
int sr_gemm_f16_scaled(const void* d_A, const void* d_W, int32_t M, int32_t N, int32_t K, const float* d_a_scale,
                       const float* d_w_scale, float* d_C, sr_stream stream);
int sr_gemm_qkv_rope(const void* d_A, const void* d_W, int32_t M, int32_t N, int32_t K, void* d_C,
                     const int32_t* d_pos, const float* d_rope_cos, const float* d_rope_sin,
                     int32_t n_rope, int32_t head_dim, sr_stream stream);
int sr_attention_varlen(const void* d_qkv, void* d_out, const int32_t* d_cu_seqlens, const int32_t* d_pos,
                        const uint8_t* d_key_valid, const float* d_rope_cos, const float* d_rope_sin,
                        int32_t B, int32_t num_heads, int32_t num_kv_heads, int32_t head_dim, sr_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* SR_HIP_H */
