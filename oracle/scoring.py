"""ORACLE (test infrastructure, not product code): restatement of the reference's
scoring stage.  numpy versions follow the reference line by line; the C versions
(oracle/score_cpu.c -> liboracle_score.so) are the same algorithms compiled, used
for larger parity cases and as bench.py's cpu_baseline ("port").

  sparse  numba_score_float / select_topk   /root/reference/scaling_retriever/indexer.py:315-344
          PINNED by tests/golden/sparse_score.npz
  dense   DenseFlatIndexer.search_knn       /root/reference/scaling_retriever/indexer.py:210-214
          over faiss IndexFlatIP [3P faiss-cpu==1.8.0, absent]: "parity unpinned",
          anchored on brute force Q @ D.T.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build_c(force=False):
    so = os.path.join(_HERE, "liboracle_score.so")
    if force or not os.path.exists(so):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return so


def clib():
    global _LIB
    if _LIB is None:
        so = build_c()
        try:
            _LIB = ctypes.CDLL(so)
        except OSError:
            _LIB = ctypes.CDLL(build_c(force=True))
        L = _LIB
        i64p, i32p, f32p = (ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int32),
                            ctypes.POINTER(ctypes.c_float))
        L.oracle_sparse_score.restype = ctypes.c_int64
        L.oracle_sparse_score.argtypes = [i64p, i32p, f32p, i32p, f32p, ctypes.c_int, ctypes.c_float,
                                          ctypes.c_int64, f32p, i64p, f32p, ctypes.c_int]
        L.oracle_select_topk.restype = ctypes.c_int64
        L.oracle_select_topk.argtypes = [i64p, f32p, ctypes.c_int64, ctypes.c_int, i64p, f32p]
        L.oracle_sparse_retrieve.restype = None
        L.oracle_sparse_retrieve.argtypes = [i64p, i32p, f32p, i64p, i32p, f32p, ctypes.c_int64, ctypes.c_int,
                                             ctypes.c_float, ctypes.c_int64, i64p, f32p, i64p,
                                             ctypes.c_int, ctypes.c_int]
        L.oracle_dense_scores_fma.restype = None
        L.oracle_dense_scores_fma.argtypes = [f32p, f32p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, i32p, f32p]
        L.oracle_topk_rows.restype = None
        L.oracle_topk_rows.argtypes = [f32p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, f32p, i64p]
        L.oracle_heap_block.restype = None
        L.oracle_heap_block.argtypes = [f32p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int,
                                        ctypes.c_void_p, i32p, ctypes.c_int]
        L.oracle_heap_finish.restype = None
        L.oracle_heap_finish.argtypes = [ctypes.c_void_p, i32p, ctypes.c_int64, ctypes.c_int, f32p, i64p]
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(ctypes.POINTER(t))


# ----------------------------------------------------------------- sparse (numpy)
def numba_score_float(indptr, doc_ids, vals, indexes_to_retrieve, query_values, threshold, size_collection):
    """indexer.py:324-344, with the posting lists held as CSR (indptr/doc_ids/vals)
    instead of numba.typed.Dict.  Term-serial; `scores[ids] += q * v` is an unfused
    fp32 multiply then add per posting (doc ids are unique within one list)."""
    scores = np.zeros(size_collection, dtype=np.float32)
    for t, q in zip(indexes_to_retrieve, query_values):
        b, e = indptr[t], indptr[t + 1]
        scores[doc_ids[b:e]] += np.float32(q) * vals[b:e]
    filtered = np.argwhere(scores > threshold)[:, 0]
    return filtered, -scores[filtered]


def select_topk(filtered_indexes, neg_scores, k):
    """indexer.py:315-322, canonicalised: the reference keeps the k smallest
    neg_scores in arbitrary (argpartition) order; here the same k by
    (score desc, index asc), sorted.  Equal as a set unless scores tie at the cut."""
    scores = -neg_scores
    order = np.lexsort((filtered_indexes, -scores.astype(np.float64)))[:k]
    return filtered_indexes[order], scores[order]


def sparse_retrieve_c(indptr, doc_ids, vals, q_indptr, q_cols, q_vals, k, threshold, N,
                      q_threads=4, inner_threads=1):
    L = clib()
    indptr = np.ascontiguousarray(indptr, np.int64)
    doc_ids = np.ascontiguousarray(doc_ids, np.int32)
    vals = np.ascontiguousarray(vals, np.float32)
    q_indptr = np.ascontiguousarray(q_indptr, np.int64)
    q_cols = np.ascontiguousarray(q_cols, np.int32)
    q_vals = np.ascontiguousarray(q_vals, np.float32)
    Nq = len(q_indptr) - 1
    oi = np.empty((Nq, k), np.int64)
    os_ = np.empty((Nq, k), np.float32)
    oc = np.empty(Nq, np.int64)
    L.oracle_sparse_retrieve(_p(indptr, ctypes.c_int64), _p(doc_ids, ctypes.c_int32), _p(vals, ctypes.c_float),
                             _p(q_indptr, ctypes.c_int64), _p(q_cols, ctypes.c_int32), _p(q_vals, ctypes.c_float),
                             Nq, k, threshold, N, _p(oi, ctypes.c_int64), _p(os_, ctypes.c_float),
                             _p(oc, ctypes.c_int64), q_threads, inner_threads)
    return oi, os_, oc


# ------------------------------------------------------------------ dense (numpy)
def flat_ip_search(Q, D, k, block=65536):
    """IndexFlatIP.search restated: exact fp32 inner products via BLAS sgemm over
    doc blocks + running top-k, rows sorted by (score desc, index asc); when
    k > N the tail is (-FLT_MAX, -1) as faiss pads."""
    Q = np.ascontiguousarray(Q, np.float32)
    D = np.ascontiguousarray(D, np.float32)
    Nq, N = Q.shape[0], D.shape[0]
    best_s = np.full((Nq, 0), 0, np.float32)
    best_i = np.full((Nq, 0), 0, np.int64)
    for b in range(0, N, block):
        S = Q @ D[b:b + block].T
        idx = np.broadcast_to(np.arange(b, b + S.shape[1], dtype=np.int64), S.shape)
        cs = np.concatenate([best_s, S], axis=1)
        ci = np.concatenate([best_i, idx], axis=1)
        if cs.shape[1] > k:
            part = np.argpartition(-cs, k - 1, axis=1)[:, :k]
            # argpartition is tie-arbitrary at the cut: widen to every tie of the k-th score
            kth = np.take_along_axis(cs, part, 1).min(axis=1, keepdims=True)
            rows_s, rows_i = [], []
            for r in range(Nq):
                keep = np.nonzero(cs[r] >= kth[r])[0]
                o = np.lexsort((ci[r, keep], -cs[r, keep].astype(np.float64)))[:k]
                rows_s.append(cs[r, keep][o])
                rows_i.append(ci[r, keep][o])
            best_s, best_i = np.stack(rows_s), np.stack(rows_i)
        else:
            best_s, best_i = cs, ci
    out_s = np.full((Nq, k), -3.402823466e38, np.float32)
    out_i = np.full((Nq, k), -1, np.int64)
    for r in range(Nq):
        o = np.lexsort((best_i[r], -best_s[r].astype(np.float64)))[:k]
        out_s[r, :len(o)] = best_s[r, o]
        out_i[r, :len(o)] = best_i[r, o]
    return out_s, out_i


def flat_ip_search_blas_heap(Q, D, k, d_block=16384, q_block=4096, stats=None, threads=None):
    """IndexFlatIP.search the way faiss-cpu runs it (knn_inner_product -> exhaustive_inner_product_blas; the reference calls
    it at /root/reference/scaling_retriever/indexer.py:210-214): sgemm over (query block, database block) pairs - the host
    BLAS behind numpy (OpenBLAS), limited to `threads` threads - and one heap per query fed from the block's scores
    (oracle/score_cpu.c, OpenMP over queries).  faiss's own blocks are 4096 queries x 1024 vectors; 16 384 vectors per block
    here keep the threaded sgemm efficient on a many-core host (to the baseline's advantage).  threads=None: the thread count
    with the best sgemm rate on this host among 16 / 32 / 64 / 128 / all (hyper-threaded 256-thread pools run the BLAS at a
    tenth of its 32-thread rate on the GPU boxes).  The cpu_baseline of bench.py.  stats (dict): receives the seconds spent in
    sgemm and in the heaps, the sgemm rate and the thread count."""
    import os
    import time
    from threadpoolctl import threadpool_limits
    L = clib()
    Q = np.ascontiguousarray(Q, np.float32)
    D = np.ascontiguousarray(D, np.float32)
    Nq, N = Q.shape[0], D.shape[0]
    H = Q.shape[1]
    if threads is None:
        cores = os.cpu_count() or 1
        best, threads = 0.0, 1
        pq, pd = Q[:min(Nq, 1024)], D[:min(N, d_block)]
        for nt in sorted({t for t in (16, 32, 64, 128, cores) if t <= cores}):
            with threadpool_limits(limits=nt):
                pq @ pd.T
                t0 = time.perf_counter()
                pq @ pd.T
                rate = 2.0 * pq.shape[0] * pd.shape[0] * H / (time.perf_counter() - t0)
            if rate > best:
                best, threads = rate, nt
    out_s = np.empty((Nq, k), np.float32)
    out_i = np.empty((Nq, k), np.int64)
    t_mm = t_heap = 0.0
    with threadpool_limits(limits=threads):
        for q0 in range(0, Nq, q_block):
            Qb = Q[q0:q0 + q_block]
            nqb = Qb.shape[0]
            heaps = np.zeros((nqb, k, 2), np.int64)          # cand_t = {float, int64}: 16 bytes
            heap_n = np.zeros(nqb, np.int32)
            S = np.empty((nqb, d_block), dtype=np.float32)
            for b in range(0, N, d_block):
                nb = min(d_block, N - b)
                t0 = time.perf_counter()
                if nb == d_block:
                    np.matmul(Qb, D[b:b + nb].T, out=S)
                else:
                    S[:, :nb] = Qb @ D[b:b + nb].T
                t1 = time.perf_counter()
                L.oracle_heap_block(_p(S, ctypes.c_float), nqb, nb, d_block, b, k, ctypes.c_void_p(heaps.ctypes.data),
                                    _p(heap_n, ctypes.c_int32), int(threads))
                t2 = time.perf_counter()
                t_mm += t1 - t0
                t_heap += t2 - t1
            L.oracle_heap_finish(ctypes.c_void_p(heaps.ctypes.data), _p(heap_n, ctypes.c_int32), nqb, k,
                                 _p(out_s[q0:q0 + nqb], ctypes.c_float), _p(out_i[q0:q0 + nqb], ctypes.c_int64))
    if stats is not None:
        stats.update({"sgemm_s": t_mm, "heap_s": t_heap, "sgemm_gflops": 2.0 * Nq * N * H / max(t_mm, 1e-9) / 1e9, "threads": int(threads)})
    return out_s, out_i


def flat_ip_search_fast(Q, D, k, block=32768):
    """Throughput-oriented variant of flat_ip_search for the cpu_baseline timing:
    BLAS sgemm blocks + argpartition (ties at the cut arbitrary, like a heap)."""
    Q = np.ascontiguousarray(Q, np.float32)
    Nq, N = Q.shape[0], D.shape[0]
    best_s = np.empty((Nq, 0), np.float32)
    best_i = np.empty((Nq, 0), np.int64)
    for b in range(0, N, block):
        S = Q @ D[b:b + block].T
        kk = min(k, S.shape[1])
        part = np.argpartition(-S, kk - 1, axis=1)[:, :kk]
        cs = np.concatenate([best_s, np.take_along_axis(S, part, 1)], axis=1)
        ci = np.concatenate([best_i, part.astype(np.int64) + b], axis=1)
        if cs.shape[1] > k:
            p2 = np.argpartition(-cs, k - 1, axis=1)[:, :k]
            cs, ci = np.take_along_axis(cs, p2, 1), np.take_along_axis(ci, p2, 1)
        best_s, best_i = cs, ci
    o = np.argsort(-best_s, axis=1, kind="stable")
    return np.take_along_axis(best_s, o, 1), np.take_along_axis(best_i, o, 1)


def mfma_korder(H):
    """k visiting order of the HIP dense-score kernels' fp32 MFMA chain
    (scaling_retriever_amd/csrc/dense_score.hip): per 8-wide k group s, step j=0..3
    accumulates k = 8s+j (lane half 0) then k = 8s+4+j (lane half 1)."""
    o = []
    for s in range(H // 8):
        for j in range(4):
            o += [8 * s + j, 8 * s + 4 + j]
    return np.array(o, np.int32)


def mfma_korder16(H):
    """k visiting order of the streaming small-batch kernel (csrc/dense_stream.hip, v_mfma_f32_16x16x4_f32):
    per 16-wide k group s, for jj = 0..3, k = 16s + 4g + jj for lane group g = 0..3."""
    o = []
    for s in range(H // 16):
        for jj in range(4):
            o += [16 * s + 4 * g + jj for g in range(4)]
    return np.array(o, np.int32)


def dense_korder(nq, H):
    """Which accumulation order sr_dense_search uses: the streaming kernel serves nq <= 64 when H % 256 == 0."""
    return mfma_korder16(H) if (nq <= 64 and H % 256 == 0) else mfma_korder(H)


def dense_scores_fma(Q, D, korder=None):
    L = clib()
    Q = np.ascontiguousarray(Q, np.float32)
    D = np.ascontiguousarray(D, np.float32)
    Nq, H = Q.shape
    N = D.shape[0]
    out = np.empty((Nq, N), np.float32)
    ko = None if korder is None else np.ascontiguousarray(korder, np.int32)
    L.oracle_dense_scores_fma(_p(Q, ctypes.c_float), _p(D, ctypes.c_float), Nq, N, H,
                              _p(ko, ctypes.c_int32) if ko is not None else None, _p(out, ctypes.c_float))
    return out


def topk_rows(S, k):
    L = clib()
    S = np.ascontiguousarray(S, np.float32)
    Nq, N = S.shape
    os_ = np.empty((Nq, k), np.float32)
    oi = np.empty((Nq, k), np.int64)
    L.oracle_topk_rows(_p(S, ctypes.c_float), Nq, N, k, _p(os_, ctypes.c_float), _p(oi, ctypes.c_int64))
    return os_, oi
