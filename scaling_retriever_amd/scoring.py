"""Thin torch-tensor front ends over the scoring entry points of libsr_hip.so.

These hold device memory (torch tensors) and call the C ABI; all arithmetic is in
the HIP kernels (csrc/dense_score.hip, csrc/sparse_score.hip, csrc/topk.hip).
The reference-shaped classes in indexer.py are built on top of these.
"""
import ctypes

import numpy as np
import torch

from . import _lib


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


class DenseIndexHIP:
    """Flat inner-product index resident in HBM (segments of fp32 [n, dim] rows)."""

    def __init__(self, dim, device=None):
        _lib.require_gpu()
        self.lib = _lib.load()
        self.dim = int(dim)
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self._h = ctypes.c_void_p()
        _lib.check(self.lib.sr_dense_index_create(ctypes.byref(self._h), self.dim), "sr_dense_index_create")
        self._segments = []  # keeps the device tensors alive (the C side holds non-owning views)

    def add_device_rows(self, rows, id_base=None, id_stride=1):
        """rows: fp32 cuda tensor [n, dim] (kept alive by this object, not copied)."""
        if rows.dtype != torch.float32 or rows.dim() != 2 or rows.shape[1] != self.dim:
            raise ValueError(f"expected float32 [n, {self.dim}] rows, got {rows.dtype} {tuple(rows.shape)}")
        if not rows.is_cuda:
            raise ValueError("add_device_rows needs a cuda tensor")
        rows = rows.contiguous()
        if id_base is None:
            id_base = self.ntotal
        if rows.device != self.device:
            raise ValueError(f"rows live on {rows.device}, the index on {self.device}")
        with torch.cuda.device(self.device):
            _lib.check(self.lib.sr_dense_index_add(self._h, _ptr(rows), rows.shape[0], int(id_base), int(id_stride)),
                       "sr_dense_index_add")
        self._segments.append(rows)

    def add_host_rows(self, rows, buffer_size=50000, id_base=None, id_stride=1, piece_bytes=64 << 20, n_buffers=8, n_threads=8):
        """rows: np.float32 [n, dim] - an in-memory array or an np.load(..., mmap_mode="r") view of a shard file.
        Streamed into ONE HBM segment through a ring of pinned staging buffers: worker threads copy pieces of the
        source into pinned memory (numpy releases the GIL for the copy, a memory-mapped source is read from the page
        cache / the file right there) and queue the H2D copy of each piece on a side stream, so file reads, host
        copies and PCIe transfers overlap and no second host copy of the matrix ever exists (the reference
        concatenates every shard on the host, eval_dense.py:113-121, then faiss copies it again, indexer.py:203).
        `buffer_size` (rows per add, indexer.py:198-208) is accepted for signature compatibility."""
        import threading
        from concurrent.futures import ThreadPoolExecutor
        if rows.dtype != np.float32 or rows.ndim != 2 or rows.shape[1] != self.dim:
            if rows.ndim != 2 or rows.shape[1] != self.dim:
                raise ValueError(f"expected [n, {self.dim}] rows, got {rows.shape}")
            rows = np.asarray(rows, dtype=np.float32)
        n = rows.shape[0]
        dev = torch.empty((n, self.dim), dtype=torch.float32, device=self.device)
        if n == 0:
            self.add_device_rows(dev, id_base=id_base, id_stride=id_stride)
            return
        piece_rows = max(1, int(piece_bytes) // (4 * self.dim))
        pieces = [(r0, min(n, r0 + piece_rows)) for r0 in range(0, n, piece_rows)]
        n_buffers = max(1, min(n_buffers, len(pieces)))
        with torch.cuda.device(self.device):
            side = torch.cuda.Stream()
            # `dev` came from the caching allocator on the current stream: a recycled block may still have kernels queued there
            side.wait_stream(torch.cuda.current_stream(self.device))
            bufs = [torch.empty((piece_rows, self.dim), dtype=torch.float32, pin_memory=True) for _ in range(n_buffers)]
            free = [torch.cuda.Event() for _ in range(n_buffers)]
            locks = [threading.Lock() for _ in range(n_buffers)]

            def move(i):
                r0, r1 = pieces[i]
                b = i % n_buffers
                with locks[b]:                          # pieces i, i + n_buffers, ... share buffer b, in order
                    free[b].synchronize()               # its previous H2D copy has left the buffer
                    np.copyto(bufs[b].numpy()[:r1 - r0], rows[r0:r1])
                    with torch.cuda.device(self.device), torch.cuda.stream(side):
                        dev[r0:r1].copy_(bufs[b][:r1 - r0], non_blocking=True)
                        free[b].record(side)
            with ThreadPoolExecutor(max_workers=max(1, min(n_threads, n_buffers))) as pool:
                list(pool.map(move, range(len(pieces))))
            side.synchronize()
        self.add_device_rows(dev, id_base=id_base, id_stride=id_stride)

    def add_npy_file(self, path, id_base=None, id_stride=1, piece_bytes=64 << 20, n_buffers=8, n_threads=8):
        """One embs_{rank}_{chunk}.npy shard file -> one HBM segment, never loaded whole on the host: worker threads
        `preadv` pieces of the file straight into a ring of pinned staging buffers (one kernel copy out of the page cache,
        no per-page mapping faults as a memory-mapped source costs, GIL released) and queue each piece's H2D copy on a side
        stream."""
        import os
        import threading
        from concurrent.futures import ThreadPoolExecutor
        arr = np.load(path, mmap_mode="r")              # header only: shape, dtype, data offset
        if arr.dtype != np.float32 or arr.ndim != 2 or arr.shape[1] != self.dim or not arr.flags["C_CONTIGUOUS"]:
            return self.add_host_rows(np.ascontiguousarray(arr, dtype=np.float32), id_base=id_base, id_stride=id_stride)
        n, offset0 = arr.shape[0], arr.offset
        del arr
        dev = torch.empty((n, self.dim), dtype=torch.float32, device=self.device)
        if n == 0:
            return self.add_device_rows(dev, id_base=id_base, id_stride=id_stride)
        row_bytes = 4 * self.dim
        piece_rows = max(1, int(piece_bytes) // row_bytes)
        pieces = [(r0, min(n, r0 + piece_rows)) for r0 in range(0, n, piece_rows)]
        n_buffers = max(1, min(n_buffers, len(pieces)))
        fd = os.open(path, os.O_RDONLY)
        try:
            with torch.cuda.device(self.device):
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream(self.device))      # see add_host_rows
                bufs = [torch.empty((piece_rows, self.dim), dtype=torch.float32, pin_memory=True) for _ in range(n_buffers)]
                free = [torch.cuda.Event() for _ in range(n_buffers)]
                locks = [threading.Lock() for _ in range(n_buffers)]

                def move(i):
                    r0, r1 = pieces[i]
                    b = i % n_buffers
                    with locks[b]:
                        free[b].synchronize()
                        view = memoryview(bufs[b].numpy()[:r1 - r0]).cast("B")
                        got, want = 0, (r1 - r0) * row_bytes
                        while got < want:
                            k = os.preadv(fd, [view[got:want]], offset0 + r0 * row_bytes + got)
                            if k <= 0:
                                raise IOError(f"{path}: short read at row {r0}")
                            got += k
                        with torch.cuda.device(self.device), torch.cuda.stream(side):
                            dev[r0:r1].copy_(bufs[b][:r1 - r0], non_blocking=True)
                            free[b].record(side)
                with ThreadPoolExecutor(max_workers=max(1, min(n_threads, n_buffers))) as pool:
                    list(pool.map(move, range(len(pieces))))
                side.synchronize()
        finally:
            os.close(fd)
        self.add_device_rows(dev, id_base=id_base, id_stride=id_stride)

    @property
    def ntotal(self):
        return int(self.lib.sr_dense_index_ntotal(self._h))

    def set_workspace_limit(self, nbytes):
        _lib.check(self.lib.sr_dense_index_set_workspace_limit(self._h, int(nbytes)))

    def set_batch_invariant(self, on=True):
        """One k order for every batch size: a query's results are the same bits alone and inside any batch (batches of <= 64
        queries then run the tiled kernels instead of the streaming one, ~4.4 instead of ~5.9 TB/s).  Off by default."""
        _lib.check(self.lib.sr_dense_index_set_batch_invariant(self._h, 1 if on else 0))

    def set_precision(self, mode):
        """"fp32" (default: the exact kernel), "fp32_filtered" (the same results bit for bit through a certified fp16 filter +
        exact re-score, ~7x faster for batches > 64 queries, one fp16 plane of the corpus in HBM), "bf16x3" / "bf16x6"
        (split-bf16 scores, not bit-identical)."""
        code = {"fp32": 0, "bf16x3": 1, "bf16x6": 2, "fp32_filtered": 3}[mode]
        with torch.cuda.device(self.device):
            _lib.check(self.lib.sr_dense_index_set_precision(self._h, code), "sr_dense_index_set_precision")

    def filter_stats(self):
        """(searches answered by the certified filter alone, searches where some or all queries were re-done by the exact kernel)."""
        a, b = ctypes.c_int64(0), ctypes.c_int64(0)
        _lib.check(self.lib.sr_dense_index_filter_stats(self._h, ctypes.byref(a), ctypes.byref(b)))
        return a.value, b.value

    def filter_query_stats(self):
        """(queries certified by the filter, queries re-done by the exact kernel) so far."""
        a, b = ctypes.c_int64(0), ctypes.c_int64(0)
        _lib.check(self.lib.sr_dense_index_filter_query_stats(self._h, ctypes.byref(a), ctypes.byref(b)))
        return a.value, b.value

    def search(self, queries, k):
        """queries: fp32 cuda tensor [nq, dim] -> (scores fp32 [nq,k], ids int64 [nq,k]) cuda tensors."""
        if queries.dtype != torch.float32 or queries.dim() != 2 or queries.shape[1] != self.dim:
            raise ValueError(f"expected float32 [nq, {self.dim}] queries, got {queries.dtype} {tuple(queries.shape)}")
        if not queries.is_cuda:
            raise ValueError("queries must be a cuda tensor")
        queries = queries.contiguous()
        nq = queries.shape[0]
        scores = torch.empty((nq, k), dtype=torch.float32, device=queries.device)
        ids = torch.empty((nq, k), dtype=torch.int64, device=queries.device)
        if queries.device != self.device:
            raise ValueError(f"queries live on {queries.device}, the index on {self.device}")
        with torch.cuda.device(self.device):      # the library allocates its workspace on the current device
            _lib.check(self.lib.sr_dense_search(self._h, _ptr(queries), nq, int(k), _ptr(scores), _ptr(ids),
                                                _lib.stream_ptr()), "sr_dense_search")
        return scores, ids

    def search_begin(self, queries, k, share):
        """First half of a doc-sharded search (include/sr_hip.h sr_dense_search_begin): returns lower [nq] fp32 (cuda) - per
        query a value at least ceil(k / share) documents of this index reach exactly (-inf where the filter does not apply).
        `queries` must be the same contiguous tensor that is passed to search_finish."""
        if queries.dtype != torch.float32 or queries.dim() != 2 or queries.shape[1] != self.dim or not queries.is_cuda or not queries.is_contiguous():
            raise ValueError(f"expected a contiguous float32 cuda tensor [nq, {self.dim}], got {queries.dtype} {tuple(queries.shape)}")
        lower = torch.empty((queries.shape[0],), dtype=torch.float32, device=queries.device)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.sr_dense_search_begin(self._h, _ptr(queries), queries.shape[0], int(k), int(share), _ptr(lower),
                                                      _lib.stream_ptr()), "sr_dense_search_begin")
        return lower

    def search_finish(self, queries, k, threshold=None):
        """Second half: threshold [nq] fp32 = the minimum of search_begin's values over all shards (None: a plain search).
        Returns (scores [nq, k], ids [nq, k]); rows may end in padding (id -1) - the shard returns what can reach the global top-k."""
        nq = queries.shape[0]
        scores = torch.empty((nq, k), dtype=torch.float32, device=queries.device)
        ids = torch.empty((nq, k), dtype=torch.int64, device=queries.device)
        if threshold is not None:
            threshold = threshold.to(device=queries.device, dtype=torch.float32).contiguous()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.sr_dense_search_finish(self._h, _ptr(queries), nq, int(k), _ptr(threshold), _ptr(scores), _ptr(ids),
                                                       _lib.stream_ptr()), "sr_dense_search_finish")
        return scores, ids

    def close(self):
        if self._h:
            self.lib.sr_dense_index_destroy(self._h)
            self._h = ctypes.c_void_p()
        self._segments = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class SparseIndexHIP:
    """CSR-by-term inverted index resident in HBM.

    indptr int64 [V+1], doc_ids int32 [nnz] (strictly ascending inside each term),
    vals fp32 [nnz]; n_docs = IndexDictOfArray.nb_docs().
    """

    def __init__(self, indptr, doc_ids, vals, n_docs, device=None):
        _lib.require_gpu()
        self.lib = _lib.load()
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)

        def to_dev(x, dt):
            if isinstance(x, np.ndarray):
                x = torch.from_numpy(np.ascontiguousarray(x))
            return x.to(device=self.device, dtype=dt).contiguous()
        self.indptr = to_dev(indptr, torch.int64)
        self.doc_ids = to_dev(doc_ids, torch.int32)
        self.vals = to_dev(vals, torch.float32)
        if self.doc_ids.numel() == 0:  # keep valid pointers
            self.doc_ids = torch.zeros(1, dtype=torch.int32, device=self.device)
            self.vals = torch.zeros(1, dtype=torch.float32, device=self.device)
        self.n_terms = self.indptr.numel() - 1
        self.n_docs = int(n_docs)
        self._h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.sr_sparse_index_create(ctypes.byref(self._h), _ptr(self.indptr), _ptr(self.doc_ids),
                                                       _ptr(self.vals), self.n_terms, self.n_docs, _lib.stream_ptr()),
                       "sr_sparse_index_create")

    def set_workspace_limit(self, nbytes):
        _lib.check(self.lib.sr_sparse_index_set_workspace_limit(self._h, int(nbytes)))

    def block_stats(self):
        """{"dense_terms", "block_calls", "fallback_calls"}: which of the two bit-identical scoring paths ran."""
        a, b, c = ctypes.c_int64(0), ctypes.c_int64(0), ctypes.c_int64(0)
        _lib.check(self.lib.sr_sparse_index_block_stats(self._h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)))
        return {"dense_terms": a.value, "block_calls": b.value, "fallback_calls": c.value}

    def cert_stats(self):
        """Certified two-stage scorer (csrc/sparse_cert.hip): {"present", "dense_terms", "searches", "queries", "redone_exact", "doc_tiles",
        "candidates_rescored" (exact chains run by the certified path, rounded up to 16 per query), "batches_without_memory" (query batches
        the exact kernels served because the scorer's per-call buffers did not fit)}."""
        out = (ctypes.c_int64 * 8)()
        _lib.check(self.lib.sr_sparse_index_cert_stats(self._h, out), "sr_sparse_index_cert_stats")
        keys = ("present", "dense_terms", "searches", "queries", "redone_exact", "doc_tiles", "candidates_rescored", "batches_without_memory")
        return {k_: int(v) for k_, v in zip(keys, out)}

    def cert_record_keys(self, enable):
        """Test hook: keep the stage-1 keys of every (query, doc) pair of the following searches."""
        with torch.cuda.device(self.device):
            _lib.check(self.lib.sr_sparse_index_cert_debug(self._h, 1 if enable else 0, None, 0, None, 0, None, None),
                       "sr_sparse_index_cert_debug")

    def cert_recorded_keys(self, nq):
        """(keys uint16 [nq_pad, n_tiles * 1024], consts fp32 [nq_pad, 6] = (c_q, s_q, rare terms, query terms, rare terms left out, the k-th best
        key the last top-k select of the scan saw: 0 if none ran), vscale, T) of the last search (after cert_record_keys(True))."""
        nq_pad = (nq + 31) // 32 * 32
        stride = self.cert_stats()["doc_tiles"] * 1024
        keys = np.empty((nq_pad, stride), dtype=np.uint16)
        consts = np.empty((nq_pad, 6), dtype=np.float32)
        vs, T = ctypes.c_float(0), ctypes.c_int32(0)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.sr_sparse_index_cert_debug(self._h, 2, keys.ctypes.data_as(ctypes.c_void_p), keys.size,
                                                           consts.ctypes.data_as(ctypes.c_void_p), nq_pad, ctypes.byref(vs), ctypes.byref(T)),
                       "sr_sparse_index_cert_debug")
        return keys, consts, vs.value, T.value

    def work_counters(self, enable):
        """Switch the query-block kernel's work counters on / off; returns what was counted since the last call:
        {"dense_columns_loaded", "dense_column_applications", "light_postings", "grouped_postings", "plan_entries", "workgroup_tiles"}."""
        out = (ctypes.c_uint64 * 6)()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.sr_sparse_index_work_counters(self._h, 1 if enable else 0, out), "sr_sparse_index_work_counters")
        keys = ("dense_columns_loaded", "dense_column_applications", "light_postings", "grouped_postings", "plan_entries", "workgroup_tiles")
        return {k_: int(v) for k_, v in zip(keys, out)}

    def search(self, q_indptr, q_cols, q_vals, k, threshold=0.0, id_base=0, id_stride=1):
        """Queries as CSR tensors. Returns (scores [nq,k], ids [nq,k], counts [nq]) cuda tensors."""
        def to_dev(x, dt):
            if isinstance(x, np.ndarray):
                x = torch.from_numpy(np.ascontiguousarray(x))
            return x.to(device=self.device, dtype=dt).contiguous()
        q_indptr = to_dev(q_indptr, torch.int64)
        q_cols = to_dev(q_cols, torch.int32)
        q_vals = to_dev(q_vals, torch.float32)
        nq = q_indptr.numel() - 1
        if q_cols.numel() == 0:
            q_cols = torch.zeros(1, dtype=torch.int32, device=self.device)
            q_vals = torch.zeros(1, dtype=torch.float32, device=self.device)
        scores = torch.empty((nq, k), dtype=torch.float32, device=self.device)
        ids = torch.empty((nq, k), dtype=torch.int64, device=self.device)
        counts = torch.empty((nq,), dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.sr_sparse_search(self._h, _ptr(q_indptr), _ptr(q_cols), _ptr(q_vals), nq, int(k),
                                                 float(threshold), int(id_base), int(id_stride), _ptr(scores), _ptr(ids),
                                                 _ptr(counts), _lib.stream_ptr()), "sr_sparse_search")
        return scores, ids, counts

    def close(self):
        if self._h:
            self.lib.sr_sparse_index_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def sparse_csr_build(rows, cols, vals, n_terms, n_docs=0, sort_docs=False):
    """Doc-major postings (cuda tensors: rows int32 = global doc row, cols int32 = term, vals fp32; insertion order) -> CSR by term
    (indptr int64 [n_terms + 1], doc_ids int32, vals fp32) with sr_sparse_csr_build: this library's stable radix sort on the device.
    Replaces the per-posting append of IndexDictOfArray.add_batch_document (inverted_index.py:67-76)."""
    _lib.require_gpu()
    lib = _lib.load()
    rows = rows.to(torch.int32).contiguous()
    cols = cols.to(torch.int32).contiguous()
    vals = vals.to(torch.float32).contiguous()
    nnz = rows.numel()
    dev = rows.device
    indptr = torch.empty(int(n_terms) + 1, dtype=torch.int64, device=dev)
    out_rows = torch.empty(max(1, nnz), dtype=torch.int32, device=dev)
    out_vals = torch.empty(max(1, nnz), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.sr_sparse_csr_build(_ptr(rows), _ptr(cols), _ptr(vals), nnz, int(n_terms), int(n_docs), 1 if sort_docs else 0,
                                           _ptr(indptr), _ptr(out_rows), _ptr(out_vals), _lib.stream_ptr()), "sr_sparse_csr_build")
    return indptr, out_rows[:nnz], out_vals[:nnz]


def sparse_csr_expand_terms(indptr, nnz):
    """Term of every posting of a CSR-by-term index (indptr int64 [V + 1] on the device) -> int32 [nnz]: sr_sparse_csr_expand_terms."""
    _lib.require_gpu()
    lib = _lib.load()
    indptr = indptr.to(torch.int64).contiguous()
    out = torch.empty(max(1, int(nnz)), dtype=torch.int32, device=indptr.device)
    with torch.cuda.device(indptr.device):
        _lib.check(lib.sr_sparse_csr_expand_terms(_ptr(indptr), indptr.numel() - 1, int(nnz), _ptr(out), _lib.stream_ptr()),
                   "sr_sparse_csr_expand_terms")
    return out[:int(nnz)]


def topk_merge(scores, ids, pad_score=-3.402823466e38):
    """Merge per-shard top-k lists: scores fp32 [W, nq, k], ids int64 [W, nq, k] (cuda) -> ([nq,k], [nq,k])."""
    _lib.require_gpu()
    lib = _lib.load()
    if scores.dim() != 3 or ids.shape != scores.shape:
        raise ValueError("expected scores/ids of shape [n_lists, nq, k]")
    scores = scores.contiguous().float()
    ids = ids.contiguous().to(torch.int64)
    W, nq, k = scores.shape
    out_s = torch.empty((nq, k), dtype=torch.float32, device=scores.device)
    out_i = torch.empty((nq, k), dtype=torch.int64, device=scores.device)
    _lib.check(lib.sr_topk_merge(_ptr(scores), _ptr(ids), W, nq, k, float(pad_score), _ptr(out_s), _ptr(out_i),
                                 _lib.stream_ptr()), "sr_topk_merge")
    return out_s, out_i
