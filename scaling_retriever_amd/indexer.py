"""Retrieval / indexing layer: host-side mirror of
/root/reference/scaling_retriever/indexer.py for the eval path -
store_embs (:26-97), DenseIndexer / DenseFlatIndexer (:127-217), SparseIndexer (:220-308),
SparseRetrieval (:311-540) - with every arithmetic step on the MI355X:

  reference (CPU)                                   here (HIP, via libsr_hip.so)
  faiss.IndexFlatIP.add / .search                   DenseIndexHIP: D resident in HBM, fp32 MFMA + fused top-k
  numba_score_float + select_topk, 4 threads        SparseIndexHIP: LDS-tiled posting scan + fused top-k, batched
  torch.nonzero + python per-posting append         sr_sparse_compact (device CSR) + one sort by term
  model.encode under torch.autocast(bf16)           LlamaBiDense/LlamaBiSparse.encode (HIP Llama forward)

Same class / method names, arguments, artefact files (embs_{rank}_{chunk}.npy, ids_*.npy,
plan.json, doc_ids.pkl, index_dist.json, index_stats.json, run.json, q_stats.json).
"""
import ctypes
import json
import logging
import os
import pickle
from collections import defaultdict
from typing import List

import numpy as np
import torch
from tqdm import tqdm

from . import _lib
from .scoring import DenseIndexHIP, SparseIndexHIP, sparse_csr_build, sparse_csr_expand_terms
from .utils.inverted_index import IndexDictOfArray
from .utils.run_file import IdTable, PiecewiseRunWriter, RunResult, id_table, to_host, write_run_json
from .utils.utils import get_rank, get_world_size, is_first_worker, to_list

logger = logging.getLogger()


def _unwrap(model):
    return model.module if hasattr(model, "module") else model


def batch_groups(loader, max_rows=16384):
    """Consecutive collator batches of `loader`, grouped until `max_rows` rows: one LlamaBi*.encode_batches call per group.
    The reference's query loaders yield eval_batch_size (128) rows at a time (eval_dense.py:219-220, eval_sparse.py:125-127);
    the engine wants tens of thousands of tokens per pass."""
    group, rows = [], 0
    for batch in loader:
        n = len(batch["ids"])
        if group and rows + n > max_rows:
            yield group
            group, rows = [], 0
        group.append(batch)
        rows += n
    if group:
        yield group


def encode_group(model, group, device, which="encode"):
    """Encode one group of collator batches: encode_batches when the model has it (bit-identical rows, one engine pass),
    else batch by batch like the reference."""
    inputs = [{k: v.to(device) for k, v in batch.items() if k != "ids"} for batch in group]
    if len(inputs) > 1 and hasattr(model, "encode_batches"):
        return model.encode_batches(inputs)
    outs = [getattr(model, which)(**i) for i in inputs]
    if isinstance(outs[0], tuple):
        return tuple(torch.cat(o) for o in zip(*outs))
    return torch.cat(outs) if len(outs) > 1 else outs[0]


STORE_GROUP_ROWS = 512          # store_embs: fixed-size loader batches encoded per engine pass (4 batches of 128)


# =============================================================== dense: corpus encode
def store_embs(model, collection_loader, local_rank, index_dir, device, chunk_size=2_000_000, use_fp16=False,
               is_query=False, idx_to_id=None):
    """indexer.py:26-97.  Encodes this rank's shard and writes embs_{rank}_{chunk}.npy (fp32 [n, H]),
    ids_{rank}_{chunk}.npy and (rank 0) plan.json.  Embeddings stay on the device until a chunk is full,
    so there is one D2H copy per chunk instead of one per batch (indexer.py:56)."""
    if is_query:
        raise NotImplementedError
    batch_size = getattr(collection_loader, "batch_size", None)
    if is_first_worker():
        print("batch_size: {}, chunk_size: {}".format(batch_size if batch_size else "token budget", chunk_size))
    os.makedirs(index_dir, exist_ok=True)
    enc = _unwrap(model)
    embeddings, embeddings_ids = [], []
    chunk_idx = 0
    # a chunk is closed after `chunk_size` passages; with fixed-size batches that is every chunk_size // batch_size
    # batches, exactly the reference's write_freq (indexer.py:32-33, :60)
    chunk_docs = max(1, chunk_size // batch_size) * batch_size if batch_size else chunk_size

    # A full chunk leaves the device on a side stream and is written by ONE background thread while the next chunk is being
    # encoded: at MS MARCO scale a chunk is 2 M x H fp32 = 16 GB - a third of a second of PCIe and several seconds of np.save
    # during which the GPU would otherwise idle.  One chunk in flight; files appear in chunk order.
    from concurrent.futures import ThreadPoolExecutor
    writer = ThreadPoolExecutor(max_workers=1)
    pending = []
    on_gpu = torch.cuda.is_available() and torch.device(device).type == "cuda"
    side = torch.cuda.Stream(device=device) if on_gpu else None

    def flush():
        nonlocal embeddings, embeddings_ids, chunk_idx
        embs_dev = torch.cat(embeddings).float()
        ids = embeddings_ids
        if isinstance(ids[0], int):
            ids = np.array(ids, dtype=np.int64)
        assert len(embs_dev) == len(ids), (len(embs_dev), len(ids))
        while pending:                          # at most one chunk in flight (host and device memory stay bounded)
            pending.pop().result()
        if on_gpu:
            host = torch.empty(embs_dev.shape, dtype=torch.float32, pin_memory=True)
            side.wait_stream(torch.cuda.current_stream(embs_dev.device))
            with torch.cuda.stream(side):
                host.copy_(embs_dev, non_blocking=True)
                done = torch.cuda.Event()
                done.record(side)
            embs_dev.record_stream(side)
        else:
            host, done = embs_dev, None

        def write(host=host, ids=ids, done=done, ci=chunk_idx):
            if done is not None:
                done.synchronize()
            np.save(os.path.join(index_dir, "embs_{}_{}.npy".format(local_rank, ci)), host.numpy())
            np.save(os.path.join(index_dir, "ids_{}_{}.npy".format(local_rank, ci)), ids)
        pending.append(writer.submit(write))
        embeddings, embeddings_ids = [], []
        chunk_idx += 1

    try:
        total = len(collection_loader)
    except TypeError:
        total = None              # token-budget loaders do not know their batch count up front
    # The reference's loader yields fixed batches of per_device_eval_batch_size passages (128: ~9 600 real tokens), less than the
    # engine wants per pass.  Fixed-size batches are therefore encoded a few at a time in ONE engine pass (encode_batches: every row
    # keeps the positions of its own batch, so the vectors are those of batch-by-batch doc_encode calls, bit for bit); a
    # token-budget loader's batches are already sized for the engine and go one by one.  Chunk files keep the reference's row
    # counts: a group's rows are appended batch by batch, with the flush check in between.
    group_rows = STORE_GROUP_ROWS if (batch_size and hasattr(enc, "encode_batches")) else 0
    progress = tqdm(disable=not is_first_worker(), desc="encode # {} seqs".format(total if total is not None else "?"), total=total)
    for group in (batch_groups(collection_loader, group_rows) if group_rows else ([b] for b in collection_loader)):
        with torch.inference_mode(), torch.autocast("cuda", dtype=torch.bfloat16):      # indexer.py:46-52
            if len(group) > 1:
                reps_g = encode_group(enc, group, device, which="doc_encode")
            else:
                inputs = {k: v.to(device, non_blocking=True) for k, v in group[0].items() if k != "ids"}
                reps_g = enc.doc_encode(**inputs)
        r0 = 0
        for batch in group:
            text_ids = batch["ids"]
            assert isinstance(text_ids, list)
            embeddings.append(reps_g[r0:r0 + len(text_ids)])
            r0 += len(text_ids)
            embeddings_ids.extend(text_ids)
            progress.update(1)
            if len(embeddings_ids) >= chunk_docs:
                flush()
        assert r0 == len(reps_g), (r0, len(reps_g))
    progress.close()
    if len(embeddings) != 0:
        print("last embedddings shape = {}".format((sum(len(e) for e in embeddings), embeddings[0].shape[1])))
        flush()
    while pending:
        pending.pop().result()                  # the last chunk is on disk (an exception of the writer surfaces here)
    writer.shutdown()

    plan = {"nranks": get_world_size(), "num_chunks": chunk_idx, "index_path": os.path.join(index_dir, "model.index")}
    print("plan: ", plan)
    if is_first_worker():
        with open(os.path.join(index_dir, "plan.json"), "w") as fout:
            json.dump(plan, fout)


# ================================================================ dense: flat index
class DenseIndexer(object):
    """indexer.py:127-188."""

    def __init__(self, buffer_size: int = 50000):
        self.buffer_size = buffer_size
        self.index_id_to_db_id = []
        self.index = None

    def init_index(self, vector_sz: int):
        raise NotImplementedError

    def index_data(self, doc_reps, doc_ids):
        raise NotImplementedError

    def get_index_name(self):
        raise NotImplementedError

    def search_knn(self, query_vectors, top_docs: int):
        raise NotImplementedError

    def get_files(self, path: str):
        if os.path.isdir(path):
            return os.path.join(path, "index.dpr"), os.path.join(path, "index_meta.dpr")
        return path + ".{}.dpr".format(self.get_index_name()), path + ".{}_meta.dpr".format(self.get_index_name())

    def index_exists(self, path: str):
        index_file, meta_file = self.get_files(path)
        return os.path.isfile(index_file) and os.path.isfile(meta_file)

    def _update_id_mapping(self, db_ids: List):
        self.index_id_to_db_id.extend(db_ids)
        self._id_table = self._run_table = None
        return len(self.index_id_to_db_id)

    def id_table(self):
        """index_id_to_db_id as one object array (+ a trailing None for faiss' label -1), rebuilt when the list grew: the id
        mapping of search_knn is ONE numpy take instead of a Python loop over every hit (indexer.py:212-213)."""
        ids = self.index_id_to_db_id
        n = len(ids)
        # the reference lets callers assign or edit index_id_to_db_id directly (deserialize, :189): the cache is tied to the list object,
        # its length and its first / last entries, so a re-assigned or re-loaded list of the same length is not served stale ids
        stamp = (id(ids), n, ids[0] if n else None, ids[-1] if n else None)
        cached = getattr(self, "_id_table", None)
        if cached is None or getattr(self, "_id_table_stamp", None) != stamp:
            cached = np.empty(n + 1, dtype=object)
            cached[:n] = ids
            cached[n] = None
            self._id_table = cached
            self._id_table_stamp = stamp
            self._run_table = None
        return cached

    def run_table(self):
        """The same ids as an IdTable (typed keys of run.json)."""
        self.id_table()
        if getattr(self, "_run_table", None) is None:
            self._run_table = IdTable(self.index_id_to_db_id)
        return self._run_table


class DenseFlatIndexer(DenseIndexer):
    """indexer.py:191-217 over a flat inner-product index resident in HBM."""

    def __init__(self, buffer_size: int = 50000):
        super().__init__(buffer_size)
        self.hidden_dim = None

    def init_index(self, hidden_dim):
        self.hidden_dim = int(hidden_dim)
        self.index = DenseIndexHIP(self.hidden_dim)
        # IndexFlatIP's exact results (bit-identical to the exact fp32 kernel) through the certified bf16 filter + exact
        # re-score; the library uses the exact kernel by itself when HBM has no room for the filter's bf16 planes
        if self.hidden_dim % 64 == 0:
            self.index.set_precision("fp32_filtered")

    def index_data(self, doc_reps, doc_ids):
        assert len(doc_reps) == len(doc_ids)
        n = len(doc_reps)
        if isinstance(doc_reps, torch.Tensor) and doc_reps.is_cuda:
            self.index.add_device_rows(doc_reps.float())
        else:
            self.index.add_host_rows(doc_reps if isinstance(doc_reps, np.ndarray) else np.asarray(doc_reps), buffer_size=self.buffer_size)
        n_total = self._update_id_mapping(list(doc_ids))
        logger.info("total data indexed %d", n_total)
        assert self.index.ntotal == n_total, (self.index.ntotal, n_total)
        return n

    def id_lists(self, positions):
        """[[db id of p for p in row] for row in positions] (indexer.py:212-213) for a C-contiguous int64 [nq, k] array of index
        positions; -1 -> None.  csrc/host_lists.c visits the id objects in index order instead of hit order (7 M references over a
        500 MB object heap: a TLB miss each in hit order); the lists hold the index's own id objects, as the reference's do."""
        from . import _host_lists
        table = self.id_table()
        positions = np.ascontiguousarray(positions, dtype=np.int64)
        return _host_lists.take_rows(table.ctypes.data, len(table) - 1, positions.ctypes.data, positions.shape[0], positions.shape[1])

    KNN_CHUNKS = 2          # search_knn pipelines the query set in this many pieces when it is large (>= 1 024 queries)

    def search_knn(self, query_reps, top_docs: int):
        """indexer.py:210-214: (list of db-id lists, fp32 scores [nq, k]); label -1 (fewer than k vectors) -> None.  The lists come
        from id_lists (csrc/host_lists.c).  A large query set is searched in KNN_CHUNKS pieces, the GPU working on piece c + 1 (in a
        worker thread: the C call releases the GIL) while this thread builds the lists of piece c - the exact results do not depend
        on how the queries are batched (pieces stay above 64 queries: one kernel family, one k order)."""
        if isinstance(query_reps, torch.Tensor):
            q = query_reps.to(device=self.index.device, dtype=torch.float32)
        else:
            q = torch.from_numpy(np.ascontiguousarray(query_reps, dtype=np.float32)).to(self.index.device)
        nq = q.shape[0]
        n_chunks = self.KNN_CHUNKS if nq >= 1024 else 1
        if n_chunks == 1:
            scores, indexes = self.search_arrays(q, top_docs)
            return self.id_lists(indexes), scores
        from concurrent.futures import ThreadPoolExecutor
        per = (nq + n_chunks - 1) // n_chunks
        bounds = [(c0, min(nq, c0 + per)) for c0 in range(0, nq, per)]
        dev = self.index.device
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(dev))          # q may come from a side stream of the caller

        def gpu(c):
            with torch.cuda.device(dev):
                torch.cuda.current_stream(dev).wait_event(ready)          # the worker thread's current stream is the default one
                return self.search_arrays(q[bounds[c][0]:bounds[c][1]], top_docs)
        top_doc_ids, score_parts = [], []
        with ThreadPoolExecutor(max_workers=1) as pool:
            fut = pool.submit(gpu, 0)
            for c in range(len(bounds)):
                scores, indexes = fut.result()
                if c + 1 < len(bounds):
                    fut = pool.submit(gpu, c + 1)
                top_doc_ids.extend(self.id_lists(indexes))
                score_parts.append(scores)
        return top_doc_ids, np.concatenate(score_parts)

    def search_arrays(self, query_reps, top_docs: int):
        """(scores fp32 [nq, k], index positions int64 [nq, k]; -1 = fewer than k vectors) as host arrays: what search_knn maps
        to db ids, and what the run.json writer takes as they are (utils/run_file.py)."""
        if isinstance(query_reps, torch.Tensor):
            q = query_reps.to(device=self.index.device, dtype=torch.float32)
        else:
            q = torch.from_numpy(np.ascontiguousarray(query_reps, dtype=np.float32)).to(self.index.device)
        scores, indexes = self.index.search(q, top_docs)
        return to_host(scores), to_host(indexes)

    def get_index_name(self):
        return "flat_index"

    def serialize(self, file: str):
        """indexer.py:145-159 writes a faiss file; here the same two files hold the raw fp32 rows (npy) +
        the pickled id map (faiss' binary format is not reproduced)."""
        if os.path.isdir(file):
            index_file, meta_file = os.path.join(file, "index.dpr"), os.path.join(file, "index_meta.dpr")
        else:
            index_file, meta_file = file + ".index.dpr", file + ".index_meta.dpr"
        rows = torch.cat([s for s in self.index._segments]).cpu().numpy()
        with open(index_file, "wb") as f:
            np.save(f, rows)
        with open(meta_file, mode="wb") as f:
            pickle.dump(self.index_id_to_db_id, f)

    def deserialize(self, path: str):
        index_file, meta_file = self.get_files(path)
        if not os.path.isfile(index_file):   # the reference's serialize() naming (".index.dpr")
            index_file, meta_file = path + ".index.dpr", path + ".index_meta.dpr"
        with open(index_file, "rb") as f:
            rows = np.load(f)
        self.init_index(rows.shape[1])
        self.index.add_host_rows(rows, buffer_size=self.buffer_size)
        with open(meta_file, "rb") as reader:
            self.index_id_to_db_id = pickle.load(reader)
        assert len(self.index_id_to_db_id) == self.index.ntotal, \
            "Deserialized index_id_to_db_id should match faiss index size"


# ======================================================================= sparse reps
def sparse_reps_to_csr(reps):
    """[B, V] fp32 cuda tensor -> (row_ptr int64 [B+1], cols int32 [nnz], vals fp32 [nnz]) cuda tensors in
    torch.nonzero order (indexer.py:259-260, :393-399), via sr_sparse_compact."""
    lib = _lib.load()
    reps = reps.contiguous().float()
    B, V = reps.shape
    row_ptr = torch.empty(B + 1, dtype=torch.int64, device=reps.device)
    cap = max(1, min(B * V, 1 << 22))
    n = ctypes.c_int64(0)
    with torch.cuda.device(reps.device):
        while True:
            cols = torch.empty(cap, dtype=torch.int32, device=reps.device)
            vals = torch.empty(cap, dtype=torch.float32, device=reps.device)
            rc = lib.sr_sparse_compact(reps.data_ptr(), B, V, row_ptr.data_ptr(), cols.data_ptr(), vals.data_ptr(), cap,
                                       ctypes.byref(n), _lib.stream_ptr())
            if rc == _lib.SR_ERR_NOMEM and n.value > cap:
                cap = n.value
                continue
            _lib.check(rc, "sr_sparse_compact")
            break
    return row_ptr, cols[:n.value], vals[:n.value]


class L0:
    """losses/regulariaztion.py:9-14 (index statistics only)."""

    def __call__(self, batch_rep):
        return torch.count_nonzero(batch_rep, dim=-1).float().mean()


# ===================================================================== sparse: indexing
class SparseIndexer:
    """indexer.py:220-308.  The per-batch nonzero extraction runs on the device; postings are appended
    as arrays (no per-posting Python loop)."""

    def __init__(self, model, index_dir, device, compute_stats=False, dim_voc=None, force_new=True,
                 filename="array_index.h5py", **kwargs):
        self.model = model
        self.model.eval()
        self.index_dir = index_dir
        self.sparse_index = IndexDictOfArray(self.index_dir, dim_voc=dim_voc, force_new=force_new, filename=filename)
        self.compute_stats = compute_stats
        self.device = device
        if self.compute_stats:
            self.l0 = L0()
        self.model.to(self.device)
        self.local_rank = get_rank()
        self.world_size = get_world_size()
        print("world_size: {}, local_rank: {}".format(self.world_size, self.local_rank))

    GROUP_ROWS = 256          # index(): fixed-size loader batches encoded per engine pass (4 batches of 64)

    def _encode_batch(self, inputs, batch_ids):
        return self.model.encode(**inputs)

    def index(self, collection_loader, id_dict=None):
        doc_ids = {}
        stats = defaultdict(float)
        count = 0
        # index() builds a whole index: postings are adopted with set_csr, so a container that already holds postings
        # (a second index() call, or force_new=False over an existing file) would lose them while keeping their doc count
        if self.sparse_index.nb_docs() != 0 or len(self.sparse_index.doc_ids) != 0:
            raise ValueError("SparseIndexer.index builds a new index; this container already holds documents "
                             "(use force_new=True, or IndexDictOfArray.add_batch_document to append)")
        # COO triples stay on the device; ONE stable sort by term at the end builds the CSR (the reference appends
        # posting by posting in Python, inverted_index.py:74-76).  12 B per posting: 13.5 GB for MS MARCO at L0_d = 128.
        dev_rows, dev_cols, dev_vals = [], [], []
        n_batches = 0
        # The reference's loader yields fixed batches (64 passages: ~4 800 real tokens, a fraction of what the engine wants per
        # pass): fixed-size batches are encoded a few at a time in ONE engine pass (encode_batches: rows bit-identical to
        # batch-by-batch encode calls), then handled batch by batch exactly as before - statistics, postings and doc ids do not
        # change.  Token-budget loaders and subclasses with their own _encode_batch (HybridIndexer) go batch by batch.
        group_rows = (self.GROUP_ROWS if getattr(collection_loader, "batch_size", None) and hasattr(self.model, "encode_batches")
                      and type(self)._encode_batch is SparseIndexer._encode_batch else 0)

        def encoded_batches():
            for group in (batch_groups(collection_loader, group_rows) if group_rows else ([b] for b in collection_loader)):
                if len(group) > 1:
                    with torch.inference_mode(), torch.autocast("cuda", dtype=torch.bfloat16):  # indexer.py:255-256
                        reps_g = encode_group(self.model, group, self.device)
                    r0 = 0
                    for b in group:
                        n = len(b["ids"])
                        yield b, reps_g[r0:r0 + n]
                        r0 += n
                else:
                    yield group[0], None

        for t, (batch, batch_documents) in enumerate(tqdm(encoded_batches(), disable=not is_first_worker())):
            n_batches += 1
            batch_ids = to_list(batch["ids"]) if isinstance(batch["ids"], torch.Tensor) else batch["ids"]
            assert isinstance(batch_ids, list)
            if id_dict:
                batch_ids = [id_dict[x] for x in batch_ids]
            if batch_documents is None:
                inputs = {k: v.to(self.device) for k, v in batch.items() if k not in {"ids"}}
                with torch.inference_mode(), torch.autocast("cuda", dtype=torch.bfloat16):  # indexer.py:255-256
                    batch_documents = self._encode_batch(inputs, batch_ids)      # [bz, vocab_size] fp32 on device
            if self.compute_stats:
                stats["L0_d"] += self.l0(batch_documents).item()
            row_ptr, col, data = sparse_reps_to_csr(batch_documents)
            nnz_per_row = (row_ptr[1:] - row_ptr[:-1])
            row = torch.repeat_interleave(torch.arange(len(nnz_per_row), device=row_ptr.device), nnz_per_row) + count
            if (count + len(nnz_per_row)) * self.world_size + self.local_rank >= 2 ** 31:
                raise OverflowError("global document index exceeds int32 (the posting lists hold int32 doc ids, inverted_index.py:22-55)")
            dev_rows.append((row * self.world_size + self.local_rank).to(torch.int32))   # g_row = (row + count) * W + rank
            dev_cols.append(col.clone())
            dev_vals.append(data.clone())
            has_posting = np.nonzero((nnz_per_row > 0).cpu().numpy())[0]       # docs without any posting get no entry (:271-283)
            all_idxes = (count + has_posting) * self.world_size + self.local_rank
            doc_ids.update(zip(all_idxes.tolist(), (batch_ids[_i] for _i in has_posting.tolist())))
            count += len(batch_ids)
        if dev_rows:
            rows_t, cols_t, vals_t = torch.cat(dev_rows), torch.cat(dev_cols), torch.cat(dev_vals)
            del dev_rows, dev_cols, dev_vals
            V = max(int(self.sparse_index.dim_voc or 0), int(cols_t.max().item()) + 1 if cols_t.numel() else 0)
            # sr_sparse_csr_build (csrc/sparse_build.hip): this library's stable radix sort by term - insertion order kept inside a
            # term, i.e. the posting lists add_batch_document's appends would hold (inverted_index.py:67-76)
            indptr, rows_t, vals_t = sparse_csr_build(rows_t, cols_t, vals_t, V)
            del cols_t
            n_docs = (count - 1) * self.world_size + self.local_rank + 1 if self.world_size > 1 else count   # nb_docs() = max g_row + 1
            if self.index_dir is not None:       # the host copy is needed to write the files; the 12 B per posting in HBM are released
                self.sparse_index.set_csr(indptr.cpu().numpy(), rows_t.cpu().numpy(), vals_t.cpu().numpy(), n_docs)
                self.device_csr = None
            else:
                # the container carries the postings like the reference's in-memory index_d (host copy made on first use);
                # an immediately following SparseRetrieval(index_d=...) scores from the device arrays: no 13 GB round trip
                self.sparse_index.set_device_csr(indptr, rows_t, vals_t, n_docs)
                self.device_csr = (indptr, rows_t, vals_t, int(n_docs))

        if self.compute_stats:
            stats = {key: value / max(1, n_batches) for key, value in stats.items()}     # mean over batches, as the reference
        if self.index_dir is not None:
            self.sparse_index.save()
            pickle.dump(doc_ids, open(os.path.join(self.index_dir, "doc_ids.pkl"), "wb"))
            print("done iterating over the corpus...")
            print("index contains {} posting lists".format(len(self.sparse_index)))
            print("index contains {} documents".format(len(doc_ids)))
            if self.compute_stats:
                with open(os.path.join(self.index_dir, "index_stats.json"), "w") as handler:
                    json.dump(stats, handler)
        else:
            out = {"index": self.sparse_index, "ids_mapping": doc_ids, "device_csr": getattr(self, "device_csr", None)}
            if self.compute_stats:
                out["stats"] = stats
            return out


# ==================================================================== sparse: retrieval
def _csr_sorted_by_doc(indptr, doc_ids, vals, device):
    """Sort every posting list by doc id on the device (a merged multi-rank index is rank-major inside
    a term, inverted_index.py:139-146; per-doc sums do not depend on the order inside a list)."""
    indptr_t = torch.from_numpy(np.ascontiguousarray(indptr)).to(device)
    ids_t = torch.from_numpy(np.ascontiguousarray(doc_ids)).to(device)
    vals_t = torch.from_numpy(np.ascontiguousarray(vals)).to(device)
    if ids_t.numel():
        # sr_sparse_csr_build with sort_docs: stable radix passes over the doc rows, then over the terms (csrc/sparse_build.hip)
        term = sparse_csr_expand_terms(indptr_t, ids_t.numel())
        indptr2, ids_t, vals_t = sparse_csr_build(ids_t, term, vals_t, len(indptr) - 1, n_docs=int(ids_t.max().item()) + 1, sort_docs=True)
        assert torch.equal(indptr2, indptr_t.to(torch.int64))
    return indptr_t, ids_t, vals_t


class QueryCSR:
    """What _generate_query_vecs returns as `sparse_query_vecs`: the reference's list of (cols int32, vals fp32) pairs per query
    (indexer.py:393-401), held as ONE CSR on the device (row_ptr int64 [nq + 1], cols int32, vals fp32) - the form
    sr_sparse_search takes.  It still reads like the list (len, indexing, iteration give the per-query numpy pairs, from a host
    copy made on first use), so callers of the reference's interface keep working; the HIP scorer never leaves the device."""

    def __init__(self, row_ptr, cols, vals):
        self.row_ptr, self.cols, self.vals = row_ptr, cols, vals
        self._host = None

    @classmethod
    def cat(cls, parts):
        if len(parts) == 1:
            return parts[0]
        ptrs, base = [parts[0].row_ptr], int(parts[0].row_ptr[-1])
        for p_ in parts[1:]:
            ptrs.append(p_.row_ptr[1:] + base)
            base += int(p_.row_ptr[-1])
        return cls(torch.cat(ptrs), torch.cat([p_.cols for p_ in parts]), torch.cat([p_.vals for p_ in parts]))

    def host(self):
        if self._host is None:
            self._host = (self.row_ptr.cpu().numpy(), self.cols.cpu().numpy().astype(np.int32), self.vals.cpu().numpy().astype(np.float32))
        return self._host

    def __len__(self):
        return int(self.row_ptr.numel()) - 1

    def __getitem__(self, q):
        ptr, cols, vals = self.host()
        if isinstance(q, slice):
            return [self[i] for i in range(*q.indices(len(self)))]
        if q < 0:
            q += len(self)
        return cols[ptr[q]:ptr[q + 1]], vals[ptr[q]:ptr[q + 1]]

    def __iter__(self):
        return (self[i] for i in range(len(self)))

    def mean_l0(self):
        n = len(self)
        return float(self.cols.numel()) / n if n else 0.0


def _as_query_csr(sparse_query_vecs, device):
    if isinstance(sparse_query_vecs, QueryCSR):
        return sparse_query_vecs
    counts = [len(c) for c, _ in sparse_query_vecs]
    row_ptr = torch.from_numpy(np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)).to(device)
    cols = torch.from_numpy(np.concatenate([np.asarray(c, np.int32) for c, _ in sparse_query_vecs]) if counts else np.zeros(0, np.int32)).to(device)
    vals = torch.from_numpy(np.concatenate([np.asarray(v, np.float32) for _, v in sparse_query_vecs]) if counts else np.zeros(0, np.float32)).to(device)
    return QueryCSR(row_ptr, cols, vals)


def _doc_id_table(doc_ids, n_docs):
    """doc_ids.pkl (dict: global doc index -> collection id, docs without a posting absent; indexer.py:271-283) as an array over
    [0, n_docs): one vectorised look-up per result instead of a dict access per hit."""
    n = max(int(n_docs), (max(doc_ids) + 1) if len(doc_ids) else 0)
    keys = np.fromiter(doc_ids.keys(), dtype=np.int64, count=len(doc_ids))
    vals = list(doc_ids.values())
    # Docs without a posting have no id (they can never be hits); their slots get DISTINCT placeholders, so that a few empty documents
    # in a real collection do not make the id table look like it held duplicates (which sends run.json through the dict path)
    holes = np.ones(n, dtype=bool)
    holes[keys] = False
    hole_pos = np.nonzero(holes)[0]
    if vals and all(isinstance(v, (int, np.integer)) and not isinstance(v, bool) for v in vals):
        table = np.full(n, -1, dtype=np.int64)
        table[keys] = np.asarray(vals, dtype=np.int64)
        if len(hole_pos):
            lo = min(int(table[keys].min()) if len(keys) else 0, 0)
            table[hole_pos] = lo - 1 - np.arange(len(hole_pos), dtype=np.int64)       # below every real id
        return table
    table = np.empty(n, dtype=object)
    table[keys] = np.asarray(vals, dtype=object)
    for j, pos in enumerate(hole_pos.tolist()):
        table[pos] = f"\x00no-posting-{j}"
    return table


class SparseRetrieval:
    """indexer.py:311-540."""

    _static_cache = None      # (the caller's dict - held, so its id cannot be recycled -, device index, n_terms)

    def __init__(self, model, config, dim_voc, device, dataset_name=None, index_d=None, compute_stats=False,
                 is_beir=False, **kwargs):
        self.model = model
        self.model.eval()
        assert ("index_dir" in config and index_d is None) or ("index_dir" not in config and index_d is not None)
        if "index_dir" in config:
            self.sparse_index, self.doc_ids = self._open_index(config["index_dir"], dim_voc)
        else:
            self.sparse_index = index_d["index"]
            self.doc_ids = index_d["ids_mapping"]
        self.device = device
        self.model.to(device)
        dev = torch.device(device if not isinstance(device, int) else f"cuda:{device}")
        self._dev = dev
        dcsr = index_d.get("device_csr") if index_d is not None else None
        with torch.cuda.device(dev):
            if dcsr is not None and get_world_size() == 1:
                # built by SparseIndexer.index in this process: term-major, doc-ascending inside a term, already in HBM
                indptr_t, ids_t, vals_t, n_docs = dcsr
                if dim_voc is not None and indptr_t.numel() - 1 < dim_voc:
                    indptr_t = torch.cat([indptr_t, indptr_t[-1:].expand(dim_voc + 1 - indptr_t.numel())])
                self.hip_index = SparseIndexHIP(indptr_t, ids_t, vals_t, max(1, n_docs), device=dev)
            else:
                self.hip_index = self._build_hip_index(dim_voc, dev)
        self.out_dir = os.path.join(config["out_dir"], dataset_name) if (dataset_name is not None and not is_beir) \
            else config["out_dir"]
        self.doc_stats = index_d["stats"] if (index_d is not None and compute_stats) else None
        self.compute_stats = compute_stats
        if self.compute_stats:
            self.l0 = L0()

    def _open_index(self, index_dir, dim_voc):
        with open(os.path.join(index_dir, "doc_ids.pkl"), "rb") as f:
            return IndexDictOfArray(index_dir, dim_voc=dim_voc), pickle.load(f)

    def _build_hip_index(self, dim_voc, dev):
        indptr, ids, vals = self.sparse_index.csr(dim_voc)
        indptr_t, ids_t, vals_t = _csr_sorted_by_doc(indptr, ids, vals, dev)
        return SparseIndexHIP(indptr_t, ids_t, vals_t, max(1, self.sparse_index.nb_docs()), device=dev)

    # -- kept for callers of the reference's static helpers (indexer.py:315-344) ------------
    @staticmethod
    def select_topk(filtered_indexes, scores, k):
        if len(filtered_indexes) > k:
            order = np.lexsort((filtered_indexes, scores))[:k]      # scores are negated: ascending = best first
            return filtered_indexes[order], -scores[order]
        return filtered_indexes, -scores

    @staticmethod
    def numba_score_float(inverted_index_ids, inverted_index_floats, indexes_to_retrieve, query_values, threshold,
                          size_collection):
        """Same signature/returns as the reference: (doc indexes with score > threshold ascending int64,
        NEGATED scores fp32).  Scores come from the HIP scorer with k = all candidates capped at the top-k
        width, so use SparseRetrieval.retrieve for bulk work; this entry point exists for drop-in callers."""
        cache = SparseRetrieval._static_cache
        if cache is not None and cache[0] is inverted_index_ids and cache[3] == size_collection:
            hit = cache[1]
        else:
            terms = sorted(t for t in inverted_index_ids.keys() if len(inverted_index_ids[t]))
            V = (max(terms) + 1) if terms else 1
            counts = np.zeros(V, np.int64)
            for t in terms:
                counts[t] = len(inverted_index_ids[t])
            indptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
            ids = np.concatenate([np.asarray(inverted_index_ids[t], np.int32) for t in terms]) if terms else np.zeros(0, np.int32)
            vals = np.concatenate([np.asarray(inverted_index_floats[t], np.float32) for t in terms]) if terms else np.zeros(0, np.float32)
            dev = torch.device("cuda", torch.cuda.current_device())
            hit = SparseIndexHIP(*_csr_sorted_by_doc(indptr, ids, vals, dev), size_collection, device=dev)
            SparseRetrieval._static_cache = (inverted_index_ids, hit, V, size_collection)
        k = _lib.load().sr_max_topk()
        cols = np.asarray(indexes_to_retrieve, np.int32)
        s, i, c = hit.search(np.array([0, len(cols)], np.int64), cols, np.asarray(query_values, np.float32), k,
                             threshold=float(threshold))
        c = int(c.item())
        idx = i[0, :c].cpu().numpy()
        sc = s[0, :c].cpu().numpy()
        order = np.argsort(idx, kind="stable")
        return idx[order].astype(np.int64), -sc[order]

    QUERY_GROUP_ROWS = 2048      # rows per encode_batches call: the [rows, V] fp32 reps of a group live in HBM (1 GB at V = 128 256)

    def _generate_query_vecs(self, q_loader):
        """indexer.py:382-403: encode queries, keep the nonzero (col, value) pairs per query.  The loader's batches are encoded
        group-wise in one engine pass each (encode_batches: same bits as batch by batch) and the pairs stay on the device as ONE
        CSR (QueryCSR) - no per-batch .cpu(), no per-query slicing."""
        parts, qids = [], []
        for group in tqdm(batch_groups(q_loader, self.QUERY_GROUP_ROWS), desc="generate query vecs", disable=not is_first_worker()):
            with torch.inference_mode(), torch.autocast("cuda", dtype=torch.bfloat16):  # indexer.py:390-391
                reps = encode_group(self.model, group, self.device)
            for batch in group:
                qids.extend(batch["ids"] if isinstance(batch["ids"], list) else to_list(batch["ids"]))
            parts.append(QueryCSR(*sparse_reps_to_csr(reps)))
            del reps
        if not parts:
            dev = self._dev
            return QueryCSR(torch.zeros(1, dtype=torch.int64, device=dev), torch.zeros(0, dtype=torch.int32, device=dev),
                            torch.zeros(0, dtype=torch.float32, device=dev)), qids
        return QueryCSR.cat(parts), qids

    def doc_id_table(self):
        if getattr(self, "_doc_table", None) is None:
            self._doc_table = IdTable(_doc_id_table(self.doc_ids, self.sparse_index.nb_docs()))
        return self._doc_table

    def _sparse_retrieve_multithreaded(self, sparse_query_vecs, qids, threshold=0., topk=1000):
        """indexer.py:405-474 runs 4 Python threads x numba and fills res[str(qid)][str(doc_ids[id_])] hit by hit; here the
        whole query set is one batched HIP search (the doc space is tiled across workgroups instead) and `res` is a RunResult
        over the result arrays: the same mapping, without 7 M dict insertions."""
        q = _as_query_csr(sparse_query_vecs, self._dev)
        scores, ids, counts = self.hip_index.search(q.row_ptr, q.cols, q.vals, topk, threshold=threshold)
        res = RunResult(qids, to_host(scores), to_host(ids), self.doc_id_table(), to_host(counts))
        stats = defaultdict(float)
        stats["L0_q"] = q.mean_l0()
        return res, stats

    def _write_outputs(self, res, stats):
        os.makedirs(self.out_dir, exist_ok=True)
        if self.compute_stats:
            with open(os.path.join(self.out_dir, "q_stats.json"), "w") as handler:
                json.dump(stats, handler)
        res.dump(os.path.join(self.out_dir, "run.json"))        # sr_write_run_json: the bytes json.dump(res) writes

    def retrieve(self, q_loader, topk, threshold=0.):
        """indexer.py:530-540.  Query groups of QUERY_GROUP_ROWS rows go through encode -> search one after the other while a worker
        thread writes the previous group's piece of run.json (sr_write_run_json_part): formatting and the page-cache copy of a Dev-sized
        file take as long as the encode, and nothing in them needs the GPU.  A query's rows do not depend on the batch it is searched in
        (certified or exact: the same bits), and the file is the one-call file byte for byte (tests/test_boundary_gpu.py)."""
        groups = list(batch_groups(q_loader, self.QUERY_GROUP_ROWS))
        group_qids = [[x for batch in g for x in (batch["ids"] if isinstance(batch["ids"], list) else to_list(batch["ids"]))] for g in groups]
        qids = [x for g in group_qids for x in g]
        table = self.doc_id_table()
        if len(groups) < 2 or not (id_table(qids).distinct and id_table(table).distinct):
            sparse_query_vecs, qids = self._generate_query_vecs([batch for g in groups for batch in g])
            res, stats = self._sparse_retrieve_multithreaded(sparse_query_vecs, qids, threshold=threshold, topk=topk)
            self._write_outputs(res, stats)
            return res
        os.makedirs(self.out_dir, exist_ok=True)
        path = os.path.join(self.out_dir, "run.json")
        pieces, nnz = [], 0
        with PiecewiseRunWriter(path) as writer:                       # run.json appears only when its last piece is written
            for gi, group in enumerate(groups):
                with torch.inference_mode(), torch.autocast("cuda", dtype=torch.bfloat16):  # indexer.py:390-391
                    reps = encode_group(self.model, group, self.device)
                q = QueryCSR(*sparse_reps_to_csr(reps))
                del reps
                nnz += int(q.cols.numel())
                scores, ids, counts = self.hip_index.search(q.row_ptr, q.cols, q.vals, topk, threshold=threshold)
                piece = (to_host(scores), to_host(ids), to_host(counts))
                pieces.append(piece)
                writer.add(group_qids[gi], piece[0], piece[1], table, piece[2], last=gi + 1 == len(groups))
            writer.finish()
        res = RunResult(qids, np.concatenate([p_[0] for p_ in pieces]), np.concatenate([p_[1] for p_ in pieces]), table,
                        np.concatenate([p_[2] for p_ in pieces]))
        if self.compute_stats:
            stats = defaultdict(float)
            stats["L0_q"] = float(nnz) / len(qids) if qids else 0.0
            with open(os.path.join(self.out_dir, "q_stats.json"), "w") as handler:
                json.dump(stats, handler)
        return res


class ShardedSparseRetrieval(SparseRetrieval):
    """Doc-sharded SparseRetrieval for world_size > 1 (the reference asserts world_size == 1, eval_sparse.py:114, and
    needs a merge_indexes pass first, scripts/eval_sparse.sh:19): rank r opens the directory `{index_dir}_{r}` the indexing
    task wrote (eval_sparse.py:98-100), scores its own documents and the per-shard top-k meet in ONE gather
    (distributed.ShardedSparseRetriever).  Every rank encodes only its block of the queries; the query CSR is
    all-gathered (distributed.all_gather_query_csr).  run.json / q_stats.json are written by rank 0."""

    def __init__(self, model, config, dim_voc, device, rank=None, world_size=None, **kwargs):
        from .distributed import ShardedSparseRetriever
        self.rank = get_rank() if rank is None else rank
        self.world_size = get_world_size() if world_size is None else world_size
        base = config["index_dir"][:-1] if config["index_dir"].endswith("/") else config["index_dir"]
        self.shard_dir = f"{base}_{self.rank}"
        self._base_dir = base
        self._shard_factory = ShardedSparseRetriever
        super().__init__(model, dict(config, index_dir=self.shard_dir), dim_voc, device, **kwargs)

    def _open_index(self, index_dir, dim_voc):
        # a shard's doc_ids.pkl holds only the global rows r, r + W, ...: count them here, not with the single-index rule
        # (IndexDictOfArray: min key == 0, inverted_index.py:44-55)
        with open(os.path.join(index_dir, "doc_ids.pkl"), "rb") as f:
            doc_ids = pickle.load(f)
        index = IndexDictOfArray(index_dir, dim_voc=dim_voc, _count_docs=False)
        index.n = (max(doc_ids) + 1) if len(doc_ids) else 0      # max g_row + 1 of THIS shard
        return index, doc_ids

    def _build_hip_index(self, dim_voc, dev):
        indptr, ids, vals = self.sparse_index.csr(dim_voc)
        self.sharded = self._shard_factory(indptr, ids, vals, self.sparse_index.nb_docs(), rank=self.rank,
                                           world_size=self.world_size, device=dev)
        return self.sharded.index

    def _all_doc_ids(self):
        ids = {}
        for r in range(self.world_size):
            with open(os.path.join(f"{self._base_dir}_{r}", "doc_ids.pkl"), "rb") as f:
                ids.update(pickle.load(f))
        return ids

    def retrieve(self, q_loader, topk, threshold=0.):
        """q_loader yields THIS rank's block of the queries (distributed.query_slice order) or all of them when
        `q_loader.replicated` is set."""
        from .distributed import all_gather_query_csr
        import torch.distributed as dist
        q, qids = self._generate_query_vecs(q_loader)
        row_ptr, cols, vals = q.row_ptr, q.cols, q.vals
        if self.world_size > 1 and not getattr(q_loader, "replicated", False):
            all_qids = [None] * self.world_size
            dist.all_gather_object(all_qids, qids)
            qids = [q_ for part in all_qids for q_ in part]
            row_ptr, cols, vals = all_gather_query_csr(row_ptr, cols, vals, len(qids))
        scores, ids, counts_t = self.sharded.search(row_ptr, cols, vals, topk, threshold=threshold)
        if scores is None:
            return None
        doc_ids = self._all_doc_ids()
        n_docs = (max(doc_ids) + 1) if len(doc_ids) else 0
        res = RunResult(qids, to_host(scores), to_host(ids), IdTable(_doc_id_table(doc_ids, n_docs)), to_host(counts_t))
        nnz = (row_ptr[1:] - row_ptr[:-1])
        stats = {"L0_q": float(nnz.float().mean().item()) if nnz.numel() else 0.0}
        self._write_outputs(res, stats)
        return res


# ============================================================================== hybrid
class HybridIndexer(SparseIndexer):
    """indexer.py:710-857: ONE pass over the collection writes both the inverted index (sparse_index_dir) and the dense
    shard files embs_{rank}_{chunk}.npy / ids_*.npy / plan.json (dense_index_dir).  model.encode returns
    (sparse reps [B, V], dense reps [B, H]) from one backbone pass (LlamaBiHybrid -> sr_encode_both)."""

    def __init__(self, model, sparse_index_dir, dense_index_dir, device, chunk_size=2_000_000, compute_stats=False,
                 dim_voc=None, force_new=True, filename="array_index.h5py", **kwargs):
        super().__init__(model, sparse_index_dir, device, compute_stats=compute_stats, dim_voc=dim_voc, force_new=force_new,
                         filename=filename)
        self.sparse_index_dir, self.dense_index_dir, self.chunk_size = sparse_index_dir, dense_index_dir, chunk_size
        os.makedirs(dense_index_dir, exist_ok=True)
        self._dense, self._dense_ids, self._chunk_idx = [], [], 0

    def _flush_dense(self):
        embs = torch.cat(self._dense).numpy()
        ids = self._dense_ids
        if isinstance(ids[0], int):
            ids = np.array(ids, dtype=np.int64)
        assert len(embs) == len(ids), (len(embs), len(ids))
        np.save(os.path.join(self.dense_index_dir, "embs_{}_{}.npy".format(self.local_rank, self._chunk_idx)), embs)
        np.save(os.path.join(self.dense_index_dir, "ids_{}_{}.npy".format(self.local_rank, self._chunk_idx)), ids)
        self._dense, self._dense_ids = [], []
        self._chunk_idx += 1

    def _encode_batch(self, inputs, batch_ids):
        sparse, dense = self.model.encode(**inputs)
        # to the host batch by batch, as store_embs does (indexer.py:56): a whole chunk of 2 M x H fp32 rows (16 GB at H = 2048,
        # 32 GB at 4096) would otherwise sit in HBM next to the COO triples of the sparse index
        self._dense.append(dense.float().cpu())
        self._dense_ids.extend(batch_ids)
        if len(self._dense_ids) >= self.chunk_size:
            self._flush_dense()
        return sparse

    def index(self, collection_loader, id_dict=None):
        out = super().index(collection_loader, id_dict=id_dict)
        if self._dense:
            print("last embedddings shape = {}".format((len(self._dense_ids), self._dense[0].shape[1])))
            self._flush_dense()
        plan = {"nranks": get_world_size(), "num_chunks": self._chunk_idx, "index_path": os.path.join(self.dense_index_dir, "model.index")}
        print("plan: ", plan)
        if is_first_worker():
            with open(os.path.join(self.dense_index_dir, "plan.json"), "w") as fout:
                json.dump(plan, fout)
        return out


class HybridRetriever(SparseRetrieval):
    """indexer.py:859-1019: queries are encoded once (both heads), scored against the inverted index and against the flat
    dense index; writes {out_dir}/sparse/run.json + q_stats.json and {out_dir}/dense/run.json."""

    def __init__(self, model, sparse_index_dir, dense_index_dir, out_dir, dim_voc, device, **kwargs):
        from .utils.utils import obtain_doc_vec_dir_files
        super().__init__(model, {"index_dir": sparse_index_dir, "out_dir": out_dir}, dim_voc, device, compute_stats=True)
        self.sparse_out_dir, self.dense_out_dir = os.path.join(out_dir, "sparse"), os.path.join(out_dir, "dense")
        if is_first_worker():
            os.makedirs(self.sparse_out_dir, exist_ok=True)
            os.makedirs(self.dense_out_dir, exist_ok=True)
        self.dense_index = DenseFlatIndexer()
        self.dense_index.init_index(self.model.hidden_size)
        self._index_encoded_data(*obtain_doc_vec_dir_files(dense_index_dir))

    def _index_encoded_data(self, doc_vec_files, doc_id_files):
        total = 0
        for doc_file, id_file in zip(doc_vec_files, doc_id_files):       # one HBM segment per shard file
            total += self.dense_index.index_data(np.load(doc_file, mmap_mode="r"), np.load(id_file).tolist())
        print("size of doc reps to index: ", total)

    def _generate_query_vecs(self, q_loader):
        parts, dense_query_vecs, qids = [], [], []
        for group in tqdm(batch_groups(q_loader, self.QUERY_GROUP_ROWS), desc="generate query vecs", disable=not is_first_worker()):
            with torch.inference_mode(), torch.autocast("cuda", dtype=torch.bfloat16):      # indexer.py:938-939
                batch_sparse_reps, batch_dense_reps = encode_group(self.model, group, self.device)
            for batch in group:
                qids.extend(batch["ids"] if isinstance(batch["ids"], list) else to_list(batch["ids"]))
            dense_query_vecs.append(batch_dense_reps)
            parts.append(QueryCSR(*sparse_reps_to_csr(batch_sparse_reps)))
            del batch_sparse_reps
        sparse_query_vecs = QueryCSR.cat(parts)
        dense_query_vecs = torch.cat(dense_query_vecs)
        assert len(sparse_query_vecs) == len(dense_query_vecs) == len(qids)
        return sparse_query_vecs, dense_query_vecs, qids

    def _dense_retrieve(self, query_reps, qids, topk=1000):
        """indexer.py:973-982, as a RunResult over the result arrays (label -1 rows = fewer than k vectors are skipped)."""
        scores, positions = self.dense_index.search_arrays(query_reps, topk)
        return RunResult(qids, scores, positions, self.dense_index.run_table())

    def _sparse_retrieve(self, sparse_query_vecs, qids, threshold=0., topk=1000):
        return self._sparse_retrieve_multithreaded(sparse_query_vecs, qids, threshold=threshold, topk=topk)

    def retrieve(self, q_loader, topk, id_dict=False, threshold=0.):
        sparse_query_vecs, dense_query_vecs, qids = self._generate_query_vecs(q_loader)
        sparse_res, sparse_stats = self._sparse_retrieve(sparse_query_vecs, qids, threshold=threshold, topk=topk)
        dense_res = self._dense_retrieve(dense_query_vecs, qids, topk=topk)
        with open(os.path.join(self.sparse_out_dir, "q_stats.json"), "w") as handler:
            json.dump(sparse_stats, handler)
        sparse_res.dump(os.path.join(self.sparse_out_dir, "run.json"))
        dense_res.dump(os.path.join(self.dense_out_dir, "run.json"))
        return sparse_res, dense_res
