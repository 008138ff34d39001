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
from .scoring import DenseIndexHIP, SparseIndexHIP
from .utils.inverted_index import IndexDictOfArray
from .utils.utils import get_rank, get_world_size, is_first_worker, to_list

logger = logging.getLogger()


def _unwrap(model):
    return model.module if hasattr(model, "module") else model


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

    def flush():
        nonlocal embeddings, embeddings_ids, chunk_idx
        embs = torch.cat(embeddings).float().cpu().numpy()
        ids = embeddings_ids
        if isinstance(ids[0], int):
            ids = np.array(ids, dtype=np.int64)
        assert len(embs) == len(ids), (len(embs), len(ids))
        np.save(os.path.join(index_dir, "embs_{}_{}.npy".format(local_rank, chunk_idx)), embs)
        np.save(os.path.join(index_dir, "ids_{}_{}.npy".format(local_rank, chunk_idx)), ids)
        embeddings, embeddings_ids = [], []
        chunk_idx += 1

    try:
        total = len(collection_loader)
    except TypeError:
        total = None              # token-budget loaders do not know their batch count up front
    for idx, batch in tqdm(enumerate(collection_loader), disable=not is_first_worker(),
                           desc="encode # {} seqs".format(total if total is not None else "?"), total=total):
        inputs = {k: v.to(device, non_blocking=True) for k, v in batch.items() if k != "ids"}
        with torch.inference_mode(), torch.autocast("cuda", dtype=torch.bfloat16):      # indexer.py:46-52
            reps = enc.doc_encode(**inputs)
        text_ids = batch["ids"]
        assert isinstance(text_ids, list)
        embeddings.append(reps)
        embeddings_ids.extend(text_ids)
        if len(embeddings_ids) >= chunk_docs:
            flush()
    if len(embeddings) != 0:
        print("last embedddings shape = {}".format((sum(len(e) for e in embeddings), embeddings[0].shape[1])))
        flush()

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
        return len(self.index_id_to_db_id)


class DenseFlatIndexer(DenseIndexer):
    """indexer.py:191-217 over a flat inner-product index resident in HBM."""

    def __init__(self, buffer_size: int = 50000):
        super().__init__(buffer_size)
        self.hidden_dim = None

    def init_index(self, hidden_dim):
        self.hidden_dim = int(hidden_dim)
        self.index = DenseIndexHIP(self.hidden_dim)

    def index_data(self, doc_reps, doc_ids):
        assert len(doc_reps) == len(doc_ids)
        n = len(doc_reps)
        if isinstance(doc_reps, torch.Tensor) and doc_reps.is_cuda:
            self.index.add_device_rows(doc_reps.float())
        else:
            self.index.add_host_rows(np.asarray(doc_reps), buffer_size=self.buffer_size)
        n_total = self._update_id_mapping(list(doc_ids))
        logger.info("total data indexed %d", n_total)
        assert self.index.ntotal == n_total, (self.index.ntotal, n_total)
        return n

    def search_knn(self, query_reps, top_docs: int):
        if isinstance(query_reps, torch.Tensor):
            q = query_reps.to(device=self.index.device, dtype=torch.float32)
        else:
            q = torch.from_numpy(np.ascontiguousarray(query_reps, dtype=np.float32)).to(self.index.device)
        scores, indexes = self.index.search(q, top_docs)
        scores, indexes = scores.cpu().numpy(), indexes.cpu().numpy()
        table = np.empty(len(self.index_id_to_db_id) + 1, dtype=object)
        table[:-1] = self.index_id_to_db_id
        table[-1] = None                                    # faiss label -1 (fewer than k vectors)
        top_doc_ids = [list(table[row]) for row in indexes]
        return top_doc_ids, scores

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

    def index(self, collection_loader, id_dict=None):
        doc_ids = {}
        stats = defaultdict(float)
        count = 0
        # COO triples stay on the device; ONE stable sort by term at the end builds the CSR (the reference appends
        # posting by posting in Python, inverted_index.py:74-76).  12 B per posting: 13.5 GB for MS MARCO at L0_d = 128.
        dev_rows, dev_cols, dev_vals = [], [], []
        for t, batch in enumerate(tqdm(collection_loader, disable=not is_first_worker())):
            inputs = {k: v.to(self.device) for k, v in batch.items() if k not in {"ids"}}
            with torch.inference_mode(), torch.autocast("cuda", dtype=torch.bfloat16):  # indexer.py:255-256
                batch_documents = self.model.encode(**inputs)      # [bz, vocab_size] fp32 on device
            if self.compute_stats:
                stats["L0_d"] += self.l0(batch_documents).item()
            row_ptr, col, data = sparse_reps_to_csr(batch_documents)
            nnz_per_row = (row_ptr[1:] - row_ptr[:-1])
            row = torch.repeat_interleave(torch.arange(len(nnz_per_row), device=row_ptr.device), nnz_per_row) + count
            dev_rows.append((row * self.world_size + self.local_rank).to(torch.int32))   # g_row = (row + count) * W + rank
            dev_cols.append(col.clone())
            dev_vals.append(data.clone())
            batch_ids = to_list(batch["ids"]) if isinstance(batch["ids"], torch.Tensor) else batch["ids"]
            assert isinstance(batch_ids, list)
            if id_dict:
                batch_ids = [id_dict[x] for x in batch_ids]
            has_posting = (nnz_per_row > 0).cpu().numpy()
            all_idxes = (count + np.arange(len(batch_ids))) * self.world_size + self.local_rank
            for _i, _idx in enumerate(all_idxes):            # docs without any posting get no entry (:271-283)
                if has_posting[_i]:
                    doc_ids[int(_idx)] = batch_ids[_i]
            count += len(batch_ids)
        if dev_rows:
            rows_t, cols_t, vals_t = torch.cat(dev_rows), torch.cat(dev_cols), torch.cat(dev_vals)
            del dev_rows, dev_cols, dev_vals
            order = torch.sort(cols_t.to(torch.int32), stable=True).indices      # insertion order kept inside a term
            V = max(int(self.sparse_index.dim_voc or 0), int(cols_t.max().item()) + 1 if cols_t.numel() else 0)
            counts = torch.bincount(cols_t.long(), minlength=V)
            indptr = torch.cat([torch.zeros(1, dtype=torch.int64, device=counts.device), torch.cumsum(counts, 0)])
            self.sparse_index.set_csr(indptr.cpu().numpy(), rows_t[order].cpu().numpy(), vals_t[order].cpu().numpy(),
                                      self.sparse_index.nb_docs() + count)

        if self.compute_stats:
            stats = {key: value / len(collection_loader) for key, value in stats.items()}
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
            out = {"index": self.sparse_index, "ids_mapping": doc_ids}
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
        term = torch.repeat_interleave(torch.arange(len(indptr) - 1, device=device), indptr_t[1:] - indptr_t[:-1])
        key = term * (int(ids_t.max().item()) + 1) + ids_t.long()
        order = torch.argsort(key)
        ids_t, vals_t = ids_t[order].contiguous(), vals_t[order].contiguous()
    return indptr_t, ids_t, vals_t


class SparseRetrieval:
    """indexer.py:311-540."""

    _static_cache = {}

    def __init__(self, model, config, dim_voc, device, dataset_name=None, index_d=None, compute_stats=False,
                 is_beir=False, **kwargs):
        self.model = model
        self.model.eval()
        assert ("index_dir" in config and index_d is None) or ("index_dir" not in config and index_d is not None)
        if "index_dir" in config:
            self.sparse_index = IndexDictOfArray(config["index_dir"], dim_voc=dim_voc)
            self.doc_ids = pickle.load(open(os.path.join(config["index_dir"], "doc_ids.pkl"), "rb"))
        else:
            self.sparse_index = index_d["index"]
            self.doc_ids = index_d["ids_mapping"]
        self.device = device
        self.model.to(device)
        dev = torch.device(device if not isinstance(device, int) else f"cuda:{device}")
        indptr, ids, vals = self.sparse_index.csr(dim_voc)
        with torch.cuda.device(dev):
            indptr_t, ids_t, vals_t = _csr_sorted_by_doc(indptr, ids, vals, dev)
            self.hip_index = SparseIndexHIP(indptr_t, ids_t, vals_t, max(1, self.sparse_index.nb_docs()), device=dev)
        self.out_dir = os.path.join(config["out_dir"], dataset_name) if (dataset_name is not None and not is_beir) \
            else config["out_dir"]
        self.doc_stats = index_d["stats"] if (index_d is not None and compute_stats) else None
        self.compute_stats = compute_stats
        if self.compute_stats:
            self.l0 = L0()

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
        key = id(inverted_index_ids)
        hit = SparseRetrieval._static_cache.get(key)
        if hit is None:
            terms = sorted(inverted_index_ids.keys())
            V = (max(terms) + 1) if terms else 1
            counts = np.zeros(V, np.int64)
            for t in terms:
                counts[t] = len(inverted_index_ids[t])
            indptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
            ids = np.concatenate([np.asarray(inverted_index_ids[t], np.int32) for t in terms]) if terms else np.zeros(0, np.int32)
            vals = np.concatenate([np.asarray(inverted_index_floats[t], np.float32) for t in terms]) if terms else np.zeros(0, np.float32)
            dev = torch.device("cuda", torch.cuda.current_device())
            hit = SparseIndexHIP(*_csr_sorted_by_doc(indptr, ids, vals, dev), size_collection, device=dev)
            SparseRetrieval._static_cache = {key: hit}
        k = _lib.load().sr_max_topk()
        cols = np.asarray(indexes_to_retrieve, np.int32)
        s, i, c = hit.search(np.array([0, len(cols)], np.int64), cols, np.asarray(query_values, np.float32), k,
                             threshold=float(threshold))
        c = int(c.item())
        idx = i[0, :c].cpu().numpy()
        sc = s[0, :c].cpu().numpy()
        order = np.argsort(idx, kind="stable")
        return idx[order].astype(np.int64), -sc[order]

    def _generate_query_vecs(self, q_loader):
        """indexer.py:382-403: encode queries, keep the nonzero (col, value) pairs per query."""
        sparse_query_vecs, qids = [], []
        for t, batch in enumerate(tqdm(q_loader, total=len(q_loader), desc="generate query vecs",
                                       disable=not is_first_worker())):
            inputs = {k: v.to(self.device) for k, v in batch.items() if k not in {"ids"}}
            with torch.inference_mode(), torch.autocast("cuda", dtype=torch.bfloat16):  # indexer.py:390-391
                batch_sparse_reps = self.model.encode(**inputs)
            qids.extend(batch["ids"] if isinstance(batch["ids"], list) else to_list(batch["ids"]))
            row_ptr, cols, vals = sparse_reps_to_csr(batch_sparse_reps)
            row_ptr, cols, vals = row_ptr.cpu().numpy(), cols.cpu().numpy(), vals.cpu().numpy()
            for b in range(len(row_ptr) - 1):
                sparse_query_vecs.append((cols[row_ptr[b]:row_ptr[b + 1]].astype(np.int32),
                                          vals[row_ptr[b]:row_ptr[b + 1]].astype(np.float32)))
        return sparse_query_vecs, qids

    def _sparse_retrieve_multithreaded(self, sparse_query_vecs, qids, threshold=0., topk=1000):
        """indexer.py:405-474 runs 4 Python threads x numba; here the whole query set is one batched HIP
        search (the doc space is tiled across workgroups instead)."""
        q_indptr = np.concatenate([[0], np.cumsum([len(c) for c, _ in sparse_query_vecs])]).astype(np.int64)
        q_cols = np.concatenate([c for c, _ in sparse_query_vecs]) if len(sparse_query_vecs) else np.zeros(0, np.int32)
        q_vals = np.concatenate([v for _, v in sparse_query_vecs]) if len(sparse_query_vecs) else np.zeros(0, np.float32)
        scores, ids, counts = self.hip_index.search(q_indptr, q_cols, q_vals, topk, threshold=threshold)
        scores, ids, counts = scores.cpu().numpy(), ids.cpu().numpy(), counts.cpu().numpy()
        res = defaultdict(dict)
        stats = defaultdict(float)
        for qi, qid in enumerate(qids):
            r = res[str(qid)]
            for id_, sc in zip(ids[qi, :counts[qi]], scores[qi, :counts[qi]]):
                r[str(self.doc_ids[int(id_)])] = float(sc)
            stats["L0_q"] += len(sparse_query_vecs[qi][0]) / max(1, len(qids))
        return res, stats

    def retrieve(self, q_loader, topk, threshold=0.):
        sparse_query_vecs, qids = self._generate_query_vecs(q_loader)
        res, stats = self._sparse_retrieve_multithreaded(sparse_query_vecs, qids, threshold=threshold, topk=topk)
        os.makedirs(self.out_dir, exist_ok=True)
        if self.compute_stats:
            with open(os.path.join(self.out_dir, "q_stats.json"), "w") as handler:
                json.dump(stats, handler)
        with open(os.path.join(self.out_dir, "run.json"), "w") as handler:
            json.dump(res, handler)
        return res
