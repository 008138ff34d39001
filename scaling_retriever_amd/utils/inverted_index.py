"""Inverted-index container: mirror of
/root/reference/scaling_retriever/utils/inverted_index.py (IndexDictOfArray :15-105,
merge_indexes :108-170) with a CSR-by-term core.

The reference keeps `dict[term] -> array('I') doc ids / array('f') values` and appends one
posting per Python loop iteration (:67-76).  Here a batch is appended as whole COO arrays
and the per-term view (`index_doc_id[t]`, `index_doc_value[t]`) is materialised on demand
from one stable sort by term, which yields exactly the reference's posting order
(insertion order inside each term).  On-disk layout: the reference's h5py datasets
(`dim`, `index_doc_id_{t}`, `index_doc_value_{t}`) when h5py is importable, otherwise one
`array_index.npz` holding the CSR (indptr, doc_ids, vals, dim) - same information.
"""
import json
import os
import pickle

import numpy as np

try:  # optional, the reference's file format
    import h5py  # noqa: F401
    HAVE_H5PY = True
except Exception:  # pragma: no cover
    HAVE_H5PY = False


class _TermView:
    """dict-like `term -> np.ndarray` view over a CSR (missing terms -> empty arrays)."""

    def __init__(self, owner, which):
        self.o, self.which = owner, which

    def __getitem__(self, t):
        o = self.o
        o._finalize()
        if t < 0 or t >= len(o.indptr) - 1:
            return np.array([], dtype=np.int32 if self.which == "ids" else np.float32)
        b, e = o.indptr[t], o.indptr[t + 1]
        return (o.doc_ids if self.which == "ids" else o.vals)[b:e]

    def __contains__(self, t):
        o = self.o
        o._finalize()
        return 0 <= t < len(o.indptr) - 1 and o.indptr[t + 1] > o.indptr[t]

    def keys(self):
        o = self.o
        o._finalize()
        return [int(t) for t in np.nonzero(np.diff(o.indptr))[0]]

    def items(self):
        return [(t, self[t]) for t in self.keys()]

    def __len__(self):
        return len(self.keys())


class IndexDictOfArray:
    def __init__(self, index_path=None, force_new=False, filename="array_index.h5py", dim_voc=None, _count_docs=True):
        self.dim_voc = dim_voc
        self._pending = []          # [(rows int64, cols int32, vals fp32)]
        self.indptr = np.zeros(1 if dim_voc is None else dim_voc + 1, dtype=np.int64)
        self.doc_ids = np.zeros(0, dtype=np.int32)
        self.vals = np.zeros(0, dtype=np.float32)
        self.n = 0
        self.index_path = index_path
        self.filename = None
        if index_path is not None:
            os.makedirs(index_path, exist_ok=True)
            self.filename = os.path.join(index_path, filename)
            existing = self._existing_file()
            if existing and not force_new:
                print("index already exists, loading...")
                self._load(existing, dim_voc)
                doc_ids = pickle.load(open(os.path.join(index_path, "doc_ids.pkl"), "rb")) if _count_docs else []
                if isinstance(doc_ids, list):
                    self.n = len(doc_ids)
                else:  # dict g_row -> pid (inverted_index.py:44-55)
                    keys = list(doc_ids)
                    assert min(keys) == 0, min(keys)
                    self.n = max(keys) + 1
                print("done loading index...")
            else:
                print("initializing new index...")
        else:
            print("initializing new index...")
        self.index_doc_id = _TermView(self, "ids")
        self.index_doc_value = _TermView(self, "vals")

    # ---- build ---------------------------------------------------------------------
    def add_batch_document(self, row, col, data, n_docs=-1):
        """inverted_index.py:67-76: append (doc row, term col, value) triples."""
        row = np.asarray(row).astype(np.int64, copy=False)
        col = np.asarray(col).astype(np.int32, copy=False)
        data = np.asarray(data).astype(np.float32, copy=False)
        self.n += len(set(row.tolist())) if n_docs < 0 else n_docs
        if len(row):
            self._pending.append((row, col, data))

    def set_device_csr(self, indptr, doc_ids, vals, n_docs):
        """Adopt a CSR that still lives on the device (torch tensors; SparseIndexer.index without an index_dir): the host copy is
        made by the first access that needs one (csr(), a per-term view, save(), merge) - an immediately following
        SparseRetrieval scores from the device arrays and never triggers it."""
        self._pending = []
        self._device_csr = (indptr, doc_ids, vals)
        self.n = int(n_docs)

    def _finalize(self):
        dev = getattr(self, "_device_csr", None)
        if dev is not None:
            self._device_csr = None
            self.indptr = np.ascontiguousarray(dev[0].cpu().numpy(), dtype=np.int64)
            self.doc_ids = np.ascontiguousarray(dev[1].cpu().numpy(), dtype=np.int32)
            self.vals = np.ascontiguousarray(dev[2].cpu().numpy(), dtype=np.float32)
        if not self._pending:
            return
        rows = np.concatenate([p[0] for p in self._pending])
        cols = np.concatenate([p[1] for p in self._pending])
        vals = np.concatenate([p[2] for p in self._pending])
        self._pending = []
        V = max(int(cols.max()) + 1, len(self.indptr) - 1, self.dim_voc or 0)
        if len(self.doc_ids):  # merge with what is already in CSR form (keeps old postings first)
            old_cols = np.repeat(np.arange(len(self.indptr) - 1, dtype=np.int32), np.diff(self.indptr))
            rows = np.concatenate([self.doc_ids.astype(np.int64), rows])
            cols = np.concatenate([old_cols, cols])
            vals = np.concatenate([self.vals, vals])
        order = np.argsort(cols, kind="stable")
        self.doc_ids = rows[order].astype(np.int32)
        self.vals = vals[order]
        counts = np.bincount(cols, minlength=V).astype(np.int64)
        self.indptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)

    def set_csr(self, indptr, doc_ids, vals, n_docs):
        """Adopt a finished CSR (built on the device by SparseIndexer): postings inside a term in insertion order."""
        self._pending = []
        self._device_csr = None
        self.indptr = np.ascontiguousarray(indptr, dtype=np.int64)
        self.doc_ids = np.ascontiguousarray(doc_ids, dtype=np.int32)
        self.vals = np.ascontiguousarray(vals, dtype=np.float32)
        self.n = int(n_docs)

    def csr(self, dim_voc=None):
        """(indptr int64 [V+1], doc_ids int32, vals fp32) with V >= dim_voc."""
        self._finalize()
        V = max(len(self.indptr) - 1, dim_voc or 0, self.dim_voc or 0)
        indptr = self.indptr
        if len(indptr) - 1 < V:
            indptr = np.concatenate([indptr, np.full(V - (len(indptr) - 1), indptr[-1], dtype=np.int64)])
        return indptr, self.doc_ids, self.vals

    def __len__(self):
        return len(self.index_doc_id)

    def nb_docs(self):
        return self.n

    # ---- disk ----------------------------------------------------------------------
    def _existing_file(self):
        if self.filename and os.path.exists(self.filename):
            return self.filename
        npz = os.path.join(self.index_path, "array_index.npz") if self.index_path else None
        if npz and os.path.exists(npz):
            return npz
        return None

    def _load(self, path, dim_voc):
        if path.endswith(".npz"):
            z = np.load(path)
            self.indptr, self.doc_ids, self.vals = z["indptr"], z["doc_ids"], z["vals"]
            if dim_voc is not None and dim_voc + 1 < len(self.indptr):   # reference reads range(dim_voc) only
                self.indptr = self.indptr[:dim_voc + 1]
                self.doc_ids, self.vals = self.doc_ids[:self.indptr[-1]], self.vals[:self.indptr[-1]]
            return
        if not HAVE_H5PY:
            raise ImportError(f"{path} is an h5py index but h5py is not installed")
        import h5py
        with h5py.File(path, "r") as f:
            dim = dim_voc if dim_voc is not None else int(f["dim"][()])
            ids, vals, counts = [], [], np.zeros(dim, dtype=np.int64)
            for key in range(dim):
                name = "index_doc_id_{}".format(key)
                if name in f:
                    a = np.array(f[name], dtype=np.int32)
                    ids.append(a)
                    vals.append(np.array(f["index_doc_value_{}".format(key)], dtype=np.float32))
                    counts[key] = len(a)
        self.doc_ids = np.concatenate(ids) if ids else np.zeros(0, np.int32)
        self.vals = np.concatenate(vals) if vals else np.zeros(0, np.float32)
        self.indptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)

    def save(self, dim=None):
        """inverted_index.py:84-105 (+ index_dist.json: size of each posting list)."""
        self._finalize()
        keys = self.index_doc_id.keys()
        if HAVE_H5PY and self.filename.endswith(".h5py"):
            import h5py
            with h5py.File(self.filename, "w") as f:
                f.create_dataset("dim", data=int(dim) if dim else len(keys))
                for key in keys:
                    f.create_dataset("index_doc_id_{}".format(key), data=self.index_doc_id[key])
                    f.create_dataset("index_doc_value_{}".format(key), data=self.index_doc_value[key])
        else:
            np.savez(os.path.join(self.index_path, "array_index.npz"), indptr=self.indptr, doc_ids=self.doc_ids,
                     vals=self.vals, dim=np.int64(int(dim) if dim else len(keys)))
        index_dist = {int(k): int(self.indptr[k + 1] - self.indptr[k]) for k in keys}
        json.dump(index_dist, open(os.path.join(self.index_path, "index_dist.json"), "w"))


def merge_indexes(model_name_or_path, filename="array_index.h5py", index_name="index", index_dir=None):
    """inverted_index.py:108-170: concatenate the per-rank indexes term by term (rank dirs in
    os.listdir order), union doc_ids, average L0_d."""
    with open(os.path.join(model_name_or_path, "config.json")) as fin:
        dim_voc = json.load(fin)["vocab_size"]
    root = index_dir if index_dir is not None else model_name_or_path
    index_dirs = [os.path.join(root, d) for d in os.listdir(root) if d.startswith(index_name)]
    assert len(index_dirs) in [1, 2, 4, 8], index_dirs
    if len(index_dirs) == 1:
        print("only one index, no need to merge")
        return
    out = IndexDictOfArray(dim_voc=dim_voc)
    doc_ids, index_dist, index_stats = {}, {}, {"L0_d": 0}
    for d in index_dirs:
        part = IndexDictOfArray(d, dim_voc=dim_voc, filename=filename, _count_docs=False)
        indptr, ids, vals = part.csr(dim_voc)
        cols = np.repeat(np.arange(len(indptr) - 1, dtype=np.int32), np.diff(indptr))
        out.add_batch_document(ids.astype(np.int64), cols, vals, n_docs=0)
        with open(os.path.join(d, "doc_ids.pkl"), "rb") as f:
            doc_ids.update(pickle.load(f))
        with open(os.path.join(d, "index_dist.json")) as f:
            index_dist.update(json.load(f))
        with open(os.path.join(d, "index_stats.json")) as f:
            index_stats["L0_d"] += json.load(f)["L0_d"] / len(index_dirs)
    out_dir = os.path.join(root, index_name)
    os.makedirs(out_dir, exist_ok=True)
    out.index_path, out.filename = out_dir, os.path.join(out_dir, filename)
    out.save(dim=dim_voc)
    with open(os.path.join(out_dir, "doc_ids.pkl"), "wb") as f:
        pickle.dump(doc_ids, f)
    with open(os.path.join(out_dir, "index_dist.json"), "w") as f:
        json.dump(index_dist, f)
    with open(os.path.join(out_dir, "index_stats.json"), "w") as f:
        json.dump(index_stats, f)


if __name__ == "__main__":  # scripts/eval_sparse.sh:19 `python -m utils.inverted_index --model_name_or_path ...`
    import argparse
    parser = argparse.ArgumentParser()
    parser.add_argument("--model_name_or_path", type=str, required=True)
    parser.add_argument("--index_name", default="index", type=str)
    parser.add_argument("--index_dir", default=None, type=str)
    a = parser.parse_args()
    merge_indexes(a.model_name_or_path, index_name=a.index_name, index_dir=a.index_dir)
