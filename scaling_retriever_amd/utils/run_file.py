"""run.json without a per-hit Python loop.

The reference's retrieval drivers turn the top-k arrays into a nested dict hit by hit and json.dump it:
  /root/reference/eval_dense.py:225-241           qid_to_rankdata[str(qid)][str(docid)] = float(score); ujson.dump
  /root/reference/scaling_retriever/indexer.py:431-432,530-540   res[str(qid)][str(doc_ids[id_])] = float(sc); json.dump(res)
At MS MARCO Dev (6 980 x 1 000) that is 7 M dict insertions and a 217 MB dump - ~15 s of host time behind a 0.25 s search.
Here the result stays in the typed arrays the search returned:

  IdTable     the index-position -> document-id table as typed arrays (int64, fixed-width ASCII, or escaped JSON string bodies);
  RunResult   what `retrieve()` returns: a read-only Mapping with the dict's behaviour (res[qid][docid] -> float, iteration,
              len, == with a dict of dicts, .to_dict()) that materialises a query's dict only when it is asked for;
  write_run_json   the file, written by libsr_hip.so's sr_write_run_json (host threads): byte for byte what Python's json.dump
              writes for the nested dict - which is what the reference's sparse path calls (indexer.py:537-538).  Its dense path
              calls ujson.dump (eval_dense.py:240): same content - json.load gives equal dicts - but compact separators, its own
              float text and "\\/" escapes, so the BYTES of that file differ.

RunResult is a read-only Mapping, not a dict: callers that json.dump() it or assign into res[qid] need res.to_dict() first (the
reference's own callers only read it, write it or hand it to the metrics; utils/metrics.py accepts both).
"""
import ctypes
import json
from collections.abc import Mapping

import numpy as np

from .. import _lib


def to_host(t):
    """Device tensor -> numpy array through a PINNED staging buffer of torch's caching host allocator: the 28 + 56 MB result
    arrays of a Dev-sized search come down at link speed, and a repeated search reuses the buffer instead of faulting in a fresh
    pageable allocation page by page (first-touch page faults of a result-sized buffer cost more than the copy itself)."""
    import torch
    if t.device.type == "cpu":
        return t.numpy()
    h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    h.copy_(t, non_blocking=True)
    torch.cuda.current_stream(t.device).synchronize()
    return h.numpy()


def _escaped_blob(keys):
    """[str] -> (bytes, int64 offsets [n + 1]): the body of json.dumps(key) for every key."""
    parts = [json.dumps(str(k))[1:-1].encode("ascii") for k in keys]
    off = np.zeros(len(parts) + 1, dtype=np.int64)
    if parts:
        np.cumsum([len(p) for p in parts], out=off[1:])
    return b"".join(parts), off


class IdTable:
    """Keys of one side of run.json (query ids, or the document-id table indexed by result position), in the cheapest typed
    form that reproduces str(key):
      "i64"    every key is an int (or numpy integer): decimal text;
      "fixed"  ASCII strings without a character JSON escapes: fixed-width NUL-padded bytes (a numpy 'S' array);
      "blob"   anything else: json.dumps bodies + offsets.
    `distinct` (checked once) says whether str(key) is unique - the fast writer needs it (a dict would merge duplicates)."""

    def __init__(self, keys):
        self.n = len(keys)
        self.i64 = self.fixed = self.blob = self.off = None
        self.width = 0
        arr = keys if isinstance(keys, np.ndarray) else None
        if arr is None:
            if self.n and all(isinstance(k, (int, np.integer)) and not isinstance(k, bool) for k in keys):
                arr = np.asarray(keys, dtype=np.int64)
            else:
                arr = np.asarray([str(k) for k in keys]) if self.n else np.zeros(0, dtype=np.int64)
        if arr.dtype.kind in "iu":
            self.kind, self.i64 = "i64", np.ascontiguousarray(arr, dtype=np.int64)
        else:
            if arr.dtype.kind == "O":
                arr = np.asarray([str(k) for k in arr])
            fixed = None
            if arr.dtype.kind in "US":
                try:
                    fixed = np.ascontiguousarray(arr.astype("S") if arr.dtype.kind == "U" else arr)
                    v = fixed.view(np.uint8)
                    if v.size and bool((((v < 32) & (v != 0)) | (v == 34) | (v == 92) | (v > 126)).any()):
                        fixed = None                       # a character json.dumps would escape
                    elif arr.dtype.kind == "U" and bool((np.char.str_len(arr) != np.char.str_len(fixed)).any()):
                        fixed = None                       # an embedded NUL
                except UnicodeEncodeError:
                    fixed = None
            if fixed is not None and fixed.dtype.itemsize > 0:
                self.kind, self.fixed, self.width = "fixed", fixed, fixed.dtype.itemsize
            else:
                self.kind = "blob"
                self.blob, self.off = _escaped_blob(arr.tolist())
        self._distinct = None
        self._keys = keys

    @property
    def distinct(self):
        if self._distinct is None:
            if self.kind == "i64":
                self._distinct = np.unique(self.i64).size == self.n
            elif self.kind == "fixed":
                self._distinct = np.unique(self.fixed).size == self.n
            else:
                self._distinct = len(set(self.key(i) for i in range(self.n))) == self.n
        return self._distinct

    def key(self, i):
        """str(key i), as the reference's str(qid) / str(docid)."""
        if self.kind == "i64":
            return str(int(self.i64[i]))
        if self.kind == "fixed":
            return self.fixed[i].decode("ascii")
        return json.loads(b'"' + self.blob[self.off[i]:self.off[i + 1]] + b'"')

    def keys_of(self, positions):
        """[str] for an int array of positions."""
        if self.kind == "i64":
            return [str(v) for v in self.i64[positions].tolist()]
        if self.kind == "fixed":
            return [b.decode("ascii") for b in self.fixed[positions].tolist()]
        return [self.key(int(p)) for p in positions]

    def c_args(self):
        """(i64 pointer, bytes pointer, offsets pointer, fixed width) for sr_write_run_json."""
        if self.kind == "i64":
            return self.i64.ctypes.data, None, None, 0
        if self.kind == "fixed":
            return None, self.fixed.ctypes.data, None, self.width
        return None, self.blob, self.off.ctypes.data, 0


def id_table(keys):
    return keys if isinstance(keys, IdTable) else IdTable(keys)


def write_run_json(path, qids, scores, positions, doc_table, counts=None, n_threads=0, part=0):
    """Write {str(qid): {str(doc id): float(score)}} for scores fp32 [nq, k] / positions int64 [nq, k] (rows of `doc_table`; negative =
    padding) / counts [nq] (hits per row, optional).  Returns the file size.  Falls back to json.dump of the materialised dict
    when keys repeat (a dict merges them; the array writer would not).
    part: 0 = the whole file; 1 / 2 / 3 = first / middle / last piece of a file written piece by piece (sr_write_run_json_part: the
    pieces' qids must be distinct across pieces too - the caller's concern - and the finished file equals one call over all rows)."""
    scores = np.ascontiguousarray(scores, dtype=np.float32)
    positions = np.ascontiguousarray(positions, dtype=np.int64)
    nq, k = scores.shape if scores.ndim == 2 else (0, 0)
    assert positions.shape == scores.shape
    qt, dt = id_table(qids), id_table(doc_table)
    assert qt.n == nq, (qt.n, nq)
    if counts is not None:
        counts = np.ascontiguousarray(counts, dtype=np.int32)
    if not (qt.distinct and dt.distinct):
        if part:
            raise ValueError("write_run_json: a file written in pieces needs distinct query and document keys")
        with open(path, "w") as f:
            json.dump(RunResult(qt, scores, positions, dt, counts).to_dict(), f)
        import os
        return os.path.getsize(path)
    lib = _lib.load()
    qa, da = qt.c_args(), dt.c_args()
    nbytes = ctypes.c_int64(0)
    if part:
        _lib.check(lib.sr_write_run_json_part(str(path).encode(), int(part), nq, k, scores.ctypes.data, positions.ctypes.data,
                                              counts.ctypes.data if counts is not None else None,
                                              qa[0], qa[1], qa[2], qa[3], da[0], da[1], da[2], da[3], dt.n, int(n_threads), ctypes.byref(nbytes)),
                   "sr_write_run_json_part")
        return nbytes.value
    _lib.check(lib.sr_write_run_json(str(path).encode(), nq, k, scores.ctypes.data, positions.ctypes.data,
                                     counts.ctypes.data if counts is not None else None,
                                     qa[0], qa[1], qa[2], qa[3], da[0], da[1], da[2], da[3], dt.n, int(n_threads), ctypes.byref(nbytes)),
               "sr_write_run_json")
    return nbytes.value


class PiecewiseRunWriter:
    """run.json written piece by piece on ONE worker thread (the pieces reach the file in order) while the caller encodes and searches the
    next piece.  The pieces go to `path + ".tmp"`; the file takes its name only when the last piece is down (os.replace), so a failure in
    a later piece - in the caller's encode / search or in the writer - leaves no half-written run.json behind (a reader of the one-shot
    path never saw one either).  add() re-raises the exception of an earlier piece's write instead of computing on for nothing."""

    def __init__(self, path):
        import os
        from concurrent.futures import ThreadPoolExecutor
        self.path, self.tmp = str(path), str(path) + ".tmp"
        self._os = os
        self._pool = ThreadPoolExecutor(max_workers=1)
        self._writes = []
        self._closed = False

    def _check(self):
        for w in self._writes:
            if w.done() and w.exception() is not None:
                raise w.exception()

    def add(self, qids, scores, positions, doc_table, counts=None, last=False):
        self._check()
        part = 3 if last else (1 if not self._writes else 2)
        if last and not self._writes:
            part = 0                                            # a single piece: the one-call writer
        self._writes.append(self._pool.submit(write_run_json, self.tmp, qids, scores, positions, doc_table, counts, 0, part))
        self._closed = self._closed or last

    def finish(self):
        """Wait for every piece; returns the file size.  Raises what a write raised."""
        size = 0
        for w in self._writes:
            size = w.result()
        if not self._closed:
            raise RuntimeError("PiecewiseRunWriter.finish: the last piece was never added")
        self._os.replace(self.tmp, self.path)
        return size

    def __enter__(self):
        return self

    def __exit__(self, exc_type, exc, tb):
        self._pool.shutdown(wait=True)
        if self._os.path.exists(self.tmp):                      # finish() was not reached, or a write failed
            try:
                self._os.remove(self.tmp)
            except OSError:
                pass
        return False


class RunResult(Mapping):
    """The `res` / `qid_to_rankdata` dict of the reference's retrieval drivers, backed by the result arrays.
    res[str(qid)] -> {str(doc id): float(score)} in rank order (built on access); queries without a hit have no entry, as in the
    reference; `res == some_dict` compares contents; `.to_dict()` materialises everything; `.dump(path)` writes run.json."""

    def __init__(self, qids, scores, positions, doc_table, counts=None):
        self.qids, self.docs = id_table(qids), id_table(doc_table)
        self.scores = np.ascontiguousarray(scores, dtype=np.float32)
        self.positions = np.ascontiguousarray(positions, dtype=np.int64)
        self.counts = None if counts is None else np.ascontiguousarray(counts, dtype=np.int32)
        self._row_of = None                     # built on first access: a caller that only wanted the files pays nothing for the mapping

    def _keep(self, r):
        keep = self.positions[r] >= 0
        if self.counts is not None:
            keep &= np.arange(keep.shape[0]) < self.counts[r]
        return keep

    def _rows(self):
        if self._row_of is None:
            k = self.scores.shape[1] if self.scores.ndim == 2 else 0
            valid = self.positions >= 0
            if self.counts is not None:
                valid &= np.arange(k)[None, :] < self.counts[:, None]
            has_hit = valid.any(1) if valid.size else np.zeros(len(self.scores), dtype=bool)
            rows = {}
            for r in np.nonzero(has_hit)[0].tolist():
                rows.setdefault(self.qids.key(r), []).append(r)
            self._row_of = rows
        return self._row_of

    def _row_dict(self, r):
        keep = self._keep(r)
        return dict(zip(self.docs.keys_of(self.positions[r][keep]), self.scores[r][keep].astype(np.float64).tolist()))

    def __getitem__(self, qid):
        rows = self._rows()[qid]
        out = self._row_dict(rows[0])
        for r in rows[1:]:                     # a repeated qid: later rows update the dict, as the reference's loops do
            out.update(self._row_dict(r))
        return out

    def __iter__(self):
        return iter(self._rows())

    def __len__(self):
        return len(self._rows())

    def __contains__(self, qid):
        return qid in self._rows()

    def to_dict(self):
        return {q: self[q] for q in self}

    def dump(self, path, n_threads=0):
        return write_run_json(path, self.qids, self.scores, self.positions, self.docs, self.counts, n_threads=n_threads)
