"""The reference's on-disk inverted index is one h5py file of per-term datasets
(/root/reference/scaling_retriever/utils/inverted_index.py:22-55 read, :84-105 write: `dim`, `index_doc_id_{t}`,
`index_doc_value_{t}`).  IndexDictOfArray reads and writes that layout when h5py is importable.  It is NOT in this image: the test
then runs against tests/h5py_double.py (the five calls the reference makes, datasets kept in a pickle) so that the h5py branches
execute at all; with the real library installed the same test uses it.  The CSR .npz twin is what the rest of the suite exercises."""
import os
import pickle
import sys

import numpy as np
import pytest

try:
    import h5py
    REAL_H5PY = True
except Exception:
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import h5py_double as h5py
    REAL_H5PY = False


@pytest.fixture(autouse=True)
def _h5py_available(monkeypatch):
    """Without the real library: make `import h5py` inside inverted_index resolve to the double for this module's tests."""
    from scaling_retriever_amd.utils import inverted_index
    if not REAL_H5PY:
        monkeypatch.setitem(sys.modules, "h5py", h5py)
        monkeypatch.setattr(inverted_index, "HAVE_H5PY", True)
    yield


def test_reads_a_reference_layout_file_and_round_trips(tmp_path):
    from scaling_retriever_amd.utils.inverted_index import IndexDictOfArray
    d = str(tmp_path / "index")
    os.makedirs(d)
    postings = {0: ([0, 3, 4], [0.5, 1.5, 2.0]), 2: ([1], [3.0]), 5: ([0, 4], [0.25, 4.0])}
    with h5py.File(os.path.join(d, "array_index.h5py"), "w") as f:          # hand-built, exactly as the reference writes it
        f.create_dataset("dim", data=6)
        for t, (ids, vals) in postings.items():
            f.create_dataset(f"index_doc_id_{t}", data=np.array(ids, np.int32))
            f.create_dataset(f"index_doc_value_{t}", data=np.array(vals, np.float32))
    pickle.dump({i: f"p{i}" for i in range(5)}, open(os.path.join(d, "doc_ids.pkl"), "wb"))
    idx = IndexDictOfArray(d, dim_voc=6)
    assert idx.nb_docs() == 5 and sorted(idx.index_doc_id.keys()) == [0, 2, 5]
    for t, (ids, vals) in postings.items():
        assert idx.index_doc_id[t].tolist() == ids and idx.index_doc_value[t].tolist() == vals
    indptr, ids, vals = idx.csr(6)
    assert indptr.tolist() == [0, 3, 3, 4, 4, 4, 6]
    d2 = str(tmp_path / "index2")
    out = IndexDictOfArray(d2, dim_voc=6, force_new=True)
    out.set_csr(indptr, ids, vals, 5)
    out.save(dim=6)
    with h5py.File(os.path.join(d2, "array_index.h5py"), "r") as f:
        assert int(f["dim"][()]) == 6 and sorted(k for k in f.keys() if k.startswith("index_doc_id_")) == \
            ["index_doc_id_0", "index_doc_id_2", "index_doc_id_5"]
        assert np.array(f["index_doc_value_5"]).tolist() == [0.25, 4.0]


def test_sparse_index_directory_in_the_reference_layout_is_searchable(tmp_path):
    """save() -> a fresh IndexDictOfArray over the same directory -> the same CSR: what SparseRetrieval(index_dir=...) loads when
    the index was written by the reference (inverted_index.py:84-105) or by SparseIndexer here with h5py present."""
    from scaling_retriever_amd.utils.inverted_index import IndexDictOfArray
    rng = np.random.default_rng(0)
    V, N = 40, 30
    dense = (rng.random((N, V)) < 0.2) * rng.random((N, V)).astype(np.float32)
    rows, cols = np.nonzero(dense)
    d = str(tmp_path / "index")
    idx = IndexDictOfArray(d, dim_voc=V, force_new=True)
    idx.add_batch_document(rows.astype(np.int64), cols.astype(np.int32), dense[rows, cols].astype(np.float32), n_docs=N)
    idx.save(dim=V)
    assert os.path.exists(os.path.join(d, "array_index.h5py")) and not os.path.exists(os.path.join(d, "array_index.npz"))
    pickle.dump([f"p{i}" for i in range(N)], open(os.path.join(d, "doc_ids.pkl"), "wb"))
    again = IndexDictOfArray(d, dim_voc=V)
    a, b = idx.csr(V), again.csr(V)
    assert all(np.array_equal(x, y) for x, y in zip(a, b)) and again.nb_docs() == N
    for t in range(V):                                           # posting order inside a term = ascending doc (insertion order)
        ids = again.index_doc_id[t]
        assert np.array_equal(ids, np.nonzero(dense[:, t])[0]) and np.array_equal(again.index_doc_value[t], dense[ids, t])
