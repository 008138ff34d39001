"""The reference's on-disk inverted index is one h5py file of per-term datasets
(/root/reference/scaling_retriever/utils/inverted_index.py:22-55 read, :84-105 write: `dim`, `index_doc_id_{t}`,
`index_doc_value_{t}`).  IndexDictOfArray reads and writes that layout when h5py is importable (it is NOT in this image, so
the test skips here and the CSR .npz twin is what the rest of the suite exercises)."""
import os
import pickle

import numpy as np
import pytest

h5py = pytest.importorskip("h5py")


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
