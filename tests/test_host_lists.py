"""CPU: csrc/host_lists.c - the list-of-lists of db ids of DenseFlatIndexer.search_knn
(/root/reference/scaling_retriever/indexer.py:210-214) - against the reference's comprehension."""
import sys

import numpy as np
import pytest

from scaling_retriever_amd import _host_lists as HL


def _table(ids):
    t = np.empty(len(ids) + 1, dtype=object)
    t[:len(ids)] = ids
    t[len(ids)] = None
    return t


@pytest.mark.parametrize("n_docs,nq,k,kind", [(1000, 7, 13, "str"), (300_000, 300, 1000, "str"), (50_000, 513, 100, "int"), (5, 3, 9, "str")])
def test_take_rows_equals_the_reference_comprehension(n_docs, nq, k, kind):
    ids = [f"D{7 * i}" for i in range(n_docs)] if kind == "str" else [7 * i + 1_000_000 for i in range(n_docs)]
    index_id_to_db_id = ids
    table = _table(ids)
    rng = np.random.default_rng(n_docs + nq)
    idx = rng.integers(0, n_docs, size=(nq, k)).astype(np.int64)
    idx[rng.random((nq, k)) < 0.01] = -1                      # faiss pads with label -1 (fewer than k vectors)
    probe = ids[int(idx[idx >= 0][0])]
    before = sys.getrefcount(probe)
    out = HL.take_rows(table.ctypes.data, n_docs, idx.ctypes.data, nq, k)
    ref = [[index_id_to_db_id[i] if i >= 0 else None for i in row] for row in idx]          # indexer.py:212-213
    assert out == ref and all(type(r) is list for r in out)
    assert all(a is b for a, b in zip(out[0], ref[0]))        # the index's own objects, not copies
    del out, ref
    assert sys.getrefcount(probe) == before                   # every reference the lists held was counted once


def test_take_rows_edges_and_errors():
    table = _table(["a", "b", "c"])
    idx = np.array([[2, 0, -1], [1, 1, 2]], dtype=np.int64)
    assert HL.take_rows(table.ctypes.data, 3, idx.ctypes.data, 2, 3) == [["c", "a", None], ["b", "b", "c"]]
    assert HL.take_rows(table.ctypes.data, 3, idx.ctypes.data, 0, 3) == []
    assert HL.take_rows(table.ctypes.data, 3, idx.ctypes.data, 2, 0) == [[], []]
    for bad in (3, -2):
        b = np.array([[bad]], dtype=np.int64)
        with pytest.raises(IndexError):
            HL.take_rows(table.ctypes.data, 3, b.ctypes.data, 1, 1)
    with pytest.raises(ValueError):
        HL.take_rows(table.ctypes.data, -1, idx.ctypes.data, 1, 1)


def test_concurrent_calls_keep_their_results():
    """Two threads inside take_rows at once (the sort runs without the GIL): the cached scratch goes to one, the other gets its own."""
    from concurrent.futures import ThreadPoolExecutor
    n_docs = 400_000
    ids = [str(i) for i in range(n_docs)]
    table = _table(ids)
    rng = np.random.default_rng(3)
    idxs = [rng.integers(0, n_docs, size=(600, 1000)).astype(np.int64) for _ in range(4)]
    with ThreadPoolExecutor(max_workers=4) as pool:
        outs = list(pool.map(lambda ix: HL.take_rows(table.ctypes.data, n_docs, ix.ctypes.data, ix.shape[0], ix.shape[1]), idxs))
    for ix, out in zip(idxs, outs):
        assert out == [[ids[i] for i in row] for row in ix]
