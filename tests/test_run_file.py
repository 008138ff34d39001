"""CPU: run.json written from the result arrays (sr_write_run_json, a host function of the C ABI) is byte for byte what the
reference's per-hit loops + json.dump write (/root/reference/eval_dense.py:225-241; scaling_retriever/indexer.py:431-432,536-537),
and RunResult behaves like the nested dict those loops build."""
import json
import os

import numpy as np
import pytest

from scaling_retriever_amd.utils.run_file import IdTable, RunResult, write_run_json


def _reference_dict(qids, scores, positions, doc_ids, counts=None):
    """The reference's loop, literally (eval_dense.py:225-234 with the padding rows of a short index dropped)."""
    run = {}
    for r, qid in enumerate(qids):
        n = scores.shape[1] if counts is None else int(counts[r])
        for j in range(n):
            if positions[r, j] < 0:
                continue
            qid_s, docid, score = str(qid), str(doc_ids[positions[r, j]]), float(scores[r, j])
            if qid_s not in run:
                run[qid_s] = {docid: score}
            else:
                run[qid_s][docid] = score
    return run


def _case(rng, nq, k, N):
    scores = rng.standard_normal((nq, k)).astype(np.float32)
    positions = np.stack([rng.permutation(N)[:k] for _ in range(nq)]).astype(np.int64)
    return scores, positions


SPECIAL = [0.0, -0.0, 1e-5, 1.5e-5, 9.9999e-5, 1e-4, 0.0001234, 1e15, 9.99e15, 1e16, 1.2345e16, 123456.0, 1.0, 100000.0, 3.4028235e38,
           1e-45, 1.17549435e-38, np.inf, -np.inf, np.nan, 16777216.0, 0.1, 2.5e-7, -3.3333333, 0.3, 7.0e22]


@pytest.mark.parametrize("threads", [1, 3])
def test_int_keys_byte_identical_to_json_dump(tmp_path, threads):
    rng = np.random.default_rng(0)
    nq, k, N = 700, 50, 5000
    scores, positions = _case(rng, nq, k, N)
    scores[0, :len(SPECIAL)] = np.array(SPECIAL, dtype=np.float32)
    positions[3, 20:] = -1                                   # fewer than k documents (faiss label -1)
    positions[9, :] = -1                                     # a query without any hit: no entry
    qids = list(range(1000, 1000 + nq))
    docs = rng.permutation(10 ** 7)[:N].astype(np.int64)
    p = tmp_path / "run.json"
    n = write_run_json(p, qids, scores, positions, docs, n_threads=threads)
    want = json.dumps(_reference_dict(qids, scores, positions, docs))
    assert p.read_text() == want and n == len(want)
    assert "1009" not in json.loads(want)


@pytest.mark.parametrize("threads", [1, 2, 64])
def test_several_rounds_of_slabs_and_empty_leading_queries(tmp_path, threads):
    """More queries than one round of slabs holds (a thread's slab is capped at 1 024 queries), the first queries - a whole first
    slab with 64 threads - without any hit: the file's first entry must come without a separator wherever it is."""
    rng = np.random.default_rng(4)
    nq, k, N = 2300, 4, 900
    scores, positions = _case(rng, nq, k, N)
    positions[:40, :] = -1
    positions[1500:1510, :] = -1
    qids = [str(10 * i) for i in range(nq)]
    docs = np.arange(N, dtype=np.int64) * 3
    p = tmp_path / "r.json"
    n = write_run_json(p, qids, scores, positions, docs, n_threads=threads)
    want = json.dumps(_reference_dict(qids, scores, positions, docs))
    assert p.read_text() == want and n == len(want)
    # an existing longer file is replaced, not overwritten in place
    p.write_text("x" * (2 * len(want)))
    write_run_json(p, qids, scores, positions, docs, n_threads=threads)
    assert p.read_text() == want


def test_large_result_written_through_the_mapped_path(tmp_path):
    """Rounds of 8 MB and more are copied into a shared mapping of the file by the formatting threads (buffered writes to one file
    serialise on its inode): same bytes, and the file ends where the text ends."""
    rng = np.random.default_rng(5)
    nq, k, N = 2500, 200, 200_000
    scores = rng.standard_normal((nq, k)).astype(np.float32)
    positions = rng.integers(0, N, size=(nq, k)).astype(np.int64)
    for r in range(nq):                                        # the reference's dict keeps the LAST score of a repeated docid
        positions[r] = rng.permutation(N)[:k]
    positions[17, 150:] = -1
    qids = [str(3 * i) for i in range(nq)]
    docs = np.arange(N).astype("U7")
    p = tmp_path / "big.json"
    from scaling_retriever_amd import _lib
    before = _lib.load().sr_run_writer_mapped_rounds()
    n = write_run_json(p, qids, scores, positions, docs, n_threads=8)
    assert _lib.load().sr_run_writer_mapped_rounds() == before + 1       # the mapping was taken, not the pwrite fallback
    want = json.dumps(_reference_dict(qids, scores, positions, docs))
    assert n == len(want) >= (8 << 20) and os.path.getsize(p) == n
    assert p.read_text() == want
    # one thread: pwrite, no mapping, same bytes
    n1 = write_run_json(tmp_path / "big1.json", qids, scores, positions, docs, n_threads=1)
    assert _lib.load().sr_run_writer_mapped_rounds() == before + 1 and n1 == n
    assert (tmp_path / "big1.json").read_text() == want


def test_counts_string_keys_and_escapes(tmp_path):
    rng = np.random.default_rng(1)
    nq, k, N = 64, 30, 400
    scores, positions = _case(rng, nq, k, N)
    counts = rng.integers(0, k + 1, nq).astype(np.int32)
    counts[0] = 0
    # MS MARCO shape: ids are decimal STRINGS (dataset.py keeps them as read), stored by store_embs as a numpy unicode array
    qids = [str(5 + 3 * i) for i in range(nq)]
    docs = np.array([str(i * 7) for i in range(N)])
    assert IdTable(docs).kind == "fixed" and IdTable(qids).kind == "fixed"
    p = tmp_path / "a.json"
    write_run_json(p, qids, scores, positions, docs, counts)
    assert p.read_text() == json.dumps(_reference_dict(qids, scores, positions, docs, counts))
    # BEIR shape: free-form string ids, some need escaping / are not ASCII
    docs2 = ['doc "%d"\\x' % i if i % 5 == 0 else "dé-%d" % i if i % 7 == 0 else "D%d\t" % i for i in range(N)]
    qids2 = ["q中%d" % i for i in range(nq)]
    assert IdTable(docs2).kind == "blob"
    p2 = tmp_path / "b.json"
    write_run_json(p2, qids2, scores, positions, docs2, counts)
    assert p2.read_text() == json.dumps(_reference_dict(qids2, scores, positions, docs2, counts))
    assert json.loads(p2.read_text()) == _reference_dict(qids2, scores, positions, docs2, counts)


def test_duplicate_keys_fall_back_to_dict_semantics(tmp_path):
    rng = np.random.default_rng(2)
    scores, positions = _case(rng, 6, 8, 40)
    qids = [1, 2, 2, 3, 4, 4]                               # repeated qids merge, later rows win (reference loop semantics)
    docs = np.arange(40) // 2                               # and so do repeated document ids
    p = tmp_path / "run.json"
    write_run_json(p, qids, scores, positions, docs)
    want = _reference_dict(qids, scores, positions, docs)
    assert json.loads(p.read_text()) == want
    assert p.read_text() == json.dumps(want)


def test_run_result_is_the_reference_dict():
    rng = np.random.default_rng(3)
    nq, k, N = 40, 25, 300
    scores, positions = _case(rng, nq, k, N)
    positions[5, 10:] = -1
    positions[7, :] = -1
    qids = [str(100 + i) for i in range(nq)]
    docs = np.array([str(9000 + i) for i in range(N)])
    want = _reference_dict(qids, scores, positions, docs)
    res = RunResult(qids, scores, positions, docs)
    assert len(res) == len(want) == nq - 1 and "107" not in res and list(res) == list(want)
    assert res == want and dict(res) == want and res.to_dict() == want
    assert res["105"] == want["105"] and list(res["100"].items()) == list(want["100"].items())     # rank order kept
    assert res["100"][str(docs[positions[0, 0]])] == float(scores[0, 0])
    with pytest.raises(KeyError):
        res["107"]


def test_empty_result(tmp_path):
    p = tmp_path / "run.json"
    write_run_json(p, [], np.zeros((0, 10), np.float32), np.zeros((0, 10), np.int64), np.arange(5))
    assert p.read_text() == "{}"


@pytest.mark.parametrize("cuts", [(0, 700), (300, 301), (0, 0), (1000, 1000)])
def test_file_written_in_pieces_equals_the_whole(tmp_path, cuts):
    """sr_write_run_json_part: first / middle / last pieces (the host writes piece c while the GPU searches piece c + 1) give the bytes
    of one call over all rows - also when a piece is empty or holds only queries without hits."""
    rng = np.random.default_rng(11)
    nq, k, N = 1000, 37, 5000
    scores, positions = _case(rng, nq, k, N)
    positions[:40] = -1                                         # the first 40 queries have no hit: the file's first entry comes later
    qids = [str(10 * i + 3) for i in range(nq)]
    docs = np.arange(N) * 3
    whole, parts = tmp_path / "whole.json", tmp_path / "parts.json"
    n = write_run_json(whole, qids, scores, positions, docs, n_threads=3)
    a, b = cuts
    bounds = [(0, a), (a, b), (b, nq)]
    size = 0
    for c, (r0, r1) in enumerate(bounds):
        size = write_run_json(parts, qids[r0:r1], scores[r0:r1], positions[r0:r1], docs, n_threads=3, part=c + 1)
    assert size == n and parts.read_bytes() == whole.read_bytes()
    assert json.loads(parts.read_text()) == _reference_dict(qids, scores, positions, docs)


def test_piecewise_writer_publishes_run_json_only_when_complete(tmp_path):
    """PiecewiseRunWriter (SparseRetrieval.retrieve, eval_dense write_run): the pieces go to run.json.tmp and the file takes its name after
    the last piece - the bytes of one call; a failure between pieces (the caller's next encode / search, or a write) leaves the
    previous run.json untouched and no .tmp behind, and a failed write is re-raised at the next add()."""
    from scaling_retriever_amd.utils.run_file import PiecewiseRunWriter
    rng = np.random.default_rng(5)
    nq, k, N = 600, 20, 3000
    scores, positions = _case(rng, nq, k, N)
    qids = [str(i) for i in range(nq)]
    docs = np.arange(N)
    whole, path = tmp_path / "whole.json", tmp_path / "run.json"
    write_run_json(whole, qids, scores, positions, docs)
    with PiecewiseRunWriter(path) as w:
        for c, (r0, r1) in enumerate([(0, 256), (256, 512), (512, nq)]):
            w.add(qids[r0:r1], scores[r0:r1], positions[r0:r1], docs, last=c == 2)
            assert not path.exists()                                   # nothing under the final name before the end
        size = w.finish()
    assert path.read_bytes() == whole.read_bytes() and size == path.stat().st_size and not (tmp_path / "run.json.tmp").exists()
    with PiecewiseRunWriter(tmp_path / "single.json") as w:            # one piece: the one-call writer
        w.add(qids, scores, positions, docs, last=True)
        w.finish()
    assert (tmp_path / "single.json").read_bytes() == whole.read_bytes()
    before = path.read_bytes()
    with pytest.raises(RuntimeError, match="search failed"):           # the caller fails after the first piece
        with PiecewiseRunWriter(path) as w:
            w.add(qids[:256], scores[:256], positions[:256], docs)
            raise RuntimeError("search failed")
    assert path.read_bytes() == before and not (tmp_path / "run.json.tmp").exists()
    with pytest.raises(AssertionError):                                # a write fails (row count mismatch): re-raised, not swallowed
        with PiecewiseRunWriter(path) as w:
            w.add(qids[:256], scores[:255], positions[:255], docs)
            import time
            for _ in range(200):
                if w._writes[0].done():
                    break
                time.sleep(0.01)
            w.add(qids[256:], scores[256:], positions[256:], docs, last=True)
    assert path.read_bytes() == before and not (tmp_path / "run.json.tmp").exists()
