"""Small GPU regressions: the drop-in static scorer's index cache, unknown query terms, shard-file ingest, the opt-in
README known-answer test."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dicts(rng, V, N, density):
    ids, vals = {}, {}
    for t in range(V):
        nz = np.nonzero(rng.random(N) < density)[0].astype(np.int32)
        ids[t], vals[t] = nz, (rng.random(len(nz), dtype=np.float32) + 0.1)
    return ids, vals


def test_static_numba_score_float_never_scores_a_stale_index():
    """SparseRetrieval.numba_score_float caches the device index of the caller's dict; the cache holds the dict itself, so
    a NEW dict that happens to reuse the address (id) of a dropped one cannot hit it."""
    import gc
    from oracle import scoring as SC
    from scaling_retriever_amd.indexer import SparseRetrieval
    rng = np.random.default_rng(0)
    cols, qv = np.array([1, 4, 7], np.int32), np.array([0.5, 1.5, 2.0], np.float32)
    for _ in range(4):                             # fresh dicts each round; CPython readily reuses the freed address
        ids, vals = _dicts(rng, 10, 500, 0.1)
        got_i, got_s = SparseRetrieval.numba_score_float(ids, vals, cols, qv, 0.0, 500)
        ptr = np.concatenate([[0], np.cumsum([len(ids[t]) for t in range(10)])]).astype(np.int64)
        ref_i, ref_s = SC.numba_score_float(ptr, np.concatenate([ids[t] for t in range(10)]),
                                            np.concatenate([vals[t] for t in range(10)]), cols, qv, 0.0, 500)
        assert np.array_equal(got_i, ref_i) and np.array_equal(got_s, ref_s)
        assert SparseRetrieval._static_cache[0] is ids
        del ids, vals
        gc.collect()


def test_unknown_query_terms_are_empty_posting_lists():
    from scaling_retriever_amd.scoring import SparseIndexHIP
    rng = np.random.default_rng(1)
    ids, vals = _dicts(rng, 20, 3000, 0.05)
    ptr = np.concatenate([[0], np.cumsum([len(ids[t]) for t in range(20)])]).astype(np.int64)
    idx = SparseIndexHIP(ptr, np.concatenate(list(ids.values())), np.concatenate(list(vals.values())), 3000)
    q = np.array([2, 5, 11], np.int32)
    w = np.array([1.0, 2.0, 0.5], np.float32)
    s0, i0, c0 = idx.search(np.array([0, 3], np.int64), q, w, 10)
    # the same query with term ids outside [0, n_terms) mixed in (the reference's dict holds an empty array for every id)
    q2 = np.array([-3, 2, 5, 11, 20, 999999], np.int32)
    w2 = np.array([9.0, 1.0, 2.0, 0.5, 7.0, 3.0], np.float32)
    s1, i1, c1 = idx.search(np.array([0, 6], np.int64), q2, w2, 10)
    assert torch.equal(s0, s1) and torch.equal(i0, i1) and torch.equal(c0, c1)


def test_shard_file_ingest_mmap_pinned_ring(tmp_path):
    from oracle import scoring as SC
    from scaling_retriever_amd.scoring import DenseIndexHIP
    rng = np.random.default_rng(2)
    rows = rng.standard_normal((70001, 128), dtype=np.float32)
    np.save(tmp_path / "embs_0_0.npy", rows)
    a, b = DenseIndexHIP(128), DenseIndexHIP(128)
    a.add_npy_file(str(tmp_path / "embs_0_0.npy"))                                     # memory-mapped, 64 MB pieces
    b.add_host_rows(rows, piece_bytes=1 << 20, n_buffers=3, n_threads=5)              # many small pieces, ring reuse
    assert a.ntotal == b.ntotal == 70001
    assert torch.equal(a._segments[0], torch.from_numpy(rows).cuda()) and torch.equal(b._segments[0], a._segments[0])
    Q = rng.standard_normal((3, 128), dtype=np.float32)
    s, i = a.search(torch.from_numpy(Q).cuda(), 5)
    es, ei = SC.topk_rows(SC.dense_scores_fma(Q, rows, SC.dense_korder(3, 128)), 5)
    assert np.array_equal(i.cpu().numpy(), ei) and np.array_equal(s.cpu().numpy(), es)
    with pytest.raises(ValueError):
        a.add_host_rows(rows[:, :64])


def test_readme_known_answers_when_checkpoints_are_present():
    """/root/reference/README.md:56-66 - the only real-weights pin the reference offers.  Opt-in: point SR_LION_SP_1B and
    SR_LION_DS_1B at local copies of hzeng/Lion-SP-1B-llama3-marco-mntp and hzeng/Lion-DS-1B-llama3-marco-mntp (adapter +
    tokenizer; adapter_config.json's base_model_name_or_path must resolve locally).  There is no network here, so the test
    skips unless they exist."""
    sp, ds = os.environ.get("SR_LION_SP_1B"), os.environ.get("SR_LION_DS_1B")
    if not (sp and ds and os.path.isdir(sp) and os.path.isdir(ds)):
        pytest.skip("set SR_LION_SP_1B / SR_LION_DS_1B to local checkpoint directories")
    from transformers import AutoTokenizer
    from scaling_retriever_amd.modeling.llm_encoder import LlamaBiDense, LlamaBiSparse
    queries = ["What is the capital of France?", "Who wrote '1984'?"]
    passages = ["Paris is the capital of France.", "George Orwell wrote '1984'."]
    expect = {"sparse": [[14.8352, 0.0264], [0.0055, 13.9098]], "dense": [[0.2878, 0.1321], [0.1041, 0.2922]]}
    for kind, path, cls in (("sparse", sp, LlamaBiSparse), ("dense", ds, LlamaBiDense)):
        model = cls.load_from_lora(path).to("cuda").eval()
        tok = AutoTokenizer.from_pretrained(path)
        tok.padding_side = "left"                                              # examples/quick_start.py:9
        tq = tok(queries, max_length=192, truncation=True, padding="longest", return_tensors="pt")
        tp = tok(passages, max_length=192, truncation=True, padding="longest", return_tensors="pt")
        with torch.no_grad():                                                  # quick_start.py runs fp32, no autocast
            q = model.query_encode(**{k: v.cuda() for k, v in tq.items()})
            p = model.doc_encode(**{k: v.cuda() for k, v in tp.items()})
        scores = torch.matmul(q, p.T).cpu().numpy()
        print(kind, scores.tolist())
        tol = 1e-3 * max(1.0, float(np.abs(expect[kind]).max()))              # the README prints 4 decimals
        assert np.allclose(scores, np.array(expect[kind]), atol=tol), (kind, scores)
