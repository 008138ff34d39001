"""Property tests (hypothesis) of the oracle and the host logic on small random inputs - the pins SURVEY.md 8c asks for where
the reference has no tests of its own: the restated scorers against brute force, the C ports against the numpy
restatements, the metrics against their definitions."""
import numpy as np
from hypothesis import given, settings, strategies as st

from oracle import scoring as SC

SET = settings(max_examples=40, deadline=None)


def _random_csr(rng, V, N, density):
    lists = []
    for _ in range(V):
        n = rng.binomial(N, density)
        ids = np.sort(rng.choice(N, size=n, replace=False)).astype(np.int32)
        lists.append(ids)
    indptr = np.concatenate([[0], np.cumsum([len(x) for x in lists])]).astype(np.int64)
    doc_ids = np.concatenate(lists).astype(np.int32) if lists else np.zeros(0, np.int32)
    vals = rng.random(len(doc_ids), dtype=np.float32) * 3
    return indptr, doc_ids, vals


@SET
@given(st.integers(0, 2 ** 31 - 1), st.integers(1, 40), st.integers(1, 300), st.floats(0.0, 0.6), st.integers(1, 12),
       st.integers(1, 30), st.sampled_from([0.0, 0.5, -1.0]))
def test_sparse_oracle_against_dense_brute_force(seed, V, N, density, nterms, k, threshold):
    rng = np.random.default_rng(seed)
    indptr, doc_ids, vals = _random_csr(rng, V, N, density)
    nterms = min(nterms, V)
    cols = np.sort(rng.choice(V, size=nterms, replace=False)).astype(np.int32)
    qv = (rng.random(nterms, dtype=np.float32) * 2).astype(np.float32)
    filt, neg = SC.numba_score_float(indptr, doc_ids, vals, cols, qv, threshold, N)
    dense = np.zeros((V, N), np.float32)
    for t in range(V):
        dense[t, doc_ids[indptr[t]:indptr[t + 1]]] = vals[indptr[t]:indptr[t + 1]]
    ref = np.zeros(N, np.float64)
    for t, q in zip(cols, qv):
        ref += np.float64(q) * dense[t].astype(np.float64)
    np.testing.assert_allclose(-neg, ref[filt], rtol=1e-5, atol=1e-6)
    # exactly the docs above the threshold (away from rounding distance of it)
    clear = np.abs(ref - threshold) > 1e-4
    assert set(filt[clear[filt]].tolist()) == set(np.nonzero((ref > threshold) & clear)[0].tolist())
    ids, sc = SC.select_topk(filt, neg, k)
    assert len(ids) == min(k, len(filt)) and np.all(sc[:-1] >= sc[1:])
    if len(filt) > k:
        assert sc[-1] >= np.sort(-neg)[::-1][k - 1] - 1e-7
    # the C port is the same algorithm: identical bits
    oi, os_, oc = SC.sparse_retrieve_c(indptr, doc_ids, vals, np.array([0, nterms], np.int64), cols, qv, k, threshold, N,
                                       q_threads=1, inner_threads=1)
    assert oc[0] == len(ids) and np.array_equal(oi[0, :oc[0]], ids) and np.array_equal(os_[0, :oc[0]], sc)


@SET
@given(st.integers(0, 2 ** 31 - 1), st.integers(1, 9), st.integers(1, 200), st.sampled_from([16, 32, 64]), st.integers(1, 40),
       st.booleans())
def test_dense_oracle_topk_against_argsort(seed, nq, n, h, k, with_ties):
    rng = np.random.default_rng(seed)
    D = rng.standard_normal((n, h), dtype=np.float32)
    if with_ties and n > 3:
        D[n // 2] = D[0]                    # duplicate rows: equal scores must come out index-ascending
        D[n - 1] = D[0]
    Q = rng.standard_normal((nq, h), dtype=np.float32)
    s, i = SC.flat_ip_search(Q, D, k)
    S = Q @ D.T
    for q in range(nq):
        order = np.lexsort((np.arange(n), -S[q].astype(np.float64)))[:k]
        m = len(order)
        assert np.array_equal(i[q, :m], order) and np.array_equal(s[q, :m], S[q, order])
        assert np.all(i[q, m:] == -1)
    # the k-ordered fmaf chain is a reordering of the same sum
    F = SC.dense_scores_fma(Q, D, SC.dense_korder(nq, h))
    np.testing.assert_allclose(F, S, rtol=2e-5, atol=2e-5)
    fs, fi = SC.topk_rows(F, min(k, n))
    assert np.all(fs[:, :-1] >= fs[:, 1:])


@SET
@given(st.integers(0, 2 ** 31 - 1), st.integers(1, 30), st.integers(1, 25))
def test_metric_definitions(seed, nq, ndocs):
    from scaling_retriever_amd.utils.metrics import mrr_k, ndcg_k, recall_k
    rng = np.random.default_rng(seed)
    run, qrel, expect = {}, {}, []
    for q in range(nq):
        scores = rng.permutation(ndocs).astype(float)           # distinct scores
        run[f"q{q}"] = {f"d{d}": float(scores[d]) for d in range(ndocs)}
        rel = int(rng.integers(0, ndocs))
        qrel[f"q{q}"] = {f"d{rel}": 1}
        rank = int((scores > scores[rel]).sum()) + 1
        expect.append(1.0 / rank if rank <= 10 else 0.0)
    assert abs(mrr_k(run, qrel, 10) - float(np.mean(expect))) < 1e-12
    nd = ndcg_k(run, qrel, 10)
    assert 0.0 <= nd <= 1.0 + 1e-12 and (nd > 0) == (max(expect) > 0)
    assert abs(recall_k(run, qrel, ndocs) - 1.0) < 1e-12          # the whole list always contains the relevant doc
