"""GPU parity of the 4-queries-per-workgroup inverted-index kernel (heavy terms as dense columns held in registers,
csrc/sparse_score.hip sparse_block_kernel) against the oracle's term-serial fp32 sums
(reference: SparseRetrieval.numba_score_float + select_topk, scaling_retriever/indexer.py:315-344).  Bit-exact."""
import numpy as np
import pytest
import torch

from oracle import scoring as O

pytestmark = pytest.mark.gpu


def _index(rng, V, N, heavy, mid_df, light_df):
    """heavy: {term: fraction of docs}; every 7th other term mid (df up to mid_df), the rest light; a few empty."""
    indptr, ids, vals = [0], [], []
    for t in range(V):
        if t in heavy:
            df = int(N * heavy[t])
        elif t % 11 == 5:
            df = 0
        elif t % 7 == 3:
            df = int(rng.integers(1, mid_df))
        else:
            df = int(rng.integers(1, light_df))
        docs = np.sort(rng.choice(N, size=min(df, N), replace=False)).astype(np.int32)
        ids.append(docs)
        vals.append(np.log1p(rng.uniform(0, 20, size=len(docs))).astype(np.float32))
        indptr.append(indptr[-1] + len(docs))
    return np.array(indptr, np.int64), np.concatenate(ids), np.concatenate(vals)


def _queries(rng, V, nq, max_terms, always=(), order="asc"):
    qi, qc, qv = [0], [], []
    for q in range(nq):
        L0 = int(rng.integers(0, max_terms + 1))
        cols = set(rng.choice(V, size=L0, replace=False).tolist())
        for t in always:
            if rng.uniform() < 0.7:
                cols.add(t)
        cols = np.array(sorted(cols), np.int32)
        if order == "desc":
            cols = cols[::-1].copy()
        elif order == "mixed" and q % 5 == 2:
            cols = rng.permutation(cols).astype(np.int32)
        qc.append(cols)
        qv.append(np.log1p(rng.uniform(0, 20, size=len(cols))).astype(np.float32))
        qi.append(qi[-1] + len(cols))
    return np.array(qi, np.int64), np.concatenate(qc), np.concatenate(qv)


DENSE_DIV, DENSE_MAX = 4, 64      # csrc/sparse_score.hip SP_DENSE_DIV / SP_DENSE_MAX


def _n_dense(indptr, N):
    lens = np.diff(indptr)
    return int(min(DENSE_MAX, ((lens > 0) & (lens * DENSE_DIV >= N)).sum()))


def _check(indptr, ids, vals, N, qi, qc, qv, k, thr, expect_dense=None, expect_fallback=None):
    from scaling_retriever_amd.scoring import SparseIndexHIP
    idx = SparseIndexHIP(indptr, ids, vals, N)
    if expect_dense == "auto":
        expect_dense = _n_dense(indptr, N)
        assert expect_dense > 0
    s, i, c = idx.search(qi, qc, qv, k, threshold=thr)
    torch.cuda.synchronize()
    st = idx.block_stats()
    if expect_dense is not None:
        assert st["dense_terms"] == expect_dense, st
        assert st["block_calls"] == (1 if expect_dense else 0), st
    if expect_fallback is not None:
        assert st["fallback_calls"] == expect_fallback, st
    s, i, c = s.cpu().numpy(), i.cpu().numpy(), c.cpu().numpy()
    for q in range(len(qi) - 1):
        cols, v = qc[qi[q]:qi[q + 1]], qv[qi[q]:qi[q + 1]]
        known = (cols >= 0) & (cols < len(indptr) - 1)
        fi, neg = O.numba_score_float(indptr, ids, vals, cols[known], v[known], thr, N)
        ei, es = O.select_topk(fi, neg, k)
        assert c[q] == len(ei), (q, c[q], len(ei))
        assert np.array_equal(i[q, :c[q]], ei), q
        assert np.array_equal(s[q, :c[q]], es), q
        assert (i[q, c[q]:] == -1).all()
    return idx


@pytest.mark.parametrize("V,N,heavy,nq,max_terms,k,thr", [
    (60, 3000, {0: 1.0, 1: 0.9, 2: 0.55}, 13, 10, 20, 0.0),                      # heavy prefix, one tile, ragged last block
    (120, 20000, {3: 1.0, 17: 0.8, 40: 0.6, 41: 0.5, 99: 0.75}, 37, 24, 100, 0.0),  # heavy ids scattered: runs alternate
    (300, 30000, {0: 1.0, 1: 1.0, 5: 0.7, 150: 0.9, 299: 0.6}, 22, 60, 1000, 0.0),  # union > 64 entries: two plan batches
    (80, 17000, {2: 0.95, 9: 0.5}, 9, 12, 50, 2.5),                              # positive threshold
    (80, 17000, {2: 0.95, 9: 0.5}, 6, 12, 50, -1.0),                             # negative threshold
])
def test_block_kernel_bit_exact(V, N, heavy, nq, max_terms, k, thr):
    rng = np.random.default_rng(V * 7 + N)
    indptr, ids, vals = _index(rng, V, N, heavy, N // 4, max(2, N // 50))
    qi, qc, qv = _queries(rng, V, nq, min(max_terms, V), always=tuple(heavy))
    _check(indptr, ids, vals, N, qi, qc, qv, k, thr, expect_dense="auto", expect_fallback=0)


def test_block_kernel_zero_weights_unknown_terms_and_empty_queries():
    rng = np.random.default_rng(5)
    V, N = 90, 12000
    heavy = {1: 1.0, 30: 0.8, 31: 0.6}
    indptr, ids, vals = _index(rng, V, N, heavy, 3000, 200)
    qi, qc, qv = _queries(rng, V, 11, 14, always=tuple(heavy))
    qv[::3] = 0.0                                  # explicit zero weights: add exact zeros
    qv[1::7] = -qv[1::7]                           # negative weights
    # a query made of unknown terms only, and one whose last term is unknown (ascending order kept)
    qc = np.concatenate([qc, np.array([V, V + 5], np.int32), np.array([1, 30, V + 2], np.int32)])
    qv = np.concatenate([qv, np.array([1.0, 2.0], np.float32), np.array([0.5, 1.5, 3.0], np.float32)])
    qi = np.concatenate([qi, [qi[-1] + 2, qi[-1] + 5]])
    _check(indptr, ids, vals, N, qi, qc, qv, 25, 0.0, expect_dense="auto", expect_fallback=0)


@pytest.mark.parametrize("order", ["desc", "mixed"])
def test_non_ascending_queries_take_the_per_query_kernel(order):
    """The accumulation order is the query's own term order: blocks holding a query whose terms do not ascend are scored
    by the per-query kernel in that order, the others by the block kernel, in one call."""
    rng = np.random.default_rng(11)
    V, N = 70, 15000
    heavy = {0: 1.0, 20: 0.7, 50: 0.55}
    indptr, ids, vals = _index(rng, V, N, heavy, 4000, 300)
    qi, qc, qv = _queries(rng, V, 23, 16, always=tuple(heavy), order=order)
    _check(indptr, ids, vals, N, qi, qc, qv, 40, 0.0, expect_dense="auto", expect_fallback=1)


def test_no_heavy_term_means_no_block_path():
    rng = np.random.default_rng(2)
    V, N = 50, 9000
    indptr, ids, vals = _index(rng, V, N, {}, 2000, 100)
    assert _n_dense(indptr, N) == 0
    qi, qc, qv = _queries(rng, V, 9, 10)
    _check(indptr, ids, vals, N, qi, qc, qv, 30, 0.0, expect_dense=0)


def test_more_heavy_terms_than_columns_and_many_queries():
    """Over 64 heavy terms: the longest 64 get columns, the others stay posting lists (some walked in groups, some in one
    step); > 1024 queries: two query batches."""
    rng = np.random.default_rng(8)
    V, N = 100, 10000
    heavy = {t: 0.5 + 0.005 * t for t in range(0, 80)}
    indptr, ids, vals = _index(rng, V, N, heavy, 2000, 100)
    qi, qc, qv = _queries(rng, V, 1100, 6)
    from scaling_retriever_amd.scoring import SparseIndexHIP
    idx = SparseIndexHIP(indptr, ids, vals, N)
    idx.set_workspace_limit(64 << 20)
    assert idx.block_stats()["dense_terms"] == 64
    s, i, c = idx.search(qi, qc, qv, 20)
    es_i, es_s, es_c = O.sparse_retrieve_c(indptr, ids, vals, qi, qc, qv, 20, 0.0, N, q_threads=4)
    assert np.array_equal(c.cpu().numpy(), es_c)
    assert np.array_equal(i.cpu().numpy(), es_i)
    assert np.array_equal(s.cpu().numpy(), es_s)


@pytest.mark.parametrize("seed", range(6))
def test_block_kernel_randomised(seed):
    """Random shapes around the kernel's internal sizes (4096-doc sub-tiles, 64-entry plan batches, 64-posting short runs,
    256-posting groups, blocks of 4 queries): index sizes just below / above tile multiples, heavy terms at random ids and
    densities, run lengths straddling the short / long boundary, query sets that are not a multiple of 4, some queries in
    non-ascending order (per-query kernel in the same call), thresholds of either sign, small candidate workspace."""
    from scaling_retriever_amd.scoring import SparseIndexHIP
    rng = np.random.default_rng(1000 + seed)
    V = int(rng.integers(40, 260))
    N = int(rng.choice([4095, 4096, 4097, 8191, 8193, 12289, 20000, 33000]))
    n_heavy = int(rng.integers(1, 12))
    heavy = {int(t): float(rng.uniform(0.26, 1.0)) for t in rng.choice(V, size=n_heavy, replace=False)}
    mid_df = max(3, int(N * rng.uniform(0.02, 0.24)))            # up to ~1000 postings per sub-tile: several groups
    light_df = max(2, int(N * rng.uniform(0.001, 0.03)))           # around the 64-postings-per-sub-tile boundary
    indptr, ids, vals = _index(rng, V, N, heavy, mid_df, light_df)
    nq = int(rng.integers(1, 70))
    order = "mixed" if seed % 2 else "asc"
    qi, qc, qv = _queries(rng, V, nq, min(V, int(rng.integers(1, 90))), always=tuple(heavy), order=order)
    if seed % 3 == 0 and len(qv):
        qv[rng.integers(0, max(1, len(qv)), size=max(1, len(qv) // 7))] *= -1.0
    thr = float(rng.choice([0.0, 0.0, 1.5, -0.5]))
    k = int(rng.choice([1, 10, 100, 1000]))
    idx = SparseIndexHIP(indptr, ids, vals, N)
    if seed % 2 == 0:
        idx.set_workspace_limit(8 << 20)
    s, i, c = idx.search(qi, qc, qv, k, threshold=thr)
    es_i, es_s, es_c = O.sparse_retrieve_c(indptr, ids, vals, qi, qc, qv, k, thr, N, q_threads=4)
    assert idx.block_stats()["dense_terms"] == _n_dense(indptr, N)
    assert np.array_equal(c.cpu().numpy(), es_c)
    assert np.array_equal(i.cpu().numpy(), es_i)
    assert np.array_equal(s.cpu().numpy(), es_s)


def test_block_order_does_not_change_results(monkeypatch):
    """The batch's queries are sorted (scatter work, dense-term set) before they are cut into blocks of 4; with the sort
    switched off (blocks of consecutive queries) every query's ids and scores are the same bits."""
    from scaling_retriever_amd.scoring import SparseIndexHIP
    rng = np.random.default_rng(77)
    V, N = 150, 26000
    heavy = {int(t): float(rng.uniform(0.3, 1.0)) for t in rng.choice(V, size=9, replace=False)}
    indptr, ids, vals = _index(rng, V, N, heavy, 5000, 400)
    qi, qc, qv = _queries(rng, V, 1301, 20, always=tuple(heavy))            # two batches of queries, ragged last block
    idx = SparseIndexHIP(indptr, ids, vals, N)
    s1, i1, c1 = idx.search(qi, qc, qv, 50)
    monkeypatch.setenv("SR_SPARSE_SORT", "0")
    s0, i0, c0 = idx.search(qi, qc, qv, 50)
    assert torch.equal(s1, s0) and torch.equal(i1, i0) and torch.equal(c1, c0)
    es_i, es_s, es_c = O.sparse_retrieve_c(indptr, ids, vals, qi, qc, qv, 50, 0.0, N, q_threads=4)
    assert np.array_equal(i1.cpu().numpy(), es_i) and np.array_equal(s1.cpu().numpy(), es_s) and np.array_equal(c1.cpu().numpy(), es_c)
