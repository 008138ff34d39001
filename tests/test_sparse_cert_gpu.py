"""GPU parity of the certified two-stage inverted-index scorer (csrc/sparse_cert.hip: fp16 MFMA + fixed-point LDS atomics, certificate,
exact re-score from a forward index) against the oracle's term-serial fp32 sums (reference: SparseRetrieval.numba_score_float +
select_topk, scaling_retriever/indexer.py:315-344).  Bit-exact results; the stage-1 keys are checked against the proven bound for
every (query, doc) pair; everything the fast path does not accept must come back identical through the exact kernels."""
import numpy as np
import pytest
import torch

from oracle import scoring as O

pytestmark = pytest.mark.gpu


def _zipf_index(rng, V, N, L0_d, vals="log1p", cap=1.0):
    """df_r ~ 1/r capped at cap * N, mean L0_d postings per doc; docs of a term drawn without replacement."""
    r = np.arange(1, V + 1, dtype=np.float64)
    lo, hi = 0.0, float(N) * L0_d * V
    for _ in range(100):
        C = 0.5 * (lo + hi)
        s_ = np.minimum(cap * N, C / r).sum()
        lo, hi = (C, hi) if s_ < N * L0_d else (lo, C)
    df = np.maximum(1, np.minimum(cap * N, C / r)).astype(np.int64)
    indptr, ids, vs = [0], [], []
    for t in range(V):
        n = int(df[t])
        docs = np.sort(rng.choice(N, size=n, replace=False)).astype(np.int32) if n < N else np.arange(N, dtype=np.int32)
        ids.append(docs)
        if vals == "log1p":
            v = np.log1p(rng.uniform(0, 20, size=n))
        elif vals == "wide":            # six decades of dynamic range
            v = np.exp(rng.uniform(np.log(1e-5), np.log(8.0), size=n))
        else:                           # few distinct values: massive score ties
            v = rng.choice([0.5, 1.0, 2.0], size=n)
        vs.append(v.astype(np.float32))
        indptr.append(indptr[-1] + n)
    return np.array(indptr, np.int64), np.concatenate(ids), np.concatenate(vs)


def _zipf_queries(rng, V, nq, L0_q, vals="log1p"):
    w = 1.0 / np.arange(1, V + 1)
    w /= w.sum()
    qi, qc, qv = [0], [], []
    for _ in range(nq):
        n = int(rng.integers(max(1, L0_q // 2), L0_q + 1))
        cols = np.sort(rng.choice(V, size=min(n, V), replace=False, p=w)).astype(np.int32)
        qc.append(cols)
        if vals == "ties":
            qv.append(rng.choice([1.0, 2.0], size=len(cols)).astype(np.float32))
        else:
            qv.append(np.log1p(rng.uniform(0, 20, size=len(cols))).astype(np.float32))
        qi.append(qi[-1] + len(cols))
    return np.array(qi, np.int64), np.concatenate(qc), np.concatenate(qv)


def _search_and_compare(idx, indptr, ids, vals, N, qi, qc, qv, k, thr=0.0, id_base=0, id_stride=1):
    s, i, c = idx.search(qi, qc, qv, k, threshold=thr, id_base=id_base, id_stride=id_stride)
    torch.cuda.synchronize()
    ei, es, ec = O.sparse_retrieve_c(indptr, ids, vals, qi, qc, qv, k, thr, N, q_threads=4)
    ei = np.where(ei >= 0, id_base + ei * id_stride, ei)
    s, i, c = s.cpu().numpy(), i.cpu().numpy(), c.cpu().numpy()
    assert np.array_equal(c, ec)
    bad = [q for q in range(len(qi) - 1) if not (np.array_equal(i[q], ei[q]) and np.array_equal(s[q], es[q]))]
    assert not bad, f"queries with different rows: {bad[:10]} ({len(bad)} of {len(qi) - 1})"
    return s, i, c


@pytest.fixture
def forced(monkeypatch):
    monkeypatch.setenv("SR_SPARSE_CERT", "1")        # build + use the certified scorer below its size threshold as well


def test_stage1_keys_obey_the_bound_for_every_query_doc_pair(forced):
    import scipy.sparse as sp
    from scaling_retriever_amd.scoring import SparseIndexHIP
    rng = np.random.default_rng(3)
    V, N, nq = 1500, 21000, 45
    indptr, ids, vals = _zipf_index(rng, V, N, 40)
    qi, qc, qv = _zipf_queries(rng, V, nq, 24)
    idx = SparseIndexHIP(indptr, ids, vals, N)
    st = idx.cert_stats()
    assert st["present"] == 1 and st["doc_tiles"] == (N + 1023) // 1024 and st["dense_terms"] % 16 == 0
    idx.cert_record_keys(True)
    _search_and_compare(idx, indptr, ids, vals, N, qi, qc, qv, 100)
    keys, consts, vscale, T = idx.cert_recorded_keys(nq)
    idx.cert_record_keys(False)
    assert T == st["dense_terms"]
    term_of = np.repeat(np.arange(V), np.diff(indptr))
    D = sp.csr_matrix((vals.astype(np.float64), (ids, term_of)), shape=(N, V))
    Q = sp.csr_matrix((qv.astype(np.float64), qc, qi), shape=(nq, V))
    true = np.asarray((Q @ D.T).todense())                                   # real-arithmetic scores, [nq, N]
    dd = 2.0 * 2.0 ** -11 + 2.4e-7 + T * 2.4e-7 + 1.0e-5
    checked = 0
    for q in range(nq):
        cq, sq, n_rare, n_qt, n_drop = consts[q, :5]
        if cq == 0:
            continue
        tf = 65535.0 * float(sq) * true[q]
        kq = keys[q, :N].astype(np.float64)
        assert (kq >= tf * (1 - dd) - 1.2 - 2.03 * n_drop).all(), (q, float((tf * (1 - dd) - 1.2 - kq).max()))
        assert (kq <= tf * (1 + dd) + 1.2 + 1.01 * n_rare).all(), (q, float((kq - tf * (1 + dd) - 1.2 - 1.01 * n_rare).max()))
        assert kq.max() <= 65535 * 0.99                                      # no key near its half word's limit
        assert (keys[q, N:] == 0).all()                                      # docs beyond the collection
        checked += 1
    assert checked >= nq - 2
    assert idx.cert_stats()["redone_exact"] <= 2


@pytest.mark.parametrize("V,N,L0_d,nq,L0_q,k,thr", [
    (3000, 66000, 48, 70, 32, 1000, 0.0),        # the headline's shape in small: Zipf index, k = 1000
    (800, 40000, 30, 33, 16, 100, 0.0),          # ragged last query block, 40 tiles
    (5000, 30000, 64, 64, 64, 10, 0.0),          # long queries, small k
    (2000, 50001, 40, 40, 24, 500, 3.0),         # positive threshold, one doc in the last tile
    (2000, 25000, 40, 20, 24, 200, -1.0),        # negative threshold: docs with score 0 qualify -> decided by the exact kernels where needed
])
def test_certified_search_is_bit_exact(forced, V, N, L0_d, nq, L0_q, k, thr):
    from scaling_retriever_amd.scoring import SparseIndexHIP
    rng = np.random.default_rng(V + N + k)
    indptr, ids, vals = _zipf_index(rng, V, N, L0_d)
    qi, qc, qv = _zipf_queries(rng, V, nq, L0_q)
    idx = SparseIndexHIP(indptr, ids, vals, N)
    assert idx.cert_stats()["present"] == 1
    _search_and_compare(idx, indptr, ids, vals, N, qi, qc, qv, k, thr)
    st = idx.cert_stats()
    assert st["searches"] == 1 and st["queries"] == nq
    if thr >= 0:
        assert st["redone_exact"] <= nq // 8, st          # the fast path carries the call


def test_queries_outside_the_fast_path_come_back_identical(forced):
    """Negative weights, descending / shuffled term order, duplicate terms, empty queries, unknown terms, zero weights, more than 256
    terms: all flagged by the plan kernel and served by the exact kernels inside the same call, next to certified queries (one of them
    with 90 rare terms: on the fast path since round 6).  A batch with NO query for the fast path costs the scorer no pass at all."""
    from scaling_retriever_amd.scoring import SparseIndexHIP
    rng = np.random.default_rng(11)
    V, N = 2500, 33000
    indptr, ids, vals = _zipf_index(rng, V, N, 40)
    qi, qc, qv = _zipf_queries(rng, V, 24, 20)
    qc, qv, qi = list(np.split(qc, qi[1:-1])), list(np.split(qv, qi[1:-1])), None
    qv[1] = -qv[1]                                               # all negative
    qv[2][::3] *= -1                                             # some negative
    qc[3], qv[3] = qc[3][::-1].copy(), qv[3][::-1].copy()        # descending order (the accumulation order is the query's own)
    p = rng.permutation(len(qc[4])); qc[4], qv[4] = qc[4][p], qv[4][p]
    qc[5] = np.concatenate([qc[5], qc[5][-1:]]); qv[5] = np.concatenate([qv[5], qv[5][-1:]])     # a duplicate term
    qc[6], qv[6] = np.zeros(0, np.int32), np.zeros(0, np.float32)                                # empty
    qc[7], qv[7] = np.array([V, V + 3], np.int32), np.array([1.0, 2.0], np.float32)              # unknown terms only
    qc[8] = np.concatenate([qc[8], [V + 1]]).astype(np.int32); qv[8] = np.concatenate([qv[8], [2.0]]).astype(np.float32)   # known + unknown: fast path
    qv[9][::2] = 0.0                                             # zero weights: fast path (they add exact zeros)
    qc[10] = np.sort(rng.choice(np.arange(200, V), size=90, replace=False)).astype(np.int32)     # 90 rare terms
    qv[10] = np.log1p(rng.uniform(0, 20, size=90)).astype(np.float32)
    qc[11] = np.sort(rng.choice(V, size=300, replace=False)).astype(np.int32)                    # 300 terms
    qv[11] = np.log1p(rng.uniform(0, 20, size=300)).astype(np.float32)
    qv[12][0] = np.float32(np.inf)
    qi = np.concatenate([[0], np.cumsum([len(c) for c in qc])]).astype(np.int64)
    qc, qv = np.concatenate(qc), np.concatenate(qv)
    idx = SparseIndexHIP(indptr, ids, vals, N)
    s, i, c = idx.search(qi, qc, qv, 50)
    torch.cuda.synchronize()
    st = idx.cert_stats()
    assert st["searches"] == 1 and 9 <= st["redone_exact"] <= 14, st
    s, i, c = s.cpu().numpy(), i.cpu().numpy(), c.cpu().numpy()
    for q in range(len(qi) - 1):
        if q == 12:
            continue                  # inf * 0-free products: the chain holds inf / nan, order of equal keys is not pinned by the oracle
        cols, v = qc[qi[q]:qi[q + 1]], qv[qi[q]:qi[q + 1]]
        known = (cols >= 0) & (cols < V)
        fi, neg = O.numba_score_float(indptr, ids, vals, cols[known], v[known], 0.0, N)
        ei, es = O.select_topk(fi, neg, 50)
        assert c[q] == len(ei), (q, c[q], len(ei))
        assert np.array_equal(i[q, :c[q]], ei) and np.array_equal(s[q, :c[q]], es), q


def test_ties_and_near_ties_at_the_cut(forced):
    """Few distinct values: thousands of docs share a score.  Either the band of undecided keys fits (certified) or the query is
    re-done exactly; the rows are the oracle's (score desc, doc asc) either way."""
    from scaling_retriever_amd.scoring import SparseIndexHIP
    rng = np.random.default_rng(5)
    V, N = 600, 30000
    indptr, ids, vals = _zipf_index(rng, V, N, 20, vals="ties")
    qi, qc, qv = _zipf_queries(rng, V, 40, 8, vals="ties")
    idx = SparseIndexHIP(indptr, ids, vals, N)
    _search_and_compare(idx, indptr, ids, vals, N, qi, qc, qv, 100)
    assert idx.cert_stats()["searches"] == 1


def test_wide_dynamic_range_of_values(forced):
    from scaling_retriever_amd.scoring import SparseIndexHIP
    rng = np.random.default_rng(9)
    V, N = 1200, 24000
    indptr, ids, vals = _zipf_index(rng, V, N, 30, vals="wide")
    qi, qc, qv = _zipf_queries(rng, V, 30, 16)
    qv[::5] *= 1e-4                                              # tiny query weights next to ordinary ones
    idx = SparseIndexHIP(indptr, ids, vals, N)
    _search_and_compare(idx, indptr, ids, vals, N, qi, qc, qv, 100)


def test_doc_shard_ids_and_k_beyond_the_band(forced):
    from scaling_retriever_amd.scoring import SparseIndexHIP
    rng = np.random.default_rng(21)
    V, N = 1000, 20000
    indptr, ids, vals = _zipf_index(rng, V, N, 30)
    qi, qc, qv = _zipf_queries(rng, V, 17, 16)
    idx = SparseIndexHIP(indptr, ids, vals, N)
    _search_and_compare(idx, indptr, ids, vals, N, qi, qc, qv, 200, id_base=3, id_stride=8)      # rank 3 of 8
    assert idx.cert_stats()["searches"] == 1
    _search_and_compare(idx, indptr, ids, vals, N, qi, qc, qv, 3500)                             # k + 1024 > 4096: exact kernels
    assert idx.cert_stats()["searches"] == 1


def test_same_rows_with_the_certified_scorer_switched_off(forced, monkeypatch):
    from scaling_retriever_amd.scoring import SparseIndexHIP
    rng = np.random.default_rng(31)
    V, N = 2000, 45000
    indptr, ids, vals = _zipf_index(rng, V, N, 40)
    qi, qc, qv = _zipf_queries(rng, V, 100, 32)
    idx = SparseIndexHIP(indptr, ids, vals, N)
    s1, i1, c1 = idx.search(qi, qc, qv, 1000)
    assert idx.cert_stats()["searches"] == 1
    monkeypatch.setenv("SR_SPARSE_CERT_SEARCH", "0")
    s0, i0, c0 = idx.search(qi, qc, qv, 1000)
    assert idx.cert_stats()["searches"] == 1
    assert torch.equal(s1, s0) and torch.equal(i1, i0) and torch.equal(c1, c0)
    # and again: the scorer's workspaces are reused call after call
    monkeypatch.setenv("SR_SPARSE_CERT_SEARCH", "1")
    s2, i2, c2 = idx.search(qi, qc, qv, 1000)
    assert torch.equal(s1, s2) and torch.equal(i1, i2) and torch.equal(c1, c2)


def test_indexes_the_scorer_does_not_take(forced):
    """A negative value, or a doc with more postings than the forward-index sort handles: no certified scorer, exact kernels only."""
    from scaling_retriever_amd.scoring import SparseIndexHIP
    rng = np.random.default_rng(41)
    V, N = 1500, 9000
    indptr, ids, vals = _zipf_index(rng, V, N, 30)
    qi, qc, qv = _zipf_queries(rng, V, 9, 12)
    v2 = vals.copy()
    v2[7] = -v2[7]
    idx = SparseIndexHIP(indptr, ids, v2, N)
    assert idx.cert_stats()["present"] == 0
    _search_and_compare(idx, indptr, ids, v2, N, qi, qc, qv, 20)
    # one doc holding 1 200 terms
    big = np.sort(rng.choice(V, size=1200, replace=False))
    rows = np.concatenate([np.repeat(np.arange(V), np.diff(indptr)), big])
    docs = np.concatenate([ids, np.full(1200, N, np.int32)])
    vv = np.concatenate([vals, np.ones(1200, np.float32)])
    o = np.lexsort((docs, rows))
    ip = np.concatenate([[0], np.cumsum(np.bincount(rows, minlength=V))]).astype(np.int64)
    idx2 = SparseIndexHIP(ip, docs[o].astype(np.int32), vv[o], N + 1)
    assert idx2.cert_stats()["present"] == 0
    _search_and_compare(idx2, ip, docs[o].astype(np.int32), vv[o], N + 1, qi, qc, qv, 20)


def test_auto_mode_keeps_small_collections_on_the_exact_kernels():
    from scaling_retriever_amd.scoring import SparseIndexHIP
    rng = np.random.default_rng(51)
    indptr, ids, vals = _zipf_index(rng, 500, 12000, 20)
    idx = SparseIndexHIP(indptr, ids, vals, 12000)
    assert idx.cert_stats()["present"] == 0


def test_more_rare_postings_per_tile_than_a_wave_stages(forced, monkeypatch):
    """Only 16 terms on the matrix pipe and a dense-ish index: the runs of a wave's 4 queries inside a tile hold several times the 512
    quads staged per step, so most of them go through the overflow windows of the flat walk (and, with runs of > 1 000 postings, past
    every per-window size).  Same bits as the oracle."""
    from scaling_retriever_amd.scoring import SparseIndexHIP
    monkeypatch.setenv("SR_SPARSE_CERT_T", "16")
    rng = np.random.default_rng(61)
    V, N = 300, 23000
    indptr, ids, vals = _zipf_index(rng, V, N, 90, cap=0.6)          # every second term in a third of the docs or more
    qi, qc, qv = _zipf_queries(rng, V, 37, 60)
    idx = SparseIndexHIP(indptr, ids, vals, N)
    st = idx.cert_stats()
    assert st["present"] == 1 and st["dense_terms"] == 16
    _search_and_compare(idx, indptr, ids, vals, N, qi, qc, qv, 100)
    st = idx.cert_stats()
    assert st["searches"] == 1 and st["redone_exact"] <= 4, st


def test_tiny_rare_weights_are_left_out_of_stage_one_and_still_exact(forced):
    """A rare term whose weight falls below fp16's normal range after the query's scaling is not scored in stage 1; the certificate
    widens by its bounded worth and the exact re-score brings it back: certified, same bits."""
    from scaling_retriever_amd.scoring import SparseIndexHIP
    rng = np.random.default_rng(71)
    V, N = 2500, 30000
    indptr, ids, vals = _zipf_index(rng, V, N, 40)
    qi, qc, qv = _zipf_queries(rng, V, 32, 24)
    rare = qc >= 200
    qv[rare & (np.arange(len(qv)) % 3 == 0)] *= np.float32(3e-7)
    idx = SparseIndexHIP(indptr, ids, vals, N)
    idx.cert_record_keys(True)
    _search_and_compare(idx, indptr, ids, vals, N, qi, qc, qv, 100)
    _, consts, _, _ = idx.cert_recorded_keys(32)
    idx.cert_record_keys(False)
    assert (consts[:32, 4] > 0).sum() >= 16          # most queries had a term left out ...
    assert idx.cert_stats()["redone_exact"] <= 2      # ... and were certified all the same


@pytest.mark.parametrize("sel_over", ["2.0", "1.0"])
def test_band_filter_threshold_is_valid_and_changes_no_row(forced, monkeypatch, sel_over):
    """While the scan runs, docs are filtered by the cut that follows from the k-th best key SO FAR (second rank of the top-k select,
    topk_compact2): it must never exceed the k-th best key of the whole collection, and rows must equal the oracle's and those of a scan
    that filters with the (k + band)-th best key only.  sel_over = 1.0: a select (and a fresh threshold) in every launch."""
    from scaling_retriever_amd.scoring import SparseIndexHIP
    rng = np.random.default_rng(77)
    V, N, nq, k = 2500, 230000, 40, 60                     # 225 tiles: launches of 2, 4, ... 128 tiles; k + band = 1 084 keys held
    indptr, ids, vals = _zipf_index(rng, V, N, 24)
    qi, qc, qv = _zipf_queries(rng, V, nq, 20)
    monkeypatch.setenv("SR_DEV_SWITCHES", "1")
    monkeypatch.setenv("SR_SPARSE_CERT_SELOVER", sel_over)
    idx = SparseIndexHIP(indptr, ids, vals, N)
    idx.cert_record_keys(True)
    s1, i1, c1 = _search_and_compare(idx, indptr, ids, vals, N, qi, qc, qv, k)
    keys, consts, _, _ = idx.cert_recorded_keys(nq)
    idx.cert_record_keys(False)
    assert idx.cert_stats()["redone_exact"] == 0
    seen = 0
    for q in range(nq):
        kth_all = np.sort(keys[q, :N])[-k]
        assert consts[q, 5] <= kth_all, (q, consts[q, 5], kth_all)
        seen += consts[q, 5] > 0
    assert seen >= nq // 2                                  # the second rank was found for most queries (a select ran)
    monkeypatch.setenv("SR_SPARSE_CERT_BAND", "0")
    s0, i0, c0 = idx.search(qi, qc, qv, k)
    assert np.array_equal(s0.cpu().numpy(), s1) and np.array_equal(i0.cpu().numpy(), i1) and np.array_equal(c0.cpu().numpy(), c1)


def test_long_queries_take_the_scalar_ordered_sum(forced):
    """More than 64 query terms (at most 64 of them rare, or the query is not on the fast path at all): the re-score adds a row's products
    in its scalar loop instead of the 64-entry LDS lists; rows must still be the oracle's."""
    from scaling_retriever_amd.scoring import SparseIndexHIP
    rng = np.random.default_rng(91)
    V, N, nq = 1500, 70000, 24
    indptr, ids, vals = _zipf_index(rng, V, N, 60)
    qi, qc, qv = [0], [], []
    for q in range(nq):
        heavy = rng.choice(120, size=70 + q % 20, replace=False)              # the longest lists: on the matrix pipe
        rare = 200 + rng.choice(V - 200, size=20 + q % 30, replace=False)
        cols = np.sort(np.concatenate([heavy, rare])).astype(np.int32)
        qc.append(cols)
        qv.append(np.log1p(rng.uniform(0, 20, size=len(cols))).astype(np.float32))
        qi.append(qi[-1] + len(cols))
    qi, qc, qv = np.array(qi, np.int64), np.concatenate(qc), np.concatenate(qv)
    idx = SparseIndexHIP(indptr, ids, vals, N)
    _search_and_compare(idx, indptr, ids, vals, N, qi, qc, qv, 100)
    st = idx.cert_stats()
    assert st["searches"] == 1 and st["redone_exact"] <= nq // 4 and st["candidates_rescored"] > 0


def test_query_set_in_batches_and_without_memory_for_the_scorer(forced, monkeypatch):
    """sr_sparse_search hands the certified scorer at most 8 192 queries at a time (its workspace is ~200 KB per query) and serves a
    batch with the exact kernels when that workspace cannot be allocated - the call does not fail, the rows are the same.  Here: batches
    of 64 (three whole ones and a ragged one), then every batch 'out of memory', then the scorer again (its buffers come back)."""
    from scaling_retriever_amd.scoring import SparseIndexHIP
    rng = np.random.default_rng(77)
    V, N, k = 1500, 40000, 200
    indptr, ids, vals = _zipf_index(rng, V, N, 40)
    qi, qc, qv = _zipf_queries(rng, V, 230, 24)
    idx = SparseIndexHIP(indptr, ids, vals, N)
    s1, i1, c1 = _search_and_compare(idx, indptr, ids, vals, N, qi, qc, qv, k)
    st = idx.cert_stats()
    assert st["searches"] == 1 and st["queries"] == 230 and st["batches_without_memory"] == 0
    monkeypatch.setenv("SR_SPARSE_CERT_BATCH", "64")
    s2, i2, c2 = _search_and_compare(idx, indptr, ids, vals, N, qi, qc, qv, k)
    st = idx.cert_stats()
    assert st["searches"] == 1 + 4 and st["queries"] == 460
    assert np.array_equal(s1, s2) and np.array_equal(i1, i2) and np.array_equal(c1, c2)
    monkeypatch.setenv("SR_SPARSE_CERT_FAKE_OOM", "1")
    s3, i3, c3 = _search_and_compare(idx, indptr, ids, vals, N, qi, qc, qv, k)
    st = idx.cert_stats()
    assert st["searches"] == 5 and st["batches_without_memory"] == 4
    assert np.array_equal(s1, s3) and np.array_equal(i1, i3)
    monkeypatch.delenv("SR_SPARSE_CERT_FAKE_OOM")
    monkeypatch.delenv("SR_SPARSE_CERT_BATCH")
    s4, i4, c4 = _search_and_compare(idx, indptr, ids, vals, N, qi, qc, qv, k)
    assert idx.cert_stats()["searches"] == 6 and np.array_equal(s1, s4) and np.array_equal(i1, i4)


def test_up_to_256_rare_terms_stay_on_the_fast_path(forced):
    """A query's rare terms (outside the index's heaviest ones) beyond the first 64 are added by a plain per-lane walk inside the same
    kernel step (cert_extra_groups) instead of sending the query to the exact kernels.  65, 128, 129, 200 and 256 rare terms next to
    short queries in the same blocks of 32, heavy terms mixed in, single postings and runs: rows are the oracle's bit for bit, the
    stage-1 keys obey the proven bound for every (query, doc) pair, and only the 257-term query is handed back."""
    from scaling_retriever_amd.scoring import SparseIndexHIP
    rng = np.random.default_rng(123)
    V, N, k = 3000, 70000, 100
    indptr, ids, vals = _zipf_index(rng, V, N, 60)
    df = np.diff(indptr)
    order = np.argsort(-df, kind="stable")
    heavy, rare = order[:128], order[128:]
    qi, qc, qv = [0], [], []
    n_rare_list = [65, 5, 128, 129, 30, 200, 256, 64, 257, 12, 100, 70] * 3        # 36 queries: two blocks of 32
    for q, nr in enumerate(n_rare_list):
        nh = 0 if nr >= 250 else int(rng.integers(0, 30))
        cols = np.sort(np.concatenate([rng.choice(heavy, size=nh, replace=False), rng.choice(rare[:1500] if q % 2 else rare, size=nr, replace=False)])).astype(np.int32)
        qc.append(cols)
        qv.append(np.log1p(rng.uniform(0, 20, size=len(cols))).astype(np.float32))
        qi.append(qi[-1] + len(cols))
    qi, qc, qv = np.array(qi, np.int64), np.concatenate(qc), np.concatenate(qv)
    idx = SparseIndexHIP(indptr, ids, vals, N)
    idx.cert_record_keys(True)
    _search_and_compare(idx, indptr, ids, vals, N, qi, qc, qv, k)
    st = idx.cert_stats()
    n_over = sum(1 for nr in n_rare_list if nr > 256)
    assert st["searches"] == 1 and n_over <= st["redone_exact"] <= n_over + 4, st
    # the bound of DESIGN.md 4.6 on the recorded keys, with the rare-term count the plan kernel reports (up to 256 now)
    keys, consts, vscale, T = idx.cert_recorded_keys(len(n_rare_list))
    assert sorted(set(int(x) for x in consts[:len(n_rare_list), 2])) [-1] > 64
    for q in (0, 2, 3, 5, 6, 10):
        cq, sq, n_r, n_qt, n_drop = (float(x) for x in consts[q, :5])
        assert cq > 0 and int(n_r) == n_rare_list[q]
        cols, v = qc[qi[q]:qi[q + 1]], qv[qi[q]:qi[q + 1]]
        true = np.zeros(N, np.float64)
        for t, w in zip(cols, v):
            true[ids[indptr[t]:indptr[t + 1]]] += float(w) * vals[indptr[t]:indptr[t + 1]].astype(np.float64)
        tf = true * sq * 65535.0
        dd = 2.0 * 2.0 ** -11 + 2.4e-7 + T * 2.4e-7 + 1.0e-5               # as in test_stage1_keys_obey_the_bound_for_every_query_doc_pair
        kq = keys[q, :N].astype(np.float64)
        assert (kq >= tf * (1 - dd) - 1.2 - 2.03 * n_drop).all(), q
        assert (kq <= tf * (1 + dd) + 1.2 + 1.01 * n_r).all(), q
        assert kq.max() <= 65535 * 0.99


def test_band_grows_with_the_rare_terms_and_a_failed_batch_is_retried_wider(forced, monkeypatch):
    """The certificate's band (keys kept beyond k) is chosen per batch from its largest rare-term count - each rare term widens the stretch
    of keys the 16-bit arithmetic cannot tell from the k-th by one unit: 1 024 keys up to 96 rare terms, 2 048 up to 160, 3 072 beyond -
    and when at least 256 queries of a batch are handed back under a band that could still grow, they go through the scorer once more
    with the widest one before the exact kernels get what is left.  Rows are the oracle's either way."""
    from scaling_retriever_amd.scoring import SparseIndexHIP
    rng = np.random.default_rng(321)
    V, N, k = 3000, 120000, 100
    indptr, ids, vals = _zipf_index(rng, V, N, 60)
    order = np.argsort(-np.diff(indptr), kind="stable")
    heavy, rare = order[:128], order[128:]

    def queries(nq, n_rare):
        qi, qc, qv = [0], [], []
        for _ in range(nq):
            cols = np.sort(np.concatenate([rng.choice(heavy, size=20, replace=False), rng.choice(rare, size=n_rare, replace=False)])).astype(np.int32)
            qc.append(cols)
            qv.append(np.log1p(rng.uniform(0, 20, size=len(cols))).astype(np.float32))
            qi.append(qi[-1] + len(cols))
        return np.array(qi, np.int64), np.concatenate(qc), np.concatenate(qv)
    idx = SparseIndexHIP(indptr, ids, vals, N)
    for n_rare in (30, 120, 200):                                   # bands 1 024 / 2 048 / 3 072
        qi, qc, qv = queries(40, n_rare)
        st0 = idx.cert_stats()
        _search_and_compare(idx, indptr, ids, vals, N, qi, qc, qv, k)
        st1 = idx.cert_stats()
        assert st1["searches"] == st0["searches"] + 1 and st1["redone_exact"] - st0["redone_exact"] <= 4, (n_rare, st0, st1)
    # the retry: a band of 64 keys certifies next to nothing of 300 near-tie queries; the second pass with the widest band does
    indptr_t, ids_t, vals_t = _zipf_index(rng, 1200, 90000, 30, vals="ties")
    qi, qc, qv = _zipf_queries(rng, 1200, 300, 16, vals="ties")
    idx_t = SparseIndexHIP(indptr_t, ids_t, vals_t, 90000)
    s_ref, i_ref, c_ref = _search_and_compare(idx_t, indptr_t, ids_t, vals_t, 90000, qi, qc, qv, 50)
    base = idx_t.cert_stats()
    monkeypatch.setenv("SR_SPARSE_CERT_BANDKEYS", "64")
    s2, i2, c2 = _search_and_compare(idx_t, indptr_t, ids_t, vals_t, 90000, qi, qc, qv, 50)
    st = idx_t.cert_stats()
    assert np.array_equal(s_ref, s2) and np.array_equal(i_ref, i2)
    assert st["searches"] == base["searches"] + 1 and st["queries"] == base["queries"] + 300        # a retried sub-batch is counted once


def test_batch_without_a_fast_path_query_costs_no_pass(forced):
    """Every query negative / unordered / too long: the plan kernel reports no query for the scorer, stage 1 does not run (no
    candidates re-scored, every query handed back) and the exact kernels serve the batch - the oracle's rows."""
    from scaling_retriever_amd.scoring import SparseIndexHIP
    rng = np.random.default_rng(55)
    V, N = 2500, 70000
    indptr, ids, vals = _zipf_index(rng, V, N, 40)
    qi, qc, qv = _zipf_queries(rng, V, 40, 20)
    qv = -qv                                                     # all weights negative
    idx = SparseIndexHIP(indptr, ids, vals, N)
    s, i, c = idx.search(qi, qc, qv, 50)
    torch.cuda.synchronize()
    st = idx.cert_stats()
    assert st["searches"] == 1 and st["queries"] == 40 and st["redone_exact"] == 40 and st["candidates_rescored"] == 0
    s, i, c = s.cpu().numpy(), i.cpu().numpy(), c.cpu().numpy()
    for q in range(40):
        cols, v = qc[qi[q]:qi[q + 1]], qv[qi[q]:qi[q + 1]]
        fi, neg = O.numba_score_float(indptr, ids, vals, cols, v, 0.0, N)
        ei, es = O.select_topk(fi, neg, 50)
        assert c[q] == len(ei) and np.array_equal(i[q, :c[q]], ei) and np.array_equal(s[q, :c[q]], es), q
