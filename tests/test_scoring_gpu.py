"""GPU parity: HIP scoring kernels (through the C ABI) vs the CPU oracle.

Dense: reference = DenseFlatIndexer.search_knn over faiss IndexFlatIP
(/root/reference/scaling_retriever/indexer.py:191-217) - exact fp32 IP, top-k desc.
Sparse: reference = SparseRetrieval.numba_score_float + select_topk
(/root/reference/scaling_retriever/indexer.py:315-344).
"""
import numpy as np
import pytest
import torch

from oracle import scoring as O

pytestmark = pytest.mark.gpu

FLT_MIN = np.float32(-3.402823466e38)


def _dense_case(nq, n, h, k, seed, segments=1, scale=1.0):
    from scaling_retriever_amd.scoring import DenseIndexHIP
    rng = np.random.default_rng(seed)
    Q = (rng.standard_normal((nq, h), dtype=np.float32) * scale).astype(np.float32)
    D = (rng.standard_normal((n, h), dtype=np.float32) * scale).astype(np.float32)
    idx = DenseIndexHIP(h)
    bounds = np.linspace(0, n, segments + 1).astype(int)
    for a, b in zip(bounds[:-1], bounds[1:]):
        idx.add_host_rows(D[a:b], buffer_size=1000)
    s, i = idx.search(torch.from_numpy(Q).cuda(), k)
    torch.cuda.synchronize()
    return Q, D, s.cpu().numpy(), i.cpu().numpy()


def _check_dense_exact(Q, D, s, i, k):
    """Bit-exact against the k-ordered fmaf chain oracle + (score desc, id asc) top-k."""
    F = O.dense_scores_fma(Q, D, O.dense_korder(Q.shape[0], Q.shape[1]))
    es, ei = O.topk_rows(F, k)
    assert np.array_equal(i, ei), f"id mismatch: {np.argwhere(i != ei)[:5]}"
    assert np.array_equal(s, es)


@pytest.mark.parametrize("nq,n,h,k", [
    (1, 1000, 64, 10),          # single query (HBM-bound config)
    (16, 5000, 128, 100),
    (33, 3000, 64, 7),          # TN=64 config, ragged
    (100, 2500, 256, 1000),     # TN=128 config, k close to n
    (300, 4097, 64, 50),        # TN=256 config, ragged docs
    (1, 70001, 256, 10),        # streaming kernel (nq <= 32, H % 256 == 0), ragged docs
    (16, 9000, 512, 100),       # streaming, one 16-query block
    (17, 3000, 256, 1000),      # streaming, two query blocks
    (32, 200000, 256, 64),      # streaming, several geometric chunks
    (33, 3000, 256, 20),        # streaming, three query blocks
    (64, 5000, 512, 50),        # streaming, four query blocks
    (65, 3000, 256, 20),        # just above the streaming range
])
def test_dense_search_bit_exact(nq, n, h, k):
    Q, D, s, i = _dense_case(nq, n, h, k, seed=nq + n)
    _check_dense_exact(Q, D, s, i, k)


@pytest.mark.parametrize("k", [1, 100, 1000, 4096])
def test_dense_topk_running_set_under_adversarial_doc_orders(k):
    """The fused top-k keeps a running set of 2k keys and cuts it back to k only on overflow.  Doc orders that stress it: scores
    ascending along the index (every chunk beats tau: the set overflows again and again), descending (nothing passes after the
    first chunks), blocks of exact duplicates straddling chunk boundaries (ties by ascending doc index), plus random rows.
    Several launches per search (small candidate workspace); bit-exact vs the oracle."""
    from scaling_retriever_amd.scoring import DenseIndexHIP
    rng = np.random.default_rng(k)
    nq, n, h = 70, 40000, 64
    D = rng.standard_normal((n, h), dtype=np.float32)
    Q = rng.standard_normal((nq, h), dtype=np.float32)
    u = np.zeros(h, np.float32); u[0] = 1.0
    Q[0] = u; Q[1] = -u; Q[2] = u
    D[:, 0] = np.linspace(-3.0, 3.0, n, dtype=np.float32)            # query 0: ascending along the index, query 1: descending
    D[5000:5600] = D[5000]                                            # 600 exact duplicates
    D[20000:26000] = D[39999]                                         # 6000 copies of the best row of query 0 (> 2k of them at k <= 1000)
    idx = DenseIndexHIP(h)
    idx.add_host_rows(D)
    idx.set_workspace_limit(8 << 20)
    s, i = idx.search(torch.from_numpy(Q).cuda(), k)
    _check_dense_exact(Q, D, s.cpu().numpy(), i.cpu().numpy(), k)
    filt = DenseIndexHIP(h)
    filt.set_precision("fp32_filtered")
    filt.add_host_rows(D)
    filt.set_workspace_limit(8 << 20)
    fs, fi = filt.search(torch.from_numpy(Q).cuda(), k)
    assert torch.equal(fi, i) and torch.equal(fs, s)


def test_dense_matches_sgemm_oracle():
    """Against the faiss-style restatement (BLAS sgemm): same ids except near-ties, scores to 1e-5 rel."""
    nq, n, h, k = 64, 20000, 256, 100
    Q, D, s, i = _dense_case(nq, n, h, k, seed=5)
    es, ei = O.flat_ip_search(Q, D, k)
    np.testing.assert_allclose(s, es, rtol=2e-5, atol=2e-5)
    mism = (i != ei)
    if mism.any():  # only allowed where neighbouring scores are within rounding
        r, c = np.nonzero(mism)
        assert np.all(np.abs(es[r, c] - s[r, c]) <= 2e-5 * np.maximum(1, np.abs(es[r, c])))
    assert mism.mean() < 0.01


def test_dense_k_larger_than_n_pads_like_faiss():
    Q, D, s, i = _dense_case(5, 40, 64, 64, seed=9)
    F = O.dense_scores_fma(Q, D, O.mfma_korder(64))
    es, ei = O.topk_rows(F, 64)
    assert np.array_equal(i, ei) and np.array_equal(s, es)
    assert (i[:, 40:] == -1).all() and (s[:, 40:] == FLT_MIN).all()


def test_dense_ties_resolve_to_lowest_doc_index():
    from scaling_retriever_amd.scoring import DenseIndexHIP
    h, n = 64, 2000
    rng = np.random.default_rng(0)
    base = rng.standard_normal((10, h), dtype=np.float32)
    D = base[rng.integers(0, 10, size=n)]          # heavy duplication -> many exact ties
    Q = rng.standard_normal((7, h), dtype=np.float32)
    idx = DenseIndexHIP(h)
    idx.add_host_rows(D)
    s, i = idx.search(torch.from_numpy(Q).cuda(), 100)
    _check_dense_exact(Q, D, s.cpu().numpy(), i.cpu().numpy(), 100)


def test_dense_multi_segment_and_strided_ids():
    """Segments with id_base/id_stride (doc-sharded layout g_row = row*W + rank, indexer.py:262)."""
    from scaling_retriever_amd.scoring import DenseIndexHIP
    h, n, W, k = 64, 3000, 3, 50
    rng = np.random.default_rng(11)
    D = rng.standard_normal((n, h), dtype=np.float32)
    Q = rng.standard_normal((20, h), dtype=np.float32)
    idx = DenseIndexHIP(h)
    for r in range(W):
        idx.add_device_rows(torch.from_numpy(D[r::W].copy()).cuda(), id_base=r, id_stride=W)
    s, i = idx.search(torch.from_numpy(Q).cuda(), k)
    _check_dense_exact(Q, D, s.cpu().numpy(), i.cpu().numpy(), k)


def test_dense_small_workspace_forces_many_chunks():
    from scaling_retriever_amd.scoring import DenseIndexHIP
    h, n, k = 64, 6000, 30
    rng = np.random.default_rng(12)
    D = rng.standard_normal((n, h), dtype=np.float32)
    D = D[np.argsort(D @ np.ones(h, np.float32))]     # ascending along one direction: adversarial for tau
    Q = np.ones((3, h), np.float32) + 0.01 * rng.standard_normal((3, h), dtype=np.float32)
    idx = DenseIndexHIP(h)
    idx.add_host_rows(D)
    idx.set_workspace_limit(1 << 20)
    s, i = idx.search(torch.from_numpy(Q).cuda(), k)
    _check_dense_exact(Q, D, s.cpu().numpy(), i.cpu().numpy(), k)


def test_dense_linearity_property_large():
    """Size-independent property at a larger size: scores are linear in the query, so
    search(2q) returns the same ids with exactly doubled scores."""
    from scaling_retriever_amd.scoring import DenseIndexHIP
    h, n, k = 128, 300000, 100
    g = torch.Generator(device="cuda").manual_seed(1)
    D = torch.randn((n, h), device="cuda", generator=g)
    Q = torch.randn((40, h), device="cuda", generator=g)
    idx = DenseIndexHIP(h)
    idx.add_device_rows(D)
    s1, i1 = idx.search(Q, k)
    s2, i2 = idx.search(2 * Q, k)
    assert torch.equal(i1, i2) and torch.equal(2 * s1, s2)
    assert (s1[:, :-1] >= s1[:, 1:]).all()           # sorted descending
    ref = torch.topk(Q @ D.T, k, dim=1)
    assert (ref.indices == i1).float().mean() > 0.999
    torch.testing.assert_close(s1, ref.values, rtol=1e-4, atol=1e-4)


def test_dense_rejects_bad_arguments():
    from scaling_retriever_amd.scoring import DenseIndexHIP
    with pytest.raises(ValueError):
        DenseIndexHIP(30)
    idx = DenseIndexHIP(64)
    idx.add_host_rows(np.zeros((10, 64), np.float32))
    with pytest.raises(ValueError):
        idx.search(torch.zeros((2, 64), device="cuda"), 5000)
    with pytest.raises(ValueError):
        idx.search(torch.zeros((2, 32), device="cuda"), 5)


# ---------------------------------------------------------------------- sparse
def _random_index(rng, V, N, max_df, sort=True):
    indptr, ids, vals = [0], [], []
    for t in range(V):
        df = int(rng.integers(0, max_df)) if t % 5 else 0
        docs = rng.choice(N, size=min(df, N), replace=False).astype(np.int32)
        if sort:
            docs = np.sort(docs)
        ids.append(docs)
        vals.append(np.log1p(rng.uniform(0, 20, size=len(docs))).astype(np.float32))
        indptr.append(indptr[-1] + len(docs))
    return np.array(indptr, np.int64), np.concatenate(ids), np.concatenate(vals)


def _random_queries(rng, V, nq, max_terms):
    qi, qc, qv = [0], [], []
    for _ in range(nq):
        L0 = int(rng.integers(0, max_terms + 1))
        cols = np.sort(rng.choice(V, size=L0, replace=False)).astype(np.int32)
        qc.append(cols)
        qv.append(np.log1p(rng.uniform(0, 20, size=L0)).astype(np.float32))
        qi.append(qi[-1] + L0)
    return np.array(qi, np.int64), np.concatenate(qc), np.concatenate(qv)


def _check_sparse(indptr, ids, vals, N, qi, qc, qv, k, thr):
    from scaling_retriever_amd.scoring import SparseIndexHIP
    idx = SparseIndexHIP(indptr, ids, vals, N)
    s, i, c = idx.search(qi, qc, qv, k, threshold=thr)
    torch.cuda.synchronize()
    s, i, c = s.cpu().numpy(), i.cpu().numpy(), c.cpu().numpy()
    for q in range(len(qi) - 1):
        cols, v = qc[qi[q]:qi[q + 1]], qv[qi[q]:qi[q + 1]]
        fi, neg = O.numba_score_float(indptr, ids, vals, cols, v, thr, N)
        ei, es = O.select_topk(fi, neg, k)
        assert c[q] == len(ei), (q, c[q], len(ei))
        assert np.array_equal(i[q, :c[q]], ei), q
        assert np.array_equal(s[q, :c[q]], es), q     # bit-exact: term-serial unfused fp32 sums
        assert (i[q, c[q]:] == -1).all()


def test_sparse_golden_index(golden_dir):
    """The golden index of tests/golden/sparse_score.npz (posting lists unsorted, as in a
    merged multi-rank index): sort each list by doc id, then compare with the reference outputs."""
    import os
    from scaling_retriever_amd.scoring import SparseIndexHIP
    z = np.load(os.path.join(golden_dir, "sparse_score.npz"))
    indptr, ids, vals, N = z["indptr"], z["doc_ids"].copy(), z["vals"].copy(), int(z["N"])
    for t in range(len(indptr) - 1):
        b, e = indptr[t], indptr[t + 1]
        o = np.argsort(ids[b:e], kind="stable")
        ids[b:e], vals[b:e] = ids[b:e][o], vals[b:e][o]
    idx = SparseIndexHIP(indptr, ids, vals, N)
    for q in range(int(z["nq"])):
        cols, v = z[f"q{q}:cols"], z[f"q{q}:vals"]
        thr, k = float(z[f"q{q}:threshold"]), int(z[f"q{q}:k"])
        s, i, c = idx.search(np.array([0, len(cols)], np.int64), cols, v, k, threshold=thr)
        s, i, c = s.cpu().numpy()[0], i.cpu().numpy()[0], int(c.cpu().numpy()[0])
        o = np.argsort(i[:c])
        assert np.array_equal(i[:c][o], z[f"q{q}:topk_idx_sorted"]), q
        assert np.array_equal(s[:c][o], z[f"q{q}:topk_score_sorted"]), q


@pytest.mark.parametrize("V,N,max_df,nq,max_terms,k,thr", [
    (50, 300, 80, 10, 8, 10, 0.0),
    (200, 20000, 3000, 40, 30, 100, 0.0),       # 3 doc tiles
    (300, 70000, 20000, 25, 300, 1000, 0.0),    # > 256 query terms: two term batches
    (100, 9000, 2000, 12, 10, 50, 3.0),         # positive threshold
    (100, 9000, 2000, 6, 10, 50, -1.0),         # negative threshold: untouched docs qualify too
])
def test_sparse_search_bit_exact(V, N, max_df, nq, max_terms, k, thr):
    rng = np.random.default_rng(V + N)
    indptr, ids, vals = _random_index(rng, V, N, max_df)
    qi, qc, qv = _random_queries(rng, V, nq, min(max_terms, V))
    _check_sparse(indptr, ids, vals, N, qi, qc, qv, k, thr)


def test_sparse_rejects_unsorted_postings():
    from scaling_retriever_amd.scoring import SparseIndexHIP
    indptr = np.array([0, 3], np.int64)
    with pytest.raises(ValueError):
        SparseIndexHIP(indptr, np.array([5, 2, 9], np.int32), np.ones(3, np.float32), 10)
    with pytest.raises(ValueError):
        SparseIndexHIP(indptr, np.array([1, 2, 11], np.int32), np.ones(3, np.float32), 10)


def test_sparse_many_queries_batches_and_small_workspace():
    rng = np.random.default_rng(77)
    V, N = 400, 50000
    indptr, ids, vals = _random_index(rng, V, N, 5000)
    qi, qc, qv = _random_queries(rng, V, 1500, 12)          # > 1024 queries: two query batches
    from scaling_retriever_amd.scoring import SparseIndexHIP
    idx = SparseIndexHIP(indptr, ids, vals, N)
    idx.set_workspace_limit(64 << 20)
    s, i, c = idx.search(qi, qc, qv, 20)
    es_i, es_s, es_c = O.sparse_retrieve_c(indptr, ids, vals, qi, qc, qv, 20, 0.0, N, q_threads=4)
    assert np.array_equal(c.cpu().numpy(), es_c)
    assert np.array_equal(i.cpu().numpy(), es_i)
    assert np.array_equal(s.cpu().numpy(), es_s)


# ----------------------------------------------------------------------- merge
def test_topk_merge_equals_global_topk():
    from scaling_retriever_amd.scoring import topk_merge
    rng = np.random.default_rng(3)
    W, nq, k = 4, 9, 25
    scores = rng.standard_normal((W, nq, k)).astype(np.float32)
    scores[0, 0, :] = scores[1, 0, :]                       # exact ties across shards
    ids = np.stack([np.stack([rng.choice(1000, size=k, replace=False) * W + w for _ in range(nq)]) for w in range(W)])
    ids = ids.astype(np.int64)
    ids[2, 3, 10:] = -1                                      # a short list
    s, i = topk_merge(torch.from_numpy(scores).cuda(), torch.from_numpy(ids).cuda())
    s, i = s.cpu().numpy(), i.cpu().numpy()
    for q in range(nq):
        cs, ci = scores[:, q].ravel(), ids[:, q].ravel()
        keep = ci >= 0
        o = np.lexsort((ci[keep], -cs[keep].astype(np.float64)))[:k]
        assert np.array_equal(i[q], ci[keep][o]) and np.array_equal(s[q], cs[keep][o])


@pytest.mark.parametrize("nq,n,h,k", [(300, 70001, 256, 100), (700, 33000, 2048, 1000), (129, 513, 64, 10), (100, 50001, 512, 100),
                                      (65, 700, 64, 10), (128, 33000, 2048, 1000)])
def test_dense_pipelined_kernel_equals_plain_kernel(nq, n, h, k, monkeypatch):
    """The default score kernel for nq > 128 (three LDS stages, fragment prefetch across the barrier, memory operations
    dealt out between the MFMAs) accumulates every score in the same k order as the plain double-buffered kernel:
    ids and fp32 scores are bit-identical, ragged tiles included."""
    from scaling_retriever_amd.scoring import DenseIndexHIP
    g = torch.Generator(device="cuda").manual_seed(nq + n + h)
    D = torch.randn((n, h), device="cuda", generator=g)
    Q = torch.randn((nq, h), device="cuda", generator=g)
    idx = DenseIndexHIP(h)
    idx.add_device_rows(D)
    s5, i5 = idx.search(Q, k)
    monkeypatch.setenv("SR_DENSE_VARIANT", "1")
    s1, i1 = idx.search(Q, k)
    assert torch.equal(s1, s5) and torch.equal(i1, i5)


def test_topk_ties_straddling_rank_k_resolve_by_ascending_doc_index():
    """The reference's select_topk keeps an ARBITRARY k of the docs tied at the cut (np.argpartition, indexer.py:315-322) and
    faiss keeps whichever it met first; here the rule is fixed - score descending, then doc index ascending - and must hold
    exactly when a group of equal scores straddles rank k (a comparison sorted by id cannot see a wrong choice there)."""
    from scaling_retriever_amd.scoring import DenseIndexHIP, SparseIndexHIP
    # sparse: one posting list, all values equal except a few higher ones -> 20 000 docs tied below 5 leaders
    N, k = 20000, 37
    ids = np.arange(N, dtype=np.int32)
    vals = np.full(N, 0.5, np.float32)
    leaders = np.array([19000, 7, 12345, 8191, 8192])
    vals[leaders] = 2.0
    idx = SparseIndexHIP(np.array([0, N], np.int64), ids, vals, N)
    s, i, c = idx.search(np.array([0, 1], np.int64), np.array([0], np.int32), np.array([1.5], np.float32), k)
    tied = np.setdiff1d(np.arange(N), leaders)[:k - len(leaders)]              # the LOWEST doc indices of the tie group
    assert int(c) == k
    assert i[0].cpu().numpy().tolist() == sorted(leaders.tolist()) + tied.tolist()
    assert np.array_equal(s[0].cpu().numpy(), np.float32([3.0] * 5 + [0.75] * (k - 5)))
    oi, os_ = O.select_topk(np.arange(N, dtype=np.int64), -(vals * np.float32(1.5)), k)
    assert np.array_equal(oi, i[0].cpu().numpy()) and np.array_equal(os_, s[0].cpu().numpy())
    # dense: duplicated rows -> exact score ties across tile and segment boundaries
    rng = np.random.default_rng(0)
    base = rng.standard_normal((3, 64), dtype=np.float32)
    D = np.repeat(base, 3000, axis=0)[rng.permutation(9000)]
    q = rng.standard_normal((70, 64), dtype=np.float32)                        # 70 queries: the tiled MFMA kernel
    d = DenseIndexHIP(64)
    d.add_host_rows(D[:5000])
    d.add_host_rows(D[5000:])
    for Q in (q, q[:4]):                                                       # ... and the streaming kernel
        s, i = d.search(torch.from_numpy(Q).cuda(), 3500)
        es, ei = O.topk_rows(O.dense_scores_fma(Q, D, O.dense_korder(Q.shape[0], 64)), 3500)
        assert np.array_equal(i.cpu().numpy(), ei) and np.array_equal(s.cpu().numpy(), es)
        row = i[0].cpu().numpy()
        sc = s[0].cpu().numpy()
        for g0 in np.flatnonzero(np.diff(sc, prepend=np.inf) != 0):            # inside every tie group: ascending ids
            g1 = g0 + np.argmax(np.append(sc[g0:] != sc[g0], True))
            assert np.all(np.diff(row[g0:g1]) > 0)


@pytest.mark.parametrize("precision", ["fp32", "fp32_filtered"])
def test_batch_invariant_mode_gives_a_query_the_same_bits_alone_and_in_any_batch(precision):
    """sr_dense_index_set_batch_invariant: one k order for every batch size.  A query alone, among 8, among 40
    (the streaming kernel's range when the mode is off) and inside a batch of 300 (the tiled kernels / the certified filter) returns
    identical ids and fp32 scores through sr_dense_search; with the mode off the small batches are exact chains in another k order
    (equal to 1 ulp, the documented default - as faiss between its small-batch loop and its sgemm path)."""
    from scaling_retriever_amd.scoring import DenseIndexHIP
    dev = torch.device("cuda", 0)
    N, H, k = 200_000, 512, 100
    g = torch.Generator(device=dev).manual_seed(77)
    D = torch.randn((N, H), device=dev, generator=g) * (0.5 / H ** 0.5)
    Q = torch.randn((300, H), device=dev, generator=g) * (0.5 / H ** 0.5)
    idx = DenseIndexHIP(H, device=dev)
    idx.set_precision(precision)
    idx.add_device_rows(D)
    idx.set_batch_invariant(True)
    s300, i300 = idx.search(Q, k)
    for n in (1, 8, 40, 64, 65):
        s, i = idx.search(Q[:n].contiguous(), k)
        assert torch.equal(s, s300[:n]) and torch.equal(i, i300[:n]), n
    s1, i1 = idx.search(Q[7:8].contiguous(), k)
    assert torch.equal(s1, s300[7:8]) and torch.equal(i1, i300[7:8])
    # the mode changes the kernel family of small batches, not the large-batch result
    idx.set_batch_invariant(False)
    s300b, i300b = idx.search(Q, k)
    assert torch.equal(s300b, s300) and torch.equal(i300b, i300)
    s8, i8 = idx.search(Q[:8].contiguous(), k)
    assert torch.allclose(s8, s300[:8], rtol=1e-6, atol=1e-7)
    # and the results are the oracle's (k-ordered fmaf chain of the tiled kernels)
    Qh, Dh = Q[:4].cpu().numpy(), D.cpu().numpy()
    F = O.dense_scores_fma(Qh, Dh, O.dense_korder(300, H))
    es, ei = O.topk_rows(F, k)
    assert np.array_equal(i300[:4].cpu().numpy(), ei) and np.array_equal(s300[:4].cpu().numpy(), es)
