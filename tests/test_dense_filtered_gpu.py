"""SR_PRECISION_FP32_FILTERED: sr_dense_search through the certified fp16 filter + exact re-score must return EXACTLY what
the exact fp32 kernel returns - ids and fp32 scores, bit for bit (= the oracle's k-ordered fmaf chain, i.e. what
faiss.IndexFlatIP.search is restated as, /root/reference/scaling_retriever/indexer.py:210-214) - on benign and on
adversarial data, and must fall back to the exact kernel when the certificate cannot be given.  Also pins the error bound
the certificate rests on."""
import numpy as np
import pytest
import torch

from oracle import scoring as O

pytestmark = pytest.mark.gpu


def _both(D_parts, Q, k, bases=None):
    from scaling_retriever_amd.scoring import DenseIndexHIP
    H = Q.shape[1]
    exact, filt = DenseIndexHIP(H), DenseIndexHIP(H)
    filt.set_precision("fp32_filtered")
    for i, part in enumerate(D_parts):
        kw = {} if bases is None else {"id_base": bases[i][0], "id_stride": bases[i][1]}
        exact.add_host_rows(part, **kw)
        filt.add_host_rows(part, **kw)
    q = torch.from_numpy(Q).cuda()
    es, ei = exact.search(q, k)
    fs, fi = filt.search(q, k)
    return (es, ei), (fs, fi), filt


@pytest.mark.parametrize("H,N,nq,k", [(128, 30000, 200, 100), (256, 12000, 70, 1000), (2048, 40000, 300, 1000), (4096, 9000, 130, 50)])
def test_filtered_equals_exact_and_oracle(H, N, nq, k):
    rng = np.random.default_rng(H + N)
    D = (rng.standard_normal((N, H), dtype=np.float32) * (0.5 / np.sqrt(H))).astype(np.float32)
    Q = (rng.standard_normal((nq, H), dtype=np.float32) * (0.5 / np.sqrt(H))).astype(np.float32)
    (es, ei), (fs, fi), filt = _both([D[:N // 3], D[N // 3:]], Q, k)
    assert torch.equal(fi, ei) and torch.equal(fs, es)
    assert filt.filter_stats() == (1, 0)                                  # answered by the filter, not by the fallback
    n_or = min(nq, 40)
    os_, oi = O.topk_rows(O.dense_scores_fma(Q[:n_or], D, O.mfma_korder(H)), k)
    assert np.array_equal(fi[:n_or].cpu().numpy(), oi) and np.array_equal(fs[:n_or].cpu().numpy(), os_)


@pytest.mark.parametrize("seed", range(8))
def test_filtered_randomised_shapes(seed):
    """Random dims (64 .. 1024, multiples of 64), doc counts around tile multiples, query counts just above the 64-query
    threshold of the tiled kernels, k from 1 to 2000, 1-3 segments with strided ids, value scales from 1e-3 to 1e3: the
    filter's answer equals the exact kernel's, bit for bit, whichever of its paths (1 or 2 products, fallback) ran."""
    rng = np.random.default_rng(500 + seed)
    H = 64 * int(rng.integers(1, 17))
    N = int(rng.choice([255, 256, 257, 5000, 8191, 12289, 30001]))
    nq = int(rng.choice([65, 66, 127, 129, 256, 300]))
    k = int(rng.choice([1, 7, 100, 1000, 2000]))
    scale = float(10.0 ** rng.uniform(-3, 3))
    D = (rng.standard_normal((N, H), dtype=np.float32) * scale).astype(np.float32)
    Q = (rng.standard_normal((nq, H), dtype=np.float32) / scale).astype(np.float32)
    if seed % 2:
        D[N // 2:N // 2 + min(50, N // 4)] = D[0]                 # duplicates
    nseg = int(rng.integers(1, 4))
    parts = [D[j::nseg] for j in range(nseg)]
    (es, ei), (fs, fi), filt = _both(parts, Q, k, bases=[(j, nseg) for j in range(nseg)])
    assert torch.equal(fi, ei) and torch.equal(fs, es), (H, N, nq, k, nseg)
    os_, oi = O.topk_rows(O.dense_scores_fma(Q[:8], D, O.mfma_korder(H)), k)
    assert np.array_equal(fi[:8].cpu().numpy(), oi) and np.array_equal(fs[:8].cpu().numpy(), os_)


def test_filtered_adversarial_data_still_exact():
    """Same-sign vectors (no cancellation: the largest |S_a - S_x| per unit of |q||d|), a 1e6 dynamic range inside the rows,
    documents of very different norms, near-duplicate documents around the cut, strided global ids."""
    rng = np.random.default_rng(7)
    H, N, nq, k = 512, 20000, 96, 200
    D = np.abs(rng.standard_normal((N, H), dtype=np.float32))
    D *= np.exp(rng.uniform(-7, 7, size=(1, H))).astype(np.float32)               # per-column scales over 1e6
    D *= np.exp(rng.uniform(-2, 2, size=(N, 1))).astype(np.float32)               # norms spread over e^4
    base = D[100].copy()
    for j in range(150):                                                           # 150 near-duplicates of one document
        D[200 + j] = base * np.float32(1.0 + 1e-7 * j)
    Q = np.abs(rng.standard_normal((nq, H), dtype=np.float32))
    Q[0] = base                                                                    # ... which the first query hits head on
    (es, ei), (fs, fi), filt = _both([D[0::2], D[1::2]], Q, k, bases=[(0, 2), (1, 2)])
    assert torch.equal(fi, ei) and torch.equal(fs, es)
    os_, oi = O.topk_rows(O.dense_scores_fma(Q[:16], D, O.mfma_korder(H)), k)
    assert np.array_equal(fi[:16].cpu().numpy(), oi) and np.array_equal(fs[:16].cpu().numpy(), os_)


def test_filtered_redoes_what_it_cannot_certify():
    """More than kp - k documents tie with the k-th one (exact duplicates): no certificate -> the exact kernel answers, the
    result is still the exact kernel's (ties by ascending doc index).  A query without a certificate is re-done ALONE."""
    rng = np.random.default_rng(3)
    H, k = 128, 50
    base = rng.standard_normal((8, H), dtype=np.float32)
    D = np.repeat(base, 3000, axis=0)[rng.permutation(24000)]                      # 2 999 exact twins each: more than kp - k
    Q = rng.standard_normal((100, H), dtype=np.float32)
    (es, ei), (fs, fi), filt = _both([D], Q, k)
    assert torch.equal(fi, ei) and torch.equal(fs, es)
    assert filt.filter_stats() == (0, 1) and filt.filter_query_stats() == (0, 100)
    # 1 499 twins each fit inside the candidates: the ties at the cut are all re-scored, the certificate holds
    D2 = np.repeat(base, 1500, axis=0)[rng.permutation(12000)]
    (es, ei), (fs, fi), filt = _both([D2], Q, k)
    assert torch.equal(fi, ei) and torch.equal(fs, es)
    assert filt.filter_stats() == (1, 0)
    # a zero query: every score is 0, nothing can be separated - that query alone goes to the exact kernel
    Q[5] = 0
    (es, ei), (fs, fi), filt = _both([rng.standard_normal((5000, H), dtype=np.float32)], Q, k)
    assert torch.equal(fi, ei) and torch.equal(fs, es)
    assert filt.filter_stats() == (0, 1) and filt.filter_query_stats() == (99, 1)
    # a non-finite query: the same
    Q[5] = rng.standard_normal(H)
    Q[7, 3] = np.inf
    Dg = rng.standard_normal((5000, H), dtype=np.float32)
    (es, ei), (fs, fi), filt = _both([Dg], Q, k)
    ok = np.ones(100, bool)
    ok[7] = False                                      # (inf and NaN scores: the two kernels agree on ids only where scores order)
    assert torch.equal(fi[ok], ei[ok]) and torch.equal(fs[ok], es[ok])
    assert filt.filter_query_stats() == (99, 1)
    # non-finite data: the index is never filtered
    Dn = rng.standard_normal((5000, H), dtype=np.float32)
    Dn[17, 3] = np.inf
    _, (fs, fi), filt = _both([Dn], rng.standard_normal((100, H), dtype=np.float32), k)
    assert filt.filter_stats() == (0, 1)


def test_filter_redoes_a_few_queries_in_the_tiled_k_order():
    """A handful of uncertifiable queries (each aimed at a block of 3 000 exact twins) inside a batch of 300: they are re-done
    by the exact kernel in a batch of their own - far fewer than 64 queries, where sr_dense_search would pick the streaming
    kernel and ITS accumulation order - and must still match the exact kernel run on the whole batch bit for bit."""
    rng = np.random.default_rng(17)
    H, k, nq = 256, 100, 300
    D = rng.standard_normal((40000, H), dtype=np.float32)
    twins = rng.standard_normal((3, H), dtype=np.float32)
    for t in range(3):
        D[5000 + 3000 * t:8000 + 3000 * t] = twins[t]
    Q = rng.standard_normal((nq, H), dtype=np.float32)
    for t, q in enumerate((4, 150, 299)):
        Q[q] = twins[t] * 3.0
    (es, ei), (fs, fi), filt = _both([D[:20000], D[20000:]], Q, k)
    assert torch.equal(fi, ei) and torch.equal(fs, es)
    cert, redone = filt.filter_query_stats()          # the 3 aimed queries + the few random ones whose rank-k cut falls into a twin block
    assert filt.filter_stats() == (0, 1) and cert + redone == nq and 3 <= redone <= 15, (cert, redone)
    os_, oi = O.topk_rows(O.dense_scores_fma(Q[[4, 150, 299]], D, O.mfma_korder(H)), k)
    assert np.array_equal(fi[[4, 150, 299]].cpu().numpy(), oi) and np.array_equal(fs[[4, 150, 299]].cpu().numpy(), os_)


def test_filtered_small_index_and_small_batches():
    """Fewer documents than candidates (every document is re-scored), fewer than k documents (padding), and batches of <= 64
    queries, which the streaming kernel answers in its own k order whatever the mode."""
    from scaling_retriever_amd.scoring import DenseIndexHIP
    rng = np.random.default_rng(5)
    H = 256
    for N, k in ((1500, 1000), (700, 1000)):
        D = rng.standard_normal((N, H), dtype=np.float32)
        Q = rng.standard_normal((80, H), dtype=np.float32)
        (es, ei), (fs, fi), filt = _both([D], Q, k)
        assert torch.equal(fi, ei) and torch.equal(fs, es) and filt.filter_stats() == (1, 0)
        assert int((fi[0] >= 0).sum()) == min(N, k)
    D = rng.standard_normal((20000, H), dtype=np.float32)
    Q = rng.standard_normal((16, H), dtype=np.float32)
    (es, ei), (fs, fi), filt = _both([D], Q, 100)
    assert torch.equal(fi, ei) and torch.equal(fs, es) and filt.filter_stats() == (0, 0)


def _restated_upper_bound(Q, D):
    """The filter's U(q, j) restated in float64 from its definition (csrc/dense_filter.hip): power-of-two scales, fp16 planes,
    actual residual norms.  Returns (U, e), both [nq, N], in the true domain."""
    H = Q.shape[1]
    sigma = 1.25 * (3.0 * H * 2.0 ** -24) + 1.0e-5
    d = torch.from_numpy(D).cuda()
    q = torch.from_numpy(Q).cuda()
    sd = 2.0 ** (15 - int(np.frexp(float(np.abs(D).max()))[1]))
    sq = torch.exp2(15 - torch.frexp(q.abs().amax(dim=1)).exponent.double())[:, None]
    dp, qp = d.double() * sd, q.double() * sq
    d0, q0 = (d * sd).half().double(), (q * sq.float()).half().double()
    X = ((dp - d0).norm(dim=1) + sigma * dp.norm(dim=1)) * 1.001
    Y = d0.norm(dim=1) * 1.001
    A = qp.norm(dim=1) * 1.001
    B = (qp - q0).norm(dim=1) * 1.001
    e = (A[:, None] * X[None, :] + B[:, None] * Y[None, :]) / (sq * sd)
    return (q0 @ d0.T) / (sq * sd) + e, e


def test_filter_upper_bound_dominates_the_exact_score():
    """The certificate rests on U(q, j) >= S_x(q, j) >= U - 2 e for EVERY pair.  U is restated in float64 from the definition
    (the MFMA's fp32 accumulation adds at most what sigma reserves for it and is checked on the device for every re-scored
    pair: a violation re-does the query, which filter_query_stats would show), S_x comes from the exact kernel.  Gaussian,
    same-sign, adversarially aligned data (every fp16 rounding error at its maximum, all pushed the same way) and rows whose
    values span 1e6."""
    from scaling_retriever_amd.scoring import DenseIndexHIP
    rng = np.random.default_rng(11)
    for H in (256, 2048):
        for kind in ("gauss", "same_sign", "aligned", "wide"):
            D = rng.standard_normal((2000, H), dtype=np.float32)
            Q = rng.standard_normal((128, H), dtype=np.float32)
            if kind in ("same_sign", "aligned"):
                D, Q = np.abs(D), np.abs(Q)
            if kind == "aligned":            # values just below an fp16 rounding boundary: d - d0 = -2^-11 |d| for every element
                D = (np.float32(1.0) + np.float32(2.0 ** -10) * np.float32(0.499)) * np.exp2(rng.integers(-3, 3, size=D.shape)).astype(np.float32)
            if kind == "wide":
                D *= np.exp(rng.uniform(-7, 7, size=(1, H))).astype(np.float32)
            a = DenseIndexHIP(H)
            a.add_host_rows(D)
            q = torch.from_numpy(Q).cuda()
            es, ei = a.search(q, 2000)
            ex = torch.zeros((128, 2000), device="cuda").scatter_(1, ei, es).double()
            U, e = _restated_upper_bound(Q, D)
            bound = torch.from_numpy(np.linalg.norm(Q, axis=1)[:, None] * np.linalg.norm(D, axis=1)[None, :]).cuda()
            slack_hi = float(((U - ex) / bound).min())
            slack_lo = float(((ex - (U - 2 * e)) / bound).min())
            print(f"H {H} {kind}: e / (|q||d|) = {float((e / bound).mean()):.2e}; min (U - S_x) / |q||d| = {slack_hi:.2e}, "
                  f"min (S_x - U + 2e) / |q||d| = {slack_lo:.2e}")
            assert slack_hi >= 0 and slack_lo >= 0
            f = DenseIndexHIP(H)
            f.set_precision("fp32_filtered")
            f.add_host_rows(D)
            fs, fi = f.search(q, 50)
            assert torch.equal(fs, es[:, :50]) and torch.equal(fi, ei[:, :50])
            assert f.filter_query_stats() == (128, 0), (H, kind, f.filter_query_stats())      # no pair violated the bound on the device


def test_filtered_with_tiny_norm_rows_next_to_a_large_absmax():
    """the filter's fp16 plane is scaled by ONE power of two per segment (from the segment's absmax), so rows 2^-20 times
    smaller land in fp16's subnormal range (or flush to zero) - their plane entries carry almost no bits and the bound has to say
    so through the per-document residual norms.  A segment mixing unit-scale rows, rows at 2^-12 and rows at 2^-20 of the absmax
    (some of them the best matches of queries that point at them), one huge outlier element: ids and fp32 scores equal the exact
    kernel's and the oracle's."""
    rng = np.random.default_rng(77)
    H, N, nq, k = 256, 20000, 130, 200
    D = (rng.standard_normal((N, H), dtype=np.float32) * (0.5 / np.sqrt(H))).astype(np.float32)
    D[1000:6000] *= np.float32(2.0 ** -12)
    D[6000:12000] *= np.float32(2.0 ** -20)
    D[17, 3] = 900.0                                   # the segment's absmax: pushes everything else 2^11 further down the plane
    Q = (rng.standard_normal((nq, H), dtype=np.float32) * (0.5 / np.sqrt(H))).astype(np.float32)
    Q[:20] = D[6000:6020] * np.float32(2.0 ** 20)      # queries aligned with tiny rows: those rows' scores are tiny but positive
    Q[20:40] = -Q[20:40] + D[2000:2020] * np.float32(2.0 ** 13)
    Q[40:50, 3] = -1.0                                 # the outlier row is the worst match for these
    (es, ei), (fs, fi), filt = _both([D], Q, k)
    assert torch.equal(fi, ei) and torch.equal(fs, es)
    cert, redone = filt.filter_query_stats()
    assert cert + redone == nq
    os_, oi = O.topk_rows(O.dense_scores_fma(Q[:50], D, O.mfma_korder(H)), k)
    assert np.array_equal(fi[:50].cpu().numpy(), oi) and np.array_equal(fs[:50].cpu().numpy(), os_)
    # the same rows as a segment of their own get their own scale: still exact
    (es2, ei2), (fs2, fi2), _ = _both([D[:6000], D[6000:12000], D[12000:]], Q, k)
    assert torch.equal(fi2, ei2) and torch.equal(fs2, es2) and torch.equal(es2, es) and torch.equal(ei2, ei)


def test_second_threshold_mixed_norm_segments(monkeypatch):
    """The upper-bound pass drops a pair when its bound lies under the kp-th largest bound so far OR under U_(k) - 2 e_max(q), the k-th
    largest bound less twice the largest error term any document of the index can have (dense_filter.h).  Segments of very different
    norms (e_max comes from the large one, the scores that matter partly from the small one), documents of zero norm, near-ties at the
    cut and k close to N: the result equals the exact kernel's bit for bit, with the second threshold (default) and without it."""
    from scaling_retriever_amd.scoring import DenseIndexHIP
    rng = np.random.default_rng(11)
    H, nq = 256, 160
    small = (rng.standard_normal((40000, H), dtype=np.float32) * 1e-3).astype(np.float32)
    large = rng.standard_normal((25000, H), dtype=np.float32)
    large[::7] *= np.float32(30.0)                                   # a few documents carry the segment's largest x and y
    large[5:2000:5] = 0
    mid = (rng.standard_normal((9000, H), dtype=np.float32) * 0.05).astype(np.float32)
    Q = rng.standard_normal((nq, H), dtype=np.float32)
    Q[1] = -large[7] / np.float32(30.0)                              # the big documents at the BOTTOM of this query's ranking
    Q[2] = small[11] * np.float32(1e3)
    for j in range(300):                                             # near-ties around rank k in the small segment
        small[20000 + j] = small[11] * np.float32(1.0 - 1e-6 * j)
    parts = [small, large, mid]
    for k in (10, 1000, 3000):
        exact = DenseIndexHIP(H)
        for p_ in parts:
            exact.add_host_rows(p_)
        es, ei = exact.search(torch.from_numpy(Q).cuda(), k)
        for on in ("1", "0"):
            monkeypatch.setenv("SR_DEV_SWITCHES", "1")
            monkeypatch.setenv("SR_FILTER_TAU2", on)
            filt = DenseIndexHIP(H)
            filt.set_precision("fp32_filtered")
            for p_ in parts:
                filt.add_host_rows(p_)
            fs, fi = filt.search(torch.from_numpy(Q).cuda(), k)
            assert torch.equal(fi, ei) and torch.equal(fs, es), (k, on)
            assert sum(filt.filter_stats()) == 1
