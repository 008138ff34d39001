"""SR_PRECISION_FP32_FILTERED: sr_dense_search through the certified bf16 filter + exact re-score must return EXACTLY what
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


def test_filtered_falls_back_when_it_cannot_certify():
    """More than kp - k documents tie with the k-th one (exact duplicates): no certificate -> the exact kernel answers, the
    result is still the exact kernel's (ties by ascending doc index)."""
    rng = np.random.default_rng(3)
    H, k = 128, 50
    base = rng.standard_normal((8, H), dtype=np.float32)
    D = np.repeat(base, 1500, axis=0)[rng.permutation(12000)]                      # 1 499 exact twins each: more than kp - k
    Q = rng.standard_normal((100, H), dtype=np.float32)
    (es, ei), (fs, fi), filt = _both([D], Q, k)
    assert torch.equal(fi, ei) and torch.equal(fs, es)
    assert filt.filter_stats() == (0, 1)
    # a zero query: every score is 0, nothing can be separated
    Q[5] = 0
    (es, ei), (fs, fi), filt = _both([rng.standard_normal((5000, H), dtype=np.float32)], Q, k)
    assert torch.equal(fi, ei) and torch.equal(fs, es) and filt.filter_stats() == (0, 1)
    # non-finite data: never certified
    Dn = rng.standard_normal((5000, H), dtype=np.float32)
    Dn[17, 3] = np.inf
    _, (fs, fi), filt = _both([Dn], rng.standard_normal((100, H), dtype=np.float32), k)
    assert filt.filter_stats() == (0, 1)


def test_filter_raises_its_plane_products_when_one_is_not_enough():
    """Score gaps between rank k and rank kp that the one-product bound (2^-8 |q||d|) cannot separate but the two-product
    bound (2^-9) can: the first search fails its certificate once, switches the index to two products for good, and is
    answered by the filter - identical to the exact kernel - without the exact kernel running."""
    from scaling_retriever_amd.scoring import DenseIndexHIP
    rng = np.random.default_rng(21)
    H, N, nq, k = 256, 30000, 100, 100                       # kp = k + 2048
    c = 2.0 ** -9 * 1.004 + 1.25 * (3.0 * H * 2.0 ** -24) + 1.0e-5
    c1 = c + 2.0 ** -9 * 1.004
    # every document = s_i * u + small noise, queries = u: scores = s_i |u|^2 (+ noise), descending in i with a relative step
    # chosen so that the gap between rank k and rank kp is 1.5 c |q|: above E = c |q| (two products certify), below
    # E1 = c1 |q| ~ 2 c |q| (one product does not)
    u = rng.standard_normal(H).astype(np.float32)
    u /= np.linalg.norm(u)
    kp = k + 2048
    assert c < 1.5 * c < c1
    step = 1.5 * c / (kp - k)
    s = (1.0 - step * np.arange(N)).astype(np.float32)
    s[s < 0.2] = 0.2
    D = (s[:, None] * u[None, :]).astype(np.float32)
    Q = np.repeat(u[None, :], nq, axis=0).astype(np.float32) * rng.uniform(0.5, 2.0, size=(nq, 1)).astype(np.float32)
    exact = DenseIndexHIP(H)
    exact.add_host_rows(D)
    filt = DenseIndexHIP(H)
    filt.set_precision("fp32_filtered")
    filt.add_host_rows(D)
    q = torch.from_numpy(Q).cuda()
    es, ei = exact.search(q, k)
    assert filt.filter_products() == (1, 0)
    fs, fi = filt.search(q, k)
    assert torch.equal(fi, ei) and torch.equal(fs, es)
    assert filt.filter_products() == (2, 1), filt.filter_products()
    assert filt.filter_stats() == (1, 0)
    fs, fi = filt.search(q, k)                                # stays on two products
    assert torch.equal(fi, ei) and torch.equal(fs, es)
    assert filt.filter_products() == (2, 1) and filt.filter_stats() == (2, 0)


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


def test_filter_error_bound_dominates_the_filter_score():
    """The certificate's E = c(H) |q| |d| must dominate |S_a - S_x| for S_a = (q0 + q1) . d0 (two bf16 planes of the query, one
    of the document) and, with c1(H), for S_a = q0 . d0 (the one-product pass tried first).  S_a is restated in float64 from the planes (the MFMA's fp32 accumulation adds at most what the bound
    reserves for it and is checked on the device for every re-scored pair), S_x comes from the exact kernel; Gaussian,
    same-sign and adversarially aligned data (every rounding error pushed the same way)."""
    from scaling_retriever_amd.scoring import DenseIndexHIP
    rng = np.random.default_rng(11)
    for H in (256, 2048):
        c = 2.0 ** -9 * 1.004 + 1.25 * (3.0 * H * 2.0 ** -24) + 1.0e-5
        c1 = 2.0 ** -8 * 1.004 + 1.25 * (3.0 * H * 2.0 ** -24) + 1.0e-5
        for kind in ("gauss", "same_sign", "aligned"):
            D = rng.standard_normal((2000, H), dtype=np.float32)
            Q = rng.standard_normal((128, H), dtype=np.float32)
            if kind != "gauss":
                D, Q = np.abs(D), np.abs(Q)
            if kind == "aligned":            # values just below a bf16 rounding boundary: d - d0 is -2^-9 |d| for every element
                D = (np.float32(1.0) + np.float32(2.0 ** -8) * np.float32(0.499)) * np.exp2(rng.integers(-3, 3, size=D.shape)).astype(np.float32)
            a = DenseIndexHIP(H)
            a.add_host_rows(D)
            q = torch.from_numpy(Q).cuda()
            es, ei = a.search(q, 2000)
            ex = torch.zeros((128, 2000), device="cuda").scatter_(1, ei, es).double()
            d0 = torch.from_numpy(D).cuda().bfloat16().double()
            q0 = q.bfloat16()
            q1 = (q - q0.float()).bfloat16()
            sa = (q0.double() + q1.double()) @ d0.T
            bound = torch.from_numpy(np.linalg.norm(Q, axis=1)[:, None] * np.linalg.norm(D, axis=1)[None, :]).cuda()
            worst = float(((ex - sa).abs() / bound).max())
            worst1 = float(((ex - q0.double() @ d0.T).abs() / bound).max())
            print(f"H {H} {kind}: max |S_a - S_x| / (|q||d|) = {worst:.2e} (bound c = {c:.2e}); one product {worst1:.2e} (c1 = {c1:.2e})")
            assert worst < c and worst1 < c1
