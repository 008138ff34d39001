"""GPU property tests (hypothesis): sr_dense_search / sr_sparse_search through the C ABI against the oracle on random
shapes - ragged tiles, every kernel regime (streaming <= 64 queries, 128- and 256-query tiles, the template fallbacks for
dims that are not a multiple of 256), k above and below the corpus size, ties.  Bit-exact ids and fp32 scores."""
import numpy as np
import pytest
import torch
from hypothesis import given, settings, strategies as st

from oracle import scoring as SC

pytestmark = pytest.mark.gpu
SET = settings(max_examples=30, deadline=None)


@SET
@given(st.integers(0, 2 ** 31 - 1), st.sampled_from([1, 3, 17, 64, 65, 100, 128, 129, 200, 257, 300]), st.integers(1, 3000),
       st.sampled_from([16, 48, 64, 128, 256, 272, 512]), st.integers(1, 70), st.booleans())
def test_dense_search_random_shapes(seed, nq, n, h, k, ties):
    from scaling_retriever_amd.scoring import DenseIndexHIP
    rng = np.random.default_rng(seed)
    D = rng.standard_normal((n, h), dtype=np.float32)
    if ties and n > 4:
        D[n // 2] = D[1]
        D[n - 1] = D[1]
    Q = rng.standard_normal((nq, h), dtype=np.float32)
    idx = DenseIndexHIP(h)
    idx.add_device_rows(torch.from_numpy(D).cuda())
    s, i = idx.search(torch.from_numpy(Q).cuda(), k)
    s, i = s.cpu().numpy(), i.cpu().numpy()
    m = min(k, n)
    es, ei = SC.topk_rows(SC.dense_scores_fma(Q, D, SC.dense_korder(nq, h)), m)
    assert np.array_equal(i[:, :m], ei) and np.array_equal(s[:, :m], es)
    assert np.all(i[:, m:] == -1)
    idx.close()


@SET
@given(st.integers(0, 2 ** 31 - 1), st.integers(1, 60), st.integers(1, 20000), st.floats(0.0, 0.3), st.integers(1, 40),
       st.integers(1, 25), st.integers(1, 50), st.sampled_from([0.0, 0.7]))
def test_sparse_search_random_shapes(seed, V, N, density, nq, nterms, k, threshold):
    from scaling_retriever_amd.scoring import SparseIndexHIP
    rng = np.random.default_rng(seed)
    lists = [np.sort(rng.choice(N, size=rng.binomial(N, density), replace=False)).astype(np.int32) for _ in range(V)]
    indptr = np.concatenate([[0], np.cumsum([len(x) for x in lists])]).astype(np.int64)
    doc_ids = np.concatenate(lists).astype(np.int32)
    vals = (rng.random(len(doc_ids), dtype=np.float32) * 3).astype(np.float32)
    nterms = min(nterms, V)
    cols = np.concatenate([np.sort(rng.choice(V, size=nterms, replace=False)) for _ in range(nq)]).astype(np.int32)
    qv = (rng.random(nq * nterms, dtype=np.float32) * 2).astype(np.float32)
    q_indptr = np.arange(0, nq * nterms + 1, nterms, dtype=np.int64)
    index = SparseIndexHIP(torch.from_numpy(indptr).cuda(), torch.from_numpy(doc_ids).cuda(), torch.from_numpy(vals).cuda(), N)
    s, i, c = index.search(torch.from_numpy(q_indptr).cuda(), torch.from_numpy(cols).cuda(), torch.from_numpy(qv).cuda(), k,
                           threshold=threshold)
    oi, os_, oc = SC.sparse_retrieve_c(indptr, doc_ids, vals, q_indptr, cols, qv, k, threshold, N, q_threads=2)
    s, i, c = s.cpu().numpy(), i.cpu().numpy(), c.cpu().numpy()
    assert np.array_equal(c, oc)
    for q in range(nq):
        assert np.array_equal(i[q, :c[q]], oi[q, :c[q]]) and np.array_equal(s[q, :c[q]], os_[q, :c[q]])
    index.close()


@settings(max_examples=60, deadline=None)
@given(st.integers(0, 2 ** 31 - 1), st.integers(20, 400), st.integers(1, 9000), st.floats(0.002, 0.25), st.integers(1, 70),
       st.integers(1, 40), st.sampled_from([1, 7, 100, 1000, 3000]), st.sampled_from([0.0, 0.7]), st.booleans())
def test_certified_sparse_scorer_random_shapes(monkeypatch_module, seed, V, N, density, nq, nterms, k, threshold, zeros):
    """The certified two-stage scorer forced on (it is sized for collections of >= 64 tiles): collections of one ragged tile up to nine,
    term laws with and without lists that cover every doc, queries with zero-valued terms, k above the collection size, a positive
    threshold - every row must be the oracle's, whichever of its paths (certified, handed back to the exact kernels) served it."""
    from scaling_retriever_amd.scoring import SparseIndexHIP
    rng = np.random.default_rng(seed)
    dens = np.minimum(1.0, density * 8.0 / np.arange(1, V + 1) ** 0.7)            # a few heavy terms, a long tail
    lists = [np.sort(rng.choice(N, size=rng.binomial(N, d_), replace=False)).astype(np.int32) for d_ in dens]
    indptr = np.concatenate([[0], np.cumsum([len(x) for x in lists])]).astype(np.int64)
    doc_ids = np.concatenate(lists).astype(np.int32)
    if len(doc_ids) == 0:
        return
    vals = np.log1p(rng.random(len(doc_ids)) * 20).astype(np.float32)
    nterms = min(nterms, V)
    cols = np.concatenate([np.sort(rng.choice(V, size=nterms, replace=False)) for _ in range(nq)]).astype(np.int32)
    qv = np.log1p(rng.random(nq * nterms) * 20).astype(np.float32)
    if zeros:
        qv[rng.random(len(qv)) < 0.2] = 0.0
    q_indptr = np.arange(0, nq * nterms + 1, nterms, dtype=np.int64)
    index = SparseIndexHIP(torch.from_numpy(indptr).cuda(), torch.from_numpy(doc_ids).cuda(), torch.from_numpy(vals).cuda(), N)
    s, i, c = index.search(torch.from_numpy(q_indptr).cuda(), torch.from_numpy(cols).cuda(), torch.from_numpy(qv).cuda(), k, threshold=threshold)
    oi, os_, oc = SC.sparse_retrieve_c(indptr, doc_ids, vals, q_indptr, cols, qv, k, threshold, N, q_threads=2)
    s, i, c = s.cpu().numpy(), i.cpu().numpy(), c.cpu().numpy()
    assert np.array_equal(c, oc)
    for q in range(nq):
        assert np.array_equal(i[q, :c[q]], oi[q, :c[q]]) and np.array_equal(s[q, :c[q]], os_[q, :c[q]]), q
    st_ = index.cert_stats()
    assert st_["present"] == 1 and (st_["searches"] == 1 or k + 1024 > 4096)
    index.close()


@pytest.fixture(scope="module")
def monkeypatch_module():
    """SR_SPARSE_CERT=1 (dev switch) for the whole module's forced-scorer tests: hypothesis re-enters the test function many times, and a
    function-scoped monkeypatch may not be combined with @given."""
    import os
    old = {k_: os.environ.get(k_) for k_ in ("SR_SPARSE_CERT", "SR_DEV_SWITCHES")}
    os.environ["SR_SPARSE_CERT"], os.environ["SR_DEV_SWITCHES"] = "1", "1"
    yield
    for k_, v in old.items():
        if v is None:
            os.environ.pop(k_, None)
        else:
            os.environ[k_] = v


@pytest.fixture(scope="module")
def tiny_models(golden_dir):
    import json
    import os
    from golden_weights import make_weights
    from scaling_retriever_amd.modeling.llm_encoder import LlamaBiDense, LlamaBiSparse
    z = np.load(os.path.join(golden_dir, "enc_tiny_a.npz"))
    cfg = json.loads(str(z["config_json"]))
    w = make_weights(cfg, 99)
    return cfg, w, LlamaBiDense.from_weights(cfg, w, precision="bf16").to("cuda").eval(), LlamaBiSparse.from_weights(cfg, w, precision="bf16").to("cuda").eval()


@settings(max_examples=12, deadline=None)
@given(st.integers(0, 2 ** 31 - 1), st.integers(1, 10), st.integers(1, 40), st.sampled_from(["left", "right"]))
def test_encoder_random_batches(tiny_models, seed, B, max_len, side):
    """Ragged random batches (length-1 sequences, either padding side) through both heads: within the encoder tolerance of
    the fp32 oracle, and a row's output does not depend on what else is in the batch (bitwise)."""
    from oracle import llama_bi as LB
    cfg, w, dense, sparse = tiny_models
    rng = np.random.default_rng(seed)
    lens = rng.integers(1, max_len + 1, size=B)
    S = int(lens.max())
    ids = np.zeros((B, S), np.int64)
    mask = np.zeros((B, S), np.int64)
    for b, l in enumerate(lens):
        sl = slice(S - l, S) if side == "left" else slice(0, l)
        ids[b, sl] = rng.integers(3, cfg["vocab_size"], size=l)
        mask[b, sl] = 1
    t_ids, t_mask = torch.from_numpy(ids).cuda(), torch.from_numpy(mask).cuda()
    for model, fn in ((dense, LB.dense_encode), (sparse, LB.sparse_encode)):
        out = model.doc_encode(input_ids=t_ids, attention_mask=t_mask)
        ref = fn(w, cfg, ids, mask)
        got = out.cpu().numpy()
        err = np.linalg.norm(got.astype(np.float64) - ref) / max(np.linalg.norm(ref), 1e-30)
        assert err < 1.5e-2, err
        # row 0 alone (same padding inside its own row span) gives the same bits as inside the batch
        alone = model.doc_encode(input_ids=t_ids[:1].contiguous(), attention_mask=t_mask[:1].contiguous())
        assert torch.equal(alone[0], out[0])


@settings(max_examples=40, deadline=None)
@given(st.integers(0, 2 ** 31 - 1), st.integers(1, 2600), st.integers(1, 40), st.integers(1, 12), st.sampled_from([0, 1, 2, 4]),
       st.sampled_from(["", "", "128", "256", "split:256", "split:1024"]))
def test_gemm_random_shapes(seed, M, n32, k64, epi, tile):
    """sr_gemm_bf16 on random shapes (any M, N = 32 n, K = 64 k) for the store / residual / SwiGLU epilogues and every
    tiling (automatic plan, forced tiles, forced row cuts) against a plain PyTorch fp32 reference of the same op."""
    import os
    from scaling_retriever_amd import _lib as L
    lib = L.load()
    old = os.environ.get("SR_GEMM_TILE")
    os.environ["SR_GEMM_TILE"] = tile
    try:
        N, K = 32 * n32, 64 * k64
        g = torch.Generator(device="cuda").manual_seed(seed % (2 ** 31))
        A = torch.randn((M, K), device="cuda", generator=g).bfloat16()
        W = (torch.randn((N, K), device="cuda", generator=g) / K ** 0.5).bfloat16()
        ref = A.float() @ W.float().T
        if epi == 0:
            C = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
        elif epi == 1:
            C = torch.randn((M, N), device="cuda", generator=g)
            want = C + ref
        elif epi == 2:
            C = torch.empty((M, N // 2), dtype=torch.bfloat16, device="cuda")
        else:
            C = torch.empty((M, N), dtype=torch.float32, device="cuda")
        L.check(lib.sr_gemm_bf16(A.data_ptr(), W.data_ptr(), M, N, K, epi, C.data_ptr(), None, L.stream_ptr()), "sr_gemm_bf16")
        torch.cuda.synchronize()
        if epi == 0:
            torch.testing.assert_close(C.float(), ref, rtol=1e-2, atol=1e-2)
        elif epi == 1:
            torch.testing.assert_close(C, want, rtol=1e-4, atol=1e-4)
        elif epi == 2:
            # gate / up rows are interleaved in 16-row blocks: block 2b = gate rows, block 2b + 1 = up rows
            y = ref.reshape(M, N // 32, 2, 16)
            sw = (torch.nn.functional.silu(y[:, :, 0]) * y[:, :, 1]).reshape(M, N // 2)
            torch.testing.assert_close(C.float(), sw, rtol=2e-2, atol=2e-2)
        else:
            torch.testing.assert_close(C, ref, rtol=1e-4, atol=1e-4)
    finally:
        if old is None:
            os.environ.pop("SR_GEMM_TILE", None)
        else:
            os.environ["SR_GEMM_TILE"] = old
