"""GPU: edge cases the reference's callers can produce - ragged / tiny / maximum sizes, long sequences
(BEIR uses 512 tokens, scripts/beir/eval_beir_dense.sh:23-24), empty posting lists, k = 1."""
import json
import os

import numpy as np
import pytest
import torch

from golden_weights import make_weights
from oracle import llama_bi as LB
from oracle import scoring as SC

pytestmark = pytest.mark.gpu


def rel(a, b):
    return float(np.linalg.norm(np.asarray(a, np.float64) - b) / max(np.linalg.norm(b), 1e-30))


@pytest.fixture(scope="module")
def tiny(golden_dir):
    z = np.load(os.path.join(golden_dir, "enc_hd64.npz"))
    cfg = json.loads(str(z["config_json"]))
    return cfg, make_weights(cfg, int(z["weight_seed"]))


def test_long_sequences_use_the_general_attention_path(tiny):
    """Sequences of 300 and 512 tokens (> the 256-token fast path) mixed with short ones, dense and sparse heads."""
    from scaling_retriever_amd.modeling.llm_encoder import LlamaBiDense, LlamaBiSparse
    cfg, w = tiny
    rng = np.random.default_rng(0)
    lens, L = [512, 300, 7, 257], 512
    ids = np.full((4, L), cfg["vocab_size"] - 1, np.int64)
    mask = np.zeros((4, L), np.int64)
    for r, n in enumerate(lens):
        ids[r, L - n:] = rng.integers(0, cfg["vocab_size"], size=n)
        mask[r, L - n:] = 1
    t_ids, t_mask = torch.from_numpy(ids).cuda(), torch.from_numpy(mask).cuda()
    d = LlamaBiDense.from_weights(cfg, w, precision="bf16").to("cuda").encode(input_ids=t_ids, attention_mask=t_mask).cpu().numpy()
    s = LlamaBiSparse.from_weights(cfg, w, precision="bf16").to("cuda").encode(input_ids=t_ids, attention_mask=t_mask).cpu().numpy()
    assert rel(d, LB.dense_encode(w, cfg, ids, mask)) < 1.5e-2
    assert rel(s, LB.sparse_encode(w, cfg, ids, mask)) < 1.5e-2


def test_single_token_single_row_batch(tiny):
    from scaling_retriever_amd.modeling.llm_encoder import LlamaBiDense
    cfg, w = tiny
    ids, mask = np.array([[5]], np.int64), np.array([[1]], np.int64)
    out = LlamaBiDense.from_weights(cfg, w, precision="bf16").to("cuda").encode(input_ids=torch.from_numpy(ids).cuda(),
                                                               attention_mask=torch.from_numpy(mask).cuda())
    assert rel(out.cpu().numpy(), LB.dense_encode(w, cfg, ids, mask)) < 1.5e-2


def test_batch_larger_than_workspace_is_split(tiny):
    """B x L above max_batch_tokens: the wrapper feeds the engine in row slabs; results do not depend on the split."""
    from scaling_retriever_amd.modeling.llm_encoder import LlamaBiDense
    cfg, w = tiny
    rng = np.random.default_rng(1)
    B, L = 40, 24
    ids = rng.integers(0, cfg["vocab_size"], size=(B, L)).astype(np.int64)
    mask = np.ones((B, L), np.int64)
    mask[::3, :10] = 0                                   # some left padding
    big = LlamaBiDense.from_weights(cfg, w, max_batch_tokens=4096, precision="bf16").to("cuda")
    small = LlamaBiDense.from_weights(cfg, w, max_batch_tokens=256, max_batch_seqs=16, precision="bf16").to("cuda")   # 10 rows per call
    a = big.encode(input_ids=torch.from_numpy(ids).cuda(), attention_mask=torch.from_numpy(mask).cuda())
    b = small.encode(input_ids=torch.from_numpy(ids).cuda(), attention_mask=torch.from_numpy(mask).cuda())
    assert torch.allclose(a, b, rtol=0, atol=2e-3)
    with pytest.raises(ValueError):
        small.encode(input_ids=torch.zeros((1, 300), dtype=torch.int64).cuda(), attention_mask=torch.ones((1, 300), dtype=torch.int64).cuda())


def test_dense_search_tiny_and_k1():
    from scaling_retriever_amd.scoring import DenseIndexHIP
    rng = np.random.default_rng(2)
    for nq, n, h, k in [(1, 1, 64, 1), (3, 5, 64, 1), (70, 17, 128, 3), (2, 300, 256, 300)]:
        Q, D = rng.standard_normal((nq, h), dtype=np.float32), rng.standard_normal((n, h), dtype=np.float32)
        idx = DenseIndexHIP(h)
        idx.add_host_rows(D)
        s, i = idx.search(torch.from_numpy(Q).cuda(), k)
        es, ei = SC.topk_rows(SC.dense_scores_fma(Q, D, SC.dense_korder(nq, h)), k)
        assert np.array_equal(i.cpu().numpy(), ei) and np.array_equal(s.cpu().numpy(), es), (nq, n, h, k)
    empty = DenseIndexHIP(64)
    s, i = empty.search(torch.zeros((2, 64), device="cuda"), 4)
    assert (i.cpu().numpy() == -1).all()
    s, i = idx.search(torch.zeros((0, 256), device="cuda"), 4)
    assert tuple(s.shape) == (0, 4)


def test_sparse_search_empty_and_degenerate():
    from scaling_retriever_amd.scoring import SparseIndexHIP
    V, N = 20, 9000
    indptr = np.zeros(V + 1, np.int64)
    empty = SparseIndexHIP(indptr, np.zeros(0, np.int32), np.zeros(0, np.float32), N)
    s, i, c = empty.search(np.array([0, 2], np.int64), np.array([1, 3], np.int32), np.ones(2, np.float32), 5)
    assert int(c.item()) == 0 and (i.cpu().numpy() == -1).all()
    # one posting in the last doc of the last (partial) tile; a query without terms; k = 1
    indptr[8:] = 1
    one = SparseIndexHIP(indptr, np.array([N - 1], np.int32), np.array([2.0], np.float32), N)
    s, i, c = one.search(np.array([0, 1, 1], np.int64), np.array([7], np.int32), np.array([1.5], np.float32), 1)
    assert c.cpu().tolist() == [1, 0] and int(i[0, 0]) == N - 1 and float(s[0, 0]) == 3.0


def test_search_handles_serialise_concurrent_callers():
    """The reference drives retrieval from a 4-thread ThreadPoolExecutor (indexer.py:459).  A search handle owns one
    top-k workspace, so concurrent sr_dense_search / sr_sparse_search calls on the SAME handle must serialise inside the
    library: four threads, each on its own stream, get exactly the single-threaded answers."""
    import threading
    from scaling_retriever_amd.scoring import DenseIndexHIP, SparseIndexHIP
    g = torch.Generator(device="cuda").manual_seed(21)
    N, H, V = 30000, 128, 500
    D = torch.randn((N, H), device="cuda", generator=g)
    dense = DenseIndexHIP(H)
    dense.add_device_rows(D)
    Qs = [torch.randn((n, H), device="cuda", generator=g) for n in (7, 150, 64, 300)]
    want_dense = [dense.search(Q, 25) for Q in Qs]
    # sparse index: random postings, doc ids ascending inside each term
    term = torch.randint(0, V, (200000,), device="cuda", generator=g)
    doc = torch.randint(0, N, (200000,), device="cuda", generator=g)
    key = torch.unique(term.long() * N + doc.long())
    t_of = torch.div(key, N, rounding_mode="floor")
    ids = (key - t_of * N).int().contiguous()
    vals = torch.rand(ids.numel(), device="cuda", generator=g)
    indptr = torch.zeros(V + 1, dtype=torch.int64, device="cuda")
    indptr[1:] = torch.cumsum(torch.bincount(t_of, minlength=V), 0)
    sparse = SparseIndexHIP(indptr, ids, vals, N)
    qsets = []
    for nq in (3, 40, 11, 90):
        cols = torch.stack([torch.randperm(V, device="cuda", generator=g)[:12].sort().values for _ in range(nq)]).int().reshape(-1)
        qsets.append((torch.arange(0, nq * 12 + 1, 12, device="cuda", dtype=torch.int64), cols.contiguous(),
                      torch.rand(nq * 12, device="cuda", generator=g)))
    want_sparse = [sparse.search(*q, 20) for q in qsets]
    torch.cuda.synchronize()
    got_dense, got_sparse, errors = [None] * 4, [None] * 4, []

    def work(t):
        try:
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                for _ in range(3):
                    got_dense[t] = dense.search(Qs[t], 25)
                    got_sparse[t] = sparse.search(*qsets[t], 20)
            st.synchronize()
        except Exception as e:          # noqa: BLE001 - surfaced below
            errors.append(e)

    threads = [threading.Thread(target=work, args=(t,)) for t in range(4)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
    for t in range(4):
        assert all(torch.equal(a, b) for a, b in zip(got_dense[t], want_dense[t]))
        assert all(torch.equal(a, b) for a, b in zip(got_sparse[t], want_sparse[t]))
