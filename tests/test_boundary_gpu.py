"""GPU: the drop-in boundary at speed keeps the reference's results.

  * LlamaBi*.encode_batches (sr_encode_rows): several collator batches in one engine pass == the batch-by-batch encode calls of
    DenseRetriever.generate_query_vecs (/root/reference/eval_dense.py:94-106) and SparseRetrieval._generate_query_vecs
    (/root/reference/scaling_retriever/indexer.py:382-403), bit for bit - every row keeps the position_ids of its own batch;
  * a sequence gets the same bits alone and next to a longer one (per-sequence kernel choice of the fp32 attention);
  * DenseFlatIndexer.search_knn maps ids with one take (indexer.py:210-214), LocalFaissDenseRetriever.write_run writes the file
    the reference's loop + json.dump writes (eval_dense.py:225-241);
  * sr_dense_search_begin's lower bound is an exact score that ceil(k / W) documents of the shard reach.
"""
import json
import os

import numpy as np
import pytest
import torch

from golden_weights import make_weights

pytestmark = pytest.mark.gpu

CFG_1B_2L = {"hidden_size": 2048, "intermediate_size": 8192, "num_attention_heads": 32, "num_key_value_heads": 8, "head_dim": 64,
             "num_hidden_layers": 2, "vocab_size": 128256, "rms_norm_eps": 1e-5, "rope_theta": 500000.0,
             "tie_word_embeddings": True, "max_position_embeddings": 512}


def _batches(rng, vocab, sizes, lo, hi, side="left"):
    """Collator-shaped batches (pad to the batch's longest row)."""
    out = []
    for n in sizes:
        lens = rng.integers(lo, hi + 1, size=n)
        L = int(lens.max())
        ids = np.zeros((n, L), np.int64)
        mask = np.zeros((n, L), np.int64)
        for r, l in enumerate(lens):
            sl = slice(L - l, L) if side == "left" else slice(0, l)
            ids[r, sl] = rng.integers(3, vocab, size=l)
            mask[r, sl] = 1
        out.append({"input_ids": torch.from_numpy(ids).cuda(), "attention_mask": torch.from_numpy(mask).cuda(),
                    "ids": [f"r{len(out)}_{i}" for i in range(n)]})
    return out


@pytest.fixture(scope="module")
def tiny(golden_dir):
    z = np.load(os.path.join(golden_dir, "enc_tiny_a.npz"))
    cfg = json.loads(str(z["config_json"]))
    return cfg, make_weights(cfg, int(z["weight_seed"]))


@pytest.fixture(scope="module")
def wide():
    return CFG_1B_2L, make_weights(CFG_1B_2L, 99, embed_std=0.05)


def _check_encode_batches(cls, cfg, w, batches, autocast, **kw):
    model = cls.from_weights(cfg, w, **kw).to("cuda").eval()
    ctx = torch.autocast("cuda", dtype=torch.bfloat16) if autocast else torch.autocast("cuda", enabled=False)
    with torch.inference_mode(), ctx:
        one_by_one = [model.encode(input_ids=b["input_ids"], attention_mask=b["attention_mask"]) for b in batches]
        together = model.encode_batches(batches)
    if isinstance(together, tuple):
        for h in range(len(together)):
            assert torch.equal(together[h], torch.cat([o[h] for o in one_by_one])), f"head {h}"
    else:
        assert torch.equal(together, torch.cat(one_by_one))
    del model
    torch.cuda.empty_cache()


def test_encode_batches_is_batch_by_batch_encode_bit_for_bit_tiny(tiny):
    from scaling_retriever_amd.modeling.llm_encoder import LlamaBiDense, LlamaBiHybrid, LlamaBiSparse
    cfg, w = tiny
    rng = np.random.default_rng(5)
    bl = _batches(rng, cfg["vocab_size"], [4, 7, 1, 5, 3], 1, 20)
    _check_encode_batches(LlamaBiDense, cfg, w, bl, autocast=False)          # the dense-query regime (fp32)
    _check_encode_batches(LlamaBiDense, cfg, w, bl, autocast=True)
    _check_encode_batches(LlamaBiSparse, cfg, w, bl, autocast=True)          # the sparse-query regime
    _check_encode_batches(LlamaBiSparse, cfg, w, _batches(rng, cfg["vocab_size"], [3, 6, 2], 1, 20, side="right"), autocast=True)
    _check_encode_batches(LlamaBiHybrid, cfg, w, bl, autocast=True)
    # a token budget smaller than the group: the engine cuts the rows itself
    _check_encode_batches(LlamaBiDense, cfg, w, bl, autocast=False, max_batch_tokens=64, max_batch_seqs=6)


def test_encode_batches_at_1b_width_query_shape(wide):
    """55 loader batches of 128 queries are what the reference's retrieval drivers produce for MS MARCO Dev; here 12 of them, at
    the production layer width, in both regimes: one engine pass returns the per-batch bits."""
    from scaling_retriever_amd.modeling.llm_encoder import LlamaBiDense, LlamaBiSparse
    cfg, w = wide
    rng = np.random.default_rng(6)
    bl = _batches(rng, cfg["vocab_size"], [128] * 11 + [70], 4, 40)
    _check_encode_batches(LlamaBiDense, cfg, w, bl, autocast=False, max_batch_tokens=32768, max_batch_seqs=2048)
    _check_encode_batches(LlamaBiSparse, cfg, w, bl[:5], autocast=True, max_batch_tokens=32768, max_batch_seqs=2048)


def test_a_short_sequence_gets_the_same_bits_next_to_a_long_one(wide):
    """fp32 regime: sequences of <= 64 tokens run the fp32-MFMA attention kernel, longer ones the FMA kernel - chosen per
    SEQUENCE, so a short passage alone and next to a 100-token passage (token-budget batches mix lengths) encode identically."""
    from scaling_retriever_amd.modeling.llm_encoder import LlamaBiDense
    cfg, w = wide
    model = LlamaBiDense.from_weights(cfg, w, precision="fp32").to("cuda").eval()
    rng = np.random.default_rng(7)
    short = rng.integers(3, cfg["vocab_size"], size=(3, 30))
    long_ = rng.integers(3, cfg["vocab_size"], size=(2, 100))
    ids = np.zeros((5, 100), np.int64)
    mask = np.zeros((5, 100), np.int64)
    ids[:3, 70:], mask[:3, 70:] = short, 1
    ids[3:], mask[3:] = long_, 1
    with torch.inference_mode():
        mixed = model.encode(input_ids=torch.from_numpy(ids).cuda(), attention_mask=torch.from_numpy(mask).cuda())
        alone_s = model.encode(input_ids=torch.from_numpy(ids[:3]).cuda(), attention_mask=torch.from_numpy(mask[:3]).cuda())
        alone_l = model.encode(input_ids=torch.from_numpy(ids[3:]).cuda(), attention_mask=torch.from_numpy(mask[3:]).cuda())
    assert torch.equal(mixed[:3], alone_s) and torch.equal(mixed[3:], alone_l)


def test_search_knn_and_write_run_match_the_reference_loop(tiny, tmp_path):
    import eval_dense
    from scaling_retriever_amd.indexer import DenseFlatIndexer
    from scaling_retriever_amd.modeling.llm_encoder import LlamaBiDense
    cfg, w = tiny
    H = cfg["hidden_size"]
    rng = np.random.default_rng(8)
    model = LlamaBiDense.from_weights(cfg, w).to("cuda").eval()
    for id_kind in ("str", "int"):
        n_docs = 3000
        embs = rng.standard_normal((n_docs, H)).astype(np.float32)
        doc_ids = [str(7 * i + 1) for i in range(n_docs)] if id_kind == "str" else [7 * i + 1 for i in range(n_docs)]
        index = DenseFlatIndexer()
        index.init_index(H)
        index.index_data(embs[:2000], doc_ids[:2000])
        index.index_data(embs[2000:], doc_ids[2000:])
        retriever = eval_dense.LocalFaissDenseRetriever(model, index=index, device="cuda")
        loader = _batches(np.random.default_rng(9), cfg["vocab_size"], [16] * 9 + [5], 2, 12)
        for i, b in enumerate(loader):
            b["ids"] = [str(1000 * i + j) for j in range(len(b["ids"]))]
        qids, top_ids, top_scores = retriever.get_top_docs(loader, top_docs=50)
        assert isinstance(top_ids, list) and isinstance(top_ids[0], list) and len(top_ids) == len(qids) == 149
        assert type(top_ids[0][0]) is type(doc_ids[0]) and top_scores.dtype == np.float32
        # the reference's loop over get_top_docs' output (eval_dense.py:225-241), then json.dump
        run = {}
        for qid, ids_, scores_ in zip(qids, top_ids, top_scores):
            for docid, score in zip(ids_, scores_):
                run.setdefault(str(qid), {})[str(docid)] = float(score)
        path = tmp_path / f"run_{id_kind}.json"
        n_q, n_bytes = retriever.write_run(loader, 50, str(path))
        assert n_q == 149 and path.read_text() == json.dumps(run) and n_bytes == len(json.dumps(run))
    # fewer than k vectors: faiss pads with label -1, the mapping gives None there and the run skips it
    small = DenseFlatIndexer()
    small.init_index(H)
    small.index_data(embs[:7], doc_ids[:7])
    ids7, sc7 = small.search_knn(embs[:3], 10)
    assert all(row[7:] == [None] * 3 and None not in row[:7] for row in ids7)


def test_search_knn_in_pipelined_pieces_equals_one_search():
    """>= 1 024 queries: search_knn searches in KNN_CHUNKS pieces (GPU on piece c + 1 while the host maps piece c); the lists and
    scores are those of one search over the whole set."""
    from scaling_retriever_amd.indexer import DenseFlatIndexer
    rng = np.random.default_rng(12)
    H, n_docs, nq, k = 64, 30_000, 1101, 40
    embs = rng.standard_normal((n_docs, H)).astype(np.float32)
    doc_ids = [f"D{3 * i}" for i in range(n_docs)]
    index = DenseFlatIndexer()
    index.init_index(H)
    index.index_data(embs, doc_ids)
    q = rng.standard_normal((nq, H)).astype(np.float32)
    ids, scores = index.search_knn(q, k)
    whole_s, whole_i = index.search_arrays(q, k)
    assert DenseFlatIndexer.KNN_CHUNKS > 1 and np.array_equal(scores, whole_s) and scores.dtype == np.float32
    assert ids == [[doc_ids[j] for j in row] for row in whole_i]
    assert ids[0][0] is doc_ids[int(whole_i[0, 0])]          # the index's own objects, as the reference's id list gives


def test_search_begin_lower_bound_is_an_exact_score_j_documents_reach():
    from scaling_retriever_amd.scoring import DenseIndexHIP
    dev = torch.device("cuda", 0)
    N, H, nq, k, W = 150_000, 256, 160, 300, 2
    g = torch.Generator(device=dev).manual_seed(21)
    D = torch.randn((N, H), device=dev, generator=g) * (0.5 / H ** 0.5)
    Q = torch.randn((nq, H), device=dev, generator=g) * (0.5 / H ** 0.5)
    exact = DenseIndexHIP(H, device=dev)
    exact.add_device_rows(D)
    es, _ = exact.search(Q, k)
    idx = DenseIndexHIP(H, device=dev)
    idx.set_precision("fp32_filtered")
    idx.add_device_rows(D)
    j = (k + W - 1) // W
    lower = idx.search_begin(Q, k, W)
    assert bool(torch.isfinite(lower).all())
    # valid: at least j documents score >= lower exactly, i.e. lower <= the j-th best exact score ...
    assert bool((lower <= es[:, j - 1]).all())
    # ... and it IS one of the exact scores (the smallest among the j candidates with the largest upper bounds): tight to a
    # handful of ranks, not loosened by twice the error bound as min(U - 2e) was
    assert bool(((lower[:, None] == es).any(1)).all())
    assert bool((lower >= es[:, j + 7]).all())
    s, i = idx.search_finish(Q, k, lower)          # one shard standing in for all: every document >= lower comes back exactly
    keep = es >= lower[:, None]
    assert bool(((s == es) | ~keep).all()) and bool(((i >= 0) | ~keep).all())


def test_write_run_in_pipelined_pieces_equals_one_file(tiny, tmp_path):
    """>= 1 024 queries: write_run searches in RUN_PIECES pieces (GPU on piece c + 1 while the host formats and writes piece c through
    sr_write_run_json_part); the file holds the bytes of the one-piece run, which are json.dump's of the reference's loop."""
    import eval_dense
    from scaling_retriever_amd.indexer import DenseFlatIndexer
    from scaling_retriever_amd.modeling.llm_encoder import LlamaBiDense
    cfg, w = tiny
    H = cfg["hidden_size"]
    rng = np.random.default_rng(18)
    model = LlamaBiDense.from_weights(cfg, w).to("cuda").eval()
    n_docs = 20_000
    index = DenseFlatIndexer()
    index.init_index(H)
    index.index_data(rng.standard_normal((n_docs, H)).astype(np.float32), [f"P{5 * i}" for i in range(n_docs)])
    retriever = eval_dense.LocalFaissDenseRetriever(model, index=index, device="cuda")
    loader = _batches(np.random.default_rng(19), cfg["vocab_size"], [64] * 17 + [13], 2, 12)          # 1 101 queries
    for i, b in enumerate(loader):
        b["ids"] = [str(1000 * i + j) for j in range(len(b["ids"]))]
    p3, p1 = tmp_path / "run3.json", tmp_path / "run1.json"
    assert retriever.RUN_PIECES == 4
    n_q, n3 = retriever.write_run(loader, 30, str(p3))
    retriever.RUN_PIECES = 1
    _, n1 = retriever.write_run(loader, 30, str(p1))
    assert n_q == 1101 and n3 == n1 == p1.stat().st_size and p3.read_bytes() == p1.read_bytes()
    run = json.loads(p3.read_text())
    assert len(run) == 1101 and all(len(v) == 30 for v in run.values())
