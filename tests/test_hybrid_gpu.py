"""Hybrid model + HybridIndexer / HybridRetriever (/root/reference/scaling_retriever/indexer.py:710-1019): both heads
from ONE backbone pass (sr_encode_both) equal the two single-head encoders; one indexing pass writes the inverted index
and the dense shard files; retrieval writes sparse/run.json and dense/run.json equal to the single-head pipelines'."""
import json
import os

import numpy as np
import pytest
import torch
from torch.utils.data import DataLoader

from golden_weights import make_weights

pytestmark = pytest.mark.gpu


def _case(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    cfg = json.loads(str(z["config_json"]))
    return z, cfg, make_weights(cfg, int(z["weight_seed"]))


@pytest.mark.parametrize("name", ["enc_hd64", "enc_hd128", "enc_tiny_a"])
@pytest.mark.parametrize("prec", ["bf16", "fp32"])
def test_both_heads_from_one_pass_equal_the_single_head_encoders(golden_dir, name, prec):
    from scaling_retriever_amd.modeling.llm_encoder import LlamaBiDense, LlamaBiHybrid, LlamaBiSparse
    z, cfg, w = _case(golden_dir, name)
    hyb = LlamaBiHybrid.from_weights(cfg, w, precision=prec).to("cuda").eval()
    dense = LlamaBiDense.from_weights(cfg, {k: v for k, v in w.items() if not k.startswith("lm_head")}, precision=prec).to("cuda").eval()
    sparse = LlamaBiSparse.from_weights(cfg, w, precision=prec).to("cuda").eval()
    assert hyb.hidden_size == cfg["hidden_size"] and hyb.vocab_size == cfg["vocab_size"]
    for side in ("left", "right"):
        ids = torch.from_numpy(z[f"{side}:input_ids"]).cuda()
        mask = torch.from_numpy(z[f"{side}:attention_mask"]).cuda()
        s, d = hyb.encode(input_ids=ids, attention_mask=mask)
        assert torch.equal(d, dense.encode(input_ids=ids, attention_mask=mask)), side
        ref_s = sparse.encode(input_ids=ids, attention_mask=mask)
        # right padding: the dense span also computes the trailing pad rows (masked keys, skipped by the max) - same values
        assert torch.equal(s, ref_s) if side == "left" else bool((s - ref_s).abs().max() <= 1e-6 * ref_s.abs().max()), side
        # ... and BOTH heads of the one-pass encoder against the reference's own head classes (tests/golden/enc_*.npz: fp32
        # outputs of LlamaBiDense / LlamaBiSparse .encode, /root/reference/scaling_retriever/modeling/llm_encoder.py:186-196,424-443,
        # the heads HybridIndexer / HybridRetriever call, indexer.py:764,939): 2e-5 in the fp32 regime, the bf16-autocast band otherwise
        tol = 2e-5 if prec == "fp32" else 1.5e-2
        for got, key in ((d, "dense"), (s, "sparse")):
            ref = z[f"{side}:{key}"]
            err = float(np.linalg.norm(got.cpu().numpy() - ref) / np.linalg.norm(ref))
            assert err < tol, (name, prec, side, key, err)


def test_hybrid_indexer_and_retriever_end_to_end(golden_dir, tmp_path):
    from fake_tokenizer import FakeTokenizer
    from scaling_retriever_amd.dataset.data_collator import LlamaSparseCollectionCollator
    from scaling_retriever_amd.indexer import (DenseFlatIndexer, HybridIndexer, HybridRetriever, SparseIndexer, SparseRetrieval,
                                               store_embs)
    from scaling_retriever_amd.modeling.llm_encoder import LlamaBiDense, LlamaBiHybrid, LlamaBiSparse
    from scaling_retriever_amd.utils.utils import obtain_doc_vec_dir_files
    from test_pipeline import ListDataset, _corpus
    z, cfg, w = _case(golden_dir, "enc_tiny_a")
    tok = FakeTokenizer(vocab_size=cfg["vocab_size"], padding_side="left")
    docs, queries = ListDataset(_corpus(90, seed=1, max_words=20)), ListDataset([(f"q{i}", t) for i, (_, t) in enumerate(_corpus(8, seed=2, max_words=6))])
    collate_d, collate_q = LlamaSparseCollectionCollator(tok, 24), LlamaSparseCollectionCollator(tok, 8)
    loader = lambda: DataLoader(docs, batch_size=16, shuffle=False, collate_fn=collate_d)      # noqa: E731
    q_loader = lambda: DataLoader(queries, batch_size=4, shuffle=False, collate_fn=collate_q)  # noqa: E731
    hyb = LlamaBiHybrid.from_weights(cfg, w).to("cuda").eval()
    sp_dir, de_dir, out = str(tmp_path / "sp"), str(tmp_path / "de"), str(tmp_path / "out")
    HybridIndexer(hyb, sp_dir, de_dir, device="cuda", chunk_size=40, compute_stats=True, dim_voc=hyb.vocab_size).index(loader())
    assert json.load(open(os.path.join(de_dir, "plan.json")))["num_chunks"] == 2          # 48 + 42 passages
    for f in ("doc_ids.pkl", "index_stats.json", "index_dist.json"):
        assert os.path.exists(os.path.join(sp_dir, f))
    sparse_res, dense_res = HybridRetriever(hyb, sp_dir, de_dir, out, dim_voc=hyb.vocab_size, device="cuda").retrieve(q_loader(), topk=10)
    assert json.load(open(os.path.join(out, "sparse", "run.json"))) == {q: dict(r) for q, r in sparse_res.items()}
    assert json.load(open(os.path.join(out, "dense", "run.json"))) == {q: dict(r) for q, r in dense_res.items()}
    assert "L0_q" in json.load(open(os.path.join(out, "sparse", "q_stats.json")))

    # the two single-head pipelines over the same batches give the same runs
    sparse = LlamaBiSparse.from_weights(cfg, w).to("cuda").eval()
    sp2 = str(tmp_path / "sp2")
    SparseIndexer(sparse, sp2, device="cuda", compute_stats=True, dim_voc=sparse.vocab_size).index(loader())
    ref_sparse = SparseRetrieval(sparse, {"index_dir": sp2, "out_dir": str(tmp_path / "o2")}, sparse.vocab_size, "cuda").retrieve(q_loader(), topk=10)
    assert {q: dict(r) for q, r in ref_sparse.items()} == {q: dict(r) for q, r in sparse_res.items()}
    dense = LlamaBiDense.from_weights(cfg, {k: v for k, v in w.items() if not k.startswith("lm_head")}).to("cuda").eval()
    de2 = str(tmp_path / "de2")
    store_embs(dense, loader(), 0, de2, "cuda", chunk_size=40)
    index = DenseFlatIndexer()
    index.init_index(dense.hidden_size)
    for vf, idf in zip(*obtain_doc_vec_dir_files(de2)):
        index.index_data(np.load(vf), np.load(idf).tolist())
    q_reps, qids = [], []
    for b in q_loader():
        with torch.autocast("cuda", dtype=torch.bfloat16):          # HybridRetriever encodes queries under autocast (indexer.py:938)
            q_reps.append(dense.query_encode(input_ids=b["input_ids"].cuda(), attention_mask=b["attention_mask"].cuda()))
        qids += b["ids"]
    top_ids, top_scores = index.search_knn(torch.cat(q_reps), 10)
    for qid, dids, scs in zip(qids, top_ids, top_scores):
        assert list(dense_res[qid].keys()) == [str(d) for d in dids]
        assert np.array_equal(np.array(list(dense_res[qid].values()), np.float32), scs)
