"""GPU: north_star's acceptance bar restated on a synthetic collection - the whole HIP pipeline (doc_encode ->
index -> query_encode -> top-10) must reproduce the ranking METRICS of the reference pipeline (oracle fp32 encoder +
oracle scoring) within 1e-3 MRR@10 / nDCG@10, dense and sparse.

Collection: 3000 random passages and 4000 queries, each query a noisy sub-sample of one passage's tokens, which is
its single relevant document (qrels do not depend on either system).  Random-init weights have no trained margin,
so near-ties are far more frequent here than on MS MARCO; the oracle's own bf16-autocast emulation
(oracle.llama_bi.Hooks(bf16=True), what the reference runs under torch.autocast) is evaluated alongside to show the
spread two faithful mixed-precision runs have on this data."""
import json
import os

import numpy as np
import pytest
import torch

from golden_weights import make_weights
from oracle import llama_bi as LB
from oracle import scoring as SC

pytestmark = pytest.mark.gpu

N_DOCS, N_QUERIES, K = 3000, 4000, 10
TOL = 1e-3


def _collection(vocab, seed):
    rng = np.random.default_rng(seed)
    docs = [rng.integers(3, vocab, size=rng.integers(8, 25)) for _ in range(N_DOCS)]
    src = rng.integers(0, N_DOCS, size=N_QUERIES)
    queries = []
    for d in src:
        keep = rng.choice(docs[d], size=min(6, len(docs[d])), replace=False)
        noise = rng.integers(3, vocab, size=2)
        q = np.concatenate([keep, noise])
        rng.shuffle(q)
        queries.append(q)
    return docs, queries, src


def _pad(seqs, side):
    S = max(len(s) for s in seqs)
    ids = np.zeros((len(seqs), S), np.int64)
    mask = np.zeros((len(seqs), S), np.int64)
    for i, s in enumerate(seqs):
        if side == "left":
            ids[i, S - len(s):], mask[i, S - len(s):] = s, 1
        else:
            ids[i, :len(s)], mask[i, :len(s)] = s, 1
    return ids, mask


def _oracle_encode(fn, w, cfg, seqs, side, hooks=None, batch=250):
    out = []
    for b in range(0, len(seqs), batch):
        ids, mask = _pad(seqs[b:b + batch], side)
        out.append(fn(w, cfg, ids, mask, hooks))
    return np.concatenate(out)


def _hip_encode(model, seqs, side, query, batch=128, autocast=True):
    """autocast=True: the reference's document / sparse-query regime (indexer.py:46-52, :255-256, :390-391); False: its
    dense-query regime (eval_dense.py:94-106)."""
    import contextlib
    out = []
    for b in range(0, len(seqs), batch):
        ids, mask = _pad(seqs[b:b + batch], side)
        f = model.query_encode if query else model.doc_encode
        with (torch.autocast("cuda", dtype=torch.bfloat16) if autocast else contextlib.nullcontext()):
            out.append(f(input_ids=torch.from_numpy(ids).cuda(), attention_mask=torch.from_numpy(mask).cuda()))
    return torch.cat(out)


def _metrics(ids, scores, src):
    from scaling_retriever_amd.utils.metrics import mrr_k, ndcg_k
    run = {f"q{q}": {f"d{int(d)}": float(s) for d, s in zip(ids[q], scores[q])} for q in range(len(ids))}
    qrel = {f"q{q}": {f"d{int(src[q])}": 1} for q in range(len(ids))}
    return mrr_k(run, qrel, K), ndcg_k(run, qrel, K)


@pytest.fixture(scope="module")
def setup(golden_dir):
    z = np.load(os.path.join(golden_dir, "enc_hd64.npz"))
    cfg = json.loads(str(z["config_json"]))
    w = make_weights(cfg, 4242)
    return cfg, w, _collection(cfg["vocab_size"], 17)


def test_dense_pipeline_reproduces_mrr_and_ndcg(setup):
    from scaling_retriever_amd.modeling.llm_encoder import LlamaBiDense
    from scaling_retriever_amd.scoring import DenseIndexHIP
    cfg, w, (docs, queries, src) = setup
    ref = {}
    for name, hooks in (("fp32", None), ("bf16", LB.Hooks(bf16=True))):
        d = _oracle_encode(LB.dense_encode, w, cfg, docs, "left", hooks)
        q = _oracle_encode(LB.dense_encode, w, cfg, queries, "left", hooks)
        s, i = SC.flat_ip_search(q, d, K)
        ref[name] = _metrics(i, s, src)
    # the reference's precision map: documents under bf16 autocast, queries in fp32 (the model follows the autocast context)
    model = LlamaBiDense.from_weights(cfg, w).to("cuda").eval()
    index = DenseIndexHIP(cfg["hidden_size"])
    index.add_device_rows(_hip_encode(model, docs, "left", False, autocast=True))
    s, i = index.search(_hip_encode(model, queries, "left", True, autocast=False), K)
    got = _metrics(i.cpu().numpy(), s.cpu().numpy(), src)
    print("dense MRR@10/nDCG@10: hip", got, "oracle fp32", ref["fp32"], "oracle bf16-autocast", ref["bf16"])
    assert 0.2 < ref["fp32"][0] < 0.999                      # the task is neither trivial nor hopeless
    assert abs(got[0] - ref["fp32"][0]) <= TOL and abs(got[1] - ref["fp32"][1]) <= TOL


def test_sparse_pipeline_reproduces_mrr_and_ndcg(setup):
    from scaling_retriever_amd.modeling.llm_encoder import LlamaBiSparse
    cfg, _, (docs, queries, src) = setup
    # a random-init expansion head carries no lexical signal; tie it to small embeddings and damp the layers so that a
    # passage's own tokens get the largest term weights and there is a ranking to reproduce
    cfg = dict(cfg, tie_word_embeddings=True)
    w = make_weights(cfg, 4243, embed_std=0.05)
    for name in w:
        if "proj" in name:
            w[name] = (w[name] * np.float32(0.2)).astype(np.float32)
    ref = {}
    for name, hooks in (("fp32", None), ("bf16", LB.Hooks(bf16=True))):
        d = _oracle_encode(LB.sparse_encode, w, cfg, docs, "right", hooks)
        q = _oracle_encode(LB.sparse_encode, w, cfg, queries, "right", hooks)
        sc = q @ d.T
        i = np.argsort(-sc, axis=1, kind="stable")[:, :K]
        ref[name] = _metrics(i, np.take_along_axis(sc, i, 1), src)
    model = LlamaBiSparse.from_weights(cfg, w).to("cuda").eval()          # both sides under autocast, as eval_sparse.py does
    d_reps = _hip_encode(model, docs, "right", False)
    q_reps = _hip_encode(model, queries, "right", True)
    from scaling_retriever_amd.scoring import SparseIndexHIP
    from scaling_retriever_amd.indexer import sparse_reps_to_csr
    d_ptr, cols, vals = sparse_reps_to_csr(d_reps)
    rows = torch.repeat_interleave(torch.arange(N_DOCS, device="cuda"), d_ptr[1:] - d_ptr[:-1])
    # document-major postings -> term-major CSR (what SparseIndexer builds on the device)
    order = torch.argsort(cols.long() * N_DOCS + rows)
    t_sorted, d_sorted, v_sorted = cols[order].long(), rows[order].int(), vals[order]
    indptr = torch.zeros(cfg["vocab_size"] + 1, dtype=torch.int64, device="cuda")
    indptr[1:] = torch.cumsum(torch.bincount(t_sorted, minlength=cfg["vocab_size"]), 0)
    index = SparseIndexHIP(indptr, d_sorted.contiguous(), v_sorted.contiguous(), N_DOCS)
    q_indptr, qc, qv = sparse_reps_to_csr(q_reps)
    s, i, c = index.search(q_indptr, qc.int().contiguous(), qv.contiguous(), K)
    assert (c == K).all()
    got = _metrics(i.cpu().numpy(), s.cpu().numpy(), src)
    print("sparse MRR@10/nDCG@10: hip", got, "oracle fp32", ref["fp32"], "oracle bf16-autocast", ref["bf16"])
    assert 0.2 < ref["fp32"][0] < 0.999
    assert abs(got[0] - ref["fp32"][0]) <= TOL and abs(got[1] - ref["fp32"][1]) <= TOL
