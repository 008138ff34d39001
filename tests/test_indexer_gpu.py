"""GPU: the reference-shaped retrieval layer (scaling_retriever_amd/indexer.py) end to end on a tiny
model - store_embs artefacts, DenseFlatIndexer, SparseIndexer -> SparseRetrieval - against the oracle.
Callers mirrored: /root/reference/eval_dense.py:158-241, /root/reference/eval_sparse.py:75-151."""
import json
import os
import pickle

import numpy as np
import pytest
import torch

from golden_weights import make_weights
from oracle import llama_bi as LB
from oracle import scoring as SC

pytestmark = pytest.mark.gpu


class FakeLoader:
    """Stands in for DataLoader(collate_fn=LlamaDenseCollectionCollator): yields
    {"input_ids", "attention_mask", "ids"} with left padding to the longest row."""

    def __init__(self, seqs, ids, batch_size, pad_id):
        self.seqs, self.ids, self.batch_size, self.pad_id = seqs, ids, batch_size, pad_id

    def __len__(self):
        return (len(self.seqs) + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        for b0 in range(0, len(self.seqs), self.batch_size):
            chunk = self.seqs[b0:b0 + self.batch_size]
            L = max(len(s) for s in chunk)
            ids = np.full((len(chunk), L), self.pad_id, np.int64)
            mask = np.zeros((len(chunk), L), np.int64)
            for r, s in enumerate(chunk):
                ids[r, L - len(s):] = s
                mask[r, L - len(s):] = 1
            yield {"input_ids": torch.from_numpy(ids), "attention_mask": torch.from_numpy(mask),
                   "ids": list(self.ids[b0:b0 + self.batch_size])}


def _corpus(rng, n, V, lo, hi):
    return [rng.integers(0, V - 1, size=int(rng.integers(lo, hi + 1))) for _ in range(n)]


def _oracle_encode(fn, w, cfg, seqs, pad_id):
    out = []
    for s in seqs:
        out.append(fn(w, cfg, np.asarray(s)[None, :], np.ones((1, len(s)), np.int64))[0])
    return np.stack(out)


@pytest.fixture(scope="module")
def tiny(golden_dir):
    z = np.load(os.path.join(golden_dir, "enc_tiny_a.npz"))
    cfg = json.loads(str(z["config_json"]))
    return cfg, make_weights(cfg, int(z["weight_seed"]))


def test_store_embs_then_dense_retrieval(tiny, tmp_path):
    from scaling_retriever_amd.indexer import DenseFlatIndexer, store_embs
    from scaling_retriever_amd.modeling.llm_encoder import LlamaBiDense
    from scaling_retriever_amd.utils.utils import obtain_doc_vec_dir_files
    cfg, w = tiny
    V = cfg["vocab_size"]
    rng = np.random.default_rng(0)
    docs, queries = _corpus(rng, 70, V, 3, 30), _corpus(rng, 9, V, 2, 8)
    pids = [f"p{i}" for i in range(len(docs))]
    model = LlamaBiDense.from_weights(cfg, w, precision="bf16").to("cuda").eval()
    loader = FakeLoader(docs, pids, batch_size=16, pad_id=V - 1)
    out_dir = str(tmp_path / "embs")
    store_embs(model, loader, local_rank=0, index_dir=out_dir, device="cuda", chunk_size=32)   # 2 batches per chunk
    plan = json.load(open(os.path.join(out_dir, "plan.json")))
    assert plan["nranks"] == 1 and plan["num_chunks"] == 3
    vec_files, id_files = obtain_doc_vec_dir_files(out_dir)
    embs = np.concatenate([np.load(f) for f in vec_files])
    ids = np.concatenate([np.load(f) for f in id_files]).tolist()
    assert embs.dtype == np.float32 and embs.shape == (70, cfg["hidden_size"]) and ids == pids
    ref_d = _oracle_encode(LB.dense_encode, w, cfg, docs, V - 1)
    assert np.linalg.norm(embs - ref_d) / np.linalg.norm(ref_d) < 1.5e-2
    # store_embs encodes the loader's fixed-size batches a few at a time in one engine pass: the rows are those of the reference's
    # batch-by-batch doc_encode calls under autocast (indexer.py:46-52), bit for bit
    with torch.inference_mode(), torch.autocast("cuda", dtype=torch.bfloat16):
        per_batch = torch.cat([model.doc_encode(input_ids=b["input_ids"].cuda(), attention_mask=b["attention_mask"].cuda()) for b in loader])
    assert np.array_equal(embs, per_batch.float().cpu().numpy())
    # retrieval task (eval_dense.py:190-241)
    index = DenseFlatIndexer()
    index.init_index(cfg["hidden_size"])
    index.index_data(embs, ids)
    qloader = FakeLoader(queries, [f"q{i}" for i in range(len(queries))], batch_size=4, pad_id=V - 1)
    q_reps = np.concatenate([model.query_encode(input_ids=b["input_ids"].cuda(), attention_mask=b["attention_mask"].cuda()).cpu().numpy()
                             for b in qloader])
    top_ids, top_scores = index.search_knn(q_reps, 10)
    es, ei = SC.topk_rows(SC.dense_scores_fma(q_reps, embs, SC.mfma_korder(cfg["hidden_size"])), 10)
    assert top_scores.dtype == np.float32 and np.array_equal(top_scores, es)
    assert top_ids == [[pids[j] for j in row] for row in ei]
    assert (np.diff(top_scores, axis=1) <= 0).all()
    # serialize / deserialize round trip
    index.serialize(str(tmp_path / "ix"))
    index2 = DenseFlatIndexer()
    index2.deserialize(str(tmp_path / "ix"))
    ids2, sc2 = index2.search_knn(q_reps, 10)
    assert ids2 == top_ids and np.array_equal(sc2, top_scores)


def test_sparse_index_then_retrieval(tiny, tmp_path):
    from scaling_retriever_amd.indexer import SparseIndexer, SparseRetrieval
    from scaling_retriever_amd.modeling.llm_encoder import LlamaBiSparse
    cfg, w = tiny
    V = cfg["vocab_size"]
    rng = np.random.default_rng(1)
    docs, queries = _corpus(rng, 50, V, 1, 6), _corpus(rng, 7, V, 1, 3)
    pids, qids = [f"p{i}" for i in range(len(docs))], [f"q{i}" for i in range(len(queries))]
    model = LlamaBiSparse.from_weights(cfg, w, precision="bf16").to("cuda").eval()
    index_dir = str(tmp_path / "index")
    indexer = SparseIndexer(model, index_dir=index_dir, compute_stats=True, dim_voc=model.vocab_size, device="cuda")
    indexer.index(FakeLoader(docs, pids, batch_size=8, pad_id=V - 1))
    doc_ids = pickle.load(open(os.path.join(index_dir, "doc_ids.pkl"), "rb"))
    assert doc_ids == {i: p for i, p in enumerate(pids)}
    stats = json.load(open(os.path.join(index_dir, "index_stats.json")))
    # postings equal the nonzeros of the HIP reps, in doc order inside every term
    reps = np.concatenate([model.encode(input_ids=b["input_ids"].cuda(), attention_mask=b["attention_mask"].cuda()).cpu().numpy()
                           for b in FakeLoader(docs, pids, 8, V - 1)])
    assert stats["L0_d"] == pytest.approx(np.mean([(reps[i:i + 8] != 0).sum(1).mean() for i in range(0, 50, 8)]), rel=1e-5)
    ref = _oracle_encode(LB.sparse_encode, w, cfg, docs, V - 1)
    assert np.linalg.norm(reps - ref) / np.linalg.norm(ref) < 1.5e-2
    retr = SparseRetrieval(config={"index_dir": index_dir, "out_dir": str(tmp_path / "out")}, model=model,
                           compute_stats=True, dim_voc=model.vocab_size, device="cuda")
    assert retr.sparse_index.nb_docs() == 50
    t = 5
    assert np.array_equal(retr.sparse_index.index_doc_id[t], np.nonzero(reps[:, t])[0])
    np.testing.assert_array_equal(retr.sparse_index.index_doc_value[t], reps[reps[:, t] != 0, t])
    res = retr.retrieve(FakeLoader(queries, qids, batch_size=4, pad_id=V - 1), topk=10, threshold=0.0)
    run = json.load(open(tmp_path / "out" / "run.json"))
    # `res` is the reference's nested dict as a read-only mapping over the result arrays; the file holds json.dump of that dict
    assert run == res.to_dict() == dict(res) and set(run) == set(qids)
    assert open(tmp_path / "out" / "run.json").read() == json.dumps(res.to_dict())
    assert res[qids[0]] == run[qids[0]] and list(res[qids[0]]) == list(run[qids[0]])
    assert "L0_q" in json.load(open(tmp_path / "out" / "q_stats.json"))
    # oracle: numba_score_float + select_topk on the SAME index and the SAME query vectors
    indptr, ids, vals = retr.sparse_index.csr(V)
    qvecs, _ = retr._generate_query_vecs(FakeLoader(queries, qids, batch_size=4, pad_id=V - 1))
    for qi, (cols, qv) in enumerate(qvecs):
        assert np.all(np.diff(cols) > 0) and cols.dtype == np.int32 and qv.dtype == np.float32
        fi, neg = SC.numba_score_float(indptr, ids, vals, cols, qv, 0.0, 50)
        ei, es = SC.select_topk(fi, neg, 10)
        got = run[qids[qi]]
        assert list(got.keys()) == [pids[j] for j in ei]
        np.testing.assert_array_equal(np.array(list(got.values()), np.float32), es)
    # the reference's static helper signature still works (indexer.py:324-344)
    cols, qv = qvecs[0]
    fi2, neg2 = SparseRetrieval.numba_score_float(
        {t: retr.sparse_index.index_doc_id[t] for t in range(V)}, {t: retr.sparse_index.index_doc_value[t] for t in range(V)},
        cols, qv, threshold=0.0, size_collection=50)
    fi, neg = SC.numba_score_float(indptr, ids, vals, cols, qv, 0.0, 50)
    assert fi2.dtype == np.int64 and np.array_equal(fi2, fi) and np.array_equal(neg2, neg)
    ti, ts = SparseRetrieval.select_topk(fi2, neg2, 3)
    ei, es = SC.select_topk(fi, neg, 3)
    assert np.array_equal(ti, ei) and np.array_equal(ts, es)


def test_sharded_dense_retriever_single_process_fake_world():
    """Fake-world check of the multi-GPU layout: W shards built with id_base=rank, id_stride=W and merged
    with sr_topk_merge equal the single-index search bit for bit."""
    from scaling_retriever_amd.scoring import DenseIndexHIP, topk_merge
    W, n, h, k = 4, 5003, 64, 100
    g = torch.Generator(device="cuda").manual_seed(0)
    D = torch.randn((n, h), device="cuda", generator=g)
    Q = torch.randn((33, h), device="cuda", generator=g)
    full = DenseIndexHIP(h)
    full.add_device_rows(D)
    fs, fi = full.search(Q, k)
    ss, si = [], []
    for r in range(W):
        shard = DenseIndexHIP(h)
        shard.add_device_rows(D[r::W].contiguous(), id_base=r, id_stride=W)
        s, i = shard.search(Q, k)
        ss.append(s)
        si.append(i)
    ms, mi = topk_merge(torch.stack(ss), torch.stack(si))
    assert torch.equal(mi, fi) and torch.equal(ms, fs)


def test_sparse_retrieve_in_pipelined_groups_equals_one_group(tiny, tmp_path, monkeypatch):
    """SparseRetrieval.retrieve takes the query groups one after the other (encode -> search) while a worker thread writes the previous
    group's piece of run.json: same mapping, same file bytes and same q_stats.json as the one-group path."""
    from scaling_retriever_amd.indexer import SparseIndexer, SparseRetrieval
    from scaling_retriever_amd.modeling.llm_encoder import LlamaBiSparse
    cfg, w = tiny
    V = cfg["vocab_size"]
    rng = np.random.default_rng(5)
    docs, queries = _corpus(rng, 300, V, 1, 6), _corpus(rng, 37, V, 1, 3)
    pids, qids = [f"p{i}" for i in range(len(docs))], [f"q{7 * i}" for i in range(len(queries))]
    model = LlamaBiSparse.from_weights(cfg, w, precision="bf16").to("cuda").eval()
    index_dir = str(tmp_path / "index")
    SparseIndexer(model, index_dir=index_dir, compute_stats=True, dim_voc=model.vocab_size, device="cuda").index(
        FakeLoader(docs, pids, batch_size=16, pad_id=V - 1))
    outs = []
    for name, rows in (("one", 2048), ("many", 8)):           # 37 queries in loader batches of 4: one group / five groups of 8 rows
        monkeypatch.setattr(SparseRetrieval, "QUERY_GROUP_ROWS", rows)
        retr = SparseRetrieval(config={"index_dir": index_dir, "out_dir": str(tmp_path / name)}, model=model, compute_stats=True,
                               dim_voc=model.vocab_size, device="cuda")
        res = retr.retrieve(FakeLoader(queries, qids, batch_size=4, pad_id=V - 1), topk=20, threshold=0.0)
        outs.append((res.to_dict(), (tmp_path / name / "run.json").read_bytes(), json.load(open(tmp_path / name / "q_stats.json"))))
    assert outs[0][0] == outs[1][0] and len(outs[0][0]) > 30
    assert outs[0][1] == outs[1][1]
    assert outs[0][2]["L0_q"] == pytest.approx(outs[1][2]["L0_q"], rel=1e-6)
