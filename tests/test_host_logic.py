"""CPU: host-side logic that needs no GPU - inverted-index container, plan files, metrics,
shard assignment, and the N>1 gather plumbing over gloo (world_size 2)."""
import json
import os
import pickle
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_index_dict_of_array_matches_reference_semantics(tmp_path):
    """Posting order inside a term = insertion order; nb_docs from n_docs; save/load round trip
    (inverted_index.py:67-105)."""
    from scaling_retriever_amd.utils.inverted_index import IndexDictOfArray
    idx = IndexDictOfArray(str(tmp_path / "ix"), force_new=True, dim_voc=10)
    idx.add_batch_document(np.array([0, 0, 2]), np.array([3, 7, 3]), np.array([1.0, 2.0, 3.0]), n_docs=3)
    idx.add_batch_document(np.array([5, 4]), np.array([3, 1]), np.array([4.0, 5.0]), n_docs=2)
    assert idx.nb_docs() == 5
    assert list(idx.index_doc_id[3]) == [0, 2, 5] and list(idx.index_doc_value[3]) == [1.0, 3.0, 4.0]
    assert list(idx.index_doc_id[1]) == [4] and len(idx.index_doc_id[9]) == 0
    assert len(idx) == 3 and 3 in idx.index_doc_id and 2 not in idx.index_doc_id
    idx.save()
    pickle.dump({0: "a", 2: "b", 4: "c", 5: "d"}, open(tmp_path / "ix" / "doc_ids.pkl", "wb"))
    dist_json = json.load(open(tmp_path / "ix" / "index_dist.json"))
    assert dist_json == {"1": 1, "3": 3, "7": 1}
    re = IndexDictOfArray(str(tmp_path / "ix"), dim_voc=10)
    assert re.nb_docs() == 6                      # max g_row + 1 for dict doc_ids (inverted_index.py:44-55)
    assert list(re.index_doc_id[3]) == [0, 2, 5]
    indptr, ids, vals = re.csr(12)
    assert len(indptr) == 13 and indptr[-1] == 5 and ids.dtype == np.int32 and vals.dtype == np.float32


def test_merge_indexes_concatenates_rank_dirs(tmp_path):
    from scaling_retriever_amd.utils.inverted_index import IndexDictOfArray, merge_indexes
    json.dump({"vocab_size": 6}, open(tmp_path / "config.json", "w"))
    for r in range(2):
        d = tmp_path / f"index_{r}"
        ix = IndexDictOfArray(str(d), force_new=True, dim_voc=6)
        rows = np.array([0, 1, 2]) * 2 + r          # g_row = row * W + rank
        ix.add_batch_document(rows, np.array([1, 1, 4]), np.array([1.0, 2.0, 3.0]) + r, n_docs=3)
        ix.save()
        pickle.dump({int(g): f"p{g}" for g in rows}, open(d / "doc_ids.pkl", "wb"))
        json.dump({"L0_d": 1.0 + r}, open(d / "index_stats.json", "w"))
    merge_indexes(str(tmp_path), index_name="index", index_dir=str(tmp_path))
    m = IndexDictOfArray(str(tmp_path / "index"), dim_voc=6)
    assert sorted(m.index_doc_id[1].tolist()) == [0, 1, 2, 3] and m.nb_docs() == 6
    assert json.load(open(tmp_path / "index" / "index_stats.json"))["L0_d"] == 1.5


def test_plan_files_order_matches_golden(golden_dir, tmp_path):
    from scaling_retriever_amd.utils.utils import obtain_doc_vec_dir_files
    z = np.load(os.path.join(golden_dir, "plan_files.npz"))
    nr, nc = int(z["nranks"]), int(z["num_chunks"])
    json.dump({"nranks": nr, "num_chunks": nc, "index_path": "x"}, open(tmp_path / "plan.json", "w"))
    for r in range(nr):
        for c in range(nc):
            np.save(tmp_path / f"embs_{r}_{c}.npy", np.zeros((1, 2), np.float32))
            np.save(tmp_path / f"ids_{r}_{c}.npy", np.zeros(1, np.int64))
    vec, ids = obtain_doc_vec_dir_files(str(tmp_path))
    assert [os.path.basename(p) for p in vec] == list(z["vec"])
    assert [os.path.basename(p) for p in ids] == list(z["ids"])


def test_metrics_hand_made_runs():
    """MRR@10 = reciprocal rank of the first relevant doc within the top 10 (metrics.py:13-29)."""
    from scaling_retriever_amd.utils.metrics import evaluate, mrr_k, ndcg_k, recall_k
    qrel = {"q1": {"d1": 1}, "q2": {"d9": 1, "d3": 2}, "q3": {"d5": 1}}
    run = {"q1": {"d7": 3.0, "d1": 2.0, "d2": 1.0},                       # relevant at rank 2
           "q2": {f"d{i}": 20.0 - i for i in range(12)},                  # d3 at rank 4, d9 at rank 10
           "q4": {"d1": 1.0}}                                             # not in qrel: ignored
    assert mrr_k(run, qrel, 10) == pytest.approx((1 / 2 + 1 / 4) / 2)
    assert mrr_k(run, qrel, 3) == pytest.approx((1 / 2 + 0) / 2)
    assert recall_k(run, qrel, 5) == pytest.approx((1.0 + 0.5) / 2)
    dcg = 2 / np.log2(5) + 1 / np.log2(11)
    idcg = 2 / np.log2(2) + 1 / np.log2(3)
    assert ndcg_k(run, qrel, 10) == pytest.approx((1 / np.log2(3) + dcg / idcg) / 2)
    assert evaluate(run, qrel, "recall", select=5) == pytest.approx(0.75)
    # ties: trec_eval orders equal scores by docid descending
    assert mrr_k({"q3": {"d5": 1.0, "d6": 1.0}}, qrel, 10) == pytest.approx(0.5)


def test_shard_rows_match_distributed_sampler_without_padding():
    """rank r takes rows r, r+W, ... exactly like DistributedSampler(shuffle=False) (eval_dense.py:178),
    minus its wrap-around padding duplicates."""
    from torch.utils.data.distributed import DistributedSampler
    from scaling_retriever_amd.distributed import shard_rows, shard_size
    for n, W in [(10, 4), (8, 4), (1, 2), (8841823, 8)]:
        seen = []
        for r in range(W):
            mine = list(shard_rows(n, r, W)) if n < 100 else None
            if mine is not None:
                ref = list(DistributedSampler(range(n), num_replicas=W, rank=r, shuffle=False))
                assert mine == ref[:len(mine)] and set(ref[len(mine):]) <= set(range(W))   # extras = wrapped duplicates
                seen += mine
            assert shard_size(n, r, W) == len(range(r, n, W))
        if n < 100:
            assert sorted(seen) == list(range(n))


def test_pack_unpack_topk_roundtrip():
    from scaling_retriever_amd.distributed import pack_topk, unpack_topk
    s = torch.tensor([[1.5, -2.25, 0.0, -3.4e38, float("inf")]])
    i = torch.tensor([[0, 4294967294, 17, -1, 8841822]])
    s2, i2 = unpack_topk(pack_topk(s, i))
    assert torch.equal(s, s2) and torch.equal(i, i2)


_WORKER = r'''
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from scaling_retriever_amd.distributed import gather_topk, shard_rows
from oracle import scoring as SC
dist.init_process_group("gloo")
rank, W = dist.get_rank(), dist.get_world_size()
from scaling_retriever_amd.distributed import all_gather_query_reps, query_slice
for nq in (7, 8, 1):                                     # ragged, even and fewer-queries-than-ranks splits
    Qall = torch.arange(nq * 3, dtype=torch.float32).reshape(nq, 3)
    lo, hi = query_slice(nq, rank, W)
    assert torch.equal(all_gather_query_reps(Qall[lo:hi].contiguous(), nq), Qall), nq
# the threshold exchange of the doc-sharded dense search: element-wise minimum over the ranks (distributed.all_reduce_min)
from scaling_retriever_amd.distributed import all_reduce_min
mine = torch.tensor([1.0 + rank, -2.0 - rank, float("-inf") if rank == 1 else 5.0, 0.25])
assert torch.equal(all_reduce_min(mine), torch.tensor([1.0, -2.0 - (W - 1), float("-inf"), 0.25]))
rng = np.random.default_rng(0)
D = rng.standard_normal((501, 32), dtype=np.float32); Q = rng.standard_normal((6, 32), dtype=np.float32); k = 20
rows = np.array(list(shard_rows(len(D), rank, W)))
s, li = SC.flat_ip_search(Q, D[rows], k)                 # test double for the per-shard HIP search
gi = np.where(li >= 0, li * W + rank, -1)                # g_row = local * W + rank
gs, gids = gather_topk(torch.from_numpy(s), torch.from_numpy(gi), dst=0)
# the plain gather of the padded [nq, k] buffer - what travels over RCCL (no size exchange, no host sync) - and the compacting one (this
# transport's default) hand rank 0 the same rows; padded slots (id -1) included
s_pad, gi_pad = s.copy(), gi.copy()
s_pad[:, k // 2:], gi_pad[:, k // 2:] = 0.0, -1
for padded in (False, True):
    a_s, a_i = (s_pad, gi_pad) if padded else (s, gi)
    p_s, p_i = gather_topk(torch.from_numpy(a_s), torch.from_numpy(a_i), dst=0, compact=False)
    c_s, c_i = gather_topk(torch.from_numpy(a_s), torch.from_numpy(a_i), dst=0, compact=True)
    if rank == 0:
        assert torch.equal(p_i, c_i) and torch.equal(p_s[c_i >= 0], c_s[c_i >= 0]) and p_i.shape == (W, 6, k), padded
    else:
        assert p_s is None and c_s is None
if rank == 0:
    assert gs.shape == (W, 6, k)
    cs = gs.permute(1, 0, 2).reshape(6, -1).numpy(); ci = gids.permute(1, 0, 2).reshape(6, -1).numpy()
    es, ei = SC.flat_ip_search(Q, D, k)
    for q in range(6):
        o = np.lexsort((ci[q], -cs[q].astype(np.float64)))[:k]
        assert np.array_equal(ci[q][o], ei[q]), q
    print("MERGE_OK")
else:
    assert gs is None

# ---- sparse twin: query CSR all-gather + doc-sharded inverted-index search (per-shard scorer = oracle test double)
from scaling_retriever_amd.distributed import all_gather_query_csr
V, N, nq, k = 40, 203, 9, 12
cnt = rng.integers(0, 6, size=nq); cnt[3] = 0                                  # ragged, one empty query
q_ptr = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64)
q_cols = np.concatenate([np.sort(rng.choice(V, size=c, replace=False)) for c in cnt]).astype(np.int32)
q_vals = rng.random(len(q_cols), dtype=np.float32) + 0.1
lo, hi = query_slice(nq, rank, W)
p, c, v = all_gather_query_csr(torch.from_numpy(q_ptr[lo:hi + 1] - q_ptr[lo]), torch.from_numpy(q_cols[q_ptr[lo]:q_ptr[hi]]),
                               torch.from_numpy(q_vals[q_ptr[lo]:q_ptr[hi]]), nq)
assert np.array_equal(p.numpy(), q_ptr) and np.array_equal(c.numpy(), q_cols) and np.array_equal(v.numpy(), q_vals)
dense = (rng.random((N, V)) < 0.2) * (rng.random((N, V), dtype=np.float32) + 0.05)
def csr_of(rows):                                                              # CSR by term over the given doc rows
    sub = dense[rows]
    ids, vals, ptr = [], [], [0]
    for t in range(V):
        nz = np.nonzero(sub[:, t])[0]
        ids.append(nz.astype(np.int32)); vals.append(sub[nz, t].astype(np.float32)); ptr.append(ptr[-1] + len(nz))
    return np.array(ptr, np.int64), np.concatenate(ids), np.concatenate(vals)
rows = np.arange(rank, N, W)                                                   # this rank's documents: g_row = local * W + rank
ptr_r, ids_r, vals_r = csr_of(rows)
li, ls, lc = SC.sparse_retrieve_c(ptr_r, ids_r, vals_r, q_ptr, q_cols, q_vals, k, 0.0, len(rows))
gi = np.where(np.arange(k)[None, :] < lc[:, None], li * W + rank, -1)
ls = np.where(gi >= 0, ls, 0.0).astype(np.float32)
gs, gids = gather_topk(torch.from_numpy(ls), torch.from_numpy(gi), dst=0)
if rank == 0:
    ptr_a, ids_a, vals_a = csr_of(np.arange(N))
    ei, es, ec = SC.sparse_retrieve_c(ptr_a, ids_a, vals_a, q_ptr, q_cols, q_vals, k, 0.0, N)
    cs = gs.permute(1, 0, 2).reshape(nq, -1).numpy(); ci = gids.permute(1, 0, 2).reshape(nq, -1).numpy()
    for q in range(nq):
        keep = ci[q] >= 0
        o = np.lexsort((ci[q][keep], -cs[q][keep].astype(np.float64)))[:k]
        assert np.array_equal(ci[q][keep][o], ei[q, :ec[q]]), q              # ids AND fp32 scores equal the single-index search
        assert np.array_equal(cs[q][keep][o], es[q, :ec[q]]), q
    print("SPARSE_MERGE_OK")
dist.barrier(); dist.destroy_process_group()
'''


def test_gather_topk_world_size_2_gloo(tmp_path):
    """N>1 path on CPU: 2 processes over gloo, per-shard top-k -> ONE gather -> merged == global top-k."""
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29611", str(script), ROOT],
                         capture_output=True, text=True, env=env, timeout=240)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "MERGE_OK" in out.stdout and "SPARSE_MERGE_OK" in out.stdout


def test_in_memory_index_carries_its_postings():
    """SparseIndexer.index(index_dir=None) hands the device CSR to the container; every consumer of
    index_d["index"] (csr(), per-term views, save, the world_size > 1 retrieval path) must see the postings, as with the
    reference's in-memory index_d (indexer.py:239-308)."""
    import torch
    from scaling_retriever_amd.utils.inverted_index import IndexDictOfArray
    idx = IndexDictOfArray(dim_voc=6)
    indptr = torch.tensor([0, 2, 2, 5, 5, 5, 6])
    ids = torch.tensor([0, 3, 1, 2, 3, 0], dtype=torch.int32)
    vals = torch.tensor([1.0, 2.0, 3.0, 4.0, 5.0, 6.0])
    idx.set_device_csr(indptr, ids, vals, 4)
    assert idx.nb_docs() == 4
    assert len(idx) == 3 and idx.index_doc_id[2].tolist() == [1, 2, 3] and idx.index_doc_value[5].tolist() == [6.0]
    ip, di, va = idx.csr(8)
    assert ip.tolist() == [0, 2, 2, 5, 5, 5, 6, 6, 6] and di.tolist() == ids.tolist() and va.tolist() == vals.tolist()
