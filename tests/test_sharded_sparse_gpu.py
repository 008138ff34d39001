"""Doc-sharded sparse retrieval (BASELINE.json north_star: the corpus shards by document; the reference scores on one
process, /root/reference/eval_sparse.py:114, after merge_indexes): per-rank CSR over local docs + ONE gather of per-shard
top-k + sr_topk_merge must equal the single-index search bit for bit - ids, fp32 scores and counts."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _random_index(rng, V, N, density):
    dense = (rng.random((N, V)) < density) * (rng.random((N, V), dtype=np.float32) + 0.05).astype(np.float32)
    ids, vals, ptr = [], [], [0]
    for t in range(V):
        nz = np.nonzero(dense[:, t])[0]
        ids.append(nz.astype(np.int32))
        vals.append(dense[nz, t].astype(np.float32))
        ptr.append(ptr[-1] + len(nz))
    return np.array(ptr, np.int64), np.concatenate(ids), np.concatenate(vals)


def _queries(rng, V, nq, lo, hi):
    cnt = rng.integers(lo, hi + 1, size=nq)
    ptr = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64)
    cols = np.concatenate([np.sort(rng.choice(V, size=c, replace=False)) for c in cnt]).astype(np.int32)
    return ptr, cols, (rng.random(len(cols), dtype=np.float32) + 0.1)


@pytest.mark.parametrize("W", [2, 3, 8])
def test_fake_world_equals_single_index(W):
    from scaling_retriever_amd.distributed import ShardedSparseRetriever
    from scaling_retriever_amd.scoring import SparseIndexHIP, topk_merge
    rng = np.random.default_rng(W)
    V, N, k = 300, 20011, 50                      # 3 doc tiles of 8192 on one index, N / W docs per shard
    ptr, ids, vals = _random_index(rng, V, N, 0.02)
    q_ptr, q_cols, q_vals = _queries(rng, V, 40, 0, 12)
    single = SparseIndexHIP(ptr, ids, vals, N)
    es, ei, ec = single.search(q_ptr, q_cols, q_vals, k)
    term = np.repeat(np.arange(V), np.diff(ptr))
    parts_s, parts_i = [], []
    for r in range(W):
        sel = (ids % W) == r                      # what index_dir_{r} holds: postings of global rows r, r + W, ...
        ptr_r = np.concatenate([[0], np.cumsum(np.bincount(term[sel], minlength=V))]).astype(np.int64)
        n_r = int(ids[sel].max()) + 1 if sel.any() else 0      # nb_docs() of the shard = max g_row + 1
        shard = ShardedSparseRetriever(ptr_r, ids[sel], vals[sel], n_r, rank=r, world_size=W)
        s, i, c = shard.index.search(q_ptr, q_cols, q_vals, k, id_base=r, id_stride=W)
        assert bool(((i < 0) | (i % W == r)).all())
        parts_s.append(s)
        parts_i.append(i)
    ms, mi = topk_merge(torch.stack(parts_s), torch.stack(parts_i), pad_score=0.0)
    assert torch.equal(mi, ei) and torch.equal(ms, es)
    assert torch.equal((mi >= 0).sum(1).to(torch.int32), ec)
    with pytest.raises(ValueError):               # postings of another shard's documents
        ShardedSparseRetriever(ptr, ids, vals, N, rank=0, world_size=W)


_WORKER = r'''
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
from test_sharded_sparse_gpu import _random_index, _queries
from scaling_retriever_amd.distributed import ShardedSparseRetriever, all_gather_query_csr, query_slice
from scaling_retriever_amd.scoring import SparseIndexHIP
torch.cuda.set_device(0)
dist.init_process_group("gloo")
rank, W = dist.get_rank(), dist.get_world_size()
rng = np.random.default_rng(11)
V, N, k, nq = 200, 17000, 30, 25
ptr, ids, vals = _random_index(rng, V, N, 0.03)
q_ptr, q_cols, q_vals = _queries(rng, V, nq, 1, 10)
lo, hi = query_slice(nq, rank, W)                 # every rank "encoded" only its block of the queries
p, c, v = all_gather_query_csr(torch.from_numpy(q_ptr[lo:hi + 1] - q_ptr[lo]).cuda(), torch.from_numpy(q_cols[q_ptr[lo]:q_ptr[hi]]).cuda(),
                               torch.from_numpy(q_vals[q_ptr[lo]:q_ptr[hi]]).cuda(), nq)
term = np.repeat(np.arange(V), np.diff(ptr))
sel = (ids % W) == rank
ptr_r = np.concatenate([[0], np.cumsum(np.bincount(term[sel], minlength=V))]).astype(np.int64)
shard = ShardedSparseRetriever(ptr_r, ids[sel], vals[sel], int(ids[sel].max()) + 1)
s, i, cnt = shard.search(p, c, v, k)
if rank == 0:
    es, ei, ec = SparseIndexHIP(ptr, ids, vals, N).search(q_ptr, q_cols, q_vals, k)
    assert torch.equal(i, ei) and torch.equal(s, es) and torch.equal(cnt, ec)
    print("SHARDED_SPARSE_OK")
else:
    assert s is None
dist.barrier(); dist.destroy_process_group()
'''


def test_sharded_sparse_retriever_two_processes(tmp_path):
    """Two ranks sharing this GPU (gloo rendezvous, real HIP kernels): query CSR all-gather, per-shard search, ONE gather,
    merge on rank 0."""
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29655", str(script), ROOT],
                         capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    assert "SHARDED_SPARSE_OK" in out.stdout


def test_eval_sparse_driver_two_ranks_no_merge_pass(golden_dir, tmp_path):
    """eval_sparse.py --task_name indexing then retrieval under torchrun with 2 ranks (sharing this GPU, SR_SHARE_GPU=1):
    retrieval reads index_0 / index_1 directly.  Same run as merge_indexes + the single-process retrieval of the reference."""
    from golden_weights import make_weights
    from test_eval_drivers import _texts, _write_model
    z = np.load(os.path.join(golden_dir, "enc_tiny_a.npz"))
    cfg = json.loads(str(z["config_json"]))
    w = make_weights(cfg, int(z["weight_seed"]))
    rng = np.random.default_rng(4)
    lora_s, _ = _write_model(str(tmp_path), cfg, w, rng)["sparse"]
    docs, queries = _texts(rng, 41, 3, 20), _texts(rng, 7, 2, 6)
    with open(tmp_path / "corpus.tsv", "w") as f:
        f.writelines(f"d{i}\t{t}\n" for i, t in enumerate(docs))
    with open(tmp_path / "queries.tsv", "w") as f:
        f.writelines(f"q{i}\t{t}\n" for i, t in enumerate(queries))
    index_dir = str(tmp_path / "sp" / "index")
    os.makedirs(tmp_path / "sp")
    env = dict(os.environ, SR_SHARE_GPU="1", MASTER_ADDR="127.0.0.1", TQDM_DISABLE="1")
    run2 = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
            "--master-port", "29677", os.path.join(ROOT, "eval_sparse.py"), "--model_name_or_path", lora_s]
    for extra in (["--task_name", "indexing", "--corpus_path", str(tmp_path / "corpus.tsv"), "--index_dir", index_dir,
                   "--doc_max_length", "16", "--token_budget", "64", "--tokenize_workers", "0"],
                  ["--task_name", "retrieval", "--query_path", str(tmp_path / "queries.tsv"), "--index_dir", index_dir,
                   "--out_dir", str(tmp_path / "out2"), "--top_k", "10", "--query_max_length", "8", "--eval_batch_size", "3"]):
        out = subprocess.run(run2 + extra, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-3000:]
    assert sorted(os.listdir(tmp_path / "sp")) == ["index_0", "index_1"]
    run_sharded = json.load(open(tmp_path / "out2" / "run.json"))
    # the reference's route: merge the per-rank indexes, then ONE process
    import eval_sparse
    from scaling_retriever_amd.utils.inverted_index import merge_indexes
    merge_indexes(lora_s, index_name="index", index_dir=str(tmp_path / "sp"))
    eval_sparse.main(["--task_name", "retrieval", "--model_name_or_path", lora_s, "--query_path", str(tmp_path / "queries.tsv"),
                      "--index_dir", index_dir, "--out_dir", str(tmp_path / "out1"), "--top_k", "10", "--query_max_length", "8",
                      "--eval_batch_size", "3"])
    run_single = json.load(open(tmp_path / "out1" / "run.json"))
    assert set(run_sharded) == set(run_single) == {f"q{i}" for i in range(7)}
    for q in run_single:
        assert len(run_sharded[q]) == len(run_single[q]) == 10
        common = set(run_sharded[q]) & set(run_single[q])
        assert len(common) >= 9, q                                   # query batches differ between the two routes: bf16 noise only
        for pid in common:
            assert abs(run_sharded[q][pid] - run_single[q][pid]) < 2e-2 * max(1.0, abs(run_single[q][pid]))
