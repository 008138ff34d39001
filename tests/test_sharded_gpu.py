"""GPU: the N > 1 path with real HIP kernels - 2 processes (gloo rendezvous, both on cuda:0), each holding the doc
shard rank, rank + W, ... in HBM, ShardedDenseRetriever.search = local sr_dense_search + ONE gather + sr_topk_merge
on rank 0; the result must equal the single-index search bit for bit.  (On the 8-GPU node the same code runs with
backend "nccl" = RCCL; only the transport differs.)"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

_WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from scaling_retriever_amd.distributed import ShardedDenseRetriever, shard_rows
from scaling_retriever_amd.scoring import DenseIndexHIP
torch.cuda.set_device(0)
dist.init_process_group("gloo")
rank, W = dist.get_rank(), dist.get_world_size()
g = torch.Generator(device="cuda").manual_seed(0)
n, h, k = 30001, 128, 200
D = torch.randn((n, h), device="cuda", generator=g)
Q = torch.randn((150, h), device="cuda", generator=g)           # identical on every rank (same seed)
from scaling_retriever_amd.distributed import all_gather_query_reps, query_slice
lo, hi = query_slice(Q.shape[0], rank, W)                        # each rank "encodes" only its block of queries
Qg = all_gather_query_reps(Q[lo:hi].contiguous(), Q.shape[0])
assert torch.equal(Qg, Q)
r = ShardedDenseRetriever(h)
assert (r.rank, r.world_size) == (rank, W)
r.add_local_rows(D[torch.arange(rank, n, W, device="cuda")].contiguous())
for nq in (150, 7):                                              # MFMA-tiled and streaming kernels
    s, i = r.search(Q[:nq].contiguous(), k)
    if rank == 0:
        full = DenseIndexHIP(h); full.add_device_rows(D)
        fs, fi = full.search(Q[:nq].contiguous(), k)
        assert torch.equal(i, fi) and torch.equal(s, fs), nq
    else:
        assert s is None and i is None
if rank == 0: print("SHARDED_OK")
dist.barrier(); dist.destroy_process_group()
'''


def test_sharded_dense_retriever_two_processes(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29633", str(script), ROOT],
                         capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    assert "SHARDED_OK" in out.stdout


def test_bench_multi_rank_path_dry_run():
    """bench.py's N > 1 path (query-sharded encode + all-gather, doc-sharded search, one gather of the per-shard top-k,
    merge on rank 0, max-over-ranks timing, the precision-mode and small-batch legs) end to end with two ranks sharing
    this GPU over gloo (SR_BENCH_SHARE_GPU=1; RCCL refuses two ranks on one device).  Small corpus: plumbing, not speed."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SR_BENCH_SHARE_GPU="1")
    env.pop("WORLD_SIZE", None)
    # no launcher on the command line: `bench.py --gpus 2` starts its two ranks itself (the driver's command shape)
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--n-docs", "300000",
           "--encode-passages", "4096", "--layers", "2", "--no-cpu-baseline", "--sparse-docs", "200000"]
    out = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    res = json.loads(line)
    assert res["n_gpus"] == 2 and res["scaling"] == "strong" and res["value"] > 0
    assert res["config"]["ranks"]["world_size"] == 2 and len(res["config"]["ranks"]["shard_docs"]) == 2
    assert res["roofline"]["bound"] == "mfma" and 0 < res["roofline"]["frac"] < 1 and res["cpu_baseline"] is None
    assert res["small_batch"][0]["nq"] == 1
    assert res["encode"]["passages_per_s"] > 0 and res["encode"]["sample_passages"] == 4096
    # BASELINE configs[2] doc-sharded over the two ranks: one gather of the per-shard top-k, merged rows checked against the oracle
    assert res["sparse"]["sharded"] is True and res["sparse"]["n_gpus"] == 2 and res["sparse"]["qps"] > 0
    assert res["sparse"]["oracle_bit_exact_queries"] >= 8 and res["sparse"]["redone_exact"] == 0
    assert len(line) < 4096 and out.stdout.rstrip().endswith(line)             # the contract line is the LAST line of stdout, and compact
    # the full record goes to stderr (and to gpurun_out/bench_detail.json)
    detail = json.loads([l for l in out.stderr.splitlines() if l.startswith("[bench detail] ")][-1][len("[bench detail] "):])
    assert detail["fast_mode"]["value"] > 0 and detail["value"] == res["value"] and detail["config"]["ranks"] == res["config"]["ranks"]
    assert len(detail["sparse"]["shard_postings"]) == 2 and detail["sparse"]["certified_scorer_on_every_rank"] is True


@pytest.mark.parametrize("W", [2, 8])
def test_sharded_threshold_exchange_equals_single_index(W):
    """sr_dense_search_begin / _finish: W shards in one process, each returns a lower bound of its
    ceil(k / W)-th best exact score; with the minimum over the shards as threshold every shard re-scores only what can reach the
    GLOBAL top-k, and the merge of the (padded) shard outputs equals the single index bit for bit - ids and fp32 scores - on
    Gaussian data, on data with near-duplicates around the cut, and with a zero query (re-done exactly on every shard)."""
    import torch
    from scaling_retriever_amd.scoring import DenseIndexHIP, topk_merge
    g = torch.Generator(device="cuda").manual_seed(W)
    n, h, k, nq = 60001, 256, 300, 130
    D = torch.randn((n, h), device="cuda", generator=g)
    D[1000:1400] = D[7] * (1.0 + 1e-6 * torch.arange(400, device="cuda")[:, None])        # 400 near-duplicates of one row
    Q = torch.randn((nq, h), device="cuda", generator=g)
    Q[3] = D[7]
    Q[11] = 0
    full = DenseIndexHIP(h)
    full.add_device_rows(D)
    es, ei = full.search(Q, k)
    shards = []
    for r in range(W):
        ix = DenseIndexHIP(h)
        ix.set_precision("fp32_filtered")
        ix.add_device_rows(D[torch.arange(r, n, W, device="cuda")].contiguous(), id_base=r, id_stride=W)
        shards.append(ix)
    lowers = torch.stack([ix.search_begin(Q, k, W) for ix in shards])
    thr = lowers.min(0).values
    outs = [ix.search_finish(Q, k, thr) for ix in shards]
    ms, mi = topk_merge(torch.stack([o[0] for o in outs]), torch.stack([o[1] for o in outs]))
    assert torch.equal(mi, ei) and torch.equal(ms, es)
    # the point of the exchange: a shard returns about k / W candidates, the rest of its rows is padding
    kept = torch.stack([(o[1] >= 0).sum(1) for o in outs]).float()
    ok = torch.ones(nq, dtype=torch.bool, device="cuda")
    ok[11] = False                                   # the zero query is re-done exactly: a full local top-k
    ok[3] = False                                    # the near-duplicate block sits on the cut of this one
    assert float(kept[:, ok].mean()) < 0.6 * k if W == 8 else True
    # without a threshold the pair is a plain search
    s0, i0 = shards[0].search_finish(Q, k, None)
    s1, i1 = shards[0].search(Q, k)
    assert torch.equal(s0, s1) and torch.equal(i0, i1)


def test_eval_dense_driver_two_ranks(golden_dir, tmp_path):
    """eval_dense.py --task_name write_doc_embeds then retrieval under torchrun with 2 ranks (sharing this GPU, SR_SHARE_GPU=1):
    each rank encodes its passages and, at retrieval, scores the shard files it loads; the threshold exchange, the gather of the
    per-shard top-k and the merge give the run of the single-process route (the reference scores on one process,
    /root/reference/eval_dense.py:191)."""
    import json
    import numpy as np
    from golden_weights import make_weights
    from test_eval_drivers import _texts, _write_model
    z = np.load(os.path.join(golden_dir, "enc_tiny_a.npz"))
    cfg = json.loads(str(z["config_json"]))
    w = make_weights(cfg, int(z["weight_seed"]))
    rng = np.random.default_rng(5)
    models = _write_model(str(tmp_path), cfg, w, rng)
    lora, _ = models["dense"]
    docs, queries = _texts(rng, 90, 3, 20), _texts(rng, 7, 2, 6)
    with open(tmp_path / "corpus.tsv", "w") as f:
        for i, t in enumerate(docs):
            f.write(f"d{i}\t{t}\n")
    with open(tmp_path / "queries.tsv", "w") as f:
        for i, t in enumerate(queries):
            f.write(f"q{i}\t{t}\n")
    env = dict(os.environ, SR_SHARE_GPU="1", MASTER_ADDR="127.0.0.1", TQDM_DISABLE="1")
    run2 = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
            "--master-port", "29691", os.path.join(ROOT, "eval_dense.py"), "--model_name_or_path", lora]
    emb2 = str(tmp_path / "embs2")
    for extra in (["--task_name", "write_doc_embeds", "--corpus_path", str(tmp_path / "corpus.tsv"), "--doc_embed_dir", emb2,
                   "--doc_max_length", "16", "--chunk_size", "32", "--token_budget", "64", "--tokenize_workers", "0"],
                  ["--task_name", "retrieval", "--query_path", str(tmp_path / "queries.tsv"), "--doc_embed_dir", emb2,
                   "--out_dir", str(tmp_path / "out2"), "--top_k", "10", "--query_max_length", "8"]):
        out = subprocess.run(run2 + extra, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-3000:]
    run_sharded = json.load(open(tmp_path / "out2" / "run.json"))
    # one process over the same shard files
    sys.path.insert(0, ROOT)
    import eval_dense
    eval_dense.main(["--task_name", "retrieval", "--model_name_or_path", lora, "--query_path", str(tmp_path / "queries.tsv"),
                     "--doc_embed_dir", emb2, "--out_dir", str(tmp_path / "out1"), "--top_k", "10", "--query_max_length", "8"])
    run_single = json.load(open(tmp_path / "out1" / "run.json"))
    assert run_sharded == run_single
    assert all(len(v) == 10 for v in run_sharded.values()) and len(run_sharded) == 7
