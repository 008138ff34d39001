"""RCCL executed at least once: this pool has one GPU per box, so the doc-sharded path's collectives
(gather_topk, all_gather_query_reps, all_gather_query_csr - scaling_retriever_amd/distributed.py; the single gather that
replaces the reference's one-process scoring, /root/reference/eval_dense.py:191, eval_sparse.py:114) are driven through a
1-rank backend="nccl" process group with the world-size-1 shortcut switched off.  The payloads must come back bit for bit."""
import os
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %r)
    import torch
    import torch.distributed as dist
    from scaling_retriever_amd import distributed as D
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", init_method="tcp://127.0.0.1:%d", world_size=1, rank=0, device_id=torch.device("cuda", 0))
    assert dist.get_backend() == "nccl"
    D.FORCE_COLLECTIVES = True
    g = torch.Generator(device="cuda").manual_seed(0)
    s = torch.randn((37, 50), device="cuda", generator=g)
    s[3, 7] = float("-inf")
    i = torch.randint(0, 2 ** 32 - 2, (37, 50), device="cuda", generator=g)
    i[5, 9] = -1
    gs, gi = D.gather_topk(s, i, dst=0, compact=False)
    assert gs.shape == (1, 37, 50) and torch.equal(gs[0], s) and torch.equal(gi[0], i)
    # the compacting gather ships only the valid slots (one row of padding here is too little: the full buffer travels) ...
    gs, gi = D.gather_topk(s, i, dst=0)
    assert torch.equal(gs[0], s) and torch.equal(gi[0], i)
    # ... and with most slots padding (what a shard returns after the threshold exchange) the valid entries come back first, in order
    i2 = i.clone()
    i2[:, 9:] = -1
    i2[4, :] = -1
    gs, gi = D.gather_topk(s, i2, dst=0)
    assert gs.shape == (1, 37, 50) and torch.equal(gi[0], i2) and torch.equal(gs[0][:, :9][i2[:, :9] >= 0], s[:, :9][i2[:, :9] >= 0])
    reps = torch.randn((37, 64), device="cuda", generator=g)
    out = D.all_gather_query_reps(reps, 37)
    assert out.shape == (37, 64) and torch.equal(out, reps)
    counts = torch.randint(0, 6, (37,), device="cuda", generator=g)
    row_ptr = torch.cat([torch.zeros(1, dtype=torch.int64, device="cuda"), torch.cumsum(counts, 0)])
    n = int(row_ptr[-1])
    cols = torch.randint(0, 128256, (n,), device="cuda", generator=g).to(torch.int32)
    vals = torch.rand((n,), device="cuda", generator=g)
    p, c, v = D.all_gather_query_csr(row_ptr, cols, vals, 37)
    assert torch.equal(p, row_ptr) and torch.equal(c, cols) and torch.equal(v, vals)
    # the whole sharded dense search on a world of one, through the collectives
    from scaling_retriever_amd.distributed import ShardedDenseRetriever
    from scaling_retriever_amd.scoring import DenseIndexHIP
    rows = torch.randn((5000, 128), device="cuda", generator=g)
    q = torch.randn((9, 128), device="cuda", generator=g)
    r = ShardedDenseRetriever(128)
    r.add_local_rows(rows)
    ss, ii = r.search(q, 20)
    ref = DenseIndexHIP(128)
    ref.add_device_rows(rows)
    es, ei = ref.search(q, 20)
    assert torch.equal(ss, es) and torch.equal(ii, ei)
    # a batch the certified filter serves (> 64 queries): search_begin -> all-reduce(min) over RCCL -> search_finish -> gather
    rows2 = torch.randn((40000, 128), device="cuda", generator=g)
    q2 = torch.randn((160, 128), device="cuda", generator=g)
    r2 = ShardedDenseRetriever(128)
    r2.add_local_rows(rows2)
    c0, _ = r2.index.filter_query_stats()
    ss, ii = r2.search(q2, 300)
    c1, redone = r2.index.filter_query_stats()
    assert c1 - c0 >= 150, (c0, c1, redone)            # the threshold path ran, and the filter certified the batch
    ref2 = DenseIndexHIP(128)
    ref2.add_device_rows(rows2)
    es, ei = ref2.search(q2, 300)
    assert torch.equal(ss, es) and torch.equal(ii, ei)
    dist.barrier()
    dist.destroy_process_group()
    print("rccl one-rank ok")
""")


def test_collectives_through_a_one_rank_rccl_group():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", SCRIPT % (ROOT, port)], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "rccl one-rank ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
