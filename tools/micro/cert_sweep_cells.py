import os, sys, time, json
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
os.environ["SR_DEV_SWITCHES"] = "1"
import torch
from synth import build_index, build_queries
from scaling_retriever_amd.scoring import SparseIndexHIP
dev = torch.device("cuda", 0)
V, N, k, nq = 128256, 8_841_823, 1000, 2048
for L0_d, L0_q in ((128, 64), (256, 32), (256, 64), (128, 32)):
    indptr, doc_ids, vals, df = build_index(V, N, L0_d, dev, 3)
    idx = SparseIndexHIP(indptr, doc_ids, vals, N)
    q = build_queries(V, nq, L0_q, dev, 4)
    idx.search(*q, k); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2): r = idx.search(*q, k)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 2
    os.environ["SR_SPARSE_CERT_SEARCH"] = "0"
    r0 = idx.search(*q, k); torch.cuda.synchronize()
    t0 = time.perf_counter(); r0 = idx.search(*q, k); torch.cuda.synchronize(); dt0 = time.perf_counter() - t0
    os.environ.pop("SR_SPARSE_CERT_SEARCH")
    print(L0_d, L0_q, "cert", round(nq / dt), "exact", round(nq / dt0), "same", all(torch.equal(a, b) for a, b in zip(r, r0)), idx.cert_stats()["redone_exact"], flush=True)
    idx.close(); del idx, indptr, doc_ids, vals
    torch.cuda.empty_cache()
