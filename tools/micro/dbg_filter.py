import sys, os, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from scaling_retriever_amd.scoring import DenseIndexHIP
H = 2048
for N, nq in ((500_000, 300), (2_000_000, 1000), (2_000_000, 6980), (8_841_823, 1000)):
    g = torch.Generator(device="cuda").manual_seed(1)
    D = torch.empty((N, H), dtype=torch.float32, device="cuda")
    for r0 in range(0, N, 1 << 20):
        D[r0:r0 + (1 << 20)].normal_(0.0, 0.5 / H ** 0.5, generator=g)
    Q = torch.empty((nq, H), dtype=torch.float32, device="cuda").normal_(0.0, 0.5 / H ** 0.5, generator=g)
    idx = DenseIndexHIP(H); idx.add_device_rows(D)
    es, ei = idx.search(Q, 1000)
    idx.set_precision("bf16x3"); as_, ai = idx.search(Q, 2048)
    idx.set_precision("fp32_filtered"); fs, fi = idx.search(Q, 1000)
    bad = (~((fi == ei).all(1) & (fs == es).all(1))).nonzero()[:, 0]
    print(f"N {N} nq {nq}: stats {idx.filter_stats()} differing queries {bad.numel()}", flush=True)
    if bad.numel():
        q = int(bad[0])
        cand = set(ai[q].tolist()); ex = ei[q].tolist()
        missing = [d for d in ex if d not in cand]
        print("  first bad query", q, "exact ids missing from the approx top-2048:", len(missing))
        pos = (fi[q] != ei[q]).nonzero()[:, 0]
        print("  id mismatches at ranks", pos[:10].tolist(), "score mismatches", int((fs[q] != es[q]).sum()))
        if len(pos):
            r = int(pos[0]); print("   exact", ei[q, r].item(), es[q, r].item(), "filtered", fi[q, r].item(), fs[q, r].item())
        # approx top-kp sanity: recompute approx? compare approx scores with exact ones for shared docs
        ex_map = dict(zip(ei[q].tolist(), es[q].tolist()))
        dif = [abs(s - ex_map[d]) for d, s in zip(ai[q].tolist(), as_[q].tolist()) if d in ex_map]
        print("   max |S_a - S_x| on shared docs", max(dif) if dif else None, "a_k", as_[q, 999].item(), "a_kp", as_[q, 2047].item())
    idx.close(); del idx, D, Q
    torch.cuda.empty_cache()
