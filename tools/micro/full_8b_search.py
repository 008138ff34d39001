#!/usr/bin/env python3
"""The whole MS MARCO corpus at Lion-DS-8B width on ONE MI355X: 8 841 823 x 4096 fp32 (145 GB) + its fp16 filter plane (72 GB)
resident, 6 980 queries, top-1000 - certified filter vs the exact kernel, bit for bit.  (BASELINE configs[4] spreads this over
8 GPUs; this is the capacity point.)  python3 tools/micro/full_8b_search.py [n_docs]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from scaling_retriever_amd.scoring import DenseIndexHIP  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8_841_823
nq, H, k = 6980, 4096, 1000
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(3)
D = torch.empty((N, H), dtype=torch.float32, device=dev)
for r0 in range(0, N, 1 << 19):
    D[r0:r0 + (1 << 19)].normal_(0.0, 0.5 / H ** 0.5, generator=g)
Q = torch.empty((nq, H), dtype=torch.float32, device=dev).normal_(0.0, 0.5 / H ** 0.5, generator=g)
idx = DenseIndexHIP(H, device=dev)
idx.add_device_rows(D)
torch.cuda.synchronize()
t = time.perf_counter()
es, ei = idx.search(Q, k)
torch.cuda.synchronize()
t_exact = time.perf_counter() - t
idx.set_precision("fp32_filtered")
fs, fi = idx.search(Q, k)
c0, r0 = idx.filter_query_stats()
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(2):
    idx.search(Q, k)
torch.cuda.synchronize()
t_f = (time.perf_counter() - t) / 2
c1, r1 = idx.filter_query_stats()
free, total = torch.cuda.mem_get_info()
print(f"{N} x {H}: exact kernel {t_exact * 1e3:.0f} ms ({nq / t_exact:.0f} queries/s, {2.0 * nq * N * H / t_exact / 1e12:.1f} TFLOP/s); "
      f"certified filter {t_f * 1e3:.0f} ms ({nq / t_f:.0f} queries/s), {(c1 - c0) // 2} certified / {(r1 - r0) // 2} re-done per search; "
      f"bit-identical: {bool(torch.equal(fs, es) and torch.equal(fi, ei))}; HBM in use {(total - free) / 1e9:.0f} of {total / 1e9:.0f} GB", flush=True)
