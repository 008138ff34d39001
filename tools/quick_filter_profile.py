#!/usr/bin/env python3
"""One certified-filter search at the full MSMARCO shape, for `rocprofv3 --kernel-trace --stats` (per-kernel breakdown of the
score stage).  python3 tools/quick_filter_profile.py [n_docs] [n_queries] [repeats]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scaling_retriever_amd.scoring import DenseIndexHIP  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8_841_823
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 6980
rep = int(sys.argv[3]) if len(sys.argv) > 3 else 2
H = 2048
g = torch.Generator(device="cuda").manual_seed(1)
D = torch.empty((N, H), dtype=torch.float32, device="cuda")
for r0 in range(0, N, 1 << 20):
    D[r0:r0 + (1 << 20)].normal_(0.0, 0.5 / H ** 0.5, generator=g)
Q = torch.empty((nq, H), dtype=torch.float32, device="cuda").normal_(0.0, 0.5 / H ** 0.5, generator=g)
idx = DenseIndexHIP(H)
idx.add_device_rows(D)
idx.set_precision("fp32_filtered")
idx.search(Q, 1000)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(rep):
    idx.search(Q, 1000)
torch.cuda.synchronize()
print(f"filtered search: {(time.perf_counter() - t) / rep * 1e3:.1f} ms per call, stats {idx.filter_stats()}", flush=True)
