#!/usr/bin/env python3
"""Score-stage timing on a slice of the MSMARCO shape: exact kernel, certified filter, bf16x3 pass alone.
python tools/quick_split_bench.py [n_docs] [n_queries]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scaling_retriever_amd.scoring import DenseIndexHIP  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 6980
H = 2048
g = torch.Generator(device="cuda").manual_seed(1)
D = torch.empty((N, H), dtype=torch.float32, device="cuda")
for r0 in range(0, N, 1 << 20):
    D[r0:r0 + (1 << 20)].normal_(0.0, 0.5 / H ** 0.5, generator=g)
Q = torch.empty((nq, H), dtype=torch.float32, device="cuda").normal_(0.0, 0.5 / H ** 0.5, generator=g)
idx = DenseIndexHIP(H)
idx.add_device_rows(D)
ref = None
for mode, k in (("fp32", 1000), ("fp32_filtered", 1000), ("bf16x3", 2048)):
    idx.set_precision(mode)
    out = idx.search(Q, k)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(3):
        out = idx.search(Q, k)
    torch.cuda.synchronize()
    t = (time.perf_counter() - t) / 3
    tf = 2.0 * nq * N * H / t / 1e12
    note = ""
    if mode == "fp32":
        ref = out
    elif mode == "fp32_filtered":
        note = " identical=%s stats=%s" % (bool(torch.equal(out[0], ref[0]) and torch.equal(out[1], ref[1])), idx.filter_stats())
    print(f"{mode:14s} k={k}: {t * 1e3:8.1f} ms  {tf:7.1f} algorithmic TFLOP/s{note}", flush=True)
