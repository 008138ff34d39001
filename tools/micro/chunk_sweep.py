#!/usr/bin/env python3
"""Certified-filter search time against docs per launch of the upper-bound pass (workspace limit x SR_DENSE_LAUNCH_WGS), headline
shape.  python3 tools/micro/chunk_sweep.py [n_docs]"""
import os
import sys
import time

import torch

os.environ["SR_DEV_SWITCHES"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import synth  # noqa: E402
from scaling_retriever_amd.scoring import DenseIndexHIP  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8_841_823
nq, H, k = 6980, 2048, 1000
dev = torch.device("cuda")
D = synth.dense_rows("gauss", N, H, dev, seed=11)
Q = synth.dense_queries("gauss", nq, H, dev, seed=12, D=D)
idx = DenseIndexHIP(H)
idx.set_precision("fp32_filtered")
idx.add_device_rows(D)
ref = None
for gb, wgs in ((4, 2048), (8, 4096), (16, 8192), (32, 16384), (64, 32768), (4, 2048)):
    idx.set_workspace_limit(gb << 30)
    os.environ["SR_DENSE_LAUNCH_WGS"] = str(wgs)
    s, i = idx.search(Q, k)
    if ref is None:
        ref = (s.clone(), i.clone())
    same = torch.equal(s, ref[0]) and torch.equal(i, ref[1])
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(3):
        idx.search(Q, k)
    torch.cuda.synchronize()
    t = (time.perf_counter() - t) / 3
    print(f"workspace {gb:3d} GB, launch_wgs {wgs:6d}: search {t * 1e3:7.1f} ms, certified/re-done {idx.filter_query_stats()}, same results {same}", flush=True)
