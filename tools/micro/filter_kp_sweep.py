#!/usr/bin/env python3
"""Certified filter: candidates per query (kp) against certification rate and search time, on the headline corpus and the
robustness corpora at full shape.  python3 tools/micro/filter_kp_sweep.py [n_docs]"""
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
for corpus, queries in (("gauss", "gauss"), ("aniso", "aniso"), ("aniso_dup", "aniso"), ("aniso_dup", "near_docs")):
    D = synth.dense_rows(corpus, N, H, dev, seed=11)
    Q = synth.dense_queries(queries, nq, H, dev, seed=12, D=D)
    for kp in (1536, 2048, 3072):
        os.environ["SR_FILTER_KP"] = str(kp)
        idx = DenseIndexHIP(H)
        idx.set_precision("fp32_filtered")
        idx.add_device_rows(D)
        idx.search(Q, k)
        c0, r0 = idx.filter_query_stats()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(2):
            idx.search(Q, k)
        torch.cuda.synchronize()
        t = (time.perf_counter() - t) / 2
        print(f"{corpus:10s} {queries:10s} kp {kp}: {c0} certified, {r0} re-done, search {t * 1e3:.1f} ms", flush=True)
        idx.close()
    del D, Q
    torch.cuda.empty_cache()
