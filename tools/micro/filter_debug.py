#!/usr/bin/env python3
"""Bisect a filtered-vs-exact mismatch: python3 tools/micro/filter_debug.py [n_docs] [corpus] [queries]"""
import os
import sys

import torch

os.environ["SR_DEV_SWITCHES"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import synth  # noqa: E402
from scaling_retriever_amd.scoring import DenseIndexHIP  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8_841_823
corpus = sys.argv[2] if len(sys.argv) > 2 else "aniso_dup"
queries = sys.argv[3] if len(sys.argv) > 3 else "aniso"
nq, H, k = 6980, 2048, 1000
dev = torch.device("cuda")
D = synth.dense_rows(corpus, N, H, dev, seed=11)
Q = synth.dense_queries(queries, nq, H, dev, seed=12, D=D)
ex = DenseIndexHIP(H)
ex.add_device_rows(D)
es, ei = ex.search(Q, k)
for rep in range(3):
    es2, ei2 = ex.search(Q, k)
    print("exact vs exact, rep", rep, bool(torch.equal(es, es2) and torch.equal(ei, ei2)), flush=True)
ex.close()
n_rounds = int(os.environ.get("ROUNDS", "8"))
for rnd in range(n_rounds):
    f = DenseIndexHIP(H)
    f.set_precision("fp32_filtered")
    f.add_device_rows(D)
    for rep in range(3):
        fs, fi = f.search(Q, k)
        bad = (~((fs == es).all(1) & (fi == ei).all(1))).nonzero()[:, 0]
        if bad.numel():
            print("round", rnd, "rep", rep, "stats", f.filter_stats(), f.filter_query_stats(), "queries that differ:", bad.tolist()[:20], flush=True)
            for q in bad.tolist()[:3]:
                d = ((fs[q] != es[q]) | (fi[q] != ei[q])).nonzero()[:, 0]
                j = int(d[0])
                print("   q", q, "first diff at rank", j, "of", d.numel(), "filtered", float(fs[q, j]), int(fi[q, j]), "exact", float(es[q, j]), int(ei[q, j]),
                      "exact id in filtered list:", bool((fi[q] == ei[q, j]).any()), "filtered id in exact list:", bool((ei[q] == fi[q, j]).any()), flush=True)
            # which queries does the filter re-do?  (kp at the maximum certifies the most)
    print("round", rnd, "done", f.filter_query_stats(), flush=True)
    f.close()
