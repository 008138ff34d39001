#!/usr/bin/env python3
"""How long SparseIndexHIP(...) takes at the MSMARCO shape once the CSR sits in HBM: validation + the exact kernels' block tables + the
certified scorer's operand, forward index and run tables (one-off per index load).  python3 tools/micro/sparse_index_load.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tools import synth  # noqa: E402
from scaling_retriever_amd.scoring import SparseIndexHIP  # noqa: E402

dev = torch.device("cuda", 0)
V, N = 128256, 8841823
indptr, ids, vals, _ = synth.build_index(V, N, 128, dev, 0)
torch.cuda.synchronize()
for rep in range(3):
    t = time.perf_counter()
    idx = SparseIndexHIP(indptr, ids, vals, N)
    torch.cuda.synchronize()
    print(f"SparseIndexHIP over {ids.numel()} postings: {(time.perf_counter() - t) * 1e3:.1f} ms", idx.cert_stats() if hasattr(idx, "cert_stats") else "", flush=True)
    del idx
    torch.cuda.empty_cache()
