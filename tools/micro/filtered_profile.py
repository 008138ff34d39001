#!/usr/bin/env python3
"""Five certified-filter searches at the headline shape, for rocprofv3 --kernel-trace --stats (what a search is made of)."""
import os
import sys

import torch

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
for _ in range(5):
    idx.search(Q, k)
torch.cuda.synchronize()
