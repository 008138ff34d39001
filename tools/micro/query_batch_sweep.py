"""fp32-regime query encode of the 6 980 synthetic Dev queries per loader batch size (tokens per call decide how many
whole rounds of 256 x 256 tiles the layer GEMMs make).  python tools/micro/query_batch_sweep.py [batch ...]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench  # noqa: E402
from scaling_retriever_amd.modeling.llm_encoder import LlamaBiDense  # noqa: E402

dev = torch.device("cuda", 0)
cfg = dict(bench.LION_1B)
w = bench.random_weights(cfg, dev, 0)
model = LlamaBiDense.from_weights(cfg, dict(w), max_batch_tokens=65536, max_batch_seqs=8192, fp32_planes=16).to(dev).eval()
model.base_model.precision = "fp32"
for qb in [int(a) for a in sys.argv[1:]] or [2048, 1880, 1800, 3760, 940, 6980]:
    batches, lens = bench.synth_batches(6980, qb, 2.1, 0.35, 4, 64, cfg["vocab_size"], 2, dev)
    best = 1e9
    for rep in range(4):
        torch.cuda.synchronize()
        t = time.perf_counter()
        for i, m in batches:
            model.query_encode(input_ids=i, attention_mask=m)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t)
    toks = [int(m.sum()) for _, m in batches]
    print(f"query batch {qb:5d}: {best * 1e3:7.1f} ms, tokens per call {toks}", flush=True)
