#!/usr/bin/env python3
"""Query-encode time at Lion-DS-1B dims for the 6 980 synthetic Dev queries: bf16 regime vs the fp32 regime, per query-batch
size.  planes: 16 (default) = two fp16 planes of power-of-two scaled rows, 2 / 3 = bf16 planes.
python tools/quick_query_encode.py [planes ...]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from scaling_retriever_amd.modeling.llm_encoder import LlamaBiDense  # noqa: E402

dev = torch.device("cuda", 0)
cfg = dict(bench.LION_1B)
w = bench.random_weights(cfg, dev, 0)
for planes in [int(a) for a in sys.argv[1:]] or [16]:
    model = LlamaBiDense.from_weights(cfg, dict(w), max_batch_tokens=65536, max_batch_seqs=8192, fp32_planes=planes).to(dev).eval()
    for qb in (512, 2048, 6980):
        batches, lens = bench.synth_batches(6980, qb, 2.1, 0.35, 4, 64, cfg["vocab_size"], 2, dev)
        for prec in ("bf16", "fp32"):
            model.base_model.precision = prec
            for rep in range(3):
                torch.cuda.synchronize()
                t = time.perf_counter()
                for i, m in batches:
                    model.query_encode(input_ids=i, attention_mask=m)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t
            nseg = {3: 6, 2: 3, 16: 3}[planes] if prec == "fp32" else 1
            tf = lens.sum() * bench.FLOP_PER_TOKEN_1B * nseg / dt / 1e12
            print(f"planes {planes} batch {qb:5d} {prec}: {dt * 1e3:8.1f} ms  ({int(lens.sum())} tokens, {tf:7.1f} TFLOP/s of bf16 MFMA work)", flush=True)
    del model
    torch.cuda.empty_cache()
