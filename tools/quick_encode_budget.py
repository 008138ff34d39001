#!/usr/bin/env python3
"""Corpus-encode rate (through store_embs, as bench.py's encode leg) per token budget.  python tools/quick_encode_budget.py 8192 16384 32768"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("TQDM_DISABLE", "1")
import bench  # noqa: E402
from scaling_retriever_amd.modeling.llm_encoder import LlamaBiDense  # noqa: E402

dev = torch.device("cuda", 0)
cfg = dict(bench.LION_1B)
model = LlamaBiDense.from_weights(cfg, bench.random_weights(cfg, dev, 0), max_batch_tokens=65536, max_batch_seqs=8192, fp32_planes=0).to(dev).eval()
for budget in [int(a) for a in sys.argv[1:]] or [8192, 16384, 32768]:
    args = argparse.Namespace(encode_passages=65536, token_budget=budget)
    r = bench.encode_leg(args, cfg, model, dev, 0, 1, False)
    print(budget, r["value"], "passages/s", r["roofline"]["achieved"], "TF", "gpu-only", r["roofline"]["achieved_gpu_time_only"], flush=True)
