#!/usr/bin/env python3
"""Corpus-encode rate at Lion-DS-8B dims (bench.py's config5 encode, through store_embs).  python3 tools/quick_encode_8b.py [passages]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("TQDM_DISABLE", "1")
import bench  # noqa: E402
from scaling_retriever_amd.modeling.llm_encoder import LlamaBiDense  # noqa: E402

dev = torch.device("cuda", 0)
cfg = dict(bench.LION_8B)
model = LlamaBiDense.from_weights(cfg, bench.random_weights(cfg, dev, 0), max_batch_tokens=65536, max_batch_seqs=8192, fp32_planes=0).to(dev).eval()
args = argparse.Namespace(encode_passages=int(sys.argv[1]) if len(sys.argv) > 1 else 8192, token_budget=16384)
r = bench.encode_leg(args, cfg, model, dev, 0, 1, False, flop_per_token=bench.FLOP_PER_TOKEN_8B, layers_ref=32)
print("8B", r["value"], "passages/s", r["roofline"]["achieved"], "TF", "padded-128", r["padded_batch_128_mode"]["passages_per_s"], flush=True)
