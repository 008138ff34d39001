#!/usr/bin/env python3
"""SparseIndexer.index rate at Lion-SP-1B dims (bench.py's sparse_index leg).  python3 tools/quick_sparse_index.py [passages]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("TQDM_DISABLE", "1")
import bench  # noqa: E402

args = argparse.Namespace(sparse_index_passages=int(sys.argv[1]) if len(sys.argv) > 1 else 65536, token_budget=16384)
r = bench.sparse_index_leg(args, torch.device("cuda", 0))
print("sparse_index", r["passages_per_s"], "passages/s", r["roofline"]["frac"], "of peak, L0_d", r["L0_d"], flush=True)
