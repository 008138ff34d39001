#!/usr/bin/env python3
"""Shard-file ingest rate: embs_*.npy (memory-mapped) -> pinned staging ring -> async H2D into one HBM segment
(DenseIndexHIP.add_npy_file), against the reference's route (np.load every shard, np.concatenate, faiss index.add:
two more host copies, /root/reference/eval_dense.py:113-121, scaling_retriever/indexer.py:198-208).
python tools/quick_ingest_bench.py [GiB] [dir]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scaling_retriever_amd.scoring import DenseIndexHIP  # noqa: E402

gib = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
d = sys.argv[2] if len(sys.argv) > 2 else ("/dev/shm" if os.path.isdir("/dev/shm") else "/tmp")
H = 2048
n = int(gib * (1 << 30) / (4 * H))
path = os.path.join(d, "sr_ingest_bench.npy")
rows = np.lib.format.open_memmap(path, mode="w+", dtype=np.float32, shape=(n, H))
rows[:] = 1.0
rows.flush()
del rows
try:
    for threads, bufs in ((1, 2), (4, 4), (8, 8), (16, 16)):
        idx = DenseIndexHIP(H)
        torch.cuda.synchronize()
        t = time.perf_counter()
        idx.add_host_rows(np.load(path, mmap_mode="r"), n_buffers=bufs, n_threads=threads)
        torch.cuda.synchronize()
        t = time.perf_counter() - t
        print(f"mmap -> pinned ring ({threads} threads, {bufs} x 64 MB buffers) -> HBM: {n * H * 4 / t / 1e9:.1f} GB/s ({gib:.0f} GiB in {t:.2f}s)", flush=True)
        assert float(idx._segments[0][-1, -1]) == 1.0
        idx.close()
        del idx
        torch.cuda.empty_cache()
    for threads, bufs in ((4, 4), (8, 8), (16, 16), (32, 32)):
        idx = DenseIndexHIP(H)
        torch.cuda.synchronize()
        t = time.perf_counter()
        idx.add_npy_file(path, n_buffers=bufs, n_threads=threads)
        torch.cuda.synchronize()
        t = time.perf_counter() - t
        print(f"preadv -> pinned ring ({threads} threads, {bufs} x 64 MB buffers) -> HBM: {n * H * 4 / t / 1e9:.1f} GB/s ({gib:.0f} GiB in {t:.2f}s)", flush=True)
        assert float(idx._segments[0][-1, -1]) == 1.0
        idx.close()
        del idx
        torch.cuda.empty_cache()
    pinned = torch.empty((1 << 28,), dtype=torch.float32, pin_memory=True)      # 1 GiB
    devb = torch.empty_like(pinned, device="cuda")
    devb.copy_(pinned, non_blocking=True)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(4):
        devb.copy_(pinned, non_blocking=True)
    torch.cuda.synchronize()
    print(f"raw pinned H2D copy (the link's rate): {4 * pinned.numel() * 4 / (time.perf_counter() - t) / 1e9:.1f} GB/s", flush=True)
    del pinned, devb
    t = time.perf_counter()
    host = np.load(path)                                    # the reference: whole file into host memory ...
    dev = torch.from_numpy(np.ascontiguousarray(host)).cuda()       # ... then one pageable H2D copy
    torch.cuda.synchronize()
    t = time.perf_counter() - t
    print(f"np.load + pageable copy: {n * H * 4 / t / 1e9:.1f} GB/s", flush=True)
finally:
    os.remove(path)
