#!/usr/bin/env python3
"""Where search_knn's wall time goes at the full shape: per-piece timestamps of the GPU side (worker thread: sr_dense_search + D2H)
and of the host side (numpy take + tolist per row), to see how much of the two overlaps.  python tools/micro/knn_overlap.py [n_docs]"""
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from scaling_retriever_amd.indexer import DenseFlatIndexer  # noqa: E402
from scaling_retriever_amd.scoring import DenseIndexHIP  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8_841_823
H, NQ, K = 2048, 6980, 1000
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1)
D = torch.empty((N, H), dtype=torch.float32, device=dev)
for r0 in range(0, N, 1 << 20):
    D[r0:r0 + (1 << 20)].normal_(0.0, 0.5 / H ** 0.5, generator=g)
Q = torch.randn((NQ, H), device=dev, generator=g) * (0.5 / H ** 0.5)
index = DenseIndexHIP(H, device=dev)
index.add_device_rows(D)
index.set_precision("fp32_filtered")
fi = DenseFlatIndexer()
fi.hidden_dim, fi.index = H, index
fi._update_id_mapping(np.arange(N).astype("U8").tolist())
table = fi.id_table()
T0 = time.perf_counter()


def now():
    return round((time.perf_counter() - T0) * 1e3, 1)


for rep in range(2):
    t = time.perf_counter(); s, i = fi.search_arrays(Q, K); t_arr = time.perf_counter() - t
    t = time.perf_counter(); lists = [table.take(row).tolist() for row in i]; t_map = time.perf_counter() - t
    del lists
    t = time.perf_counter(); lists = fi.id_lists(i); t_map2 = time.perf_counter() - t
    del lists
    t = time.perf_counter(); ids, sc = fi.search_knn(Q, K); t_knn = time.perf_counter() - t
    del ids
    print(f"rep {rep}: search_arrays {t_arr * 1e3:.0f} ms, numpy take per row {t_map * 1e3:.0f} ms, id_lists {t_map2 * 1e3:.0f} ms, search_knn {t_knn * 1e3:.0f} ms", flush=True)

# instrumented pipeline
for n_chunks, interval in ((4, 5e-3), (2, 5e-3), (8, 5e-3), (4, 1e-4)):
    sys.setswitchinterval(interval)
    per = (NQ + n_chunks - 1) // n_chunks
    bounds = [(c0, min(NQ, c0 + per)) for c0 in range(0, NQ, per)]
    log = []

    def gpu(c):
        a = now()
        r = fi.search_arrays(Q[bounds[c][0]:bounds[c][1]], K)
        log.append(("gpu", c, a, now()))
        return r
    t = time.perf_counter()
    out = []
    with ThreadPoolExecutor(max_workers=1) as pool:
        fut = pool.submit(gpu, 0)
        for c in range(len(bounds)):
            sc, ix = fut.result()
            if c + 1 < len(bounds):
                fut = pool.submit(gpu, c + 1)
            a = now()
            out.extend(fi.id_lists(ix))
            log.append(("map", c, a, now()))
    print(f"chunks {n_chunks} switch interval {interval}: total {(time.perf_counter() - t) * 1e3:.0f} ms", sorted(log, key=lambda x: x[2]), flush=True)
    del out
