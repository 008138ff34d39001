#!/usr/bin/env python3
"""A/B of the certified filter's upper-bound pass (dense_split_kernel<true>) at the full MSMARCO shape: average launch
duration (HIP events inside the library) under the dev switches.  python3 tools/micro/split_ab.py [n_docs]
The SR_SPLIT_DIAG runs need a diagnostic build: make -C scaling_retriever_amd/csrc clean all EXTRA=-DSR_DIAG_BUILD (never shipped)."""
import ctypes
import os
import sys

import torch

os.environ["SR_DEV_SWITCHES"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from scaling_retriever_amd import _lib  # noqa: E402
from scaling_retriever_amd.scoring import DenseIndexHIP  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8_841_823
nq, H = 6980, 2048
g = torch.Generator(device="cuda").manual_seed(1)
D = torch.empty((N, H), dtype=torch.float32, device="cuda")
for r0 in range(0, N, 1 << 20):
    D[r0:r0 + (1 << 20)].normal_(0.0, 0.5 / H ** 0.5, generator=g)
Q = torch.empty((nq, H), dtype=torch.float32, device="cuda").normal_(0.0, 0.5 / H ** 0.5, generator=g)
idx = DenseIndexHIP(H)
idx.add_device_rows(D)
lib = _lib.load()


def run(mode, k, env, reps=2):
    for kk in ("SR_SPLIT_PERSIST", "SR_SPLIT_DIAG", "SR_SPLIT_XCD", "SR_FILTER_KP", "SR_SPLIT_SEG", "SR_SPLIT_STAMPS", "SR_DENSE_LAUNCH_WGS"):
        os.environ.pop(kk, None)
    os.environ.update(env)
    idx.set_precision(mode)
    idx.search(Q, k)
    _lib.check(lib.sr_dense_index_profile(idx._h, 1))
    torch.cuda.synchronize()
    import time
    t = time.perf_counter()
    for _ in range(reps):
        idx.search(Q, k)
    torch.cuda.synchronize()
    t = (time.perf_counter() - t) / reps
    n_l, ms, fl, by = ctypes.c_int64(0), ctypes.c_double(0), ctypes.c_double(0), ctypes.c_double(0)
    _lib.check(lib.sr_dense_index_profile_read(idx._h, ctypes.byref(n_l), ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(by)))
    _lib.check(lib.sr_dense_index_profile(idx._h, 0))
    print(f"{mode:14s} {str(env):60s} search {t * 1e3:7.1f} ms, {n_l.value // reps} launches, {ms.value / max(1, n_l.value):.4f} ms per launch, "
          f"{fl.value / (ms.value * 1e-3) / 1e12:.0f} TFLOP/s algorithmic", flush=True)


for rep in range(2):
    run("fp32_filtered", 1000, {})
    run("fp32_filtered", 1000, {"SR_DENSE_LAUNCH_WGS": "4096"})
    run("fp32_filtered", 1000, {"SR_DENSE_LAUNCH_WGS": "1024"})
