#!/usr/bin/env python3
"""A/B of the certified filter's upper-bound pass (dense_split_kernel<true>) at the full MSMARCO shape: per setting of the dev
switches given on the command line, the search time and the kernel's average launch duration (HIP events inside the library,
sr_dense_index_profile).  Every setting's result is compared with the first one's (ids and scores, bit for bit).
  SR_DEV_SWITCHES=1 python3 tools/split_ab.py "SR_SPLIT_BLOCKTEST=1" "SR_SPLIT_BLOCKTEST=0" [--docs N] [--exact]"""
import ctypes
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scaling_retriever_amd import _lib  # noqa: E402
from scaling_retriever_amd.scoring import DenseIndexHIP  # noqa: E402

args = [a for a in sys.argv[1:]]
N = 8_841_823
exact = False
if "--docs" in args:
    i = args.index("--docs"); N = int(args[i + 1]); del args[i:i + 2]
if "--lib" in args:                        # another build of the library (same box, same inputs): tools/ab_libs/*.so
    i = args.index("--lib"); _lib.LIB_PATH = os.path.abspath(args[i + 1]); del args[i:i + 2]
if "--exact" in args:
    args.remove("--exact"); exact = True
settings = args or [""]
nq, H = 6980, 2048
g = torch.Generator(device="cuda").manual_seed(1)
D = torch.empty((N, H), dtype=torch.float32, device="cuda")
for r0 in range(0, N, 1 << 20):
    D[r0:r0 + (1 << 20)].normal_(0.0, 0.5 / H ** 0.5, generator=g)
Q = torch.empty((nq, H), dtype=torch.float32, device="cuda").normal_(0.0, 0.5 / H ** 0.5, generator=g)
idx = DenseIndexHIP(H)
idx.add_device_rows(D)
lib = _lib.load()
ref = None
if exact:
    idx.set_precision("fp32")
    ref = idx.search(Q, 1000)
idx.set_precision("fp32_filtered")
for rnd in range(2):                       # two rounds over the settings: the second one on a warm chip
    for st in settings:
        kv = dict(x.split("=", 1) for x in st.split(",") if x)
        for k_, v_ in kv.items():
            os.environ[k_] = v_
        out = idx.search(Q, 1000)
        torch.cuda.synchronize()
        _lib.check(lib.sr_dense_index_profile(idx._h, 1))
        t = time.perf_counter()
        for _ in range(3):
            out = idx.search(Q, 1000)
        torch.cuda.synchronize()
        t = (time.perf_counter() - t) / 3
        _lib.check(lib.sr_dense_index_profile(idx._h, 0))
        n_l, ms, fl, by = ctypes.c_int64(), ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        _lib.check(lib.sr_dense_index_profile_read(idx._h, ctypes.byref(n_l), ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(by)))
        if ref is None:
            ref = out
        same = bool(torch.equal(out[0], ref[0]) and torch.equal(out[1], ref[1]))
        print(f"[{rnd}] {st or 'default':40s} search {t * 1e3:7.1f} ms   kernel {ms.value / max(n_l.value, 1):.4f} ms x {n_l.value // 3} launches"
              f" = {ms.value / 3:.1f} ms   {fl.value / ms.value / 1e9:7.1f} TF   same_as_first={same} stats={idx.filter_stats()}", flush=True)
        for k_ in kv:
            os.environ.pop(k_, None)
