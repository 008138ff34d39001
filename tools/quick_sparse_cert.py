#!/usr/bin/env python3
"""Quick timing of sr_sparse_search at the MSMARCO shape: certified scorer on / off, statistics, bit-exactness of a sample.
  python tools/quick_sparse_cert.py [--N ...] [--L0-d 128] [--L0-q 32] [--nq 6980] [--check 64] [--exact 1]"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
os.environ["SR_DEV_SWITCHES"] = "1"
from synth import build_index, build_queries  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--V", type=int, default=128256)
    ap.add_argument("--N", type=int, default=8_841_823)
    ap.add_argument("--L0-d", type=int, default=128)
    ap.add_argument("--L0-q", type=int, default=32)
    ap.add_argument("--nq", type=int, default=6980)
    ap.add_argument("--k", type=int, default=1000)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--check", type=int, default=64)
    ap.add_argument("--exact", type=int, default=1, help="also time the exact kernels")
    ap.add_argument("--alpha", type=float, default=1.0)
    ap.add_argument("--cap-div", type=float, default=1.0)
    a = ap.parse_args()
    from scaling_retriever_amd import _lib
    from scaling_retriever_amd.scoring import SparseIndexHIP
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    indptr, doc_ids, vals, df = build_index(a.V, a.N, a.L0_d, dev, 3, alpha=a.alpha, cap=int(a.N / a.cap_div))
    q_indptr, q_cols, q_vals = build_queries(a.V, a.nq, a.L0_q, dev, 4, alpha=a.alpha)
    torch.cuda.synchronize()
    t0 = time.time()
    idx = SparseIndexHIP(indptr, doc_ids, vals, a.N)
    torch.cuda.synchronize()
    out = {"postings": int(doc_ids.numel()), "index_create_s": round(time.time() - t0, 2), "cert": idx.cert_stats()}

    def timed(label):
        idx.search(q_indptr, q_cols, q_vals, a.k)
        torch.cuda.synchronize()
        _lib.check(lib.sr_sparse_index_profile(idx._h, 1))
        ts = time.perf_counter()
        for _ in range(a.steps):
            r = idx.search(q_indptr, q_cols, q_vals, a.k)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - ts) / a.steps
        n_l, ms, by = ctypes.c_int64(0), ctypes.c_double(0), ctypes.c_double(0)
        _lib.check(lib.sr_sparse_index_profile_read(idx._h, ctypes.byref(n_l), ctypes.byref(ms), ctypes.byref(by)))
        _lib.check(lib.sr_sparse_index_profile(idx._h, 0))
        out[label] = {"qps": round(a.nq / dt, 1), "ms_per_pass": round(dt * 1e3, 2), "score_kernel_ms_per_pass": round(ms.value / a.steps, 2),
                      "launches_per_pass": n_l.value // a.steps}
        return r

    s1, i1, c1 = timed("certified")
    out["cert_after"] = idx.cert_stats()
    if a.exact:
        os.environ["SR_SPARSE_CERT_SEARCH"] = "0"
        s0, i0, c0 = timed("exact")
        os.environ.pop("SR_SPARSE_CERT_SEARCH")
        out["same_bits_as_exact_kernels"] = bool(torch.equal(s1, s0) and torch.equal(i1, i0) and torch.equal(c1, c0))
    if a.check:
        from oracle import scoring as SC
        n = a.check
        hq = q_indptr[:n + 1].cpu().numpy()
        e = int(hq[-1])
        oi, os_, oc = SC.sparse_retrieve_c(indptr.cpu().numpy(), doc_ids.cpu().numpy(), vals.cpu().numpy(), hq, q_cols[:e].cpu().numpy(),
                                           q_vals[:e].cpu().numpy(), a.k, 0.0, a.N, q_threads=min(n, os.cpu_count()))
        out["oracle_bit_exact"] = bool(np.array_equal(i1[:n].cpu().numpy(), oi) and np.array_equal(s1[:n].cpu().numpy(), os_)
                                       and np.array_equal(c1[:n].cpu().numpy(), oc))
        out["oracle_queries"] = n
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
