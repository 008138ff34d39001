"""Ad-hoc timing of the scoring kernels on one MI355X (development aid, not the contract bench)."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from scaling_retriever_amd.scoring import DenseIndexHIP, SparseIndexHIP

def timeit(fn, iters=3):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return float(np.median(ts))

def dense(N, H, nqs, k, precision="fp32"):
    g = torch.Generator(device="cuda").manual_seed(1)
    D = torch.randn((N, H), device="cuda", generator=g) * (0.5 / H ** 0.5)
    idx = DenseIndexHIP(H); idx.add_device_rows(D); idx.set_precision(precision)
    for nq in nqs:
        Q = torch.randn((nq, H), device="cuda", generator=g)
        ms = timeit(lambda: idx.search(Q, k))
        flops = 2.0 * nq * N * H
        print(json.dumps({"what": "dense", "precision": precision, "N": N, "H": H, "nq": nq, "k": k, "ms": ms, "qps": nq / ms * 1e3,
                          "TFLOPs": flops / ms / 1e9, "D_GBps": N * H * 4 / ms / 1e6}), flush=True)

if __name__ == "__main__":
    ap = argparse.ArgumentParser(); ap.add_argument("--N", type=int, default=1_000_000); ap.add_argument("--H", type=int, default=2048)
    ap.add_argument("--nq", type=str, default="1,16,32,64,128,1024,6980"); ap.add_argument("--k", type=int, default=1000)
    ap.add_argument("--precision", type=str, default="fp32")
    a = ap.parse_args()
    dense(a.N, a.H, [int(x) for x in a.nq.split(",")], a.k, a.precision)
