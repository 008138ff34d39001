#!/usr/bin/env python3
"""Does sr_dense_search repeat itself?  Full-size corpus, fixed queries, R searches per mode compared with the first.
python3 tools/micro/exact_stress.py plain:100 alt:50 filtered:100"""
import os
import sys

import torch
import subprocess
print('host', os.uname().nodename, '|', subprocess.run('rocm-smi --showserial --showuniqueid 2>/dev/null | grep -i "serial\\|unique" | head -4', shell=True, capture_output=True, text=True).stdout.replace(chr(10), ' ; '), flush=True)

os.environ["SR_DEV_SWITCHES"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from scaling_retriever_amd.scoring import DenseIndexHIP  # noqa: E402

dev = torch.device("cuda", 0)
N, H, k, nq = 8_841_823, 2048, 1000, 6980
g = torch.Generator(device=dev).manual_seed(1)
D = torch.empty((N, H), dtype=torch.float32, device=dev)
for r0 in range(0, N, 1 << 20):
    D[r0:r0 + (1 << 20)].normal_(0.0, 0.5 / H ** 0.5, generator=g)
Q = torch.empty((nq, H), dtype=torch.float32, device=dev).normal_(0.0, 0.5 / H ** 0.5, generator=g)
index = DenseIndexHIP(H, device=dev)
index.add_device_rows(D)
ref = None


def check(tag, it, s, i):
    global ref
    if ref is None:
        ref = (s.clone(), i.clone())
        return
    if not (torch.equal(s, ref[0]) and torch.equal(i, ref[1])):
        bad = (~((s == ref[0]).all(1) & (i == ref[1]).all(1))).nonzero()[:, 0]
        q = int(bad[0])
        d = ((s[q] != ref[0][q]) | (i[q] != ref[1][q])).nonzero()[:, 0]
        print(tag, "iteration", it, ":", bad.numel(), "queries differ", bad.tolist()[:10], "| q", q, "first diff at rank", int(d[0]), "of", d.numel(),
              "got", float(s[q, d[0]]), int(i[q, d[0]]), "ref", float(ref[0][q, d[0]]), int(ref[1][q, d[0]]), flush=True)


for spec in sys.argv[1:] or ["plain:60"]:
    mode, reps = spec.split(":")
    reps = int(reps)
    for it in range(reps):
        if mode in ("plain", "alt"):
            index.set_precision("fp32")
            check(mode + "/exact", it, *index.search(Q, k))
        if mode in ("filtered", "alt"):
            index.set_precision("fp32_filtered")
            check(mode + "/filtered", it, *index.search(Q, k))
    print(mode, "done:", reps, "iterations", index.filter_stats(), index.filter_query_stats(), flush=True)
