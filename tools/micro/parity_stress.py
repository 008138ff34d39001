#!/usr/bin/env python3
"""Stress the headline's parity check: bench.py's setup (HIP query encoder + the full corpus on one index), then rounds of
encode -> exact search -> filtered search on the SAME index, compared bit for bit.  python3 tools/micro/parity_stress.py [rounds]"""
import os
import sys

import torch
import subprocess
print('host', os.uname().nodename, '|', subprocess.run('rocm-smi --showserial --showuniqueid 2>/dev/null | grep -i "serial\\|unique" | head -4', shell=True, capture_output=True, text=True).stdout.replace(chr(10), ' ; '), flush=True)

os.environ["SR_DEV_SWITCHES"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench  # noqa: E402
from scaling_retriever_amd.modeling.llm_encoder import LlamaBiDense  # noqa: E402
from scaling_retriever_amd.scoring import DenseIndexHIP  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda", 0)
cfg = dict(bench.LION_1B)
model = LlamaBiDense.from_weights(cfg, bench.random_weights(cfg, dev, 0), max_batch_tokens=65536, max_batch_seqs=8192).to(dev).eval()
batches, lens = bench.synth_batches(6980, 6980, 2.1, 0.35, 4, 64, cfg["vocab_size"], 2, dev)
N, H, k = 8_841_823, 2048, 1000
D = torch.empty((N, H), dtype=torch.float32, device=dev)
g = torch.Generator(device=dev).manual_seed(1)
for r0 in range(0, N, 1 << 20):
    D[r0:r0 + (1 << 20)].normal_(0.0, 0.5 / H ** 0.5, generator=g)
index = DenseIndexHIP(H, device=dev)
index.add_device_rows(D)
index.set_precision("fp32_filtered")
envs = [{}] if len(sys.argv) < 3 else [dict(kv.split("=") for kv in a.split(",")) for a in sys.argv[2:]]
ref = None
for env in envs:
    for kk in ("SR_SPLIT_SEG", "SR_SPLIT_PERSIST"):
        os.environ.pop(kk, None)
    os.environ.update(env)
    n_bad = 0
    for rnd in range(rounds):
        with torch.no_grad():
            reps = torch.cat([model.query_encode(input_ids=i, attention_mask=m) for i, m in batches])
        if ref is None:
            reps0 = reps.clone()
        reps_ok_a = bool(torch.equal(reps, reps0))
        for _ in range(2):
            index.search(reps, k)                      # the timed steps of bench.py
        index.set_precision("fp32")
        es, ei = index.search(reps, k)
        es_b, ei_b = index.search(reps, k)
        index.set_precision("fp32_filtered")
        fs, fi = index.search(reps, k)
        reps_ok_b = bool(torch.equal(reps, reps0))
        if ref is None:
            ref = (es.clone(), ei.clone())
        ex_ok = bool(torch.equal(es, ref[0]) and torch.equal(ei, ref[1]))
        if not (reps_ok_a and reps_ok_b and torch.equal(es, es_b) and torch.equal(ei, ei_b)):
            print(env, "round", rnd, "reps == first reps after encode:", reps_ok_a, "after the searches:", reps_ok_b, "| exact twice equal:",
                  bool(torch.equal(es, es_b) and torch.equal(ei, ei_b)), "| filtered == first exact:", bool(torch.equal(fs, ref[0]) and torch.equal(fi, ref[1])),
                  "| rows of reps that differ:", (reps != reps0).any(1).nonzero()[:, 0].tolist()[:12], flush=True)
        same = bool(torch.equal(es, fs) and torch.equal(ei, fi))
        if not (same and ex_ok):
            n_bad += 1
            bad = (~((fs == es).all(1) & (fi == ei).all(1))).nonzero()[:, 0]
            print(env, "round", rnd, "exact == first exact:", ex_ok, "| filtered == exact:", same, "|", bad.numel(), "queries differ:", bad.tolist()[:12], flush=True)
            for q in bad.tolist()[:3]:
                d = ((fs[q] != es[q]) | (fi[q] != ei[q])).nonzero()[:, 0]
                j = int(d[0])
                print("   q", q, "first diff at rank", j, "of", d.numel(), "filtered", float(fs[q, j]), int(fi[q, j]), "exact", float(es[q, j]), int(ei[q, j]),
                      "| exact id in filtered list:", bool((fi[q] == ei[q, j]).any()), "| filtered id in exact list:", bool((ei[q] == fi[q, j]).any()), flush=True)
            fs2, fi2 = index.search(reps, k)
            print("   filtered again == exact:", bool(torch.equal(es, fs2) and torch.equal(ei, fi2)), flush=True)
    print(env, "rounds", rounds, "bad", n_bad, "stats", index.filter_stats(), index.filter_query_stats(), flush=True)
