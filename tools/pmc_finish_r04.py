#!/usr/bin/env python3
"""Adds to gpurun_out/r04_prof/pmc_traffic.json and pmc_sparse_traffic.json what bench.py checks before it quotes them as
`roofline.traffic`: the problem shape the passes ran on, the commit, and the sha256 of the kernel source (a later change of the
kernel makes the figure stale and bench.py then reports traffic null)."""
import hashlib
import json
import os
import subprocess
import sys

O = sys.argv[1]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def sha(rel):
    with open(os.path.join(ROOT, rel), "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()


try:
    commit = subprocess.run(["git", "-C", ROOT, "log", "-1", "--format=%h %s"], capture_output=True, text=True).stdout.strip()
except Exception:
    commit = ""
p = os.path.join(O, "pmc_traffic.json")
if os.path.exists(p):
    d = json.load(open(p))
    line = {}
    try:
        line = json.loads(open(os.path.join(O, "bench_under_rocprof.json")).read().strip().splitlines()[-1])
    except Exception:
        pass
    cfg = line.get("config", {})
    r = line.get("roofline", {})
    steps = max(1, int(line.get("steps", 1)))
    d.update({"note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over the reduced bench command of "
                      "tools/profile_r04.sh, folded by tools/pmc_traffic.py; KB per dispatch averaged over the kernel's dispatches. "
                      "traffic_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (MI355X_MICROARCH.md: gfx950 FETCH_SIZE reports half the bytes of a wide "
                      "coalesced read and counts Infinity-Cache hits).",
              "commit": commit or "(snapshot without .git: see the round's final commit)",
              "dense_split_launch": {"n_docs": cfg.get("n_docs"), "nq": cfg.get("n_queries"), "dim": cfg.get("hidden"),
                                     "launches_per_search": round(r.get("launches", 0) / steps) if r else None,
                                     "algorithmic_bytes": int(r["flop_per_launch"] / (2.0 * cfg["n_queries"] * cfg["hidden"]) * (cfg["hidden"] * 2 + 8) + cfg["n_queries"] * cfg["hidden"] * 2) if r and cfg else None,
                                     "algorithmic_note": "one fp16 plane of the launch's docs + their (x, y) + one fp16 plane of the queries, each once"},
              "kernel_source": {"file": "scaling_retriever_amd/csrc/dense_split.hip", "sha256": sha("scaling_retriever_amd/csrc/dense_split.hip")}})
    json.dump(d, open(p, "w"), indent=1)
p = os.path.join(O, "pmc_sparse_traffic.json")
if os.path.exists(p):
    d = json.load(open(p))
    d.update({"note": "separate --pmc FETCH_SIZE / --pmc WRITE_SIZE passes over `python3 tools/bench_sparse.py --no-cpu --check 0 --steps 1` (full MSMARCO "
                      "shape: V 128 256, N 8 841 823, L0_d 128, 6 980 queries of L0_q 32); per DISPATCH averages - a pass of 6 980 queries is "
                      "`dispatches_per_pass` dispatches of sparse_block_kernel",
              "commit": commit or "(snapshot without .git)",
              "shape": {"V": 128256, "N": 8841823, "L0_d": 128, "L0_q": 32, "nq": 6980},
              "passes_profiled": 2,
              "kernel_source": {"file": "scaling_retriever_amd/csrc/sparse_score.hip", "sha256": sha("scaling_retriever_amd/csrc/sparse_score.hip")}})
    json.dump(d, open(p, "w"), indent=1)
print("ok")
