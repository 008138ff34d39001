#!/usr/bin/env python3
"""Time every build_var/libsr_*.so (tools/micro/gemm_variants.sh) on the Lion-1B layer GEMM shapes: one subprocess per library.
python3 tools/micro/gemm_variant_bench.py [M,M,...]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
WORKER = r'''
import os, sys, json
os.environ["SR_DEV_SWITCHES"] = "1"
sys.path.insert(0, %r)
import torch
from scaling_retriever_amd import _lib as L
lib = L.load()
Ms = [int(x) for x in sys.argv[1].split(",")]
out = {}
g = torch.Generator(device="cuda").manual_seed(0)
for M in Ms:
    tot_ms = tot_fl = 0.0
    row = {}
    for name, N, K in (("qkv", 3072, 2048), ("o", 2048, 2048), ("gate_up", 16384, 2048), ("down", 2048, 8192)):
        A = torch.randn((M, K), device="cuda", generator=g).bfloat16()
        W = (torch.randn((N, K), device="cuda", generator=g) * 0.02).bfloat16()
        C = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
        f = lambda: L.check(lib.sr_gemm_bf16(A.data_ptr(), W.data_ptr(), M, N, K, 0, C.data_ptr(), None, L.stream_ptr()))
        for _ in range(5): f()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(20): f()
            b.record(); torch.cuda.synchronize()
            best = min(best, a.elapsed_time(b) / 20)
        row[name] = round(2.0 * M * N * K / best / 1e9, 1)
        tot_ms += best; tot_fl += 2.0 * M * N * K
    row["layer"] = round(tot_fl / tot_ms / 1e9, 1)
    out[M] = row
print(json.dumps(out))
''' % ROOT
Ms = sys.argv[1] if len(sys.argv) > 1 else "16384,38400"
libs = sorted(f for f in os.listdir(os.path.join(ROOT, "build_var")) if f.endswith(".so"))
for rnd in range(2):
    for f in libs + ["(product)"]:
        env = dict(os.environ)
        if f != "(product)":
            env["SR_HIP_LIB"] = os.path.join(ROOT, "build_var", f)
        r = subprocess.run([sys.executable, "-c", WORKER, Ms], capture_output=True, text=True, env=env)
        print(f, r.stdout.strip() or r.stderr[-500:], flush=True)
