#!/usr/bin/env python3
"""sr_sparse_csr_build with the tile scatter (default) against the per-wave scatter (SR_SPARSE_BUILD_TILE=0) on the same doc-major
postings: both results compared bit for bit with each other and with torch's stable sort, then timed.
python3 tools/micro/csr_build_ab.py [million_docs] [L0_d] [only]      ("only": the tile scatter's term build alone, for rocprofv3)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from scaling_retriever_amd.scoring import sparse_csr_build  # noqa: E402

os.environ["SR_DEV_SWITCHES"] = "1"
dev = torch.device("cuda", 0)
n_docs = int(float(sys.argv[1]) * 1e6) if len(sys.argv) > 1 else 2_000_000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 127
V = 128256
g = torch.Generator(device=dev).manual_seed(1)
w = 1.0 / torch.arange(1, V + 1, device=dev, dtype=torch.float32)
tab = torch.multinomial(w.expand(4096, V), L, replacement=False, generator=g)
cols = tab[torch.randint(0, 4096, (n_docs,), device=dev, generator=g)]
cols = ((cols + torch.randint(0, 50, (n_docs, 1), device=dev, generator=g)) % V).sort(dim=1).values
keep = torch.ones_like(cols, dtype=torch.bool)
keep[:, 1:] = cols[:, 1:] != cols[:, :-1]
rows = torch.arange(n_docs, device=dev, dtype=torch.int32)[:, None].expand(n_docs, L)[keep].contiguous()
cols = cols[keep].to(torch.int32).contiguous()
del keep, tab
vals = torch.log1p(torch.rand(cols.numel(), device=dev, generator=g) * 20)
nnz = cols.numel()
res = {}
only = len(sys.argv) > 3 and sys.argv[3] == "only"
for mode in ("1",) if only else ("1", "0"):
    os.environ["SR_SPARSE_BUILD_TILE"] = mode
    sparse_csr_build(rows[:1 << 20], cols[:1 << 20], vals[:1 << 20], V)
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        out = sparse_csr_build(rows, cols, vals, V)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    res[mode] = out
    t = min(ts)
    print(f"tile={mode}: {t * 1e3:.2f} ms for {nnz} postings = {nnz / t / 1e9:.2f} G postings/s, {nnz * 56 / t / 1e12:.3f} TB/s of 28 B per posting and pass", flush=True)
if only:
    sys.exit(0)
for a, b in zip(res["1"], res["0"]):
    assert torch.equal(a, b)
o = torch.sort(cols, stable=True).indices
assert torch.equal(res["1"][1], rows[o]) and torch.equal(res["1"][2], vals[o])
# the doc passes too (3 passes over 24 bits), from the term-major result
p = torch.randperm(nnz, device=dev, generator=g)
r2, c2, v2 = rows[p].contiguous(), cols[p].contiguous(), vals[p].contiguous()
outs = {}
for mode in ("1", "0"):
    os.environ["SR_SPARSE_BUILD_TILE"] = mode
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    outs[mode] = sparse_csr_build(r2, c2, v2, V, n_docs=n_docs, sort_docs=True)
    torch.cuda.synchronize()
    print(f"sort_docs tile={mode}: {(time.perf_counter() - t0) * 1e3:.2f} ms", flush=True)
for a, b in zip(outs["1"], outs["0"]):
    assert torch.equal(a, b)
assert torch.equal(outs["1"][1], res["1"][1]) and torch.equal(outs["1"][2], res["1"][2])
print("BIT_IDENTICAL", flush=True)
