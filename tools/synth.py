"""Synthetic workloads of BASELINE.json configs[2] (the reference publishes no L0 statistics):
V = 128 256 terms, N = 8 841 823 docs, mean L0_d postings per doc, document frequencies Zipf(1.0) (df_r ~ 1/r, capped
at N); queries: L0_q distinct terms drawn from the same Zipf, values log1p(U(0, 20)).  Used by bench.py, tools/bench_sparse.py
and tests/test_full_size_gpu.py."""
import numpy as np
import torch


def zipf_df(V, N, total):
    """df_r = min(N, C / r) with sum = total (water-filling on the cap)."""
    r = np.arange(1, V + 1, dtype=np.float64)
    lo, hi = 0.0, float(total) * V
    for _ in range(100):
        C = 0.5 * (lo + hi)
        s = np.minimum(N, C / r).sum()
        lo, hi = (C, hi) if s < total else (lo, C)
    return np.maximum(1, np.floor(np.minimum(N, C / r))).astype(np.int64)


def build_index(V, N, L0_d, device, seed):
    g = torch.Generator(device=device).manual_seed(seed)
    df = zipf_df(V, N, N * L0_d)
    heavy = int((df > N // 8).sum())          # Bernoulli masks for the heaviest lists, sampling for the rest
    ids_parts, counts = [], np.zeros(V, dtype=np.int64)
    for t in range(heavy):
        m = torch.rand(N, device=device, generator=g) < (df[t] / N)
        d = torch.nonzero(m)[:, 0].to(torch.int32)
        ids_parts.append(d)
        counts[t] = d.numel()
    light_df = torch.from_numpy(df[heavy:]).to(device)
    term = torch.repeat_interleave(torch.arange(heavy, V, device=device), light_df)
    doc = torch.randint(0, N, (int(light_df.sum().item()),), device=device, generator=g)
    key = torch.unique(term * N + doc)        # sorted by (term, doc), duplicates dropped
    del term, doc
    t_of = torch.div(key, N, rounding_mode="floor")
    ids_parts.append((key - t_of * N).to(torch.int32))
    counts[heavy:] = torch.bincount(t_of - heavy, minlength=V - heavy).cpu().numpy()
    del key, t_of
    doc_ids = torch.cat(ids_parts)
    del ids_parts
    vals = torch.log1p(torch.rand(doc_ids.numel(), device=device, generator=g) * 20.0)
    indptr = torch.from_numpy(np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)).to(device)
    return indptr, doc_ids, vals, df


def build_queries(V, nq, L0_q, device, seed):
    g = torch.Generator(device=device).manual_seed(seed)
    w = 1.0 / torch.arange(1, V + 1, device=device, dtype=torch.float32)
    cols = torch.multinomial(w.expand(nq, V), L0_q, replacement=False, generator=g)
    cols = torch.sort(cols, dim=1).values.to(torch.int32).reshape(-1).contiguous()
    vals = torch.log1p(torch.rand(nq * L0_q, device=device, generator=g) * 20.0)
    indptr = torch.arange(0, nq * L0_q + 1, L0_q, device=device, dtype=torch.int64)
    return indptr, cols, vals
