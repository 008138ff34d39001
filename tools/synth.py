"""Synthetic workloads of BASELINE.json configs[2] (the reference publishes no L0 statistics):
V = 128 256 terms, N = 8 841 823 docs, mean L0_d postings per doc, document frequencies Zipf(1.0) (df_r ~ 1/r, capped
at N); queries: L0_q distinct terms drawn from the same Zipf, values log1p(U(0, 20)).  Used by bench.py, tools/bench_sparse.py
and tests/test_full_size_gpu.py."""
import numpy as np
import torch


def zipf_df(V, N, total, alpha=1.0, cap=None):
    """df_r = min(cap, C / r^alpha) with sum = total (water-filling on the cap; cap = N unless given)."""
    cap = N if cap is None else cap
    r = np.arange(1, V + 1, dtype=np.float64) ** alpha
    lo, hi = 0.0, float(total) * V ** max(1.0, alpha)
    for _ in range(200):
        C = 0.5 * (lo + hi)
        s = np.minimum(cap, C / r).sum()
        lo, hi = (C, hi) if s < total else (lo, C)
    return np.maximum(1, np.floor(np.minimum(cap, C / r))).astype(np.int64)


def build_index(V, N, L0_d, device, seed, alpha=1.0, cap=None):
    """alpha / cap: document-frequency law df_r ~ r^-alpha capped at `cap` docs (default: Zipf(1.0) capped at N, the headline's index;
    alpha = 0.7 with cap = N / 5 is the `flat` workload of bench.py's sparse_sweep: no term reaches a quarter of the collection)."""
    g = torch.Generator(device=device).manual_seed(seed)
    df = zipf_df(V, N, N * L0_d, alpha, cap)
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


def build_queries(V, nq, L0_q, device, seed, alpha=1.0):
    g = torch.Generator(device=device).manual_seed(seed)
    w = 1.0 / torch.arange(1, V + 1, device=device, dtype=torch.float32) ** alpha
    cols = torch.multinomial(w.expand(nq, V), L0_q, replacement=False, generator=g)
    cols = torch.sort(cols, dim=1).values.to(torch.int32).reshape(-1).contiguous()
    vals = torch.log1p(torch.rand(nq * L0_q, device=device, generator=g) * 20.0)
    indptr = torch.arange(0, nq * L0_q + 1, L0_q, device=device, dtype=torch.int64)
    return indptr, cols, vals


# ---- dense corpora for the certified filter's robustness legs (bench.py `filter_robustness`, tests/test_filter_corpora_gpu.py) ----
# The headline score stage runs on isotropic Gaussian rows of near-constant norm.  Real LlamaBiDense
# embeddings are means of unit vectors: a large common component, a low-rank structure, norms spread over 0.3-0.9, and MS MARCO
# holds near-duplicate passages.  These generators reproduce those traits at any size, on the device, chunk by chunk.
def _dense_basis(H, device, seed):
    g = torch.Generator(device=device).manual_seed(seed)
    m = torch.randn(H, device=device, generator=g)
    m /= m.norm()
    W = torch.linalg.qr(torch.randn(H, 64, device=device, generator=g))[0]        # [H, 64], orthonormal columns
    return m, W


def dense_rows(kind, n, H, device, seed, basis_seed=77, norm_lo=0.3, norm_hi=0.9):
    """kind: "gauss" (N(0, 0.5 / sqrt(H)), the headline's corpus) | "aniso" (0.6 common mean + 0.6 rank-64 + 0.53 noise, unit
    direction, lognormal norms clipped to [norm_lo, norm_hi]) | "aniso_dup" (the same with 2 % of the rows rewritten as
    near-duplicate clusters of 50-5000 members, relative distance 1e-4 or 3e-2 to their centre)."""
    g = torch.Generator(device=device).manual_seed(seed)
    out = torch.empty((n, H), dtype=torch.float32, device=device)
    if kind == "gauss":
        for r0 in range(0, n, 1 << 20):
            out[r0:r0 + (1 << 20)].normal_(0.0, 0.5 / H ** 0.5, generator=g)
        return out
    m, W = _dense_basis(H, device, basis_seed)
    for r0 in range(0, n, 1 << 19):
        r1 = min(n, r0 + (1 << 19))
        z = torch.randn((r1 - r0, 64), device=device, generator=g) / 8.0
        x = torch.randn((r1 - r0, H), device=device, generator=g) * (0.53 / H ** 0.5)
        x.addmm_(z, W.T, alpha=0.6)
        x += 0.6 * m
        x /= x.norm(dim=1, keepdim=True)
        norms = torch.exp(torch.randn((r1 - r0, 1), device=device, generator=g) * 0.3 - 0.6).clamp_(norm_lo, norm_hi)
        out[r0:r1] = x * norms
    if kind == "aniso_dup":
        rng = np.random.default_rng(seed + 1)
        left, start = int(0.02 * n), 0
        perm = torch.randperm(n, device=device, generator=g)
        while left >= 50 and start < n:
            size = int(min(left, np.exp(rng.uniform(np.log(50), np.log(5000)))))
            rows = perm[start:start + size]
            centre = out[rows[0]].clone()
            rel = 1e-4 if rng.random() < 0.5 else 3e-2
            noise = torch.randn((rows.numel(), H), device=device, generator=g) * (rel * float(centre.norm()) / H ** 0.5)
            out[rows] = centre[None, :] + noise
            start += size
            left -= size
    elif kind != "aniso":
        raise ValueError(kind)
    return out


def dense_queries(kind, nq, H, device, seed, D=None, basis_seed=77):
    """"gauss" | "aniso" (same basis as the corpus, norms 0.5-0.9) | "near_docs" (every query = a random row of D, rescaled to
    norm 0.8, plus noise at cosine ~0.85: the first hit is far above the rest, the cut sits in the row's neighbourhood)."""
    g = torch.Generator(device=device).manual_seed(seed)
    if kind == "gauss":
        return torch.randn((nq, H), device=device, generator=g) * (0.5 / H ** 0.5)
    if kind == "aniso":
        m, W = _dense_basis(H, device, basis_seed)
        z = torch.randn((nq, 64), device=device, generator=g) / 8.0
        x = torch.randn((nq, H), device=device, generator=g) * (0.53 / H ** 0.5)
        x.addmm_(z, W.T, alpha=0.6)
        x += 0.6 * m
        x /= x.norm(dim=1, keepdim=True)
        return x * (0.5 + 0.4 * torch.rand((nq, 1), device=device, generator=g))
    if kind == "near_docs":
        rows = torch.randint(0, D.shape[0], (nq,), device=device, generator=g)
        x = D[rows].clone()
        x /= x.norm(dim=1, keepdim=True)
        x = 0.85 * x + torch.randn((nq, H), device=device, generator=g) * (0.527 / H ** 0.5)
        return 0.8 * x / x.norm(dim=1, keepdim=True)
    raise ValueError(kind)
