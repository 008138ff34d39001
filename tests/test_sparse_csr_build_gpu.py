"""GPU: sr_sparse_csr_build (csrc/sparse_build.hip, this library's stable radix sort) against numpy's stable sort - the posting
lists IndexDictOfArray.add_batch_document builds by per-posting append (scaling_retriever/utils/inverted_index.py:67-76) - bit for bit,
and the device exclusive scan underneath it."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _coo(rng, n_docs, V, L0, world=1, rank=0, zipf=True):
    """Doc-major postings as SparseIndexer.index collects them: rows ascending, terms ascending inside a doc."""
    per = rng.poisson(L0, size=n_docs).clip(0, V)
    w = 1.0 / np.arange(1, V + 1) if zipf else np.ones(V)
    w = w / w.sum()
    rows, cols = [], []
    for d in range(n_docs):
        c = np.sort(rng.choice(V, size=per[d], replace=False, p=w))
        cols.append(c)
        rows.append(np.full(len(c), d * world + rank))
    rows = np.concatenate(rows).astype(np.int32)
    cols = np.concatenate(cols).astype(np.int32)
    vals = np.log1p(rng.uniform(0, 20, size=len(rows))).astype(np.float32)
    return rows, cols, vals


def _reference(rows, cols, vals, V, sort_docs):
    if sort_docs:
        o = np.lexsort((rows, cols))
    else:
        o = np.argsort(cols, kind="stable")
    indptr = np.concatenate([[0], np.cumsum(np.bincount(cols, minlength=V))]).astype(np.int64)
    return indptr, rows[o], vals[o]


@pytest.mark.parametrize("n_docs,V,L0,sort_docs", [
    (3000, 500, 20, False),          # one radix pass over the terms (9 bits)
    (20000, 128256, 40, False),      # the model's vocabulary: two 9-bit passes, most terms empty
    (9000, 2000, 30, True),          # doc passes first
    (1, 7, 3, False),
])
@pytest.mark.parametrize("kernel", ["tile", "wave"])
def test_csr_build_matches_a_stable_sort(n_docs, V, L0, sort_docs, kernel, monkeypatch):
    """kernel: the workgroup-per-tile scatter (the product path) and the per-wave scatter it replaced in round 6 (dev switch)."""
    from scaling_retriever_amd.scoring import sparse_csr_build
    if kernel == "wave":
        monkeypatch.setenv("SR_DEV_SWITCHES", "1")
        monkeypatch.setenv("SR_SPARSE_BUILD_TILE", "0")
    rng = np.random.default_rng(n_docs + V)
    rows, cols, vals = _coo(rng, n_docs, V, L0)
    if sort_docs:                    # a merged multi-rank index arrives rank-major inside a term: shuffle the insertion order
        p = rng.permutation(len(rows))
        rows, cols, vals = rows[p], cols[p], vals[p]
    dev = torch.device("cuda")
    indptr, out_rows, out_vals = sparse_csr_build(torch.from_numpy(rows).to(dev), torch.from_numpy(cols).to(dev), torch.from_numpy(vals).to(dev),
                                                  V, n_docs=n_docs, sort_docs=sort_docs)
    e_ip, e_rows, e_vals = _reference(rows, cols, vals, V, sort_docs)
    assert np.array_equal(indptr.cpu().numpy(), e_ip)
    assert np.array_equal(out_rows.cpu().numpy(), e_rows)
    assert np.array_equal(out_vals.cpu().numpy(), e_vals)


def test_csr_build_keeps_insertion_order_inside_a_term_and_handles_the_edges():
    from scaling_retriever_amd.scoring import sparse_csr_build
    dev = torch.device("cuda")
    # insertion order that is NOT doc order (two ranks' batches interleaved): stays as inserted without sort_docs
    rows = torch.tensor([5, 1, 9, 1, 5, 0], dtype=torch.int32, device=dev)
    cols = torch.tensor([2, 2, 2, 0, 0, 2], dtype=torch.int32, device=dev)
    vals = torch.arange(6, dtype=torch.float32, device=dev)
    ip, r, v = sparse_csr_build(rows, cols, vals, 4)
    assert ip.tolist() == [0, 2, 2, 6, 6] and r.tolist() == [1, 5, 5, 1, 9, 0] and v.tolist() == [3.0, 4.0, 0.0, 1.0, 2.0, 5.0]
    ip, r, v = sparse_csr_build(rows, cols, vals, 4, n_docs=10, sort_docs=True)
    assert r.tolist() == [1, 5, 0, 1, 5, 9] and v.tolist() == [3.0, 4.0, 5.0, 1.0, 0.0, 2.0]
    # no postings at all
    e = torch.zeros(0, dtype=torch.int32, device=dev)
    ip, r, v = sparse_csr_build(e, e, torch.zeros(0, device=dev), 5)
    assert ip.tolist() == [0] * 6 and r.numel() == 0
    # a term outside the vocabulary is an error, not a wild write
    with pytest.raises(ValueError):
        sparse_csr_build(rows, torch.tensor([2, 2, 4, 0, 0, 2], dtype=torch.int32, device=dev), vals, 4)
    # sort_docs orders by the digits of [0, n_docs): a row beyond a stale n_docs would leave the lists not ascending - refused
    with pytest.raises(ValueError, match="n_docs"):
        sparse_csr_build(rows, cols, vals, 4, n_docs=9, sort_docs=True)
    ip, r, v = sparse_csr_build(rows, cols, vals, 4, n_docs=9)              # without sort_docs the rows are payload: any value >= 0
    assert r.tolist() == [1, 5, 5, 1, 9, 0]


def test_csr_build_of_views_that_start_anywhere():
    """The histogram reads whole tiles 16 bytes per lane when the arrays allow it: views that start 4, 8, 12 bytes into an allocation
    (and a last tile that is not full) take the other path and give the same lists."""
    from scaling_retriever_amd.scoring import sparse_csr_build
    rng = np.random.default_rng(11)
    rows, cols, vals = _coo(rng, 6000, 3000, 25)
    dev = torch.device("cuda")
    e_ip, e_rows, e_vals = _reference(rows, cols, vals, 3000, False)
    for off in (0, 1, 2, 3):
        pad = np.zeros(off, dtype=np.int32)
        r = torch.from_numpy(np.concatenate([pad, rows])).to(dev)[off:]
        c = torch.from_numpy(np.concatenate([pad, cols])).to(dev)[off:]
        v = torch.from_numpy(np.concatenate([pad.astype(np.float32), vals])).to(dev)[off:]
        assert r.is_contiguous() and r.data_ptr() % 16 == (4 * off) % 16
        ip, out_r, out_v = sparse_csr_build(r, c, v, 3000)
        assert np.array_equal(ip.cpu().numpy(), e_ip) and np.array_equal(out_r.cpu().numpy(), e_rows) and np.array_equal(out_v.cpu().numpy(), e_vals)


@pytest.mark.parametrize("seed", range(6))
def test_csr_build_random_shapes_around_the_tile_and_digit_boundaries(seed):
    """Random vocabularies (1 … 21 bits: one to three passes, digits of 1 … 9 bits), posting counts at and around multiples of the
    8 192-posting tile, skewed and flat term laws, with and without the doc passes: always the stable sort of the triples."""
    from scaling_retriever_amd.scoring import sparse_csr_build
    rng = np.random.default_rng(100 + seed)
    dev = torch.device("cuda")
    for case in range(6):
        bits = int(rng.integers(1, 22))
        V = int(rng.integers(max(2, (1 << bits) // 2 + 1), (1 << bits) + 1)) if bits > 1 else 2
        nnz = int(rng.choice([1, 63, 64, 65, 1023, 8191, 8192, 8193, 3 * 8192, 5 * 8192 + 17, 70001]))
        n_docs = int(rng.integers(1, 5000))
        if rng.random() < 0.5:
            cols = (rng.random(nnz) ** 4 * V).astype(np.int64).clip(0, V - 1)            # skewed towards the small ids
        else:
            cols = rng.integers(0, V, size=nnz)
        rows = rng.integers(0, n_docs, size=nnz)
        cols, rows = cols.astype(np.int32), rows.astype(np.int32)
        vals = rng.random(nnz).astype(np.float32)
        for sort_docs in (False, True):
            ip, r, v = sparse_csr_build(torch.from_numpy(rows).to(dev), torch.from_numpy(cols).to(dev), torch.from_numpy(vals).to(dev), V,
                                        n_docs=n_docs, sort_docs=sort_docs)
            # (equal (term, doc) pairs keep their input order under sort_docs too: lexsort is stable)
            e_ip, e_rows, e_vals = _reference(rows, cols, vals, V, sort_docs)
            tag = (seed, case, bits, V, nnz, sort_docs)
            assert np.array_equal(ip.cpu().numpy(), e_ip), tag
            assert np.array_equal(r.cpu().numpy(), e_rows), tag
            assert np.array_equal(v.cpu().numpy(), e_vals), tag


def test_csr_build_large_and_the_index_it_feeds():
    """1.3 M docs x 24 postings at V = 128 256 (31 M postings, two 9-bit passes, several thousand waves each): equals the stable sort, and the index built from it
    scores like the oracle."""
    from oracle import scoring as O
    from scaling_retriever_amd.scoring import SparseIndexHIP, sparse_csr_build
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(5)
    n_docs, V, L = 1_300_000, 128256, 24
    # L distinct terms per doc: a random start + distinct offsets from a Zipf-ish table, ascending inside the doc
    w = 1.0 / torch.arange(1, V + 1, device=dev, dtype=torch.float32)
    cols = torch.multinomial(w.expand(4096, V), L, replacement=False, generator=g)
    cols = cols[torch.randint(0, 4096, (n_docs,), device=dev, generator=g)]
    cols = ((cols + torch.randint(0, 50, (n_docs, 1), device=dev, generator=g)) % V).sort(dim=1).values
    keep = torch.ones_like(cols, dtype=torch.bool)
    keep[:, 1:] = cols[:, 1:] != cols[:, :-1]                      # distinct terms inside a doc
    rows = torch.arange(n_docs, device=dev, dtype=torch.int32)[:, None].expand(n_docs, L)[keep].contiguous()
    cols = cols[keep].to(torch.int32).contiguous()
    vals = torch.log1p(torch.rand(cols.numel(), device=dev, generator=g) * 20)
    indptr, out_rows, out_vals = sparse_csr_build(rows, cols, vals, V)
    o = torch.sort(cols, stable=True).indices
    assert torch.equal(out_rows, rows[o]) and torch.equal(out_vals, vals[o])
    counts = torch.bincount(cols.long(), minlength=V)
    assert torch.equal(indptr[1:] - indptr[:-1], counts) and int(indptr[0]) == 0
    idx = SparseIndexHIP(indptr, out_rows, out_vals, n_docs)
    rng = np.random.default_rng(0)
    qi = np.arange(0, 16 * 12 + 1, 12, dtype=np.int64)
    qc = np.concatenate([np.sort(rng.choice(V, size=12, replace=False)) for _ in range(16)]).astype(np.int32)
    qv = np.log1p(rng.uniform(0, 20, size=len(qc))).astype(np.float32)
    s, i, c = idx.search(qi, qc, qv, 100)
    ei, es, ec = O.sparse_retrieve_c(indptr.cpu().numpy(), out_rows.cpu().numpy(), out_vals.cpu().numpy(), qi, qc, qv, 100, 0.0, n_docs, q_threads=4)
    assert np.array_equal(i.cpu().numpy(), ei) and np.array_equal(s.cpu().numpy(), es) and np.array_equal(c.cpu().numpy(), ec)
