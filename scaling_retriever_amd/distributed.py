"""Doc-sharded retrieval across the GPUs of one node (one process per GPU, RCCL over xGMI).

The reference shards the corpus by document for ENCODING only - DistributedSampler(shuffle=False):
rank r takes dataset rows r, r+W, ... (/root/reference/eval_dense.py:178, eval_sparse.py:96), the
global row of local row i being g_row = i*W + rank (/root/reference/scaling_retriever/indexer.py:262)
- and then scores on ONE process (eval_dense.py:191, eval_sparse.py:114).  Here each GPU also keeps
and scores the documents it encoded; the only exchange is ONE gather of the per-shard top-k to rank 0
(6980 x 1000 x 8 B = 55.8 MB per rank), followed by sr_topk_merge.

Unlike the sampler, shards are NOT padded by wrap-around: no duplicate documents exist, so the merged
result equals the single-GPU result bit for bit (same keys, same tie rule).
"""
import torch
import torch.distributed as dist

# The collectives below are no-ops on a group of one rank.  Tests set this to run them through torch.distributed anyway (a
# 1-rank "nccl" group on one MI355X is the only way RCCL executes on a one-GPU box: tests/test_rccl_one_rank_gpu.py).
FORCE_COLLECTIVES = False


def _single(group):
    return not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not FORCE_COLLECTIVES)


def shard_rows(n_total, rank, world_size):
    """Dataset rows of this rank: rank, rank + W, ... (no wrap-around padding)."""
    return range(rank, n_total, world_size)


def shard_size(n_total, rank, world_size):
    return len(shard_rows(n_total, rank, world_size))


def pack_topk(scores, ids):
    """(fp32 [nq,k], int64 [nq,k]) -> one int64 [nq,k] tensor: id in the high 32 bits (-1 -> 0xffffffff),
    score bits in the low 32, so that the shard result travels in ONE collective."""
    bits = scores.contiguous().view(torch.int32).to(torch.int64) & 0xFFFFFFFF
    return ((ids.to(torch.int64) & 0xFFFFFFFF) << 32) | bits


def unpack_topk(packed):
    lo = packed & 0xFFFFFFFF
    lo = torch.where(lo >= 2 ** 31, lo - 2 ** 32, lo).to(torch.int32)   # back to the original bit pattern
    scores = lo.view(torch.float32)
    ids = (packed >> 32) & 0xFFFFFFFF
    ids = torch.where(ids == 0xFFFFFFFF, torch.full_like(ids, -1), ids)
    return scores, ids


def gather_topk(scores, ids, dst=0, group=None, compact=None):
    """The single data collective of doc-sharded retrieval.  Every rank passes its local (scores, global ids) [nq, k]; rank `dst`
    gets (scores [W, nq, k], ids [W, nq, k]), the others (None, None).

    compact=None (default): over RCCL the padded [nq, k] buffer travels as it is, in ONE gather and without a host sync - 55.8 MB per
    rank at MS MARCO Dev, 0.4 ms on a rank's own xGMI link (153 GB/s; the 7 senders use 7 different links of rank dst) - because what
    compaction saves there (0.3 ms at W = 8) is what its size exchange, its host syncs and the re-padding on rank dst cost; over gloo
    (CPU tests, ranks sharing a GPU: a transport of ~1 GB/s) the valid slots alone travel.  True / False force either.

    compact: ship only the VALID slots.  With the threshold exchange a dense shard returns what can reach the global top-k -
    193 of 1 000 slots per query on average at W = 8 - and the rest of its [nq, k] output is padding (id -1); a sparse shard pads
    rows with fewer than k hits.  A rank's payload is one int64 buffer [nq + cap]: per-row counts, then the packed valid entries
    in row order; cap = the largest payload over the ranks (one 8-byte all-gather sizes it).  Rank dst lays the entries back into
    padded [nq, k] rows (valid entries first), which is what sr_topk_merge takes.  When compaction would save less than a quarter
    of the bytes the full [nq, k] buffer travels instead (no size exchange then is not possible - the ranks must agree - so the
    decision is taken on the all-gathered sizes)."""
    if _single(group):
        return scores.unsqueeze(0), ids.unsqueeze(0)
    W, rank = dist.get_world_size(group), dist.get_rank(group)
    packed = pack_topk(scores, ids)
    device = packed.device
    use_cpu = dist.get_backend(group) == "gloo"       # gloo (CPU tests, or several ranks sharing one GPU) has no CUDA collectives
    nq, k = packed.shape
    if compact is None:
        compact = use_cpu
    if compact:
        valid = ids >= 0
        counts = valid.sum(1).to(torch.int64)
        n_local = counts.sum().reshape(1)
        sizes = [torch.zeros(1, dtype=torch.int64, device="cpu" if use_cpu else device) for _ in range(W)]
        dist.all_gather(sizes, n_local.cpu() if use_cpu else n_local, group=group)
        cap = max(int(t.item()) for t in sizes)
        compact = (nq + cap) * 4 <= nq * k * 3
    if not compact:
        payload = packed.cpu() if use_cpu and packed.is_cuda else packed
        bufs = [torch.empty_like(payload) for _ in range(W)] if rank == dst else None
        dist.gather(payload, gather_list=bufs, dst=dst, group=group)
        if rank != dst:
            return None, None
        s, i = zip(*[unpack_topk(b.to(device)) for b in bufs])
        return torch.stack(s), torch.stack(i)
    buf = torch.zeros(nq + cap, dtype=torch.int64, device=device)
    buf[:nq] = counts
    buf[nq:nq + int(n_local.item())] = packed[valid]
    payload = buf.cpu() if use_cpu and buf.is_cuda else buf
    bufs = [torch.empty_like(payload) for _ in range(W)] if rank == dst else None
    dist.gather(payload, gather_list=bufs, dst=dst, group=group)
    if rank != dst:
        return None, None
    out = torch.full((W, nq, k), -1 << 32, dtype=torch.int64, device=device)      # id -1, score bits 0: padding
    rows_all = torch.arange(nq, device=device)
    for r, b in enumerate(bufs):
        b = b.to(device)
        c = b[:nq]
        n = int(c.sum().item())
        if n == 0:
            continue
        rows = torch.repeat_interleave(rows_all, c)
        cols = torch.arange(n, device=device) - (torch.cumsum(c, 0) - c)[rows]
        out[r, rows, cols] = b[nq:nq + n]
    s, i = unpack_topk(out)
    return s, i


def query_slice(n_queries, rank, world_size):
    """Contiguous block of queries rank `rank` encodes: [lo, hi) of ceil(n/W)-sized blocks (the last ones may be short)."""
    per = (n_queries + world_size - 1) // world_size
    return min(rank * per, n_queries), min((rank + 1) * per, n_queries)


def all_gather_query_reps(local_reps, n_queries, group=None):
    """Second (small) collective of the sharded path: every rank encodes only ITS block of queries
    (query_slice) and the fp32 embeddings are all-gathered (6980 x 2048 x 4 B = 57 MB in total), instead of every
    rank re-encoding all queries.  local_reps: [hi - lo, H]; returns [n_queries, H] on every rank."""
    if _single(group):
        return local_reps
    W = dist.get_world_size(group)
    per = (n_queries + W - 1) // W
    H = local_reps.shape[1]
    padded = torch.zeros((per, H), dtype=local_reps.dtype, device=local_reps.device)
    padded[:local_reps.shape[0]] = local_reps
    out = torch.empty((W * per, H), dtype=local_reps.dtype, device=local_reps.device)
    if dist.get_backend(group) == "gloo" and padded.is_cuda:
        chunks = [torch.empty((per, H), dtype=padded.dtype) for _ in range(W)]
        dist.all_gather(chunks, padded.cpu(), group=group)
        out = torch.cat(chunks).to(local_reps.device)
    else:
        dist.all_gather_into_tensor(out, padded, group=group)
    return out[:n_queries].contiguous()


def all_gather_query_csr(row_ptr, cols, vals, n_queries, group=None):
    """Sparse twin of all_gather_query_reps: every rank encodes only ITS block of queries (query_slice) and the CSR pieces
    (row_ptr int64 [n_local + 1], cols int32, vals fp32; 6980 x 32 x 8 B = 1.8 MB in total at MS MARCO Dev) are
    all-gathered in ONE collective: a rank's piece travels as one int64 buffer [per + cap] - per row counts, then
    (col << 32 | value bits) entries - padded to the largest piece.  Returns the CSR of ALL queries on every rank."""
    if _single(group):
        return row_ptr, cols, vals
    W = dist.get_world_size(group)
    per = (n_queries + W - 1) // W
    device = cols.device
    counts = (row_ptr[1:] - row_ptr[:-1]).to(torch.int64)
    nnz = torch.tensor([int(cols.numel())], dtype=torch.int64, device=device)
    use_cpu = dist.get_backend(group) == "gloo"
    caps = [torch.zeros_like(nnz.cpu() if use_cpu else nnz) for _ in range(W)]
    dist.all_gather(caps, nnz.cpu() if use_cpu else nnz, group=group)          # 8 bytes per rank: sizes the payload
    cap = max(int(c.item()) for c in caps)
    buf = torch.zeros(per + cap, dtype=torch.int64, device=device)
    buf[:counts.numel()] = counts
    bits = vals.contiguous().view(torch.int32).to(torch.int64) & 0xFFFFFFFF
    buf[per:per + cols.numel()] = (cols.to(torch.int64) << 32) | bits
    if use_cpu:
        chunks = [torch.empty(per + cap, dtype=torch.int64) for _ in range(W)]
        dist.all_gather(chunks, buf.cpu(), group=group)
        allb = torch.stack(chunks).to(device)
    else:
        allb = torch.empty((W, per + cap), dtype=torch.int64, device=device)
        dist.all_gather_into_tensor(allb, buf, group=group)
    all_counts, all_cols, all_vals = [], [], []
    for r in range(W):
        lo, hi = query_slice(n_queries, r, W)
        c = allb[r, :hi - lo]
        n = int(c.sum().item())
        e = allb[r, per:per + n]
        lo32 = e & 0xFFFFFFFF
        lo32 = torch.where(lo32 >= 2 ** 31, lo32 - 2 ** 32, lo32).to(torch.int32)
        all_counts.append(c)
        all_cols.append((e >> 32).to(torch.int32))
        all_vals.append(lo32.view(torch.float32))
    counts = torch.cat(all_counts)
    q_ptr = torch.cat([torch.zeros(1, dtype=torch.int64, device=device), torch.cumsum(counts, 0)])
    return q_ptr, torch.cat(all_cols).contiguous(), torch.cat(all_vals).contiguous()


def all_reduce_min(values, group=None):
    """Third (tiny) collective of the doc-sharded dense path: element-wise minimum over the ranks of an fp32 vector (nq floats:
    28 KB at MS MARCO Dev) - the shards' lower bounds of their ceil(k / W)-th best score, whose minimum bounds the global
    k-th score from below (sr_dense_search_begin / _finish)."""
    if _single(group):
        return values
    if dist.get_backend(group) == "gloo" and values.is_cuda:
        t = values.cpu()
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
        return t.to(values.device)
    t = values.clone()
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    return t


def sharded_dense_search(index, queries, k, world_size, group=None):
    """One rank's part of a doc-sharded dense search on its DenseIndexHIP: candidates + lower bound, all-reduce(min), exact
    re-score of what can reach the GLOBAL top-k.  Returns this shard's (scores, ids) [nq, k] (possibly padded)."""
    queries = queries.contiguous()
    if (world_size <= 1 and not FORCE_COLLECTIVES) or _single(group):   # no process group: the shard's own bound is NOT a global one
        return index.search(queries, k)
    world_size = max(1, world_size)             # FORCE_COLLECTIVES on a world of one (tests): share = 1, the bound is the k-th score
    lower = index.search_begin(queries, k, world_size)
    return index.search_finish(queries, k, all_reduce_min(lower, group=group))


class ShardedSparseRetriever:
    """Doc-sharded inverted-index retrieval: rank r holds the postings of dataset rows r, r + W, ... - exactly what
    `eval_sparse.py --task_name indexing` writes into index_dir_{r} (g_row = local * W + rank,
    /root/reference/scaling_retriever/indexer.py:262; /root/reference/eval_sparse.py:98-100) - as a CSR over LOCAL doc
    indices, scores its shard (score tiles over N / W docs) and returns global indices through id_base = rank,
    id_stride = W.  The exchange is the same single gather of per-shard top-k + sr_topk_merge as the dense path; no
    merge_indexes pass, no N-sized array anywhere.  A doc's postings live on one rank and are applied in the same term
    order, so the merged result equals the single-index search bit for bit."""

    def __init__(self, indptr, global_doc_ids, vals, n_docs_global, rank=None, world_size=None, device=None):
        from .scoring import SparseIndexHIP
        self.rank = dist.get_rank() if rank is None and dist.is_initialized() else (rank or 0)
        self.world_size = dist.get_world_size() if world_size is None and dist.is_initialized() else (world_size or 1)
        W, r = self.world_size, self.rank
        device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        to = lambda x, dt: (torch.from_numpy(x) if not torch.is_tensor(x) else x).to(device=device, dtype=dt)   # noqa: E731
        g = to(global_doc_ids, torch.int64)
        if g.numel() and bool(((g % W) != r).any()):
            raise ValueError(f"rank {r} of {W}: the index holds documents of other shards (global row % W != rank)")
        local = torch.div(g - r, W, rounding_mode="floor").to(torch.int32)
        indptr_t, vals_t = to(indptr, torch.int64), to(vals, torch.float32)
        self.n_local = max(1, len(range(r, int(n_docs_global), W)))
        if local.numel():      # posting lists sorted by doc (a merged / bucketed build is not necessarily): this library's stable radix
            # passes over the local rows, then over the terms (sr_sparse_csr_build with sort_docs, csrc/sparse_build.hip)
            from .scoring import sparse_csr_build, sparse_csr_expand_terms
            term = sparse_csr_expand_terms(indptr_t, local.numel())
            n_rows = max(self.n_local, int(local.max().item()) + 1)
            indptr2, local, vals_t = sparse_csr_build(local, term, vals_t, indptr_t.numel() - 1, n_docs=n_rows, sort_docs=True)
            assert torch.equal(indptr2, indptr_t)
        self.index = SparseIndexHIP(indptr_t, local, vals_t, self.n_local, device=device)

    def search(self, q_indptr, q_cols, q_vals, k, threshold=0.0, dst=0):
        """Queries replicated on every rank (CSR).  Returns (scores [nq, k], global ids [nq, k], counts [nq]) on rank dst,
        (None, None, None) elsewhere; rows padded with (0, -1) like sr_sparse_search."""
        from .scoring import topk_merge
        s, i, c = self.index.search(q_indptr, q_cols, q_vals, k, threshold=threshold, id_base=self.rank, id_stride=self.world_size)
        gs, gi = gather_topk(s, i, dst=dst)
        if gs is None:
            return None, None, None
        if gs.shape[0] == 1:
            return s, i, c
        ms, mi = topk_merge(gs, gi, pad_score=0.0)
        return ms, mi, (mi >= 0).sum(1).to(torch.int32)


class ShardedDenseRetriever:
    """Each rank holds rows rank, rank+W, ... of the corpus in its own HBM (DenseIndexHIP with
    id_base = rank, id_stride = W) and scores the replicated query matrix against them."""

    def __init__(self, hidden_dim, rank=None, world_size=None, precision="fp32_filtered"):
        from .scoring import DenseIndexHIP
        self.rank = dist.get_rank() if rank is None and dist.is_initialized() else (rank or 0)
        self.world_size = dist.get_world_size() if world_size is None and dist.is_initialized() else (world_size or 1)
        self.index = DenseIndexHIP(hidden_dim)
        self.index.set_precision(precision)      # the same ids and fp32 scores as "fp32", through the certified filter

    def add_local_rows(self, rows):
        """rows: fp32 cuda tensor [n_local, H] = the embeddings of dataset rows rank, rank+W, ..."""
        self.index.add_device_rows(rows, id_base=self.rank, id_stride=self.world_size)

    def search(self, queries, k, dst=0):
        """queries replicated on every rank.  Returns (scores, global ids) on rank dst, (None, None) elsewhere."""
        from .scoring import topk_merge
        s, i = sharded_dense_search(self.index, queries, k, self.world_size)
        gs, gi = gather_topk(s, i, dst=dst)
        if gs is None:
            return None, None
        if gs.shape[0] == 1:
            return gs[0], gi[0]
        return topk_merge(gs, gi)
