"""Tokenise / collate pipeline feeding doc_encode at GPU speed.

The reference feeds its encoder from DataLoader(batch_size=128, num_workers=1) + LlamaDenseCollectionCollator
(/root/reference/eval_dense.py:171-179, /root/reference/scaling_retriever/dataset/data_collator.py:177-190): one worker
tokenises, every batch is 128 passages padded to its longest.  At ~7 000 passages/s per GPU that loader is the
bottleneck, and the batch shape (whatever 128 passages happen to add up to) is not what the GEMMs want.

Here:
  * tokenisation runs in `num_workers` processes over chunks of texts (no padding: a row's tokens do not depend on its
    batch - truncation to max_length and the BOS token are per text - so the result per passage is exactly what the
    reference's collator produces for it);
  * inside a window of consecutive passages the rows are bucketed by length (stable sort) and packed greedily into
    batches of at most `max_tokens` REAL tokens (the encoder computes no pad tokens, so the GEMM M dimension is the token
    count: a budget that is a multiple of 8 192 makes every layer GEMM a whole number of 256-CU rounds of 256 x 256
    tiles) and at most `max_seqs` rows; bucketing also keeps the padded [B, L] tensors that cross PCIe small;
  * batches are collated into a ring of pinned host buffers (pad-to-longest, the tokenizer's padding side).
This is parity-neutral: pads are masked keys that never enter the computation, each passage's embedding depends only on
its own tokens, and the ids travel with the rows (ids_*.npy records the order), so the artefacts hold the same
(pid -> vector) content as the sequential loader's (tests/test_pipeline.py on CPU; on the GPU tests/test_eval_drivers.py compares both loaders' artefacts with the oracle).
"""
import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset


def token_budget_batches(order, lengths, max_tokens, max_seqs):
    """Greedy packing of the row indices `order` (kept in that order) into batches with sum(lengths) <= max_tokens and
    len <= max_seqs.  A row longer than max_tokens gets a batch of its own."""
    out, cur, tok = [], [], 0
    for i in order:
        n = int(lengths[i])
        if cur and (tok + n > max_tokens or len(cur) >= max_seqs):
            out.append(cur)
            cur, tok = [], 0
        cur.append(int(i))
        tok += n
    if cur:
        out.append(cur)
    return out


def length_bucketed_batches(lengths, max_tokens, max_seqs, window):
    """Row indices 0..n-1 -> list of batches.  Rows are bucketed by length only inside windows of `window` consecutive
    rows, so a chunk file still holds a contiguous range of the collection."""
    lengths = np.asarray(lengths)
    out = []
    for w0 in range(0, len(lengths), window):
        idx = np.arange(w0, min(w0 + window, len(lengths)))
        order = idx[np.argsort(lengths[idx], kind="stable")]
        out.extend(token_budget_batches(order, lengths, max_tokens, max_seqs))
    return out


def _identity(x):
    return x


class _TokenizeChunks(Dataset):
    """Item c = rows [c * chunk, (c + 1) * chunk) of this rank's shard, tokenised without padding."""

    def __init__(self, dataset, rows, tokenizer, max_length, chunk):
        self.dataset, self.rows, self.tokenizer, self.max_length, self.chunk = dataset, rows, tokenizer, max_length, chunk

    def __len__(self):
        return (len(self.rows) + self.chunk - 1) // self.chunk

    def __getitem__(self, c):
        rows = self.rows[c * self.chunk:(c + 1) * self.chunk]
        ids, texts = zip(*[self.dataset[r] for r in rows])
        tok = self.tokenizer(list(texts), max_length=self.max_length, truncation=True, padding=False)["input_ids"]
        lengths = np.fromiter((len(t) for t in tok), dtype=np.int32, count=len(tok))
        flat = np.concatenate([np.asarray(t, dtype=np.int32) for t in tok]) if len(tok) else np.zeros(0, np.int32)
        return list(ids), flat, lengths


class PinnedBatchRing:
    """`depth` pairs of pinned int64 [max_seqs * max_len] buffers; a collated batch stays valid until `depth - 1` more have
    been drawn (store_embs / SparseIndexer.index copy a batch to the device before asking for the next one)."""

    def __init__(self, max_elems, depth=4, pin=True):
        pin = bool(pin and torch.cuda.is_available())
        self.bufs = [(torch.empty(max_elems, dtype=torch.int64, pin_memory=pin),
                      torch.empty(max_elems, dtype=torch.int64, pin_memory=pin)) for _ in range(depth)]
        self.i = 0

    def take(self, B, L):
        ids, mask = self.bufs[self.i]
        self.i = (self.i + 1) % len(self.bufs)
        if B * L > ids.numel():          # a batch beyond the planned size: one-off buffers
            return torch.empty((B, L), dtype=torch.int64), torch.empty((B, L), dtype=torch.int64)
        return ids[:B * L].view(B, L), mask[:B * L].view(B, L)


def collate_rows(rows, flat, offsets, lengths, pad_id, left, ring):
    """Pad-to-longest collation of ragged token rows (what tokenizer(..., padding="longest") returns)."""
    B = len(rows)
    L = int(max(lengths[r] for r in rows)) if B else 0
    ids, mask = ring.take(B, max(L, 1))
    ids.fill_(pad_id)
    mask.zero_()
    ids_np, mask_np = ids.numpy(), mask.numpy()
    for b, r in enumerate(rows):
        n = int(lengths[r])
        t = flat[offsets[r]:offsets[r] + n]
        if left:
            ids_np[b, L - n:] = t
            mask_np[b, L - n:] = 1
        else:
            ids_np[b, :n] = t
            mask_np[b, :n] = 1
    return ids, mask


class TokenBudgetCollectionLoader:
    """Iterable over {"input_ids", "attention_mask", "ids"} batches of this rank's shard of `dataset` (items = (id, text)).

    rank r takes dataset rows r, r + W, ... (the DistributedSampler(shuffle=False) assignment of eval_dense.py:178
    without its wrap-around duplicates).  `tokenized` may instead supply an iterator of already tokenised chunks
    (ids list, flat int32 tokens, int32 lengths) - used by bench.py, which has no tokenizer offline."""

    batch_size = None        # batches are sized by tokens, not rows

    def __init__(self, dataset=None, tokenizer=None, max_length=192, max_tokens=16384, max_seqs=1024, window=32768,
                 num_workers=4, chunk=2048, rank=0, world_size=1, pad_token_id=None, padding_side=None, tokenized=None,
                 pin_memory=True):
        assert (dataset is None) != (tokenized is None), "pass either (dataset, tokenizer) or tokenized chunks"
        self.dataset, self.tokenizer, self.max_length = dataset, tokenizer, int(max_length)
        self.max_tokens, self.max_seqs, self.window = int(max_tokens), int(max_seqs), int(window)
        self.num_workers, self.chunk = int(num_workers), int(chunk)
        self.rows = list(range(rank, len(dataset), world_size)) if dataset is not None else None
        self.tokenized = tokenized
        if pad_token_id is None:
            pad_token_id = getattr(tokenizer, "pad_token_id", None)
        if pad_token_id is None:
            raise ValueError("pad_token_id is not set (the reference asserts pad_token == eos_token, eval_dense.py:186)")
        self.pad_id = int(pad_token_id)
        side = padding_side or getattr(tokenizer, "padding_side", "right")
        self.left = side == "left"
        self.ring = PinnedBatchRing(self.max_seqs * self.max_length, pin=pin_memory)
        self.n_rows = len(self.rows) if self.rows is not None else None

    def _chunks(self):
        if self.tokenized is not None:
            yield from self.tokenized
            return
        ds = _TokenizeChunks(self.dataset, self.rows, self.tokenizer, self.max_length, self.chunk)
        kw = dict(prefetch_factor=4, persistent_workers=False) if self.num_workers > 0 else {}
        loader = DataLoader(ds, batch_size=None, shuffle=False, num_workers=self.num_workers, collate_fn=_identity, **kw)
        yield from loader

    def _emit(self, ids, flats, lens):
        lengths = np.concatenate(lens)
        flat = np.concatenate(flats)
        offsets = np.concatenate([[0], np.cumsum(lengths[:-1], dtype=np.int64)]) if len(lengths) else np.zeros(0, np.int64)
        for rows in length_bucketed_batches(lengths, self.max_tokens, self.max_seqs, max(len(lengths), 1)):
            t_ids, t_mask = collate_rows(rows, flat, offsets, lengths, self.pad_id, self.left, self.ring)
            yield {"input_ids": t_ids, "attention_mask": t_mask, "ids": [ids[r] for r in rows]}

    def __iter__(self):
        ids, flats, lens, n = [], [], [], 0
        for c_ids, c_flat, c_len in self._chunks():
            ids.extend(c_ids)
            flats.append(np.asarray(c_flat, dtype=np.int32))
            lens.append(np.asarray(c_len, dtype=np.int32))
            n += len(c_ids)
            if n >= self.window:
                yield from self._emit(ids, flats, lens)
                ids, flats, lens, n = [], [], [], 0
        if n:
            yield from self._emit(ids, flats, lens)
