"""CPU: host logic of store_embs (/root/reference/scaling_retriever/indexer.py:26-97) with a stand-in encoder - grouping of the
reference's fixed-size loader batches into one engine pass must not change the artefacts: chunk files keep the reference's row
counts (write_freq = chunk_size // batch_size batches, :32-33, :60), ids and rows stay in loader order, plan.json names the chunks."""
import json
import os

import numpy as np
import torch

from scaling_retriever_amd import indexer
from scaling_retriever_amd.utils.utils import obtain_doc_vec_dir_files


class _Enc(torch.nn.Module):
    """doc_encode: a deterministic function of each row's real tokens; encode_batches: the per-batch results, counted."""
    H = 6

    def __init__(self, with_groups):
        super().__init__()
        self.calls, self.group_calls = 0, 0
        if with_groups:
            self.encode_batches = self._encode_batches

    def doc_encode(self, input_ids, attention_mask):
        self.calls += 1
        x = (input_ids * attention_mask).to(torch.float32)
        feats = [x.sum(1), (x * x).sum(1), attention_mask.sum(1).float(), x.max(1).values, x[:, -1], (x % 7).sum(1)]
        return torch.stack(feats, 1)

    def _encode_batches(self, batches):
        self.group_calls += 1
        return torch.cat([self.doc_encode(**b) for b in batches])


class _Loader:
    def __init__(self, n, batch_size, fixed=True):
        self.n, self.bs = n, batch_size
        if fixed:
            self.batch_size = batch_size

    def __len__(self):
        return (self.n + self.bs - 1) // self.bs

    def __iter__(self):
        rng = np.random.default_rng(1)
        for b0 in range(0, self.n, self.bs):
            m = min(self.bs, self.n - b0)
            L = int(rng.integers(3, 9))
            ids = torch.from_numpy(rng.integers(1, 50, size=(m, L)))
            mask = torch.ones((m, L), dtype=torch.int64)
            mask[:, :int(rng.integers(0, 2))] = 0
            yield {"input_ids": ids, "attention_mask": mask, "ids": [f"p{b0 + i}" for i in range(m)]}


def _run(tmp_path, name, enc, loader, chunk_size):
    d = str(tmp_path / name)
    indexer.store_embs(enc, loader, local_rank=0, index_dir=d, device="cpu", chunk_size=chunk_size)
    vec_files, id_files = obtain_doc_vec_dir_files(d)
    plan = json.load(open(os.path.join(d, "plan.json")))
    return [np.load(f) for f in vec_files], [np.load(f).tolist() for f in id_files], plan


def test_grouped_fixed_size_batches_write_the_per_batch_artefacts(tmp_path, monkeypatch):
    monkeypatch.setattr(indexer, "STORE_GROUP_ROWS", 24)          # 3 batches of 8 per engine pass
    n, bs, chunk = 77, 8, 20                                       # chunk closes every 20 // 8 = 2 batches = 16 rows
    grouped, plain = _Enc(True), _Enc(False)
    e1, i1, p1 = _run(tmp_path, "grouped", grouped, _Loader(n, bs), chunk)
    e2, i2, p2 = _run(tmp_path, "plain", plain, _Loader(n, bs), chunk)
    assert grouped.group_calls == 3 and grouped.calls == 10 and plain.calls == 10      # 10 batches: groups of 3, 3, 3; the last batch goes alone
    assert p1["num_chunks"] == p2["num_chunks"] == 5 and p1["nranks"] == 1
    assert [len(x) for x in e1] == [16, 16, 16, 16, 13] == [len(x) for x in e2]
    assert i1 == i2 and sum(i1, []) == [f"p{i}" for i in range(n)]
    assert all(np.array_equal(a, b) and a.dtype == np.float32 for a, b in zip(e1, e2))


def test_token_budget_loaders_are_not_grouped(tmp_path):
    enc = _Enc(True)
    e, ids, plan = _run(tmp_path, "budget", enc, _Loader(30, 7, fixed=False), 12)      # no batch_size attribute: sized for the engine already
    assert enc.group_calls == 0 and enc.calls == 5 and sum(ids, []) == [f"p{i}" for i in range(30)]
    assert sum(len(x) for x in e) == 30
