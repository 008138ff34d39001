"""CPU tests of the tokenise / collate pipeline (scaling_retriever_amd/dataset/pipeline.py): the token-budget batches
hold exactly the rows, tokens and ids the reference's sequential collator
(/root/reference/scaling_retriever/dataset/data_collator.py:177-190) produces, only grouped differently."""
import numpy as np
import pytest
import torch
from torch.utils.data import DataLoader

from fake_tokenizer import FakeTokenizer
from scaling_retriever_amd.dataset.data_collator import LlamaDenseCollectionCollator
from scaling_retriever_amd.dataset.pipeline import (TokenBudgetCollectionLoader, length_bucketed_batches,
                                                    token_budget_batches)


class ListDataset(torch.utils.data.Dataset):
    def __init__(self, items):
        self.items = items

    def __len__(self):
        return len(self.items)

    def __getitem__(self, i):
        return self.items[i]


def _corpus(n, seed=0, max_words=60):
    rng = np.random.default_rng(seed)
    return [(f"p{i}", " ".join(f"w{int(x)}" for x in rng.integers(0, 5000, size=int(rng.integers(1, max_words)))))
            for i in range(n)]


def _rows(batch):
    """{id: unpadded token list} of a collated batch."""
    out = {}
    for i, pid in enumerate(batch["ids"]):
        m = batch["attention_mask"][i].bool()
        out[pid] = batch["input_ids"][i][m].tolist()
    return out


def test_token_budget_packing_invariants():
    rng = np.random.default_rng(1)
    lengths = rng.integers(1, 193, size=5000)
    batches = length_bucketed_batches(lengths, max_tokens=4096, max_seqs=64, window=1000)
    flat = [i for b in batches for i in b]
    assert sorted(flat) == list(range(5000))                      # every row exactly once
    for b in batches:
        assert len(b) <= 64 and (lengths[b].sum() <= 4096 or len(b) == 1)
        assert max(b) // 1000 == min(b) // 1000                   # a batch never straddles a window
        assert list(lengths[b]) == sorted(lengths[b])             # bucketed: ascending lengths inside a batch
    # budgets are used: all but the last batch of a window are within one max-length row of full (or at the row cap)
    full = [b for b in batches if len(b) < 64]
    assert np.mean([lengths[b].sum() for b in full]) > 0.9 * 4096
    assert token_budget_batches([0, 1, 2], [5000, 10, 10], 4096, 8) == [[0], [1, 2]]     # an over-long row stands alone


@pytest.mark.parametrize("side", ["left", "right"])
@pytest.mark.parametrize("workers", [0, 2])
def test_pipeline_matches_sequential_collator(side, workers):
    tok = FakeTokenizer(padding_side=side)
    data = ListDataset(_corpus(700))
    ref_loader = DataLoader(data, batch_size=128, shuffle=False, collate_fn=LlamaDenseCollectionCollator(tok, max_length=48))
    ref = {}
    for b in ref_loader:
        ref.update(_rows(b))
    loader = TokenBudgetCollectionLoader(data, tok, max_length=48, max_tokens=1024, max_seqs=40, window=256,
                                         num_workers=workers, chunk=100, pin_memory=False)
    got, n_batches = {}, 0
    for b in loader:
        B, L = b["input_ids"].shape
        assert b["attention_mask"].shape == (B, L) and len(b["ids"]) == B and B <= 40
        assert int(b["attention_mask"].sum()) <= 1024
        lens = b["attention_mask"].sum(1)
        assert int(lens.max()) == L                                # pad-to-longest
        assert bool((b["input_ids"][b["attention_mask"] == 0] == tok.pad_token_id).all())
        if side == "left":
            assert bool((b["attention_mask"][:, -1] == 1).all())   # what the dense head's [-length:] slice needs
        else:
            assert bool((b["attention_mask"][:, 0] == 1).all())
        rows = _rows(b)
        assert not (set(rows) & set(got))
        got.update(rows)
        n_batches += 1
    assert got == ref                                              # same ids, same tokens per id (truncation included)
    assert n_batches < 700 / 20


def test_pipeline_rank_shards_cover_the_collection_once():
    tok = FakeTokenizer()
    data = ListDataset(_corpus(333, seed=3))
    seen = []
    for r in range(4):
        loader = TokenBudgetCollectionLoader(data, tok, max_length=32, max_tokens=512, max_seqs=64, num_workers=0, rank=r,
                                             world_size=4, pin_memory=False)
        ids = [pid for b in loader for pid in b["ids"]]
        assert sorted(ids, key=lambda p: int(p[1:])) == [f"p{i}" for i in range(r, 333, 4)]     # rows r, r + W, ...
        seen += ids
    assert sorted(seen) == sorted(f"p{i}" for i in range(333))     # no wrap-around duplicates


def test_pretokenized_chunks_and_missing_pad_token():
    lens = np.array([3, 1, 2], np.int32)
    chunk = (["a", "b", "c"], np.arange(6, dtype=np.int32), lens)
    loader = TokenBudgetCollectionLoader(tokenized=[chunk], max_length=8, max_tokens=4, max_seqs=8, pad_token_id=99,
                                         padding_side="left", pin_memory=False)
    batches = [{k: (v.clone() if torch.is_tensor(v) else v) for k, v in b.items()} for b in loader]
    assert [b["ids"] for b in batches] == [["b", "c"], ["a"]]
    assert batches[0]["input_ids"].tolist() == [[99, 3], [4, 5]] and batches[1]["input_ids"].tolist() == [[0, 1, 2]]
    with pytest.raises(ValueError):
        TokenBudgetCollectionLoader(tokenized=[chunk], padding_side="left")
