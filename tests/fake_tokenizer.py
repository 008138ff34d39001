"""A deterministic stand-in for the HF tokenizer call surface the eval collators use (no tokenizer files offline):
tokenizer(texts, max_length=, truncation=True, padding="longest" | False, return_tensors="pt" | None) -> input_ids
(+ attention_mask), BOS prepended, nothing appended, pad_token == eos_token, padding_side attribute."""
import zlib

import torch


class FakeTokenizer:
    def __init__(self, vocab_size=512, padding_side="left"):
        self.vocab_size = vocab_size
        self.bos_token_id, self.eos_token_id = 1, 2
        self.pad_token_id = self.eos_token_id
        self.pad_token = self.eos_token = "</s>"
        self.padding_side = padding_side

    def _encode(self, text, max_length, truncation):
        toks = [self.bos_token_id] + [3 + zlib.crc32(w.encode()) % (self.vocab_size - 3) for w in text.split()]
        return toks[:max_length] if (truncation and max_length) else toks

    def __call__(self, texts, max_length=None, truncation=False, padding=False, return_tensors=None):
        if isinstance(texts, str):
            texts = [texts]
        rows = [self._encode(t, max_length, truncation) for t in texts]
        if not padding:
            return {"input_ids": rows, "attention_mask": [[1] * len(r) for r in rows]}
        L = max(len(r) for r in rows)
        ids = torch.full((len(rows), L), self.pad_token_id, dtype=torch.int64)
        mask = torch.zeros((len(rows), L), dtype=torch.int64)
        for i, r in enumerate(rows):
            if self.padding_side == "left":
                ids[i, L - len(r):] = torch.tensor(r)
                mask[i, L - len(r):] = 1
            else:
                ids[i, :len(r)] = torch.tensor(r)
                mask[i, :len(r)] = 1
        return {"input_ids": ids, "attention_mask": mask}
