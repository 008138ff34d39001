"""Eval collator: (id, text) pairs -> {"input_ids", "attention_mask", "ids"}.

Mirror of LlamaSparseCollectionCollator = LlamaDenseCollectionCollator
(/root/reference/scaling_retriever/dataset/data_collator.py:177-193): tokenizer(texts, max_length, truncation=True,
padding="longest", return_tensors="pt"); special tokens on (BOS), nothing appended.  The padding side is the
tokenizer's (eval_dense.py sets it to "left", which the dense head's `[-length:]` slice requires)."""


class LlamaSparseCollectionCollator:
    def __init__(self, tokenizer, max_length):
        self.tokenizer = tokenizer
        self.max_length = max_length

    def __call__(self, batch):
        ids, texts = [list(xs) for xs in zip(*batch)]
        tok = self.tokenizer(texts, max_length=self.max_length, truncation=True, padding="longest", return_tensors="pt")
        return {**{k: v for k, v in tok.items()}, "ids": ids}


LlamaDenseCollectionCollator = LlamaSparseCollectionCollator
