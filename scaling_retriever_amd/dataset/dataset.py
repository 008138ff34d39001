"""Eval-path datasets: the input side of the hot path (boundary contract only).

Same record shapes as /root/reference/scaling_retriever/dataset/dataset.py:
read_msmarco_corpus / read_msmarco_query (:21-35), CollectionDataset (:170-186) -> (pid, text),
MSMARCOQueryDataset (:205-218) -> (qid, query).  MS MARCO TSV: `id<TAB>text`, ids stay strings.
"""
from torch.utils.data import Dataset


def _read_two_column_tsv(path):
    out = {}
    with open(path) as fin:
        for line in fin:
            key, text = line.rstrip("\n").split("\t", 1)
            out[key] = text.strip()
    return out


def read_msmarco_corpus(corpus_path):
    return {pid: (None, text) for pid, text in _read_two_column_tsv(corpus_path).items()}


def read_msmarco_query(query_path):
    return _read_two_column_tsv(query_path)


def read_wiki_corpus(corpus_path):
    """psgs_w100-style TSV with a header: id, text, title."""
    out = {}
    with open(corpus_path) as fin:
        for i, line in enumerate(fin):
            if i == 0:
                continue
            pid, text, title = line.strip().split("\t")
            out[pid] = (title, text)
    return out


def get_doc_text(title, text):
    return text if title is None else f"title: {title} | context: {text}"


class CollectionDataset(Dataset):
    def __init__(self, corpus_path, data_source=None):
        if data_source == "msmarco":
            self.pid_to_doc = read_msmarco_corpus(corpus_path)
        elif data_source == "wiki":
            self.pid_to_doc = read_wiki_corpus(corpus_path)
        else:
            raise NotImplementedError(f"data_source={data_source!r}")
        self.pids = list(self.pid_to_doc.keys())

    def __len__(self):
        return len(self.pids)

    def __getitem__(self, idx):
        pid = self.pids[idx]
        return pid, get_doc_text(*self.pid_to_doc[pid])


class MSMARCOQueryDataset(Dataset):
    def __init__(self, query_path):
        self.qid_to_query = read_msmarco_query(query_path)
        self.qids = list(self.qid_to_query.keys())

    def __len__(self):
        return len(self.qids)

    def __getitem__(self, idx):
        qid = self.qids[idx]
        return qid, self.qid_to_query[qid]


class BeirDataset(Dataset):
    """dict id -> {"title", "text"} (corpus) or id -> text (queries), as beir's GenericDataLoader returns."""

    def __init__(self, data, information_type="document"):
        self.ids = list(data.keys())
        if information_type == "document":
            self.texts = [get_doc_text(d.get("title") or None, d["text"]) for d in data.values()]
        else:
            self.texts = list(data.values())

    def __len__(self):
        return len(self.ids)

    def __getitem__(self, idx):
        return self.ids[idx], self.texts[idx]
