"""BEIR dataset folders without the `beir` package (not installed; no network).

The reference downloads `{beir_dataset}.zip` and reads it with beir's GenericDataLoader
(/root/reference/eval_dense.py:164-168,211-215,245-248; /root/reference/eval_sparse.py:80-85,116-121,189-192).  The
folder layout that loader reads - and this one reads from `{beir_dataset_dir}/{beir_dataset}` - is
  corpus.jsonl    {"_id": ..., "title": ..., "text": ...} per line
  queries.jsonl   {"_id": ..., "text": ...} per line
  qrels/{split}.tsv   header line, then  query-id <TAB> corpus-id <TAB> score
and, like GenericDataLoader.load, only the queries that have a qrel entry in the split are returned."""
import csv
import json
import os


def beir_data_path(beir_dataset_dir, beir_dataset):
    path = os.path.join(beir_dataset_dir, beir_dataset) if beir_dataset else beir_dataset_dir
    if not os.path.isdir(path):
        raise FileNotFoundError(f"BEIR folder {path} not found (downloading is not possible here: unpack {beir_dataset}.zip there)")
    return path


def load_beir(beir_dataset_dir, beir_dataset=None, split="test"):
    """-> (corpus {id: {"title", "text"}}, queries {id: text}, qrels {qid: {docid: int}})."""
    path = beir_data_path(beir_dataset_dir, beir_dataset)
    corpus, queries, qrels = {}, {}, {}
    with open(os.path.join(path, "corpus.jsonl"), encoding="utf8") as f:
        for line in f:
            if line.strip():
                d = json.loads(line)
                corpus[d["_id"]] = {"text": d.get("text", ""), "title": d.get("title", "")}
    with open(os.path.join(path, "queries.jsonl"), encoding="utf8") as f:
        for line in f:
            if line.strip():
                d = json.loads(line)
                queries[d["_id"]] = d.get("text", "")
    with open(os.path.join(path, "qrels", split + ".tsv"), encoding="utf8") as f:
        reader = csv.reader(f, delimiter="\t", quoting=csv.QUOTE_MINIMAL)
        next(reader)
        for row in reader:
            qid, did, score = row[0], row[1], int(row[2])
            qrels.setdefault(qid, {})[did] = score
    queries = {qid: queries[qid] for qid in qrels if qid in queries}
    return corpus, queries, qrels
