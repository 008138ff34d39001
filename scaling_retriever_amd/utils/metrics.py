"""Ranking metrics for the parity judge: MRR@k, recall@k, nDCG@k over {qid: {docid: score}} runs.

Mirror of the definitions in /root/reference/scaling_retriever/utils/metrics.py
(truncate_run :13-19, mrr_k :22-29, evaluate :47-65, load_and_evaluate :67-83), which call
pytrec_eval (C extension, not installed here) - restated from trec_eval's published
definitions: documents ranked by score descending, ties by docid descending; a query counts
only if it appears in both run and qrel; recip_rank = 1 / rank of the first doc with
relevance > 0; ndcg uses gains rel / log2(rank + 1) against the ideal ordering of the qrel.
"""
import json
import math


def truncate_run(run, k):
    out = {}
    for q_id in run:
        ranked = sorted(run[q_id].items(), key=lambda item: item[1], reverse=True)
        out[q_id] = dict(ranked[:k])
    return out


def _ranking(docs):
    return [d for d, _ in sorted(docs.items(), key=lambda kv: (kv[1], kv[0]), reverse=True)]


def mrr_k(run, qrel, k, agg=True):
    truncated = truncate_run(run, k)
    per_q = {}
    for q, docs in truncated.items():
        if q not in qrel:
            continue
        rr = 0.0
        for rank, d in enumerate(_ranking(docs), start=1):
            if qrel[q].get(d, 0) > 0:
                rr = 1.0 / rank
                break
        per_q[q] = {"recip_rank": rr}
    if agg:
        return sum(d["recip_rank"] for d in per_q.values()) / max(1, len(per_q))
    return per_q


def recall_k(run, qrel, k, agg=True):
    per_q = {}
    for q, docs in run.items():
        if q not in qrel:
            continue
        rel = {d for d, r in qrel[q].items() if r > 0}
        top = _ranking(docs)[:k]
        per_q[q] = (len(rel.intersection(top)) / len(rel)) if rel else 0.0
    return sum(per_q.values()) / max(1, len(per_q)) if agg else per_q


def ndcg_k(run, qrel, k, agg=True):
    per_q = {}
    for q, docs in run.items():
        if q not in qrel:
            continue
        top = _ranking(docs)[:k]
        dcg = sum(max(qrel[q].get(d, 0), 0) / math.log2(i + 2) for i, d in enumerate(top))
        ideal = sorted((r for r in qrel[q].values() if r > 0), reverse=True)[:k]
        idcg = sum(r / math.log2(i + 2) for i, r in enumerate(ideal))
        per_q[q] = dcg / idcg if idcg > 0 else 0.0
    return sum(per_q.values()) / max(1, len(per_q)) if agg else per_q


def evaluate(run, qrel, metric, agg=True, select=None):
    """metric in {"recall", "ndcg_cut", "recip_rank"}; returns trec-style keys (recall_10, ndcg_cut_10 ...)."""
    cuts = [5, 10, 15, 20, 30, 100, 200, 500, 1000]
    if metric == "recall":
        res = {f"recall_{c}": recall_k(run, qrel, c) for c in cuts}
    elif metric == "ndcg_cut":
        res = {f"ndcg_cut_{c}": ndcg_k(run, qrel, c) for c in cuts}
    elif metric == "recip_rank":
        res = {"recip_rank": mrr_k(run, qrel, 10 ** 9)}
    else:
        raise ValueError("provide valid metric")
    if select is not None:
        return res.get("{}_{}".format(metric, select), 0)
    return res


def load_and_evaluate(qrel_file_path, run_file_path, metric):
    with open(qrel_file_path) as reader:
        qrel = json.load(reader)
    with open(run_file_path) as reader:
        run = json.load(reader)
    if metric == "mrr_10":
        res = mrr_k(run, qrel, k=10)
        print("MRR@10:", res)
        return {"mrr_10": res}
    res = evaluate(run, qrel, metric=metric)
    print(metric, "==>", res)
    return res


def recall_cap_k(run, qrel, k, agg=True):
    """beir's capped recall R_cap@k [3P beir==2.0.0 EvaluateRetrieval.evaluate_custom(metric="r_cap") -> custom_metrics.recall_cap,
    restated from its published source]: for EVERY query of the run, relevant docs among its top k (by score) over
    min(k, number of relevant docs), averaged over len(run).  As in beir, a query of the run that the qrels do not hold is a
    KeyError and a query without relevant documents a ZeroDivisionError (the earlier version skipped the former and
    scored the latter 0, so R_cap@100 could differ from the reference's perf.json when the run held queries outside the split)."""
    per_q = {}
    for q, docs in run.items():
        rel = {d for d, r in qrel[q].items() if r > 0}
        top = [d for d, _ in sorted(docs.items(), key=lambda kv: kv[1], reverse=True)[:k]]
        per_q[q] = len([d for d in top if qrel[q].get(d, 0) > 0]) / min(len(rel), k)
    return sum(per_q.values()) / len(run) if agg else per_q


def evaluate_beir(args, qrels):
    """/root/reference/scaling_retriever/utils/metrics.py:131-151: drop hits whose doc id equals the query id, then
    NDCG@10, Recall@100 (trec_eval definitions, over the queries of the qrels that the run answers) and beir's
    R_cap@100; written to {out_dir}/perf.json.  beir rounds its figures to 5 decimals."""
    import os
    with open(os.path.join(args.out_dir, "run.json")) as reader:
        run = json.load(reader)
    print("Removing query id from document list")
    new_run = {qid: {d: v for d, v in docs.items() if d != qid} for qid, docs in run.items()}
    qrels = {str(q): {str(d): int(r) for d, r in docs.items()} for q, docs in qrels.items()}
    res = {"NDCG@10": round(ndcg_k(new_run, qrels, 10), 5), "Recall@100": round(recall_k(new_run, qrels, 100), 5),
           "R_cap@100": round(recall_cap_k(new_run, qrels, 100), 5)}
    with open(os.path.join(args.out_dir, "perf.json"), "w") as writer:
        json.dump(res, writer, indent=4)
    return res
