#!/usr/bin/env python3
"""Sparse eval driver on MI355X: same tasks, flags and artefacts as /root/reference/eval_sparse.py
(SparseArguments :34-72; tasks indexing :75-106, retrieval :109-151, evaluate_msmarco), running the
HIP LlamaBiSparse encoder and the HIP inverted-index scorer.

  torchrun --nproc_per_node=2 eval_sparse.py --task_name indexing --model_name_or_path <lora dir> \
           --corpus_path <tsv> --index_dir <dir>/index --eval_batch_size 64 --doc_max_length 192
  python   -m scaling_retriever_amd.utils.inverted_index --model_name_or_path <base dir> --index_dir <dir>   (merge)
  python   eval_sparse.py --task_name retrieval --model_name_or_path <lora dir> --query_path <tsv> \
           --index_dir <dir>/index --out_dir <dir> --top_k 1000
"""
import argparse
import ast
import json
import os

import torch
import torch.distributed as dist
from torch.utils.data import DataLoader
from torch.utils.data.distributed import DistributedSampler

CORPUS_DATASOURCE = {"./data/msmarco-full/full_collection/raw.tsv": "msmarco"}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    for name in ["model_name_or_path", "corpus_path", "index_dir", "out_dir", "query_path", "eval_run_path",
                 "eval_qrel_path", "eval_metric", "beir_dataset", "beir_dataset_dir", "access_token"]:
        ap.add_argument("--" + name, type=str, default=None)
    ap.add_argument("--task_name", type=str, default="")
    ap.add_argument("--data_source", type=str, default=None)
    ap.add_argument("--is_beir", action="store_true")
    ap.add_argument("--eval_batch_size", type=int, default=128)   # SparseArguments.eval_batch_size (eval_sparse.py:50)
    ap.add_argument("--doc_max_length", type=int, default=192)
    ap.add_argument("--query_max_length", type=int, default=64)
    ap.add_argument("--top_k", type=int, default=100)             # SparseArguments.top_k (eval_sparse.py:56); scripts pass 1000
    ap.add_argument("--local_rank", type=int, default=-1)
    ap.add_argument("--world_size", type=int, default=1)
    ap.add_argument("--token_budget", type=int, default=16384,
                    help="real tokens per encode batch of the indexing task (length-bucketed, multi-worker tokenisation); "
                         "0 = the reference's loader (eval_sparse.py:94-97)")
    ap.add_argument("--tokenize_workers", type=int, default=4)
    args = ap.parse_args(argv)
    if args.eval_metric:
        args.eval_metric = ast.literal_eval(args.eval_metric)
    return args


def ddp_setup(args):
    if "LOCAL_RANK" in os.environ and int(os.environ.get("WORLD_SIZE", "1")) > 1:
        args.local_rank = int(os.environ["LOCAL_RANK"])
        if os.environ.get("SR_SHARE_GPU") == "1":
            # dry run of the multi-rank path on a ONE-GPU box (tests): every rank on cuda:0, gloo instead of RCCL, which refuses
            # two ranks on one device; real runs leave it unset
            args.local_rank = 0
            torch.cuda.set_device(0)
            dist.init_process_group(backend="gloo")
        else:
            torch.cuda.set_device(args.local_rank)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", args.local_rank))
        args.world_size = dist.get_world_size()
    else:
        args.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(args.local_rank)
        args.world_size = 1


def _tokenizer(path):
    from transformers import AutoTokenizer
    tok = AutoTokenizer.from_pretrained(path)
    if tok.pad_token is None:
        tok.pad_token = tok.eos_token
    return tok     # eval_sparse.py never sets padding_side: the max-pool head does not depend on it


def _query_dataset(args):
    from scaling_retriever_amd.dataset.dataset import BeirDataset, MSMARCOQueryDataset
    if args.is_beir and args.beir_dataset is not None:             # eval_sparse.py:116-121
        from scaling_retriever_amd.utils.beir import load_beir
        _, queries, _ = load_beir(args.beir_dataset_dir, args.beir_dataset, split="test")
        return BeirDataset(queries, information_type="query")
    return MSMARCOQueryDataset(args.query_path)


def _collection_dataset(args):
    from scaling_retriever_amd.dataset.dataset import BeirDataset, CollectionDataset
    if args.is_beir and args.beir_dataset is not None:             # eval_sparse.py:80-85
        from scaling_retriever_amd.utils.beir import load_beir
        corpus, _, _ = load_beir(args.beir_dataset_dir, args.beir_dataset, split="test")
        return BeirDataset(corpus, information_type="document")
    source = args.data_source or CORPUS_DATASOURCE.get(args.corpus_path, "msmarco")
    return CollectionDataset(corpus_path=args.corpus_path, data_source=source)


def sparse_index(args):
    from scaling_retriever_amd.dataset.data_collator import LlamaSparseCollectionCollator
    from scaling_retriever_amd.indexer import SparseIndexer
    from scaling_retriever_amd.modeling.llm_encoder import LlamaBiSparse
    tokenizer = _tokenizer(args.model_name_or_path)
    collection = _collection_dataset(args)
    model = LlamaBiSparse.load_from_lora(args.model_name_or_path)
    if args.token_budget > 0:
        from scaling_retriever_amd.dataset.pipeline import TokenBudgetCollectionLoader
        loader = TokenBudgetCollectionLoader(collection, tokenizer, max_length=args.doc_max_length, max_tokens=args.token_budget,
                                             max_seqs=256, num_workers=args.tokenize_workers,
                                             rank=dist.get_rank() if args.world_size > 1 else 0, world_size=args.world_size)
    else:
        sampler = DistributedSampler(collection, shuffle=False) if args.world_size > 1 else None
        loader = DataLoader(collection, batch_size=args.eval_batch_size, shuffle=False, num_workers=2, sampler=sampler,
                            collate_fn=LlamaSparseCollectionCollator(tokenizer=tokenizer, max_length=args.doc_max_length))
    index_dir = args.index_dir[:-1] if args.index_dir.endswith("/") else args.index_dir
    if args.world_size > 1:
        index_dir = f"{index_dir}_{dist.get_rank()}"               # eval_sparse.py:98-100
    indexer = SparseIndexer(model, index_dir=index_dir, compute_stats=True, dim_voc=model.vocab_size, device=args.local_rank)
    indexer.index(loader)
    if args.world_size > 1:
        dist.barrier()


def sparse_retrieval(args):
    from scaling_retriever_amd.dataset.data_collator import LlamaSparseCollectionCollator
    from scaling_retriever_amd.indexer import SparseRetrieval
    from scaling_retriever_amd.modeling.llm_encoder import LlamaBiSparse
    tokenizer = _tokenizer(args.model_name_or_path)
    queries = _query_dataset(args)
    model = LlamaBiSparse.load_from_lora(args.model_name_or_path)
    collate = LlamaSparseCollectionCollator(tokenizer=tokenizer, max_length=args.query_max_length)
    os.makedirs(args.out_dir, exist_ok=True)
    config = {"index_dir": args.index_dir, "out_dir": args.out_dir}
    if args.world_size > 1:
        # The reference asserts world_size == 1 here (eval_sparse.py:114) and needs merge_indexes first.  Doc-sharded: each rank
        # scores the index_dir_{rank} it built, encodes its block of the queries, ONE gather of per-shard top-k.
        from torch.utils.data import Subset
        from scaling_retriever_amd.distributed import query_slice
        from scaling_retriever_amd.indexer import ShardedSparseRetrieval
        lo, hi = query_slice(len(queries), dist.get_rank(), args.world_size)
        q_loader = DataLoader(Subset(queries, range(lo, hi)), batch_size=args.eval_batch_size, shuffle=False, num_workers=0,
                              collate_fn=collate)
        retriever = ShardedSparseRetrieval(config=config, model=model, compute_stats=True, dim_voc=model.vocab_size,
                                           device=args.local_rank)
        res = retriever.retrieve(q_loader, topk=args.top_k, threshold=0.0)
        dist.barrier()
        return res
    q_loader = DataLoader(queries, batch_size=args.eval_batch_size, shuffle=False, num_workers=0, collate_fn=collate)
    retriever = SparseRetrieval(config=config, model=model, compute_stats=True, dim_voc=model.vocab_size, device=args.local_rank)
    return retriever.retrieve(q_loader, topk=args.top_k, threshold=0.0)


def evaluate_msmarco(args):
    from scaling_retriever_amd.utils.metrics import load_and_evaluate
    res = {metric: load_and_evaluate(args.eval_qrel_path, args.eval_run_path, metric) for metric in args.eval_metric}
    os.makedirs(args.out_dir, exist_ok=True)
    with open(os.path.join(args.out_dir, "perf.json"), "w") as fout:
        json.dump(res, fout, indent=4)
    return res


def main(argv=None):
    args = parse_args(argv)
    if args.task_name not in ["evaluate_msmarco", "evaluate_beir"]:
        with open(os.path.join(args.model_name_or_path, "config.json")) as f:
            model_type = json.load(f).get("model_type", "llama")
        assert model_type == "llama", model_type                   # the HIP path implements the Llama family only
        ddp_setup(args)
    if args.task_name == "indexing":
        sparse_index(args)
    elif args.task_name == "retrieval":
        sparse_retrieval(args)
    elif args.task_name == "evaluate_msmarco":
        evaluate_msmarco(args)
    elif args.task_name == "evaluate_beir":                        # eval_sparse.py:188-193
        from scaling_retriever_amd.utils.beir import load_beir
        from scaling_retriever_amd.utils.metrics import evaluate_beir
        _, _, qrels = load_beir(args.beir_dataset_dir, args.beir_dataset, split="test")
        return evaluate_beir(args, qrels)
    else:
        raise NotImplementedError(args.task_name)
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
