#!/usr/bin/env python3
"""Dense eval driver on MI355X: same tasks, flags and artefacts as /root/reference/eval_dense.py
(DenseRetrievalArguments :35-74; tasks write_doc_embeds :158-189, retrieval :190-241,
evaluate_msmarco :138-145,242-243), running the HIP encoder + HIP flat index.

  torchrun --nproc_per_node=8 eval_dense.py --task_name write_doc_embeds --model_name_or_path <lora dir> \
           --corpus_path <tsv> --doc_embed_dir <dir> --eval_batch_size 128 --doc_max_length 192
  python   eval_dense.py --task_name retrieval --model_name_or_path <lora dir> --query_path <tsv> \
           --doc_embed_dir <dir> --out_dir <dir> --top_k 1000
  torchrun --nproc_per_node=8 eval_dense.py --task_name retrieval ...      (doc-sharded: the reference asserts
           world_size == 1, eval_dense.py:191; here each rank scores the shard files it loads, one RCCL gather)
  python   eval_dense.py --task_name evaluate_msmarco --eval_qrel_path q.json --eval_run_path run.json \
           --eval_metric '["mrr_10","recall"]' --out_dir <dir>
"""
import argparse
import ast
import json
import os

import numpy as np
import torch
import torch.distributed as dist
from torch.utils.data import DataLoader
from torch.utils.data.distributed import DistributedSampler

CORPUS_DATASOURCE = {"./data/msmarco-full/full_collection/raw.tsv": "msmarco"}   # constants.py:10-13


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    for name in ["model_name_or_path", "ckpt_path", "model_type", "corpus_path", "doc_embed_dir", "index_dir", "out_dir",
                 "query_path", "eval_run_path", "eval_qrel_path", "eval_metric", "beir_dataset", "beir_dataset_dir",
                 "access_token"]:
        ap.add_argument("--" + name, type=str, default=None)
    ap.add_argument("--task_name", type=str, default="")
    ap.add_argument("--data_source", type=str, default=None, help="msmarco | wiki (default: looked up from corpus_path, else msmarco)")
    ap.add_argument("--is_beir", action="store_true")
    ap.add_argument("--eval_batch_size", type=int, default=128)
    ap.add_argument("--doc_max_length", type=int, default=192)
    ap.add_argument("--query_max_length", type=int, default=64)
    ap.add_argument("--hidden_dim", type=int, default=768)
    ap.add_argument("--top_k", type=int, default=1000)
    ap.add_argument("--local_rank", type=int, default=-1)
    ap.add_argument("--world_size", type=int, default=1)
    ap.add_argument("--chunk_size", type=int, default=2_000_000)
    ap.add_argument("--token_budget", type=int, default=16384,
                    help="real tokens per doc_encode batch (length-bucketed, multi-worker tokenisation); 0 = the reference's "
                         "loader: eval_batch_size passages padded to the longest, one worker (eval_dense.py:171-179)")
    ap.add_argument("--tokenize_workers", type=int, default=4)
    args = ap.parse_args(argv)
    if args.eval_metric:
        args.eval_metric = ast.literal_eval(args.eval_metric)     # the reference uses eval() (eval_dense.py:70)
    return args


def ddp_setup(args):
    """eval_dense.py:29-32.  Works without torchrun too (single process, no process group)."""
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


def _tokenizer(path, access_token=None):
    from transformers import AutoTokenizer
    tok = AutoTokenizer.from_pretrained(path, token=access_token) if access_token else AutoTokenizer.from_pretrained(path)
    tok.padding_side = "left"                                     # eval_dense.py:185,206
    if tok.pad_token is None:
        tok.pad_token = tok.eos_token
    assert tok.pad_token == tok.eos_token                         # eval_dense.py:186,208
    return tok


def write_doc_embeds(args):
    from scaling_retriever_amd.dataset.data_collator import LlamaDenseCollectionCollator
    from scaling_retriever_amd.dataset.dataset import CollectionDataset
    from scaling_retriever_amd.indexer import store_embs
    from scaling_retriever_amd.modeling.llm_encoder import LlamaBiDense
    from scaling_retriever_amd.utils.utils import is_first_worker
    if is_first_worker():
        os.makedirs(args.doc_embed_dir, exist_ok=True)
    tokenizer = _tokenizer(args.model_name_or_path, args.access_token)
    if args.is_beir and args.beir_dataset is not None:            # eval_dense.py:164-169
        from scaling_retriever_amd.dataset.dataset import BeirDataset
        from scaling_retriever_amd.utils.beir import load_beir
        corpus, _, _ = load_beir(args.beir_dataset_dir, args.beir_dataset, split="test")
        dataset = BeirDataset(corpus, information_type="document")
    else:
        source = args.data_source or CORPUS_DATASOURCE.get(args.corpus_path, "msmarco")
        dataset = CollectionDataset(corpus_path=args.corpus_path, data_source=source)
    if args.token_budget > 0:
        from scaling_retriever_amd.dataset.pipeline import TokenBudgetCollectionLoader
        rank = dist.get_rank() if args.world_size > 1 else 0
        loader = TokenBudgetCollectionLoader(dataset, tokenizer, max_length=args.doc_max_length, max_tokens=args.token_budget,
                                             max_seqs=1024, num_workers=args.tokenize_workers, rank=rank,
                                             world_size=args.world_size)
    else:
        sampler = DistributedSampler(dataset, shuffle=False) if args.world_size > 1 else None
        loader = DataLoader(dataset, batch_size=args.eval_batch_size, shuffle=False, num_workers=1, sampler=sampler,
                            collate_fn=LlamaDenseCollectionCollator(tokenizer=tokenizer, max_length=args.doc_max_length))
    model = LlamaBiDense.load_from_lora(args.model_name_or_path, access_token=args.access_token)
    model.to(args.local_rank)
    model.eval()
    # file names carry the rank (= LOCAL_RANK on the single node the reference runs on, eval_dense.py:188)
    store_embs(model=model, collection_loader=loader, local_rank=dist.get_rank() if args.world_size > 1 else args.local_rank,
               index_dir=args.doc_embed_dir, device=args.local_rank, chunk_size=args.chunk_size)
    if args.world_size > 1:
        dist.barrier()


def generate_query_vecs(model, dataloader, device, group_rows=16384):
    """DenseRetriever.generate_query_vecs (eval_dense.py:94-106): no autocast, i.e. the fp32 regime.  The loader's batches
    (--eval_batch_size rows each, 128 by default: ~1 100 query tokens) are encoded group-wise, one engine pass per group
    (LlamaBiDense.encode_batches: every row keeps the positions of its own batch, so the vectors are the ones batch-by-batch
    query_encode calls return, bit for bit), and come back in loader order."""
    from scaling_retriever_amd.indexer import batch_groups, encode_group
    reps, qids = [], []
    for group in batch_groups(dataloader, group_rows):
        with torch.no_grad():
            reps.append(encode_group(model, group, device, which="query_encode"))
        for batch in group:
            qids.extend(batch["ids"])
    return torch.cat(reps), qids


class DenseRetriever:
    """eval_dense.py:88-106."""

    def __init__(self, model, device):
        self.model = model
        self.model.eval()
        self.device = device

    def generate_query_vecs(self, dataloader):
        reps, qids = generate_query_vecs(self.model, dataloader, self.device)
        return reps.cpu().numpy(), qids


class LocalFaissDenseRetriever(DenseRetriever):
    """eval_dense.py:108-135 with the flat index resident in HBM (DenseFlatIndexer over the HIP scorer)."""

    last_run_timeline = None  # write_run: wall-clock stages of the last call (bench.py reports them)
    RUN_PIECES = 4            # write_run: pieces of the query set (search of piece c + 1 beside the writing of piece c)
    RUN_PIECE_SHARES = (9, 8, 7, 4)

    def __init__(self, model, device, index):
        super().__init__(model, device)
        self.index = index

    def index_encoded_data(self, doc_vec_files, doc_id_files):
        # one HBM segment per shard file instead of one concatenated host copy (eval_dense.py:113-121)
        total = 0
        for doc_file, id_file in zip(doc_vec_files, doc_id_files):
            total += self.index.index_data(np.load(doc_file, mmap_mode="r"), np.load(id_file).tolist())
        print("size of doc reps indexed: ", total)

    def get_top_docs(self, dataloader, top_docs):
        query_reps, qids = generate_query_vecs(self.model, dataloader, self.device)      # stays on the device
        top_doc_ids, top_scores = self.index.search_knn(query_reps, top_docs)
        assert len(qids) == len(query_reps), (len(qids), len(query_reps))
        return qids, top_doc_ids, top_scores

    def write_run(self, dataloader, top_docs, path):
        """get_top_docs + the run.json loop of eval_dense.py:225-241 in one go, without materialising a Python object per hit:
        encode, search, and sr_write_run_json over the result arrays.  Returns (number of queries, file size).
        A large query set goes through in pieces: this thread drives the GPU (encode piece c, search piece c) while a worker thread formats
        and writes piece c - 1 (the C call releases the GIL).  Exact results do not depend on how the queries are batched (pieces stay above
        64 queries) and the file is the one-call file byte for byte."""
        import time
        from scaling_retriever_amd.utils.run_file import PiecewiseRunWriter, id_table, write_run_json
        t_start = time.perf_counter()
        batches = list(dataloader)
        sizes = [len(b["ids"]) for b in batches]
        nq = sum(sizes)
        all_qids = [x for b in batches for x in b["ids"]]
        table = self.index.run_table()
        n_pieces = self.RUN_PIECES if nq >= 1024 and table.distinct and id_table(all_qids).distinct else 1
        tl = self.last_run_timeline = {"pieces": []}
        if n_pieces == 1:
            query_reps, qids = generate_query_vecs(self.model, batches, self.device)
            scores, positions = self.index.search_arrays(query_reps, top_docs)
            return nq, write_run_json(path, qids, scores, positions, table)
        # piece boundaries: whole loader batches, as close as they come to whole 256-query tiles (the scorer pads a query set to tiles:
        # 3 x 2 327 queries are 30 of them, 6 980 are 28), shrinking towards the end - only the LAST piece's writing stays on the critical
        # path (60-75 MB into the page cache took 18-90 ms over runs)
        tiles = (nq + 255) // 256
        shares, acc, targets = self.RUN_PIECE_SHARES[:n_pieces], 0.0, []
        for w in shares[:-1]:
            acc += w / sum(shares)
            targets.append(min(nq, 256 * max(1, round(tiles * acc))))
        cum, cuts = 0, [0]
        for bi, n in enumerate(sizes):
            cum += n
            if len(cuts) - 1 < len(targets) and cum >= targets[len(cuts) - 1] and nq - cum > 64 and cum - sum(sizes[:cuts[-1]]) > 64:
                cuts.append(bi + 1)
        cuts.append(len(batches))
        pieces = [batches[cuts[c]:cuts[c + 1]] for c in range(len(cuts) - 1) if cuts[c + 1] > cuts[c]]
        with PiecewiseRunWriter(path) as writer:                       # run.json appears only when its last piece is written
            for c, piece in enumerate(pieces):
                t0 = time.perf_counter()
                reps, qids = generate_query_vecs(self.model, piece, self.device)
                scores, positions = self.index.search_arrays(reps, top_docs)
                t1 = time.perf_counter()
                writer.add(qids, scores, positions, table, None, last=c + 1 == len(pieces))
                tl["pieces"].append({"queries": len(qids), "encode_and_search_ms": round((t1 - t0) * 1e3, 1)})
            t2 = time.perf_counter()
            size = writer.finish()
            tl["waited_for_the_writer_after_the_last_search_ms"] = round((time.perf_counter() - t2) * 1e3, 1)
        tl["total_ms"] = round((time.perf_counter() - t_start) * 1e3, 1)
        return nq, size


def retrieval(args):
    from scaling_retriever_amd.dataset.data_collator import LlamaDenseCollectionCollator
    from scaling_retriever_amd.dataset.dataset import MSMARCOQueryDataset
    from scaling_retriever_amd.distributed import gather_topk, sharded_dense_search
    from scaling_retriever_amd.modeling.llm_encoder import LlamaBiDense
    from scaling_retriever_amd.scoring import DenseIndexHIP, topk_merge
    from scaling_retriever_amd.utils.utils import obtain_doc_vec_dir_files
    rank, world = (dist.get_rank(), dist.get_world_size()) if args.world_size > 1 else (0, 1)
    device = torch.device("cuda", args.local_rank)
    if rank == 0:
        os.makedirs(args.out_dir, exist_ok=True)
    model = LlamaBiDense.load_from_lora(args.model_name_or_path, access_token=args.access_token)
    model.to(device)
    model.eval()
    tokenizer = _tokenizer(args.model_name_or_path, args.access_token)
    if args.is_beir and args.beir_dataset is not None:            # eval_dense.py:211-216
        from scaling_retriever_amd.dataset.dataset import BeirDataset
        from scaling_retriever_amd.utils.beir import load_beir
        _, beir_queries, _ = load_beir(args.beir_dataset_dir, args.beir_dataset, split="test")
        query_dataset = BeirDataset(beir_queries, information_type="query")
    else:
        query_dataset = MSMARCOQueryDataset(args.query_path)
    q_loader = DataLoader(query_dataset, batch_size=args.eval_batch_size, shuffle=False, num_workers=0,
                          collate_fn=LlamaDenseCollectionCollator(tokenizer=tokenizer, max_length=args.query_max_length))
    vec_files, id_files = obtain_doc_vec_dir_files(args.doc_embed_dir)
    # files in plan order (rank-major) get consecutive positions, exactly the order the reference concatenates them in
    # (eval_dense.py:113-121); every retrieval rank takes a round-robin subset of the FILES as HBM segments.
    sizes = [int(np.load(f, mmap_mode="r").shape[0]) for f in id_files]
    offsets = np.concatenate([[0], np.cumsum(sizes)])
    index = DenseIndexHIP(model.hidden_size, device=device)
    if model.hidden_size % 64 == 0:
        index.set_precision("fp32_filtered")          # exact results, ~3x faster for the whole query set (csrc/dense_filter.hip)
    for fi in range(rank, len(vec_files), world):
        index.add_npy_file(vec_files[fi], id_base=int(offsets[fi]))        # mmap -> pinned ring -> async H2D
    q_reps, qids = generate_query_vecs(model, q_loader, device)
    scores, idx = sharded_dense_search(index, q_reps, args.top_k, world)
    if world > 1:
        gs, gi = gather_topk(scores, idx, dst=0)
        if gs is not None:
            scores, idx = topk_merge(gs, gi)
    if rank == 0:
        # run.json (eval_dense.py:225-241) straight from the result arrays: the bytes json.dump of the reference's nested dict gives
        from scaling_retriever_amd.utils.run_file import to_host, write_run_json
        doc_ids = np.concatenate([np.load(f) for f in id_files])
        write_run_json(os.path.join(args.out_dir, "run.json"), qids, to_host(scores), to_host(idx), doc_ids)
    if world > 1:
        dist.barrier()


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
        ddp_setup(args)
        print("world_size = {}, local_rank = {}".format(args.world_size, args.local_rank))
    if args.task_name == "write_doc_embeds":
        write_doc_embeds(args)
    elif args.task_name == "retrieval":
        retrieval(args)
    elif args.task_name == "evaluate_msmarco":
        return evaluate_msmarco(args)
    elif args.task_name == "evaluate_beir":                       # eval_dense.py:244-249
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
