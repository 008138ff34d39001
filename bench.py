#!/usr/bin/env python3
"""Contract benchmark: MSMARCO-Dev queries/s end-to-end on the Lion-DS-1B dense configuration
(BASELINE.json configs[1]; configs[3] for --gpus > 1).

One step = one pass of the hot path over the whole Dev query set: encode 6 980 synthetic queries with
the HIP LlamaBiDense encoder (1B dims, random weights), score them against the doc-sharded fp32
embedding matrix resident in HBM (8 841 823 x 2048, synthetic), fused top-1000, and - for N > 1 - an all-reduce(min)
of nq floats (the threshold exchange), one RCCL gather of the per-shard top-k + merge on rank 0.  value = queries / s
over the whole job.

Precision: queries are encoded in the encoder's fp32 regime, as the reference does (eval_dense.py:94-106: no
autocast) - fp16-plane GEMMs carrying the error of an fp32 GEMM - documents in the bf16-autocast regime
(indexer.py:46-52); scores are exact fp32: a certified fp16 upper-bound filter + exact re-score, bit-identical to the
exact fp32 MFMA kernel (asserted on the whole problem in every run: `parity`).

Also reported on the same JSON line: `roofline` for the dominant kernel (dense_split_kernel<true>, the filter's
upper-bound pass, 16-bit MFMA bound; durations from HIP events recorded around every launch inside the timed region),
`cpu_baseline` (faiss's flat-IP algorithm - host BLAS sgemm blocks + a heap per query - on the host cores, bounded
sample, rank 0 at N = 1 only), and the other stages / configs of the headline metric:
  `exact_kernel_mode`  the same step with every product on the fp32 MFMA pipe: the data-independent floor;
  `filter_robustness`  the search stage on anisotropic / near-duplicate corpora and queries drawn near documents (full shape),
                       each compared with the exact kernel on all queries, with the certified / re-done query counts;
  `shard_1of8`         BASELINE.json configs[3] on one GPU: one of 8 doc shards with and without the threshold exchange;
  `drop_in`            the reference's own entry points at the full shape, inputs as its DataLoader yields them (55 batches of 128
                       queries, string ids): generate_query_vecs, DenseFlatIndexer.search_knn (list of lists of db ids),
                       get_top_docs, and the retrieval task including run.json (eval_dense.py:94-135,225-241);
  `encode`             passages/s of the corpus-encode task: >= 100 000 synthetic passages through store_embs (token-budget
                       batches -> doc_encode under autocast -> D2H -> embs_*.npy / ids_*.npy / plan.json), MFMA roofline;
                       `padded_batch_128_mode`: the same through the reference's loader shape (batches of 128 padded passages);
  `sparse`             BASELINE.json configs[2] (N = 1 only): queries/s of sr_sparse_search on the MSMARCO-shaped synthetic
                       inverted index, all CPU-scored queries compared bit for bit with the oracle, the 32-thread CPU baseline,
                       `bounds` (the kernel's VALU / L2 / LDS floors from work counted on the device), `roofline.traffic` (PMC),
                       `drop_in` (SparseRetrieval.retrieve incl. run.json) and `sparse_sweep` (L0_d x L0_q x two distributions);
  `small_batch`        1 and 16 queries (the HBM-bound regime), each also in the batch-invariant mode;
  `config5_8b`         BASELINE.json configs[4]: one GPU's share at Lion-DS-8B dims.

  python bench.py --gpus 1 --steps 3 --warmup 1
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus 8 --steps 3 --warmup 1
"""
import argparse
import contextlib
import ctypes
import json
import os
import sys
import time

os.environ.setdefault("TQDM_DISABLE", "1")      # stdout carries ONE JSON line; progress bars would drown stderr

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

LION_1B = dict(vocab_size=128256, hidden_size=2048, intermediate_size=8192, num_hidden_layers=16,
               num_attention_heads=32, num_key_value_heads=8, head_dim=64, rms_norm_eps=1e-5, rope_theta=500000.0,
               tie_word_embeddings=True,
               rope_scaling={"rope_type": "llama3", "factor": 32.0, "low_freq_factor": 1.0, "high_freq_factor": 4.0,
                             "original_max_position_embeddings": 8192})
PEAK_F32_MFMA_TF = 157.3     # MI355X_MICROARCH.md: fp32-in MFMA = fp32 vector peak
PEAK_BF16_MFMA_TF = 2500.0   # dense bf16 MFMA peak
PEAK_HBM_GBPS = 8000.0       # HBM3E spec peak (about 6300 GB/s is what a streaming copy reaches)
PEAK_L2_GBPS = 34500.0       # MI355X_MICROARCH.md 'L2 (per XCD)': 4 MiB per XCD, ~34.5 TB/s aggregate
PEAK_VALU_F32_OPS = 256 * 4 * 16 * 2 * 2.4e9     # packed fp32 lane-operations per second (a multiply or an add each): 78.6e12
FLOP_PER_TOKEN_1B = 1.946e9  # 2 x linear params of the 1B body


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def _get(d, *path):
    for k in path:
        if not isinstance(d, dict) or d.get(k) is None:
            return None
        d = d[k]
    return d


def _finite(o):
    """Non-finite floats become null: the contract line is strict JSON (no NaN / Infinity tokens)."""
    if isinstance(o, float):
        return o if o == o and abs(o) != float("inf") else None
    if isinstance(o, dict):
        return {k: _finite(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_finite(v) for v in o]
    return o


CONTRACT_LINE_MAX = 4096


def contract_line(res):
    """The ONE stdout line the driver parses: the contract fields, `roofline` and `cpu_baseline` of the headline, and a few scalars
    per leg.  Everything else (`res`: notes, sweeps, timelines) goes to stderr and to gpurun_out/bench_detail.json."""
    r, c = res.get("roofline") or {}, res.get("cpu_baseline") or {}
    cfg = res["config"]
    sp, enc, c5, di = res.get("sparse") or {}, res.get("encode") or {}, res.get("config5_8b") or {}, res.get("drop_in") or {}
    sweep = _get(sp, "sparse_sweep", "rows") or []
    line = {k: res[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                "vs_baseline", "dtype", "data")}
    line["config"] = {"workload": f"Lion-DS-1B dense, BASELINE configs[{1 if res['n_gpus'] == 1 else 3}]: {cfg['n_queries']} Dev queries encoded "
                                  f"(HIP LlamaBiDense 1B dims, {cfg['layers']} layers, fp32 regime) + exact fp32 top-{cfg['topk']} over "
                                  f"{cfg['n_docs']} x {cfg['hidden']} embeddings resident in HBM",
                      "n_docs": cfg["n_docs"], "n_queries": cfg["n_queries"], "hidden": cfg["hidden"], "topk": cfg["topk"],
                      "layers": cfg["layers"], "parallelism": f"doc-shard x{res['n_gpus']}", "ranks": cfg.get("ranks")}
    line["roofline"] = {"kernel": (r.get("kernel") or "").split(" (")[0], "bound": r.get("bound"), "achieved": r.get("achieved"),
                        "peak": r.get("peak"), "unit": "TFLOP/s", "frac": r.get("frac"), "traffic": r.get("traffic"),
                        "launches": r.get("launches"), "avg_launch_ms": r.get("avg_launch_ms"), "flop_per_launch": r.get("flop_per_launch"),
                        "kernel_share_of_step": r.get("kernel_share_of_step")} if r else None
    line["cpu_baseline"] = {"value": c.get("value"), "unit": c.get("unit"), "cores": c.get("cores"), "kind": c.get("kind"),
                            "threads": c.get("cores"), "host_cpu": c.get("host_cpu"),
                            "sample": (c.get("sample_short") or (c.get("sample") or "")[:160])} if c else None
    line["parity"] = "bit-identical to the exact fp32 kernel, all queries" if res.get("parity") and "bit-identical" in res["parity"] else res.get("parity")
    line["breakdown"] = {"query_encode_ms": _get(res, "breakdown", "query_encode_ms"), "search_ms": _get(res, "breakdown", "search_ms"),
                         "query_encode_frac": (round(_get(res, "breakdown", "query_encode_mfma_TFLOPs") / PEAK_BF16_MFMA_TF, 4)
                                               if _get(res, "breakdown", "query_encode_mfma_TFLOPs") else None)}
    line["exact_kernel"] = {"qps": _get(res, "exact_kernel_mode", "value"), "frac_fp32_mfma": _get(res, "exact_kernel_mode", "roofline", "frac")}
    line["drop_in"] = {"retrieval_qps": _get(di, "retrieval_task_with_run_json", "queries_per_s"), "search_knn_qps": _get(di, "search_knn", "queries_per_s")}
    line["encode"] = {"passages_per_s": enc.get("value"), "frac": _get(enc, "roofline", "frac"), "sample_passages": enc.get("sample_passages"),
                      "padded_128_passages_per_s": _get(enc, "padded_batch_128_mode", "passages_per_s")} if enc else None
    if sp and sp.get("n_gpus", 1) > 1:                    # the doc-sharded leg of a multi-GPU run
        line["sparse"] = {"qps": sp["value"], "ms_per_pass": sp["ms_per_pass"], "n_gpus": sp["n_gpus"], "sharded": True,
                          "redone_exact": sum(sp["queries_redone_by_the_exact_kernels"]), "oracle_bit_exact_queries": sp["oracle_bit_exact_queries"]}
        sp = {}
    else:
        line["sparse"] = None
    if sp:
      line["sparse"] = {"qps": sp.get("value"), "ms_per_pass": sp.get("ms_per_pass"), "kernel": _get(sp, "roofline", "kernel"),
                      "bound": _get(sp, "roofline", "bound"), "achieved_GBps": _get(sp, "roofline", "achieved"),
                      "frac": _get(sp, "roofline", "frac"), "traffic": _get(sp, "roofline", "traffic"),
                      "kernel_ms_per_pass": _get(sp, "roofline", "kernel_ms_per_pass"),
                      "kernel_over_sum_of_floors": _get(sp, "bounds", "kernel_over_sum_of_floors"),
                      "redone_exact": _get(sp, "path", "queries_redone_by_the_exact_kernels_per_pass"),
                      "exact_kernels_qps": _get(sp, "exact_kernels", "queries_per_s"),
                      "oracle_bit_exact_queries": sp.get("oracle_bit_exact_queries"),
                      "cpu_qps": _get(sp, "cpu_baseline", "value"), "cpu_cores": _get(sp, "cpu_baseline", "cores"),
                      "retrieve_qps": _get(sp, "drop_in", "retrieve", "queries_per_s"),
                      "index_build_postings_per_s": _get(sp, "index_build", "postings_per_s"),
                      "index_build_frac": _get(sp, "index_build", "roofline", "frac"),
                      "index_passages_per_s": _get(sp, "sparse_index", "passages_per_s"),
                      "index_frac": _get(sp, "sparse_index", "roofline", "frac"),
                      "index_sample_passages": _get(sp, "sparse_index", "sample_passages"),
                      "sweep": {"cells": len(sweep), "min_qps": min((x["queries_per_s"] for x in sweep), default=None),
                                "max_L0_q": max((x["L0_q"] for x in sweep), default=None),
                                "redone_exact": sum(x["queries_redone_by_the_exact_kernels"] for x in sweep),
                                "all_bit_exact": all(x.get("queries_bit_exact_vs_oracle", 0) >= 64 for x in sweep)} if sweep else None}
    line["config5_8b"] = {"encode_passages_per_s": _get(c5, "encode", "value"), "encode_frac": _get(c5, "encode", "roofline", "frac"),
                          "score_shard_qps": _get(c5, "score_shard_filtered", "queries_per_s")} if c5 else None
    line["small_batch"] = [{"nq": x["nq"], "frac_hbm": x["frac"]} for x in (res.get("small_batch") or [])] or None
    line["detail"] = "gpurun_out/bench_detail.json"
    line = _finite(line)
    text = json.dumps(line, allow_nan=False, separators=(",", ":"))
    for optional in ("small_batch", "config5_8b", "exact_kernel", "drop_in", "breakdown"):     # never lose a finished run to a long line: shed the optional legs first
        if len(text) < CONTRACT_LINE_MAX:
            break
        line.pop(optional, None)
        text = json.dumps(line, allow_nan=False, separators=(",", ":"))
    assert len(text) < 2 * CONTRACT_LINE_MAX and "Infinity" not in text and "NaN" not in text, len(text)
    return text


def emit(res):
    """Full record -> stderr + side file; the compact contract line -> the LAST line of stdout."""
    full = json.dumps(_finite(res), allow_nan=False)
    log("[bench detail]", full)
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "bench_detail.json"), "w") as f:
            f.write(full + "\n")
    except OSError as e:
        log("[bench detail] side file not written:", e)
    sys.stdout.flush()
    print(contract_line(res), flush=True)


def random_weights(cfg, device, seed):
    """Random-init weights of the named architecture, generated on the device in bf16 (no checkpoints offline)."""
    g = torch.Generator(device=device).manual_seed(seed)
    H, I, V = cfg["hidden_size"], cfg["intermediate_size"], cfg["vocab_size"]
    nq, nkv = cfg["num_attention_heads"] * cfg["head_dim"], cfg["num_key_value_heads"] * cfg["head_dim"]

    def lin(o, i):
        return (torch.randn((o, i), device=device, generator=g, dtype=torch.float32) * 0.02).bfloat16()
    w = {"model.embed_tokens.weight": lin(V, H), "model.norm.weight": torch.ones(H, device=device)}
    if not cfg.get("tie_word_embeddings", False):
        w["lm_head.weight"] = lin(V, H)
    for li in range(cfg["num_hidden_layers"]):
        p = f"model.layers.{li}."
        w[p + "self_attn.q_proj.weight"] = lin(nq, H)
        w[p + "self_attn.k_proj.weight"] = lin(nkv, H)
        w[p + "self_attn.v_proj.weight"] = lin(nkv, H)
        w[p + "self_attn.o_proj.weight"] = lin(H, nq)
        w[p + "mlp.gate_proj.weight"] = lin(I, H)
        w[p + "mlp.up_proj.weight"] = lin(I, H)
        w[p + "mlp.down_proj.weight"] = lin(H, I)
        w[p + "input_layernorm.weight"] = torch.ones(H, device=device)
        w[p + "post_attention_layernorm.weight"] = torch.ones(H, device=device)
    return w


def synth_batches(n, batch, mu, sigma, lo, hi, vocab, seed, device, rows=None):
    """Left-padded, pad-to-longest batches like the reference's collator + padding_side='left'
    (data_collator.py:184-186, eval_dense.py:185,206).  Row r's tokens depend only on (seed, r), so a rank that
    encodes rows [rows[0], rows[1]) of the set sees exactly the tokens a single-GPU run sees for those rows."""
    lens = np.clip(np.round(np.random.default_rng(seed).lognormal(mu, sigma, size=n)), lo, hi).astype(np.int64)
    r0, r1 = rows if rows is not None else (0, n)
    out = []
    for b0 in range(r0, r1, batch):
        ls = lens[b0:min(b0 + batch, r1)]
        L = int(ls.max())
        ids = np.full((len(ls), L), vocab - 1, dtype=np.int64)
        mask = np.zeros((len(ls), L), dtype=np.int64)
        for r, l in enumerate(ls):
            ids[r, L - l:] = np.random.default_rng((seed, b0 + r)).integers(0, vocab - 1, size=l)
            mask[r, L - l:] = 1
        out.append((torch.from_numpy(ids).to(device), torch.from_numpy(mask).to(device)))
    return out, lens


class _TimedEncoder:
    """Stands where store_embs expects the model: forwards doc_encode and brackets every call with events on torch's
    current stream (the stream the C ABI launches on), so the GPU share of the pass is measured, not guessed."""

    def __init__(self, model):
        self.model = model
        self.events = []
        self.tokens = 0
        self.sq_tokens = 0.0

    def doc_encode(self, **inputs):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        out = self.model.doc_encode(**inputs)
        b.record()
        self.events.append((a, b))
        return out

    def gpu_seconds(self):
        torch.cuda.synchronize()
        return sum(a.elapsed_time(b) for a, b in self.events) * 1e-3


def synth_token_chunks(n, mu, sigma, lo, hi, vocab, seed, rows, chunk=8192):
    """Already tokenised passages for TokenBudgetCollectionLoader (no tokenizer offline): (ids, flat int32 tokens, lengths)
    chunks; lengths ~ clip(lognormal(mu, sigma)), ids uniform."""
    lens = np.clip(np.round(np.random.default_rng(seed).lognormal(mu, sigma, size=n)), lo, hi).astype(np.int32)
    r0, r1 = rows
    chunks = []
    for c0 in range(r0, r1, chunk):
        ls = lens[c0:min(c0 + chunk, r1)]
        flat = np.random.default_rng((seed, c0)).integers(0, vocab - 1, size=int(ls.sum()), dtype=np.int32)
        chunks.append((list(range(c0, c0 + len(ls))), flat, ls))
    return chunks, lens[r0:r1]


def encode_leg(args, cfg, model, device, rank, world, share_gpu, flop_per_token=FLOP_PER_TOKEN_1B, layers_ref=16):
    import shutil
    import tempfile
    from scaling_retriever_amd.dataset.pipeline import TokenBudgetCollectionLoader
    from scaling_retriever_amd.distributed import query_slice
    from scaling_retriever_amd.indexer import store_embs
    H, L = cfg["hidden_size"], cfg["num_hidden_layers"]
    rows = query_slice(args.encode_passages, rank, world)           # contiguous block per rank
    chunks, lens = synth_token_chunks(args.encode_passages, 4.25, 0.35, 8, 192, cfg["vocab_size"], 3, rows)
    tmp = tempfile.mkdtemp(prefix="sr_bench_embs_")
    try:
        def loader():
            return TokenBudgetCollectionLoader(tokenized=chunks, max_length=192, max_tokens=args.token_budget, max_seqs=1024,
                                               window=32768, pad_token_id=cfg["vocab_size"] - 1, padding_side="left")
        warm = TokenBudgetCollectionLoader(tokenized=chunks[:1], max_length=192, max_tokens=args.token_budget, max_seqs=1024,
                                           pad_token_id=cfg["vocab_size"] - 1, padding_side="left")
        for i_, b_ in enumerate(warm):                              # warm-up: two batches
            with torch.autocast("cuda", dtype=torch.bfloat16):
                model.doc_encode(input_ids=b_["input_ids"].to(device), attention_mask=b_["attention_mask"].to(device))
            if i_ >= 1:
                break
        timed = _TimedEncoder(model)
        n_batches = sum(1 for _ in loader())
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        te = time.perf_counter()
        with contextlib.redirect_stdout(sys.stderr):              # store_embs prints its plan like the reference does
            store_embs(timed, loader(), rank, tmp, device, chunk_size=65536)
        torch.cuda.synchronize()
        te = time.perf_counter() - te
        gpu_s = timed.gpu_seconds()
        if world > 1:
            t = torch.tensor([te, gpu_s], dtype=torch.float64, device="cpu" if share_gpu else device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            te, gpu_s = float(t[0].item()), float(t[1].item())
        written = sum(np.load(os.path.join(tmp, f), mmap_mode="r").shape[0] for f in os.listdir(tmp) if f.startswith("embs_"))
        assert written == len(lens), (written, len(lens))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    # ---- the reference's own loader shape (scripts/eval_dense.sh:11-16, eval_dense.py:171-179): 128 consecutive passages per
    #      doc_encode call, padded to the batch's longest (pads are not computed here either) - through store_embs as well
    padded = None
    if world == 1 and not getattr(args, "no_padded_mode", False):
        n_p = min(len(lens), 32768)
        flat_all = np.concatenate([c[1] for c in chunks])
        offs = np.concatenate([[0], np.cumsum(lens)])

        class _Padded128:
            batch_size = 128

            def __len__(self):
                return (n_p + 127) // 128

            def __iter__(self):
                for b0 in range(0, n_p, 128):
                    ls = lens[b0:b0 + 128]
                    L_ = int(ls.max())
                    ids_ = np.full((len(ls), L_), cfg["vocab_size"] - 1, dtype=np.int64)
                    mask_ = np.zeros((len(ls), L_), dtype=np.int64)
                    for r_, l_ in enumerate(ls):
                        ids_[r_, L_ - l_:] = flat_all[offs[b0 + r_]:offs[b0 + r_ + 1]]
                        mask_[r_, L_ - l_:] = 1
                    yield {"input_ids": torch.from_numpy(ids_), "attention_mask": torch.from_numpy(mask_), "ids": list(range(rows[0] + b0, rows[0] + b0 + len(ls)))}
        tmp2 = tempfile.mkdtemp(prefix="sr_bench_embs128_")
        try:
            torch.cuda.synchronize()
            tp = time.perf_counter()
            with contextlib.redirect_stdout(sys.stderr):
                store_embs(model, _Padded128(), rank, tmp2, device, chunk_size=65536)
            torch.cuda.synchronize()
            tp = time.perf_counter() - tp
        finally:
            shutil.rmtree(tmp2, ignore_errors=True)
        padded = {"passages_per_s": round(n_p / tp, 1), "sample_passages": int(n_p), "wall_s": round(tp, 3),
                  "note": "the reference's loader shape: batches of 128 consecutive passages padded to the batch's longest (host-side collation in "
                          "this process, ~9 600 real tokens per batch); store_embs encodes four such batches per engine pass (rows bit-identical to "
                          "batch-by-batch doc_encode calls); the token-budget loader above is the drivers' default"}
    # whole-job figures: every rank encoded 1/W of the sample, times are the max over ranks
    all_lens = np.clip(np.round(np.random.default_rng(3).lognormal(4.25, 0.35, size=args.encode_passages)), 8, 192)
    tokens = float(all_lens.sum())
    flop = tokens * flop_per_token * L / layers_ref + 4.0 * float((all_lens.astype(np.float64) ** 2).sum()) * H * L
    ach, ach_gpu = flop / te / 1e12 / world, flop / gpu_s / 1e12 / world
    return {"value": round(args.encode_passages / te, 1), "unit": "passages/s (whole job, through store_embs: encode + D2H + .npy files)",
            "passages_per_s_per_gpu": round(args.encode_passages / te / world, 1), "sample_passages": int(args.encode_passages),
            "padded_batch_128_mode": padded,
            "mean_tokens_per_passage": round(float(all_lens.mean()), 1), "token_budget": args.token_budget, "batches_per_rank": n_batches,
            "dtype": "bf16 GEMM inputs / fp32 accumulate (autocast regime, indexer.py:46-52)",
            "wall_s": round(te, 3), "gpu_encode_s": round(gpu_s, 3),
            "roofline": {"kernel": "gemm_bf16_kernel (bf16 MFMA 16x16x32, 256 x 256 tiles) - the layer GEMMs are 88 % of the pass",
                         "bound": "mfma", "achieved": round(ach, 1), "peak": PEAK_BF16_MFMA_TF, "unit": "TFLOP/s", "frac": round(ach / PEAK_BF16_MFMA_TF, 4),
                         "achieved_gpu_time_only": round(ach_gpu, 1), "frac_gpu_time_only": round(ach_gpu / PEAK_BF16_MFMA_TF, 4),
                         "traffic": None,
                         "note": "achieved = algorithmic FLOP of the real (non-pad) tokens (2 x linear params + attention 4 S H per layer) / wall time "
                                 "of the whole store_embs pass per GPU; gpu_time_only divides by the summed doc_encode durations (HIP events)"}}


def sparse_leg(args, device):
    """BASELINE.json configs[2]: one MI355X vs the 32-thread CPU shape the reference names (README.md:89-94)."""
    import synth
    from scaling_retriever_amd import _lib
    from scaling_retriever_amd.scoring import SparseIndexHIP
    lib = _lib.load()
    V, N, L0_d, L0_q, nq, k = 128256, 8_841_823, 128, 32, 6980, 1000
    t0 = time.time()
    indptr, doc_ids, vals, _ = synth.build_index(V, N, L0_d, device, 3)
    q_indptr, q_cols, q_vals = synth.build_queries(V, nq, L0_q, device, 4)
    idx = SparseIndexHIP(indptr, doc_ids, vals, N, device=device)
    torch.cuda.synchronize()
    nnz = int(doc_ids.numel())
    log(f"[sparse] {nnz} postings ({nnz * 8 / 1e9:.2f} GB) built in {time.time() - t0:.1f}s")
    lens = indptr[1:] - indptr[:-1]
    touched = lens[q_cols.long()].reshape(nq, L0_q).sum(1).double()
    # bytes of index a batch of queries needs at least once: the posting lists of its DISTINCT terms.  The exact kernels walk the
    # query set in batches of 1024 (round 4's figure), the certified scorer takes it in one batch
    unique_bytes_1024 = 0.0
    for qb in range(0, nq, 1024):
        terms = torch.unique(q_cols[qb * L0_q:min(nq, qb + 1024) * L0_q].long())
        unique_bytes_1024 += 8.0 * float(lens[terms].sum().item())
    unique_bytes_all = 8.0 * float(lens[torch.unique(q_cols.long())].sum().item())
    def timed(steps=3):
        idx.search(q_indptr, q_cols, q_vals, k)
        torch.cuda.synchronize()
        _lib.check(lib.sr_sparse_index_profile(idx._h, 1))
        ts = time.perf_counter()
        for _ in range(steps):
            r = idx.search(q_indptr, q_cols, q_vals, k)
        torch.cuda.synchronize()
        dt_ = (time.perf_counter() - ts) / steps
        n_l, ms, by = ctypes.c_int64(0), ctypes.c_double(0), ctypes.c_double(0)
        _lib.check(lib.sr_sparse_index_profile_read(idx._h, ctypes.byref(n_l), ctypes.byref(ms), ctypes.byref(by)))
        _lib.check(lib.sr_sparse_index_profile(idx._h, 0))
        return r, dt_, ms.value * 1e-3 / steps, int(n_l.value) // steps

    cs0 = idx.cert_stats()
    (s, i, c), dt, kernel_s, launches = timed()
    cs1 = idx.cert_stats()
    certified = cs1["present"] == 1 and cs1["searches"] > cs0["searches"]
    redone = (cs1["redone_exact"] - cs0["redone_exact"]) // max(1, cs1["searches"] - cs0["searches"])
    # the exact kernels alone on the same queries (dev switch): the product path must return their bits
    exact = None
    if certified:
        old = {kk: os.environ.get(kk) for kk in ("SR_DEV_SWITCHES", "SR_SPARSE_CERT_SEARCH")}
        os.environ["SR_DEV_SWITCHES"], os.environ["SR_SPARSE_CERT_SEARCH"] = "1", "0"
        try:
            (s0, i0, c0), dt0, kernel0_s, launches0 = timed(steps=2)
        finally:
            for kk, vv in old.items():
                if vv is None:
                    os.environ.pop(kk, None)
                else:
                    os.environ[kk] = vv
        assert torch.equal(s, s0) and torch.equal(i, i0) and torch.equal(c, c0), "certified scorer and exact kernels disagree"
        exact = {"queries_per_s": round(nq / dt0, 1), "ms_per_pass": round(dt0 * 1e3, 1), "kernel_ms_per_pass": round(kernel0_s * 1e3, 1),
                 "launches_per_pass": launches0, "kernels": "sparse_block_kernel / sparse_score_kernel (exact term-serial chain for every (query, doc))",
                 "same_bits_as_the_product_path": True}
        del s0, i0, c0
    unique_bytes = unique_bytes_all if certified else unique_bytes_1024
    hbm_gbps = unique_bytes / kernel_s / 1e9
    T = cs1["dense_terms"] if certified else 0
    if certified:
        # work of cert_score_kernel per pass, from the index and the query set themselves
        nq_pad, n_tiles = (nq + 31) // 32 * 32, cs1["doc_tiles"]
        flop = 2.0 * nq_pad * T * n_tiles * 1024                           # fp16 MFMA product of the heavy terms, every (query, doc)
        a_bytes = (nq_pad // 32) * float(n_tiles) * 1024 * T * 2           # the [docs x T] operand is streamed once per block of 32 queries
        order = torch.argsort(lens, descending=True)
        is_rare = torch.ones(V, dtype=torch.bool, device=device)
        is_rare[order[:T]] = False
        rare_adds = float((lens * is_rare)[q_cols.long()].double().sum().item())   # one fixed-point LDS atomic per posting of a rare query term
        lds_peak, lds_src = None, None
        try:
            with open(os.path.join(ROOT, "profiles", "r02_lds_rmw.json")) as f:
                mb = json.load(f)
            lds_peak = max(v for kk, v in mb.items() if kk.startswith("random_b32_wg"))
            lds_src = "profiles/r02_lds_rmw.json (tools/micro/lds_rmw.hip: random read-modify-writes on a 32 KB LDS tile; the atomics' own peak was not measured)"
        except Exception:
            pass
        bounds = {
            "mfma_heavy_terms": {"bound": "mfma", "achieved": round(flop / kernel_s / 1e12, 1), "peak": PEAK_BF16_MFMA_TF, "unit": "TFLOP/s (fp16 in, fp32 accumulate)",
                                 "frac": round(flop / kernel_s / 1e12 / PEAK_BF16_MFMA_TF, 4), "flop_per_pass": flop,
                                 "floor_ms_per_pass": round(flop / (PEAK_BF16_MFMA_TF * 1e12) * 1e3, 1)},
            "l2_matrix_operand": {"bound": "l2", "achieved": round(a_bytes / kernel_s / 1e9, 1), "peak": PEAK_L2_GBPS, "unit": "GB/s out of L2 (MI355X_MICROARCH.md)",
                                  "frac": round(a_bytes / kernel_s / 1e9 / PEAK_L2_GBPS, 4), "bytes_per_pass": a_bytes,
                                  "floor_ms_per_pass": round(a_bytes / (PEAK_L2_GBPS * 1e9) * 1e3, 1),
                                  "note": "32 queries per workgroup (what 128 KB of 16-bit LDS slots for two 1 024-doc tiles allow): one MFMA per 1 KB fragment"},
            "lds_rare_postings": {"bound": "lds", "achieved": rare_adds / kernel_s, "peak": lds_peak, "unit": "LDS atomic adds/s, chip-wide",
                                  "frac": round(rare_adds / kernel_s / lds_peak, 4) if lds_peak else None, "adds_per_pass": rare_adds, "peak_source": lds_src,
                                  "floor_ms_per_pass": round(rare_adds / lds_peak * 1e3, 1) if lds_peak else None},
        }
        fl_ = [bounds[b_]["floor_ms_per_pass"] for b_ in bounds if bounds[b_]["floor_ms_per_pass"]]
        bounds["sum_of_floors_ms"] = round(sum(fl_), 1)
        bounds["kernel_over_sum_of_floors"] = round(kernel_s * 1e3 / max(sum(fl_), 1e-9), 2)
        bounds["note"] = ("the matrix waves (MFMA + operand stream) and the scatter waves (LDS atomics) of a workgroup work side by side on different "
                          "tiles, so the floors overlap rather than add; their sum is the conservative yardstick")
    else:
        bounds = None
    # fabric traffic of the dominant kernel from the committed PMC passes (rocprofv3 cannot run inside this process)
    sp_traffic, sp_traffic_src = None, None
    try:
        import hashlib
        with open(os.path.join(ROOT, "profiles", "r06_pmc_sparse_traffic.json")) as f:
            pm = json.load(f)
        with open(os.path.join(ROOT, pm["kernel_source"]["file"]), "rb") as f:
            same = hashlib.sha256(f.read()).hexdigest() == pm["kernel_source"]["sha256"]
        sh = pm["shape"]
        if same and (sh["V"], sh["N"], sh["L0_d"], sh["L0_q"], sh["nq"]) == (V, N, L0_d, L0_q, nq):
            sp_traffic = int(pm["traffic_bytes_per_pass"])
            sp_traffic_src = "profiles/r06_pmc_sparse_traffic.json (" + pm.get("how", "") + ")"
    except Exception:
        sp_traffic = None
    out = {"metric": "sparse inverted-index queries/s (index resident in HBM, top-%d)" % k, "value": round(nq / dt, 1), "unit": "queries/s",
           "ms_per_pass": round(dt * 1e3, 1), "dtype": "f32", "data": "synthetic",
           "config": {"workload": "Lion-SP-1B sparse scoring (BASELINE.json configs[2]), synthetic Zipf(1.0) index", "V": V, "N": N, "L0_d": L0_d,
                      "L0_q": L0_q, "nq": nq, "k": k, "postings": nnz, "mean_postings_touched_per_query": float(touched.mean().item())},
           "path": ({"scorer": "certified two-stage (csrc/sparse_cert.hip): fp16 MFMA + fixed-point LDS atomics -> certificate -> exact fp32 chain of the candidates",
                     "heavy_terms_on_the_matrix_pipe": T, "queries_redone_by_the_exact_kernels_per_pass": redone} if certified else
                    {"scorer": "exact kernels only (this index / k has no certified scorer)"}),
           "roofline": {"kernel": "cert_score_kernel" if certified else ("sparse_block_kernel" if idx.block_stats()["block_calls"] else "sparse_score_kernel"),
                        "bound": "hbm", "achieved": round(hbm_gbps, 1), "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                        "frac": round(hbm_gbps / PEAK_HBM_GBPS, 4), "traffic": sp_traffic, "traffic_source": sp_traffic_src,
                        "traffic_unit": "bytes beyond L2 per pass of the whole query set (the algorithmic figure beside it: unique_index_bytes_per_pass)",
                        "launches": launches,
                        "kernel_ms_per_pass": round(kernel_s * 1e3, 1), "unique_index_bytes_per_pass": unique_bytes,
                        "unique_index_bytes_per_pass_in_batches_of_1024": unique_bytes_1024,
                        "frac_on_the_batches_of_1024_figure": round(unique_bytes_1024 / kernel_s / 1e9 / PEAK_HBM_GBPS, 4),
                        "hbm_floor_ms_per_pass": round(unique_bytes / (PEAK_HBM_GBPS * 1e9) * 1e3, 2),
                        "other_kernels_ms_per_pass": round((dt - kernel_s) * 1e3, 1),
                        "note": "SURVEY.md 8(d) convention: achieved = posting bytes the query batches need at least once (lists of their distinct terms, 8 B per "
                                "posting) / time of the dominant kernel.  The kernel is not HBM-bound; `bounds` prices what it does: the MFMA product of "
                                "the heavy terms, the operand bytes it pulls out of L2, the LDS atomics of the other terms' postings.  "
                                "other_kernels_ms_per_pass: plan, top-k compaction of every launch, the sort of k + 1024 keys per query, certificate, "
                                "exact re-score, final top-k"},
           "bounds": bounds, "exact_kernels": exact}
    from oracle import scoring as SC
    h_indptr, h_ids, h_vals = indptr.cpu().numpy(), doc_ids.cpu().numpy(), vals.cpu().numpy()
    nqc = max(args.sparse_cpu_queries, 4)
    check = nqc          # every CPU-scored query is compared
    hq_indptr = q_indptr[:nqc + 1].cpu().numpy()
    hq_cols, hq_vals = q_cols[:nqc * L0_q].cpu().numpy(), q_vals[:nqc * L0_q].cpu().numpy()
    cores = os.cpu_count()
    runs = []
    for qt, it in ((4, 8), (4, max(1, cores // 4))):   # first = the 32-thread shape BASELINE.json names; then all cores
        if (qt, it) in [(r["q_threads"], r["inner_threads"]) for r in runs]:
            continue
        tc = time.perf_counter()
        oi, os_, oc = SC.sparse_retrieve_c(h_indptr, h_ids, h_vals, hq_indptr, hq_cols, hq_vals, k, 0.0, N, q_threads=qt, inner_threads=it)
        tc = time.perf_counter() - tc
        runs.append({"q_threads": qt, "inner_threads": it, "qps": nqc / tc, "seconds": tc})
        log("[sparse] cpu:", runs[-1])
    gi, gs, gc = i[:check].cpu().numpy(), s[:check].cpu().numpy(), c[:check].cpu().numpy()
    for q in range(check):
        assert gc[q] == oc[q], (q, gc[q], oc[q])
        assert np.array_equal(gi[q, :gc[q]], oi[q, :oc[q]]) and np.array_equal(gs[q, :gc[q]], os_[q, :oc[q]]), q
    out["oracle_bit_exact_queries"] = int(check)
    out["parity"] = f"{check} queries bit-exact (ids and fp32 scores) vs the oracle's C port of numba_score_float + select_topk at full size"
    best = max(runs, key=lambda r: r["qps"])
    out["cpu_baseline"] = {"value": round(runs[0]["qps"], 3), "unit": "queries/s", "cores": 32, "kind": "port",
                           "sample": f"{nqc} queries on the full index, oracle_sparse_retrieve (C/OpenMP port of numba_score_float + select_topk), "
                                     f"4 query threads x 8 posting threads = the 32-thread shape of README.md:89-94 / indexer.py:459, {runs[0]['seconds']:.1f}s",
                           "best_shape_on_this_host": {"value": round(best["qps"], 3), "threads": best["q_threads"] * best["inner_threads"],
                                                       "host_cores": cores}}
    idx.close()
    del idx, h_indptr, h_ids, h_vals
    torch.cuda.empty_cache()
    out["index_build"] = sparse_build_only(indptr, doc_ids, vals, V, N, device)
    if not args.no_drop_in:
        with contextlib.redirect_stdout(sys.stderr):          # the reference-shaped classes print like the reference does
            out["drop_in"] = drop_in_sparse_leg(args, dict(LION_1B), (indptr, doc_ids, vals, N), (q_indptr, q_cols, q_vals, nq), device)
    del indptr, doc_ids, vals
    torch.cuda.empty_cache()
    if not args.no_sparse_sweep:
        out["sparse_sweep"] = sparse_sweep_leg(args, device)
    return out


def sparse_sharded_leg(args, device, rank, world, share_gpu):
    """BASELINE.json configs[2] doc-sharded over the ranks of a multi-GPU run (the reference scores on one process after a merge_indexes
    pass, eval_sparse.py:98-114): every rank builds the SAME synthetic index from the same seed, keeps the postings of the docs
    r, r + W, ... (what `eval_sparse.py --task_name indexing` leaves in index_dir_{r}), scores its shard and the per-shard top-k meet in
    ONE gather + sr_topk_merge on rank 0 (distributed.ShardedSparseRetriever).  Timed like the headline: barrier + synchronize on both
    sides, max over ranks.  A sample of the merged rows is compared with the oracle on rank 0."""
    import synth
    from scaling_retriever_amd.distributed import ShardedSparseRetriever
    from scaling_retriever_amd.scoring import sparse_csr_expand_terms
    V, N, L0_d, L0_q, nq, k = 128256, args.sparse_docs, 128, 32, min(args.n_queries, 6980), min(args.topk, 1000)
    indptr, doc_ids, vals, _ = synth.build_index(V, N, L0_d, device, 3)
    q_indptr, q_cols, q_vals = synth.build_queries(V, nq, L0_q, device, 4)
    # this rank's shard: the postings of its docs, term order kept
    mine = (doc_ids % world) == rank
    term = sparse_csr_expand_terms(indptr, doc_ids.numel())
    counts = torch.bincount(term[mine].long(), minlength=V)
    s_indptr = torch.cat([torch.zeros(1, dtype=torch.int64, device=device), torch.cumsum(counts, 0)])
    shard = ShardedSparseRetriever(s_indptr, doc_ids[mine].to(torch.int64), vals[mine], N, rank=rank, world_size=world, device=device)
    nnz_local = int(mine.sum().item())
    host = None
    if rank == 0 and args.sparse_cpu_queries > 0:
        host = (indptr.cpu().numpy(), doc_ids.cpu().numpy(), vals.cpu().numpy())
    del indptr, doc_ids, vals, mine, term
    torch.cuda.empty_cache()

    def barrier():
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
    out = shard.search(q_indptr, q_cols, q_vals, k)          # warm-up
    barrier()
    t0 = time.perf_counter()
    steps = 3
    for _ in range(steps):
        out = shard.search(q_indptr, q_cols, q_vals, k)
    barrier()
    dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device="cpu" if share_gpu else device)
    dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    dt = float(dt.item()) / steps
    cs = shard.index.cert_stats()
    info = torch.tensor([nnz_local, cs["redone_exact"], cs["present"]], dtype=torch.int64, device="cpu" if share_gpu else device)
    got = [torch.zeros_like(info) for _ in range(world)]
    dist.all_gather(got, info)
    res = None
    if rank == 0:
        s, i, c = out
        checked = 0
        if host is not None:
            from oracle import scoring as O
            nqc = min(nq, max(8, args.sparse_cpu_queries // 12))
            hq = (q_indptr[:nqc + 1].cpu().numpy(), q_cols[:nqc * L0_q].cpu().numpy(), q_vals[:nqc * L0_q].cpu().numpy())
            oi, os_, oc = O.sparse_retrieve_c(host[0], host[1], host[2], hq[0], hq[1], hq[2], k, 0.0, N, q_threads=4,
                                              inner_threads=max(1, min(32, os.cpu_count() or 1) // 4))
            gi, gs, gc = i[:nqc].cpu().numpy(), s[:nqc].cpu().numpy(), c[:nqc].cpu().numpy()
            for q in range(nqc):
                assert gc[q] == oc[q] and np.array_equal(gi[q, :gc[q]], oi[q, :oc[q]]) and np.array_equal(gs[q, :gc[q]], os_[q, :oc[q]]), q
            checked = nqc
        res = {"metric": "sparse inverted-index queries/s, corpus doc-sharded over the ranks, ONE gather of the per-shard top-k + merge",
               "value": round(nq / dt, 1), "unit": "queries/s", "ms_per_pass": round(dt * 1e3, 1), "n_gpus": world,
               "config": {"V": V, "N": N, "L0_d": L0_d, "L0_q": L0_q, "nq": nq, "k": k},
               "shard_postings": [int(x[0]) for x in got], "certified_scorer_on_every_rank": all(int(x[2]) == 1 for x in got),
               "queries_redone_by_the_exact_kernels": [int(x[1]) for x in got], "oracle_bit_exact_queries": checked}
        log("[sparse sharded]", res)
    del shard
    torch.cuda.empty_cache()
    return res


def sparse_build_only(indptr, doc_ids, vals, V, N, device):
    """sr_sparse_csr_build (csrc/sparse_build.hip) on the whole MSMARCO-shaped collection: the index's postings are first turned into
    the doc-major triples SparseIndexer.index collects (insertion order = ascending doc, terms ascending inside a doc: itself a
    sr_sparse_csr_build, by doc), then the CSR by term is rebuilt from them, timed, and compared bit for bit with the index it came from
    (a stable sort by term of doc-ordered triples gives posting lists ascending by doc)."""
    from scaling_retriever_amd.scoring import sparse_csr_build, sparse_csr_expand_terms
    nnz = int(doc_ids.numel())
    term_of = sparse_csr_expand_terms(indptr, nnz)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    f_indptr, f_terms, f_vals = sparse_csr_build(term_of, doc_ids, vals, N)          # "terms" = docs here: the forward (doc-major) index
    torch.cuda.synchronize()
    t_fwd = time.perf_counter() - t0
    del term_of
    rows = sparse_csr_expand_terms(f_indptr, nnz)
    del f_indptr
    sparse_csr_build(rows[:1 << 20], f_terms[:1 << 20], f_vals[:1 << 20], V)          # warm-up (allocator, code objects)
    torch.cuda.synchronize()
    ts = []
    for _ in range(2):
        t0 = time.perf_counter()
        b_indptr, b_ids, b_vals = sparse_csr_build(rows, f_terms, f_vals, V)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    same = bool(torch.equal(b_indptr, indptr) and torch.equal(b_ids, doc_ids) and torch.equal(b_vals, vals))
    assert same, "sr_sparse_csr_build does not reproduce the index"
    t = min(ts)
    n_pass = 2 if V > 512 else 1
    alg_bytes = nnz * (n_pass * (4.0 + 12.0 + 12.0)) + 8.0 * V            # per pass: keys for the histogram, triples in, triples out
    out = {"workload": f"CSR by term from {nnz} doc-major postings ({N} docs x V = {V}): stable radix sort, {n_pass} passes of 8- and 9-bit digits",
           "postings_per_s": round(nnz / t, 1), "seconds": round(t, 4), "passages_per_s_at_this_L0_d": round(N / t, 1),
           "bit_identical_to_the_source_index": same,
           "roofline": {"kernel": "radix_scatter_tile_kernel + radix_hist_kernel", "bound": "hbm", "achieved": round(alg_bytes / t / 1e9, 1), "peak": PEAK_HBM_GBPS,
                        "unit": "GB/s", "frac": round(alg_bytes / t / 1e9 / PEAK_HBM_GBPS, 4), "algorithmic_bytes": alg_bytes,
                        "note": "wall time of the call incl. its allocation and the two exclusive scans; 28 B per posting and pass"},
           "forward_index_by_doc": {"seconds": round(t_fwd, 4), "postings_per_s": round(nnz / t_fwd, 1),
                                    "note": "the same call with the docs as sort key (24 bits: 3 passes), from term-major input"}}
    log("[sparse index_build]", out)
    del rows, f_terms, f_vals, b_indptr, b_ids, b_vals
    torch.cuda.empty_cache()
    return out


class _ThresholdedSparseDocs:
    """Stands where SparseIndexer expects the model.  Random-init weights give document reps with about half the vocabulary active
    (a trained Lion-SP model: L0_d ~ 100-200), so the REAL HIP encoder + sparse head run on every batch and a constant is subtracted
    from the reps (relu after it) that leaves ~L0_d entries per passage: encode, compaction and index build then see realistic sizes."""

    def __init__(self, model, shift):
        self.model, self.shift = model, float(shift)
        self.vocab_size = model.vocab_size

    def eval(self):
        return self

    def to(self, *a, **k):
        return self

    def encode(self, **inputs):
        reps = self.model.encode(**inputs)
        return reps.sub_(self.shift).clamp_min_(0.0)


def sparse_index_leg(args, device):
    """BASELINE.json's 'passages/sec encode' for configs[2]: SparseIndexer.index (indexer.py:239-308) over synthetic passages at
    Lion-SP-1B dims - LlamaBiSparse.doc_encode (body + the 128 256-wide head), sr_sparse_compact per batch, sr_sparse_csr_build at the end."""
    from scaling_retriever_amd.dataset.pipeline import TokenBudgetCollectionLoader
    from scaling_retriever_amd.indexer import SparseIndexer
    from scaling_retriever_amd.modeling.llm_encoder import LlamaBiSparse
    cfg = dict(LION_1B)
    V, H = cfg["vocab_size"], cfg["hidden_size"]
    n_pass = args.sparse_index_passages
    model = LlamaBiSparse.from_weights(cfg, random_weights(cfg, device, seed=9), max_batch_tokens=args.token_budget, max_batch_seqs=4096,
                                       fp32_planes=0).to(device).eval()
    chunks, lens = synth_token_chunks(n_pass, 4.25, 0.35, 8, 192, V, 5, (0, n_pass))

    def loader():
        return TokenBudgetCollectionLoader(tokenized=chunks, max_length=192, max_tokens=args.token_budget, max_seqs=1024,
                                           window=32768, pad_token_id=V - 1, padding_side="left")
    # The constant that leaves 128 entries per passage on average.  The loader's batches hold passages of similar length, and a rep's
    # scale grows with the number of tokens under its max: one batch calibrates its own length only (first batch: L0_d 316-390 over the
    # whole run) - so every 16th batch of the first length-bucketing window is encoded and the constant bisected over all of them.
    import itertools
    sample = []
    with torch.inference_mode(), torch.autocast("cuda", dtype=torch.bfloat16):
        for b in itertools.islice(loader(), 0, 160, 16):
            sample.append(model.encode(input_ids=b["input_ids"].to(device), attention_mask=b["attention_mask"].to(device)).float())
    rows_s = sum(int(r.shape[0]) for r in sample)
    lo, hi = 0.0, max(float(r.max().item()) for r in sample)
    for _ in range(40):
        mid = 0.5 * (lo + hi)
        cnt = sum(int((r > mid).sum().item()) for r in sample)
        lo, hi = (mid, hi) if cnt > 128 * rows_s else (lo, mid)
    shift = hi
    del sample
    stub = _ThresholdedSparseDocs(model, shift)
    with contextlib.redirect_stdout(sys.stderr):
        SparseIndexer(stub, None, device, compute_stats=True, dim_voc=V).index(
            TokenBudgetCollectionLoader(tokenized=chunks[:1], max_length=192, max_tokens=args.token_budget, max_seqs=1024,
                                        pad_token_id=V - 1, padding_side="left"))                     # warm-up
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = SparseIndexer(stub, None, device, compute_stats=True, dim_voc=V).index(loader())
        torch.cuda.synchronize()
        t = time.perf_counter() - t0
    indptr, rows, vals, n_docs = res["device_csr"]
    nnz = int(rows.numel())
    assert n_docs == n_pass and int(indptr[-1]) == nnz and len(res["ids_mapping"]) <= n_pass
    # every posting list ascends by doc (insertion order of a single rank), checked on the device
    seg = sparse_csr_expand_terms_cached(indptr, nnz)
    assert bool(((rows[1:] > rows[:-1]) | (seg[1:] != seg[:-1])).all()), "posting lists do not ascend by doc"
    tokens = int(lens.sum())
    flop = tokens * (FLOP_PER_TOKEN_1B + 2.0 * H * V) + 4.0 * float((lens.astype(np.float64) ** 2).sum()) * H * cfg["num_hidden_layers"]
    ach = flop / t / 1e12
    out = {"workload": f"SparseIndexer.index: {n_pass} synthetic passages (mean {tokens / n_pass:.1f} tokens) at Lion-SP-1B dims, token-budget loader, real HIP "
                       f"encoder + sparse head, reps thresholded to L0_d ~ 128 (random weights give no realistic sparsity), sr_sparse_compact per batch, "
                       f"sr_sparse_csr_build at the end, index kept on the device",
           "passages_per_s": round(n_pass / t, 1), "sample_passages": int(n_pass), "seconds": round(t, 2), "tokens_per_s": round(tokens / t, 1), "L0_d": round(nnz / n_pass, 1),
           "postings": nnz, "stats": {k_: round(float(v_), 2) for k_, v_ in res.get("stats", {}).items()},
           "roofline": {"kernel": "gemm_bf16_kernel (encoder body + lm_head with the segmented-max epilogue)", "bound": "mfma", "achieved": round(ach, 1),
                        "peak": PEAK_BF16_MFMA_TF, "unit": "TFLOP/s", "frac": round(ach / PEAK_BF16_MFMA_TF, 4),
                        "flop_per_token": FLOP_PER_TOKEN_1B + 2.0 * H * V,
                        "note": "SURVEY.md 8(d): 1.946 GFLOP/token body + 0.525 GFLOP/token head + attention; wall time of index() incl. the loader, "
                                "compaction, the doc id dict and the final CSR build"},
           "msmarco_extrapolation_minutes": round(8_841_823 / (n_pass / t) / 60.0, 1)}
    log("[sparse_index]", out)
    del model, stub, res
    torch.cuda.empty_cache()
    return out


def sparse_csr_expand_terms_cached(indptr, nnz):
    from scaling_retriever_amd.scoring import sparse_csr_expand_terms
    return sparse_csr_expand_terms(indptr, nnz)


def sparse_sweep_leg(args, device):
    """SURVEY.md 8(d) config 3 "as a function of L0": L0_d in {64, 128, 256} x L0_q in {16, 32, 64} (+ 128, 256 at L0_d = 128) on the Zipf(1.0) index, and one
    flatter index (df ~ r^-0.7 capped at N / 5: no term in a quarter of the documents, so no dense column and the per-query
    kernel serves every block) - queries/s, which kernel ran, and ids + fp32 scores of the first queries compared bit for bit with the
    oracle's C port of numba_score_float + select_topk (indexer.py:315-344) at full collection size."""
    import synth
    from oracle import scoring as SC
    from scaling_retriever_amd.scoring import SparseIndexHIP
    V, N, k, nq = 128256, 8_841_823, 1000, 2048
    rows = []
    for name, alpha, cap, l0ds in (("zipf1.0", 1.0, None, (64, 128, 256)), ("flat0.7", 0.7, N // 5, (128,))):
        for L0_d in l0ds:
            indptr, doc_ids, vals, df = synth.build_index(V, N, L0_d, device, 3, alpha=alpha, cap=cap)
            idx = SparseIndexHIP(indptr, doc_ids, vals, N, device=device)
            host = None
            # L0_q 128 / 256 (on the L0_d = 128 indexes): queries with 70-150 terms outside the 128 heaviest lists - more than the 64 a
            # scatter wave stages; the certified scorer adds the rest by its plain walk, `queries_redone_by_the_exact_kernels` says how
            # many it still hands back
            for L0_q in ((16, 32, 64, 128, 256) if L0_d == 128 else (16, 32, 64)):
                q_indptr, q_cols, q_vals = synth.build_queries(V, nq, L0_q, device, 4, alpha=alpha)
                s, i, c = idx.search(q_indptr, q_cols, q_vals, k)
                torch.cuda.synchronize()
                st0 = idx.block_stats()
                cs0 = idx.cert_stats()
                t0 = time.perf_counter()
                for _ in range(2):
                    s, i, c = idx.search(q_indptr, q_cols, q_vals, k)
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / 2
                st1 = idx.block_stats()
                cs1 = idx.cert_stats()
                n_check = 64
                if host is None:
                    host = (indptr.cpu().numpy(), doc_ids.cpu().numpy(), vals.cpu().numpy())
                hq = (q_indptr[:n_check + 1].cpu().numpy(), q_cols[:n_check * L0_q].cpu().numpy(), q_vals[:n_check * L0_q].cpu().numpy())
                oi, os_, oc = SC.sparse_retrieve_c(*host, *hq, k, 0.0, N, q_threads=4, inner_threads=max(1, (os.cpu_count() or 8) // 4))
                gi, gs, gc = i[:n_check].cpu().numpy(), s[:n_check].cpu().numpy(), c[:n_check].cpu().numpy()
                for q in range(n_check):
                    assert gc[q] == oc[q], (name, L0_d, L0_q, q)
                    assert np.array_equal(gi[q, :gc[q]], oi[q, :oc[q]]) and np.array_equal(gs[q, :gc[q]], os_[q, :oc[q]]), (name, L0_d, L0_q, q)
                touched = float((indptr[1:] - indptr[:-1])[q_cols.long()].reshape(nq, L0_q).sum(1).double().mean().item())
                rows.append({"index": name, "L0_d": L0_d, "L0_q": L0_q, "queries_per_s": round(nq / dt, 1), "ms_per_1k_queries": round(dt / nq * 1e6, 1),
                             "kernel": ("cert_score_kernel" if cs1["searches"] > cs0["searches"] else
                                        ("sparse_block_kernel" if st1["block_calls"] > st0["block_calls"] and st1["fallback_calls"] == st0["fallback_calls"]
                                         else ("sparse_score_kernel" if st1["block_calls"] == st0["block_calls"] else "both"))),
                             "queries_redone_by_the_exact_kernels": (cs1["redone_exact"] - cs0["redone_exact"]) // 2,
                             "heavy_terms_on_the_matrix_pipe": cs1["dense_terms"],
                             "dense_column_terms": st1["dense_terms"], "postings": int(doc_ids.numel()),
                             "mean_postings_touched_per_query": touched, "queries_bit_exact_vs_oracle": n_check})
                log("[sparse_sweep]", rows[-1])
            idx.close()
            del idx, indptr, doc_ids, vals, host
            torch.cuda.empty_cache()
    return {"nq": nq, "k": k, "N": N, "V": V, "rows": rows,
            "note": "batches of 2 048 queries; Zipf(1.0): document frequencies ~ 1 / rank capped at N, query terms drawn from the same law; "
                    "flat0.7: ~ rank^-0.7 capped at N / 5 (no dense columns)"}


def _loader_batches(q_batches_dev, qids, batch):
    """What DataLoader(query_dataset, batch_size=128, collate_fn=LlamaDenseCollectionCollator) yields: CPU tensors, left-padded to
    the batch's longest row, ids = the query ids (strings, as MSMARCOQueryDataset keeps them)."""
    out, r0 = [], 0
    for ids, mask in q_batches_dev:
        n = ids.shape[0]
        out.append({"input_ids": ids.cpu(), "attention_mask": mask.cpu(), "ids": qids[r0:r0 + n]})
        r0 += n
    assert all(len(b["ids"]) <= batch for b in out)
    return out


def drop_in_dense_leg(args, cfg, model, index, device, n_local):
    """The reference's own call path at the full shape: DataLoader batches of --eval_batch_size (128) queries ->
    LocalFaissDenseRetriever.get_top_docs (generate_query_vecs + DenseFlatIndexer.search_knn, eval_dense.py:94-135) -> the run.json
    of eval_dense.py:225-241.  Wall times of host + device work, inputs as the loader hands them over (CPU tensors)."""
    import tempfile
    import eval_dense
    from scaling_retriever_amd.indexer import DenseFlatIndexer
    B = 128
    q_batches, _ = synth_batches(args.n_queries, B, 2.1, 0.35, 4, 64, cfg["vocab_size"], 2, device)
    qids = [str(1_000_000 + 7 * i) for i in range(args.n_queries)]
    loader = _loader_batches(q_batches, qids, B)
    fi = DenseFlatIndexer()
    fi.hidden_dim, fi.index = cfg["hidden_size"], index                         # the bench's resident index (no second copy of D)
    fi._update_id_mapping(np.arange(n_local).astype("U8").tolist())             # MS MARCO pids are decimal strings
    fi.id_table(), fi.run_table()                                               # built once per index, like the id list itself
    retriever = eval_dense.LocalFaissDenseRetriever(model, device=device, index=fi)

    def wall(fn, n=2):
        """Mean wall time of fn() over n calls after one warm-up call.  The previous call's result is dropped OUTSIDE the timed
        region: freeing 7 M references of a list-of-lists result is the caller's business, not the call's."""
        r = fn()
        total = 0.0
        for _ in range(n):
            r = None
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            r = fn()
            torch.cuda.synchronize()
            total += time.perf_counter() - t0
        return total / n, r

    def per_batch_encode():                     # the reference's loop, one query_encode call per loader batch
        with torch.no_grad():
            return torch.cat([model.query_encode(input_ids=b["input_ids"].to(device), attention_mask=b["attention_mask"].to(device)) for b in loader])
    t_pb, reps_pb = wall(per_batch_encode, 1)
    t_gq, (reps, _) = wall(lambda: eval_dense.generate_query_vecs(model, loader, device))
    same_bits = bool(torch.equal(reps, reps_pb))
    t_knn, (top_ids, top_scores) = wall(lambda: fi.search_knn(reps, args.topk))
    t_arr, (_, arr_idx) = wall(lambda: fi.search_arrays(reps, args.topk))
    table = fi.id_table()
    t_map, _ = wall(lambda: fi.id_lists(arr_idx))
    t_map_np, _ = wall(lambda: [table.take(row).tolist() for row in arr_idx], 1)
    assert isinstance(top_ids[0], list) and len(top_ids) == args.n_queries
    t_top, _ = wall(lambda: retriever.get_top_docs(loader, args.topk))
    tmp = tempfile.mkdtemp(prefix="sr_bench_run_")
    path = os.path.join(tmp, "run.json")
    try:
        t_run, (nq_, nbytes) = wall(lambda: retriever.write_run(loader, args.topk, path))
        # the file is the reference's: parse a slice of it back and compare with search_knn's lists
        with open(path) as f:
            head = f.read(1 << 20)
        first = json.loads(head[:head.index("}") + 1] + "}")
        q0 = next(iter(first))
        assert q0 == qids[0] and list(first[q0])[:50] == [str(x) for x in top_ids[0][:50]]
        assert abs(list(first[q0].values())[0] - float(top_scores[0][0])) == 0.0
    finally:
        import shutil
        shutil.rmtree(tmp, ignore_errors=True)
    nq = args.n_queries
    out = {"workload": f"{nq} Dev-shaped queries in {len(loader)} loader batches of {B} (CPU tensors, as the DataLoader yields them), "
                       f"top-{args.topk} over {n_local} x {cfg['hidden_size']}, document ids = decimal strings",
           "generate_query_vecs": {"ms": round(t_gq * 1e3, 1), "queries_per_s": round(nq / t_gq, 1),
                                   "note": "loader batches coalesced into one engine pass (encode_batches / sr_encode_rows)",
                                   "bit_identical_to_one_call_per_batch": same_bits,
                                   "one_query_encode_call_per_loader_batch_ms": round(t_pb * 1e3, 1)},
           "search_knn": {"ms": round(t_knn * 1e3, 1), "queries_per_s": round(nq / t_knn, 1),
                          "search_arrays_ms": round(t_arr * 1e3, 1), "id_mapping_alone_ms": round(t_map * 1e3, 1),
                          "id_mapping_by_numpy_take_per_row_ms": round(t_map_np * 1e3, 1),
                          "note": "returns the reference's list of lists of db ids: the query set is searched in pieces (sr_dense_search + D2H "
                                  "through pinned memory), the host builds the lists of piece c (csrc/host_lists.c: hits radix-sorted by index position, id objects "
                                  "visited in index order) while the GPU searches piece c + 1; search_arrays_ms = the whole set in one search + D2H, "
                                  "id_mapping_alone_ms = the 7 M references by themselves (the numpy take + tolist per row it replaces beside it)"},
           "get_top_docs": {"ms": round(t_top * 1e3, 1), "queries_per_s": round(nq / t_top, 1)},
           "retrieval_task_with_run_json": {"ms": round(t_run * 1e3, 1), "queries_per_s": round(nq / t_run, 1), "run_json_bytes": int(nbytes),
                                            "timeline_of_the_last_call": retriever.last_run_timeline,
                                            "note": "generate_query_vecs + search + sr_write_run_json (the bytes json.dump of the reference's nested "
                                                    "dict gives, tests/test_run_file.py): what eval_dense.py --task_name retrieval does after the index is resident"}}
    assert same_bits, "coalesced query encode differs from the per-batch calls"
    log("[drop_in dense]", out)
    return out


class _SyntheticSparseQueries:
    """Stands where SparseRetrieval expects the model.  Random-init weights give sparse reps with ~half the vocabulary active, which
    no trained Lion-SP model does (L0_q ~ 32), so the REAL HIP encoder runs on every batch (its cost is what is timed) and the
    reps handed on are the synthetic Zipf query vectors of tools/synth.py for those rows."""

    def __init__(self, model, q_indptr, q_cols, q_vals, V, rows_of):
        self.model, self.V, self.rows_of = model, V, rows_of
        self.q_indptr, self.q_cols, self.q_vals = q_indptr, q_cols, q_vals
        self.vocab_size = V

    def eval(self):
        return self

    def to(self, *a, **k):
        return self

    def _reps(self, first_row, n):
        dev = self.q_cols.device
        lo, hi = int(self.q_indptr[first_row]), int(self.q_indptr[first_row + n])
        rows = torch.repeat_interleave(torch.arange(n, device=dev), self.q_indptr[first_row + 1:first_row + n + 1] - self.q_indptr[first_row:first_row + n])
        reps = torch.zeros((n, self.V), dtype=torch.float32, device=dev)
        reps[rows, self.q_cols[lo:hi].long()] = self.q_vals[lo:hi]
        return reps

    def encode(self, **inputs):
        self.model.encode(**inputs)
        return self._reps(self.rows_of[inputs["input_ids"].data_ptr()], inputs["input_ids"].shape[0])

    def encode_batches(self, batches):
        self.model.encode_batches(batches)
        first = self.rows_of[batches[0]["input_ids"].data_ptr()]
        return self._reps(first, sum(b["input_ids"].shape[0] for b in batches))


def drop_in_sparse_leg(args, cfg_model, idx_parts, q_parts, device):
    """SparseRetrieval.retrieve (indexer.py:530-540) at the full shape: loader batches of 128 -> _generate_query_vecs (real HIP sparse
    encoder at Lion-SP-1B dims on every batch; see _SyntheticSparseQueries) -> sr_sparse_search -> q_stats.json + run.json."""
    import shutil
    import tempfile
    from scaling_retriever_amd.indexer import SparseRetrieval
    from scaling_retriever_amd.modeling.llm_encoder import LlamaBiSparse
    from scaling_retriever_amd.utils.inverted_index import IndexDictOfArray
    indptr, doc_ids, vals, N = idx_parts
    q_indptr, q_cols, q_vals, nq = q_parts
    V, B = cfg_model["vocab_size"], 128
    model = LlamaBiSparse.from_weights(cfg_model, random_weights(cfg_model, device, seed=9), max_batch_tokens=32768, max_batch_seqs=4096,
                                       fp32_planes=0).to(device).eval()
    q_batches, _ = synth_batches(nq, B, 2.1, 0.35, 4, 64, V, 2, device)
    qids = [str(1_000_000 + 7 * i) for i in range(nq)]
    loader = _loader_batches(q_batches, qids, B)

    class _Loader(list):
        pass
    # SparseRetrieval moves a batch to the device before encode(): remember which rows a batch holds by its position
    rows_of, r0 = {}, 0
    dev_loader = _Loader()
    for b in loader:
        db = {"input_ids": b["input_ids"].to(device), "attention_mask": b["attention_mask"].to(device), "ids": b["ids"]}
        rows_of[db["input_ids"].data_ptr()] = r0
        r0 += len(b["ids"])
        dev_loader.append(db)
    stub = _SyntheticSparseQueries(model, q_indptr, q_cols, q_vals, V, rows_of)
    container = IndexDictOfArray(dim_voc=V)
    container.set_device_csr(indptr, doc_ids, vals, N)
    index_d = {"index": container, "ids_mapping": {i: str(i) for i in range(N)}, "device_csr": (indptr, doc_ids, vals, N), "stats": {}}
    tmp = tempfile.mkdtemp(prefix="sr_bench_sprun_")
    retr = None
    try:
        retr = SparseRetrieval(stub, {"out_dir": tmp}, V, device, index_d=index_d, compute_stats=True)
        retr.doc_id_table()                                                  # per index, like doc_ids.pkl itself
        retr.retrieve(dev_loader, args.topk, threshold=0.0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = retr.retrieve(dev_loader, args.topk, threshold=0.0)
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t0
        t0 = time.perf_counter()
        qv, _ = retr._generate_query_vecs(dev_loader)
        torch.cuda.synchronize()
        t_gen = time.perf_counter() - t0
        t0 = time.perf_counter()
        retr._sparse_retrieve_multithreaded(qv, qids, threshold=0.0, topk=args.topk)
        torch.cuda.synchronize()
        t_ret = time.perf_counter() - t0
        nbytes = os.path.getsize(os.path.join(tmp, "run.json"))
        with open(os.path.join(tmp, "run.json")) as f:
            head = f.read(1 << 20)
        first = json.loads(head[:head.index("}") + 1] + "}")
        assert next(iter(first)) == qids[0] and first[qids[0]] == res[qids[0]]
        l0 = json.load(open(os.path.join(tmp, "q_stats.json")))["L0_q"]
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
        if retr is not None:
            retr.hip_index.close()
    del model
    out = {"workload": f"SparseRetrieval.retrieve: {nq} queries in {len(dev_loader)} loader batches of {B}, real HIP LlamaBiSparse encode at 1B dims per "
                       f"group + the synthetic Zipf query vectors (random weights give no realistic L0), top-{args.topk}, q_stats.json + run.json",
           "retrieve": {"ms": round(t_all * 1e3, 1), "queries_per_s": round(nq / t_all, 1), "run_json_bytes": int(nbytes), "L0_q": l0},
           "generate_query_vecs_ms": round(t_gen * 1e3, 1), "search_to_RunResult_ms": round(t_ret * 1e3, 1)}
    log("[drop_in sparse]", out)
    return out


LION_8B = dict(vocab_size=128256, hidden_size=4096, intermediate_size=14336, num_hidden_layers=32, num_attention_heads=32,
               num_key_value_heads=8, head_dim=128, rms_norm_eps=1e-5, rope_theta=500000.0, tie_word_embeddings=False)
FLOP_PER_TOKEN_8B = 13.96e9  # 2 x linear params of the 8B body


def config5_leg(args, device):
    """BASELINE.json configs[4] on ONE of its 8 GPUs: Lion-DS-8B dims (llama-3-8b, train_configs/mntp/meta_llama3_8b_msmarco.json:2),
    bf16-autocast corpus encode through store_embs and the score stage over this GPU's 1/8 doc shard at H = 4096."""
    from scaling_retriever_amd.modeling.llm_encoder import LlamaBiDense
    from scaling_retriever_amd.scoring import DenseIndexHIP
    cfg = dict(LION_8B)
    H, L = cfg["hidden_size"], cfg["num_hidden_layers"]
    model = LlamaBiDense.from_weights(cfg, random_weights(cfg, device, seed=5), max_batch_tokens=32768, max_batch_seqs=2048,
                                      fp32_planes=0).to(device).eval()          # documents only here: no fp32-regime planes
    sub = argparse.Namespace(encode_passages=args.config5_passages, token_budget=args.token_budget)
    enc = encode_leg(sub, cfg, model, device, 0, 1, False, flop_per_token=FLOP_PER_TOKEN_8B, layers_ref=32)
    del model
    torch.cuda.empty_cache()
    n_shard = (args.n_docs + 7) // 8
    D = torch.empty((n_shard, H), dtype=torch.float32, device=device)
    g = torch.Generator(device=device).manual_seed(7)
    for r0 in range(0, n_shard, 1 << 19):
        D[r0:r0 + (1 << 19)].normal_(0.0, 0.5 / H ** 0.5, generator=g)
    Q = torch.empty((args.n_queries, H), dtype=torch.float32, device=device).normal_(0.0, 0.5 / H ** 0.5, generator=g)
    index = DenseIndexHIP(H, device=device)
    index.add_device_rows(D, id_base=0, id_stride=8)
    index.search(Q, args.topk)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(3):
        index.search(Q, args.topk)
    torch.cuda.synchronize()
    t = (time.perf_counter() - t) / 3
    tf = 2.0 * args.n_queries * n_shard * H / t / 1e12
    q1 = Q[:1].contiguous()
    index.search(q1, args.topk)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(5):
        index.search(q1, args.topk)
    torch.cuda.synchronize()
    t1 = (time.perf_counter() - t1) / 5
    gbps = n_shard * H * 4 / t1 / 1e9
    # the headline's method on this shard: certified fp16 upper-bound filter + exact re-score, bit-identical to the exact kernel
    es, ei = index.search(Q, args.topk)
    index.set_precision("fp32_filtered")
    fs, fi = index.search(Q, args.topk)
    c0, r0 = index.filter_query_stats()
    if not (torch.equal(fs, es) and torch.equal(fi, ei)):
        raise AssertionError("config5 shard: filtered search differs from the exact kernel")
    torch.cuda.synchronize()
    tf_ = time.perf_counter()
    for _ in range(3):
        index.search(Q, args.topk)
    torch.cuda.synchronize()
    tf_ = (time.perf_counter() - tf_) / 3
    c1, r1 = index.filter_query_stats()
    index.close()
    return {"workload": "Lion-DS-8B dims (H 4096, 32 layers, 32/8 heads of 128, MLP 14336), one GPU of the 8: bf16-autocast corpus encode + "
                        f"exact fp32 score stage over its {n_shard} x {H} doc shard",
            "encode": enc,
            "score_shard": {"queries_per_s": round(args.n_queries / t, 1), "ms": round(t * 1e3, 1),
                            "roofline": {"kernel": "dense_score_pipe_kernel", "bound": "mfma", "achieved": round(tf, 1), "peak": PEAK_F32_MFMA_TF,
                                         "unit": "TFLOP/s", "frac": round(tf / PEAK_F32_MFMA_TF, 4)}},
            "score_shard_filtered": {"queries_per_s": round(args.n_queries / tf_, 1), "ms": round(tf_ * 1e3, 1),
                                     "parity": "ids and fp32 scores bit-identical to the exact kernel on this shard",
                                     "queries_certified": int(c1 - c0), "queries_redone_by_exact_kernel": int(r1 - r0),
                                     "algorithmic_TFLOPs": round(2.0 * args.n_queries * n_shard * H / tf_ / 1e12, 1)},
            "score_shard_one_query": {"ms": round(t1 * 1e3, 2),
                                      "roofline": {"kernel": "dense_stream_kernel", "bound": "hbm", "achieved": round(gbps, 1), "peak": PEAK_HBM_GBPS,
                                                   "unit": "GB/s", "frac": round(gbps / PEAK_HBM_GBPS, 4)}}}


def launch_ranks(n, argv, dry_run=False):
    """Start `n` ranks of this script on this node (one per GPU) through torch.distributed.run and relay rank 0's line.
    Returns the exit status: 0 only if every rank exited cleanly.  Called before any GPU call of this process."""
    import socket
    import subprocess
    share = os.environ.get("SR_BENCH_SHARE_GPU") == "1"          # dry run of the multi-rank path on a one-GPU box (gloo, every rank on cuda:0)
    have = torch.cuda.device_count()
    if have < n and not share:
        log(f"error: --gpus {n} needs {n} visible GPUs, found {have} (set SR_BENCH_SHARE_GPU=1 only for a plumbing dry run on one GPU)")
        return 2
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + [a for a in argv if a != "--print-launch"]
    if dry_run:
        print(json.dumps({"launch": cmd}), flush=True)
        return 0
    log("[launch]", " ".join(cmd))
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    proc = subprocess.Popen(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, text=True)
    last = None
    for line in proc.stdout:                                    # the children's stderr goes straight through
        if line.startswith("{"):
            last = line
        else:
            sys.stderr.write(line)
    rc = proc.wait()
    if rc != 0:
        log(f"error: the {n}-rank run exited with status {rc}")
        return rc if 0 < rc < 256 else 1
    if last is None:
        log("error: the ranks exited cleanly but rank 0 printed no result line")
        return 1
    sys.stdout.write(last)
    sys.stdout.flush()
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n-docs", type=int, default=8_841_823)
    ap.add_argument("--n-queries", type=int, default=6980)
    ap.add_argument("--topk", type=int, default=1000)
    ap.add_argument("--query-batch", type=int, default=6980,
                    help="queries per query_encode call (eval_dense.py's --eval_batch_size; the reference script uses 128).  The whole Dev set "
                         "per call lets the encoder cut it by its own token budget; 1 800 / 2 048 / 940 per call: +1 / +6 / +12 % encode time")
    ap.add_argument("--encode-passages", type=int, default=131072, help="synthetic passages pushed through store_embs for the encode figure")
    ap.add_argument("--token-budget", type=int, default=16384, help="real tokens per doc_encode batch of the encode leg")
    ap.add_argument("--no-encode", action="store_true")
    ap.add_argument("--no-sparse", action="store_true")
    ap.add_argument("--sparse-docs", type=int, default=8_841_823, help="collection size of the doc-sharded sparse leg of a multi-GPU run")
    ap.add_argument("--no-config5", action="store_true", help="skip the Lion-DS-8B leg (BASELINE.json configs[4], one GPU's share)")
    ap.add_argument("--config5-passages", type=int, default=8192)
    ap.add_argument("--sparse-cpu-queries", type=int, default=768, help="bounded CPU sample of the sparse baseline (~10 s per threading shape)")
    ap.add_argument("--layers", type=int, default=None, help="override num layers (debug only; invalidates the number)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--exact-kernel", action="store_true", help="headline through the exact fp32 MFMA kernel instead of the certified filter")
    ap.add_argument("--no-fast-mode", action="store_true", help="skip the extra bf16x3 / bf16x6 precision-mode measurements")
    ap.add_argument("--no-shard-leg", action="store_true", help="skip the shard_1of8 leg (one of 8 doc shards with the threshold exchange)")
    ap.add_argument("--no-drop-in", action="store_true", help="skip the drop_in legs (the reference's own call path: loader batches of 128 -> "
                    "get_top_docs / SparseRetrieval.retrieve -> run.json)")
    ap.add_argument("--no-sparse-sweep", action="store_true", help="skip the L0_d x L0_q / flat-distribution sweep of the sparse scorer")
    ap.add_argument("--no-sparse-index", action="store_true", help="skip the SparseIndexer.index leg (sparse passages/s)")
    ap.add_argument("--sparse-index-passages", type=int, default=131072, help="synthetic passages of the sparse_index leg")
    ap.add_argument("--print-launch", action="store_true", help="with --gpus N > 1 and no launcher: print the command that would start the ranks and exit")
    ap.add_argument("--no-robustness", action="store_true", help="skip the filter_robustness legs (anisotropic / near-duplicate corpora at full shape)")
    args = ap.parse_args()

    # `python bench.py --gpus N` with N > 1 and no launcher around it starts the N ranks itself (before anything touches a GPU:
    # device_count() does not initialise one): one child `python -m torch.distributed.run --nproc-per-node N bench.py <same arguments>`,
    # whose stdout (rank 0's JSON line) is relayed; the exit status is the child's.  Nothing is retried or re-executed in place.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:], dry_run=args.print_launch))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        log(f"error: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
        sys.exit(2)
    # SR_BENCH_SHARE_GPU=1: dry run of the multi-rank path on a ONE-GPU box (every rank on cuda:0, gloo instead of RCCL,
    # which refuses two ranks on one device); the driver's real runs leave it unset
    share_gpu = os.environ.get("SR_BENCH_SHARE_GPU") == "1"
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        if share_gpu:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=device)

    from scaling_retriever_amd import _lib
    from scaling_retriever_amd.distributed import all_gather_query_reps, gather_topk, query_slice, shard_size, sharded_dense_search
    from scaling_retriever_amd.modeling.llm_encoder import LlamaBiDense
    from scaling_retriever_amd.scoring import DenseIndexHIP, topk_merge
    lib = _lib.load()

    cfg = dict(LION_1B)
    if args.layers:
        cfg["num_hidden_layers"] = args.layers
    H = cfg["hidden_size"]
    t_setup = time.time()
    model = LlamaBiDense.from_weights(cfg, random_weights(cfg, device, seed=0), max_batch_tokens=65536, max_batch_seqs=8192).to(device).eval()
    assert model.base_model.resolve_precision() == "fp32"      # no autocast here: the reference's dense-query regime
    # every rank encodes only its block of the queries; the embeddings are all-gathered (second, 57 MB collective)
    q_rows = query_slice(args.n_queries, rank, world)
    q_batches, q_lens = synth_batches(args.n_queries, args.query_batch, 2.1, 0.35, 4, 64, cfg["vocab_size"], 2, device, rows=q_rows)
    n_local = shard_size(args.n_docs, rank, world)
    D = torch.empty((n_local, H), dtype=torch.float32, device=device)
    g = torch.Generator(device=device).manual_seed(1 + rank)
    for r0 in range(0, n_local, 1 << 20):
        D[r0:r0 + (1 << 20)].normal_(0.0, 0.5 / H ** 0.5, generator=g)
    index = DenseIndexHIP(H, device=device)
    index.add_device_rows(D, id_base=rank, id_stride=world)
    # exact results through the certified bf16 filter + exact re-score (bit-identical to the exact fp32 kernel, checked below);
    # without room for the bf16 plane of D the library uses the exact kernel by itself
    index.set_precision("fp32" if args.exact_kernel else "fp32_filtered")
    torch.cuda.synchronize()
    log(f"[rank {rank}] setup {time.time() - t_setup:.1f}s: {n_local} docs x {H} fp32 = {n_local * H * 4 / 1e9:.1f} GB resident; "
        f"{args.n_queries} queries, mean {q_lens.mean():.1f} tokens")

    def encode_queries():
        with torch.no_grad():                                   # eval_dense.py:101 - and no autocast: fp32 regime
            local = [model.query_encode(input_ids=i, attention_mask=m) for i, m in q_batches]
        local = torch.cat(local) if local else torch.zeros((0, H), dtype=torch.float32, device=device)
        return all_gather_query_reps(local, args.n_queries)

    def step():
        reps = encode_queries()
        s, i = sharded_dense_search(index, reps, args.topk, world)      # world > 1: + one all-reduce(min) of nq floats
        if world > 1:
            gs, gi = gather_topk(s, i, dst=0)
            if gs is not None:
                s, i = topk_merge(gs, gi)
        return s, i

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    _lib.check(lib.sr_dense_index_profile(index._h, 1))
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    barrier()
    dt = time.perf_counter() - t0
    _lib.check(lib.sr_dense_index_profile(index._h, 0))
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if share_gpu else device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    n_l, ms, fl, by = ctypes.c_int64(0), ctypes.c_double(0), ctypes.c_double(0), ctypes.c_double(0)
    _lib.check(lib.sr_dense_index_profile_read(index._h, ctypes.byref(n_l), ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(by)))
    achieved_tf = fl.value / (ms.value * 1e-3) / 1e12 if ms.value > 0 else 0.0
    # HBM/fabric traffic per launch of the dominant kernel comes from the committed PMC passes (rocprofv3
    # cannot run inside this process); used only when it was measured on the same launch shape.
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "r01_pmc_summary.json")) as f:
            pmc = json.load(f)
        shape = pmc["dense_score_launch"]
        docs_per_launch = fl.value / max(1, n_l.value) / (2.0 * args.n_queries * H)
        if shape["nq"] == args.n_queries and shape["dim"] == H and abs(docs_per_launch / shape["docs_per_launch"] - 1) < 0.02:
            traffic = [v["traffic_bytes"] for k, v in pmc["kernels"].items() if k.startswith("dense_score_pipe_kernel")][0]
    except Exception:
        traffic = None
    n_filtered, n_fallback = index.filter_stats()
    filtered = (not args.exact_kernel) and n_filtered > 0
    # HBM / fabric bytes per launch of the filter pass: from the committed PMC passes of THIS round's kernel (rocprofv3 cannot
    # run inside this process), used only when measured on the same problem shape; the file names the commit it was taken at
    split_traffic, split_traffic_src = None, None
    try:
        with open(os.path.join(ROOT, "profiles", "r06_pmc_traffic.json")) as f:
            pmc2 = json.load(f)
        shape2 = pmc2["dense_split_launch"]
        import hashlib
        with open(os.path.join(ROOT, pmc2["kernel_source"]["file"]), "rb") as f:
            same_kernel = hashlib.sha256(f.read()).hexdigest() == pmc2["kernel_source"]["sha256"]       # stale once the kernel changes
        if (same_kernel and shape2["nq"] == args.n_queries and shape2["dim"] == H and shape2["n_docs"] == n_local and world == 1
                and abs(n_l.value / max(1, args.steps) / shape2["launches_per_search"] - 1) < 0.02):
            split_traffic = [v for kname, v in pmc2["kernels"].items() if kname.startswith("dense_split_kernel") and "<false>" not in kname][0]["traffic_bytes"]
            split_traffic_src = "profiles/r06_pmc_traffic.json (separate FETCH_SIZE / WRITE_SIZE passes, " + pmc2.get("commit", "?") + ")"
    except Exception:
        split_traffic = None
    if filtered:
        # dominant kernel: dense_split_kernel<true> - ONE fp16 plane product per algorithmic multiply-add on the 16-bit MFMA pipe
        qc, qr = index.filter_query_stats()
        roofline = {"kernel": "dense_split_kernel<upper bound> (fp16 MFMA 16x16x32, 256 docs x 256 queries per tile, persistent workgroups; the "
                              "certified filter's pass: q0 . d0 + e(q, j), 1 plane product per fp32 multiply-add)",
                    "bound": "mfma", "achieved": round(achieved_tf, 1), "peak": PEAK_BF16_MFMA_TF,
                    "unit": "TFLOP/s (fp16 MFMA work = algorithmic 2 nq N H; fp16 and bf16 MFMA have the same peak)",
                    "frac": round(achieved_tf / PEAK_BF16_MFMA_TF, 4), "filter_plane_products": 1,
                    "algorithmic_TFLOPs": round(achieved_tf, 1), "traffic": split_traffic,
                    "traffic_source": split_traffic_src,
                    "launches": int(n_l.value), "avg_launch_ms": round(ms.value / max(1, n_l.value), 4),
                    "flop_per_launch": fl.value / max(1, n_l.value), "kernel_share_of_step": round(ms.value * 1e-3 / dt, 3),
                    "searches_through_filter": int(n_filtered), "searches_with_queries_redone_by_exact_kernel": int(n_fallback),
                    "queries_certified": int(qc), "queries_redone_by_exact_kernel": int(qr)}
    else:
        roofline = {"kernel": "dense_score_pipe_kernel (fp32 MFMA 32x32x2, 256 docs x 256 queries per workgroup, 8 waves, 3 LDS stages)",
                    "bound": "mfma", "achieved": round(achieved_tf, 2), "peak": PEAK_F32_MFMA_TF, "unit": "TFLOP/s",
                    "frac": round(achieved_tf / PEAK_F32_MFMA_TF, 4), "traffic": traffic,
                    "launches": int(n_l.value), "avg_launch_ms": round(ms.value / max(1, n_l.value), 4),
                    "flop_per_launch": fl.value / max(1, n_l.value), "kernel_share_of_step": round(ms.value * 1e-3 / dt, 3)}

    # ---- where a step's time goes (one extra pass of each stage, synchronised; not part of the timed region) ----
    torch.cuda.synchronize()
    tb = time.perf_counter()
    reps_b = encode_queries()
    torch.cuda.synchronize()
    t_enc = time.perf_counter() - tb
    tb = time.perf_counter()
    index.search(reps_b, args.topk)
    torch.cuda.synchronize()
    t_search = time.perf_counter() - tb
    breakdown = {"query_encode_ms": round(t_enc * 1e3, 1), "search_ms": round(t_search * 1e3, 1),
                 "query_tokens": int(q_lens.sum()), "query_encode_calls": len(q_batches)}

    # ---- the reference's own call path at the same shape: loader batches of 128 -> get_top_docs -> run.json ----
    drop_in = None
    if world == 1 and filtered and not args.no_drop_in:
        with contextlib.redirect_stdout(sys.stderr):
            drop_in = drop_in_dense_leg(args, cfg, model, index, device, n_local)

    # ---- the same step with the score kernel in split-bf16 arithmetic (opt-in precision modes of sr_dense_search;
    #      fp32 operands split into bf16 planes, 3 / 6 plane products on the bf16 MFMA pipe, fp32 accumulate) ----
    mode_steps = min(args.steps, 3)     # secondary figures: a few steps are enough, the default run must stay short

    def timed_mode(mode, n_prod, note):
        try:
            index.set_precision(mode)
            step()
            _lib.check(lib.sr_dense_index_profile(index._h, 1))
            barrier()
            tf0 = time.perf_counter()
            for _ in range(mode_steps):
                step()
            barrier()
            dtf = time.perf_counter() - tf0
            _lib.check(lib.sr_dense_index_profile(index._h, 0))
            if world > 1:
                t = torch.tensor([dtf], dtype=torch.float64, device="cpu" if share_gpu else device)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dtf = float(t.item())
            _lib.check(lib.sr_dense_index_profile_read(index._h, ctypes.byref(n_l), ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(by)))
            eq_tf = fl.value / (ms.value * 1e-3) / 1e12 if ms.value > 0 else 0.0
            return {"precision": mode, "value": round(args.n_queries * mode_steps / dtf, 2), "unit": "queries/s",
                    "ms_per_step": round(dtf / mode_steps * 1e3, 2), "steps": mode_steps,
                    "roofline": {"kernel": f"dense_split_kernel (bf16 MFMA 16x16x32, {n_prod} products per fp32-equivalent FMA)", "bound": "mfma",
                                 "achieved": round(n_prod * eq_tf, 1), "peak": PEAK_BF16_MFMA_TF,
                                 "unit": f"TFLOP/s (bf16 MFMA work = {n_prod} x algorithmic)",
                                 "frac": round(n_prod * eq_tf / PEAK_BF16_MFMA_TF, 4), "fp32_equivalent_TFLOPs": round(eq_tf, 1),
                                 "launches": int(n_l.value), "avg_launch_ms": round(ms.value / max(1, n_l.value), 4)},
                    "note": note}
        except MemoryError as e:
            return {"precision": mode, "skipped": str(e)}
        finally:
            index.set_precision("fp32" if args.exact_kernel else "fp32_filtered")

    # ---- the same step through the exact fp32 MFMA kernel (what the filter must reproduce), and the proof on THIS run's data ----
    exact_mode = None
    parity = None
    parity_ref = None
    if filtered:
        index.set_precision("fp32")
        es, ei = index.search(reps_b, args.topk)
        index.set_precision("fp32_filtered")
        fs, fi = index.search(reps_b, args.topk)
        same = bool(torch.equal(es, fs) and torch.equal(ei, fi))
        if not same:          # say what differs before the run aborts
            bad = (~((fs == es).all(1) & (fi == ei).all(1))).nonzero()[:, 0]
            log(f"[parity] {bad.numel()} queries differ: {bad.tolist()[:16]}; stats {index.filter_stats()} {index.filter_query_stats()}")
            for q_ in bad.tolist()[:4]:
                d_ = ((fs[q_] != es[q_]) | (fi[q_] != ei[q_])).nonzero()[:, 0]
                j_ = int(d_[0])
                log(f"   q {q_}: first diff at rank {j_} of {d_.numel()}: filtered ({float(fs[q_, j_])}, {int(fi[q_, j_])}) exact ({float(es[q_, j_])}, {int(ei[q_, j_])}); "
                    f"exact id in the filtered list: {bool((fi[q_] == ei[q_, j_]).any())}; filtered id in the exact list: {bool((ei[q_] == fi[q_, j_]).any())}")
            fs2, fi2 = index.search(reps_b, args.topk)
            log(f"   second filtered search equals exact: {bool(torch.equal(es, fs2) and torch.equal(ei, fi2))}; equals the first filtered: {bool(torch.equal(fs, fs2) and torch.equal(fi, fi2))}")
            index.set_precision("fp32")
            es2, ei2 = index.search(reps_b, args.topk)
            index.set_precision("fp32_filtered")
            log(f"   second exact search equals the first exact: {bool(torch.equal(es, es2) and torch.equal(ei, ei2))}; equals the first filtered: {bool(torch.equal(fs, es2) and torch.equal(fi, ei2))}")
        parity = (f"filter + exact re-score vs exact fp32 kernel on this run's {args.n_queries} x {n_local} problem: ids and fp32 scores "
                  + ("bit-identical" if same else "DIFFER"))
        assert same, parity
        parity_ref = (es, ei)
        del fs, fi
        exact_mode = timed_mode("fp32", 1, "every one of the nq x N products on the fp32 MFMA pipe (dense_score_pipe_kernel); the filtered "
                                "headline returns the same bits")
        if exact_mode and "roofline" in exact_mode:
            r_ = exact_mode["roofline"]
            r_.update({"kernel": "dense_score_pipe_kernel (fp32 MFMA 32x32x2)", "peak": PEAK_F32_MFMA_TF, "unit": "TFLOP/s",
                       "achieved": r_["fp32_equivalent_TFLOPs"], "frac": round(r_["fp32_equivalent_TFLOPs"] / PEAK_F32_MFMA_TF, 4), "traffic": traffic})
    fast = fp32_class = None
    if not args.no_fast_mode:
        fp32_class = timed_mode("bf16x6", 6, "3 bf16 planes = the whole fp32 significand: scores within 4e-7*|q||d| of the exact fp32 path, the "
                                "error class of an fp32 dot product (tests/test_dense_bf16x3_gpu.py); needs 3 bf16 planes of D (+1.5x)")
        fast = timed_mode("bf16x3", 3, "scores within 2.5e-6*|q||d| of the exact fp32 path (tests/test_dense_bf16x3_gpu.py); "
                          "uses the first 2 bf16 planes of D")

    # ---- online / small-batch regime (north_star's "HBM-bound score"): one pass over the resident fp32
    #      matrix for 1 and 16 queries; bytes = this rank's rows x H x 4 ----
    small = []
    for nq_s in (1, 16):
        qs = reps_b[:nq_s].contiguous()
        index.search(qs, args.topk)
        torch.cuda.synchronize()
        ts = time.perf_counter()
        for _ in range(3):
            index.search(qs, args.topk)
        torch.cuda.synchronize()
        t_s = (time.perf_counter() - ts) / 3
        gbps = n_local * H * 4 / t_s / 1e9
        # the same batch with one k order for every batch size (sr_dense_index_set_batch_invariant: the tiled kernel's 32-query
        # configuration instead of the streaming kernel; the bits then equal the query's row of a 6 980-query search)
        index.set_batch_invariant(True)
        s_inv, i_inv = index.search(qs, args.topk)
        torch.cuda.synchronize()
        ts = time.perf_counter()
        for _ in range(3):
            index.search(qs, args.topk)
        torch.cuda.synchronize()
        t_inv = (time.perf_counter() - ts) / 3
        index.set_batch_invariant(False)
        same_as_big = bool(torch.equal(s_inv, out[0][:nq_s]) and torch.equal(i_inv, out[1][:nq_s])) if (world == 1 and filtered) else None
        small.append({"nq": nq_s, "ms_per_search": round(t_s * 1e3, 2), "achieved": round(gbps, 1), "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                      "frac": round(gbps / PEAK_HBM_GBPS, 4), "bound": "hbm", "kernel": "dense_stream_kernel (exact fp32, D read once)",
                      "batch_invariant_mode": {"ms_per_search": round(t_inv * 1e3, 2), "achieved": round(n_local * H * 4 / t_inv / 1e9, 1),
                                               "bits_equal_the_rows_of_the_full_batch_search": same_as_big}})

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import scoring as SC
        # Two FIXED thread shapes, median of three runs each (round 4's "thread count with the best sgemm rate" flipped between
        # boxes: 12.6 vs 50.9 queries/s): 32 threads - the count BASELINE.json and README.md:89-94 name - is the reported value,
        # every hardware thread of the host is given beside it.  ~25 s of host work.
        ns, nqs = min(n_local, 1_200_000), min(args.n_queries, 1024)
        Dh = D[:ns].cpu().numpy()
        Qh = encode_queries()[:nqs].cpu().numpy()
        cores = os.cpu_count() or 1
        shapes = {}
        for nt in sorted({min(32, cores), cores}):
            SC.flat_ip_search_blas_heap(Qh[:256], Dh[:65536], min(args.topk, 1000), threads=nt)       # warm-up: BLAS thread pool, page faults
            runs_ = []
            for _ in range(3):
                st = {}
                tc = time.perf_counter()
                cs, ci = SC.flat_ip_search_blas_heap(Qh, Dh, args.topk, stats=st, threads=nt)
                tc = time.perf_counter() - tc
                runs_.append((tc, st))
            runs_.sort(key=lambda r_: r_[0])
            tc, st = runs_[1]
            shapes[nt] = {"threads": nt, "value": round(nqs / (tc * args.n_docs / ns), 3), "seconds": [round(r_[0], 2) for r_ in runs_],
                          "sgemm_gflops": round(st["sgemm_gflops"], 1), "sgemm_s": round(st["sgemm_s"], 2), "heap_s": round(st["heap_s"], 2)}
            log("[cpu_baseline]", shapes[nt])
        try:
            host = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
        except Exception:
            host = "unknown"
        main_shape = shapes[min(32, cores)]
        cpu = {"value": main_shape["value"], "unit": "queries/s", "cores": main_shape["threads"], "kind": "port", "host_cpu": host,
               "host_hardware_threads": cores, "sgemm_gflops": main_shape["sgemm_gflops"],
               "all_hardware_threads": shapes[cores] if cores != main_shape["threads"] else None,
               "sample_short": f"score stage only: faiss flat-IP port (host sgemm + heap, oracle/score_cpu.c), {nqs} queries x {ns} docs, "
                               f"median of 3, scaled to {args.n_docs} docs",
               "sample": f"scoring stage only (oracle.scoring.flat_ip_search_blas_heap = faiss IndexFlatIP.search as faiss-cpu runs it: "
                         f"host BLAS (OpenBLAS behind numpy) sgemm over (query block, database block) pairs + one heap per query in C / OpenMP, "
                         f"oracle/score_cpu.c): {nqs} queries x {ns} docs x {H}, top-{args.topk}, {main_shape['threads']} threads, median of 3 runs "
                         f"({main_shape['seconds']} s); extrapolated linearly to {args.n_docs} docs; query encoding not included",
               "note": f"fixed thread shapes, no search for the fastest one; the host BLAS is the limiter (sgemm at {main_shape['sgemm_gflops']:.0f} GFLOP/s, "
                       f"a few per cent of this host's fp32 peak) - a reported baseline, not a target"}
        del Dh, cs, ci


    # ---- query-encode regimes side by side (one pass each, synchronised): the headline runs the fp32 regime ----
    def timed_query_encode(prec):
        model.base_model.precision = prec
        try:
            encode_queries()
            torch.cuda.synchronize()
            tq = time.perf_counter()
            encode_queries()
            torch.cuda.synchronize()
            return round((time.perf_counter() - tq) * 1e3, 1)
        finally:
            model.base_model.precision = "auto"
    breakdown["query_encode_ms_bf16_regime"] = timed_query_encode("bf16")
    n_prod = {16: 3, 3: 6, 2: 3}[model.base_model.fp32_planes]
    breakdown["query_encode_regime"] = ("fp32 (reference: no autocast, eval_dense.py:94-106): every GEMM on "
                                        + ("two fp16 planes of power-of-two scaled rows per operand, 3 plane products"
                                           if model.base_model.fp32_planes == 16 else f"bf16 planes, {n_prod} plane products")
                                        + ", fp32 accumulate; fp32 attention")
    q_tokens = int(q_lens.sum())
    breakdown["query_encode_mfma_TFLOPs"] = round(q_tokens * FLOP_PER_TOKEN_1B * cfg["num_hidden_layers"] / 16 * n_prod / t_enc / 1e12, 1)

    # ---- the certified filter on corpora shaped like real embeddings (full shape, search stage only): the headline's data is
    #      isotropic Gaussian with near-constant norms; real LlamaBiDense vectors are anisotropic, their norms spread, and MS MARCO
    #      holds near-duplicate passages.  Every leg is compared with the exact kernel on all queries. ----
    # ---- BASELINE.json configs[3] on one GPU: what ONE of 8 doc shards does per search, with and without the threshold exchange
    #      (sr_dense_search_begin / _finish: every shard re-scores only what can reach the global top-k) ----
    shard_leg = None
    if filtered and world == 1 and not args.no_shard_leg:
        from scaling_retriever_amd.scoring import topk_merge
        index.close()
        index = None
        W8 = 8
        per = (n_local + W8 - 1) // W8
        shards = []
        for r_ in range(W8):
            ix = DenseIndexHIP(H, device=device)
            ix.set_precision("fp32_filtered")
            ix.add_device_rows(D[r_ * per:min(n_local, (r_ + 1) * per)], id_base=r_ * per, id_stride=1)
            shards.append(ix)
        lowers = torch.stack([ix.search_begin(reps_b, args.topk, W8) for ix in shards])
        thr_g = lowers.min(0).values
        outs = [ix.search_finish(reps_b, args.topk, thr_g) for ix in shards]
        ms_, mi_ = topk_merge(torch.stack([o[0] for o in outs]), torch.stack([o[1] for o in outs]))
        merged_ok = None
        if parity_ref is not None:
            merged_ok = bool(torch.equal(ms_, parity_ref[0]) and torch.equal(mi_, parity_ref[1]))
            assert merged_ok, "shard_1of8: the merge of the 8 shard results differs from the single index"
        kept = float(torch.stack([(o[1] >= 0).sum(1) for o in outs]).float().mean())

        def t_of(fn, n=3):
            fn()
            torch.cuda.synchronize()
            t0_ = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0_) / n
        t_plain = t_of(lambda: shards[0].search(reps_b, args.topk))
        t_begin = t_of(lambda: shards[0].search_begin(reps_b, args.topk, W8))

        def both():
            shards[0].search_begin(reps_b, args.topk, W8)
            shards[0].search_finish(reps_b, args.topk, thr_g)
        t_both = t_of(both)
        shard_leg = {"workload": f"one of 8 doc shards ({per} x {H}) of the headline corpus, {args.n_queries} queries, top-{args.topk}",
                     "search_ms_plain": round(t_plain * 1e3, 1), "search_ms_with_threshold_exchange": round(t_both * 1e3, 1),
                     "of_which_candidates_pass_ms": round(t_begin * 1e3, 1),
                     "mean_candidates_returned_per_query_and_shard": round(kept, 1),
                     "merge_of_8_shards_equals_single_index": merged_ok,
                     "note": "the all-reduce(min) of nq floats between the two halves is not in these times (one GPU here); "
                             "scaling over RCCL stays unmeasured on this pool"}
        log("[shard_1of8]", shard_leg)
        for ix in shards:
            ix.close()
        del shards, outs, lowers

    robustness = None
    if filtered and world == 1 and not args.no_robustness:
        import synth
        robustness = []
        if index is not None:
            index.close()
        for corpus, queries in (("aniso", "aniso"), ("aniso_dup", "aniso"), ("aniso_dup", "near_docs")):
            if not robustness or robustness[-1]["corpus"] != corpus:
                del D
                torch.cuda.empty_cache()
                D = synth.dense_rows(corpus, n_local, H, device, seed=11)
            Qr = synth.dense_queries(queries, args.n_queries, H, device, seed=12, D=D)
            ex_i = DenseIndexHIP(H, device=device)
            ex_i.add_device_rows(D)
            es, ei = ex_i.search(Qr, args.topk)
            ex_i.close()
            fi_i = DenseIndexHIP(H, device=device)
            fi_i.set_precision("fp32_filtered")
            fi_i.add_device_rows(D)
            fs, fi = fi_i.search(Qr, args.topk)
            same = bool(torch.equal(es, fs) and torch.equal(ei, fi))
            if not same:
                bad = (~((fs == es).all(1) & (fi == ei).all(1))).nonzero()[:, 0]
                log(f"[filter_robustness] {corpus}/{queries}: {bad.numel()} queries differ: {bad.tolist()[:16]}; stats {fi_i.filter_stats()} {fi_i.filter_query_stats()}")
                for q_ in bad.tolist()[:4]:
                    d_ = ((fs[q_] != es[q_]) | (fi[q_] != ei[q_])).nonzero()[:, 0]
                    j_ = int(d_[0])
                    log(f"   q {q_}: first diff at rank {j_} of {d_.numel()}: filtered ({float(fs[q_, j_])}, {int(fi[q_, j_])}) exact ({float(es[q_, j_])}, {int(ei[q_, j_])}); "
                        f"exact id in the filtered list: {bool((fi[q_] == ei[q_, j_]).any())}; filtered id in the exact list: {bool((ei[q_] == fi[q_, j_]).any())}")
                fs2, fi2 = fi_i.search(Qr, args.topk)
                log(f"   second filtered search equals exact: {bool(torch.equal(es, fs2) and torch.equal(ei, fi2))}; equals the first filtered: {bool(torch.equal(fs, fs2) and torch.equal(fi, fi2))}")
                ex2 = DenseIndexHIP(H, device=device)
                ex2.add_device_rows(D)
                es2, ei2 = ex2.search(Qr, args.topk)
                log(f"   second exact search equals the first exact: {bool(torch.equal(es, es2) and torch.equal(ei, ei2))}; equals the filtered: {bool(torch.equal(fs, es2) and torch.equal(fi, ei2))}")
                ex2.close()
            assert same, f"filter_robustness {corpus}/{queries}: results differ from the exact kernel"
            c0, r0_ = fi_i.filter_query_stats()
            torch.cuda.synchronize()
            tr = time.perf_counter()
            for _ in range(2):
                fi_i.search(Qr, args.topk)
            torch.cuda.synchronize()
            tr = (time.perf_counter() - tr) / 2
            c1, r1_ = fi_i.filter_query_stats()
            nf_, nb_ = fi_i.filter_stats()
            robustness.append({"corpus": corpus, "queries": queries, "search_queries_per_s": round(args.n_queries / tr, 1),
                               "search_ms": round(tr * 1e3, 1), "queries_certified_per_search": int((c1 - c0) // 2),
                               "queries_redone_by_exact_kernel_per_search": int((r1_ - r0_) // 2),
                               "searches_through_filter": int(nf_), "searches_with_queries_redone_by_exact_kernel": int(nb_),
                               "bit_identical_to_exact_kernel": same})
            log("[filter_robustness]", robustness[-1])
            fi_i.close()
            del es, ei, fs, fi, Qr
        index = None

    # everything below needs the HBM the corpus matrix holds
    if index is not None:
        index.close()
    del index, D, reps_b
    torch.cuda.empty_cache()

    # ---- stage 2 of the metric: passages/s of the corpus-encode task, through store_embs ----
    encode = None
    if not args.no_encode:
        encode = encode_leg(args, cfg, model, device, rank, world, share_gpu)

    # ---- BASELINE.json configs[2]: sparse inverted-index scoring (single GPU) ----
    sparse = None
    if rank == 0 and world == 1 and not args.no_sparse:
        del model
        torch.cuda.empty_cache()
        sparse = sparse_leg(args, device)
        if not args.no_sparse_index:
            torch.cuda.empty_cache()
            sparse["sparse_index"] = sparse_index_leg(args, device)
    if world > 1 and not args.no_sparse:                  # configs[2] doc-sharded over the ranks (every rank takes part)
        del model
        torch.cuda.empty_cache()
        sparse = sparse_sharded_leg(args, device, rank, world, share_gpu)
    config5 = None
    if rank == 0 and world == 1 and not args.no_config5 and not args.layers:
        torch.cuda.empty_cache()
        config5 = config5_leg(args, device)

    # self-check of a multi-GPU run: what torch.distributed really set up, and what every rank holds
    ranks_info = {"world_size": world, "backend": dist.get_backend() if world > 1 else None, "shard_docs": [n_local],
                  "queries_encoded_per_rank": [q_rows[1] - q_rows[0]]}
    if world > 1:
        t = torch.tensor([n_local, q_rows[1] - q_rows[0], torch.cuda.current_device()], dtype=torch.int64, device="cpu" if share_gpu else device)
        got = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(got, t)
        ranks_info.update({"shard_docs": [int(x[0]) for x in got], "queries_encoded_per_rank": [int(x[1]) for x in got],
                           "cuda_device_of_rank": [int(x[2]) for x in got]})
        assert sum(ranks_info["shard_docs"]) == args.n_docs and sum(ranks_info["queries_encoded_per_rank"]) == args.n_queries
    if rank == 0:
        res = {
            "metric": "MSMARCO-Dev queries/sec end-to-end (query encode + dense brute-force top-1000)",
            "value": round(args.n_queries * args.steps / dt, 2), "unit": "queries/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 2), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "Lion-DS-1B dense: 6980 Dev queries encoded (HIP LlamaBiDense, 1B dims, random init, fp32 regime as the reference: "
                                   "split-plane GEMMs with the error of an fp32 GEMM, fp32 attention) "
                                   f"+ brute-force fp32 top-{args.topk} over {args.n_docs} x {H} passage embeddings resident in HBM",
                       "n_docs": args.n_docs, "n_queries": args.n_queries, "hidden": H, "topk": args.topk,
                       "query_batch": args.query_batch, "layers": cfg["num_hidden_layers"],
                       "query_encode_precision": "fp32 regime (two fp16 planes of power-of-two scaled rows per operand, 3 products, fp32 accumulate: "
                                                 "the error of an fp32 GEMM)",
                       "doc_encode_precision": "bf16 autocast regime", "score_precision": "exact fp32 (k-ordered fmaf chain)" + ("" if args.exact_kernel else ": certified fp16 upper-bound filter + exact re-score of 1-2k candidates per query, "
                                                                               "bit-identical to the exact kernel (parity field)"),
                       "ranks": ranks_info,
                       # the figures of the other legs a reader looks for first, where a record that keeps only `config` still has them
                       "drop_in_retrieval_qps": (drop_in or {}).get("retrieval_task_with_run_json", {}).get("queries_per_s") if drop_in else None,
                       "drop_in_search_knn_qps": (drop_in or {}).get("search_knn", {}).get("queries_per_s") if drop_in else None,
                       "sparse_qps": sparse["value"] if sparse else None,
                       "sparse_qps_exact_kernels": (sparse.get("exact_kernels") or {}).get("queries_per_s") if sparse else None,
                       "sparse_index_passages_per_s": (sparse.get("sparse_index") or {}).get("passages_per_s") if sparse else None,
                       "sparse_index_build_postings_per_s": (sparse.get("index_build") or {}).get("postings_per_s") if sparse else None,
                       "encode_passages_per_s": (encode or {}).get("value") if encode else None,
                       "parallelism": f"doc-shard x{world}" + (" (queries encoded 1/W per rank + all-gather of the 57 MB query matrix; "
                                                                       "1 RCCL gather of per-shard top-k; merge on rank 0)" if world > 1 else "")},
            "roofline": roofline, "cpu_baseline": cpu, "parity": parity, "exact_kernel_mode": exact_mode, "breakdown": breakdown, "small_batch": small, "fp32_class_mode": fp32_class, "fast_mode": fast,
            "drop_in": drop_in, "shard_1of8": shard_leg, "filter_robustness": robustness, "encode": encode, "sparse": sparse, "config5_8b": config5,
        }
        emit(res)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
