#!/usr/bin/env python3
"""Contract benchmark: MSMARCO-Dev queries/s end-to-end on the Lion-DS-1B dense configuration
(BASELINE.json configs[1]; configs[3] for --gpus > 1).

One step = one pass of the hot path over the whole Dev query set: encode 6 980 synthetic queries with
the HIP LlamaBiDense encoder (1B dims, random weights), score them against the doc-sharded fp32
embedding matrix resident in HBM (8 841 823 x 2048, synthetic), fused top-1000, and - for N > 1 - one
RCCL gather of the per-shard top-k + merge on rank 0.  value = queries / s over the whole job.

Also reported on the same JSON line: `roofline` for the dominant kernel (dense_score_kernel, fp32 MFMA
bound; durations from HIP events recorded around every launch inside the timed region), `encode`
(passages/s of doc_encode on a sample, MFMA bf16 bound), `cpu_baseline` (the oracle's faiss-style
flat IP search on the host cores, bounded sample, rank 0 at N = 1 only).

  python bench.py --gpus 1 --steps 3 --warmup 1
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus 8 --steps 3 --warmup 1
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

LION_1B = dict(vocab_size=128256, hidden_size=2048, intermediate_size=8192, num_hidden_layers=16,
               num_attention_heads=32, num_key_value_heads=8, head_dim=64, rms_norm_eps=1e-5, rope_theta=500000.0,
               tie_word_embeddings=True,
               rope_scaling={"rope_type": "llama3", "factor": 32.0, "low_freq_factor": 1.0, "high_freq_factor": 4.0,
                             "original_max_position_embeddings": 8192})
PEAK_F32_MFMA_TF = 157.3     # MI355X_MICROARCH.md: fp32-in MFMA = fp32 vector peak
PEAK_BF16_MFMA_TF = 2500.0   # dense bf16 MFMA peak
PEAK_HBM_GBPS = 8000.0       # HBM3E spec peak (about 6300 GB/s is what a streaming copy reaches)
FLOP_PER_TOKEN_1B = 1.946e9  # 2 x linear params of the 1B body


def log(*a):
    print(*a, file=sys.stderr, flush=True)


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n-docs", type=int, default=8_841_823)
    ap.add_argument("--n-queries", type=int, default=6980)
    ap.add_argument("--topk", type=int, default=1000)
    ap.add_argument("--query-batch", type=int, default=2048, help="queries per query_encode call (eval_batch_size; the reference script uses 128)")
    ap.add_argument("--encode-batches", type=int, default=16, help="passage batches (x128) for the encode figure")
    ap.add_argument("--layers", type=int, default=None, help="override num layers (debug only; invalidates the number)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fast-mode", action="store_true", help="skip the extra bf16x3 / bf16x6 precision-mode measurements")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        log(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE")
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
    from scaling_retriever_amd.distributed import all_gather_query_reps, gather_topk, query_slice, shard_size
    from scaling_retriever_amd.modeling.llm_encoder import LlamaBiDense
    from scaling_retriever_amd.scoring import DenseIndexHIP, topk_merge
    lib = _lib.load()

    cfg = dict(LION_1B)
    if args.layers:
        cfg["num_hidden_layers"] = args.layers
    H = cfg["hidden_size"]
    t_setup = time.time()
    model = LlamaBiDense.from_weights(cfg, random_weights(cfg, device, seed=0), max_batch_tokens=65536, max_batch_seqs=4096).to(device).eval()
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
    torch.cuda.synchronize()
    log(f"[rank {rank}] setup {time.time() - t_setup:.1f}s: {n_local} docs x {H} fp32 = {n_local * H * 4 / 1e9:.1f} GB resident; "
        f"{args.n_queries} queries, mean {q_lens.mean():.1f} tokens")

    def encode_queries():
        local = [model.query_encode(input_ids=i, attention_mask=m) for i, m in q_batches]
        local = torch.cat(local) if local else torch.zeros((0, H), dtype=torch.float32, device=device)
        return all_gather_query_reps(local, args.n_queries)

    def step():
        reps = encode_queries()
        s, i = index.search(reps, args.topk)
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

    # ---- the same step with the score kernel in split-bf16 arithmetic (opt-in precision modes of sr_dense_search;
    #      fp32 operands split into bf16 planes, 3 / 6 plane products on the bf16 MFMA pipe, fp32 accumulate) ----
    def timed_mode(mode, n_prod, note):
        try:
            index.set_precision(mode)
            step()
            _lib.check(lib.sr_dense_index_profile(index._h, 1))
            barrier()
            tf0 = time.perf_counter()
            for _ in range(args.steps):
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
            return {"precision": mode, "value": round(args.n_queries * args.steps / dtf, 2), "unit": "queries/s",
                    "ms_per_step": round(dtf / args.steps * 1e3, 2),
                    "roofline": {"kernel": f"dense_split_kernel (bf16 MFMA 16x16x32, {n_prod} products per fp32-equivalent FMA)", "bound": "mfma",
                                 "achieved": round(n_prod * eq_tf, 1), "peak": PEAK_BF16_MFMA_TF,
                                 "unit": f"TFLOP/s (bf16 MFMA work = {n_prod} x algorithmic)",
                                 "frac": round(n_prod * eq_tf / PEAK_BF16_MFMA_TF, 4), "fp32_equivalent_TFLOPs": round(eq_tf, 1),
                                 "launches": int(n_l.value), "avg_launch_ms": round(ms.value / max(1, n_l.value), 4)},
                    "note": note}
        except MemoryError as e:
            return {"precision": mode, "skipped": str(e)}
        finally:
            index.set_precision("fp32")

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
        small.append({"nq": nq_s, "ms_per_search": round(t_s * 1e3, 2), "achieved": round(gbps, 1), "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                      "frac": round(gbps / PEAK_HBM_GBPS, 4), "bound": "hbm", "kernel": "dense_stream_kernel (exact fp32, D read once)"})

    # ---- secondary figure: passages/s of doc_encode (same engine, doc-length batches) ----
    def encode_rate(batch):
        d_batches, d_lens = synth_batches(args.encode_batches * 128, batch, 4.25, 0.35, 8, 192, cfg["vocab_size"], 3, device)
        model.doc_encode(input_ids=d_batches[0][0], attention_mask=d_batches[0][1])
        torch.cuda.synchronize()
        te = time.perf_counter()
        for i_, m_ in d_batches:
            model.doc_encode(input_ids=i_, attention_mask=m_)
        torch.cuda.synchronize()
        te = time.perf_counter() - te
        tokens = int(d_lens.sum())
        L = cfg["num_hidden_layers"]
        flop = tokens * FLOP_PER_TOKEN_1B * L / 16 + 4.0 * float((d_lens.astype(np.float64) ** 2).sum()) * H * L
        return {"batch": batch, "passages_per_s_per_gpu": round(len(d_lens) / te, 1), "tokens_per_s_per_gpu": round(tokens / te, 1),
                "achieved_TFLOPs": round(flop / te / 1e12, 1), "frac_of_bf16_mfma_peak": round(flop / te / 1e12 / PEAK_BF16_MFMA_TF, 4),
                "mean_tokens_per_passage": round(float(d_lens.mean()), 1), "sample_passages": int(len(d_lens))}
    e128, e512 = encode_rate(128), encode_rate(512)
    encode = {"passages_per_s_per_gpu": e512["passages_per_s_per_gpu"], "dtype": "bf16 GEMM / fp32 accumulate",
              "reference_batch_128": e128, "batch_512": e512}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import scoring as SC
        ns, nqs = min(n_local, 1_200_000), min(args.n_queries, 1024)   # ~10 s of host work on the GPU box
        Dh = D[:ns].cpu().numpy()
        Qh = encode_queries()[:nqs].cpu().numpy()
        SC.flat_ip_search_fast(Qh[:64], Dh[:20000], min(args.topk, 1000))
        tc = time.perf_counter()
        SC.flat_ip_search_fast(Qh, Dh, args.topk)
        tc = time.perf_counter() - tc
        qps_full = nqs / (tc * args.n_docs / ns)
        cpu = {"value": round(qps_full, 3), "unit": "queries/s", "cores": os.cpu_count(), "kind": "port",
               "sample": f"scoring stage only (oracle.scoring.flat_ip_search_fast: numpy BLAS sgemm blocks + argpartition top-{args.topk}, "
                         f"the faiss IndexFlatIP algorithm): {nqs} queries x {ns} docs x {H} took {tc:.2f}s; "
                         f"extrapolated linearly to {args.n_docs} docs; query encoding not included"}

    if rank == 0:
        res = {
            "metric": "MSMARCO-Dev queries/sec end-to-end (query encode + dense brute-force top-1000)",
            "value": round(args.n_queries * args.steps / dt, 2), "unit": "queries/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 2), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "Lion-DS-1B dense: 6980 Dev queries encoded (HIP LlamaBiDense, 1B dims, random init, bf16 GEMMs) "
                                   f"+ brute-force fp32 top-{args.topk} over {args.n_docs} x {H} passage embeddings resident in HBM",
                       "n_docs": args.n_docs, "n_queries": args.n_queries, "hidden": H, "topk": args.topk,
                       "query_batch": args.query_batch, "layers": cfg["num_hidden_layers"],
                       "parallelism": f"doc-shard x{world}" + (" (queries encoded 1/W per rank + all-gather of the 57 MB query matrix; "
                                                                       "1 RCCL gather of per-shard top-k; merge on rank 0)" if world > 1 else "")},
            "roofline": roofline, "breakdown": breakdown, "small_batch": small, "fp32_class_mode": fp32_class, "fast_mode": fast, "encode": encode, "cpu_baseline": cpu,
        }
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
