#!/usr/bin/env python3
"""Regenerates the measurement table of DESIGN.md section 6 from a bench record (the full one: gpurun_out/bench_detail.json, kept as profiles/r06_bench_detail.json), so that the document cannot drift from the numbers:
python tools/design_table.py profiles/r06_bench_detail.json  (rewrites the block between the BEGIN / END markers in DESIGN.md)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r06_bench_detail.json")
r = json.load(open(src))
rf, bd = r["roofline"], r["breakdown"]
rows = []


def row(a, b):
    rows.append(f"| {a} | {b} |")


row("queries/s end to end at the reference's precision: fp32-regime query encode + exact fp32 top-1000 (headline)",
    f"**{r['value']:.0f}** ({r['ms_per_step']:.1f} ms per step over {r['steps']} steps: query encode {bd['query_encode_ms']:.0f} ms, search {bd['search_ms']:.0f} ms; "
    f"ids and fp32 scores bit-identical to the exact kernel on the full problem in the same run: `parity`)")
row("dominant kernel `dense_split_kernel<true>` (the certified filter's upper-bound pass, fp16 MFMA)",
    f"{rf['achieved']:.0f} TFLOP/s = **{rf['frac']:.3f} of the 16-bit MFMA peak**; {rf['avg_launch_ms']:.4f} ms per launch by HIP events, {rf['launches'] // r['steps']} launches per search = "
    f"{100 * rf['kernel_share_of_step']:.0f} % of the step; {rf['queries_certified']} queries certified, {rf['queries_redone_by_exact_kernel']} re-done; "
    f"traffic beyond L2 per launch: {(rf['traffic'] or 0) / 1e9:.2f} GB (`profiles/r06_pmc_traffic.json`)")
em = r.get("exact_kernel_mode") or {}
if em:
    row("the same step through the exact fp32 MFMA kernel (`exact_kernel_mode`: the data-independent floor)",
        f"{em['value']:.0f} queries/s; {em['roofline']['achieved']:.1f} TFLOP/s = **{em['roofline']['frac']:.3f} of the fp32 MFMA peak** ({em['roofline']['avg_launch_ms']:.3f} ms per launch)")
row("query encode, fp32 regime", f"{bd['query_encode_ms']:.0f} ms for {bd['query_tokens']} tokens = {bd['query_encode_mfma_TFLOPs']:.0f} TFLOP/s of fp16 MFMA work (3 plane products per fp32 product); "
    f"bf16 regime: {bd['query_encode_ms_bf16_regime']:.0f} ms")
di = r.get("drop_in")
if di:
    g, kn, td, rt = di["generate_query_vecs"], di["search_knn"], di["get_top_docs"], di["retrieval_task_with_run_json"]
    row("drop-in: `generate_query_vecs` over 55 loader batches of 128 (`eval_dense.py:94-106`)",
        f"{g['ms']} ms coalesced ({g['one_query_encode_call_per_loader_batch_ms']} ms with one call per batch); same bits: {g['bit_identical_to_one_call_per_batch']}")
    row("drop-in: `DenseFlatIndexer.search_knn` → list of lists of db ids (`indexer.py:210-214`)",
        f"**{kn['queries_per_s']:.0f} queries/s** ({kn['ms']} ms; one search + D2H {kn.get('search_arrays_ms')} ms, the 7 M-object id mapping alone {kn.get('id_mapping_alone_ms')} ms)")
    row("drop-in: `get_top_docs` (encode + `search_knn`)", f"{td['queries_per_s']:.0f} queries/s ({td['ms']} ms)")
    row("drop-in: retrieval task incl. `run.json` (`eval_dense.py:225-241`)", f"**{rt['queries_per_s']:.0f} queries/s** ({rt['ms']} ms, {rt['run_json_bytes'] / 1e6:.0f} MB file)")
for m, name in (("fp32_class_mode", "bf16x6"), ("fast_mode", "bf16x3")):
    if r.get(m) and "value" in r[m]:
        row(f"{name} score MODE (fp32-class scores, not bit-identical; same query encode)", f"{r[m]['value']:.0f} queries/s, kernel at {r[m]['roofline']['frac']:.3f} of bf16 MFMA peak")
for x in r.get("filter_robustness") or []:
    row(f"filter on the `{x['corpus']}` corpus, `{x['queries']}` queries (search stage, full shape)",
        f"{x['search_queries_per_s']:.0f} queries/s ({x['search_ms']:.0f} ms); {x['queries_certified_per_search']} certified, {x['queries_redone_by_exact_kernel_per_search']} re-done by the exact kernel; "
        f"bit-identical to the exact kernel: {x['bit_identical_to_exact_kernel']}")
sh = r.get("shard_1of8")
if sh:
    row("one of 8 doc shards (`shard_1of8`)", f"{sh['search_ms_plain']} ms per search alone, **{sh['search_ms_with_threshold_exchange']} ms** with the threshold exchange "
        f"({sh['mean_candidates_returned_per_query_and_shard']} candidates returned per query instead of {r['config']['topk']}); merge of the 8 == the single index: {sh['merge_of_8_shards_equals_single_index']}")
e = r.get("encode")
if e:
    row("corpus encode through `store_embs`, Lion-DS-1B dims", f"**{e['value']:.0f} passages/s** ({e['sample_passages']} passages, budget {e['token_budget']} tokens, {e['wall_s']} s wall) = "
        f"{e['roofline']['achieved']:.0f} TFLOP/s = **{e['roofline']['frac']:.3f} of bf16 MFMA peak** ({e['roofline']['frac_gpu_time_only']:.3f} over GPU time only)"
        + (f"; the reference's padded-batch-128 loader: {e['padded_batch_128_mode']['passages_per_s']:.0f} passages/s" if e.get("padded_batch_128_mode") else ""))
c5 = r.get("config5_8b")
if c5:
    row("corpus encode, 8B dims (configs[4], one GPU)", f"{c5['encode']['value']:.0f} passages/s = {c5['encode']['roofline']['frac']:.3f} of peak; score stage over the GPU's 1/8 shard at H 4096 through the exact "
        f"kernel: {c5['score_shard']['queries_per_s']:.0f} queries/s ({c5['score_shard']['roofline']['frac']:.3f} of fp32 MFMA peak); one query {c5['score_shard_one_query']['ms']} ms "
        f"({c5['score_shard_one_query']['roofline']['achieved'] / 1e3:.2f} TB/s)"
        + (f"; the same shard through the certified filter: **{c5['score_shard_filtered']['queries_per_s']:.0f} queries/s** ({c5['score_shard_filtered']['ms']} ms, "
           f"{c5['score_shard_filtered']['queries_redone_by_exact_kernel']} queries re-done), bit-identical to the exact kernel" if c5.get("score_shard_filtered") else ""))
for sb in r.get("small_batch") or []:
    row(f"small-batch dense score, nq = {sb['nq']} (HBM-bound, `dense_stream_kernel`)", f"{sb['achieved'] / 1e3:.2f} TB/s = {sb['frac']:.2f} of the 8 TB/s spec ({sb['ms_per_search']} ms per pass)")
sp = r.get("sparse")
if sp:
    b, rf2, ex = sp["bounds"], sp["roofline"], sp.get("exact_kernels") or {}
    row("sparse scoring, full MSMARCO shape (configs[2])", f"**{sp['value']:.0f} queries/s** ({sp['ms_per_pass']} ms per pass: `cert_score_kernel` {rf2['kernel_ms_per_pass']} ms in {rf2['launches']} launches, "
        f"the other kernels {rf2.get('other_kernels_ms_per_pass')} ms; {sp['path']['queries_redone_by_the_exact_kernels_per_pass']} queries re-done by the exact kernels); {sp['parity']}; "
        + (f"the exact kernels alone: {ex['queries_per_s']:.0f} queries/s (kernels {ex['kernel_ms_per_pass']} ms), same bits: {ex['same_bits_as_the_product_path']}" if ex else ""))
    m, l2, ld = b["mfma_heavy_terms"], b["l2_matrix_operand"], b["lds_rare_postings"]
    row("... what the scorer does, counted on the device", f"heavy terms on the matrix pipe {m['flop_per_pass']:.3g} flop: MFMA floor {m['floor_ms_per_pass']} ms; matrix operand out of L2 {l2['bytes_per_pass']:.3g} B: "
        f"L2 floor {l2['floor_ms_per_pass']} ms; rare postings {ld['adds_per_pass']:.3g} LDS adds: LDS floor {ld['floor_ms_per_pass']} ms; sum of floors {b['sum_of_floors_ms']} ms, "
        f"kernel / floors = **{b['kernel_over_sum_of_floors']}**")
    row("... HBM side of the sparse scorer", f"unique posting bytes per pass {rf2['unique_index_bytes_per_pass'] / 1e9:.1f} GB (all 6 980 queries in one batch; {rf2['unique_index_bytes_per_pass_in_batches_of_1024'] / 1e9:.1f} GB in "
        f"batches of 1 024 as round 4 ran them): HBM floor {rf2['hbm_floor_ms_per_pass']} ms, frac {rf2['frac']}"
        + (f"; traffic beyond L2 per pass {rf2['traffic'] / 1e9:.0f} GB (`profiles/r06_pmc_sparse_traffic.json`)" if rf2.get("traffic") else ""))
    ib = sp.get("index_build")
    if ib:
        row("CSR-by-term build (`sr_sparse_csr_build`), full MSMARCO shape", f"**{ib['postings_per_s'] / 1e9:.1f} G postings/s** ({ib['seconds'] * 1e3:.0f} ms for {sp['config']['postings']} postings, "
            f"{ib['roofline']['achieved']:.0f} GB/s of algorithmic bytes = {ib['roofline']['frac']:.3f} of HBM); bit-identical to the source index: {ib['bit_identical_to_the_source_index']}; "
            f"forward index by doc: {ib['forward_index_by_doc']['seconds'] * 1e3:.0f} ms")
    if sp.get("drop_in"):
        d2 = sp["drop_in"]
        row("drop-in: `SparseRetrieval.retrieve` incl. `run.json` + `q_stats.json` (`indexer.py:530-540`)",
            f"**{d2['retrieve']['queries_per_s']:.0f} queries/s** ({d2['retrieve']['ms']} ms: query encode {d2['generate_query_vecs_ms']} ms, search → RunResult {d2['search_to_RunResult_ms']} ms)")
    sw = sp.get("sparse_sweep")
    if sw:
        cells = "; ".join(f"{x['index']} {x['L0_d']}/{x['L0_q']}: {x['queries_per_s'] / 1e3:.1f} k"
                          + (f" ({x['queries_redone_by_the_exact_kernels']} of {sw['nq']} re-done by the exact kernels)" if x["queries_redone_by_the_exact_kernels"] else "") for x in sw["rows"])
        row("sparse sweep (index L0_d/L0_q: queries/s; 64 queries per cell bit-exact vs the C oracle)", cells)
    si = sp.get("sparse_index")
    if si:
        row("drop-in: `SparseIndexer.index` (encode + sparse head + compaction + CSR build)", f"**{si['passages_per_s']:.0f} passages/s** ({si['seconds']} s, L0_d {si['L0_d']}, {si['postings']} postings) = "
            f"{si['roofline']['achieved']:.0f} TFLOP/s = {si['roofline']['frac']:.3f} of bf16 MFMA peak over the wall time; MSMARCO extrapolation {si['msmarco_extrapolation_minutes']} min")
    cb = sp["cpu_baseline"]
    row("sparse CPU baseline", f"{cb['value']} queries/s with the reference's 32 threads (4 × 8); best shape on the host: {cb['best_shape_on_this_host']['value']} with {cb['best_shape_on_this_host']['threads']} threads")
cb = r.get("cpu_baseline")
if cb:
    al = cb.get("all_hardware_threads")
    row("dense CPU baseline (faiss's algorithm: BLAS sgemm blocks + a heap per query)", f"**{cb['value']} queries/s** with {cb['cores']} threads on {cb.get('host_cpu')} (median of 3 runs, sgemm at {cb.get('sgemm_gflops')} GFLOP/s)"
        + (f"; with all {al['threads']} hardware threads: {al['value']} queries/s (runs {al['seconds']} s)" if al else ""))
table = "\n".join([f"| r06, 1 × MI355X (`{os.path.relpath(src, ROOT)}`) | value |", "|---|---|"] + rows)
p = os.path.join(ROOT, "DESIGN.md")
s = open(p).read()
B, E = "<!-- BEGIN r06 table (tools/design_table.py) -->", "<!-- END r06 table -->"
if B in s:
    s = s[:s.index(B) + len(B)] + "\n" + table + "\n" + s[s.index(E):]
    open(p, "w").write(s)
print(table)
