"""CPU: the committed bench line (profiles/r04_bench_line.json, `python bench.py --steps 20 --warmup 5` on one MI355X) carries what
the driver's contract asks of bench.py's one JSON line: the metric fields, `roofline` of the dominant kernel with a PMC traffic
figure, `cpu_baseline`, and the legs DESIGN.md section 6 describes.  Guards the line's shape, not its numbers."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_bench_line_has_the_contract_fields():
    with open(os.path.join(ROOT, "profiles", "r04_bench_line.json")) as f:
        text = f.read().strip()
    assert "\n" not in text                                     # ONE line
    d = json.loads(text)
    for key, typ in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                     ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str), ("config", dict)):
        assert isinstance(d[key], typ), key
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"] and d["config"]["ranks"]["world_size"] == 1
    assert abs(d["value"] - d["config"]["n_queries"] / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-3
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and r["traffic"] > 0
    # the achieved figure is algorithmic work per launch over the measured launch duration
    assert abs(r["flop_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e12 - r["achieved"]) / r["achieved"] < 1e-2
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["sample"] and c["unit"] == d["unit"]
    assert "bit-identical" in d["parity"]
    for leg in ("exact_kernel_mode", "drop_in", "encode", "sparse", "config5_8b", "shard_1of8", "filter_robustness", "small_batch"):
        assert d.get(leg), leg
    assert d["sparse"]["roofline"]["traffic"] > 0 and len(d["sparse"]["sparse_sweep"]["rows"]) >= 12
    assert all(row["queries_bit_exact_vs_oracle"] >= 64 for row in d["sparse"]["sparse_sweep"]["rows"])
    assert d["drop_in"]["generate_query_vecs"]["bit_identical_to_one_call_per_batch"] is True
    assert d["encode"]["padded_batch_128_mode"]["passages_per_s"] > 0
