"""CPU: the contract line bench.py prints as the LAST line of stdout is what the driver can parse - compact (the round-5 line had
grown to 23 KB and the driver's record came back unparsed), strict JSON (no NaN / Infinity tokens anywhere on it), with the metric
fields, `roofline` of the dominant kernel and `cpu_baseline`.  Checked twice: on the line `bench.contract_line` makes from a full
record (profiles/r05_bench_line.json is round 5's full record), and on the line committed for the current round
(profiles/r06_bench_line.json, `python bench.py` on one MI355X).  Guards the line's shape, not its numbers."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CURRENT = os.path.join(ROOT, "profiles", "r06_bench_line.json")


def _reject_constant(name):
    raise ValueError(f"non-finite constant {name} on the contract line")


def check_contract_line(text):
    text = text.rstrip("\n")
    assert "\n" not in text                                     # ONE line
    assert len(text) < 8192, len(text)
    assert "Infinity" not in text and "NaN" not in text
    d = json.loads(text, parse_constant=_reject_constant)
    for key, typ in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                     ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str), ("config", dict)):
        assert isinstance(d[key], typ), key
    assert d["higher_is_better"] is True and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"] and d["config"]["ranks"]["world_size"] == d["n_gpus"]
    assert abs(d["value"] - d["config"]["n_queries"] / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-3
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["traffic"] is None or r["traffic"] > 0
    # the achieved figure is algorithmic work per launch over the measured launch duration
    assert abs(r["flop_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e12 - r["achieved"]) / r["achieved"] < 1e-2
    if d["n_gpus"] == 1:
        c = d["cpu_baseline"]
        assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["sample"] and c["unit"] == d["unit"]
        assert len(c["sample"]) <= 200
    assert "bit-identical" in d["parity"]
    return d


def test_contract_line_from_a_full_record_is_compact_and_strict():
    sys.path.insert(0, ROOT)
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        import bench
    finally:
        sys.argv = argv
    with open(os.path.join(ROOT, "profiles", "r05_bench_line.json")) as f:
        full = json.load(f)
    assert "Infinity" in json.dumps(full)                       # the record that broke the driver's parser in round 5
    full["breakdown"]["query_encode_ms"] = float("inf")         # and a non-finite number on top of it
    d = check_contract_line(bench.contract_line(full))
    assert d["breakdown"]["query_encode_ms"] is None
    assert d["sparse"]["qps"] == full["sparse"]["value"] and d["encode"]["frac"] == full["encode"]["roofline"]["frac"]
    assert d["sparse"]["sweep"]["cells"] == len(full["sparse"]["sparse_sweep"]["rows"])
    assert len(bench.contract_line(full)) < bench.CONTRACT_LINE_MAX <= 4096


@pytest.mark.skipif(not os.path.exists(CURRENT), reason="no bench line committed for this round yet")
def test_committed_bench_line_of_this_round():
    with open(CURRENT) as f:
        d = check_contract_line(f.read())
    assert d["n_gpus"] == 1
    for leg in ("breakdown", "exact_kernel", "drop_in", "encode", "sparse", "config5_8b", "small_batch"):
        assert d.get(leg), leg
    assert d["sparse"]["traffic"] is None or d["sparse"]["traffic"] > 0
    assert d["sparse"]["sweep"]["cells"] >= 12 and d["sparse"]["sweep"]["all_bit_exact"] is True
    assert d["encode"]["padded_128_passages_per_s"] > 0
