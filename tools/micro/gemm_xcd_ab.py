"""A/B of the GEMM's XCD-aware tile order (dev switch SR_GEMM_XCD) on the four 1B-layer shapes: TFLOP/s per shape and per layer.
python tools/micro/gemm_xcd_ab.py [tokens,tokens,...]"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import quick_gemm_bench as qb  # noqa: E402  (sets SR_DEV_SWITCHES)

Ms = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "16384,38400").split(",")]
shapes = [("qkv", 3072, 2048, 0), ("o", 2048, 2048, 0), ("gate_up", 16384, 2048, 2), ("down", 2048, 8192, 0)]
os.environ["SR_GEMM_TILE"] = ""
for M in Ms:
    for xcd in ("0", "1", "0", "1", "0", "1"):     # alternated and repeated: the first configuration measured runs on a cold part
        os.environ["SR_GEMM_XCD"] = xcd
        tot_ms, tot_fl, row = 0.0, 0.0, {}
        for name, N, K, epi in shapes:
            ms, tf = qb.run(M, N, K, epi)
            row[name] = round(tf, 1)
            tot_ms += ms
            tot_fl += 2.0 * M * N * K
        print(json.dumps({"M": M, "xcd_order": int(xcd), "TF": row, "layer_TF": round(tot_fl / tot_ms / 1e9, 1)}), flush=True)
