import os, sys, json
os.environ["SR_DEV_SWITCHES"] = "1"
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools"))
import quick_gemm_bench as qb
shapes = [("qkv", 3072, 2048, 0), ("o", 2048, 2048, 0), ("gate_up", 16384, 2048, 2), ("down", 2048, 8192, 0)]
for M in (16384,):
    for big in ("", "s3"):
        os.environ["SR_GEMM_BIG"] = big; os.environ["SR_GEMM_TILE"] = "256"
        row = {}
        tot_ms = tot_fl = 0
        for name, N, K, epi in shapes:
            ms, tf = qb.run(M, N, K, epi); row[name] = round(tf, 1); tot_ms += ms; tot_fl += 2.0 * M * N * K
        print(json.dumps({"M": M, "big": big or "256x256 2 stages", "TF": row, "layer_TF": round(tot_fl / tot_ms / 1e9, 1)}), flush=True)
