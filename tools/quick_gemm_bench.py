"""Ad-hoc GEMM timing (development aid): the four 1B-layer GEMM shapes at a given token count, forced tilings vs the automatic tile plan."""
import os, sys
import json
os.environ["SR_DEV_SWITCHES"] = "1"   # the library reads its development switches only with this set
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from scaling_retriever_amd import _lib
L = _lib; lib = L.load()
def run(M, N, K, epi, iters=20):
    g = torch.Generator(device="cuda").manual_seed(0)
    A = torch.randn((M, K), device="cuda", generator=g).bfloat16()
    W = (torch.randn((N, K), device="cuda", generator=g) * 0.02).bfloat16()
    C = torch.zeros((M, N), dtype=torch.float32, device="cuda")
    def f(): L.check(lib.sr_gemm_bf16(A.data_ptr(), W.data_ptr(), M, N, K, epi, C.data_ptr(), None, L.stream_ptr()))
    for _ in range(3): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): f()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / iters
    return ms, 2.0 * M * N * K / ms / 1e9
if __name__ == "__main__":
    Ms = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "9600,38400").split(",")]
    shapes = [("qkv", 3072, 2048, 0), ("o", 2048, 2048, 0), ("gate_up", 16384, 2048, 2), ("down", 2048, 8192, 0)]
    for M in Ms:
        for tile in ("128", "256", "auto"):
            for persist in ("1",):
                os.environ["SR_GEMM_TILE"] = "" if tile == "auto" else tile; os.environ["SR_GEMM_PERSIST"] = persist
                tot_ms, tot_fl, row = 0, 0, {}
                for name, N, K, epi in shapes:
                    ms, tf = run(M, N, K, epi)
                    row[name] = round(tf, 1); tot_ms += ms; tot_fl += 2.0 * M * N * K
                print(json.dumps({"M": M, "tile": tile, "persist": persist, "TF": row, "layer_TF": round(tot_fl / tot_ms / 1e9, 1)}), flush=True)
