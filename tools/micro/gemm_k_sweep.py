"""Fixed (X) vs per-k-step (s) cost of the 256x256 GEMM tile: time at M = N = 8192 (1024 tiles = 4 full rounds) over K."""
import os, sys
import json
os.environ["SR_DEV_SWITCHES"] = "1"   # the library reads its development switches only with this set
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from scaling_retriever_amd import _lib as L
lib = L.load()
os.environ["SR_GEMM_TILE"] = sys.argv[1] if len(sys.argv) > 1 else "256"
M = N = 8192
EPI = int(sys.argv[2]) if len(sys.argv) > 2 else 0
for K in (256, 512, 1024, 2048, 4096, 8192):
    g = torch.Generator(device="cuda").manual_seed(0)
    A = torch.randn((M, K), device="cuda", generator=g).bfloat16(); W = (torch.randn((N, K), device="cuda", generator=g) * 0.02).bfloat16()
    C = torch.zeros((M, N), dtype=torch.float32, device="cuda")
    f = lambda: L.check(lib.sr_gemm_bf16(A.data_ptr(), W.data_ptr(), M, N, K, EPI, C.data_ptr(), None, L.stream_ptr()))
    for _ in range(3): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): f()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 10
    print(json.dumps({"K": K, "ms": round(ms, 4), "TF": round(2.0 * M * N * K / ms / 1e9, 1), "us_per_tile_round": round(ms * 1e3 / 4, 2),
                      "us_per_kstep": round(ms * 1e3 / 4 / (K / 64), 3)}), flush=True)
