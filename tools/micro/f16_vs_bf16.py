#!/usr/bin/env python3
"""Does the fp16 MFMA loop run slower than the bf16 one on this part?  The same GEMM (fp32 residual-add epilogue) through
sr_gemm_bf16 (epilogue 1) and sr_gemm_f16_scaled, same shapes, same random operands (unit scales), on the eight- and four-wave
256 x 256 loops (SR_GEMM_BIG), alternating, one box.  The certified filter's pass and the fp32-regime encoder GEMMs are fp16 loops."""
import json
import os
import sys

os.environ["SR_DEV_SWITCHES"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from scaling_retriever_amd import _lib as L  # noqa: E402

lib = L.load()
M = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
ITERS = 30


def timed(fn):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(ITERS):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / ITERS


g = torch.Generator(device="cuda").manual_seed(0)
for N, K in ((2048, 2048), (16384, 2048), (2048, 6144), (2048, 8192)):
    A = torch.randn((M, K), device="cuda", generator=g)
    W = torch.randn((N, K), device="cuda", generator=g) * 0.02
    Ab, Wb, Ah, Wh = A.bfloat16(), W.bfloat16(), A.half(), W.half()
    C = torch.zeros((M, N), dtype=torch.float32, device="cuda")
    ones_m, ones_n = torch.ones(M, device="cuda"), torch.ones(N, device="cuda")
    fl = 2.0 * M * N * K
    row = {"M": M, "N": N, "K": K}
    for loop in ("8w", "4w"):
        def bf():
            L.check(lib.sr_gemm_bf16(Ab.data_ptr(), Wb.data_ptr(), M, N, K, 1, C.data_ptr(), None, L.stream_ptr()))

        def hf():
            L.check(lib.sr_gemm_f16_scaled(Ah.data_ptr(), Wh.data_ptr(), M, N, K, ones_m.data_ptr(), ones_n.data_ptr(), C.data_ptr(), L.stream_ptr()))
        os.environ["SR_GEMM_BIG"] = loop
        best = {"bf16": 0.0, "f16": 0.0}
        for _ in range(3):
            best["bf16"] = max(best["bf16"], fl / timed(bf) / 1e9)
            best["f16"] = max(best["f16"], fl / timed(hf) / 1e9)
        row[loop] = {k: round(v, 1) for k, v in best.items()}
        row[loop]["f16_over_bf16"] = round(best["f16"] / best["bf16"], 3)
    os.environ.pop("SR_GEMM_BIG", None)
    print(json.dumps(row), flush=True)
