"""Reference point only (not used by the product): what the vendor library (hipBLASLt through torch.matmul) reaches on the
encoder's GEMM shapes on this box, next to tools/quick_gemm_bench.py."""
import sys, json, torch
Ms = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "9600,38400").split(",")]
for M in Ms:
    row = {}
    for name, N, K in (("qkv", 3072, 2048), ("o", 2048, 2048), ("gate_up", 16384, 2048), ("down", 2048, 8192)):
        A = torch.randn((M, K), device="cuda").bfloat16(); W = (torch.randn((N, K), device="cuda") * 0.02).bfloat16()
        for _ in range(3): torch.matmul(A, W.T)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20): torch.matmul(A, W.T)
        b.record(); torch.cuda.synchronize()
        row[name] = round(2.0 * M * N * K / (a.elapsed_time(b) / 20) / 1e9, 1)
    print(json.dumps({"M": M, "hipBLASLt_TF": row}), flush=True)
