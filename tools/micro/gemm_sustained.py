"""Is the in-situ GEMM rate (1 150 TF in the encode pass) a sustained-load effect or a cold-operand effect?  The gate/up GEMM at 16 384 tokens:
(a) 20 back-to-back launches on the same operands (what tools/quick_gemm_bench.py times), (b) 4 000 of them (several seconds of load),
(c) 4 000 launches cycling over 16 different weight matrices and 4 activation buffers (operands from HBM every time, as in a 16-layer pass)."""
import os, sys, time
os.environ["SR_DEV_SWITCHES"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from scaling_retriever_amd import _lib as L
lib = L.load()
M, N, K, epi = 16384, 16384, 2048, 2
g = torch.Generator(device="cuda").manual_seed(0)
As = [torch.randn((M, K), device="cuda", generator=g).bfloat16() for _ in range(4)]
Ws = [(torch.randn((N, K), device="cuda", generator=g) * 0.02).bfloat16() for _ in range(16)]
C = torch.zeros((M, N // 2), dtype=torch.bfloat16, device="cuda")
def run(n, cycle):
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(n):
        A, W = (As[i % 4], Ws[i % 16]) if cycle else (As[0], Ws[0])
        L.check(lib.sr_gemm_bf16(A.data_ptr(), W.data_ptr(), M, N, K, epi, C.data_ptr(), None, L.stream_ptr()))
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / n
    return round(2.0 * M * N * K / ms / 1e9, 1)
run(5, False)
print("burst of 20, same operands      :", run(20, False), "TF")
print("4000 launches, same operands    :", run(4000, False), "TF")
print("4000 launches, cycling operands :", run(4000, True), "TF")
print("burst of 20 after that          :", run(20, False), "TF")
