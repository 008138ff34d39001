"""Where a 256x256 GEMM tile's time goes: s_memrealtime stamps (100 MHz) from inside gemm_bf16_kernel (diagnostic path)."""
import os, sys
import json
os.environ["SR_DEV_SWITCHES"] = "1"   # the library reads its development switches only with this set
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["SR_GEMM_STAMPS"] = "1"; os.environ["SR_GEMM_TILE"] = "256"
import numpy as np, torch
from scaling_retriever_amd import _lib as L
lib = L.load()
M = N = 8192
for K, epi in ((2048, 0), (2048, 2), (8192, 0)):
    g = torch.Generator(device="cuda").manual_seed(0)
    A = torch.randn((M, K), device="cuda", generator=g).bfloat16(); W = (torch.randn((N, K), device="cuda", generator=g) * 0.02).bfloat16()
    C = torch.zeros((M, N), dtype=torch.float32, device="cuda")
    st = torch.zeros((256, 64), dtype=torch.int64, device="cuda")
    for _ in range(30):
        L.check(lib.sr_gemm_bf16(A.data_ptr(), W.data_ptr(), M, N, K, epi, C.data_ptr(), st.data_ptr(), L.stream_ptr()))
    torch.cuda.synchronize()
    raw = st.cpu().numpy()
    ticks = (raw[:, 63] - raw[:, 62]).astype(np.float64)
    rt = (raw[:, 6] - raw[:, 5]).astype(np.float64)            # tile 1: realtime at k-loop start / end
    clock_ghz = float(np.median(ticks / rt) * 0.1)
    s = raw.reshape(256, 16, 4)[:, :4, :].astype(np.float64) * 0.01   # us
    t0 = s[:, 0, 0].min()
    prolog = s[:, :, 1] - s[:, :, 0]; loop = s[:, :, 2] - s[:, :, 1]; epil = s[:, :, 3] - s[:, :, 2]
    print(json.dumps({"K": K, "epi": epi, "start_skew_us": round(float(s[:, 0, 0].max() - t0), 2),
                      "prologue_us_by_tile": [round(float(x), 2) for x in prolog.mean(0)],
                      "kloop_us_by_tile": [round(float(x), 2) for x in loop.mean(0)],
                      "epilogue_us_by_tile": [round(float(x), 2) for x in epil.mean(0)],
                      "total_us": round(float(s[:, 3, 3].max() - t0), 2), "in_kernel_clock_GHz": round(clock_ghz, 3),
                      "mfma_cycles_ideal_per_kstep": 2048, "cycles_per_kstep": round(float(loop.mean(0)[1]) / (K / 64) * clock_ghz * 1e3, 0)}), flush=True)
