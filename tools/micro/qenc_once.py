"""ONE fp32-regime encode of the 6 980 synthetic Dev queries (a single engine pass, as the headline step does), after a warm-up: for
rocprofv3 --kernel-trace --stats.  python3 tools/micro/qenc_once.py [reps]"""
import os, sys, time, torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tools"))
import bench
from scaling_retriever_amd.modeling.llm_encoder import LlamaBiDense
dev = torch.device("cuda", 0)
cfg = dict(bench.LION_1B)
model = LlamaBiDense.from_weights(cfg, bench.random_weights(cfg, dev, 0), max_batch_tokens=65536, max_batch_seqs=8192, fp32_planes=16).to(dev).eval()
batches, lens = bench.synth_batches(6980, 6980, 2.1, 0.35, 4, 64, cfg["vocab_size"], 2, dev)
model.base_model.precision = "fp32"
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    torch.cuda.synchronize(); t = time.perf_counter()
    for i, m in batches:
        model.query_encode(input_ids=i, attention_mask=m)
    torch.cuda.synchronize(); print("ms", round((time.perf_counter() - t) * 1e3, 1), "tokens", int(lens.sum()), flush=True)
