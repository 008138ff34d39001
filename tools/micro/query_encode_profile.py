import os, sys, time, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import bench
from scaling_retriever_amd.modeling.llm_encoder import LlamaBiDense
dev = torch.device("cuda", 0)
cfg = dict(bench.LION_1B)
w = bench.random_weights(cfg, dev, 0)
model = LlamaBiDense.from_weights(cfg, dict(w), max_batch_tokens=65536, max_batch_seqs=8192, fp32_planes=16).to(dev).eval()
batches, lens = bench.synth_batches(6980, 2048, 2.1, 0.35, 4, 64, cfg["vocab_size"], 2, dev)
model.base_model.precision = "fp32"
for rep in range(3):
    torch.cuda.synchronize(); t = time.perf_counter()
    for i, m in batches:
        model.query_encode(input_ids=i, attention_mask=m)
    torch.cuda.synchronize(); print("ms", (time.perf_counter() - t) * 1e3, flush=True)
