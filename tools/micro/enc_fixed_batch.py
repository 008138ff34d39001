"""ONE token-budget batch (16 384 tokens budget) encoded 30 times, for rocprofv3 --kernel-trace --stats: every kernel call has the same shape, so
the per-kernel averages can be held against isolated GEMM timings (tools/quick_gemm_bench.py)."""
import os, sys, torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tools"))
os.environ.setdefault("TQDM_DISABLE", "1")
import bench
from scaling_retriever_amd.dataset.pipeline import TokenBudgetCollectionLoader
from scaling_retriever_amd.modeling.llm_encoder import LlamaBiDense
dev = torch.device("cuda", 0)
cfg = dict(bench.LION_1B)
model = LlamaBiDense.from_weights(cfg, bench.random_weights(cfg, dev, 0), max_batch_tokens=65536, max_batch_seqs=8192, fp32_planes=0).to(dev).eval()
chunks, lens = bench.synth_token_chunks(8192, 4.25, 0.35, 8, 192, cfg["vocab_size"], 5, (0, 8192))
loader = TokenBudgetCollectionLoader(tokenized=chunks, max_length=192, max_tokens=16384, max_seqs=1024, window=32768, pad_token_id=cfg["vocab_size"] - 1, padding_side="left")
b = next(iter(loader))
ids, mask = b["input_ids"].to(dev), b["attention_mask"].to(dev)
print("batch", tuple(ids.shape), "real tokens", int(mask.sum()), flush=True)
with torch.inference_mode(), torch.autocast("cuda", dtype=torch.bfloat16):
    for _ in range(30):
        model.doc_encode(input_ids=ids, attention_mask=mask)
torch.cuda.synchronize()
