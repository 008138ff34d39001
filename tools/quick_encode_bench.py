"""Ad-hoc doc_encode timing with Lion-1B dims (development aid; run under rocprofv3 --stats for a kernel breakdown)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench as B
from scaling_retriever_amd.modeling.llm_encoder import LlamaBiDense, LlamaBiSparse
if __name__ == "__main__":
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    nb = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    sparse = len(sys.argv) > 3 and sys.argv[3] == "sparse"
    dev = torch.device("cuda", 0)
    cfg = dict(B.LION_1B)
    flop_tok = 1.946e9
    if os.environ.get("SR_MODEL") == "8b":   # Lion-*-8B dims (llama-3-8b): BASELINE.json configs[4]
        cfg.update(hidden_size=4096, intermediate_size=14336, num_hidden_layers=32, num_attention_heads=32,
                   num_key_value_heads=8, head_dim=128, tie_word_embeddings=False, rope_scaling=None)
        flop_tok = 13.96e9
    cls = LlamaBiSparse if sparse else LlamaBiDense
    model = cls.from_weights(cfg, B.random_weights(cfg, dev, 0), max_batch_tokens=65536, max_batch_seqs=2048).to(dev).eval()
    batches, lens = B.synth_batches(nb * batch, batch, 4.25, 0.35, 8, 192, cfg["vocab_size"], 3, dev)
    model.doc_encode(input_ids=batches[0][0], attention_mask=batches[0][1]); torch.cuda.synchronize()
    t = time.perf_counter()
    for i, m in batches: model.doc_encode(input_ids=i, attention_mask=m)
    torch.cuda.synchronize(); t = time.perf_counter() - t
    tok = int(lens.sum())
    print(json.dumps({"sparse": sparse, "batch": batch, "passages_per_s": round(len(lens) / t, 1), "tokens_per_s": round(tok / t),
                      "TF_body": round(tok * flop_tok / t / 1e12, 1), "model": os.environ.get("SR_MODEL", "1b")}))
