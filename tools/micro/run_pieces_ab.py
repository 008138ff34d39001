"""eval_dense's retrieval task incl. run.json at the full shape for several piece plans of write_run (RUN_PIECES / RUN_PIECE_SHARES):
the GPU encodes and searches piece c + 1 while a worker thread writes piece c.  python3 tools/micro/run_pieces_ab.py"""
import argparse, os, sys, tempfile, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tools"))
os.environ.setdefault("TQDM_DISABLE", "1")
import numpy as np, torch
import bench, eval_dense
from scaling_retriever_amd.modeling.llm_encoder import LlamaBiDense
from scaling_retriever_amd.indexer import DenseFlatIndexer
from scaling_retriever_amd.scoring import DenseIndexHIP
dev = torch.device("cuda", 0)
cfg = dict(bench.LION_1B)
H, N, nq = 2048, 8_841_823, 6980
model = LlamaBiDense.from_weights(cfg, bench.random_weights(cfg, dev, 0), max_batch_tokens=65536, max_batch_seqs=8192).to(dev).eval()
g = torch.Generator(device=dev).manual_seed(1)
D = torch.empty((N, H), dtype=torch.float32, device=dev)
for r0 in range(0, N, 1 << 20):
    D[r0:r0 + (1 << 20)].normal_(0.0, 0.5 / H ** 0.5, generator=g)
index = DenseIndexHIP(H, device=dev)
index.set_precision("fp32_filtered")
index.add_device_rows(D)
fi = DenseFlatIndexer()
fi.hidden_dim, fi.index = H, index
fi._update_id_mapping(np.arange(N).astype("U8").tolist())
fi.id_table(), fi.run_table()
qb, lens = bench.synth_batches(nq, 128, 2.1, 0.35, 4, 64, cfg["vocab_size"], 2, dev)
batches = bench._loader_batches(qb, [str(1_000_000 + 7 * i) for i in range(nq)], 128)
ret = eval_dense.LocalFaissDenseRetriever(model, device=dev, index=fi)
import scaling_retriever_amd.utils.run_file as RF
_orig = RF.write_run_json
_log = []
def _timed(*a, **k):
    t0 = time.perf_counter(); r = _orig(*a, **k); _log.append((round(t0 * 1e3 % 100000, 1), round((time.perf_counter() - t0) * 1e3, 1), len(a[1])))
    return r
RF.write_run_json = _timed
_os = eval_dense.LocalFaissDenseRetriever.write_run
with tempfile.TemporaryDirectory() as tmp:
    for plan in ((4, (9, 8, 7, 4)), (3, (10, 8, 3)), (3, (9, 7, 4)), (2, (5, 1)), (5, (9, 8, 7, 5, 3))):
        ret.RUN_PIECES, ret.RUN_PIECE_SHARES = plan
        ts = []
        for rep in range(3):
            torch.cuda.synchronize(); t = time.perf_counter()
            ret.write_run(batches, 1000, os.path.join(tmp, "run.json"))
            ts.append(time.perf_counter() - t)
        print(plan, "ms", [round(x * 1e3, 1) for x in ts], ret.last_run_timeline, flush=True)
        print("   writes (start ms mod 1e5, duration ms, queries) of the last call:", _log[-plan[0]:], "call ended at", round(time.perf_counter() * 1e3 % 100000, 1), flush=True)
