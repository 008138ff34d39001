import time, os, numpy as np, tempfile, sys
sys.path.insert(0,'/root/repo')
from scaling_retriever_amd.utils.run_file import write_run_json, id_table
nq, k, N = 2560, 1000, 8841823
rng = np.random.default_rng(0)
scores = np.sort(rng.random((nq, k), dtype=np.float32) * 100)[:, ::-1].copy()
pos = rng.integers(0, N, size=(nq, k), dtype=np.int64)
docs = np.arange(N).astype("U8")
t0=time.perf_counter(); table = id_table(docs.tolist()); print("table", time.perf_counter()-t0)
qids = [str(1000000 + 7*i) for i in range(nq)]
d = tempfile.mkdtemp(); p = os.path.join(d, "run.json")
for rep in range(3):
    t0 = time.perf_counter(); sz = write_run_json(p, qids, scores, pos, table); t1 = time.perf_counter()
    print("one call %.1f ms, %d bytes" % ((t1 - t0) * 1e3, sz))
for rep in range(2):
    ts = []
    for c, part in enumerate((1, 2, 3)):
        t0 = time.perf_counter(); sz = write_run_json(p, qids[c*800:(c+1)*800+ (160 if c==2 else 0)], scores[c*800:(c+1)*800+(160 if c==2 else 0)], pos[c*800:(c+1)*800+(160 if c==2 else 0)], table, part=part); ts.append((time.perf_counter() - t0) * 1e3)
    print("pieces", ["%.1f" % x for x in ts], sz)
