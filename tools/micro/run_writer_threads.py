import time, os, numpy as np, tempfile, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from scaling_retriever_amd.utils.run_file import write_run_json, id_table
nq, k, N = 2304, 1000, 8841823
rng = np.random.default_rng(0)
scores = np.sort(rng.random((nq, k), dtype=np.float32) * 100)[:, ::-1].copy()
pos = rng.integers(0, N, size=(nq, k), dtype=np.int64)
table = id_table(np.arange(N).astype("U8").tolist())
qids = [str(1000000 + 7*i) for i in range(nq)]
d = tempfile.mkdtemp(); p = os.path.join(d, "run.json")
write_run_json(p, qids, scores, pos, table)
for nt in (8, 16, 24, 32, 48, 64):
    ts = []
    for rep in range(3):
        t0 = time.perf_counter(); sz = write_run_json(p, qids, scores, pos, table, n_threads=nt); ts.append((time.perf_counter() - t0) * 1e3)
    print("threads", nt, ["%.1f" % x for x in ts], sz, flush=True)
