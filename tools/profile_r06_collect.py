"""Copies the judged summaries of gpurun_out/r06_prof (tools/profile_r06.sh) into profiles/ and ties the sparse traffic figure to the
kernel source it was measured on (bench.py reads it back only for the same source and shape)."""
import glob
import hashlib
import json
import os
import shutil
import subprocess
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O = os.path.join(R, "gpurun_out", "r06_prof")
P = os.path.join(R, "profiles")


def copy(src, dst):
    f = glob.glob(os.path.join(O, src))
    if f:
        shutil.copy(f[0], os.path.join(P, dst))
        print("copied", dst)
    else:
        print("missing", src, file=sys.stderr)


copy("bench_line.json", "r06_bench_line.json")
copy("bench_detail.json", "r06_bench_detail.json")
copy("bench_under_rocprof.json", "r06_bench_under_rocprof.json")
copy("bench_stats/**/bench_kernel_stats.csv", "r06_bench_kernel_stats.csv") if glob.glob(os.path.join(O, "bench_stats/**/bench_kernel_stats.csv"), recursive=True) else None
for src, dst in (("bench_stats", "r06_bench_kernel_stats.csv"), ("sparse_stats", "r06_sparse_kernel_stats.csv"), ("qenc_stats", "r06_query_encode_kernel_stats.csv"),
                 ("enc_stats", "r06_encode_kernel_stats.csv"), ("encfix_stats", "r06_encode_fixed_batch_kernel_stats.csv")):
    f = glob.glob(os.path.join(O, src, "**", "*kernel_stats.csv"), recursive=True)
    if f:
        shutil.copy(f[0], os.path.join(P, dst))
        print("copied", dst)
for src, dst in (("pmc_traffic.json", "r06_pmc_traffic.json"), ("pmc_mfma.json", "r06_pmc_mfma.json"), ("pmc_sparse.json", "r06_pmc_sparse.json"),
                 ("gpu_suite.txt", "r06_gpu_suite.txt")):
    copy(src, dst)
# dense traffic: what bench.py needs beside the per-kernel figures to know that the file matches the kernel and the shape it runs
try:
    pj = os.path.join(P, "r06_pmc_traffic.json")
    d = json.load(open(pj))
    srcf = "scaling_retriever_amd/csrc/dense_split.hip"
    d["note"] = ("rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over the reduced bench command of tools/profile_r06.sh, "
                 "folded by tools/pmc_traffic.py; KB per dispatch averaged over the kernel's dispatches. traffic_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 "
                 "(MI355X_MICROARCH.md: gfx950 FETCH_SIZE reports half the bytes of a wide coalesced read and counts hits in the 256 MB last-level cache (MALL)).")
    d["commit"] = subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=R, capture_output=True, text=True).stdout.strip() or "(snapshot without .git)"
    launches = 138                                  # 65 536 docs per launch (round 6: twice round 5's), short doubling launches first
    docs = 8841823 / launches
    d["dense_split_launch"] = {"n_docs": 8841823, "nq": 6980, "dim": 2048, "launches_per_search": launches,
                               "algorithmic_bytes": int(docs * (2048 * 2 + 8) + 6980 * (2048 * 2 + 16)),
                               "algorithmic_note": "one fp16 plane of the launch's docs + their (x, y) + one fp16 plane of the queries, each once"}
    d["kernel_source"] = {"file": srcf, "sha256": hashlib.sha256(open(os.path.join(R, srcf), "rb").read()).hexdigest()}
    json.dump(d, open(pj, "w"), indent=1)
    print("annotated r06_pmc_traffic.json")
except Exception as e:
    print("dense traffic:", e, file=sys.stderr)
# sparse traffic: per pass of 6 980 queries, tied to the kernel source
try:
    t = json.load(open(os.path.join(O, "pmc_sparse_traffic.json")))["kernels"]
    k = [v for name, v in t.items() if name.startswith("cert_score_kernel")][0]
    src = "scaling_retriever_amd/csrc/sparse_cert.hip"
    doc = {"what": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over tools/quick_sparse_cert.py --exact 0 --check 0 --steps 1 "
                   "(two searches of 6 980 queries), cert_score_kernel only",
           "how": "traffic = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 summed over the kernel's dispatches of one search (MI355X_MICROARCH.md: FETCH_SIZE "
                  "reports half the bytes of wide coalesced reads on gfx950 and counts hits in the 256 MB last-level cache (MALL); most of this kernel's reads are 16-byte loads)",
           "shape": {"V": 128256, "N": 8841823, "L0_d": 128, "L0_q": 32, "nq": 6980},
           "passes_profiled": 2, "dispatches": k["dispatches"], "traffic_bytes_per_pass": int(k["traffic_bytes"] * k["dispatches"] / 2),
           "fetch_kb_per_dispatch": k["FETCH_SIZE_KB"], "write_kb_per_dispatch": k["WRITE_SIZE_KB"],
           "kernel_source": {"file": src, "sha256": hashlib.sha256(open(os.path.join(R, src), "rb").read()).hexdigest()},
           "commit": subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=R, capture_output=True, text=True).stdout.strip()}
    json.dump(doc, open(os.path.join(P, "r06_pmc_sparse_traffic.json"), "w"), indent=1)
    print("wrote r06_pmc_sparse_traffic.json", doc["traffic_bytes_per_pass"])
except Exception as e:
    print("sparse traffic:", e, file=sys.stderr)
