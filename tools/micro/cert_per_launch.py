#!/usr/bin/env python3
"""Per-launch durations of cert_score_kernel from a rocprofv3 --kernel-trace CSV (the geometric launches 1, 1, 2, 4 ... 512 tiles, then 512 each):
  rocprofv3 --kernel-trace --output-format csv -d DIR -o t -- python3 tools/quick_sparse_cert.py --exact 0 --check 0 --steps 1
  python3 tools/micro/cert_per_launch.py DIR"""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if r["Kernel_Name"].startswith("void cert_score_kernel") or r["Kernel_Name"].startswith("cert_score_kernel")]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 24
passes = len(d) // n
last = d[(passes - 1) * n:passes * n]
print("launches per pass", n, "passes", passes, "last pass total ms %.2f" % (sum(last) / 1e3))
print("us per launch:", " ".join("%.0f" % x for x in last))
rows2 = [r for r in csv.DictReader(open(f)) if "topk_compact_kernel" in r["Kernel_Name"]]
rows2.sort(key=lambda r: int(r["Start_Timestamp"]))
d2 = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows2]
per = len(d2) // passes if passes else 0
print("compactions per pass", per, "us each:", " ".join("%.0f" % x for x in d2[(passes - 1) * per:passes * per]))
