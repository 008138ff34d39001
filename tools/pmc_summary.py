"""Summarise a rocprofv3 --pmc counter_collection.csv: per kernel name, mean of every counter per dispatch.
usage: python tools/pmc_summary.py <dir-or-csv> [name-substring]"""
import csv, glob, os, sys, collections
path = sys.argv[1]
files = [path] if path.endswith(".csv") else glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True)
sel = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
for f in files:
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0]
        if sel not in name: continue
        acc[name][r["Counter_Name"]] += float(r["Counter_Value"]); disp[name].add(r["Dispatch_Id"])
for name, c in acc.items():
    n = len(disp[name])
    print(name, "dispatches", n)
    for k, v in sorted(c.items()): print(f"   {k:32s} {v / n:18.1f}")
