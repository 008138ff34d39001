"""Per-kernel PMC summary from one or more rocprofv3 passes (each `--kernel-trace --pmc <counters>` run in its own
directory - gpurun refuses --pmc together with the sys/hip/hsa trace domains).
usage: python tools/pmc_counters.py <dir> [<dir> ...] [--out file.json] [--match substr]
Per kernel (summed over its dispatches): every counter, the kernel time, effective clock = GRBM_GUI_ACTIVE / 8 / time
(MI355X_MICROARCH.md: summed over the 8 XCDs), and the ratios that read directly:
  waves_parked     SQ_WAIT_ANY / SQ_WAVE_CYCLES           share of wave lifetime at s_waitcnt / barriers
  issue_stalled    SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES
  lds_issue_stall  SQ_WAIT_INST_LDS / SQ_WAVE_CYCLES
  lds_busy         SQ_LDS_IDX_ACTIVE / (cycles x 256 CUs)    LDS-array cycles per CU cycle
  lds_conflict     SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
  mfma_busy        SQ_VALU_MFMA_BUSY_CYCLES / (cycles x 1024 SIMDs)"""
import collections
import csv
import glob
import json
import os
import sys

args = [a for a in sys.argv[1:]]
out_path = match = None
if "--out" in args:
    i = args.index("--out"); out_path = args[i + 1]; del args[i:i + 2]
if "--match" in args:
    i = args.index("--match"); match = args[i + 1]; del args[i:i + 2]
cnt = collections.defaultdict(lambda: collections.defaultdict(float))
dur = collections.defaultdict(float)
ndisp = collections.defaultdict(set)


def short(name):
    return name.split("(")[0].replace("void ", "")


for d in args:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            cnt[short(r["Kernel_Name"])][r["Counter_Name"]] += float(r["Counter_Value"])
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        seen = collections.defaultdict(float)
        n = collections.defaultdict(int)
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            seen[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
            n[k] += 1
        for k in seen:       # several passes: keep the first pass's time and dispatch count
            if k not in dur:
                dur[k], ndisp[k] = seen[k], n[k]
out = {}
for k, c in cnt.items():
    if match and match not in k:
        continue
    t = dur.get(k, 0.0)
    row = {"dispatches": ndisp.get(k, 0), "total_ms": round(t * 1e3, 3), "counters": {a: b for a, b in sorted(c.items())}}
    cyc = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    wc = c.get("SQ_WAVE_CYCLES", 0.0)
    if cyc and t:
        row["clock_GHz"] = round(cyc / t / 1e9, 3)
    for name, num, den in (("waves_parked", "SQ_WAIT_ANY", wc), ("issue_stalled", "SQ_WAIT_INST_ANY", wc),
                           ("lds_issue_stall", "SQ_WAIT_INST_LDS", wc), ("lds_busy", "SQ_LDS_IDX_ACTIVE", cyc * 256.0),
                           ("lds_conflict", "SQ_LDS_BANK_CONFLICT", c.get("SQ_LDS_IDX_ACTIVE", 0.0)),
                           ("mfma_busy", "SQ_VALU_MFMA_BUSY_CYCLES", cyc * 1024.0)):
        if num in c and den:
            row[name] = round(c[num] / den, 4)
    out[k] = row
out = dict(sorted(out.items(), key=lambda kv: -kv[1]["total_ms"]))
if out_path:
    json.dump({"note": __doc__, "kernels": out}, open(out_path, "w"), indent=1)
print(json.dumps(out, indent=1))
