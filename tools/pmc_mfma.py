"""MFMA-pipe utilisation and effective clock per kernel from one rocprofv3 pass with
  --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE
usage: python tools/pmc_mfma.py <dir> [out.json]
mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs); clock = GRBM_GUI_ACTIVE / 8 / duration
(MI355X_MICROARCH.md: GRBM_GUI_ACTIVE is summed over the 8 XCDs; reads high on dispatches shorter than ~0.3 ms);
waves_parked = SQ_WAIT_ANY / SQ_WAVE_CYCLES (share of wave lifetime spent at s_waitcnt / barriers)."""
import collections, csv, glob, json, os, sys

d = sys.argv[1]
cnt = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        cnt[name][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[name].add(r["Dispatch_Id"])
dur = collections.defaultdict(float)
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"].split("(")[0].replace("void ", "")] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
out = {}
for k, c in cnt.items():
    if c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) <= 0 or dur[k] <= 0:
        continue
    cyc = c["GRBM_GUI_ACTIVE"] / 8.0
    out[k] = {"dispatches": len(disp[k]), "avg_us": round(dur[k] / len(disp[k]) * 1e6, 1),
              "mfma_busy": round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024.0), 4), "clock_GHz": round(cyc / dur[k] / 1e9, 3),
              "waves_parked": round(c.get("SQ_WAIT_ANY", 0) / max(1.0, c.get("SQ_WAVE_CYCLES", 1)), 3),
              "lds_bank_conflict_cycles_per_dispatch": round(c.get("SQ_LDS_BANK_CONFLICT", 0) / len(disp[k]), 1)}
out = dict(sorted(out.items(), key=lambda kv: -kv[1]["avg_us"] * kv[1]["dispatches"]))
if len(sys.argv) > 2:
    json.dump({"note": __doc__, "kernels": out}, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out, indent=1))
