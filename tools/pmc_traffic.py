"""Fold two rocprofv3 PMC passes (--pmc FETCH_SIZE and --pmc WRITE_SIZE, collected separately) into per-kernel HBM/fabric
traffic per dispatch, as profiles/r01_pmc_summary.json holds it.
usage: python tools/pmc_traffic.py <fetch-dir> <write-dir> [out.json]
traffic_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (MI355X_MICROARCH.md, HBM: on gfx950 FETCH_SIZE reports half the bytes
of a wide coalesced read and counts Infinity-Cache hits; both counters are in KB)."""
import collections, csv, glob, json, os, sys


def per_kernel(path, counter):
    tot, disp = collections.defaultdict(float), collections.defaultdict(set)
    for f in glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")
            tot[name] += float(r["Counter_Value"])
            disp[name].add(r["Dispatch_Id"])
    return {k: (tot[k] / len(disp[k]), len(disp[k])) for k in tot}


fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
out = {}
for k in sorted(fetch, key=lambda k: -fetch[k][0] * fetch[k][1]):
    if k.startswith("at::") or "rocclr" in k:
        continue
    w = write.get(k, (0.0, 0))[0]
    out[k] = {"dispatches": fetch[k][1], "FETCH_SIZE_KB": round(fetch[k][0], 1), "WRITE_SIZE_KB": round(w, 1),
              "traffic_bytes": int((2 * fetch[k][0] + w) * 1024)}
if len(sys.argv) > 3:
    doc = json.load(open(sys.argv[3])) if os.path.exists(sys.argv[3]) else {}
    doc["kernels"] = out
    json.dump(doc, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))
