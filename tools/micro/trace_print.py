import csv, glob, sys
rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + "/*kernel_trace.csv")[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
for r in rows:
    n = r["Kernel_Name"]
    if "radix" in n or "scan" in n or "indptr" in n:
        print(n[:34], round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, 1))
