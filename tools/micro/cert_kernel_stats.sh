cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
rm -rf /tmp/st1
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st1 -o t -- python3 tools/quick_sparse_cert.py --exact 0 --check 0 --steps 4 > /dev/null 2>&1
python3 - <<'PY'
import csv,glob
f=glob.glob('/tmp/st1/**/*kernel_stats.csv',recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:22]:
    print(r['Name'][:60].ljust(60), r['Calls'].rjust(6), ("%.1f"%(float(r['TotalDurationNs'])/1e3/5)).rjust(10), "us/search", ("%.1f"%(float(r['AverageNs'])/1e3)).rjust(9))
PY
