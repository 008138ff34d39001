cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for d in 512 1024; do
rm -rf /tmp/st1
SR_HIP_LIB=$GRAFT_REPO_ROOT/build_var/libsr_cert_$d.so timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st1 -o t -- python3 tools/quick_sparse_cert.py --exact 0 --check 0 --steps 2 > /dev/null 2>&1
echo "== $d"; python3 - <<'PY'
import csv,glob
f=glob.glob('/tmp/st1/**/*kernel_stats.csv',recursive=True)[0]
for r in list(csv.DictReader(open(f))):
    if 'rescore' in r['Name']: print(r['Name'][:40], r['Calls'], "%.1f us avg"%(float(r['AverageNs'])/1e3))
PY
done
