cd $GRAFT_REPO_ROOT
for cfg in "--k 10 --nq 1000" "--k 100 --nq 1000" "--k 2000 --nq 700" "--k 3072 --nq 500" "--k 1000 --nq 3000 --alpha 0.8 --cap-div 3" "--k 1000 --nq 2000 --L0-d 64 --L0-q 8" "--k 500 --nq 2000 --L0-d 200 --L0-q 48 --alpha 1.2"; do
  echo "== $cfg"
  timeout -s KILL 300 python3 tools/quick_sparse_cert.py --exact 1 --check 32 --steps 1 $cfg 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.read()); print(r['certified']['qps'], r['exact']['qps'], 'same as exact:', r['same_bits_as_exact_kernels'], 'oracle:', r['oracle_bit_exact'], 'redone', r['cert_after']['redone_exact'], 'of', r['cert_after']['queries'])"
done
