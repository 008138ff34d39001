cd $GRAFT_REPO_ROOT
for N in 16384 65536 262144 1048576; do for nq in 512 6980; do
  echo "== N $N nq $nq"
  SR_DEV_SWITCHES=1 SR_SPARSE_CERT=1 timeout -s KILL 300 python3 tools/quick_sparse_cert.py --exact 1 --check 16 --steps 3 --N $N --nq $nq --V 30000 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.read()); print('cert', r['certified']['ms_per_pass'], 'ms  exact', r['exact']['ms_per_pass'], 'ms  same:', r['same_bits_as_exact_kernels'], r['oracle_bit_exact'], 'redone', r['cert_after']['redone_exact'])"
done; done
