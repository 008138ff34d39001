#!/bin/bash
# A/B runs of the certified sparse scorer's dev switches at the MSMARCO shape (one quick_sparse_cert.py run per variant, each under its own timeout):
#   gpurun -- bash tools/micro/cert_variants.sh "name ENV=VAL ..." "name2 ENV=VAL" ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out/cert_variants; mkdir -p $O
for spec in "$@"; do
  set -- $spec; name=$1; shift
  env "$@" SR_DEV_SWITCHES=1 timeout -s KILL 240 python3 tools/quick_sparse_cert.py --exact 0 --check 64 --steps 3 > $O/$name.json 2> $O/$name.err
  echo "== $name"; python3 -c "import json,sys; r=json.load(open('$O/$name.json')); print(r.get('certified'), r.get('oracle_bit_exact'))"; grep "cert stamps" $O/$name.err
done
