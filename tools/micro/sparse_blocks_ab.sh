#!/bin/bash
# A/B of the query-block inverted-index kernel at full MSMARCO shape (dev switches; diag runs give wrong results, timing only)
export SR_DEV_SWITCHES=1
run() { echo "== $*"; env "$@" timeout 150 python tools/bench_sparse.py --no-cpu --steps 2 --check 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], 'q/s', d['ms_per_pass'], 'ms/pass')"; }
for cfg in "$@"; do run $cfg; done
