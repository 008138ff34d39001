#!/bin/bash
# Diagnostic builds of the sparse scorer (never shipped): which side of sparse_score_kernel sets its rate?
#   SP_DIAG=1  posting loads only (score tile untouched)      SP_DIAG=2  LDS read-modify-writes only (no posting loads)
#   SP_RING=n  register sets of the group walk
# usage (on the GPU box, from the repo root): bash tools/micro/sparse_diag.sh
set -e
cd "$(dirname "$0")/../../scaling_retriever_amd/csrc"
OBJS=$(ls *.o | grep -v sparse_score.o)
for cfg in "0 3" "1 3" "2 3" "0 2"; do
  set -- $cfg
  hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DSP_DIAG=$1 -DSP_RING=$2 -c sparse_score.hip -o /tmp/sp_diag.o
  hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libsr_diag_$1_$2.so /tmp/sp_diag.o $OBJS
  echo "== SP_DIAG=$1 SP_RING=$2"
  (cd ../.. && SR_HIP_LIB=/tmp/libsr_diag_$1_$2.so python tools/bench_sparse.py --no-cpu --check 0 --steps 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], 'queries/s', d['roofline']['kernel_ms_per_pass'], 'ms kernel')")
done
