#!/bin/bash
# Build variants of the query-block kernel (ring depth / postings per lane) and time them at full MSMARCO shape.
# usage (on the GPU box, from the repo root): bash tools/micro/sparse_blocks_variants.sh "6 4 1 8" ...   (SPB_RING SPB_U SPB_SUBS SPB_LRING SP_SUB SPB_WAVES_PER_SIMD)
set -e
cd "$(dirname "$0")/../../scaling_retriever_amd/csrc"
OBJS=$(ls *.o | grep -v sparse_score.o)
for cfg in "$@"; do
  set -- $cfg
  hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DSPB_RING=$1 -DSPB_U=$2 -DSPB_SUBS=$3 -DSPB_LRING=$4 -DSP_SUB=$5 -DSPB_WAVES_PER_SIMD=$6 -DSPB_DRING=$7 -DSPB_Q=$8 -c sparse_score.hip -o /tmp/sp_var.o
  hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libsr_var_$1_$2_$3_$4_$5_$6_$7_$8.so /tmp/sp_var.o $OBJS
  for diag in 0 14 12; do
    echo "== SPB_RING=$1 SPB_U=$2 SPB_SUBS=$3 SPB_LRING=$4 SP_SUB=$5 WAVES=$6 DRING=$7 Q=$8 SR_SPARSE_DIAG=$diag"
    (cd ../.. && SR_DEV_SWITCHES=1 SR_SPARSE_DIAG=$diag SR_HIP_LIB=/tmp/libsr_var_$1_$2_$3_$4_$5_$6_$7_$8.so timeout 150 python tools/bench_sparse.py --no-cpu --check 0 --steps 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], 'queries/s', d['ms_per_pass'], 'ms/pass')")
  done
done
