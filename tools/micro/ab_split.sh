#!/bin/bash
# A/B of dense_split.hip build variants on one box.  usage: bash tools/micro/ab_split.sh <n_docs> <n_queries> "<-D flags A>" "<-D flags B>"
set -e
cd "$(dirname "$0")/../../scaling_retriever_amd/csrc"
OBJS=$(ls *.o | grep -v dense_split.o)
i=0
for flags in "$3" "$4"; do
  hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 $flags -c dense_split.hip -o /tmp/ab_split_$i.o
  hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libsr_ab_$i.so /tmp/ab_split_$i.o $OBJS
  i=$((i+1))
done
cd ../..
for rep in 1 2; do
  echo "== A: $3"; SR_HIP_LIB=/tmp/libsr_ab_0.so python tools/quick_split_bench.py $1 $2 2>/dev/null | tail -2
  echo "== B: $4"; SR_HIP_LIB=/tmp/libsr_ab_1.so python tools/quick_split_bench.py $1 $2 2>/dev/null | tail -2
done
