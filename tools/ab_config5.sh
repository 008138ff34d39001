#!/bin/bash
# 8B-dims encode (bench.py config5 leg) with the product library and with tools/ab_libs/libsr_hip_prev.so, same box
cd "$(dirname "$0")/.."
F="--n-docs 300000 --steps 1 --warmup 1 --no-encode --no-sparse --no-drop-in --no-robustness --no-shard-leg --no-fast-mode --no-cpu-baseline"
for lib in tools/ab_libs/libsr_hip_prev.so scaling_retriever_amd/libsr_hip.so; do
  python3 tools/ab_run.py $lib bench.py $F 2>&1 >/dev/null | grep -o '"config5_8b": {"workload[^}]*}[^}]*"passages_per_s_per_gpu": [0-9.]*' | grep -o '"value": [0-9.]*' | head -1 | sed "s|^|$lib |"
done
