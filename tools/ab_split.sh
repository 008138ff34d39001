#!/bin/bash
# A/B of the filter's upper-bound pass on a 2 M-doc slice: block test on / off, then the filter parity tests.
cd "$(dirname "$0")/.."
for v in 1 0; do
  echo "== SR_SPLIT_BLOCKTEST=$v"
  SR_DEV_SWITCHES=1 SR_SPLIT_BLOCKTEST=$v timeout 600 python tools/quick_split_bench.py 2000000 6980 2>&1 | tail -4
done
timeout 900 python -m pytest tests/test_dense_filtered_gpu.py tests/test_filter_corpora_gpu.py -x -q 2>&1 | tail -5
