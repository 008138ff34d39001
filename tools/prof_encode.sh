#!/bin/bash
# Kernel statistics of the fp32-regime query encode and of the bf16 corpus encode:  bash tools/prof_encode.sh  (through gpurun)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_encode
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/q -o q -- python3 $R/tools/micro/qenc_once.py 4 > $O/q.log 2>&1
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/e -o e -- python3 $R/tools/quick_encode_budget.py 16384 > $O/e.log 2>&1
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
tail -3 $O/q.log $O/e.log
