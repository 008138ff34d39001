#!/bin/bash
# Kernel statistics of the certified dense search at the MSMARCO shape:  bash tools/prof_search.sh  (through gpurun)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_search
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/s -o s -- python3 $R/tools/split_ab.py $SPLIT_AB_ARGS > $O/s.log 2>&1
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
grep "^\[" $O/s.log
