R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_fixed; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/f -o f -- python3 $R/tools/micro/enc_fixed_batch.py > $O/f.log 2>&1
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
grep batch $O/f.log
