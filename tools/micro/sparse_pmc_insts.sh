#!/bin/bash
# Instruction-issue counters of sparse_block_kernel at the full MSMARCO shape (what the waves spend their non-parked cycles on).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/sparse_insts
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
SP="python3 $R/tools/bench_sparse.py --no-cpu --check 0 --steps 1"
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" "SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" "SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_FLAT"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/p$i -o p -- $SP > /dev/null 2> $O/p$i.err || echo "pass $i failed: $set"
done
cd $R
python3 tools/pmc_counters.py $O/p1 $O/p2 $O/p3 $O/p4 $O/p5 --out $O/insts.json --match sparse_block_kernel > /dev/null
find $O -name "*.csv" -delete; find $O -name "*.db" -delete
python3 -c "
import json
d=json.load(open('$O/insts.json'))['kernels']
for k,v in d.items():
    print(k, v['dispatches'], v['total_ms'])
    for c,x in v['counters'].items(): print('   ', c, '%.4g'%x)
"
