#!/bin/bash
# PMC passes over cert_score_kernel at the full MSMARCO shape: where the vector-memory path, the LDS and the issue slots stand.
# usage (GPU box, repo root): bash tools/micro/cert_pmc.sh [tag]
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-cert}
O=$R/gpurun_out/pmc_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
SP="python3 $R/tools/quick_sparse_cert.py --exact 0 --check 0 --steps 1"
i=0
for set in ${CERT_PMC_SETS:+"$CERT_PMC_SETS"} ; do :; done
IFS='|' read -ra SETS <<< "${CERT_PMC_SETS:-SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS|TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum|SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU|GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY|TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum|TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum|FETCH_SIZE|SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS_ATOMIC SQ_BUSY_CYCLES}"
for set in "${SETS[@]}"; do
  i=$((i+1))
  timeout -s KILL 150 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/p$i -o p -- $SP > /dev/null 2> $O/p$i.err || echo "pass $i failed: $set"
done
cd $R
python3 tools/pmc_counters.py $O/p* --out $O/pmc.json --match cert_score_kernel > /dev/null
find $O -name "*.csv" -delete; find $O -name "*.db" -delete
python3 -c "
import json
d=json.load(open('$O/pmc.json'))['kernels']
for k,v in d.items():
    print(k, v['dispatches'], v['total_ms'], {a:b for a,b in v.items() if a not in ('counters','dispatches','total_ms')})
    for c,x in v['counters'].items(): print('   ', c, '%.4g'%x)
"
