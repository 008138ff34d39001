#!/bin/bash
# MFMA-pipe / clock / LDS counters of the vendor GEMM (hipBLASLt through torch.matmul, tools/micro/blas_ref.py) next to
# gemm_bf16_kernel (tools/quick_gemm_bench.py) on the encoder's layer shapes at 16 384 token rows: where a difference in
# TFLOP/s comes from (clock held, MFMA-busy share, LDS conflicts).  Run through gpurun from the repo root.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/gemm_pmc
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for who in vendor ours; do
  if [ $who = vendor ]; then CMD="python3 $R/tools/micro/blas_ref.py 16384"; else CMD="python3 $R/tools/quick_gemm_bench.py 16384"; fi
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/${who}_a -o p -- $CMD > $O/${who}_a.log 2> $O/${who}_a.err
  rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY --output-format csv -d $O/${who}_b -o p -- $CMD > $O/${who}_b.log 2> $O/${who}_b.err
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM --output-format csv -d $O/${who}_c -o p -- $CMD > $O/${who}_c.log 2> $O/${who}_c.err
done
cd $R
python3 tools/pmc_counters.py $O/vendor_a $O/vendor_b $O/vendor_c --out $O/vendor.json --match Cijk > /dev/null
python3 tools/pmc_counters.py $O/ours_a $O/ours_b $O/ours_c --out $O/ours.json --match gemm_bf16 > /dev/null
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*.db" -delete
python3 - <<PY
import json
for who in ("vendor", "ours"):
    d = json.load(open("$O/%s.json" % who))
    for k, v in d["kernels"].items():
        c = v["counters"]
        print(who, k[:90], "disp", v["dispatches"], "ms", round(v["total_ms"], 2), {x: v[x] for x in v if x not in ("counters", "dispatches", "total_ms")},
              {x: "%.3g" % c[x] for x in c if x.startswith("SQ_INSTS")})
PY
