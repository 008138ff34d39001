#!/bin/bash
# Round-4 profile collection on the GPU box (run through gpurun from the repo root): rocprofv3 kernel statistics and PMC passes,
# written under gpurun_out/r04_prof (the summaries that are judged are copied into profiles/ afterwards).
# PMC passes are --kernel-trace only (gpurun refuses --pmc together with the sys / hip / hsa trace domains).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_prof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $R/bench.py --no-cpu-baseline --no-fast-mode --no-sparse --no-encode --no-config5 --no-robustness --no-shard-leg --no-drop-in"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_stats -o bench -- $BENCH --steps 3 --warmup 1 > $O/bench_under_rocprof.json 2> $O/bench_stats.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o p -- $BENCH --steps 1 --warmup 0 > /dev/null 2> $O/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o p -- $BENCH --steps 1 --warmup 0 > /dev/null 2> $O/pmc_write.err
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mfma -o p -- $BENCH --steps 1 --warmup 0 > /dev/null 2> $O/pmc_mfma.err
SP="python3 $R/tools/bench_sparse.py --no-cpu --check 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/sparse_stats -o sp -- $SP --steps 2 > $O/sparse_under_rocprof.json 2> $O/sparse_stats.err
# HBM / fabric traffic of the sparse scorer: separate FETCH_SIZE / WRITE_SIZE passes
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_sp_fetch -o p -- $SP --steps 1 > /dev/null 2> $O/pmc_sp_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_sp_write -o p -- $SP --steps 1 > /dev/null 2> $O/pmc_sp_write.err
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CU_CYCLES --output-format csv -d $O/pmc_sp1 -o p -- $SP --steps 1 > /dev/null 2> $O/pmc_sp1.err
rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sp3 -o p -- $SP --steps 1 > /dev/null 2> $O/pmc_sp3.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/qenc_stats -o q -- python3 $R/tools/quick_query_encode.py 16 > $O/qenc.log 2> $O/qenc.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/enc_stats -o e -- python3 $R/tools/quick_encode_budget.py 16384 > $O/enc.log 2> $O/enc.err
# the GEMM loop's MFMA-pipe counters on the encoder shapes
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_gemm -o p -- python3 $R/tools/quick_gemm_bench.py 16384 > /dev/null 2> $O/pmc_gemm.err
cd $R
python3 tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write $O/pmc_traffic.json > /dev/null 2> $O/pmc_traffic.err
python3 tools/pmc_traffic.py $O/pmc_sp_fetch $O/pmc_sp_write $O/pmc_sparse_traffic.json > /dev/null 2>> $O/pmc_traffic.err
python3 tools/pmc_counters.py $O/pmc_mfma --out $O/pmc_mfma.json --match dense_split > /dev/null 2>> $O/pmc_traffic.err
python3 tools/pmc_counters.py $O/pmc_gemm --out $O/pmc_gemm.json --match gemm_bf16 > /dev/null 2>> $O/pmc_traffic.err
python3 tools/pmc_counters.py $O/pmc_sp1 $O/pmc_sp3 --out $O/pmc_sparse_block.json --match sparse_block > /dev/null 2>> $O/pmc_traffic.err
python3 tools/pmc_finish_r04.py $O > /dev/null 2>> $O/pmc_traffic.err
# keep the summaries, drop the bulky per-dispatch traces
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*.db" -delete
ls -la $O $O/*/ 2>/dev/null | head -70
