#!/bin/bash
# Build k-step schedule variants of the four-wave GEMM loop (plain-store epilogue only: seconds per variant) into build_var/.
# usage (from the repo root; hipcc cross-compiles without a GPU):  bash tools/micro/gemm_variants.sh "name RD1 B1 DMA B2 RD2 [extra -D flags]" ...
# then on the GPU box:  SR_HIP_LIB=build_var/libsr_<name>.so python3 tools/gemm_vs_hipblaslt.py out.json
set -e
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
cd "$ROOT/scaling_retriever_amd/csrc"
OBJS=$(ls *.o | grep -v gemm_bf16.o)
mkdir -p "$ROOT/build_var"
for cfg in "$@"; do
  set -- $cfg
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DSR_GEMM_VARIANT_BUILD -DKL_RD1_EVERY=$2 -DKL_B1_AT=$3 -DKL_DMA_EVERY=$4 -DKL_B2_AT=$5 -DKL_RD2_EVERY=$6 $7 $8 \
      -c gemm_bf16.hip -o "$ROOT/build_var/gemm_$1.o" && \
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/build_var/libsr_$1.so" "$ROOT/build_var/gemm_$1.o" $OBJS && rm "$ROOT/build_var/gemm_$1.o" && echo "built $1" ) &
done
wait
