#!/bin/bash
# A second build of the library beside the product one, for same-box A/B runs (tools/split_ab.py --lib):
#   tools/build_variant.sh NAME [EXTRA flags]     ->  tools/ab_libs/libsr_hip_NAME.so  (built from the working tree in a scratch copy)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
T=$(mktemp -d)
mkdir -p $T/scaling_retriever_amd $T/include tools/ab_libs
cp -r scaling_retriever_amd/csrc $T/scaling_retriever_amd/
cp include/*.h $T/include/
rm -f $T/scaling_retriever_amd/csrc/*.o
make -C $T/scaling_retriever_amd/csrc -j8 EXTRA="$*" ../libsr_hip.so >/dev/null
cp $T/scaling_retriever_amd/libsr_hip.so tools/ab_libs/libsr_hip_$name.so
rm -rf $T
echo tools/ab_libs/libsr_hip_$name.so
