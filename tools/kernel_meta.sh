#!/bin/bash
# Register / scratch metadata of every kernel in a built object:  tools/kernel_meta.sh scaling_retriever_amd/csrc/dense_split.o
set -e
L=/opt/rocm/lib/llvm/bin
T=$(mktemp -d)
$L/llvm-objcopy --dump-section=.hip_fatbin=$T/fat "$1" $T/unused.o
$L/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$T/fat --output=$T/dev.co --unbundle
$L/llvm-readelf --notes $T/dev.co | grep -E "^\s+\.name:|\.vgpr_count|\.agpr_count|\.sgpr_count|private_segment_fixed|vgpr_spill|group_segment_fixed" | \
  awk '/\.name:/{if(l)print l; l=$2; next}{l=l" "$1$2}END{print l}' | sed 's/\.private_segment_fixed_size:/scratch=/;s/\.vgpr_spill_count:/vspill=/;s/\.vgpr_count:/vgpr=/;s/\.agpr_count:/agpr=/;s/\.sgpr_count:/sgpr=/;s/\.group_segment_fixed_size:/lds=/'
[ -n "$KEEP" ] && cp $T/dev.co "$KEEP"
rm -rf $T
