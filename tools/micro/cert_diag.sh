#!/bin/bash
# Timing-only variants of cert_score_kernel (never shipped; wrong results): which role sets the tile time?
#   SC_DIAG bit 1: scatter waves add no postings   bit 2: matrix waves load / multiply nothing   bit 4: no table lookups
#           bit 8: no LDS adds   16: plain read-modify-write instead of the LDS atomic   32: loads but no MFMA   64: MFMA but no loads
#           bit 128: no candidates (queries then go to the exact kernels: read the cert kernel's own time from the stamps or a trace)
#           bit 512: re-score loads its rows but intersects nothing   1024: re-score reads every row from the table's start (cache hits)
#           bit 2048: re-score adds with its scalar loop for every query
#   third argument: extra compiler flags, e.g. "-DSC_STAMPS=1 -Wno-inline-asm" for the per-role cycle stamps (SR_CERT_STAMPS=<first tile>)
# Build HERE (hipcc cross-compiles): bash tools/micro/cert_diag.sh build "0 1 2 3 5 7" ; run on the GPU box: bash tools/micro/cert_diag.sh run "0 1 2 3 5 7"
set -e
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
mkdir -p "$ROOT/build_var"
cd "$ROOT/scaling_retriever_amd/csrc"
if [ "$1" = build ]; then
  OBJS=$(ls *.o | grep -v sparse_cert.o)
  for d in $2; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DSC_DIAG=$d $3 -c sparse_cert.hip -o "$ROOT/build_var/cert_$d.o"
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/build_var/libsr_cert_$d.so" "$ROOT/build_var/cert_$d.o" $OBJS
  done
else
  cd "$ROOT"
  for d in $2; do
    echo "== SC_DIAG=$d"
    SR_CERT_STAMPS=${STAMPS_FROM:-1} SR_HIP_LIB="$ROOT/build_var/libsr_cert_$d.so" python tools/quick_sparse_cert.py --exact 0 --check 0 --steps 2 2>&1 | grep -v amdgpu.ids | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'): print(json.loads(l)['certified'])
    elif 'stamps' in l: print(l.strip())"
  done
fi
