#!/bin/bash
# Per-launch kernel times of cert_score_kernel for the product library ("new") and diagnostic builds in build_var/ (tools/micro/cert_diag.sh build):
#   gpurun -- bash tools/micro/cert_per_launch.sh new 128 256 old
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for v in "$@"; do
  unset SR_HIP_LIB
  [ $v != new ] && export SR_HIP_LIB=$GRAFT_REPO_ROOT/build_var/libsr_cert_$v.so
  rm -rf /tmp/pl_$v
  timeout -s KILL 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/pl_$v -o t -- python3 tools/quick_sparse_cert.py --exact 0 --check 0 --steps 1 > /dev/null 2>&1
  echo "== $v"; python3 tools/micro/cert_per_launch.py /tmp/pl_$v
done
