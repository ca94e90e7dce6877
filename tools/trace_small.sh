#!/bin/bash
# development aid (GPU box): workgroup phase timelines (SR_IGEMM_TRACE build of igemm.hip, linked against the shipped objects) of the
# small-batch igemm shapes: bash tools/trace_small.sh
set -e
R=$GRAFT_REPO_ROOT
C=$R/stable-renderer_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DSR_IGEMM_TRACE=1 -c $C/igemm.hip -o /tmp/igemm_trace.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libsr_trace.so /tmp/igemm_trace.o $(ls $C/_obj/*.o | grep -v igemm.o)
export SR_DEV_LIB=/tmp/libsr_trace.so
for spec in "4 8192 1 1 320 320 1" "13 8192 1 1 320 320 1" "3 8192 1 1 320 320 1" "13 512 1 1 1280 1280 1" "4 2048 1 1 640 640 1" "13 2048 1 1 640 640 1" "2 2 64 64 320 320 3" "15 2 64 64 320 320 3" "14 2 8 8 1280 1280 3"; do
  set -- $spec
  echo "== tile $1: B$2 $3x$4 C$5 N$6 k$7"
  SR_IGEMM_TILE=$1 python3 $R/tools/trace_igemm.py $2 $3 $4 $5 $6 $7 2>&1 | grep -v amdgpu
done
