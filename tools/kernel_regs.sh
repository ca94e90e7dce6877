#!/bin/bash
# development aid (no GPU needed): registers / spills / LDS of every kernel of one source file: bash tools/kernel_regs.sh igemm.hip [extra flags]
SRC=$1; shift
D=$(dirname "$0")/../stable-renderer_amd/csrc
OUT=/tmp/sr_co_$$
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only "$@" -c $D/$SRC -o $OUT.bundle || exit 1
/opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$OUT.bundle --output=$OUT.co --unbundle || exit 1
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $OUT.co | grep -E "\.name:|\.vgpr_count|\.vgpr_spill_count|\.private_segment_fixed_size|\.group_segment_fixed_size|\.sgpr_spill" | paste - - - - - - | sed 's/  */ /g'
rm -f $OUT.bundle $OUT.co
