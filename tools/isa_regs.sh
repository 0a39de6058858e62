#!/bin/bash
# register / spill counts of the kernels of one csrc file: tools/isa_regs.sh k_conv.hip [grep pattern] [extra hipcc flags]
src=$1; pat=${2:-.}; shift; shift
mkdir -p /tmp/isa
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=on --cuda-device-only "$@" -c ddmi_amd/csrc/$src -o /tmp/isa/x.co -Iinclude || exit 1
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=/tmp/isa/x.co --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=/tmp/isa/x.elf
/opt/rocm/lib/llvm/bin/llvm-readelf --notes /tmp/isa/x.elf | grep -E "\.name:|\.vgpr_count|\.sgpr_count|spill_count|\.agpr_count" | paste - - - - - - | sed 's/ \+/ /g; s/\.agpr_count: //; s/\.name: //' | grep -E "$pat" | cut -c1-220
