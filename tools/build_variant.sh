#!/bin/bash
# Builds a variant of ONE source file with extra flags and links it with the other objects of the current build:
#   tools/build_variant.sh NAME k_attnblock.hip -DAB_DBG=1   ->  ddmi_amd/lib/variants/libddmi_hip_NAME.so
# Use on the GPU box with DDMI_LIB_PATH=... (ddmi_amd/_lib.py) to A/B two builds in one call.
set -e
cd "$(dirname "$0")/.."
name=$1; src=$2; shift 2
mkdir -p ddmi_amd/lib/variants
obj=ddmi_amd/lib/variants/${src%.hip}_$name.o
extra=""
case $src in k_attention.hip|k_attn_bwd.hip|k_attnblock.hip) extra="-mllvm -amdgpu-mfma-vgpr-form -fno-slp-vectorize";; esac
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on $extra "$@" -c ddmi_amd/csrc/$src -o $obj
objs=$(ls ddmi_amd/lib/obj/*.o | grep -v "/${src%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ddmi_amd/lib/variants/libddmi_hip_$name.so $objs $obj
echo ddmi_amd/lib/variants/libddmi_hip_$name.so
