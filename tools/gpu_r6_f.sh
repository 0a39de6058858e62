#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
bash tools/gpu_r6_e.sh
echo "=== attnblock: shipped build vs AB_DBG=16 (affine transform on every x fragment, timing only)"
python3 tools/bench_attnblock.py 512 10 2>&1 | grep -v amdgpu.ids
DDMI_LIB_PATH=$PWD/ddmi_amd/lib/variants/libddmi_hip_abaff.so python3 tools/bench_attnblock.py 512 10 2>&1 | grep -v amdgpu.ids
bash tools/gpu_r6_sky_attn.sh
