#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6g; mkdir -p $O
timeout 2400 python -m pytest tests/ -x -q -m gpu > $O/alltests.log 2>&1
echo "alltests rc=$?"; tail -6 $O/alltests.log
python3 tools/bench_x3.py 128 bf16x3 2>&1 | grep -v amdgpu.ids
