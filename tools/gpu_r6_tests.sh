#!/bin/bash
# round 6: the new GPU tests + the round-5 FGN tests on the rewritten epilogue, then the small-batch kernel tables
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6b
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_round6.py tests/test_gpu_round5.py -x -q -m gpu -s > $O/tests.log 2>&1
echo "tests rc=$?"; tail -15 $O/tests.log
bash tools/gpu_r6_small.sh > $O/small.log 2>&1
head -30 gpurun_out/r6a/kernel_stats_b1.csv | cut -c1-160
