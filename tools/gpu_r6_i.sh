#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6i; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_round6.py -x -q -m gpu -k "video or triplane" > $O/tests.log 2>&1
echo "video tests rc=$?"; tail -25 $O/tests.log
timeout 1500 python -m pytest tests/test_gpu_train.py -x -q -m gpu > $O/train_tests.log 2>&1
echo "train tests rc=$?"; tail -4 $O/train_tests.log
