#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6c
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_round6.py -x -q -m gpu -k "small or one_launch or split_k" > $O/tests2.log 2>&1
echo "tests rc=$?"; tail -5 $O/tests2.log
for B in 1 8 16; do
  echo "== max_batch=$B small-batch regime on"; python3 tools/bench_small_batch.py --batches $B --plain-only 2>&1 | grep '"rows"'
done
TAG=_small BATCHES="1" bash tools/gpu_r6_trace.sh
