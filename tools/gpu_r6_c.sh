#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6c
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_round6.py -x -q -m gpu -s > $O/tests.log 2>&1
echo "tests rc=$?"; tail -12 $O/tests.log
for B in 1 8 16; do
  echo "== max_batch=$B small-batch regime on"; python3 tools/bench_small_batch.py --batches $B 2>&1 | grep '"rows"'
done
echo "== max_batch=32 (regime off by the bound)"; python3 tools/bench_small_batch.py --batches 32 --plain-only 2>&1 | grep '"rows"'
echo "== max_batch=32 with DDMI_SMALL_BATCH_MAX=32"; DDMI_SMALL_BATCH_MAX=32 python3 tools/bench_small_batch.py --batches 32 --plain-only 2>&1 | grep '"rows"'
echo "== max_batch=16 regime off"; DDMI_SMALL_BATCH=0 python3 tools/bench_small_batch.py --batches 16 --plain-only 2>&1 | grep '"rows"'
TAG=_small BATCHES="1 8" bash tools/gpu_r6_trace.sh
