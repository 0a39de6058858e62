#!/bin/bash
# round 6: the launch-by-launch trace of ONE AFHQ UNet evaluation at a small batch (rocprofv3 --kernel-trace), env passed through
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6a
mkdir -p $O
for B in ${BATCHES:-1 8}; do
  rocprofv3 --kernel-trace --output-format csv -d $O/trace_b$B -- python3 tools/bench_small_batch.py --batches $B --plain-only --iters 2 > $O/trace_b$B.log 2>&1
  f=$(find $O/trace_b$B -name '*kernel_trace.csv' | head -1)
  [ -n "$f" ] && python3 tools/trace_last_eval.py "$f" > $O/trace${TAG}_b$B.txt
  rm -rf $O/trace_b$B
  grep '"rows"' $O/trace_b$B.log | tail -1
done
