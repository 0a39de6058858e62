#!/bin/bash
# round 6: kernel tables of one AFHQ UNet evaluation at B = 1 / 8 without the autotuner's launches in the trace
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp DDMI_AUTOTUNE=0
O=gpurun_out/r6a
mkdir -p $O
for B in 1 8; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_b$B -- python3 tools/bench_small_batch.py --batches $B --plain-only --iters 20 > $O/prof_b$B.log 2>&1
  f=$(find $O/prof_b$B -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp "$f" $O/kernel_stats_notune_b$B.csv
  rm -rf $O/prof_b$B
  tail -2 $O/prof_b$B.log
done
