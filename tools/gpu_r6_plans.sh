#!/bin/bash
# round 6: the convolution plans the autotuner picks for the AFHQ UNet at a small batch, with their times (DDMI_TUNE_FILE)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6p; mkdir -p $O
for B in 8; do
  rm -f $O/plans_b$B.txt
  DDMI_TUNE_DEFAULT=/nonexistent DDMI_TUNE_FILE=$PWD/$O/plans_b$B.txt python3 tools/bench_small_batch.py --batches $B --plain-only 2>&1 | grep '"rows"'
done
