#!/bin/bash
# round 6: B = 1 and B = 8 as the handle's max_batch (autotuned at that batch), and the A/B switches that matter there
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6a
for B in 1 8; do
  echo "== max_batch=$B tuned";            python3 tools/bench_small_batch.py --batches $B --plain-only 2>&1 | grep '"rows"'
  echo "== max_batch=$B tuned, CAT off";   DDMI_CONV_CAT=0 python3 tools/bench_small_batch.py --batches $B --plain-only 2>&1 | grep '"rows"'
  echo "== max_batch=$B tuned, attn 5-launch"; DDMI_ATTN_FUSED=0 python3 tools/bench_small_batch.py --batches $B --plain-only 2>&1 | grep '"rows"'
  echo "== max_batch=$B tuned, gn fuse 2"; DDMI_GN_FUSE=2 python3 tools/bench_small_batch.py --batches $B --plain-only 2>&1 | grep '"rows"'
done
