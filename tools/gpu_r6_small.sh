#!/bin/bash
# round 6: the small-batch regime of the AFHQ UNet - timings (plain / graph) and a rocprofv3 kernel table at B = 1 and B = 8
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp DDMI_TUNE_DEFAULT="$PWD/ddmi_amd/tuned/conv_plans_mi355x.txt"
O=gpurun_out/r6a
mkdir -p $O
python3 tools/bench_small_batch.py --batches 1,2,4,8,16 --json $O/small_batch.json > $O/small_batch.log 2>&1
for B in 1 8; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_b$B -- python3 tools/bench_small_batch.py --batches $B --plain-only --iters 20 > $O/prof_b$B.log 2>&1
  f=$(find $O/prof_b$B -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp "$f" $O/kernel_stats_b$B.csv
  rm -rf $O/prof_b$B
done
tail -3 $O/small_batch.log
