#!/bin/bash
# rocprofv3 kernel statistics of the training workload (eager launches, so kernel names are the library's own).
# usage: tools/gpu_trainprof.sh <tag> [batch]
tag=$1; B=${2:-64}
mkdir -p gpurun_out
export TMPDIR=/tmp
(cd /tmp; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/gpurun_out/${tag}_trprof -- python3 $OLDPWD/bench.py --workload train --batch $B --steps 2 --warmup 1 --no-graph > $OLDPWD/gpurun_out/${tag}_trprof.log 2>&1)
echo "rocprof rc=$?"
f=$(find gpurun_out/${tag}_trprof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/${tag}_train_kernel_stats.csv && head -40 gpurun_out/${tag}_train_kernel_stats.csv | cut -c1-200
rm -rf gpurun_out/${tag}_trprof
