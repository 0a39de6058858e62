#!/bin/bash
# rocprofv3 kernel stats of one bench workload: tools/gpu_profile_workload.sh <tag> <workload> [extra bench flags]
tag=$1; wl=$2; shift; shift
mkdir -p gpurun_out
export TMPDIR=/tmp
CMD="bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline $*"
(cd /tmp; timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/gpurun_out/${tag}_${wl}_rocprof -- python3 $OLDPWD/$CMD > $OLDPWD/gpurun_out/${tag}_${wl}_rocprof.log 2>&1)
echo "rocprof $wl rc=$?"
f=$(find gpurun_out/${tag}_${wl}_rocprof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f gpurun_out/${tag}_${wl}_kernel_stats.csv && head -12 gpurun_out/${tag}_${wl}_kernel_stats.csv | cut -c1-160
tail -n 2 gpurun_out/${tag}_${wl}_rocprof.log | cut -c1-600
rm -rf gpurun_out/${tag}_${wl}_rocprof
