#!/bin/bash
# HBM traffic per launch of one bench workload's kernels (two rocprofv3 --pmc passes, nothing else traced) -> the summary bench.py's
# `roofline.traffic` of that workload looks up:   DDMI_HEAD=<sha> tools/gpu_pmc_workload.sh <tag> <shapenet|sky|train> [extra bench flags]
tag=$1; wl=$2; shift; shift
mkdir -p gpurun_out
export TMPDIR=/tmp
CMD="bench.py --workload $wl --steps 1 --warmup 1 --no-cpu-baseline $*"
[ "$wl" = train ] && CMD="$CMD --no-graph"
(cd /tmp; timeout 1200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OLDPWD/gpurun_out/${tag}_${wl}_pf -- python3 $OLDPWD/$CMD > $OLDPWD/gpurun_out/${tag}_${wl}_pf.log 2>&1)
echo "pmc fetch $wl rc=$?"
(cd /tmp; timeout 1200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OLDPWD/gpurun_out/${tag}_${wl}_pw -- python3 $OLDPWD/$CMD > $OLDPWD/gpurun_out/${tag}_${wl}_pw.log 2>&1)
echo "pmc write $wl rc=$?"
python tools/pmc_summary.py gpurun_out/${tag}_${wl}_pf gpurun_out/${tag}_${wl}_pw "rocprofv3 --pmc {FETCH_SIZE|WRITE_SIZE} -- python3 $CMD" > gpurun_out/${tag}_pmc_traffic_${wl}.json
echo "summary rc=$?"; head -c 900 gpurun_out/${tag}_pmc_traffic_${wl}.json; echo
rm -rf gpurun_out/${tag}_${wl}_pf gpurun_out/${tag}_${wl}_pw
