#!/bin/bash
# Sustained shader clock under one bench workload's kernels (one rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace pass, nothing else traced)
# -> the summary bench.py's `roofline.clock` of that workload looks up:
#   DDMI_HEAD=<sha> tools/gpu_clock_workload.sh <tag> <shapenet|sky|train> [extra bench flags]   ->  gpurun_out/<tag>_effective_clock_<workload>.json
tag=$1; wl=$2; shift; shift
mkdir -p gpurun_out
export TMPDIR=/tmp
CMD="bench.py --workload $wl --steps 1 --warmup 1 --no-cpu-baseline $*"
[ "$wl" = train ] && CMD="$CMD --no-graph"
(cd /tmp; timeout 1200 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OLDPWD/gpurun_out/${tag}_${wl}_pc -- python3 $OLDPWD/$CMD > $OLDPWD/gpurun_out/${tag}_${wl}_pc.log 2>&1)
echo "pmc clock $wl rc=$?"
python tools/pmc_clock.py gpurun_out/${tag}_${wl}_pc "rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -- python3 $CMD" > gpurun_out/${tag}_effective_clock_${wl}.json
echo "summary rc=$?"; head -c 700 gpurun_out/${tag}_effective_clock_${wl}.json; echo
rm -rf gpurun_out/${tag}_${wl}_pc
