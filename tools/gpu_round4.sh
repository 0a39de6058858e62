#!/bin/bash
# Round-4 evidence in one GPU call: tools/gpu_round4.sh <tag>   (DDMI_HEAD = the commit the numbers are made on)
tag=$1
export DDMI_HEAD=${DDMI_HEAD:-unknown}
tools/gpu_profiles.sh $tag
for wl in shapenet sky train; do tools/gpu_pmc_workload.sh $tag $wl; done
tools/gpu_attnblock_pmc.sh 512 > gpurun_out/${tag}_attnblock_pmc.txt 2>&1
DDMI_ATTNBLOCK_TS=1 python tools/bench_attnblock.py 512 10 > gpurun_out/${tag}_attnblock_bench.txt 2>&1
python tools/peaks.py > gpurun_out/${tag}_measured_peaks.txt 2>&1
ls -la gpurun_out | tail -30
