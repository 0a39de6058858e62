#!/bin/bash
# Per-workload evidence (PMC traffic + rocprofv3 kernel statistics of the shapenet / sky / train legs, the training step at B = 128):
#   DDMI_HEAD=<sha> tools/gpu_round5_workloads.sh <tag>
tag=$1
export DDMI_HEAD=${DDMI_HEAD:-unknown}
export TMPDIR=/tmp
mkdir -p gpurun_out
for wl in shapenet sky train; do tools/gpu_pmc_workload.sh $tag $wl | tail -3; done
for wl in shapenet sky; do tools/gpu_profile_workload.sh $tag $wl | tail -4; done
timeout 600 python bench.py --workload train --batch 128 --no-cpu-baseline > gpurun_out/${tag}_bench_train_n1_b128.json 2> gpurun_out/${tag}_bench_train_b128.log
echo "train B=128 rc=$?"; head -c 300 gpurun_out/${tag}_bench_train_n1_b128.json; echo
