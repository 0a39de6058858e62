#!/bin/bash
# Round-6 evidence at the shipped tree, part 1: the AFHQ leg's set (bench JSON, rocprofv3 kernel statistics, PMC traffic, sustained clock:
# tools/gpu_profiles.sh), then per workload PMC traffic + kernel statistics + clock, the training step at B = 128, the measured peaks.
#   DDMI_HEAD=<sha> tools/gpu_round6.sh <tag>
cd "$GRAFT_REPO_ROOT" || exit 1
tag=$1
export DDMI_HEAD=${DDMI_HEAD:-unknown}
export TMPDIR=/tmp
mkdir -p gpurun_out
bash tools/gpu_profiles.sh $tag 2>&1 | tail -30
for wl in shapenet sky train; do tools/gpu_pmc_workload.sh $tag $wl | tail -3; done
for wl in shapenet sky train; do tools/gpu_profile_workload.sh $tag $wl | tail -4; done
for wl in shapenet sky train; do tools/gpu_clock_workload.sh $tag $wl | tail -3; done
timeout 600 python bench.py --workload train --batch 128 --no-cpu-baseline > gpurun_out/${tag}_bench_train_n1_b128.json 2> gpurun_out/${tag}_bench_train_b128.log
echo "train B=128 rc=$?"; head -c 300 gpurun_out/${tag}_bench_train_n1_b128.json; echo
python tools/peaks.py > gpurun_out/${tag}_measured_peaks.txt 2>&1
