#!/bin/bash
# Round-5 evidence in one GPU call (after tools/gpu_profiles.sh <tag>a for the AFHQ leg): tools/gpu_round5.sh <tag>
#   the driver's command line, PMC traffic + rocprofv3 kernel statistics per workload leg, the training step at B = 128
tag=$1
export DDMI_HEAD=${DDMI_HEAD:-unknown}
export TMPDIR=/tmp
mkdir -p gpurun_out
t0=$(date +%s)
timeout 1700 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${tag}_driverlike_bench_n1_steps20.json 2> gpurun_out/${tag}_driverlike.log
echo "driver-like bench rc=$? wall $(( $(date +%s) - t0 )) s"; head -c 300 gpurun_out/${tag}_driverlike_bench_n1_steps20.json; echo
for wl in shapenet sky train; do tools/gpu_pmc_workload.sh $tag $wl | tail -4; done
for wl in shapenet sky; do tools/gpu_profile_workload.sh $tag $wl | tail -6; done
timeout 600 python bench.py --workload train --batch 128 --no-cpu-baseline > gpurun_out/${tag}_bench_train_n1_b128.json 2> gpurun_out/${tag}_bench_train_b128.log
echo "train B=128 rc=$?"; head -c 400 gpurun_out/${tag}_bench_train_n1_b128.json; echo
python tools/peaks.py > gpurun_out/${tag}_measured_peaks.txt 2>&1
ls -la gpurun_out | tail -12
