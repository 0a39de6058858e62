#!/bin/bash
# Round-6 evidence, part 2: the driver's own command line, timed.   DDMI_HEAD=<sha> tools/gpu_round6_driver.sh <tag>
cd "$GRAFT_REPO_ROOT" || exit 1
tag=$1
export TMPDIR=/tmp
mkdir -p gpurun_out
t0=$(date +%s)
timeout 1900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${tag}_driverlike_bench_n1_steps20.json 2> gpurun_out/${tag}_driverlike.log
echo "driver-like bench rc=$? wall $(( $(date +%s) - t0 )) s"; head -c 400 gpurun_out/${tag}_driverlike_bench_n1_steps20.json; echo
