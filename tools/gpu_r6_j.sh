#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6j; mkdir -p $O
t0=$(date +%s)
timeout 900 python3 bench.py --workload trainvideo --steps 4 --warmup 1 --no-cpu-baseline > $O/bench_trainvideo.json 2> $O/bench_trainvideo.log
echo "trainvideo rc=$? wall $(( $(date +%s) - t0 )) s"; head -c 1500 $O/bench_trainvideo.json; echo; tail -3 $O/bench_trainvideo.log
