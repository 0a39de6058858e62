#!/bin/bash
# round 6: bench.py sanity at a short chain (new small_batch block, sky cpu_baseline) and the headline's batch sweep
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp DDMI_TUNE_DEFAULT="$PWD/ddmi_amd/tuned/conv_plans_mi355x.txt"
O=gpurun_out/r6e
mkdir -p $O
timeout 900 python3 bench.py --nfe 10 --steps 1 --warmup 1 --no-parity-check --no-ddim --no-workloads --no-cpu-baseline > $O/bench_short.json 2> $O/bench_short.log
echo "short rc=$?"; python3 -c "
import json;d=json.load(open('$O/bench_short.json'));print(json.dumps(d['small_batch'])[:900]);print(d['value'], d['unet_step_algorithmic_frac'])"
for B in 512 768 1024; do
  timeout 900 python3 bench.py --nfe 20 --steps 1 --warmup 1 --batch $B --no-parity-check --no-ddim --no-workloads --no-cpu-baseline --no-small-batch > $O/bench_b$B.json 2> $O/bench_b$B.log
  python3 -c "
import json;d=json.load(open('$O/bench_b$B.json'));print($B, 'samples/s at nfe 20', d['value'], 'algorithmic frac', d['unet_step_algorithmic_frac'], d['last_step_phase_ms'], d['unet_memory_bytes'])"
done
timeout 900 python3 bench.py --workload sky --steps 1 --warmup 0 --nfe 4 > $O/bench_sky.json 2> $O/bench_sky.log
echo "sky rc=$?"; python3 -c "
import json;d=json.load(open('$O/bench_sky.json'));print(json.dumps(d['cpu_baseline'])[:900])"
