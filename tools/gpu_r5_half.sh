#!/bin/bash
# Round 5: the halo kernel's two-taps-per-stage tail (Cin = 64 k + 32) - parity test, per-shape A/B, sampling loop A/B
tag=${1:-r5b}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_round5.py -x -q > gpurun_out/${tag}_tests.log 2>&1; tail -3 gpurun_out/${tag}_tests.log
for h in 0 1; do
  DDMI_HALO_HALF=$h timeout 300 python tools/bench_conv.py --batch 512 --only "224->" --cfgs 19 > gpurun_out/${tag}_conv_half$h.log 2>&1
  DDMI_HALO_HALF=$h timeout 300 python tools/bench_conv.py --batch 512 --only "672->" --cfgs 19 >> gpurun_out/${tag}_conv_half$h.log 2>&1
done
paste -d'\n' gpurun_out/${tag}_conv_half0.log gpurun_out/${tag}_conv_half1.log | grep "k3 @" 
Q="--nfe 10 --steps 2 --warmup 1 --no-ddim --no-parity-check --no-cpu-baseline --no-workloads"
for h in 0 1 0 1; do
  DDMI_HALO_HALF=$h timeout 600 python bench.py $Q > gpurun_out/${tag}_bench_half$h.json 2>> gpurun_out/${tag}_bench.log
  python -c "
import json;d=json.loads(open('gpurun_out/${tag}_bench_half$h.json').read().strip().splitlines()[-1])
print('half=$h samples/s',round(d['value'],2),'loop ms',d['last_step_phase_ms']['sampling_loop'],'halo us',d['roofline']['avg_launch_us'],'frac',d['roofline']['frac'])"
done
