#!/bin/bash
# Round 5: producer-side GroupNorm + the halo kernel's two-tap tail: parity tests, per-shape A/B, sampling-loop A/B (10 NFE)
tag=${1:-r5c}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_round5.py -x -q > gpurun_out/${tag}_tests.log 2>&1; tail -4 gpurun_out/${tag}_tests.log
for h in 0 1; do
  DDMI_HALO_HALF=$h timeout 300 python tools/bench_conv.py --batch 512 --only "224->" --cfgs 19 > gpurun_out/${tag}_conv_half$h.log 2>&1
  DDMI_HALO_HALF=$h timeout 300 python tools/bench_conv.py --batch 512 --only "672->" --cfgs 19 >> gpurun_out/${tag}_conv_half$h.log 2>&1
done
paste -d'\n' gpurun_out/${tag}_conv_half0.log gpurun_out/${tag}_conv_half1.log | grep "k3 @"
Q="--nfe 10 --steps 2 --warmup 1 --no-ddim --no-parity-check --no-cpu-baseline --no-workloads"
for cfg in "1 1" "0 0" "1 0" "0 1" "1 1" "0 0"; do
  set -- $cfg
  DDMI_GN_FUSE=$1 DDMI_HALO_HALF=$2 timeout 600 python bench.py $Q > gpurun_out/${tag}_bench_f$1h$2.json 2>> gpurun_out/${tag}_bench.log
  python -c "
import json;d=json.loads(open('gpurun_out/${tag}_bench_f$1h$2.json').read().strip().splitlines()[-1])
sh=d['kernel_time_share']
print('fuse=$1 half=$2 samples/s',round(d['value'],2),'loop ms',d['last_step_phase_ms']['sampling_loop'],'halo us',d['roofline']['avg_launch_us'],'gn_apply',sh.get('gn_apply'),'conv+gn',sh.get('conv_halo+gn<bf16,256x224>'), 'algfrac', d.get('unet_step_algorithmic_frac'))"
done
