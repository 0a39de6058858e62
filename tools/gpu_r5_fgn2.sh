#!/bin/bash
tag=${1:-r5d}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_round5.py -x -q > gpurun_out/${tag}_tests.log 2>&1; tail -4 gpurun_out/${tag}_tests.log
Q="--nfe 10 --steps 2 --warmup 1 --no-ddim --no-parity-check --no-cpu-baseline --no-workloads"
for cfg in "1 1 1" "0 0 0" "1 0 0" "0 1 1" "1 1 1" "0 0 0"; do
  set -- $cfg
  DDMI_GN_FUSE=$1 DDMI_HALO_HALF=$2 DDMI_CONV_CAT=$3 timeout 300 python bench.py $Q > gpurun_out/${tag}_bench_f$1h$2c$3.json 2>> gpurun_out/${tag}_bench.log
  python -c "
import json;d=json.loads(open('gpurun_out/${tag}_bench_f$1h$2c$3.json').read().strip().splitlines()[-1])
sh=d['kernel_time_share']
print('fuse=$1 half=$2 cat=$3 samples/s',round(d['value'],2),'loop ms',d['last_step_phase_ms']['sampling_loop'],'halo us',d['roofline']['avg_launch_us'],'gn_apply',sh.get('gn_apply'),'conv+gn',sh.get('conv_halo+gn<bf16,256x224>'), '1x1', sh.get('conv_gemm1x1<bf16,256x224,persistent>'), 'cat', sh.get('conv_gemm1x1_cat<bf16,256x224,persistent>'), 'algfrac', d.get('unet_step_algorithmic_frac'))"
done
