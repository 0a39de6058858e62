#!/bin/bash
# Round 5, late: SiLU on the hardware reciprocal instead of __frcp_rn's refined one (bf16 paths) - parity tests, then A/B against a library
# built from the previous commit's k_conv.hip / k_groupnorm.hip (ddmi_amd/lib/variants/libddmi_hip_oldrcp.so)
tag=${1:-r5k}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_round5.py tests/test_gpu_ops.py tests/test_gpu_train.py -q -x -k "groupnorm or gn or resblock or fgn or norm" > gpurun_out/${tag}_tests.log 2>&1; tail -3 gpurun_out/${tag}_tests.log
OLD=$PWD/ddmi_amd/lib/variants/libddmi_hip_oldrcp.so
FGN_MODES=2 FGN_ITERS=6 DDMI_CONV_TS=1 timeout 200 python tools/dbg/fgn_scale.py 512 224 64 224 2>&1 | grep "mean per workgroup"
DDMI_LIB_PATH=$OLD FGN_MODES=2 FGN_ITERS=6 DDMI_CONV_TS=1 timeout 200 python tools/dbg/fgn_scale.py 512 224 64 224 2>&1 | grep "mean per workgroup"
Q="--nfe 10 --steps 2 --warmup 1 --no-ddim --no-parity-check --no-cpu-baseline --no-workloads"
for lib in new old new old; do
  if [ $lib = old ]; then export DDMI_LIB_PATH=$OLD; else unset DDMI_LIB_PATH; fi
  timeout 400 python bench.py $Q > gpurun_out/${tag}_bench_$lib.json 2>> gpurun_out/${tag}_bench.log
  python -c "
import json;d=json.loads(open('gpurun_out/${tag}_bench_$lib.json').read().strip().splitlines()[-1])
sh=d['kernel_time_share']
print('$lib samples/s',round(d['value'],2),'loop ms',d['last_step_phase_ms']['sampling_loop'],d['roofline']['kernel'],d['roofline']['avg_launch_us'],d['roofline']['frac'],'gn_apply',sh.get('gn_apply'),'algfrac', d.get('unet_step_algorithmic_frac'))"
done
