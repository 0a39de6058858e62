#!/bin/bash
# Runs named stages on the GPU box, each under its own timeout, logging to gpurun_out/<tag>_<stage>.log.
# usage: tools/gpu_stage.sh <tag> stage1 stage2 ...   (stages: see the case below)
tag=$1; shift
mkdir -p gpurun_out
export TMPDIR=/tmp
for st in "$@"; do
  log=gpurun_out/${tag}_${st}.log
  case $st in
    mlptests)  timeout 900 python -m pytest tests/test_gpu_networks.py tests/test_gpu_round2.py -q -k "mlp or pipeline or generate_chain or arbitrary or same_seed or noise" > $log 2>&1 ;;
    alltests)  timeout 2400 python -m pytest tests -m gpu -q > $log 2>&1 ;;
    newtests)  timeout 1800 python -m pytest tests/test_gpu_round2.py -q > $log 2>&1 ;;
    smoke)     timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $log 2>&1 ;;
    mlpbench)  timeout 300 python tools/bench_mlp.py > $log 2>&1 ;;
    mlptests8) DDMI_MLP2D_CW=8 timeout 900 python -m pytest tests/test_gpu_networks.py tests/test_gpu_round2.py -q -k "mlp or pipeline or arbitrary or noise" > $log 2>&1 ;;
    mlpbench8) DDMI_MLP2D_CW=8 timeout 300 python tools/bench_mlp.py --which 2d > $log 2>&1 ;;
    mlpts8)    DDMI_MLP2D_CW=8 DDMI_MLP2D_TS=1 timeout 300 python tools/bench_mlp.py --which 2d > $log 2>&1 ;;
    occvid)    timeout 1200 python -m pytest tests/test_gpu_occupancy.py tests/test_gpu_video.py -q > $log 2>&1 ;;
    train)     timeout 1500 python -m pytest tests/test_gpu_train.py -q -x > $log 2>&1 ;;
    trainall)  timeout 1500 python -m pytest tests/test_gpu_train.py -q > $log 2>&1 ;;
    wgradbench) timeout 600 python tools/bench_wgrad.py 128 > $log 2>&1 ;;
    wgradbenchA) DDMI_WGRAD_ALLFRAGS=4 timeout 600 python tools/bench_wgrad.py 128 > $log 2>&1 ;;
    stepbench) timeout 300 python tools/bench_step.py > $log 2>&1 ;;
    opstests)  timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_networks.py -q -k "philox or ddpm_step or pipeline or generate_chain" > $log 2>&1 ;;
    mlpts)     DDMI_MLP2D_TS=1 timeout 300 python tools/bench_mlp.py --which 2d > $log 2>&1 ;;
    bench)     timeout 1500 python bench.py --steps 1 --warmup 1 > gpurun_out/${tag}_bench.json 2> $log ;;
    benchfast) timeout 900 python bench.py --steps 1 --warmup 1 --no-ddim --no-parity-check --no-cpu-baseline > gpurun_out/${tag}_benchfast.json 2> $log ;;
    benchshape) timeout 900 python bench.py --workload shapenet --steps 1 --warmup 1 > gpurun_out/${tag}_bench_shapenet.json 2> $log ;;
    benchtrain) timeout 1200 python bench.py --workload train --steps 2 --warmup 1 > gpurun_out/${tag}_bench_train.json 2> $log ;;
    benchtrain64) timeout 1200 python bench.py --workload train --batch 64 --steps 3 --warmup 1 > gpurun_out/${tag}_bench_train64.json 2> $log ;;
    benchsky)  timeout 1200 python bench.py --workload sky --steps 1 --warmup 0 > gpurun_out/${tag}_bench_sky.json 2> $log ;;
    benchq)    timeout 900 python bench.py --nfe 10 --steps 1 --warmup 1 --no-ddim --no-parity-check --no-cpu-baseline > gpurun_out/${tag}_benchq.json 2> $log ;;
    r3tests)   timeout 1200 python -m pytest tests/test_gpu_round3.py -q -x > $log 2>&1 ;;
    benchq0)   DDMI_SKIP_FOLD=1 timeout 900 python bench.py --nfe 10 --steps 1 --warmup 1 --no-ddim --no-parity-check --no-cpu-baseline > gpurun_out/${tag}_benchq0.json 2> $log ;;
    conv1x1)   timeout 600 python tools/bench_conv.py --batch 512 --only " k1 " --cfgs 13,24,21 > $log 2>&1 ;;
    convvar)   timeout 900 python -m pytest tests/test_gpu_ops.py -q -k "every_bf16_variant" > $log 2>&1 ;;
    skyfull)   DDMI_METRICS_LOG=gpurun_out/${tag}_metrics.jsonl timeout 1500 python -m pytest tests/test_gpu_video.py -q -x -s -k "full_size_decode_vs_oracle" > $log 2>&1 ;;
    r3train)   timeout 1500 python -m pytest tests/test_gpu_round3.py -q -x -s -k "training or wgrad or ema or resume" > $log 2>&1 ;;
    benchfull) timeout 2400 python bench.py --steps 2 --warmup 1 > gpurun_out/${tag}_bench_n1_b512.json 2> $log ;;
    benchqt)   DDMI_TUNE_DEFAULT=/nonexistent DDMI_TUNE_FILE=$PWD/gpurun_out/${tag}_plans.txt timeout 900 python bench.py --nfe 10 --steps 1 --warmup 1 --no-ddim --no-parity-check --no-cpu-baseline > gpurun_out/${tag}_benchqt.json 2> $log ;;
    upbench)   timeout 600 python tools/bench_conv.py --batch 512 --only "up @" --cfgs 13,11,19,20 > $log 2>&1 ;;
    uptest)    timeout 900 python -m pytest tests/test_gpu_ops.py -q -k "upsample_parity_form" > $log 2>&1 ;;
    benchtrainL) DDMI_ATTN_BWD_LEGACY=1 timeout 1200 python bench.py --workload train --steps 2 --warmup 1 > gpurun_out/${tag}_bench_trainL.json 2> $log ;;
    attnbwd)   timeout 900 python -m pytest tests/test_gpu_round3.py tests/test_gpu_train.py -q -k "attention or upsample_conv or block_input or lse or concat" > $log 2>&1 ;;
    attnbench) timeout 300 python tools/bench_attn_mh.py 128 > $log 2>&1 ;;
    *) echo "unknown stage $st" ;;
  esac
  echo "== $st rc=$? =="; tail -n 6 $log
done
