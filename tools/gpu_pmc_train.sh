#!/bin/bash
# HBM traffic per launch of the training step's kernels (two rocprofv3 --pmc passes, nothing else traced): tools/gpu_pmc_train.sh <tag>
tag=$1
mkdir -p gpurun_out
export TMPDIR=/tmp
CMD="bench.py --workload train --steps 1 --warmup 1 --no-cpu-baseline --no-graph"
(cd /tmp; timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OLDPWD/gpurun_out/${tag}_pmc_tf -- python3 $OLDPWD/$CMD > $OLDPWD/gpurun_out/${tag}_pmc_tf.log 2>&1)
echo "pmc fetch rc=$?"
(cd /tmp; timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OLDPWD/gpurun_out/${tag}_pmc_tw -- python3 $OLDPWD/$CMD > $OLDPWD/gpurun_out/${tag}_pmc_tw.log 2>&1)
echo "pmc write rc=$?"
python tools/pmc_summary.py gpurun_out/${tag}_pmc_tf gpurun_out/${tag}_pmc_tw "rocprofv3 --pmc {FETCH_SIZE|WRITE_SIZE} -- python3 $CMD (B=128 default batch; steps / warm-up as in the command)" > gpurun_out/${tag}_pmc_traffic_train_b128.json
echo "summary rc=$?"; head -c 1500 gpurun_out/${tag}_pmc_traffic_train_b128.json
rm -rf gpurun_out/${tag}_pmc_tf gpurun_out/${tag}_pmc_tw
