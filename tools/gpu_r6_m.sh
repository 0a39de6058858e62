#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6m; mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_round6.py tests/test_gpu_train.py -x -q -m gpu -s -k "video or triplane or sky_training or attention" > $O/tests.log 2>&1
echo "tests rc=$?"; grep -v amdgpu.ids $O/tests.log | tail -8
for pad in 1 0; do
  DDMI_TRAIN_ATTN_PAD=$pad timeout 600 python3 bench.py --workload trainvideo --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read());print('pad=$pad', d['value'], d['ms_per_step'], json.dumps(d['kernel_time_share'])[:400])"
done
