#!/bin/bash
# (a) the whole parity suite with the small-batch regime ON (the suite pins the large-batch kernels by default: tests/conftest.py);
# (b) the round-5 / round-6 hand-off tests on the final tree
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6l; mkdir -p $O
DDMI_SMALL_BATCH=1 DDMI_NO_BF16_BASELINES=1 timeout 2400 python -m pytest tests/ -q -m gpu > $O/alltests_regime_on.log 2>&1
echo "regime-on suite rc=$?"; tail -30 $O/alltests_regime_on.log | cut -c1-300
timeout 1500 python -m pytest tests/test_gpu_round6.py tests/test_gpu_round5.py -x -q -m gpu > $O/r56.log 2>&1
echo "r5+r6 rc=$?"; tail -3 $O/r56.log
