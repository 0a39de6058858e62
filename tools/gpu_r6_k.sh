#!/bin/bash
# the whole GPU suite + smoke + the driver's command on the final tree
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6k; mkdir -p $O
timeout 2400 python -m pytest tests/ -x -q -m gpu > $O/alltests.log 2>&1
echo "alltests rc=$?"; tail -4 $O/alltests.log
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
t0=$(date +%s)
timeout 1900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/driverlike_final.json 2> $O/driverlike_final.log
echo "driver-like bench rc=$? wall $(( $(date +%s) - t0 )) s"; head -c 300 $O/driverlike_final.json; echo
