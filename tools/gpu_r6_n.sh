#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6n; mkdir -p $O
timeout 2400 python -m pytest tests/ -x -q -m gpu > $O/alltests.log 2>&1
echo "alltests rc=$?"; tail -4 $O/alltests.log
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
