#!/bin/bash
# round 6 (verdict item 7): the sky decode leg's attention launches BY SHAPE - duration (kernel trace), HBM bytes (two --pmc passes, gfx950
# corrections as tools/pmc_summary.py applies them) and the fraction of the MFMA peak -> gpurun_out/r6_sky_attention_by_shape.json
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6s; mkdir -p $O
R=$PWD
(cd /tmp; timeout 900 rocprofv3 --kernel-trace --output-format csv -d $R/$O/kt -- python3 $R/tools/sky_decode_once.py > $R/$O/kt.log 2>&1); echo "kt rc=$?"
(cd /tmp; timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/$O/pf -- python3 $R/tools/sky_decode_once.py > $R/$O/pf.log 2>&1); echo "pf rc=$?"
(cd /tmp; timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/$O/pw -- python3 $R/tools/sky_decode_once.py > $R/$O/pw.log 2>&1); echo "pw rc=$?"
python3 tools/sky_attn_table.py $O > gpurun_out/r6_sky_attention_by_shape.json; echo "table rc=$?"
cat gpurun_out/r6_sky_attention_by_shape.json | head -c 3000
rm -rf $O/kt $O/pf $O/pw
