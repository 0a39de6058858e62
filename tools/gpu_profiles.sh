#!/bin/bash
# Round-N evidence for profiles/: bench JSON + the conv plans it tuned, rocprofv3 kernel stats of a short run of the same
# command, and the two PMC passes for HBM traffic (separate runs, --pmc only).   usage: tools/gpu_profiles.sh <tag>
tag=$1
mkdir -p gpurun_out
export TMPDIR=/tmp
export DDMI_TUNE_FILE=$PWD/gpurun_out/${tag}_conv_plans.txt
rm -f $DDMI_TUNE_FILE
timeout 1500 python bench.py --steps 2 --warmup 1 > gpurun_out/${tag}_bench_n1_b512.json 2> gpurun_out/${tag}_bench.log
echo "bench rc=$?"; tail -c 600 gpurun_out/${tag}_bench_n1_b512.json | head -c 400; echo
SHORT="bench.py --nfe 6 --steps 1 --warmup 0 --no-cpu-baseline --no-parity-check --no-ddim"
(cd /tmp; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/gpurun_out/${tag}_rocprof -- python3 $OLDPWD/$SHORT > $OLDPWD/gpurun_out/${tag}_rocprof.log 2>&1)
echo "rocprof rc=$?"
f=$(find gpurun_out/${tag}_rocprof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/${tag}_kernel_stats.csv && head -8 gpurun_out/${tag}_kernel_stats.csv
(cd /tmp; timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OLDPWD/gpurun_out/${tag}_pmc_f -- python3 $OLDPWD/$SHORT > $OLDPWD/gpurun_out/${tag}_pmc_f.log 2>&1)
echo "pmc fetch rc=$?"
(cd /tmp; timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OLDPWD/gpurun_out/${tag}_pmc_w -- python3 $OLDPWD/$SHORT > $OLDPWD/gpurun_out/${tag}_pmc_w.log 2>&1)
echo "pmc write rc=$?"
python tools/pmc_summary.py gpurun_out/${tag}_pmc_f gpurun_out/${tag}_pmc_w "rocprofv3 --pmc {FETCH_SIZE|WRITE_SIZE} -- python3 $SHORT (B=512)" > gpurun_out/${tag}_pmc_traffic_b512.json
echo "pmc summary rc=$?"; head -c 700 gpurun_out/${tag}_pmc_traffic_b512.json
# keep the merge small: drop the raw per-dispatch CSVs
rm -rf gpurun_out/${tag}_pmc_f gpurun_out/${tag}_pmc_w gpurun_out/${tag}_rocprof
