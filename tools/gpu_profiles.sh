#!/bin/bash
# Round-N evidence for profiles/: bench JSON + the conv plans it tuned, rocprofv3 kernel stats of a short run of the same
# command, the two PMC passes for HBM traffic and the GRBM_GUI_ACTIVE pass for the sustained clock (separate runs, --pmc only /
# --pmc + --kernel-trace).   usage: DDMI_HEAD=<sha> tools/gpu_profiles.sh <tag>
tag=$1
mkdir -p gpurun_out
export TMPDIR=/tmp
export DDMI_TUNE_FILE=$PWD/gpurun_out/${tag}_conv_plans.txt
rm -f $DDMI_TUNE_FILE
timeout 2400 python bench.py --steps 2 --warmup 1 --parity-batch 128 > gpurun_out/${tag}_bench_n1_b512.json 2> gpurun_out/${tag}_bench.log
echo "bench rc=$?"; head -c 400 gpurun_out/${tag}_bench_n1_b512.json; echo
SHORT="bench.py --nfe 6 --steps 1 --warmup 0 --no-cpu-baseline --no-parity-check --no-ddim --no-workloads --no-small-batch"
(cd /tmp; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/gpurun_out/${tag}_rocprof -- python3 $OLDPWD/$SHORT > $OLDPWD/gpurun_out/${tag}_rocprof.log 2>&1)
echo "rocprof rc=$?"
f=$(find gpurun_out/${tag}_rocprof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/${tag}_kernel_stats.csv && head -8 gpurun_out/${tag}_kernel_stats.csv
(cd /tmp; timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OLDPWD/gpurun_out/${tag}_pmc_f -- python3 $OLDPWD/$SHORT > $OLDPWD/gpurun_out/${tag}_pmc_f.log 2>&1)
echo "pmc fetch rc=$?"
(cd /tmp; timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OLDPWD/gpurun_out/${tag}_pmc_w -- python3 $OLDPWD/$SHORT > $OLDPWD/gpurun_out/${tag}_pmc_w.log 2>&1)
echo "pmc write rc=$?"
python tools/pmc_summary.py gpurun_out/${tag}_pmc_f gpurun_out/${tag}_pmc_w "rocprofv3 --pmc {FETCH_SIZE|WRITE_SIZE} -- python3 $SHORT (B=512)" > gpurun_out/${tag}_pmc_traffic_b512.json
echo "pmc summary rc=$?"; head -c 500 gpurun_out/${tag}_pmc_traffic_b512.json; echo
(cd /tmp; timeout 900 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OLDPWD/gpurun_out/${tag}_pmc_c -- python3 $OLDPWD/$SHORT > $OLDPWD/gpurun_out/${tag}_pmc_c.log 2>&1)
echo "pmc clock rc=$?"
python tools/pmc_clock.py gpurun_out/${tag}_pmc_c "rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -- python3 $SHORT (B=512)" > gpurun_out/${tag}_effective_clock.json
echo "clock summary rc=$?"; head -c 600 gpurun_out/${tag}_effective_clock.json; echo
# keep the merge small: drop the raw per-dispatch CSVs
rm -rf gpurun_out/${tag}_pmc_f gpurun_out/${tag}_pmc_w gpurun_out/${tag}_pmc_c gpurun_out/${tag}_rocprof
