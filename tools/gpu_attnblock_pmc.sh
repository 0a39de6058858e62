#!/bin/bash
# The fused attention-block launch under rocprofv3: kernel durations (--kernel-trace) and L2 <-> fabric traffic (--pmc FETCH_SIZE /
# WRITE_SIZE in separate passes; FETCH_SIZE x 2 on gfx950, MI355X_MICROARCH.md) per dispatch.  usage: tools/gpu_attnblock_pmc.sh [batch]
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$PWD; B=${1:-512}
(cd /tmp; timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/ab_kt -- python3 $R/tools/bench_attnblock.py $B 3 > $R/gpurun_out/ab_kt.log 2>&1)
(cd /tmp; timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/ab_f -- python3 $R/tools/bench_attnblock.py $B 3 > $R/gpurun_out/ab_f.log 2>&1)
(cd /tmp; timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/ab_w -- python3 $R/tools/bench_attnblock.py $B 3 > $R/gpurun_out/ab_w.log 2>&1)
python3 - <<'PY'
import csv, glob, collections
def med(v):
    v = sorted(v); return v[len(v) // 2]
for f in glob.glob('gpurun_out/ab_kt/**/*kernel_trace.csv', recursive=True):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        d[r['Kernel_Name'][:60]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    for k, v in d.items():
        if any(s in k for s in ('attnblock', 'attention', 'gemm1x1', 'gn_apply')): print('kernel-trace us (median, n):', k, round(med(v), 1), len(v))
for tag, name in (('ab_f', 'FETCH_SIZE'), ('ab_w', 'WRITE_SIZE')):
    for f in glob.glob('gpurun_out/%s/**/*counter_collection.csv' % tag, recursive=True):
        d = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] == name: d[r['Kernel_Name'][:60]].append(float(r['Counter_Value']) / 1024)
        for k, v in d.items():
            if any(s in k for s in ('attnblock', 'attention', 'gemm1x1', 'gn_apply')): print(name, 'raw MB per dispatch (sorted distinct):', k, sorted(set(round(x) for x in v)))
PY
rm -rf gpurun_out/ab_kt gpurun_out/ab_f gpurun_out/ab_w
