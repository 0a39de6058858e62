#!/bin/bash
# A/B of the attention workgroup order: kernel durations (rocprofv3 --kernel-trace) and L2->fabric reads (--pmc FETCH_SIZE) per dispatch
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$PWD
(cd /tmp; timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/attn_kt -- python3 $R/tools/bench_attn_afhq.py 512 > $R/gpurun_out/attn_kt.log 2>&1)
(cd /tmp; timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/attn_pmc -- python3 $R/tools/bench_attn_afhq.py 512 > $R/gpurun_out/attn_pmc.log 2>&1)
python3 - <<'PY'
import csv, glob
for f in glob.glob('gpurun_out/attn_kt/**/*kernel_trace.csv', recursive=True):
    rows=[r for r in csv.DictReader(open(f)) if 'attention' in r['Kernel_Name']]
    rows.sort(key=lambda r:int(r['Start_Timestamp']))
    print('kernel-trace: attention dispatches (us):', [round((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3,1) for r in rows])
for f in glob.glob('gpurun_out/attn_pmc/**/*counter_collection.csv', recursive=True):
    rows=[r for r in csv.DictReader(open(f)) if 'attention' in r['Kernel_Name'] and r['Counter_Name']=='FETCH_SIZE']
    rows.sort(key=lambda r:int(r['Dispatch_Id']))
    print('FETCH_SIZE raw KiB per attention dispatch (MB):', [round(float(r['Counter_Value'])/1024,1) for r in rows])
PY
rm -rf gpurun_out/attn_kt gpurun_out/attn_pmc
