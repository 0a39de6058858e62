#!/bin/bash
# Round 5: prices the producer-side GroupNorm with TIMING-ONLY builds (wrong results on purpose) before it is built.
#   A  baseline library
#   B  out_layers GroupNorm passes skipped, NO extra epilogue work      (what removing the passes is worth at most)
#   C  passes skipped + scale/shift/SiLU VALU work in the producers' epilogues (variant library, -DGN_ABL, DDMI_CONV_DBG=32)
tag=${1:-r5a}
mkdir -p gpurun_out
export TMPDIR=/tmp
Q="--nfe 10 --steps 2 --warmup 1 --no-ddim --no-parity-check --no-cpu-baseline --no-workloads"
V=$PWD/ddmi_amd/lib/variants/libddmi_hip_gnabl.so
timeout 600 python bench.py $Q > gpurun_out/${tag}_A.json 2> gpurun_out/${tag}_A.log
DDMI_GN_ABL=1 timeout 600 python bench.py $Q > gpurun_out/${tag}_B.json 2> gpurun_out/${tag}_B.log
DDMI_LIB_PATH=$V DDMI_GN_ABL=1 DDMI_CONV_DBG=32 timeout 600 python bench.py $Q > gpurun_out/${tag}_C.json 2> gpurun_out/${tag}_C.log
timeout 600 python bench.py $Q > gpurun_out/${tag}_A2.json 2> gpurun_out/${tag}_A2.log
python - <<PY
import json
for k in "A B C A2".split():
    try:
        d = json.loads(open(f"gpurun_out/${tag}_{k}.json").read().strip().splitlines()[-1])
        sh = d.get("kernel_time_share", {})
        print(k, "samples/s", round(d["value"], 3), "loop ms", d.get("last_step_phase_ms", {}).get("sampling_loop"), "halo us", d["roofline"]["avg_launch_us"],
              "gn_apply share", sh.get("gn_apply"))
    except Exception as e:
        print(k, "failed", e)
PY
