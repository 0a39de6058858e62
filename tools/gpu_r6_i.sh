#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6i; mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_round6.py -x -q -m gpu -s -k "main_video" > $O/tests.log 2>&1
echo "video e2e rc=$?"; grep -v amdgpu.ids $O/tests.log | tail -25
for B in 8 32 64; do python3 - <<PY 2>&1 | grep -v amdgpu.ids
import sys, time, torch
sys.path.insert(0, "tests")
from ddmi_amd import train as T, initweights as iw
from helpers import SKY_UNET
B = $B
tr = T.LDMTrainStep(SKY_UNET, dict(timesteps=1000, channels=64, domain="video"), numerics="bf16", device="cuda:0")
st = {"model." + k: v for k, v in iw.fill_state("sky.unet3", [(k, tuple(p.shape)) for k, p in tr.model.state_dict().items()]).items()}
tr.load_state_dict(st, strict=False).finalize()
z = torch.randn((B, 64, 2048), device="cuda:0")
for _ in range(2): tr.train_step(z)
torch.cuda.synchronize(); t0 = time.time()
for _ in range(3): tr.train_step(z)
torch.cuda.synchronize(); dt = (time.time() - t0) / 3
print(f"sky train step bf16 B={B}: {dt*1e3:.1f} ms = {B/dt:.1f} clips/s; params {tr.flat.n/1e6:.0f} M; mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB")
PY
done
