"""Builds ddmi_amd/lib/libddmi_hip.so from ddmi_amd/csrc with hipcc for gfx950 (cross-compiles without a GPU).

    python -m ddmi_amd.build [--force]
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(LIBDIR, "obj")
LIB = os.path.join(LIBDIR, "libddmi_hip.so")
SOURCES = ["k_elementwise.hip", "k_groupnorm.hip", "k_conv.hip", "k_attention.hip", "k_attnblock.hip", "k_mlp2d.hip", "k_mlp3d.hip", "mlp3d_host.hip", "k_mlpv.hip", "mlpv_host.hip", "k_mcubes.hip", "models_video.hip", "net.hip",
           "models.hip", "mlp2d_host.hip", "capi.hip", "prof.hip", "k_train.hip", "k_attn_bwd.hip", "train_capi.hip", "k_microbench.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=on", "-Wall", "-Wno-unused-function",
         "-Wno-unused-variable", "-Wno-unused-but-set-variable"]


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC)")


def _newest_header() -> float:
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    deps.append(os.path.join(os.path.dirname(HERE), "include", "ddmi_hip.h"))
    return max(os.path.getmtime(p) for p in deps)


# per-file extras.  k_attention: keep MFMA results in VGPRs - the softmax between the two matrix products is VALU work on
# the score accumulators, and with AGPR accumulators the compiler brackets it with ~80 v_accvgpr moves per 64-key tile
# -fno-slp-vectorize: packed f32 VALU (v_pk_mul_f32 / v_pk_add_f32, which the SLP vectoriser forms from adjacent scalar multiplies)
# is an anti-lever beside MFMAs (MI355X_MICROARCH.md, per-instruction cycle constants): measured on one box, attention backward
# dQ 289 -> 266 us, dK/dV 380 -> 360, the eager-maximum forward 270 -> 261; the lazy forward has none in its hot path (unchanged)
_ATTN_FLAGS = ["-mllvm", "-amdgpu-mfma-vgpr-form", "-fno-slp-vectorize"]
EXTRA_FLAGS = {"k_attention.hip": _ATTN_FLAGS, "k_attn_bwd.hip": _ATTN_FLAGS, "k_attnblock.hip": _ATTN_FLAGS}


def _compile(src: str, force: bool, hdr_mtime: float) -> str:
    obj = os.path.join(OBJDIR, src.replace(".hip", ".o"))
    path = os.path.join(CSRC, src)
    if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(path), hdr_mtime):
        return obj
    cmd = [hipcc(), *FLAGS, *EXTRA_FLAGS.get(src, []), "-c", path, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj


def build_library(force: bool = False, jobs: int = 4) -> str:
    os.makedirs(OBJDIR, exist_ok=True)
    hdr = _newest_header()
    with ThreadPoolExecutor(max_workers=jobs) as ex:
        objs = list(ex.map(lambda s: _compile(s, force, hdr), SOURCES))
    if force or not os.path.exists(LIB) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs):
        cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv))
