#!/usr/bin/env python3
"""ShapeNet post-sampling stages per shape: tri-plane decode at batch 1 / 8 / 32, MLP3D 128^3 grid, marching cubes (HIP events)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ddmi_amd import _lib  # noqa: E402
from ddmi_amd.config import load_ldm_args  # noqa: E402
from ddmi_amd.occupancy import Autoencoder3D, MLP3D, Generator3D, occupancy_grid_logits  # noqa: E402
from ddmi_amd import initweights  # noqa: E402
from ddmi_amd.mcubes import extract_mesh  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if __name__ == "__main__":
    _lib.init_device(0)
    a = load_ldm_args(os.path.join(ROOT, "configs", "ldm", "shapenet.yaml"))
    dev = "cuda:0"
    mlp = MLP3D(**dict(a.mlpconfig), numerics="bf16")
    mlp.load_state_dict(initweights.fill_state("shapenet.mlp", mlp.manifest()))
    mlp.finalize(0)
    z = torch.randn((32, 192, 16, 16), device=dev)
    for DB in (1, 8, 32):
        vae = Autoencoder3D(a.ddconfig, a.embed_dim, numerics="bf16", max_batch=DB)
        vae.load_state_dict(initweights.fill_state("shapenet.vae", vae.manifest()))
        vae.finalize(0)
        for _ in range(2):
            vae.decode(z[:DB].contiguous())
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for j in range(0, 32, DB):
            planes = vae.decode(z[j:j + DB].contiguous())
        torch.cuda.synchronize()
        print(f"decode batch {DB:2d}: {(time.perf_counter() - t0) / 32 * 1e3:.3f} ms per shape", flush=True)
    one = tuple([t[0:1] for t in lst] for lst in planes)
    for _ in range(2):
        grid = occupancy_grid_logits(mlp, one, 128, 0.1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(8):
        grid = occupancy_grid_logits(mlp, one, 128, 0.1)
    torch.cuda.synchronize()
    print(f"mlp3d 128^3: {(time.perf_counter() - t0) / 8 * 1e3:.3f} ms per shape")
    t0 = time.perf_counter()
    for _ in range(8):
        v, f = extract_mesh(grid[0], 0.2, 0.1)
        vc, fc = v.cpu(), f.cpu()
    torch.cuda.synchronize()
    print(f"marching cubes + host copy: {(time.perf_counter() - t0) / 8 * 1e3:.3f} ms per shape ({len(fc)} faces)")
