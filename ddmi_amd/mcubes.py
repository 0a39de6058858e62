"""Post-path consumers on the device.

  marching_cubes(volume, isovalue)   drop-in for `libmcubes.marching_cubes` (convocc/src/utils/libmcubes/mcubes.pyx:22-27):
                                     same vertices (float64, bit-identical), same triangles, same ORDER as the reference's
                                     sequential sweep - computed by the HIP kernels of csrc/k_mcubes.hip
  extract_mesh(occ_hat, ...)         Generator3D.extract_mesh up to the trimesh construction (generation.py:146-183,
                                     vol_bound None): threshold in logit space, virtual -1e6 border, the 0.5 / padding
                                     shifts and the box scaling fused into the vertex store
  images_to_uint8(img)               the sample write-out of evals/eval.py:226-233: clamp -> [0,1] -> uint8 HWC

There is no CPU fallback: without the HIP library / a GPU these raise.
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np

from . import _lib


def _grid_of(vol, isovalue, pad, pad_value):
    import torch
    assert vol.is_cuda, "marching cubes runs on the device: pass a cuda tensor (or a numpy array, which is uploaded)"
    if vol.dim() != 3:
        raise RuntimeError("Only three-dimensional arrays are supported.")  # pywrapper.cpp:91-92
    if vol.dtype != torch.float64:
        vol = vol.to(torch.float32)
    vol = vol.contiguous()
    iso = float(np.float32(isovalue))  # mcubes.pyx:22: `float isovalue` is a C float
    g = _lib.McubesGrid(vol.data_ptr(), int(vol.dtype == torch.float64), *vol.shape, int(pad), float(pad_value), iso)
    return vol, g


def _run(vol, isovalue, pad, pad_value, normalize, box_size):
    import torch
    L = _lib.lib()
    vol, g = _grid_of(vol, isovalue, pad, pad_value)
    with torch.cuda.device(vol.device):
        nbytes = C.c_size_t()
        _lib.check(L.ddmi_mcubes_workspace_bytes(C.byref(g), C.byref(nbytes)))
        ws = torch.empty(max(nbytes.value, 256), dtype=torch.uint8, device=vol.device)
        nv, nt = C.c_int64(), C.c_int64()
        st = _lib.stream_ptr()
        _lib.check(L.ddmi_mcubes_count(C.byref(g), ws.data_ptr(), C.byref(nv), C.byref(nt), st))
        verts = torch.empty((nv.value, 3), dtype=torch.float64, device=vol.device)
        faces = torch.empty((nt.value, 3), dtype=torch.int64, device=vol.device)
        if nv.value or nt.value:
            _lib.check(L.ddmi_mcubes_emit(C.byref(g), ws.data_ptr(), verts.data_ptr(), faces.data_ptr(), int(normalize),
                                          float(box_size), st))
    return verts, faces


def marching_cubes(volume, isovalue, *, pad: int = 0, pad_value: float = -1e6):
    """(vertices [nv,3] float64, triangles [nf,3]) of `libmcubes.marching_cubes(volume, isovalue)`.

    A numpy volume is uploaded and numpy arrays come back (triangles as uint64, the reference's `unsigned long`);
    a cuda tensor stays on the device (triangles int64).  pad = 1 computes on np.pad(volume, 1, constant_values=pad_value)
    without building it."""
    import torch
    if isinstance(volume, np.ndarray):
        assert torch.cuda.is_available(), "ddmi_amd.mcubes needs the GPU (no CPU fallback)"
        if volume.ndim != 3:
            raise RuntimeError("Only three-dimensional arrays are supported.")
        v, f = _run(torch.from_numpy(np.ascontiguousarray(volume)).cuda(), isovalue, pad, pad_value, 0, 1.0)
        return v.cpu().numpy(), f.cpu().numpy().astype(np.uint64)
    return _run(volume, isovalue, pad, pad_value, 0, 1.0)


def extract_mesh(occ_hat, threshold: float = 0.5, padding: float = 0.1):
    """generation.py:146-183 (vol_bound None) on a device-resident logit grid [n_x, n_y, n_z]:
    returns (vertices float64 [nv,3] in the (1+padding) box, triangles int64 [nf,3]) on the device - what the reference
    hands to trimesh.Trimesh / pytorch3d Meshes (after `.float()`)."""
    thr = np.log(threshold) - np.log(1. - threshold)
    box_size = 1 + padding
    return _run(occ_hat, thr, 1, -1e6, 1, box_size)


def images_to_uint8(img):
    """evals/eval.py:226-233: fake.clamp(-1,1) -> (fake+1)/2 -> (x*255).astype(uint8), [B,C,H,W] fp32 -> [B,H,W,C] uint8."""
    import torch
    assert img.is_cuda and img.dim() == 4
    img = img.to(torch.float32).contiguous()
    B, Cc, H, W = img.shape
    out = torch.empty((B, H, W, Cc), dtype=torch.uint8, device=img.device)
    with torch.cuda.device(img.device):
        _lib.check(_lib.lib().ddmi_image_to_u8(img.data_ptr(), out.data_ptr(), B, Cc, H, W, _lib.stream_ptr()))
    return out
