"""TEST INFRASTRUCTURE ONLY: ctypes front ends of the marching-cubes oracle.

  marching_cubes(volume, isovalue)      -> oracle/_build/libmcubes_oracle.so (oracle/mcubes_oracle.c, the C restatement)
  marching_cubes_ref(volume, isovalue)  -> oracle/_ref/libmcubes_ref.so (the reference's own marchingcubes.cpp, compiled
                                           by oracle/Makefile; present wherever `make -C oracle` ran with /root/reference)
Both follow `libmcubes.marching_cubes` (mcubes.pyx:22-27): the isovalue is rounded to C float first, the results are
(vertices float64 [nv,3], triangles uint64 [nf,3]).  `extract_mesh` restates generation.py:146-183 (vol_bound None).
"""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_ORACLE = os.path.join(HERE, "_build", "libmcubes_oracle.so")
_REF = os.path.join(HERE, "_ref", "libmcubes_ref.so")


def build():
    """`make -C oracle`: the C restatement always, the reference build when its sources are present."""
    subprocess.run(["make", "-C", HERE], check=True, capture_output=True)


def have_ref() -> bool:
    return os.path.exists(_REF)


def _run(path, prefix, idx_t, volume, isovalue):
    if not os.path.exists(path):
        if path == _ORACLE:
            build()
        else:
            raise FileNotFoundError(path)
    lib = ctypes.CDLL(path)
    run = getattr(lib, prefix + "_run")
    free = getattr(lib, prefix + "_free")
    free.argtypes = [ctypes.c_void_p]
    run.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_long, ctypes.c_long, ctypes.c_double,
                    ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_void_p),
                    ctypes.POINTER(ctypes.c_size_t)]
    vol = np.asarray(volume)
    assert vol.ndim == 3, "Only three-dimensional arrays are supported."  # pywrapper.cpp:91-92
    if vol.dtype != np.float64:
        vol = vol.astype(np.float32, copy=False)
    vol = np.ascontiguousarray(vol)
    iso = float(np.float32(isovalue))  # mcubes.pyx:22 `float isovalue`
    pv, pi = ctypes.c_void_p(), ctypes.c_void_p()
    nv, ni = ctypes.c_size_t(), ctypes.c_size_t()
    rc = run(vol.ctypes.data, int(vol.dtype == np.float64), *vol.shape, iso, ctypes.byref(pv), ctypes.byref(nv),
             ctypes.byref(pi), ctypes.byref(ni))
    assert rc == 0
    verts = np.ctypeslib.as_array(ctypes.cast(pv, ctypes.POINTER(ctypes.c_double)), (max(nv.value, 1),))[:nv.value].copy()
    faces = np.ctypeslib.as_array(ctypes.cast(pi, ctypes.POINTER(idx_t)), (max(ni.value, 1),))[:ni.value].copy()
    free(pv)
    free(pi)
    return verts.reshape(-1, 3), faces.astype(np.uint64).reshape(-1, 3)


def marching_cubes(volume, isovalue):
    return _run(_ORACLE, "mcubes_oracle", ctypes.c_int64, volume, isovalue)


def marching_cubes_ref(volume, isovalue):
    return _run(_REF, "mcubes_ref", ctypes.c_size_t, volume, isovalue)


def extract_mesh(occ_hat, threshold=0.5, padding=0.1, mc=marching_cubes):
    """generation.py:146-183 with vol_bound None: logit threshold, -1e6 border, the 0.5 and padding shifts, box scaling."""
    occ_hat = np.asarray(occ_hat)
    n_x, n_y, n_z = occ_hat.shape
    box_size = 1 + padding
    thr = np.log(threshold) - np.log(1. - threshold)
    padded = np.pad(occ_hat, 1, "constant", constant_values=-1e6)
    vertices, triangles = mc(padded, thr)
    vertices -= 0.5
    vertices -= 1
    vertices /= np.array([n_x - 1, n_y - 1, n_z - 1])
    vertices = box_size * (vertices - 0.5)
    return vertices, triangles
