"""Generates tests/golden/mcubes_cases.npz from the REFERENCE's own marching cubes (oracle/_ref/libmcubes_ref.so, built
from /root/reference/convocc/src/utils/libmcubes/marchingcubes.{h,cpp} by oracle/Makefile) - run in the build container:

    make -C oracle && python oracle/gen_golden_mcubes.py

Every case stores the input volume, the isovalue as the caller passes it (a Python float; libmcubes rounds it to C float)
and the reference's (vertices float64, triangles uint64).  `mesh_*` cases store extract_mesh's inputs/outputs
(generation.py:146-183 restated in oracle/mcubes_oracle.py, run over the reference's marching cubes).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import mcubes_oracle as M  # noqa: E402


def volumes():
    rng = np.random.default_rng(20240607)
    yield "rand_f32", rng.standard_normal((17, 18, 19)).astype(np.float32), 0.0
    yield "rand_f32_iso", rng.standard_normal((9, 7, 8)).astype(np.float32), 0.3
    # integer-valued doubles: ties f1 == f2 are impossible on a cut edge, but v == iso and the fp32 rounding of iso matter
    yield "ties_f64", np.round(2 * rng.standard_normal((12, 12, 12))), -1.0
    yield "iso_rounding_f64", rng.standard_normal((8, 9, 10)), float(np.log(0.2) - np.log(0.8))
    yield "single_cell", rng.standard_normal((2, 2, 2)).astype(np.float32), 0.0
    yield "no_cells", rng.standard_normal((1, 5, 5)).astype(np.float32), 0.0
    yield "no_surface", np.full((6, 6, 6), 3.0, np.float32), 0.0
    yield "all_inside", np.full((5, 6, 7), -3.0, np.float32), 0.0


def main():
    assert M.have_ref(), "build oracle/_ref first (make -C oracle, needs /root/reference)"
    out = {}
    seen = set()
    for name, vol, iso in volumes():
        v, f = M.marching_cubes_ref(vol, iso)
        out[f"{name}.volume"], out[f"{name}.iso"] = vol, np.float64(iso)
        out[f"{name}.vertices"], out[f"{name}.triangles"] = v, f
        if vol.shape[0] > 1:
            c = (vol <= np.float32(iso)).astype(np.int64)
            cube = (c[:-1, :-1, :-1] | c[1:, :-1, :-1] << 1 | c[1:, 1:, :-1] << 2 | c[:-1, 1:, :-1] << 3 | c[:-1, :-1, 1:] << 4
                    | c[1:, :-1, 1:] << 5 | c[1:, 1:, 1:] << 6 | c[:-1, 1:, 1:] << 7)
            seen |= set(np.unique(cube).tolist())
        print(name, vol.shape, vol.dtype, v.shape, f.shape)
    assert len(seen) == 256, f"only {len(seen)} of the 256 corner masks are exercised"
    rng = np.random.default_rng(7)
    g = np.mgrid[:24, :20, :22].astype(np.float32)
    r = np.sqrt((g[0] - 11.3) ** 2 + (g[1] - 9.6) ** 2 + (g[2] - 10.2) ** 2)
    occ = (7.5 - r + 0.3 * rng.standard_normal(r.shape)).astype(np.float32)
    for tag, thr, pad in (("mesh_t02", 0.2, 0.1), ("mesh_t05", 0.5, 0.0)):
        v, f = M.extract_mesh(occ, thr, pad, mc=M.marching_cubes_ref)
        out[f"{tag}.occ"], out[f"{tag}.threshold"], out[f"{tag}.padding"] = occ, np.float64(thr), np.float64(pad)
        out[f"{tag}.vertices"], out[f"{tag}.triangles"] = v, f
        print(tag, v.shape, f.shape)
    path = os.path.join(ROOT, "tests", "golden", "mcubes_cases.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
