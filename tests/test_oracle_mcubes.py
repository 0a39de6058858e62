"""Pins the marching-cubes oracle (oracle/mcubes_oracle.c) bit-for-bit to vectors produced by the reference's own
marchingcubes.cpp (tests/golden/mcubes_cases.npz, made by oracle/gen_golden_mcubes.py) and, where the reference build
oracle/_ref/libmcubes_ref.so is present, to the reference itself on fresh random volumes."""
import numpy as np
import pytest

from oracle import mcubes_oracle as M

CASES = ["rand_f32", "rand_f32_iso", "ties_f64", "iso_rounding_f64", "single_cell", "no_cells", "no_surface", "all_inside"]


@pytest.fixture(scope="module")
def cases(golden):
    M.build()
    return golden.npz("mcubes_cases.npz")


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_reference_vectors(cases, name):
    v, f = M.marching_cubes(cases[f"{name}.volume"], float(cases[f"{name}.iso"]))
    assert v.dtype == np.float64 and f.dtype == np.uint64
    assert np.array_equal(v, cases[f"{name}.vertices"])  # bit-exact doubles
    assert np.array_equal(f, cases[f"{name}.triangles"])


def test_golden_volumes_hit_all_256_corner_masks(cases):
    seen = set()
    for name in CASES:
        vol, iso = cases[f"{name}.volume"], np.float32(cases[f"{name}.iso"])
        if min(vol.shape) < 2:
            continue
        c = (vol <= iso).astype(np.int64)
        cube = (c[:-1, :-1, :-1] | c[1:, :-1, :-1] << 1 | c[1:, 1:, :-1] << 2 | c[:-1, 1:, :-1] << 3 | c[:-1, :-1, 1:] << 4
                | c[1:, :-1, 1:] << 5 | c[1:, 1:, 1:] << 6 | c[:-1, 1:, 1:] << 7)
        seen |= set(np.unique(cube).tolist())
    assert len(seen) == 256  # so the packed case table is checked entry by entry through the outputs


@pytest.mark.parametrize("tag", ["mesh_t02", "mesh_t05"])
def test_extract_mesh_vectors(cases, tag):
    v, f = M.extract_mesh(cases[f"{tag}.occ"], float(cases[f"{tag}.threshold"]), float(cases[f"{tag}.padding"]))
    assert np.array_equal(v, cases[f"{tag}.vertices"]) and np.array_equal(f, cases[f"{tag}.triangles"])
    # the -1e6 border closes the surface (generation.py:159 "Make sure that mesh is watertight"): every edge twice
    e = np.sort(np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]]).astype(np.int64), axis=1)
    _, cnt = np.unique(e, axis=0, return_counts=True)
    assert (cnt == 2).all()


def test_isovalue_goes_through_c_float(cases):
    # mcubes.pyx:22 declares `float isovalue`: a value between a double threshold and its fp32 rounding flips sides
    thr = float(np.log(0.2) - np.log(0.8))
    lo, hi = sorted((thr, float(np.float32(thr))))
    assert lo != hi
    vol = np.full((3, 3, 3), 5.0)
    vol[1, 1, 1] = (lo + hi) / 2
    v, f = M.marching_cubes(vol, thr)
    inside = vol[1, 1, 1] <= float(np.float32(thr))
    assert (len(f) > 0) == inside


@pytest.mark.skipif(not M.have_ref(), reason="oracle/_ref/libmcubes_ref.so not built (needs /root/reference)")
@pytest.mark.parametrize("shape,iso,dtype", [((33, 31, 35), 0.1, np.float32), ((20, 40, 10), -0.7, np.float64),
                                              ((64, 64, 64), 0.0, np.float32)])
def test_oracle_equals_reference_build(shape, iso, dtype):
    M.build()
    rng = np.random.default_rng(hash(shape) % 1000)
    vol = rng.standard_normal(shape).astype(dtype)
    a, b = M.marching_cubes(vol, iso), M.marching_cubes_ref(vol, iso)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
