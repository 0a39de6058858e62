"""Post-path consumers on the GPU through the C ABI: marching cubes bit-identical to the
reference's sequential libmcubes (vertices, triangles AND their order), extract_mesh, and the uint8 image write-out."""
import numpy as np
import pytest
import torch

from ddmi_amd import mcubes
from oracle import mcubes_oracle as M

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
CASES = ["rand_f32", "rand_f32_iso", "ties_f64", "iso_rounding_f64", "single_cell", "no_cells", "no_surface", "all_inside"]


def same(a, b):
    return a.shape == b.shape and np.array_equal(a, b)


@pytest.mark.parametrize("name", CASES)
def test_reference_vectors(golden, name):
    c = golden.npz("mcubes_cases.npz")
    v, f = mcubes.marching_cubes(c[f"{name}.volume"], float(c[f"{name}.iso"]))  # numpy in -> numpy out, like libmcubes
    assert v.dtype == np.float64 and f.dtype == np.uint64
    assert same(v, c[f"{name}.vertices"]) and same(f, c[f"{name}.triangles"])


@pytest.mark.parametrize("tag", ["mesh_t02", "mesh_t05"])
def test_extract_mesh_vectors(golden, tag):
    c = golden.npz("mcubes_cases.npz")
    occ = torch.from_numpy(c[f"{tag}.occ"]).to(DEV)
    v, f = mcubes.extract_mesh(occ, float(c[f"{tag}.threshold"]), float(c[f"{tag}.padding"]))
    assert v.is_cuda and v.dtype == torch.float64 and f.dtype == torch.int64
    assert same(v.cpu().numpy(), c[f"{tag}.vertices"]) and same(f.cpu().numpy().astype(np.uint64), c[f"{tag}.triangles"])


@pytest.mark.parametrize("shape,iso,dtype", [((33, 31, 35), 0.1, np.float32), ((20, 40, 10), -0.7, np.float64),
                                              ((2, 70, 3), 0.0, np.float32), ((65, 64, 63), 0.25, np.float32),
                                              ((130, 130, 130), 1.5, np.float32)])
def test_random_volumes_against_oracle(shape, iso, dtype):
    rng = np.random.default_rng(sum(shape))
    vol = rng.standard_normal(shape).astype(dtype)
    v, f = mcubes.marching_cubes(vol, iso)
    rv, rf = M.marching_cubes(vol, iso)
    assert same(v, rv) and same(f, rf)


def test_virtual_border_equals_np_pad():
    rng = np.random.default_rng(5)
    vol = rng.standard_normal((30, 28, 26)).astype(np.float32)
    dv = torch.from_numpy(vol).to(DEV)
    v1, f1 = mcubes.marching_cubes(dv, -0.2, pad=1, pad_value=-1e6)
    rv, rf = M.marching_cubes(np.pad(vol, 1, "constant", constant_values=-1e6), -0.2)
    assert same(v1.cpu().numpy(), rv) and same(f1.cpu().numpy().astype(np.uint64), rf)


def _smooth_field(n, seed=0):
    g = torch.linspace(-1, 1, n, device=DEV)
    x, y, z = torch.meshgrid(g, g, g, indexing="ij")
    r = torch.sqrt((x - 0.05) ** 2 + (y + 0.02) ** 2 + (z - 0.03) ** 2)
    return (4.0 * (0.62 - r) + 0.3 * torch.sin(7 * x) * torch.cos(5 * y + z)).float()


def test_extract_mesh_128_against_oracle():
    occ = _smooth_field(128)
    v, f = mcubes.extract_mesh(occ, 0.2, 0.1)
    rv, rf = M.extract_mesh(occ.cpu().numpy(), 0.2, 0.1)
    assert same(v.cpu().numpy(), rv) and same(f.cpu().numpy().astype(np.uint64), rf)


def test_extract_mesh_256_is_a_closed_surface():
    # BASELINE's largest grid (256^3 + border = 17 M cells): size-independent properties instead of the oracle
    occ = _smooth_field(256)
    v, f = mcubes.extract_mesh(occ, 0.2, 0.1)
    nv = v.shape[0]
    assert nv > 100000 and int(f.min()) == 0 and int(f.max()) == nv - 1
    assert torch.unique(f).numel() == nv                      # every vertex is used
    assert float(v.abs().max()) <= 0.55 * (1 + 2.0 / 255)    # inside the (1 + padding) box (plus the border cell)
    e = torch.cat([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]])
    key = torch.minimum(e[:, 0], e[:, 1]) * nv + torch.maximum(e[:, 0], e[:, 1])
    _, cnt = torch.unique(key, return_counts=True)
    assert bool((cnt == 2).all())                             # watertight: every edge in exactly two triangles
    assert nv - cnt.numel() + f.shape[0] == 2                 # Euler characteristic of a sphere-like surface
    # a second run gives the same bytes (no atomics, no ordering dependence)
    v2, f2 = mcubes.extract_mesh(occ, 0.2, 0.1)
    assert torch.equal(v, v2) and torch.equal(f, f2)


def test_images_to_uint8_is_bit_exact():
    rng = np.random.default_rng(3)
    img = (1.3 * rng.standard_normal((3, 3, 32, 48))).astype(np.float32)
    img.reshape(-1)[:8] = [-1.0, 1.0, 0.0, -0.0, 1.0 - 2 ** -24, -1.0 + 2 ** -24, 0.00392157, 7.5]
    # every fp32 x whose (x+1)/2*255 lands next to an integer boundary
    k = np.arange(256, dtype=np.float64)
    edge = (2 * k / 255 - 1).astype(np.float32)
    img.reshape(-1)[8:8 + 256] = edge
    img.reshape(-1)[264:264 + 256] = np.nextafter(edge, np.float32(-2))
    img.reshape(-1)[520:520 + 256] = np.nextafter(edge, np.float32(2))
    out = mcubes.images_to_uint8(torch.from_numpy(img).to(DEV)).cpu().numpy()
    fake = torch.from_numpy(img).clamp(-1., 1.)          # evals/eval.py:226-232 restated
    fake2 = (fake + 1) / 2
    want = np.stack([(fake2[b].numpy().transpose(1, 2, 0) * 255).astype(np.uint8) for b in range(img.shape[0])])
    assert out.dtype == np.uint8 and out.shape == (3, 32, 48, 3)
    assert np.array_equal(out, want)


def test_generator3d_end_to_end(golden, tmp_path):
    """Generator3D.generate_mesh_fromdiffusion: latent -> tri-planes -> dense logits -> mesh, all on the device.  The mesh
    must be exactly what the reference's extract_mesh (oracle, pinned to marchingcubes.cpp) makes of the SAME logit grid."""
    from ddmi_amd.occupancy import MLP3D, Autoencoder3D, Generator3D, occupancy_grid_logits
    from helpers import named_state
    cfgs = golden.json("configs.json")
    vae = Autoencoder3D(cfgs["TOY_DD3"], cfgs["TOY_EMBED3"], numerics="fp32", max_batch=1)
    vae.load_state_dict(named_state("O1.vae3d", vae.manifest()))
    mlp = MLP3D(numerics="fp32")
    mlp.load_state_dict(named_state("O1.mlp3d", mlp.manifest()))
    z = torch.from_numpy(golden.npz("O1_occupancy.npz")["z"][:1]).to(DEV)
    grid = occupancy_grid_logits(mlp, vae.decode(z), 48, 0.1)[0]
    thr = float(torch.sigmoid(grid.median()))          # random weights: put the surface where the logits are
    gen = Generator3D(threshold=thr, resolution0=48, upsampling_steps=0, padding=0.1)
    mesh, mesh2 = gen.generate_mesh_fromdiffusion(z, vae, mlp, DEV)
    rv, rf = M.extract_mesh(grid.cpu().numpy(), thr, 0.1)
    assert len(rf) > 1000
    assert same(mesh.vertices, rv) and same(mesh.faces.astype(np.uint64), rf)
    assert mesh2["verts"].shape == (1, len(rv), 3) and mesh2["verts"].dtype == torch.float32
    assert mesh2["faces"].shape == (1, len(rf), 3) and mesh2["faces"].dtype == torch.int64
    path = mesh.export(str(tmp_path / "generation.obj"))
    lines = open(path).read().splitlines()
    assert len(lines) == len(rv) + len(rf) and lines[0].startswith("v ") and lines[-1].startswith("f ")
    assert int(lines[-1].split()[1]) == int(rf[-1, 0]) + 1   # .obj indices are 1-based
    with pytest.raises(NotImplementedError):
        Generator3D(upsampling_steps=2)


def test_occupancy_pipeline_is_batch_invariant(golden):
    """OccupancyPipeline (tools/ldm/occupancy.py:205-230 / exp/stage.py:207-241): latent chains -> meshes; a sample's mesh
    depends only on (base_seed, global index), not on what else is in the batch."""
    from ddmi_amd.pipeline import OccupancyPipeline
    cfgs = golden.json("configs.json")
    ucfg = dict(size1=4, size2=4, size3=4, image_size=4, in_channels=48, out_channels=48, model_channels=32,
                channel_mult=[1, 2], num_res_blocks=1, attention_resolutions=[2], num_head_channels=32)
    dcfg = dict(timesteps=1000, image_size=4, channels=48, clip_denoised=False, mixed_prediction=True, mixed_init=-6,
                sampling_timesteps=4, ddim_sampling_eta=0.0, domain="occupancy")
    pipe = OccupancyPipeline(ucfg, cfgs["TOY_DD3"], cfgs["MLP3D_CFG"], dcfg, cfgs["TOY_EMBED3"], threshold=0.5,
                             resolution0=32, upsampling_steps=0, padding=0.1, numerics="fp32", max_batch=3, device=DEV)
    pipe.load_named_seed("toyocc").finalize()
    meshes, z = pipe.generate([5, 6, 7], base_seed=11, return_latent=True)
    assert z.shape == (3, 48, 4, 4) and len(meshes) == 3
    assert all(m.vertices.dtype == np.float64 and m.faces.shape[1] == 3 for m in meshes)
    assert sum(len(m.faces) for m in meshes) > 0
    alone = pipe.generate([7], base_seed=11)[0]
    assert same(alone.vertices, meshes[2].vertices) and same(alone.faces, meshes[2].faces)
    other = pipe.generate([8], base_seed=11)[0]
    assert not (same(other.vertices, meshes[2].vertices) and same(other.faces, meshes[2].faces))
    # the tri-plane decoder on all three latents at once (decode_batch; the bf16 default is 32), meshes still one latent at a time:
    # the same meshes as with the reference's decode batch of 1 (the parity mode's default)
    assert pipe.decode_batch == 1
    pipe3 = OccupancyPipeline(ucfg, cfgs["TOY_DD3"], cfgs["MLP3D_CFG"], dcfg, cfgs["TOY_EMBED3"], threshold=0.5, resolution0=32,
                              upsampling_steps=0, padding=0.1, numerics="fp32", max_batch=3, device=DEV, decode_batch=3)
    pipe3.load_named_seed("toyocc").finalize()
    batched = pipe3.generate([5, 6, 7], base_seed=11)
    for m, ref in zip(batched, meshes):
        assert same(m.vertices, ref.vertices) and same(m.faces, ref.faces)
    # `sink`: every mesh handed over as it is finished, in index order, nothing kept (bench.py runs thousands of chains per call this way)
    got = []
    assert pipe3.generate([5, 6, 7], base_seed=11, sink=got.append) == [] and len(got) == 3
    for m, ref in zip(got, meshes):
        assert same(m.vertices, ref.vertices) and same(m.faces, ref.faces)


def test_occupancy_pipeline_bf16_batched_decode_and_side_stream(golden):
    """The bf16 pipeline's defaults (decoder on up to 32 latents per call, marching cubes on a side stream under the next shape's
    MLP3D launch) against the same pipeline with the reference's decode batch of 1 on one stream: the same latents, and meshes that
    differ only by bf16 rounding of the decoded planes (a few surface cells)."""
    from ddmi_amd.pipeline import OccupancyPipeline
    cfgs = golden.json("configs.json")
    ucfg = dict(size1=4, size2=4, size3=4, image_size=4, in_channels=48, out_channels=48, model_channels=32,
                channel_mult=[1, 2], num_res_blocks=1, attention_resolutions=[2], num_head_channels=32)
    dcfg = dict(timesteps=1000, image_size=4, channels=48, clip_denoised=False, mixed_prediction=True, mixed_init=-6,
                sampling_timesteps=4, ddim_sampling_eta=0.0, domain="occupancy")

    def make(db):
        p = OccupancyPipeline(ucfg, cfgs["TOY_DD3"], cfgs["MLP3D_CFG"], dcfg, cfgs["TOY_EMBED3"], threshold=0.5, resolution0=32,
                              upsampling_steps=0, padding=0.1, numerics="bf16", max_batch=4, device=DEV, decode_batch=db)
        return p.load_named_seed("toyocc").finalize()

    batched, one = make(None), make(1)
    assert batched.decode_batch == 4 and one.decode_batch == 1      # default: 32, clipped to max_batch
    mb, zb = batched.generate([1, 2, 3, 4], base_seed=5, return_latent=True)
    mo, zo = one.generate([1, 2, 3, 4], base_seed=5, return_latent=True)
    assert torch.equal(zb, zo) and len(mb) == len(mo) == 4
    for a, b in zip(mb, mo):
        fa, fb = len(a.faces), len(b.faces)
        assert abs(fa - fb) <= 0.05 * max(fa, fb, 20), (fa, fb)
        if fa and fb:
            assert np.allclose(a.vertices.min(0), b.vertices.min(0), atol=0.1) and np.allclose(a.vertices.max(0), b.vertices.max(0), atol=0.1)
