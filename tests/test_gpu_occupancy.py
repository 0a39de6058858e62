"""Occupancy path (BASELINE config 4) on the GPU against golden vectors captured from the
reference: tri-plane decoder, fused tri-plane sampling + MLP3D, dense query grid; and the ShapeNet-size UNet."""
import numpy as np
import pytest
import torch

from ddmi_amd import initweights as iw
from ddmi_amd.occupancy import MLP3D, Autoencoder3D, decoder3d_manifest, mlp3d_manifest, occupancy_grid_logits
from ddmi_amd.unet import UNetModel
from helpers import TOL_BF16, TOL_FP32, named_state, rel_err
from oracle import ddmi_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
SHAPENET_UNET = dict(image_size=16, in_channels=192, model_channels=256, out_channels=192, attention_resolutions=[8, 4, 2],
                     num_res_blocks=2, channel_mult=[1, 2, 4, 6], num_head_channels=32)
SHAPENET_DD = dict(double_z=True, z_channels=128, resolution=64, in_channels=32, out_ch=64, ch=64, ch_mult=[1, 2, 4],
                   num_res_blocks=2, attn_resolutions=[], hdbf_resolutions=[32, 16], inter_attn_resolutions=[64, 32, 16],
                   dropout=0.0, attn_type="vanilla")


@pytest.mark.parametrize("numerics,tol", [("fp32", TOL_FP32), ("bf16", TOL_BF16)])
def test_triplane_decoder_golden(golden, numerics, tol):
    o1 = golden.npz("O1_occupancy.npz")
    cfgs = golden.json("configs.json")
    vae = Autoencoder3D(cfgs["TOY_DD3"], cfgs["TOY_EMBED3"], numerics=numerics, max_batch=2)
    vae.load_state_dict(named_state("O1.vae3d", vae.manifest()))
    z = torch.from_numpy(o1["z"]).to(DEV)
    for ref_layout in (False, True):
        xy, yz, xz = vae.decode(z, reference_layout=ref_layout)
        for name, lst in (("xy", xy), ("yz", yz), ("xz", xz)):
            assert [tuple(p.shape) for p in lst] == [(2, 64, 4, 4), (2, 64, 8, 8), (2, 64, 16, 16)]
            for i, pl in enumerate(lst):
                assert rel_err(pl.float(), o1[f"{name}{i}"]) < tol, (name, i, ref_layout)


@pytest.mark.parametrize("numerics,tol", [("fp32", TOL_FP32), ("bf16", TOL_BF16)])
def test_mlp3d_golden(golden, numerics, tol):
    o1 = golden.npz("O1_occupancy.npz")
    mlp = MLP3D(numerics=numerics)
    mlp.load_state_dict(named_state("O1.mlp3d", mlp.manifest()))
    planes = tuple([torch.from_numpy(o1[f"{n}{i}"]).to(DEV) for i in range(3)] for n in ("xy", "yz", "xz"))
    logits = mlp(torch.from_numpy(o1["points"]), planes).logits          # 4096 points incl. out-of-range ones
    assert logits.shape == (2, 4096)
    assert rel_err(logits, o1["logits"]) < tol
    native = torch.float32 if numerics == "fp32" else torch.bfloat16
    planes_cl = tuple([p.to(native).contiguous(memory_format=torch.channels_last) for p in lst] for lst in planes)
    assert rel_err(mlp(torch.from_numpy(o1["points"]), planes_cl).logits, o1["logits"]) < tol
    one = tuple([p[:1] for p in lst] for lst in planes)
    grid = occupancy_grid_logits(mlp, one, 6)                            # generation.py:87-97 at nx = 6
    assert grid.shape == (1, 6, 6, 6)
    assert rel_err(grid.reshape(-1), o1["grid6_logits"]) < tol
    # ragged tail: a point count that is not a multiple of the tile
    sub = mlp(torch.from_numpy(o1["points"][:, :1000]), planes).logits
    assert rel_err(sub, o1["logits"][:, :1000]) < tol


def test_mlp3d_alternative_wave_tiles_are_bit_identical(golden, monkeypatch):
    """`DDMI_MLP3D_WAVES=4` / `=64` (wave tiles of 64 channels x 128 / 64 points: half the LDS reads per MFMA; both measured SLOWER)
    accumulate every output in the order of the shipped eight-wave form: same bits, incl. a ragged last tile."""
    o1 = golden.npz("O1_occupancy.npz")
    mlp = MLP3D(numerics="bf16")
    mlp.load_state_dict(named_state("O1.mlp3d", mlp.manifest()))
    planes = tuple([torch.from_numpy(o1[f"{n}{i}"]).to(DEV) for i in range(3)] for n in ("xy", "yz", "xz"))
    pts = torch.from_numpy(o1["points"][:, :4001])
    monkeypatch.setenv("DDMI_MLP3D_WAVES", "8")
    a = mlp(pts, planes).logits.clone()
    for form in ("4", "64"):   # four waves x (64 channels x 128 points); eight waves x (64 channels x 64 points)
        monkeypatch.setenv("DDMI_MLP3D_WAVES", form)
        b = mlp(pts, planes).logits.clone()
        assert torch.equal(a, b), form
        assert rel_err(b, o1["logits"][:, :4001]) < TOL_BF16


def test_manifests_match_reference(golden):
    man = golden.json("manifests.json")
    cfgs = golden.json("configs.json")
    same = lambda ours, ref: {k: tuple(v) for k, v in ours} == {k: tuple(v) for k, v in ref.items()}
    assert same(decoder3d_manifest(cfgs["TOY_DD3"], cfgs["TOY_EMBED3"]), man["O1.vae3d"])
    assert same(mlp3d_manifest(**cfgs["MLP3D_CFG"]), man["O1.mlp3d"])


def test_shapenet_full_size_vs_oracle():
    """configs/ldm/shapenet.yaml sizes: 678.7 M-parameter UNet on the [192,16,16] latent, the tri-plane decoder to
    3 x (16^2, 32^2, 64^2) planes, and the 128^3 query grid (2 097 152 points, ONE launch) against the CPU oracle on a
    32^3 sub-grid plus full-grid consistency."""
    unet = UNetModel(**SHAPENET_UNET, numerics="fp32", max_batch=1)
    ust = named_state("shapenet.unet", unet.manifest())
    unet.load_state_dict(ust)
    x = iw.normal("shapenet", "x", (1, 192, 16, 16))
    t = torch.tensor([777], dtype=torch.int64)
    ref = O.unet_forward(O.to_state(ust), SHAPENET_UNET, torch.from_numpy(x), t)
    assert rel_err(unet(torch.from_numpy(x).to(DEV), t.to(DEV)), ref) < TOL_FP32
    del unet
    torch.cuda.empty_cache()

    vae = Autoencoder3D(SHAPENET_DD, 64, numerics="fp32", max_batch=1)
    vst = named_state("shapenet.vae", vae.manifest())
    vae.load_state_dict(vst)
    z = iw.normal("shapenet", "z", (1, 192, 16, 16))
    ref_planes = O.vae3d_decode(O.to_state(vst), SHAPENET_DD, 64, torch.from_numpy(z))
    planes = vae.decode(torch.from_numpy(z).to(DEV))
    for got, want in zip(planes, ref_planes):
        assert [tuple(p.shape[2:]) for p in got] == [(16, 16), (32, 32), (64, 64)]
        for g, w in zip(got, want):
            assert rel_err(g, w) < TOL_FP32

    mlp = MLP3D(numerics="fp32")
    mst = named_state("shapenet.mlp", mlp.manifest())
    mlp.load_state_dict(mst)
    full = occupancy_grid_logits(mlp, planes, 128)                         # [1,128,128,128]
    assert full.shape == (1, 128, 128, 128) and bool(torch.isfinite(full).all())
    pts = O.occupancy_query_points(128).reshape(128, 128, 128, 3)[::4, ::4, ::4].reshape(1, -1, 3)   # 32^3 sub-grid
    ref_logits = O.mlp3d(O.to_state(mst), pts, ref_planes).reshape(32, 32, 32)
    assert rel_err(full[0, ::4, ::4, ::4], ref_logits) < TOL_FP32
    mlp16 = MLP3D(numerics="bf16")
    mlp16.load_state_dict(mst)
    planes16 = tuple([p.to(torch.bfloat16).contiguous(memory_format=torch.channels_last) for p in lst] for lst in planes)
    full16 = occupancy_grid_logits(mlp16, planes16, 128)
    assert rel_err(full16[0, ::4, ::4, ::4], ref_logits) < TOL_BF16
