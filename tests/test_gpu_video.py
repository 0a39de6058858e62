"""Video path on the GPU, through the C ABI: tri-plane UNet, VITAutoencoder.decode,
MLPVideo and the whole sample -> decode -> MLP chain against the vectors captured from the reference
(oracle/gen_golden_video.py), in both numerics modes."""
import numpy as np
import pytest
import torch

from ddmi_amd import initweights as iw
from ddmi_amd.ddpm import DDPM
from ddmi_amd.video import MLPVideo, UNetModel_Triplane, VITAutoencoder, convert_to_coord_format_3d
from helpers import TOL_BF16, TOL_FP32, named_state, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def make(cls, tag, numerics, *args, **kw):
    m = cls(*args, numerics=numerics, **kw)
    m.load_state_dict(named_state(tag, m.manifest()))
    return m


def toy_coords(cfgs):
    R, Tn = cfgs["TOY_DDV"]["resolution"], cfgs["TOY_FRAMES"]
    e = (R - 1) / R
    return convert_to_coord_format_3d(1, R, R, Tn, hstart=-e, hend=e, wstart=-e, wend=255 / R, tstart=-3 / 4, tend=3 / 4)


@pytest.mark.parametrize("numerics,tol", [("fp32", TOL_FP32), ("bf16", TOL_BF16)])
def test_unet_triplane_golden(golden, numerics, tol):
    v1 = golden.npz("V1_unet3.npz")
    cfg = golden.json("configs_video.json")["TOY_UNET_V"]
    unet = make(UNetModel_Triplane, "V1.unet3", numerics, max_batch=2, **cfg)
    x = torch.from_numpy(v1["x"]).to(DEV)
    for name, ts in (("t7", [7, 7]), ("t3_11", [3, 11])):
        y = unet(x, torch.tensor(ts, dtype=torch.int64, device=DEV))
        assert rel_err(y, v1[name]) < tol, name
    # a batch of one gives the same rows (per-sample ops only)
    y1 = unet(x[:1].contiguous(), torch.tensor([3], dtype=torch.int64, device=DEV))
    assert rel_err(y1, v1["t3_11"][:1]) < tol


@pytest.mark.parametrize("numerics,tol", [("fp32", TOL_FP32), ("bf16", TOL_BF16)])
def test_vit_decode_and_mlp_video_golden(golden, numerics, tol):
    cfgs = golden.json("configs_video.json")
    v2, v3 = golden.npz("V2_vit_decode.npz"), golden.npz("V3_mlpvideo.npz")
    vit = make(VITAutoencoder, "V2.vit", numerics, cfgs["TOY_DDV"], cfgs["TOY_EMBED_V"], cfgs["TOY_FRAMES"], max_batch=1)
    z = torch.from_numpy(v2["z"]).to(DEV)
    hdbf = vit.decode(z, reference_layout=True)
    for name, lst in zip(("xy", "yt", "xt"), hdbf):
        assert len(lst) == 3
        for i, pl in enumerate(lst):
            assert tuple(pl.shape) == v2[f"{name}{i}"].shape
            assert rel_err(pl, v2[f"{name}{i}"]) < tol, (name, i)
    mlp = make(MLPVideo, "V3.mlpv", numerics, **cfgs["MLPV_CFG"])
    coords = toy_coords(cfgs)
    # (1) MLP alone on the reference's planes, (2) on the planes the decoder just produced in its native layout
    ref_planes = tuple([torch.from_numpy(v2[f"{n}{i}"]).to(DEV) for i in range(3)] for n in ("xy", "yt", "xt"))
    vid = mlp(coords, ref_planes)
    assert tuple(vid.shape) == v3["video"].shape
    assert rel_err(vid, v3["video"]) < tol
    vid2 = mlp(coords, vit.decode(z))
    assert rel_err(vid2, v3["video"]) < tol


@pytest.mark.parametrize("name,T,S,eta", [("ddim_eta0", 20, 4, 0.0), ("anc", 5, 5, 1.0)])
def test_generate_video_chain_golden(golden, name, T, S, eta):
    """tools/ldm/video.py:214-230 end to end in parity mode, noise fed explicitly in the reference's draw order."""
    cfgs = golden.json("configs_video.json")
    v4 = golden.npz("V4_generate_video.npz")
    unet = make(UNetModel_Triplane, "V1.unet3", "fp32", max_batch=1, **cfgs["TOY_UNET_V"])
    C = cfgs["TOY_UNET_V"]["in_channels"]
    ddpm = DDPM(model=unet, timesteps=T, sampling_timesteps=S, ddim_sampling_eta=eta, channels=C, domain="video")
    ddpm.load_state_dict({"mixing_logit": iw.fill_state("V4.ddpm", [("mixing_logit", (1, C, 1))])["mixing_logit"]}, strict=False)
    shape = list(v4["x_T"].shape)
    ndraw = S - 1 if S < T else T
    noises = [iw.normal("V4", f"{name}.n{i}", shape) for i in range(ndraw)]
    z = ddpm.sample(shape, noise=v4["x_T"], step_noise=noises, device=DEV)
    assert rel_err(z, v4[f"{name}.z"]) < TOL_FP32
    vit = make(VITAutoencoder, "V2.vit", "fp32", cfgs["TOY_DDV"], cfgs["TOY_EMBED_V"], cfgs["TOY_FRAMES"], max_batch=1)
    mlp = make(MLPVideo, "V3.mlpv", "fp32", **cfgs["MLPV_CFG"])
    vid = mlp(toy_coords(cfgs), vit.decode(z)).clamp(-1.0, 1.0)
    assert rel_err(vid, v4[f"{name}.video"]) < TOL_FP32
