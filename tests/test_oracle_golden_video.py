"""The CPU oracle's VIDEO path against vectors captured from the real reference
(oracle/gen_golden_video.py -> tests/golden/V*.npz)."""
import numpy as np
import pytest
import torch

from conftest import rel_err
from ddmi_amd import initweights as iw
from oracle import ddmi_oracle as O

TOL = 2e-5


def vstate(golden, name):
    man = golden.json("manifests_video.json")[name]
    return O.to_state(iw.fill_state(name, [(k, tuple(v)) for k, v in man.items()]))


def toy_coords(golden):
    cfg = golden.json("configs_video.json")
    R, Tn = cfg["TOY_DDV"]["resolution"], cfg["TOY_FRAMES"]
    e = (R - 1) / R
    return O.coord_grid_3d(R, R, Tn, -e, e, -e, 255 / R, -3 / 4, 3 / 4)


def test_attention_1d_blocks(golden):
    v0 = golden.npz("V0_attn1d.npz")
    y = O.attn_block_1d(vstate(golden, "V0.attn16"), "", torch.from_numpy(v0["x16"]), 16, False)
    assert rel_err(y, v0["y16"]) < TOL
    y = O.attn_block_1d(vstate(golden, "V0.attn8e"), "", torch.from_numpy(v0["x8e"]), 8, True)
    assert rel_err(y, v0["y8e"]) < TOL


@pytest.mark.parametrize("name,ts", [("t7", [7, 7]), ("t3_11", [3, 11])])
def test_unet_triplane_forward(golden, name, ts):
    v1 = golden.npz("V1_unet3.npz")
    cfg = golden.json("configs_video.json")["TOY_UNET_V"]
    y = O.unet_triplane_forward(vstate(golden, "V1.unet3"), cfg, torch.from_numpy(v1["x"]), torch.tensor(ts))
    assert rel_err(y, v1[name]) < TOL


def test_vit_decode_and_mlp_video(golden):
    cfgs = golden.json("configs_video.json")
    v2 = golden.npz("V2_vit_decode.npz")
    hdbf = O.vit_decode(vstate(golden, "V2.vit"), cfgs["TOY_DDV"], cfgs["TOY_FRAMES"], torch.from_numpy(v2["z"]))
    for name, lst in zip(("xy", "yt", "xt"), hdbf):
        assert len(lst) == 3
        for i, pl in enumerate(lst):
            assert rel_err(pl, v2[f"{name}{i}"]) < TOL, (name, i)
    v3 = golden.npz("V3_mlpvideo.npz")
    coords = toy_coords(golden)
    for k in ("xy", "xt", "yt"):
        assert np.abs(coords[k].numpy() - v3[f"coords_{k}"]).max() < 2e-7  # torch.linspace is SIMD-width dependent at 1 ulp
    ref_planes = tuple([torch.from_numpy(v2[f"{n}{i}"]) for i in range(3)] for n in ("xy", "yt", "xt"))
    vid = O.mlp_video(vstate(golden, "V3.mlpv"), coords, ref_planes)
    assert tuple(vid.shape) == v3["video"].shape
    assert rel_err(vid, v3["video"]) < TOL


@pytest.mark.parametrize("name,T,S,eta", [("ddim_eta0", 20, 4, 0.0), ("anc", 5, 5, 1.0)])
def test_generate_video_chain(golden, name, T, S, eta):
    cfgs = golden.json("configs_video.json")
    v4 = golden.npz("V4_generate_video.npz")
    shape = v4["x_T"].shape
    ndraw = S - 1 if S < T else T
    draws = [iw.normal("V4", f"{name}.n{i}", shape) for i in range(ndraw)]
    logit = torch.from_numpy(iw.fill_state("V4.ddpm", [("mixing_logit", (1, 8, 1))])["mixing_logit"])
    out = O.generate_video(vstate(golden, "V1.unet3"), cfgs["TOY_UNET_V"], vstate(golden, "V2.vit"), cfgs["TOY_DDV"],
                           cfgs["TOY_FRAMES"], vstate(golden, "V3.mlpv"), O.make_schedule(T), logit,
                           torch.from_numpy(v4["x_T"]), draws, T, S, eta, toy_coords(golden))
    assert rel_err(out["z"], v4[f"{name}.z"]) < 1e-4
    assert rel_err(out["video"], v4[f"{name}.video"]) < 1e-4


def test_oracle_video_training_step_matches_reference_fixture_t2(golden):
    """Pins oracle.p_losses(model=unet_triplane_forward) + torch autograd - the checker of the video-domain training tests - to fixture T2 =
    the REAL reference's `DDPM(domain='video').p_losses` -> backward on a toy `UNetModel_Triplane` (oracle/gen_golden_train_video.py;
    tools/ldm/video.py:150-167): the loss, every gradient's (sum, l2), 31 tensors in full - the shared-weight convolutions over three
    planes, the per-plane AttentionBlock, the 16-head inter-plane attention, mixing_logit [1, C, 1]."""
    import numpy as np
    import torch
    from ddmi_amd import initweights as iw
    from ddmi_amd.video import unet3_manifest
    from oracle import ddmi_oracle as O
    t2 = golden.npz("T2_train_step_video.npz")
    cfg = golden.json("T2_config.json")["TOY_UNET_T2"]
    st = iw.fill_state("T2.unet3", [(k, tuple(s)) for k, s in unet3_manifest(**cfg)])
    p = {k: torch.from_numpy(np.ascontiguousarray(v)).requires_grad_(True) for k, v in st.items()}
    L = cfg["size1"] * cfg["size2"] + cfg["size1"] * cfg["size3"] + cfg["size2"] * cfg["size3"]
    logit = torch.from_numpy(iw.normal("T2", "mixing_logit", (1, 8, 1), 0.0, 0.5)).requires_grad_(True)
    x0 = torch.from_numpy(iw.normal("T2", "x_start", (2, 8, L)))
    nz = torch.from_numpy(iw.normal("T2", "noise", (2, 8, L)))
    loss = O.p_losses(p, cfg, O.make_schedule(1000), logit, x0, torch.from_numpy(t2["t"]), nz, model=O.unet_triplane_forward)
    loss.backward()
    assert abs(float(loss) - float(t2["loss"])) < 1e-6 * float(t2["loss"])
    grads = {"model." + k: v.grad for k, v in p.items()}
    grads["mixing_logit"] = logit.grad
    assert set(grads) == set(str(k) for k in t2["all_keys"])
    l2max = float(t2["all_l2"].max())
    for k, s_ref, l2_ref in zip(t2["all_keys"], t2["all_sum"], t2["all_l2"]):
        g = grads[str(k)].double()
        assert abs(float(g.norm()) - l2_ref) < 1e-4 * l2_ref + 1e-7 * l2max, k
        assert abs(float(g.sum()) - s_ref) < 1e-4 * l2_ref * np.sqrt(g.numel()) + 1e-7 * l2max, k
    for k in [k[5:] for k in t2 if k.startswith("grad.")]:
        ref = t2["grad." + k]
        assert np.abs(grads[k].numpy() - ref).max() < 1e-4 * np.abs(ref).max() + 1e-7 * l2max, k
