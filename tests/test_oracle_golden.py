"""The CPU oracle (oracle/ddmi_oracle.py) against vectors captured from the real reference
(oracle/gen_golden.py -> tests/golden).  This is what pins the oracle; see SURVEY.md 8(c)."""
import numpy as np
import pytest
import torch

from conftest import rel_err
from ddmi_amd import initweights as iw
from oracle import ddmi_oracle as O

TOL = 2e-5  # fp32 re-association noise between two CPU implementations of the same math


def state(golden, name, tag=None, overrides=None):
    man = golden.json("manifests.json")[name]
    st = iw.fill_state(tag or name, [(k, tuple(v)) for k, v in man.items()])
    return O.to_state(st)


def test_schedule_buffers_bitexact(golden):
    s1 = golden.npz("S1_schedule.npz")
    for T in (1000, 250, 10):
        sched = O.make_schedule(T)
        for k, v in sched.items():
            ref = s1[f"T{T}.{k}"]
            assert v.dtype == np.float32
            # fp64 -> fp32 of identical formulae: identical bit patterns
            assert np.array_equal(v.view(np.uint32), ref.view(np.uint32)), (T, k)


def test_ddim_time_pairs_bitexact(golden):
    pairs = golden.json("S1_time_pairs.json")
    for key, ref in pairs.items():
        T, S = map(int, key.split("_"))
        assert [list(p) for p in O.ddim_time_pairs(T, S)] == ref, key


def test_time_pairs_match_torch_linspace_everywhere():
    # every S for T=1000 and a few other T: the integer schedule must be bit-exact (north star)
    for T in (1000, 250, 100):
        for S in range(1, T):
            times = torch.linspace(-1, T - 1, steps=S + 1)
            times = list(reversed(times.int().tolist()))
            assert O.ddim_time_pairs(T, S) == list(zip(times[:-1], times[1:])), (T, S)


def test_step_arithmetic(golden):
    k1 = golden.npz("K1_step.npz")
    x, mo, nz = (torch.from_numpy(k1[k]) for k in ("x", "model_out", "noise"))
    logit = torch.from_numpy(k1["mixing_logit"])
    for T in (1000, 250):
        sched = O.make_schedule(T)
        for t in (0, 1, T // 2, T - 2, T - 1):
            out = O.p_sample_step(sched, x, mo, t, nz, logit)
            assert rel_err(out, k1[f"anc.T{T}.t{t}"]) < 1e-6, (T, t)
    sched = O.make_schedule(1000)
    out = O.p_sample_step(sched, x, mo, 500, nz, logit, clip_denoised=True)
    assert rel_err(out, k1["anc_clip.T1000.t500"]) < 1e-6
    for eta in (0.0, 1.0, 0.5):
        out = O.sample_chain(lambda xx, tt: mo, sched, x, [k1["ddim_n1"], k1["ddim_n2"]], 1000, 3, eta, logit)
        assert rel_err(out, k1[f"ddim3.eta{eta}"]) < 1e-6, eta


def test_unet_ops(golden):
    u1 = golden.npz("U1_unet_ops.npz")
    tt = torch.tensor([0, 1, 500, 999])
    for dim in (224, 32):
        assert rel_err(O.timestep_embedding(tt, dim), u1[f"temb{dim}"]) < 1e-6
    x = torch.from_numpy(u1["x"])
    p = state(golden, "U1.gn")
    assert rel_err(O.group_norm(x, p["weight"], p["bias"], 1e-5), u1["gn"]) < TOL
    assert rel_err(O.group_norm_explicit(x, p["weight"], p["bias"], 1e-5), u1["gn"]) < TOL
    assert rel_err(O.silu(O.group_norm(x, p["weight"], p["bias"], 1e-5)), u1["gn_silu"]) < TOL
    assert rel_err(O.qkv_attention_legacy(torch.from_numpy(u1["qkv"]), 2), u1["attn_legacy"]) < TOL
    emb = torch.from_numpy(u1["emb"])
    assert rel_err(O.unet_resblock(state(golden, "U1.res"), "", x, emb), u1["res"]) < TOL
    assert rel_err(O.unet_resblock(state(golden, "U1.res_same"), "", x, emb), u1["res_same"]) < TOL
    assert rel_err(O.unet_attnblock(state(golden, "U1.attn"), "", x, 32), u1["attn"]) < TOL
    p = state(golden, "U1.down")
    assert rel_err(torch.nn.functional.conv2d(x, p["op.weight"], p["op.bias"], stride=2, padding=1), u1["down"]) < TOL
    p = state(golden, "U1.up")
    up = torch.nn.functional.interpolate(x, scale_factor=2, mode="nearest")
    assert rel_err(torch.nn.functional.conv2d(up, p["conv.weight"], p["conv.bias"], padding=1), u1["up"]) < TOL


@pytest.mark.parametrize("name,cfgname", [("a", "TOY_UNET"), ("b", "TOY_UNET_B")])
def test_unet_forward(golden, name, cfgname):
    u2 = golden.npz("U2_unet.npz")
    cfg = golden.json("configs.json")[cfgname]
    p = state(golden, f"U2{name}.unet")
    eps = O.unet_forward(p, cfg, torch.from_numpy(u2[f"{name}.x"]), torch.from_numpy(u2[f"{name}.t"]))
    assert np.abs(u2[f"{name}.eps"]).max() > 0.1  # not the zero-init trap
    assert rel_err(eps, u2[f"{name}.eps"]) < TOL


def test_decoder(golden):
    d1 = golden.npz("D1_decoder.npz")
    cfg = golden.json("configs.json")["TOY_DD"]
    planes = O.vae_decode(state(golden, "D1.vae"), cfg, torch.from_numpy(d1["z"]))
    assert [tuple(p.shape[2:]) for p in planes] == [(8, 8), (16, 16), (32, 32)]
    for i, pl in enumerate(planes):
        assert rel_err(pl, d1[f"plane{i}"]) < TOL, i


def test_plane_sampling_and_mlp(golden):
    d1 = golden.npz("D1_decoder.npz")
    m1 = golden.npz("M1_mlp.npz")
    planes = [torch.from_numpy(d1[f"plane{i}"]) for i in range(3)]
    g = torch.from_numpy(m1["grid"])
    assert (g.abs() > 1).any()  # border clamp exercised
    for i, pl in enumerate(planes):
        got = O.bilinear_border(pl, g[..., 0].reshape(2, -1), g[..., 1].reshape(2, -1), align_corners=False)
        assert rel_err(got.reshape(2, 64, 5, 7), m1[f"sample{i}"]) < TOL
    p = state(golden, "M1.mlp")
    for R in (32, 48, 8):
        rgb = O.image_mlp(p, torch.from_numpy(m1[f"coords{R}"]), planes, float(m1[f"si{R}"]))
        assert rel_err(rgb, m1[f"rgb{R}"]) < TOL, R
        assert rel_err(O.generate_coords(R), m1[f"coords{R}"]) < 2e-7  # torch.linspace is SIMD-width dependent at 1 ulp
        assert O.scale_injection(R) == float(m1[f"si{R}"])
    # NoiseInjection with non-zero weights and explicit draws
    man = golden.json("manifests.json")["M1n.mlp"]
    st = iw.fill_state("M1n.mlp", [(k, tuple(v)) for k, v in man.items()])
    for k in st:
        if k.endswith("noise.weight"):
            st[k] = iw.normal("M1n.mlp", k, st[k].shape, 0.0, 0.1)
    draws = np.stack([iw.normal("M1n", f"draw{i}", (2, 1, 32, 32)).reshape(2, 32 * 32) for i in range(12)])
    rgb = O.image_mlp(O.to_state(st), torch.from_numpy(m1["coords32"]), planes, float(m1["si32"]), noise=draws)
    assert rel_err(rgb, m1["rgb32_noise"]) < TOL
    assert rel_err(m1["rgb32_noise"], m1["rgb32"]) > 1e-3  # the noise path really changes the output


def test_coords(golden):
    c1 = golden.npz("C1_coords.npz")
    # ATen's vectorised linspace evaluates base + lane*step per SIMD chunk, so the reference's own grid
    # depends on the host's vector width at the 1-ulp level; the oracle uses the scalar formula.
    for R in (4, 64, 256):
        assert rel_err(O.generate_coords(R).numpy(), c1[f"grid{R}"]) < 2e-7
    assert rel_err(O.coord_grid_2d(5, 5).numpy(), c1["grid5_default"]) < 2e-7


@pytest.mark.parametrize("name,T,S,eta", [("ddim_eta0", 1000, 10, 0.0), ("ddim_eta1", 1000, 10, 1.0),
                                          ("anc", 10, 10, 1.0), ("anc250", 250, 250, 1.0)])
def test_generate_chain(golden, name, T, S, eta):
    g1 = golden.npz("G1_generate.npz")
    cfgs = golden.json("configs.json")
    unet_p = state(golden, "U2a.unet")
    logit = torch.from_numpy(iw.fill_state("G1.ddpm", [("mixing_logit", (1, 64, 1, 1))])["mixing_logit"])
    shape = g1["x_T"].shape
    ndraw = S - 1 if S < T else T
    noises = [iw.normal("G1", f"{name}.n{i}", shape) for i in range(ndraw)]
    sched = O.make_schedule(T)
    z = O.sample_chain(lambda x, t: O.unet_forward(unet_p, cfgs["TOY_UNET"], x, t), sched, g1["x_T"], noises, T, S, eta, logit)
    assert rel_err(z, g1[f"{name}.z"]) < 1e-4, name
    if name == "anc250":
        return
    planes = O.vae_decode(state(golden, "D1.vae"), cfgs["TOY_DD"], z)
    img = O.image_mlp(state(golden, "M1.mlp"), O.generate_coords(32), planes, O.scale_injection(32)).clamp(-1, 1)
    assert rel_err(img, g1[f"{name}.img"]) < 1e-4


def test_occupancy_decoder_and_mlp3d(golden):
    """Config-4 path: Autoencoder3D.decode, tri-plane 'add' sampling incl. both clamp branches
    of normalize_coordinate, MLP3D logits, and the dense query grid."""
    o1 = golden.npz("O1_occupancy.npz")
    cfgs = golden.json("configs.json")
    xy, yz, xz = O.vae3d_decode(state(golden, "O1.vae3d"), cfgs["TOY_DD3"], cfgs["TOY_EMBED3"], torch.from_numpy(o1["z"]))
    assert [tuple(p.shape[2:]) for p in xy] == [(4, 4), (8, 8), (16, 16)]
    for name, lst in (("xy", xy), ("yz", yz), ("xz", xz)):
        for i, pl in enumerate(lst):
            assert rel_err(pl, o1[f"{name}{i}"]) < TOL, (name, i)
    ref_planes = tuple([torch.from_numpy(o1[f"{n}{i}"]) for i in range(3)] for n in ("xy", "yz", "xz"))
    pts = torch.from_numpy(o1["points"])
    assert (pts.abs() > 0.56).any()  # clamp branches exercised
    logits = O.mlp3d(state(golden, "O1.mlp3d"), pts, ref_planes)
    assert rel_err(logits, o1["logits"]) < TOL
    grid = O.occupancy_query_points(6)
    assert rel_err(grid, o1["grid6"]) < 2e-7
    one = tuple([pl[:1] for pl in lst] for lst in ref_planes)
    assert rel_err(O.mlp3d(state(golden, "O1.mlp3d"), grid[None], one)[0], o1["grid6_logits"]) < TOL


# ---------------- G2 (BASELINE config 1) and G3 (same-seed draw order), captured by oracle/gen_golden_g2.py -------------
def reference_draw_order(seed, shape, T, S, R, inject=True):
    """The reference's draws from torch's global CPU generator inside one generate(): x_T (ddpm.py:288/343),
    one randn per step (ddpm.py:278; :363 on every non-final DDIM step), then the 12 NoiseInjection normal_() draws
    [B,1,R,R] (blocks.py:293-295)."""
    torch.manual_seed(seed)
    x_T = torch.randn(shape)
    steps = [torch.randn(shape) for _ in range(S - 1 if S < T else T)]
    inj = torch.stack([torch.empty(shape[0], 1, R, R).normal_() for _ in range(12)]).reshape(12, shape[0], R * R) if inject else None
    return x_T, steps, inj


@pytest.fixture(scope="module")
def afhq_states():
    from helpers import AFHQ_DD, AFHQ_UNET
    from ddmi_amd.autoencoder import decoder_manifest
    from ddmi_amd.mlp import mlp_manifest
    from ddmi_amd.unet import unet_manifest
    st = lambda tag, man: O.to_state(iw.fill_state(tag, [(k, tuple(s)) for k, s in man]))
    return (st("afhq.unet", unet_manifest(**AFHQ_UNET)), st("afhq.vae", decoder_manifest(AFHQ_DD, 64)), st("afhq.mlp", mlp_manifest()))


@pytest.mark.parametrize("name,T,S,eta", [("ddim", 1000, 10, 0.0), ("anc", 10, 10, 1.0)])
def test_config1_generate_g2(golden, afhq_states, name, T, S, eta):
    """BASELINE config 1: AFHQ full-width nets, latent [1,64,16,16], 10 steps, R=64 (si=4), torch.manual_seed(777)."""
    from helpers import AFHQ_DD, AFHQ_UNET
    g2 = golden.npz("G2_config1.npz")
    shape = (1, 64, 16, 16)
    x_T, steps, _ = reference_draw_order(int(g2["seed"]), shape, T, S, 64, inject=False)
    assert np.array_equal(x_T.numpy(), g2["x_T"])          # torch's seeded CPU stream is what the fixture was drawn from
    logit = torch.from_numpy(iw.fill_state("afhq.ddpm", [("mixing_logit", (1, 64, 1, 1))])["mixing_logit"])
    with torch.no_grad():
        ref = O.generate(afhq_states[0], AFHQ_UNET, afhq_states[1], AFHQ_DD, afhq_states[2], O.make_schedule(T), logit, x_T,
                         steps, T, S, eta, 64)
    assert rel_err(ref["z"], g2[f"{name}.z"]) < 1e-4
    for i, pl in enumerate(ref["planes"]):
        a = pl.numpy()
        assert list(a.shape) == list(g2[f"{name}.plane{i}.shape"])
        assert rel_err(a[0, :4, :16, :16], g2[f"{name}.plane{i}.slice"]) < 1e-4
        assert abs(a.astype(np.float64).sum() - float(g2[f"{name}.plane{i}.sum"])) < 1e-4 * float(g2[f"{name}.plane{i}.abssum"])
    assert rel_err(ref["img"], g2[f"{name}.img"]) < 1e-4


@pytest.mark.parametrize("name,T,S,eta", [("ddim_eta0", 1000, 10, 0.0), ("ddim_eta1", 1000, 10, 1.0), ("anc", 10, 10, 1.0)])
def test_same_seed_draw_order_g3(golden, name, T, S, eta):
    """The reference's generate() under torch.manual_seed(777) with NOTHING patched and non-zero NoiseInjection weights:
    pins the draw order x_T -> per-step -> 12 injections (a24)."""
    g3 = golden.npz("G3_same_seed.npz")
    cfgs = golden.json("configs.json")
    shape = (2, 64, 8, 8)
    x_T, steps, inj = reference_draw_order(int(g3["seed"]), shape, T, S, 32)
    man = golden.json("manifests.json")["M1n.mlp"]
    mst = iw.fill_state("M1n.mlp", [(k, tuple(v)) for k, v in man.items()])
    for k in mst:
        if k.endswith("noise.weight"):
            mst[k] = iw.normal("M1n.mlp", k, mst[k].shape, 0.0, 0.1)
    logit = torch.from_numpy(iw.fill_state("G1.ddpm", [("mixing_logit", (1, 64, 1, 1))])["mixing_logit"])
    unet_p = state(golden, "U2a.unet")
    z = O.sample_chain(lambda x, t: O.unet_forward(unet_p, cfgs["TOY_UNET"], x, t), O.make_schedule(T), x_T, steps, T, S, eta, logit)
    assert rel_err(z, g3[f"{name}.z"]) < 1e-4
    planes = O.vae_decode(state(golden, "D1.vae"), cfgs["TOY_DD"], z)
    raw = O.image_mlp(O.to_state(mst), O.generate_coords(32), planes, O.scale_injection(32), noise=inj)
    assert rel_err(raw, g3[f"{name}.raw"]) < 1e-4
    assert rel_err(raw.clamp(-1, 1), g3[f"{name}.img"]) < 1e-4
    wrong = O.image_mlp(O.to_state(mst), O.generate_coords(32), planes, O.scale_injection(32), noise=inj.flip(0))
    assert rel_err(wrong, g3[f"{name}.raw"]) > 1e-3     # the ORDER of the 12 draws is pinned, not just their presence


def test_oracle_training_step_matches_reference_fixture_t1(golden):
    """Pins oracle.p_losses + torch autograd (the checker of the full-size training test) to fixture T1 = the REAL reference's
    `DDPM.p_losses` -> backward on the toy UNet (oracle/gen_golden_train.py): loss and every gradient's (sum, l2), 22 in full."""
    import torch
    from ddmi_amd import initweights as iw
    from ddmi_amd.unet import unet_manifest
    from oracle import ddmi_oracle as O
    t1 = golden.npz("T1_train_step.npz")
    cfg = golden.json("configs.json")["TOY_UNET"]
    st = iw.fill_state("T1.unet", [(k, tuple(s)) for k, s in unet_manifest(**cfg)])
    p = {k: torch.from_numpy(np.ascontiguousarray(v)).requires_grad_(True) for k, v in st.items()}
    logit = torch.from_numpy(iw.normal("T1", "mixing_logit", (1, 64, 1, 1), 0.0, 0.5)).requires_grad_(True)
    x0 = torch.from_numpy(iw.normal("T1", "x_start", (2, 64, 8, 8)))
    nz = torch.from_numpy(iw.normal("T1", "noise", (2, 64, 8, 8)))
    loss = O.p_losses(p, cfg, O.make_schedule(1000), logit, x0, torch.from_numpy(t1["t"]), nz)
    loss.backward()
    assert abs(float(loss) - float(t1["loss"])) < 1e-6 * float(t1["loss"])
    grads = {"model." + k: v.grad for k, v in p.items()}
    grads["mixing_logit"] = logit.grad
    l2max = float(t1["all_l2"].max())
    for k, s_ref, l2_ref in zip(t1["all_keys"], t1["all_sum"], t1["all_l2"]):
        g = grads[str(k)].double()
        assert abs(float(g.norm()) - l2_ref) < 1e-4 * l2_ref + 1e-7 * l2max, k
        assert abs(float(g.sum()) - s_ref) < 1e-4 * l2_ref * np.sqrt(g.numel()) + 1e-7 * l2max, k
    for k in [k[5:] for k in t1 if k.startswith("grad.")]:
        ref = t1["grad." + k]
        assert np.abs(grads[k].numpy() - ref).max() < 1e-4 * np.abs(ref).max() + 1e-7 * l2max, k
