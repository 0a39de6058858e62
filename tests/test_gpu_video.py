"""Video path on the GPU, through the C ABI: tri-plane UNet, VITAutoencoder.decode,
MLPVideo and the whole sample -> decode -> MLP chain against the vectors captured from the reference
(oracle/gen_golden_video.py), in both numerics modes."""
import numpy as np
import pytest
import torch

from ddmi_amd import initweights as iw
from ddmi_amd.ddpm import DDPM
from ddmi_amd.video import MLPVideo, UNetModel_Triplane, VITAutoencoder, convert_to_coord_format_3d
from helpers import TOL_BF16, TOL_FP32, assert_bf16_close, named_state, rel_err

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


# ---------------- SkyTimelapse shapes against the CPU oracle (same named-seed weights) -----------------------------
def test_sky_unet_full_size_vs_oracle():
    from helpers import SKY_UNET
    from oracle import ddmi_oracle as O
    unet32 = UNetModel_Triplane(numerics="fp32", max_batch=1, **SKY_UNET)
    st = named_state("sky.unet3", unet32.manifest())
    unet32.load_state_dict(st)
    x = iw.normal("sky", "x", (1, 64, 2048))
    t = torch.tensor([437], dtype=torch.int64)
    ref = O.unet_triplane_forward(O.to_state(st), SKY_UNET, torch.from_numpy(x), t)
    eps32 = unet32(torch.from_numpy(x).to(DEV), t.to(DEV))
    assert rel_err(eps32, ref) < TOL_FP32
    del unet32
    torch.cuda.empty_cache()
    unet16 = UNetModel_Triplane(numerics="bf16", max_batch=1, **SKY_UNET)
    unet16.load_state_dict(st)
    eps16 = unet16(torch.from_numpy(x).to(DEV), t.to(DEV))
    assert_bf16_close(eps16, ref, "sky UNet full size")


def test_sky_decoder_quarter_resolution_vs_oracle():
    """VideoDecoder_light with the real widths / heads / 16 frames at resolution 64 (latent 8x8 + 2 x 16x8): the largest
    size whose 8-head attention matrices (6144^2 at the top level) the CPU oracle can hold."""
    from helpers import SKY_DD, SKY_EMBED, SKY_FRAMES
    from oracle import ddmi_oracle as O
    dd = dict(SKY_DD, resolution=64, hdbf_resolutions=[32, 16], inter_attn_resolutions=[8, 16, 32, 64])
    vit = VITAutoencoder(dd, SKY_EMBED, SKY_FRAMES, numerics="fp32", max_batch=1)
    st = named_state("sky.vit", vit.manifest())
    vit.load_state_dict(st)
    z = iw.normal("sky", "z64", (1, SKY_EMBED, vit.seq_len))
    ref = O.vit_decode(O.to_state(st), dd, SKY_FRAMES, torch.from_numpy(z))
    got = vit.decode(torch.from_numpy(z).to(DEV), reference_layout=True)
    for lst_g, lst_r in zip(got, ref):
        for g, r in zip(lst_g, lst_r):
            assert tuple(g.shape) == tuple(r.shape)
            assert rel_err(g, r) < TOL_FP32
    vit16 = VITAutoencoder(dd, SKY_EMBED, SKY_FRAMES, numerics="bf16", max_batch=1)
    vit16.load_state_dict(st)
    got16 = vit16.decode(torch.from_numpy(z).to(DEV), reference_layout=True)
    for lst_g, lst_r in zip(got16, ref):
        for g, r in zip(lst_g, lst_r):
            assert_bf16_close(g, r, "sky decoder quarter resolution")


def test_attention_at_the_full_video_length_sampled_queries():
    """The decoder's last inter-plane attention: 73 728 tokens (256^2 + 2 x 16 x 256), 8 heads of 64 channels.  The
    full score matrix is 174 GB in fp32, so the oracle side is evaluated for a sample of query rows only."""
    import ctypes as C
    from ddmi_amd import _lib
    heads, d, N = 8, 64, 256 * 256 + 2 * 16 * 256
    g = torch.Generator().manual_seed(7)
    q = torch.randn((1, heads * d, N), generator=g) * 0.7
    k = torch.randn((1, heads * d, N), generator=g) * 0.7
    v = torch.randn((1, heads * d, N), generator=g)
    out = torch.empty((1, heads * d, N), device=DEV, dtype=torch.float32)
    qd, kd, vd = q.to(DEV), k.to(DEV), v.to(DEV)
    _lib.init_device(0)
    _lib.check(_lib.lib().ddmi_op_attention_mh(qd.data_ptr(), kd.data_ptr(), vd.data_ptr(), out.data_ptr(), 1, heads, d, N,
                                              256 * 256, 16 * 256, _lib.numerics_code("bf16")))
    out = out.cpu()
    rows = torch.tensor([0, 1, 63, 64, 4097, 65535, 65536, 65537, 69631, 69632, 73727, 31337, 12345, 50000, 70000, 255])
    qh = q[0].reshape(heads, d, N)[:, :, rows].permute(0, 2, 1)        # [H, R, d]
    kh = k[0].reshape(heads, d, N)
    vh = v[0].reshape(heads, d, N).permute(0, 2, 1)                    # [H, N, d]
    w = torch.softmax(torch.matmul(qh, kh) * d ** -0.5, dim=-1)        # [H, R, N]
    ref = torch.matmul(w, vh)                                          # [H, R, d]
    got = out[0].reshape(heads, d, N)[:, :, rows].permute(0, 2, 1)
    assert rel_err(got, ref) < 4e-2


def test_sky_full_size_decode_vs_oracle():
    """BASELINE config 5's decode at FULL size against the CPU oracle: VITAutoencoder.decode (autoencoder_vit.py:188-202 ->
    VideoDecoder_light.forward autoencoder_unet.py:1040-1144) at resolution 256 / 16 frames - inter-plane attention over
    2 560 / 5 120 / 20 480 / 73 728 tokens (:1096-1109; the last one is 8 heads x 64 channels, 11 TFLOP) - then MLPVideo over all
    16 x 256 x 256 query points.  The oracle evaluates the attention in blocks of 512 queries (oracle.sdpa_heads), so no
    N x N matrix is ever held.  bf16 mode on all six plane lists and the RGB volume (L-infinity + relative L2 + per-channel
    cosine); the fp32 parity mode at 1e-3 on the same tensors."""
    import time
    from helpers import SKY_DD, SKY_EMBED, SKY_FRAMES
    from ddmi_amd.video import video_generation_coords
    from oracle import ddmi_oracle as O
    vit32 = VITAutoencoder(SKY_DD, SKY_EMBED, SKY_FRAMES, numerics="fp32", max_batch=1)
    st = named_state("sky.vit", vit32.manifest())
    mlp32 = MLPVideo(numerics="fp32")
    mst = named_state("sky.mlpv", mlp32.manifest())
    z = iw.normal("sky", "z", (1, SKY_EMBED, vit32.seq_len))
    t0 = time.time()
    with torch.no_grad():
        ref = O.vit_decode(O.to_state(st), SKY_DD, SKY_FRAMES, torch.from_numpy(z))          # (xy, yt, xt) plane lists
        ref_vid = O.mlp_video(O.to_state(mst), O.video_coords(256, SKY_FRAMES), ref)
    print("oracle: full-size video decode + MLPVideo %.1f s on %d threads" % (time.time() - t0, torch.get_num_threads()))
    names = ("xy", "yt", "xt")

    def check(vit, mlp, bf16):
        vit.load_state_dict(st)
        mlp.load_state_dict(mst)
        got = vit.decode(torch.from_numpy(z).to(DEV), reference_layout=True)
        for nm, lst_g, lst_r in zip(names, got, ref):
            for lvl, (g, r) in enumerate(zip(lst_g, lst_r)):
                assert tuple(g.shape) == tuple(r.shape)
                if bf16:
                    assert_bf16_close(g, r, f"sky decode full size {nm}[{lvl}]")
                else:
                    assert rel_err(g, r) < TOL_FP32, (nm, lvl, rel_err(g, r))
        vid = mlp(video_generation_coords(256, SKY_FRAMES), vit.decode(torch.from_numpy(z).to(DEV)))
        assert tuple(vid.shape) == (1, 3, SKY_FRAMES, 256, 256)
        if bf16:
            assert_bf16_close(vid, ref_vid.reshape(vid.shape), "sky decode + MLPVideo full size")
        else:
            assert rel_err(vid, ref_vid.reshape(vid.shape)) < TOL_FP32

    check(vit32, mlp32, False)
    del vit32
    torch.cuda.empty_cache()
    check(VITAutoencoder(SKY_DD, SKY_EMBED, SKY_FRAMES, numerics="bf16", max_batch=1), MLPVideo(numerics="bf16"), True)


def test_sky_full_size_decode_and_mlp_run():
    """Full-size VITAutoencoder.decode + MLPVideo in perf mode (B = 1): shapes, finiteness and the value range; the
    arithmetic itself is pinned by the tests above (no CPU oracle can hold this size)."""
    import time
    from helpers import SKY_DD, SKY_EMBED, SKY_FRAMES
    from ddmi_amd.video import video_generation_coords
    vit = VITAutoencoder(SKY_DD, SKY_EMBED, SKY_FRAMES, numerics="bf16", max_batch=1)
    vit.load_state_dict(named_state("sky.vit", vit.manifest()))
    mlp = MLPVideo(numerics="bf16")
    mlp.load_state_dict(named_state("sky.mlpv", mlp.manifest()))
    z = torch.from_numpy(iw.normal("sky", "z", (1, SKY_EMBED, vit.seq_len))).to(DEV)
    coords = video_generation_coords(256, 16)
    hdbf = vit.decode(z)  # warm-up (autotune, lazy module load)
    torch.cuda.synchronize()
    t0 = time.time()
    hdbf = vit.decode(z)
    torch.cuda.synchronize()
    t1 = time.time()
    vid = mlp(coords, hdbf)
    torch.cuda.synchronize()
    t2 = time.time()
    print("sky decode %.1f ms, mlpvideo %.1f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3))
    assert [tuple(p.shape) for p in hdbf[0]] == [(1, 64, 64, 64), (1, 64, 128, 128), (1, 64, 256, 256)]
    assert [tuple(p.shape) for p in hdbf[1]] == [(1, 64, 16, 64), (1, 64, 16, 128), (1, 64, 16, 256)]
    assert tuple(vid.shape) == (1, 3, 16, 256, 256)
    assert bool(torch.isfinite(vid).all())
    for lst in hdbf:
        for p in lst:
            assert bool(torch.isfinite(p.float()).all())


def test_video_pipeline_matches_oracle_and_is_batch_invariant(golden):
    """ddmi_amd.pipeline.VideoPipeline (tools/ldm/video.py:214-236) on the toy configuration, parity mode: the chain it
    runs equals the oracle's chain fed with the same per-sample generator draws, and a sample does not depend on which
    other samples share its batch."""
    from ddmi_amd.dist import sample_seed
    from ddmi_amd.pipeline import VideoPipeline
    from oracle import ddmi_oracle as O
    cfgs = golden.json("configs_video.json")
    R, Tn = cfgs["TOY_DDV"]["resolution"], cfgs["TOY_FRAMES"]
    T, S = 20, 4
    ddpmconfig = dict(timesteps=T, sampling_timesteps=S, ddim_sampling_eta=0.5, channels=cfgs["TOY_EMBED_V"], domain="video")
    pipe = VideoPipeline(cfgs["TOY_UNET_V"], cfgs["TOY_DDV"], cfgs["MLPV_CFG"], ddpmconfig, cfgs["TOY_EMBED_V"], frames=Tn,
                         test_resolution=R, numerics="fp32", max_batch=2, device=DEV)
    pipe.load_named_seed("toyv").finalize()
    idx = [5, 9]
    vid, z = pipe.generate(idx, base_seed=3, return_latent=True)
    assert tuple(vid.shape) == (2, 3, Tn, R, R)
    # replay the draws: x_T (stream 0xFFFFFFFF), then stream n for the n-th step, from the per-chain Philox seeds
    shape = (2, cfgs["TOY_EMBED_V"], pipe.unet.seq_len)
    seeds = torch.tensor([sample_seed(3, i) for i in idx], dtype=torch.int64, device=DEV)
    draws = [DDPM.randn_from_seeds(shape, seeds, DDPM.X_T_TAG).cpu()] + [DDPM.randn_from_seeds(shape, seeds, n).cpu() for n in range(S - 1)]
    st = lambda tag, man: O.to_state(named_state(tag, man))
    logit = torch.from_numpy(iw.fill_state("toyv.ddpm", [("mixing_logit", (1, cfgs["TOY_EMBED_V"], 1))])["mixing_logit"])
    ref = O.generate_video(st("toyv.unet3", pipe.unet.manifest()), cfgs["TOY_UNET_V"], st("toyv.vit", pipe.vae.manifest()),
                           cfgs["TOY_DDV"], Tn, st("toyv.mlpv", pipe.mlp.manifest()), O.make_schedule(T), logit, draws[0],
                           draws[1:], T, S, 0.5, O.video_coords(R, Tn))
    assert rel_err(z, ref["z"]) < TOL_FP32
    assert rel_err(vid, ref["video"]) < TOL_FP32
    solo = pipe.generate([9], base_seed=3)
    assert rel_err(solo[0], vid[1]) < 1e-5
