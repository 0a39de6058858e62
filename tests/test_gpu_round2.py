"""Round-2 parity tests on the GPU (through the C ABI):
  * BASELINE config 1 (fixture G2 captured from the reference) and the same-seed draw order (G3) via rng='reference';
  * the configuration bench.py times - bf16, max_batch = 512 - pinned row by row against B = 1 (the > 2 GiB descriptor
    rebasing of the conv kernels and the B = 512 plans), fp32 at B = 64, decoder + MLP at the bench batch;
  * NoiseInjection drawn inside the MLP kernel == the explicit draws; refusal instead of silently dropping the term;
  * a reference-layout `ldm-last.pt` written with torch.save -> load_checkpoint -> HIP path == oracle on the same tensors;
  * the RCCL (`nccl`) path of ddmi_amd/dist.py."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from ddmi_amd import _lib
from ddmi_amd import initweights as iw
from ddmi_amd.autoencoder import Autoencoder
from ddmi_amd.coords import generation_coords, get_scale_injection
from ddmi_amd.ddpm import DDPM
from ddmi_amd.mlp import MLP
from ddmi_amd.pipeline import ImagePipeline
from ddmi_amd.unet import UNetModel
from helpers import AFHQ_DD, AFHQ_EMBED, AFHQ_UNET, MLP_CFG, TOL_BF16, TOL_FP32, assert_bf16_close, named_state, rel_err
from oracle import ddmi_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INJECT_TAG = 0xFFFE0000  # philox.h PHILOX_TAG_INJECT


# ---------------- G2: BASELINE config 1 -------------------------------------------------------------------------------
@pytest.mark.parametrize("numerics", ["fp32", "bf16x3"])
@pytest.mark.parametrize("name,T,S,eta", [("ddim", 1000, 10, 0.0), ("anc", 10, 10, 1.0)])
def test_config1_g2(golden, name, T, S, eta, numerics):
    """AFHQ full-width nets on a [1,64,16,16] latent, 10 steps, R = 64 (si = 4), torch.manual_seed(777): the reference's
    generate() (tools/ldm/image.py:225-249) captured by oracle/gen_golden_g2.py; both parity modes (fp32 MFMA, and bf16x3 = every
    convolution as three bf16 MFMA products) < 1e-3."""
    g2 = golden.npz("G2_config1.npz")
    ddpmconfig = dict(timesteps=T, sampling_timesteps=S, ddim_sampling_eta=eta, channels=64, image_size=16)
    pipe = ImagePipeline(AFHQ_UNET, AFHQ_DD, MLP_CFG, ddpmconfig, AFHQ_EMBED, test_resolution=64, numerics=numerics,
                         max_batch=1, device=DEV)
    pipe.load_named_seed("afhq").finalize()
    img, z = pipe.generate([0], base_seed=int(g2["seed"]), rng="reference", return_latent=True)
    assert z.shape == (1, 64, 16, 16) and img.shape == (1, 3, 64, 64)
    assert rel_err(z, g2[f"{name}.z"]) < TOL_FP32
    planes = pipe.vae.decode(z, reference_layout=True)
    for i, pl in enumerate(planes):
        a = pl.cpu().numpy()
        assert list(a.shape) == list(g2[f"{name}.plane{i}.shape"])
        assert rel_err(a[0, :4, :16, :16], g2[f"{name}.plane{i}.slice"]) < TOL_FP32
        assert abs(a.astype(np.float64).sum() - float(g2[f"{name}.plane{i}.sum"])) < TOL_FP32 * float(g2[f"{name}.plane{i}.abssum"])
    assert rel_err(img, g2[f"{name}.img"]) < TOL_FP32
    raw = pipe.mlp(pipe.coords(), hdbf=pipe.vae.decode(z), si=get_scale_injection(64))
    assert rel_err(raw, g2[f"{name}.raw"]) < TOL_FP32


# ---------------- G3: same seed -> same samples as the reference's CPU path ---------------------------------------------
def _toy_pipe(golden, numerics, T, S, eta, max_batch=2, noisy_mlp=True):
    cfgs = golden.json("configs.json")
    ddpmconfig = dict(timesteps=T, sampling_timesteps=S, ddim_sampling_eta=eta, channels=64, image_size=8)
    pipe = ImagePipeline(cfgs["TOY_UNET"], cfgs["TOY_DD"], cfgs["MLP_CFG"], ddpmconfig, cfgs["TOY_EMBED"], test_resolution=32,
                         numerics=numerics, max_batch=max_batch, device=DEV)
    pipe.unet.load_state_dict(named_state("U2a.unet", pipe.unet.manifest()))
    pipe.vae.load_state_dict(named_state("D1.vae", pipe.vae.manifest()))
    mst = named_state("M1n.mlp" if noisy_mlp else "M1.mlp", pipe.mlp.manifest())
    if noisy_mlp:
        for k in mst:
            if k.endswith("noise.weight"):
                mst[k] = iw.normal("M1n.mlp", k, mst[k].shape, 0.0, 0.1)
    pipe.mlp.load_state_dict(mst)
    pipe.ddpm.load_state_dict({"mixing_logit": iw.fill_state("G1.ddpm", [("mixing_logit", (1, 64, 1, 1))])["mixing_logit"]}, strict=False)
    return pipe.finalize(), mst


@pytest.mark.parametrize("name,T,S,eta", [("ddim_eta0", 1000, 10, 0.0), ("ddim_eta1", 1000, 10, 1.0), ("anc", 10, 10, 1.0)])
def test_same_seed_reference_rng_g3(golden, name, T, S, eta):
    """rng='reference': torch.manual_seed(777) + CPU randn in the reference's order (x_T, per step, 12 NoiseInjection
    draws) reproduces the reference's own generate() with nothing patched and non-zero noise.weight (a24)."""
    g3 = golden.npz("G3_same_seed.npz")
    pipe, _ = _toy_pipe(golden, "fp32", T, S, eta)
    assert pipe.mlp.noise_nonzero
    img, z = pipe.generate([0, 1], base_seed=int(g3["seed"]), rng="reference", return_latent=True)
    assert rel_err(z, g3[f"{name}.z"]) < TOL_FP32
    assert rel_err(img, g3[f"{name}.img"]) < TOL_FP32
    again = pipe.generate([0, 1], base_seed=int(g3["seed"]), rng="reference")
    assert torch.equal(again, img)                       # reseeded: the same samples
    cont = pipe.generate([0, 1], base_seed=int(g3["seed"]), rng="reference", reseed=False)
    assert not torch.equal(cont, img)                    # continuing the stream: the next samples


# ---------------- NoiseInjection inside the kernel ---------------------------------------------------------------------
@pytest.mark.parametrize("numerics", ["fp32", "bf16"])
def test_mlp_noise_drawn_in_kernel_equals_explicit_draws(golden, numerics):
    d1 = golden.npz("D1_decoder.npz")
    man = golden.json("manifests.json")["M1n.mlp"]
    mlp = MLP(**MLP_CFG, numerics=numerics)
    st = named_state("M1n.mlp", mlp.manifest())
    for k in st:
        if k.endswith("noise.weight"):
            st[k] = iw.normal("M1n.mlp", k, tuple(man[k]), 0.0, 0.1)
    mlp.load_state_dict(st)
    assert mlp.noise_nonzero
    planes = [torch.from_numpy(d1[f"plane{i}"]).to(DEV) for i in range(3)]
    B, R = 2, 48                                          # 2304 points per sample: tiles straddle the sample boundary
    coords = generation_coords(R)
    seeds = torch.tensor([(5 << 32) + 17, (5 << 32) + 3], dtype=torch.int64, device=DEV)
    draws = torch.stack([DDPM.randn_from_seeds((B, R * R), seeds, INJECT_TAG + c) for c in range(12)])   # [12,B,N]
    si = get_scale_injection(R)
    explicit = mlp(coords, hdbf=planes, si=si, noise=draws)
    drawn = mlp(coords, hdbf=planes, si=si, seeds=seeds)
    assert torch.equal(drawn, explicit)
    quiet = mlp(coords, hdbf=planes, si=si, noise=torch.zeros_like(draws))
    assert rel_err(drawn, quiet) > 1e-3                   # the term is really there
    ref = O.image_mlp(O.to_state(st), coords, [p.cpu() for p in planes], si, noise=draws.cpu())
    assert rel_err(drawn, ref) < (TOL_FP32 if numerics == "fp32" else TOL_BF16)
    solo = mlp(coords, hdbf=[p[1:2].contiguous() for p in planes], si=si, seeds=seeds[1:2])
    assert torch.equal(solo[0], drawn[1])                 # a sample's noise does not depend on its batch
    # the C ABI refuses to drop the term silently
    import ctypes as C
    pts = coords[0].permute(1, 2, 0).reshape(R * R, 2).to(DEV).contiguous()
    out = torch.empty((B, 3, R, R), device=DEV)
    pl = [p.float().contiguous() for p in planes]
    ptrs = (C.c_void_p * 3)(*[p.data_ptr() for p in pl])
    hw = (C.c_int * 3)(*[int(p.shape[2]) for p in pl])
    rc = _lib.lib().ddmi_hdpe_mlp2d(mlp._handle, ptrs, hw, _lib.LAYOUT_NCHW_F32, pts.data_ptr(), float(si), None,
                                    out.data_ptr(), B, R * R, _lib.stream_ptr())
    assert rc != 0 and b"NoiseInjection" in _lib.lib().ddmi_last_error()
    # default call (no noise, no seeds): fresh noise per call like new_empty(...).normal_(), never the noise-free image
    a, b = mlp(coords, hdbf=planes, si=si), mlp(coords, hdbf=planes, si=si)
    assert not torch.equal(a, b) and rel_err(a, quiet) > 1e-3


def test_pipeline_philox_noise_injection_is_batch_invariant(golden):
    pipe, _ = _toy_pipe(golden, "fp32", 6, 6, 1.0, max_batch=3)
    img = pipe.generate([4, 0, 11], base_seed=5)
    solo = pipe.generate([11], base_seed=5)
    assert rel_err(solo[0], img[2]) < 1e-5


# ---------------- the benched configuration: bf16, max_batch = 512 -------------------------------------------------------
def _rows_batch(vec, B, rows, tag):
    """[B, ...] of fresh normals with the test vector at `rows`."""
    g = torch.Generator(device=DEV).manual_seed(1234)
    x = torch.randn((B,) + tuple(vec.shape[1:]), device=DEV, generator=g)
    for r in rows:
        x[r] = vec[0]
    return x


def test_unet_bf16_b512_rows_match_b1_and_oracle():
    """bench.py's UNet: bf16, max_batch = 512 (10 GB arena, activations up to 2.8 GB: buffer-descriptor rebasing past
    2 GiB, XCD tile maps, the B = 512 autotuned plans).  Rows 0, 255 and 511 carry the B = 1 test vector: each must equal
    the B = 1 network's result (bf16 tolerance - other tiles) and the CPU oracle; the rows must be bit-identical to each
    other (same kernels, same plans, only the address differs)."""
    B, rows = 512, (0, 255, 511)
    vec = torch.from_numpy(iw.normal("afhq", "x", (1, 64, 64, 64))).to(DEV)
    t1 = torch.tensor([437], dtype=torch.int64, device=DEV)
    st = named_state("afhq.unet", UNetModel(**AFHQ_UNET, numerics="bf16", max_batch=1).manifest())
    small = UNetModel(**AFHQ_UNET, numerics="bf16", max_batch=1)
    small.load_state_dict(st)
    eps1 = small(vec, t1).clone()
    del small
    torch.cuda.empty_cache()
    big = UNetModel(**AFHQ_UNET, numerics="bf16", max_batch=B)
    big.load_state_dict(st)
    x = _rows_batch(vec, B, rows, "unet")
    eps = big(x, t1.expand(B).contiguous())
    assert big.memory()["arena"] > (2 << 30)
    assert bool(torch.isfinite(eps).all())
    for r in rows:
        assert rel_err(eps[r], eps1[0]) < TOL_BF16, r
        assert torch.equal(eps[r], eps[rows[0]]), r
    assert not torch.equal(eps[1], eps[0])
    ref = O.unet_forward(O.to_state(st), AFHQ_UNET, vec.cpu(), t1.cpu())
    assert_bf16_close(eps[511:512], ref, "AFHQ UNet, row 511 of the B = 512 batch")
    # uniform-t fast path used by the sampling loops gives the same rows
    big.set_uniform_t(True)
    eps_u = big(x, t1.expand(B).contiguous())
    big.set_uniform_t(False)
    assert torch.equal(eps_u, eps)


def test_unet_fp32_b64_rows_match_b1():
    B, rows = 64, (0, 31, 63)
    vec = torch.from_numpy(iw.normal("afhq", "x", (1, 64, 64, 64))).to(DEV)
    t1 = torch.tensor([437], dtype=torch.int64, device=DEV)
    st = named_state("afhq.unet", UNetModel(**AFHQ_UNET, numerics="fp32", max_batch=1).manifest())
    small = UNetModel(**AFHQ_UNET, numerics="fp32", max_batch=1)
    small.load_state_dict(st)
    eps1 = small(vec, t1).clone()
    del small
    torch.cuda.empty_cache()
    big = UNetModel(**AFHQ_UNET, numerics="fp32", max_batch=B)
    big.load_state_dict(st)
    eps = big(_rows_batch(vec, B, rows, "unet32"), t1.expand(B).contiguous())
    for r in rows:
        assert rel_err(eps[r], eps1[0]) < 1e-5, r


def test_decoder_and_mlp_bf16_bench_batch_rows_match_b1():
    """bench.py's decode + fused MLP at the bench batch (512 latents at once, planes up to 4.3 GB)."""
    B, rows = 512, (0, 255, 511)
    vec = torch.from_numpy(iw.normal("afhq", "z", (1, 64, 64, 64))).to(DEV)
    small = Autoencoder(AFHQ_DD, AFHQ_EMBED, numerics="bf16", max_batch=1)
    vst = named_state("afhq.vae", small.manifest())
    small.load_state_dict(vst)
    planes1 = [p.clone() for p in small.decode(vec)]
    del small
    torch.cuda.empty_cache()
    big = Autoencoder(AFHQ_DD, AFHQ_EMBED, numerics="bf16", max_batch=B)
    big.load_state_dict(vst)
    planes = big.decode(_rows_batch(vec, B, rows, "vae"))
    for p, p1 in zip(planes, planes1):
        for r in rows:
            assert rel_err(p[r].float(), p1[0].float()) < TOL_BF16, r
            assert torch.equal(p[r], p[rows[0]]), r
    mlp = MLP(**MLP_CFG, numerics="bf16")
    mlp.load_state_dict(named_state("afhq.mlp", mlp.manifest()))
    rgb = mlp(generation_coords(256), hdbf=planes, si=1.0)
    rgb1 = mlp(generation_coords(256), hdbf=[p[511:512].contiguous(memory_format=torch.channels_last) for p in planes], si=1.0)
    assert rgb.shape == (B, 3, 256, 256) and bool(torch.isfinite(rgb).all())
    assert torch.equal(rgb[511], rgb1[0])                 # a point's value does not depend on the launch it is in
    assert torch.equal(rgb[0], rgb[511])


# ---------------- f1: a reference-layout checkpoint -> HIP path == oracle ---------------------------------------------
class TrainerArgs:   # stands in for the argparse/OmegaConf objects a real checkpoint pickles (never on the allow-list)
    def __init__(self):
        self.lr = 1e-4
        self.note = "pickled by tools/ldm/image.py:106"


@pytest.mark.parametrize("mode", ["pretrained", "stage1"])
def test_checkpoint_roundtrip_matches_oracle(golden, tmp_path, mode):
    """torch.save of the dict LDMTrainer.save writes (tools/ldm/image.py:104-119; stage 1: tools/d2c_vae/image.py:120-133)
    -> ImagePipeline.load_checkpoint (restricted unpickler, EMA mapping) -> UNet / decoder / MLP on the GPU == the oracle
    run on the very tensors that were saved."""
    from ddmi_amd import checkpoint as CK
    cfgs = golden.json("configs.json")
    T = 6
    ddpmconfig = dict(timesteps=T, sampling_timesteps=T, channels=64, image_size=8)
    pipe = ImagePipeline(cfgs["TOY_UNET"], cfgs["TOY_DD"], cfgs["MLP_CFG"], ddpmconfig, cfgs["TOY_EMBED"], test_resolution=32,
                         numerics="fp32", max_batch=2, device=DEV)
    tt = lambda d: {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in d.items()}
    unet_st = named_state("ckpt.unet", pipe.unet.manifest())
    vae_st = named_state("ckpt.vae", pipe.vae.manifest())
    vae_full = dict(tt(vae_st), **{"encoder.conv_in.weight": torch.zeros(4, 3, 3, 3), "quant_conv.weight": torch.zeros(2, 2, 1, 1)})
    mlp_st = named_state("ckpt.mlp", pipe.mlp.manifest())
    logit = iw.normal("ckpt", "logit", (1, 64, 1, 1), 0.0, 0.5)
    sched = {k: torch.from_numpy(v) for k, v in O.make_schedule(T).items()}
    ema_weights = dict({"model." + k: v for k, v in tt(unet_st).items()}, mixing_logit=torch.from_numpy(logit), **sched)
    online = {k: v * 0.5 for k, v in ema_weights.items()}       # the non-EMA weights differ: the EMA copy must be picked
    if mode == "pretrained":
        ck = {"args": TrainerArgs(), "step": 2, "current_iters": 10, "vaemodel": vae_full, "mlp": tt(mlp_st), "diffusion": online,
              "dae_opt": {"state": {}, "param_groups": [{"lr": 1e-4}]}, "scaler": None,
              "ema": dict({"ema_model." + k: v for k, v in ema_weights.items()}, **{"online_model." + k: v for k, v in online.items()},
                          initted=torch.tensor(True), step=torch.tensor(10))}
        path = os.path.join(tmp_path, "ldm-last.pt")
    else:
        ck = {"args": TrainerArgs(), "step": 2, "model": vae_full, "mlp": tt(mlp_st), "opt": {"state": {}}}
        path = os.path.join(tmp_path, "model-last.pt")
        pipe.load_named_seed("ck")                              # the diffusion model keeps its own weights in this mode
        unet_st = named_state("ck.unet", pipe.unet.manifest())
        logit = iw.fill_state("ck.ddpm", [("mixing_logit", (1, 64, 1, 1))])["mixing_logit"]
    torch.save(ck, path)
    assert CK.find_checkpoint(str(tmp_path), pretrained=(mode == "pretrained")) == path
    pipe.load_checkpoint(path)
    pipe.finalize()
    shape = (2, 64, 8, 8)
    xT = iw.normal("ck", "xT", shape)
    noises = [iw.normal("ck", f"n{i}", shape) for i in range(T)]
    z = pipe.ddpm.sample(list(shape), noise=xT, step_noise=noises, device=DEV)
    img = pipe.mlp(pipe.coords(), hdbf=pipe.vae.decode(z), si=get_scale_injection(32)).clamp(-1, 1)
    ref = O.generate(O.to_state(unet_st), cfgs["TOY_UNET"], O.to_state(vae_st), cfgs["TOY_DD"], O.to_state(mlp_st),
                     O.make_schedule(T), torch.from_numpy(logit), xT, noises, T, T, 1.0, 32)
    assert rel_err(z, ref["z"]) < TOL_FP32
    assert rel_err(img, ref["img"]) < TOL_FP32


def test_checkpoint_with_other_schedule_is_refused(golden, tmp_path):
    """A checkpoint trained with T = 1000 must not load silently into a T = 6 sampler (derived buffers would disagree)."""
    cfgs = golden.json("configs.json")
    pipe = ImagePipeline(cfgs["TOY_UNET"], cfgs["TOY_DD"], cfgs["MLP_CFG"], dict(timesteps=6, sampling_timesteps=6, channels=64, image_size=8),
                         cfgs["TOY_EMBED"], test_resolution=32, numerics="fp32", max_batch=1, device=DEV)
    tt = lambda d: {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in d.items()}
    sched = {k: torch.from_numpy(v) for k, v in O.make_schedule(1000).items()}
    diff = dict({"model." + k: v for k, v in tt(named_state("ck.unet", pipe.unet.manifest())).items()}, **sched)
    path = os.path.join(tmp_path, "ldm-last.pt")
    torch.save({"vaemodel": tt(named_state("ck.vae", pipe.vae.manifest())), "mlp": tt(named_state("ck.mlp", pipe.mlp.manifest())),
                "diffusion": diff}, path)
    with pytest.raises(ValueError, match="schedule buffer"):
        pipe.load_checkpoint(path)


# ---------------- RCCL --------------------------------------------------------------------------------------------------
def test_nccl_group_of_one_runs_the_collectives(tmp_path):
    """The `nccl` (= RCCL) backend of ddmi_amd/dist.py on this box's GPU: a one-rank group runs the same all-gather /
    all-reduce calls the N-rank job issues (a 1-GPU box cannot host two RCCL ranks: one communicator per device)."""
    script = os.path.join(ROOT, "tests", "_rank_script.py")
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533",
               DDMI_FORCE_GROUP="1")
    subprocess.run([sys.executable, script, str(tmp_path)], check=True, env=env, timeout=300)
    with open(os.path.join(tmp_path, "rank0.json")) as f:
        rec = json.load(f)
    assert rec["backend"] == "nccl" and rec["ranks"] == 1 and rec["rows"] == [0.0, 0.0]


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (the driver's 8-GPU node)")
def test_bench_self_launches_two_nccl_ranks():
    """`python bench.py --gpus 2` exactly as the driver invokes N = 1: becomes the launcher, two RCCL ranks, one JSON line."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--batch", "2", "--nfe", "3", "--steps", "1",
                          "--warmup", "0", "--no-cpu-baseline", "--no-parity-check"], check=True, capture_output=True, text=True,
                         timeout=1200)
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["nccl_ranks"] == 2 and line["config"]["global_batch"] == 4


def test_groupnorm_statistics_from_the_conv_epilogue_match_the_statistics_pass(tmp_path):
    """bf16 networks take a GroupNorm's (sum, sum of squares) from the epilogue of the convolution that produced its input
    (ConvArgs::chstat -> launch_gn_apply_chstat) where every producer can deliver them; DDMI_GN_CHSTAT=0 keeps the separate pass
    over the tensor.  Same full-size AFHQ UNet, same inputs, both ways (the switch is read once per process, so the pass runs in
    a child): the outputs agree far inside the bf16 tolerance, and the statistics traffic really dropped."""
    import os
    import subprocess
    import sys
    code = """
import sys, os, json
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, 'tests'))
import numpy as np, torch
from ddmi_amd import _lib, initweights as iw
from ddmi_amd.unet import UNetModel
from helpers import AFHQ_UNET, named_state
m = UNetModel(**AFHQ_UNET, numerics='bf16', max_batch=16)
m.load_state_dict(named_state('afhq.unet', m.manifest()))
x = torch.from_numpy(iw.normal('afhq', 'x16', (16, 64, 64, 64))).cuda()
t = (torch.arange(16, dtype=torch.int64) * 61 + 5).cuda()
m(x, t)
_lib.profile_enable(1)
y = m(x, t)
torch.cuda.synchronize()
prof = _lib.profile_collect()
np.save(sys.argv[1], y.cpu().numpy())
print('STATS_BYTES', prof.get('gn_stats', {'bytes': 0})['bytes'], prof['gn_apply']['launches'])
# the D2C-VAE decoder: other channel counts (128 / 256 / 512), other tile configurations, eps 1e-6
from ddmi_amd.autoencoder import Autoencoder
from helpers import AFHQ_DD, AFHQ_EMBED
vae = Autoencoder(AFHQ_DD, AFHQ_EMBED, numerics='bf16', max_batch=2)
vae.load_state_dict(named_state('afhq.vae', vae.manifest()))
z = torch.from_numpy(iw.normal('afhq', 'z2', (2, 64, 64, 64))).cuda()
vae.decode(z)
_lib.profile_enable(1)
planes = vae.decode(z)
torch.cuda.synchronize()
prof = _lib.profile_collect()
np.save(sys.argv[1] + '.plane.npy', planes[2].float().cpu().numpy())
print('DEC_STATS_BYTES', prof.get('gn_stats', {'bytes': 0})['bytes'], prof['gn_apply']['launches'])
""" % (ROOT, ROOT)
    outs = {}
    for mode in ("1", "0"):
        path = str(tmp_path / f"eps_{mode}.npy")
        env = dict(os.environ, DDMI_GN_CHSTAT=mode, DDMI_AUTOTUNE="0")
        r = subprocess.run([sys.executable, "-c", code, path], env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout + r.stderr
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("STATS_BYTES")][0].split()
        dline = [ln for ln in r.stdout.splitlines() if ln.startswith("DEC_STATS_BYTES")][0].split()
        outs[mode] = (np.load(path), float(line[1]), int(line[2]), np.load(path + ".plane.npy"), float(dline[1]))
    on, off = outs["1"], outs["0"]
    assert on[2] == off[2]                       # the same number of GroupNorms (apply launches; round 3: a GroupNorm whose producers
    #                                              delivered the sums has NO statistics launch at all - the apply pass reduces them)
    # B = 16: the 64x64 and 32x32 levels run without split-K and deliver their sums (split-K launches finish in another kernel and
    # keep the pass; at B = 512 every level but 8x8 delivers)
    assert on[1] < 0.85 * off[1]
    # two bf16 evaluations whose statistics differ in the last bits drift apart through 60 layers of bf16 roundings (~1e-2 of
    # the output range); what must hold is that NEITHER is further from the fp32 network than the other
    m32 = UNetModel(**AFHQ_UNET, numerics="fp32", max_batch=16)
    m32.load_state_dict(named_state("afhq.unet", m32.manifest()))
    x = torch.from_numpy(iw.normal("afhq", "x16", (16, 64, 64, 64))).to(DEV)
    ref = m32(x, (torch.arange(16, dtype=torch.int64) * 61 + 5).to(DEV)).cpu()
    e_on, e_off = rel_err(torch.from_numpy(on[0]), ref), rel_err(torch.from_numpy(off[0]), ref)
    assert e_on < TOL_BF16 and e_off < TOL_BF16 and e_on < 1.25 * e_off + 1e-3, (e_on, e_off)
    assert rel_err(torch.from_numpy(on[0]), torch.from_numpy(off[0])) < 2e-2
    # decoder: 256x256 plane of both runs against the fp32 decoder
    from ddmi_amd.autoencoder import Autoencoder
    vae32 = Autoencoder(AFHQ_DD, AFHQ_EMBED, numerics="fp32", max_batch=2)
    vae32.load_state_dict(named_state("afhq.vae", vae32.manifest()))
    pref = vae32.decode(torch.from_numpy(iw.normal("afhq", "z2", (2, 64, 64, 64))).to(DEV))[2].float().cpu()
    d_on, d_off = rel_err(torch.from_numpy(on[3]), pref), rel_err(torch.from_numpy(off[3]), pref)
    assert on[4] < 0.6 * off[4]                  # no split-K at these sizes: nearly every decoder GroupNorm takes the sums
    assert d_on < TOL_BF16 and d_off < TOL_BF16 and d_on < 1.25 * d_off + 1e-3, (d_on, d_off)
