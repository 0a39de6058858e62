"""Per-kernel parity on the GPU, through the C ABI's diagnostic op entry points, against the CPU oracle / torch fp32
and the golden vectors captured from the reference (K1, U1)."""
import ctypes as C
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from ddmi_amd import _lib, initweights as iw
from ddmi_amd.ddpm import DDPM
from helpers import TOL_BF16, TOL_FP32, rel_err
from oracle import ddmi_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _dev(a):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32).to(DEV).contiguous()


def conv_op(x, w, b, stride=1, ups=0, numerics="fp32", splitk=0):
    L = _lib.lib()
    _lib.init_device(0)
    B, Cin, H, W = x.shape
    Cout, k = w.shape[0], w.shape[2]
    hin, win = (2 * H, 2 * W) if ups else (H, W)
    Ho, Wo = (hin + 2 * (k // 2) - k) // stride + 1, (win + 2 * (k // 2) - k) // stride + 1
    xd, wd, bd = _dev(x), _dev(w), _dev(b)
    y = torch.empty((B, Cout, Ho, Wo), device=DEV, dtype=torch.float32)
    _lib.check(L.ddmi_op_conv2d(xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), y.data_ptr(), B, Cin, H, W, Cout, k, stride,
                                ups, _lib.numerics_code(numerics), splitk))
    return y.cpu()


CONV_CASES = [
    # B, Cin, H, W, Cout, k, stride, ups
    (2, 64, 8, 8, 224, 3, 1, 0),      # conv_in-like, BN=112 tiles
    (1, 224, 16, 16, 224, 3, 1, 0),   # K = 9*224: 63 chunks (odd -> zero-filled tail stage in bf16)
    (2, 96, 8, 8, 192, 3, 2, 0),      # Downsample (stride 2), Cout not a tile multiple
    (2, 64, 8, 8, 64, 3, 1, 1),       # Upsample: nearest x2 fused into the gather
    (2, 448, 8, 8, 224, 1, 1, 0),     # 1x1 skip connection
    (1, 128, 32, 32, 128, 3, 1, 0),   # decoder-like, 128x128 tiles
    (3, 16, 5, 7, 32, 3, 1, 0),       # ragged: Cin < 32 (padded chunk), odd spatial size, M tail
    (1, 512, 8, 8, 512, 3, 1, 0),     # deep K, small M -> split-K picked automatically
]


@pytest.mark.parametrize("case", CONV_CASES)
@pytest.mark.parametrize("numerics", ["fp32", "bf16"])
def test_conv2d(case, numerics):
    B, Cin, H, W, Cout, k, stride, ups = case
    tag = "conv%s" % (case,)
    x = iw.normal(tag, "x", (B, Cin, H, W))
    w = iw.normal(tag, "w", (Cout, Cin, k, k), 0.0, 1.0 / math.sqrt(Cin * k * k))
    b = iw.normal(tag, "b", (Cout,), 0.0, 0.1)
    xt = torch.from_numpy(x)
    if ups:
        xt = F.interpolate(xt, scale_factor=2, mode="nearest")
    ref = F.conv2d(xt, torch.from_numpy(w), torch.from_numpy(b), stride=stride, padding=k // 2)
    got = conv_op(x, w, b, stride, ups, numerics)
    assert got.shape == ref.shape
    tol = 2e-5 if numerics == "fp32" else 1.5e-2
    assert rel_err(got, ref) < tol


@pytest.mark.parametrize("splitk", [2, 3, 8])
def test_conv2d_forced_splitk(splitk):
    x = iw.normal("sk", "x", (1, 224, 8, 8))
    w = iw.normal("sk", "w", (224, 224, 3, 3), 0.0, 0.02)
    b = iw.normal("sk", "b", (224,))
    ref = F.conv2d(torch.from_numpy(x), torch.from_numpy(w), torch.from_numpy(b), padding=1)
    assert rel_err(conv_op(x, w, b, numerics="fp32", splitk=splitk), ref) < 2e-5
    assert rel_err(conv_op(x, w, b, numerics="bf16", splitk=splitk), ref) < 1.5e-2


def test_conv2d_is_linear_at_full_size():
    """Size-independent property at an AFHQ shape (224 -> 224 @ 64x64, B=2): conv(a*x1 + x2) = a*conv(x1) + conv(x2)
    with zero bias, and a full comparison against torch CPU."""
    x1 = iw.normal("lin", "x1", (2, 224, 64, 64))
    x2 = iw.normal("lin", "x2", (2, 224, 64, 64))
    w = iw.normal("lin", "w", (224, 224, 3, 3), 0.0, 0.02)
    z = np.zeros((224,), np.float32)
    y1, y2 = conv_op(x1, w, z), conv_op(x2, w, z)
    y12 = conv_op(0.5 * x1 + x2, w, z)
    assert rel_err(y12, 0.5 * y1 + y2) < 1e-5
    ref = F.conv2d(torch.from_numpy(x1), torch.from_numpy(w), None, padding=1)
    assert rel_err(y1, ref) < 2e-5
    assert rel_err(conv_op(x1, w, z, numerics="bf16"), ref) < 1.5e-2


# every bf16 kernel variant the autotuner may pick (k_conv.hip kRing ids: ring 0-12, pipe 13-18, halo 19-20), pinned
# one at a time on shapes that cover: halo-eligible W = 16 / 32 / 64, a Cin that is not a multiple of 64 (half-filled
# last stage), a Cout tail, several samples per launch and split-K over whole channel blocks.
VARIANT_CASES = [
    (2, 224, 16, 16, 224, 3, 1, 0),
    (1, 96, 32, 32, 160, 3, 1, 0),
    (1, 192, 64, 64, 224, 3, 1, 0),
    (2, 448, 8, 8, 224, 1, 1, 0),
    (5, 128, 8, 8, 224, 3, 1, 0),     # 8x8 level: halo tiles of four whole samples, last tile runs past the batch
    (3, 64, 8, 16, 96, 3, 1, 0),      # non-square plane (video xt / yt), two samples per tile
]
_variant_cache = {}


def _variant_case(case):
    if case not in _variant_cache:
        B, Cin, H, W, Cout, k, stride, ups = case
        tag = "convv%s" % (case,)
        x = iw.normal(tag, "x", (B, Cin, H, W))
        w = iw.normal(tag, "w", (Cout, Cin, k, k), 0.0, 1.0 / math.sqrt(Cin * k * k))
        b = iw.normal(tag, "b", (Cout,), 0.0, 0.1)
        ref = F.conv2d(torch.from_numpy(x), torch.from_numpy(w), torch.from_numpy(b), stride=stride, padding=k // 2)
        _variant_cache[case] = (x, w, b, ref)
    return _variant_cache[case]


@pytest.mark.parametrize("cfg", list(range(29)))
def test_conv2d_every_bf16_variant(cfg):
    L = _lib.lib()
    try:
        for case in VARIANT_CASES:
            B, Cin, H, W, Cout, k, stride, ups = case
            x, w, b, ref = _variant_case(case)
            for splitk in (0, 2):
                _lib.check(L.ddmi_debug_conv_config(cfg))
                got = conv_op(x, w, b, stride, ups, "bf16", splitk)
                assert rel_err(got, ref) < 1.5e-2, (cfg, case, splitk)
                # a few corrupted pixels or channels hide in an RMS figure; bound the worst element as well
                assert float((got - ref).abs().max()) < 0.03 * float(ref.abs().max()), (cfg, case, splitk)
    finally:
        L.ddmi_debug_conv_config(-1)


UPSAMPLE_CASES = [(2, 64, 8, 8, 64), (1, 224, 16, 16, 224), (3, 32, 5, 7, 96), (2, 448, 32, 32, 448), (5, 128, 4, 4, 224),
                  (5, 72, 8, 8, 224), (2, 128, 64, 64, 128), (3, 64, 8, 16, 96)]   # 2x2 halo kernel: sample-packed tiles past the batch end, ragged channel block, W = 64, non-square


@pytest.mark.parametrize("case", UPSAMPLE_CASES)
def test_conv2d_upsample_parity_form(case):
    """bf16: nearest x2 upsample + 3x3 conv runs as four 2x2 parity convolutions on the source grid with pre-summed
    weights (k_conv.hip conv_out_row) - borders, odd sizes, M tails, on the ring, pipe, small-tile and (cfg 19 / 20) 2x2 halo kernels."""
    B, Cin, H, W, Cout = case
    tag = "convup%s" % (case,)
    x = iw.normal(tag, "x", (B, Cin, H, W))
    w = iw.normal(tag, "w", (Cout, Cin, 3, 3), 0.0, 1.0 / math.sqrt(Cin * 9))
    b = iw.normal(tag, "b", (Cout,), 0.0, 0.1)
    ref = F.conv2d(F.interpolate(torch.from_numpy(x), scale_factor=2, mode="nearest"), torch.from_numpy(w),
                   torch.from_numpy(b), padding=1)
    L = _lib.lib()
    try:
        for cfg in (-1, 7, 11, 13, 16, 17, 19, 20):
            _lib.check(L.ddmi_debug_conv_config(cfg))
            got = conv_op(x, w, b, 1, 1, "bf16")
            assert got.shape == ref.shape
            assert rel_err(got, ref) < 1.5e-2, (cfg, case)
            assert float((got - ref).abs().max()) < 0.03 * float(ref.abs().max()), (cfg, case)
    finally:
        L.ddmi_debug_conv_config(-1)
    assert rel_err(conv_op(x, w, b, 1, 1, "fp32"), ref) < 2e-5   # the parity path stays on the plain fused-upsample form


def test_philox_noise_streams():
    """ddmi_randn_fill / ddmi_ddpm_step_rng: per-chain counter-based normals.  Moments, independence across chains and
    steps, determinism, independence of the batch composition, and the step kernel drawing exactly the fill's values."""
    L = _lib.lib()
    _lib.init_device(0)
    B, Cc, HW = 6, 64, 64 * 64
    seeds = torch.tensor([(777 << 32) + i for i in (0, 1, 2, 3, 4, 12345)], dtype=torch.int64, device=DEV)
    a = DDPM.randn_from_seeds((B, Cc, HW), seeds, 7)
    z = a.double().cpu()
    n = z[0].numel()
    assert abs(float(z.mean())) < 3e-3 and abs(float(z.var()) - 1.0) < 5e-3
    assert abs(float((z ** 4).mean()) - 3.0) < 5e-2 and float(z.abs().max()) < 6.5 and bool(torch.isfinite(a).all())
    flat = z.reshape(B, -1)
    corr = (flat @ flat.T) / n
    assert float((corr - torch.eye(B, dtype=torch.float64)).abs().max()) < 1e-2          # chains are independent
    b = DDPM.randn_from_seeds((B, Cc, HW), seeds, 8).double().cpu()
    assert abs(float((z * b).mean())) < 3e-3 and not torch.equal(z, b)                   # steps are independent
    assert abs(float((flat[:, :-1] * flat[:, 1:]).mean())) < 3e-3                        # neighbours are independent
    assert torch.equal(a, DDPM.randn_from_seeds((B, Cc, HW), seeds, 7))                  # deterministic
    solo = DDPM.randn_from_seeds((1, Cc, HW), seeds[5:6].contiguous(), 7)
    assert torch.equal(solo[0], a[5])                                                    # batch composition is irrelevant
    # the fused step draws the same numbers in registers
    x = torch.randn((B, Cc, 64, 64), device=DEV)
    mo = torch.randn((B, Cc, 64, 64), device=DEV)
    d = DDPM(model=None, timesteps=250, sampling_timesteps=250, channels=Cc)
    out_rng, out_ref = torch.empty_like(x), torch.empty_like(x)
    d._step(x, mo, None, d.ancestral_coefs(100), _lib.STEP_ANCESTRAL, out_rng, False, seeds, 7)
    d._step(x, mo, a.reshape(x.shape).contiguous(), d.ancestral_coefs(100), _lib.STEP_ANCESTRAL, out_ref, False)
    assert torch.equal(out_rng, out_ref)


def gn_op(x, g, b, eps, silu, numerics):
    L = _lib.lib()
    _lib.init_device(0)
    B, Cc = x.shape[:2]
    HW = int(np.prod(x.shape[2:]))
    xd, gd, bd = _dev(x), _dev(g), _dev(b)
    y = torch.empty_like(xd)
    _lib.check(L.ddmi_op_groupnorm(xd.data_ptr(), gd.data_ptr(), bd.data_ptr(), y.data_ptr(), B, Cc, HW, eps, silu,
                                   _lib.numerics_code(numerics)))
    return y.cpu()


def test_groupnorm_golden(golden):
    u1 = golden.npz("U1_unet_ops.npz")
    man = golden.json("manifests.json")["U1.gn"]
    p = iw.fill_state("U1.gn", [(k, tuple(v)) for k, v in man.items()])
    assert rel_err(gn_op(u1["x"], p["weight"], p["bias"], 1e-5, 0, "fp32"), u1["gn"]) < 1e-5
    assert rel_err(gn_op(u1["x"], p["weight"], p["bias"], 1e-5, 1, "fp32"), u1["gn_silu"]) < 1e-5
    assert rel_err(gn_op(u1["x"], p["weight"], p["bias"], 1e-5, 1, "bf16"), u1["gn_silu"]) < 1.5e-2


@pytest.mark.parametrize("shape", [(2, 224, 64, 64), (1, 2688, 8, 8), (3, 32, 5, 7), (2, 128, 16, 16)])
def test_groupnorm_shapes(shape):
    x = iw.normal("gn%s" % (shape,), "x", shape, 3.0, 2.0)  # large mean: exercises E[x^2]-E[x]^2 cancellation
    g = iw.normal("gn", "g", (shape[1],), 1.0, 0.2)
    b = iw.normal("gn", "b", (shape[1],), 0.0, 0.2)
    ref = O.silu(O.group_norm_explicit(torch.from_numpy(x), torch.from_numpy(g), torch.from_numpy(b), 1e-6))
    assert rel_err(gn_op(x, g, b, 1e-6, 1, "fp32"), ref) < 2e-5


def attn_op(qkv, heads, numerics):
    L = _lib.lib()
    _lib.init_device(0)
    B, _, T = qkv.shape
    qd = _dev(qkv)
    out = torch.empty((B, heads * 32, T), device=DEV, dtype=torch.float32)
    _lib.check(L.ddmi_op_attention_legacy(qd.data_ptr(), out.data_ptr(), B, heads, T, _lib.numerics_code(numerics)))
    return out.cpu()


def attn_mh_op(q, k, v, heads, seg0, seg1, numerics):
    L = _lib.lib()
    _lib.init_device(0)
    B, C, N = q.shape
    qd, kd, vd = _dev(q), _dev(k), _dev(v)
    out = torch.empty((B, C, N), device=DEV, dtype=torch.float32)
    _lib.check(L.ddmi_op_attention_mh(qd.data_ptr(), kd.data_ptr(), vd.data_ptr(), out.data_ptr(), B, heads, C // heads, N,
                                      seg0, seg1, _lib.numerics_code(numerics)))
    return out.cpu()


# (B, heads, d, N, seg0, seg1): every head-dim template of the kernel (32 / 64 / 128 / 256 / 512) incl. the padded
# dims the video UNet produces (16, 96) and the toy dims that take the element-wise path (2, 4); token counts that are
# not tile multiples; one, two and three row segments with boundaries inside a key tile
ATTN_MH_CASES = [
    (2, 16, 2, 128, 64, 32), (1, 16, 4, 48, 16, 16), (2, 4, 16, 200, 200, 0), (1, 3, 32, 130, 100, 30),
    (1, 2, 64, 333, 111, 111), (1, 2, 96, 96, 64, 16), (1, 2, 128, 257, 257, 0), (1, 1, 256, 150, 50, 50),
    (1, 2, 512, 192, 64, 64),
]


@pytest.mark.parametrize("case", ATTN_MH_CASES)
def test_attention_multihead_segments(case):
    B, heads, d, N, seg0, seg1 = case
    tag = "attmh%s" % (case,)
    q = iw.normal(tag, "q", (B, heads * d, N), 0.0, 1.2)
    k = iw.normal(tag, "k", (B, heads * d, N), 0.0, 1.2)
    v = iw.normal(tag, "v", (B, heads * d, N))
    q[0, 0:d, N // 3] *= 3.0  # running-max jump mid-sequence
    ref = O.sdpa_heads(torch.from_numpy(q), torch.from_numpy(k), torch.from_numpy(v), heads)
    assert rel_err(attn_mh_op(q, k, v, heads, seg0, seg1, "fp32"), ref) < 2e-5
    assert rel_err(attn_mh_op(q, k, v, heads, seg0, seg1, "bf16"), ref) < 8e-2


@pytest.mark.parametrize("d", [16, 64])
def test_attention_legacy_other_head_dims(d):
    heads, T = 3, 150
    qkv = iw.normal("attl%d" % d, "qkv", (2, heads * 3 * d, T), 0.0, 1.2)
    ref = O.qkv_attention_legacy(torch.from_numpy(qkv), heads)
    L = _lib.lib()
    for numerics, tol in (("fp32", 2e-5), ("bf16", 8e-2)):
        out = torch.empty((2, heads * d, T), device=DEV, dtype=torch.float32)
        _lib.check(L.ddmi_op_attention_legacy_hd(_dev(qkv).data_ptr(), out.data_ptr(), 2, heads, d, T, _lib.numerics_code(numerics)))
        assert rel_err(out.cpu(), ref) < tol


def test_attention_golden(golden):
    u1 = golden.npz("U1_unet_ops.npz")
    assert rel_err(attn_op(u1["qkv"], 2, "fp32"), u1["attn_legacy"]) < 2e-5
    assert rel_err(attn_op(u1["qkv"], 2, "bf16"), u1["attn_legacy"]) < 2e-2


@pytest.mark.parametrize("B,heads,T", [(1, 14, 1024), (2, 3, 256), (1, 2, 16), (1, 1, 100)])
def test_attention_shapes(B, heads, T):
    qkv = iw.normal("att%d_%d_%d" % (B, heads, T), "qkv", (B, heads * 96, T), 0.0, 1.5)
    # spike one key against one query so the running max jumps mid-sequence (online-softmax rescale path)
    qkv[0, 0:32, T // 3] *= 4.0
    qkv[0, 32:64, (2 * T) // 3] = qkv[0, 0:32, T // 3] * 2.0
    ref = O.qkv_attention_legacy(torch.from_numpy(qkv), heads)
    assert rel_err(attn_op(qkv, heads, "fp32"), ref) < 2e-5
    assert rel_err(attn_op(qkv, heads, "bf16"), ref) < 8e-2  # logits of +-30 with 8-bit mantissas


def test_timestep_embedding_golden(golden):
    u1 = golden.npz("U1_unet_ops.npz")
    L = _lib.lib()
    _lib.init_device(0)
    t = torch.tensor([0, 1, 500, 999], dtype=torch.int64, device=DEV)
    for dim in (224, 32):
        out = torch.empty((4, dim), device=DEV, dtype=torch.float32)
        _lib.check(L.ddmi_op_timestep_embedding(t.data_ptr(), out.data_ptr(), 4, dim))
        # a 1-ulp difference of exp() in the frequency table moves cos/sin(999 * f) by up to ~1e-4
        assert float((out.cpu() - torch.from_numpy(u1[f"temb{dim}"])).abs().max()) < 2e-4


# ---- fused sampler step (K1) -----------------------------------------------------------------------------------
class FixedModel:
    def __init__(self, out):
        self.out = out

    def __call__(self, x, t, cond=None, out=None, **kw):
        out.copy_(self.out)
        return out

    def load_state_dict(self, *a, **k):
        return self


def test_ddpm_step_golden(golden):
    k1 = golden.npz("K1_step.npz")
    x, mo, nz = _dev(k1["x"]), _dev(k1["model_out"]), _dev(k1["noise"])
    for T in (1000, 250):
        d = DDPM(model=FixedModel(mo), timesteps=T, sampling_timesteps=T, channels=8)
        d.load_state_dict({"mixing_logit": k1["mixing_logit"]}, strict=False)
        for t in (0, 1, T // 2, T - 2, T - 1):
            out = torch.empty_like(x)
            d._step(x, mo, nz, d.ancestral_coefs(t), _lib.STEP_ANCESTRAL, out, False)
            assert rel_err(out, k1[f"anc.T{T}.t{t}"]) < 2e-6, (T, t)
    d = DDPM(model=FixedModel(mo), timesteps=1000, sampling_timesteps=1000, channels=8)
    d.load_state_dict({"mixing_logit": k1["mixing_logit"]}, strict=False)
    out = torch.empty_like(x)
    d._step(x, mo, nz, d.ancestral_coefs(500), _lib.STEP_ANCESTRAL, out, True)
    assert rel_err(out, k1["anc_clip.T1000.t500"]) < 2e-6
    for eta in (0.0, 1.0, 0.5):
        d = DDPM(model=FixedModel(mo), timesteps=1000, sampling_timesteps=3, ddim_sampling_eta=eta, channels=8)
        d.load_state_dict({"mixing_logit": k1["mixing_logit"]}, strict=False)
        out = d.sample(list(x.shape), noise=k1["x"], step_noise=[k1["ddim_n1"], k1["ddim_n2"]], device=DEV)
        assert rel_err(out, k1[f"ddim3.eta{eta}"]) < 2e-6, eta


def test_ddpm_step_full_size_matches_oracle():
    """AFHQ latent shape [8,64,64,64]; x0 branch and x_{t-1} against the oracle on the same inputs."""
    shape = (8, 64, 64, 64)
    x, mo, nz = (iw.normal("stepfull", k, shape) for k in ("x", "mo", "nz"))
    logit = iw.normal("stepfull", "logit", (1, 64, 1, 1))
    sched = O.make_schedule(250)
    d = DDPM(model=None, timesteps=250, sampling_timesteps=250, channels=64)
    d.load_state_dict({"mixing_logit": logit}, strict=False)
    out = torch.empty(shape, device=DEV)
    for t in (249, 17, 0):
        d._step(_dev(x), _dev(mo), _dev(nz), d.ancestral_coefs(t), _lib.STEP_ANCESTRAL, out, False)
        ref = O.p_sample_step(sched, torch.from_numpy(x), torch.from_numpy(mo), t, torch.from_numpy(nz), torch.from_numpy(logit))
        assert rel_err(out, ref) < 2e-6
