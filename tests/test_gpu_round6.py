"""Round-6 GPU tests (through the C ABI): the hardened fused-GroupNorm hand-off (mean-shifted statistics, the library-side fallback,
graph invalidation, two handles on two streams) and the small-batch forward."""
import ctypes as C
import math

import numpy as np
import pytest
import torch

from ddmi_amd import _lib, initweights as iw
from helpers import named_state, rel_err, rel_l2
from test_gpu_round5 import _conv_gn_ref, conv_gn_op

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

SMALL_UNET = dict(image_size=32, in_channels=16, model_channels=128, out_channels=16, attention_resolutions=[], num_res_blocks=1,
                  channel_mult=[1, 2], num_head_channels=32)


# ---- statistics of the fused GroupNorm on data that is NOT zero-mean (round-5 verdict, weak #2a) --------------------------------------
# (B, Cin, H, W, Cout): 64x64 (16 tiles per sample exchange), 32x32 (4 tiles), 16x16 (complete inside the tile)
@pytest.mark.parametrize("case", [(2, 224, 64, 64, 224), (8, 64, 64, 64, 224), (3, 448, 32, 32, 448), (2, 64, 32, 32, 256), (5, 448, 16, 16, 896)])
def test_fused_groupnorm_statistics_with_mean_shifted_inputs(case):
    """openaimodel.py:261-270 with a LARGE time-embedding row: h = conv(x) + emb_out has |mean| >> sigma per channel (rows of +-30 sigma
    of the convolution's output, bias 10).  E[x^2] - E[x]^2 on fp32 tile partials would lose the variance there; the fused epilogue takes its
    sums before the addend and carries (mean, M2) between tiles.  Against torch fp64 within the bf16 bound, and at least as close as the
    two-launch form (whose statistics pass is fp64)."""
    B, Cin, H, W, Cout = case
    tag = "fgnshift%s" % (case,)
    x = iw.normal(tag, "x", (B, Cin, H, W))
    w = iw.normal(tag, "w", (Cout, Cin, 3, 3), 0.0, 1.0 / math.sqrt(Cin * 9))   # conv output: sigma ~ 1
    b = iw.normal(tag, "b", (Cout,), 10.0, 0.1)
    emb = (30.0 * np.sign(iw.normal(tag, "es", (B, Cout))) + iw.normal(tag, "e", (B, Cout), 0.0, 0.5)).astype(np.float32)
    gamma = iw.normal(tag, "g", (Cout,), 1.0, 0.2)
    beta = iw.normal(tag, "bt", (Cout,), 0.0, 0.2)
    ref = _conv_gn_ref(x, w, b, emb, gamma, beta, 1e-5, 1)
    fused, used = conv_gn_op(x, w, b, emb, gamma, beta, fuse=2, iters=2)
    assert used == 1
    plain, used0 = conv_gn_op(x, w, b, emb, gamma, beta, fuse=0)
    assert used0 == 0
    e_f, e_p = rel_err(fused, ref), rel_err(plain, ref)
    assert e_f < 1.5e-2, (case, e_f, e_p)
    print('fgnshift', case, 'fused', rel_l2(fused, ref), 'plain', rel_l2(plain, ref))
    assert rel_l2(fused, ref) < 8e-3, (case, rel_l2(fused, ref))   # bf16 operands + one bf16 rounding of the output
    assert rel_l2(fused, ref) <= rel_l2(plain, ref) * 1.05 + 1e-4, (case, rel_l2(fused, ref), rel_l2(plain, ref))
    # a per-GROUP offset that the channels of a group share (so that it survives in the group mean: |mean| ~ 40 sigma): same bound
    emb_g = (40.0 * np.repeat(np.sign(iw.normal(tag, "eg", (B, 32))), Cout // 32, axis=1)).astype(np.float32)
    ref_g = _conv_gn_ref(x, w, b, emb_g, gamma, beta, 1e-5, 1)
    fused_g, used = conv_gn_op(x, w, b, emb_g, gamma, beta, fuse=2)
    assert used == 1 and rel_err(fused_g, ref_g) < 1.5e-2 and rel_l2(fused_g, ref_g) < 8e-3, (case, rel_err(fused_g, ref_g), rel_l2(fused_g, ref_g))


# ---- the fallback lives in the library (round-5 verdict, weak #2b; advisor: graphs, fast fall-through) --------------------------------
def _small_unet(max_batch=8, tag="fgn6.unet"):
    from ddmi_amd.unet import UNetModel
    unet = UNetModel(**SMALL_UNET, numerics="bf16", max_batch=max_batch)
    unet.load_state_dict(named_state(tag, unet.manifest()))
    return unet


def test_handoff_timeout_switches_the_fusion_off_for_the_handle_and_drops_its_graphs():
    """What a timed-out hand-off leaves behind (injected: ddmi_unet_debug_inject_gn_timeout raises the two flags as the kernel would):
    the NEXT forward refuses to run and reports it, async_status reports it once, the handle then runs - eagerly and in newly captured
    graphs - on the separate passes (fgn_launches == 0, bits equal to a network planned with the fusion off), and stays so."""
    L = _lib.lib()
    _lib.init_device(0)
    x = torch.from_numpy(iw.normal("fgn6", "x", (8, 16, 32, 32))).to(DEV)
    t = torch.tensor([3, 50, 7, 120, 999, 0, 410, 64], dtype=torch.int64, device=DEV)
    try:
        _lib.check(L.ddmi_debug_gn_fuse(0))
        ref_unfused = _small_unet()(x, t).clone()
        _lib.check(L.ddmi_debug_gn_fuse(2))
        unet = _small_unet()
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            ref_fused = unet(x, t).clone()
            assert unet.async_status() > 0
            unet.set_graph(True)
            out = torch.empty_like(ref_fused)
            for _ in range(3):
                unet(x, t, out=out)          # run + capture, replay, replay
            side.synchronize()
            assert torch.equal(out, ref_fused)
            _lib.check(L.ddmi_unet_debug_inject_gn_timeout(unet._handle))
            with pytest.raises(_lib.DdmiError, match="fused GroupNorm"):
                unet(x, t, out=out)          # the forward after the flag went up reports instead of running
            unet(x, t, out=out)              # a new capture: the separate passes
            unet(x, t, out=out)              # its replay
            side.synchronize()
            assert unet.async_status() == 0, "the handle still runs fused launches after a timeout"
            assert torch.equal(out, ref_unfused)
            unet.set_graph(False)
            assert torch.equal(unet(x, t), ref_unfused)
        # flipping the global switch after a capture never replays the graph captured under the other mode (advisor, bench.py:245)
        unet2 = _small_unet()
        with torch.cuda.stream(side):
            unet2.set_graph(True)
            for _ in range(2):
                unet2(x, t, out=out)
            side.synchronize()
            assert torch.equal(out, ref_fused) and unet2.async_status() > 0
            _lib.check(L.ddmi_debug_gn_fuse(0))
            for _ in range(2):
                unet2(x, t, out=out)
            side.synchronize()
            assert unet2.async_status() == 0 and torch.equal(out, ref_unfused)
            _lib.check(L.ddmi_debug_gn_fuse(2))
            unet2(x, t, out=out)             # back on: the first graph is still there and still right
            side.synchronize()
            assert torch.equal(out, ref_fused)
    finally:
        L.ddmi_debug_gn_fuse(-1)


def test_sampler_reruns_the_chain_when_a_handoff_timed_out():
    """DDPM.sample (ddpm.py:371-381) must not lose a batch: with the timeout flag raised in the middle of a chain the sampler gets the
    error from the library, which has switched the fusion off for the handle, and samples the chain again - the result is the chain of
    the unfused network from the same seeds."""
    from ddmi_amd.ddpm import DDPM
    L = _lib.lib()
    _lib.init_device(0)
    seeds = torch.tensor([11, 12, 13, 14, 15, 16, 17, 18], dtype=torch.int64, device=DEV)
    shape = (8, 16, 32, 32)
    try:
        _lib.check(L.ddmi_debug_gn_fuse(0))
        d0 = DDPM(model=_small_unet(), timesteps=6, sampling_timesteps=6, channels=16, image_size=32)
        want = d0.sample(shape, seeds=seeds).clone()
        assert getattr(d0, "gn_fuse_fallbacks", 0) == 0
        _lib.check(L.ddmi_debug_gn_fuse(2))
        unet = _small_unet()
        calls = {"n": 0}

        class TimesOutAtTheThirdCall:   # the UNet, with the hand-off's timeout flag raised just before its third forward
            def __call__(self, *a, **kw):
                calls["n"] += 1
                if calls["n"] == 3:
                    _lib.check(L.ddmi_unet_debug_inject_gn_timeout(unet._handle))
                return unet(*a, **kw)

            def async_status(self):
                return unet.async_status()

            def set_uniform_t(self, on):
                return unet.set_uniform_t(on)
        d = DDPM(model=TimesOutAtTheThirdCall(), timesteps=6, sampling_timesteps=6, channels=16, image_size=32)
        got = d.sample(shape, seeds=seeds)
        assert d.gn_fuse_fallbacks == 1 and "fused GroupNorm" in d.gn_fuse_fallback_message
        assert d.fused_gn_launches == 0
        assert torch.equal(got, want)
        assert calls["n"] == 3 + 6     # the third call raised instead of running; the whole chain ran again
        got2 = d.sample(shape, seeds=seeds)          # and the handle stays on the separate passes
        assert d.gn_fuse_fallbacks == 1 and torch.equal(got2, want)
    finally:
        L.ddmi_debug_gn_fuse(-1)


def test_two_unet_handles_on_two_streams_with_the_fused_handoff():
    """handles may run concurrently on their own streams.  Two UNets whose conv1 launches exchange GroupNorm statistics
    between tiles (64x64: 16 tiles per sample, 32x32: 4) run 12 forwards each on two streams at once, at a batch that fills the chip
    several times over: the progress argument of k_conv.hip holds with a second fused launch interleaved (no timeout flagged), and every
    result equals the handle's own serial result bit for bit."""
    from ddmi_amd.unet import UNetModel
    cfg = dict(image_size=64, in_channels=16, model_channels=128, out_channels=16, attention_resolutions=[], num_res_blocks=1,
               channel_mult=[1, 2], num_head_channels=32)
    L = _lib.lib()
    _lib.init_device(0)
    B = 64
    try:
        _lib.check(L.ddmi_debug_gn_fuse(2))
        nets, xs, refs = [], [], []
        for i in range(2):
            u = UNetModel(**cfg, numerics="bf16", max_batch=B)
            u.load_state_dict(named_state("fgn6.two%d" % i, u.manifest()))
            x = torch.from_numpy(iw.normal("fgn6.two", "x%d" % i, (B, 16, 64, 64))).to(DEV)
            nets.append(u)
            xs.append(x)
        t = torch.full((B,), 77, dtype=torch.int64, device=DEV)
        for u, x in zip(nets, xs):
            refs.append(u(x, t).clone())
            assert u.async_status() > 0
        torch.cuda.synchronize()
        streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        outs = [[torch.empty_like(refs[i]) for _ in range(12)] for i in range(2)]
        for k in range(12):
            for i in range(2):
                with torch.cuda.stream(streams[i]):
                    nets[i](xs[i], t, out=outs[i][k])
        torch.cuda.synchronize()
        for i in range(2):
            assert nets[i].async_status() > 0       # raises if a hand-off timed out
            for k in range(12):
                assert torch.equal(outs[i][k], refs[i]), (i, k)
    finally:
        L.ddmi_debug_gn_fuse(-1)


# ---- the small-batch regime (net.h; SURVEY 8(d) config 2 at B in {1, 8, 16}) ----------------------------------------------------------
def _dev(a):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32).to(DEV).contiguous()


# (B, C1, C2, HW, silu): GroupNorm shapes of the AFHQ UNet that fit a workgroup's registers - 64x64 x 224 (7 channels per group: 14-byte
# runs, 28 elements per thread), concatenated inputs of the output blocks (672 = 448 | 224: 21 per group, group 21 straddles the
# boundary; 2688 = 1344 | 1344 at 8x8: 84 per group; 1792 at 16x16: 56 per group = 14 per thread), the AttentionBlock norm (no SiLU), a batch of 16
GN_SMALL_CASES = [(1, 224, 0, 4096, 1), (2, 448, 224, 256, 1), (3, 896, 0, 256, 0), (1, 1344, 1344, 64, 1), (16, 448, 0, 1024, 1),
                  (8, 896, 896, 256, 1)]


@pytest.mark.parametrize("case", GN_SMALL_CASES)
def test_one_launch_groupnorm_matches_torch_and_the_two_pass_form(case):
    """util.py:198-215 (GroupNorm32) as ONE launch for small batches (k_groupnorm.hip gn_small_kernel): a workgroup per (sample, group)
    takes shifted-data statistics and applies them.  Against torch fp64 on the bf16-rounded input within one output rounding, on data with
    |mean| = 20 sigma, and bit-identical from run to run."""
    import torch.nn.functional as F
    B, C1, C2, HW, silu = case
    C = C1 + C2
    tag = "gnsmall%s" % (case,)
    L = _lib.lib()
    _lib.init_device(0)
    x1 = (iw.normal(tag, "x1", (B, C1, HW)) + 20.0).astype(np.float32)
    x2 = iw.normal(tag, "x2", (B, C2, HW), -3.0, 2.0) if C2 else None
    gamma, beta = iw.normal(tag, "g", (C,), 1.0, 0.2), iw.normal(tag, "b", (C,), 0.0, 0.2)
    xin = torch.cat([torch.from_numpy(x1)] + ([torch.from_numpy(x2)] if C2 else []), 1)
    xin = xin.to(torch.bfloat16).double()     # what the kernel reads
    ref = F.group_norm(xin, 32, torch.from_numpy(gamma).double(), torch.from_numpy(beta).double(), 1e-5)
    ref = (F.silu(ref) if silu else ref).float()
    outs = []
    for _ in range(2):
        y = torch.empty((B, C, HW), device=DEV, dtype=torch.float32)
        gd, bd, x1d = _dev(gamma), _dev(beta), _dev(x1)
        x2d = _dev(x2) if C2 else None
        _lib.check(L.ddmi_op_groupnorm_small(x1d.data_ptr(), x2d.data_ptr() if C2 else None, gd.data_ptr(), bd.data_ptr(), y.data_ptr(), B, C1, C2,
                                             HW, 1e-5, silu))
        outs.append(y.cpu())
    assert torch.equal(outs[0], outs[1])
    assert rel_l2(outs[0], ref) < 3e-3, (case, rel_l2(outs[0], ref))     # one bf16 rounding of the output
    assert rel_err(outs[0], ref) < 6e-3, (case, rel_err(outs[0], ref))


def _small_gn_count():
    n = C.c_int(-1)
    _lib.check(_lib.lib().ddmi_debug_small_gn_launches(C.byref(n)))
    return n.value


@pytest.mark.parametrize("case", [(1, 224, 64, 64, 224), (2, 448, 32, 32, 448), (8, 448, 16, 16, 896), (3, 672, 8, 8, 1344)])
def test_conv_then_one_launch_groupnorm_through_the_network_executor(case):
    """openaimodel.py:261-270 at small batch through Net::run: conv3x3 (split-K) + bias + embedding row -> the one-launch GroupNorm32 ->
    SiLU.  Same bits as the regime-off path's statistics pass + apply pass up to the statistics' rounding; torch fp64 within the bf16 bound."""
    B, Cin, H, W, Cout = case
    tag = "sgn%s" % (case,)
    L = _lib.lib()
    x = iw.normal(tag, "x", (B, Cin, H, W))
    w = iw.normal(tag, "w", (Cout, Cin, 3, 3), 0.0, 1.0 / math.sqrt(Cin * 9))
    b = iw.normal(tag, "b", (Cout,), 0.0, 0.1)
    emb = iw.normal(tag, "e", (B, Cout), 0.0, 0.5)
    gamma, beta = iw.normal(tag, "g", (Cout,), 1.0, 0.2), iw.normal(tag, "bt", (Cout,), 0.0, 0.2)
    ref = _conv_gn_ref(x, w, b, emb, gamma, beta, 1e-5, 1)
    try:
        _lib.check(L.ddmi_debug_small_batch(1))
        small, used = conv_gn_op(x, w, b, emb, gamma, beta, fuse=0, iters=2)
        assert used == 0 and _small_gn_count() == 1, case
        _lib.check(L.ddmi_debug_small_batch(0))
        big, _ = conv_gn_op(x, w, b, emb, gamma, beta, fuse=0)
        assert _small_gn_count() == 0
    finally:
        L.ddmi_debug_small_batch(-1)
    assert rel_err(small, ref) < 1.5e-2 and rel_err(big, ref) < 1.5e-2, (case, rel_err(small, ref), rel_err(big, ref))
    assert rel_l2(small, big) < 2e-3, (case, rel_l2(small, big))     # the same bf16 tensor normalised: differences are the last bit of the output


@pytest.mark.parametrize("name,cfgname", [("a", "TOY_UNET"), ("b", "TOY_UNET_B")])
def test_unet_golden_in_the_small_batch_regime(golden, name, cfgname):
    """Fixture U2 (the reference's UNetModel on toy configurations) with the small-batch regime ON: one-launch GroupNorms, the five-launch AttentionBlock, the two-launch skip connection.  Same acceptance as test_unet_golden;
    eager = hipGraph replay; and the regime's forward agrees with the large-batch kernels' on the same weights."""
    from helpers import TOL_BF16, assert_bf16_close
    from test_gpu_networks import make_unet
    u2 = golden.npz("U2_unet.npz")
    cfg = golden.json("configs.json")[cfgname]
    L = _lib.lib()
    x = torch.from_numpy(u2[f"{name}.x"]).to(DEV)
    t = torch.from_numpy(u2[f"{name}.t"]).to(DEV)
    try:
        _lib.check(L.ddmi_debug_small_batch(1))
        unet = make_unet(cfg, f"U2{name}.unet", "bf16")
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            eps = unet(x, t).clone()
            side.synchronize()
            assert _small_gn_count() > 0
            assert rel_err(eps, u2[f"{name}.eps"]) < TOL_BF16
            assert_bf16_close(eps, u2[f"{name}.eps"], f"U2{name} small-batch regime", linf=TOL_BF16, l2=2.0e-2)
            assert torch.equal(unet(x, t), eps)
            unet.set_graph(True)
            out = torch.empty_like(eps)
            for _ in range(3):
                unet(x, t, out=out)
            side.synchronize()
            assert torch.equal(out, eps)
        _lib.check(L.ddmi_debug_small_batch(0))
        big = make_unet(cfg, f"U2{name}.unet", "bf16")(x, t)
        assert _small_gn_count() == 0
        assert rel_l2(eps, big) < 2.0e-2
    finally:
        L.ddmi_debug_small_batch(-1)


def test_afhq_unet_full_size_small_batch_regime_vs_oracle():
    """The AFHQ UNet at B = 1 and B = 8 with the small-batch regime on, against the CPU oracle (openaimodel.py:745-793 restated) on the
    named-seed weights: bf16 acceptance as for the large-batch kernels; rows of the B = 8 batch equal in distribution to B = 1."""
    from helpers import AFHQ_UNET, assert_bf16_close
    from oracle import ddmi_oracle as O
    from test_gpu_networks import make_unet
    L = _lib.lib()
    try:
        _lib.check(L.ddmi_debug_small_batch(1))
        unet = make_unet(AFHQ_UNET, "afhq.unet", "bf16", max_batch=8)
        st = named_state("afhq.unet", unet.manifest())
        x = iw.normal("afhq", "x", (1, 64, 64, 64))
        t = torch.tensor([437], dtype=torch.int64)
        ref = O.unet_forward(O.to_state(st), AFHQ_UNET, torch.from_numpy(x), t)
        eps1 = unet(torch.from_numpy(x).to(DEV), t.to(DEV))
        print('AFHQ small-batch regime: one-launch GroupNorms', _small_gn_count(), 'of 61')
        assert _small_gn_count() >= 50
        assert_bf16_close(eps1, ref, "AFHQ UNet full size, small-batch regime", linf=4e-2, l2=2.0e-2)
        x8 = torch.from_numpy(np.concatenate([x] + [iw.normal("afhq", "x%d" % i, (1, 64, 64, 64)) for i in range(7)])).to(DEV)
        eps8 = unet(x8, t.to(DEV).repeat(8))
        assert_bf16_close(eps8[:1], ref, "AFHQ UNet full size, small-batch regime, row 0 of 8", linf=4e-2, l2=2.0e-2)
        assert torch.equal(unet(x8, t.to(DEV).repeat(8)), eps8)
    finally:
        L.ddmi_debug_small_batch(-1)


# ---- the video-domain training step (widening: tools/ldm/video.py:150-167; UNetModel_Triplane openaimodel.py:800-1221) ----------------
def _toy_video_trainer(golden, numerics):
    from ddmi_amd import train as T
    cfg = golden.json("T2_config.json")["TOY_UNET_T2"]
    tr = T.LDMTrainStep(cfg, dict(timesteps=1000, channels=8), numerics=numerics, lr=1e-4, device=DEV)
    assert tr.video and tuple(tr.mixing_logit.shape) == (1, 8, 1)
    st = {"model." + k: v for k, v in iw.fill_state("T2.unet3", [(k, tuple(p.shape)) for k, p in tr.model.state_dict().items()]).items()}
    st["mixing_logit"] = iw.normal("T2", "mixing_logit", (1, 8, 1), 0.0, 0.5)
    tr.load_state_dict(st).finalize()
    return tr, cfg


def test_video_training_model_has_the_reference_state_dict(golden):
    """TrainUNetModel_Triplane's parameter names and shapes are UNetModel_Triplane.state_dict()'s (fixture T2's manifest, written by the
    real reference) - `ldm-last.pt` of the video trainer loads into it unchanged."""
    from ddmi_amd import train as T
    doc = golden.json("T2_config.json")
    m = T.TrainUNetModel_Triplane(**doc["TOY_UNET_T2"], numerics="fp32", device=DEV)
    ours = {k: list(v.shape) for k, v in m.state_dict().items()}
    assert ours == doc["manifest"], sorted(set(ours) ^ set(doc["manifest"]))[:6]


def test_video_training_step_matches_reference_fixture_t2(golden):
    """`DDPM(domain='video').p_losses` -> backward -> AdamW on the toy tri-plane UNet == the REAL reference (oracle/gen_golden_train_video.py),
    fp32 parity mode < 1e-3: the loss, 31 gradients in full, every one of the 209 gradients by (sum, l2), the parameters after the step."""
    from helpers import TOL_FP32
    t2 = golden.npz("T2_train_step_video.npz")
    tr, cfg = _toy_video_trainer(golden, "fp32")
    L = tr.model.seq_len
    x0 = torch.from_numpy(iw.normal("T2", "x_start", (2, 8, L)))
    nz = torch.from_numpy(iw.normal("T2", "noise", (2, 8, L)))
    loss = tr.loss_and_backward(x0, torch.from_numpy(t2["t"]), nz)
    assert abs(float(loss) - float(t2["loss"])) < TOL_FP32 * float(t2["loss"])
    grads = {k: p.grad for k, p in tr.named_parameters()}
    assert set(grads) == set(str(k) for k in t2["all_keys"])
    floor = 1e-6 * float(t2["all_l2"].max())
    bad = {}
    for k in [k[5:] for k in t2 if k.startswith("grad.")]:
        ref = t2["grad." + k]
        if float(np.sqrt((ref.astype(np.float64) ** 2).sum())) < floor:     # analytically zero: rounding noise on both sides
            if float(grads[k].abs().max()) > floor:
                bad[k] = "noise-level gradient came back large"
        elif not rel_err(grads[k].detach().cpu(), ref) < TOL_FP32:
            bad[k] = rel_err(grads[k].detach().cpu(), ref)
    assert not bad, bad
    for k, s_ref, l2_ref in zip(t2["all_keys"], t2["all_sum"], t2["all_l2"]):
        gk = grads[str(k)].double()
        assert abs(float(gk.pow(2).sum().sqrt()) - l2_ref) < TOL_FP32 * l2_ref + floor, k
        assert abs(float(gk.sum()) - s_ref) < TOL_FP32 * l2_ref * np.sqrt(gk.numel()) + floor, k
    tr.optimizer_step()
    params = dict(tr.named_parameters())
    for k in [k[6:] for k in t2 if k.startswith("adamw.")]:
        gref = np.abs(t2["grad." + k])
        live = torch.from_numpy(gref > 1e-4 * gref.max() + floor * 1e-3).to(DEV)   # Adam's first step is lr * sign(g): noise-level entries masked
        if not bool(live.any()):
            continue
        ref = torch.from_numpy(t2["adamw." + k]).to(DEV)
        err = float(((params[k].detach() - ref).abs() * live).max() / ref.abs().max())
        assert err < 1e-5, (k, err)


def test_video_training_step_bf16_tracks_the_fixture_and_descends(golden):
    from helpers import TOL_BF16
    t2 = golden.npz("T2_train_step_video.npz")
    tr, cfg = _toy_video_trainer(golden, "bf16")
    L = tr.model.seq_len
    x0 = torch.from_numpy(iw.normal("T2", "x_start", (2, 8, L)))
    nz = torch.from_numpy(iw.normal("T2", "noise", (2, 8, L)))
    loss = tr.loss_and_backward(x0, torch.from_numpy(t2["t"]), nz)
    assert abs(float(loss) - float(t2["loss"])) < TOL_BF16 * float(t2["loss"])
    grads = {k: p.grad for k, p in tr.named_parameters()}
    floor = 1e-6 * float(t2["all_l2"].max())
    for k in [k[5:] for k in t2 if k.startswith("grad.")]:
        a, b = grads[k].double().flatten().cpu(), torch.from_numpy(t2["grad." + k]).double().flatten()
        if float(b.norm()) < floor:
            assert float(a.norm()) < 1e-3 * float(t2["all_l2"].max()), k
            continue
        cos = float((a * b).sum() / (a.norm() * b.norm() + 1e-30))
        assert cos > 0.98, (k, cos)
    l0 = float(loss)
    tr.optimizer_step()
    for _ in range(5):
        l1 = float(tr.train_step(x0, torch.from_numpy(t2["t"]), nz))
    assert l1 < l0


@pytest.mark.parametrize("heads,d,T_", [(16, 4, 128), (16, 8, 32), (4, 16, 96), (16, 16, 256), (3, 64, 80), (2, 96, 48)])
def test_attention_backward_at_the_triplane_head_dims(heads, d, T_):
    """QKVAttentionLegacy's backward (openaimodel.py:397-426) at the head dims of the tri-plane UNet - 64-channel AttentionBlock heads,
    16-head inter-plane attention of 16 .. 96 channels per head, the fixtures' 4 / 8 - against torch autograd of the oracle's attention.
    bf16 heads narrower than 32 channels at token counts that are multiples of 64 take the zero-padded MFMA path (train.py
    AttentionLegacyFn: (16, 4, 128) and (16, 16, 256) here), everything else the plain kernels."""
    from ddmi_amd import train as T
    from oracle import ddmi_oracle as O
    for numerics, tol in (("fp32", 2e-5), ("bf16", 4e-2)):
        tc = T.TrainContext(numerics, DEV)
        B = 2
        qkv = torch.from_numpy(iw.normal("ta6", "qkv%d_%d" % (heads, d), (B, heads * 3 * d, T_)))
        dout = torch.from_numpy(iw.normal("ta6", "dout%d_%d" % (heads, d), (B, heads * d, T_)))
        if numerics == "bf16":
            qkv, dout = qkv.to(torch.bfloat16).float(), dout.to(torch.bfloat16).float()
        qr = qkv.clone().requires_grad_(True)
        out_r = O.qkv_attention_legacy(qr, heads)
        out_r.backward(dout)
        qd = qkv.permute(0, 2, 1).reshape(B, 1, T_, heads * 3 * d).contiguous().to(DEV, tc.dtype).requires_grad_(True)
        out = T.AttentionLegacyFn.apply(tc, qd, heads)
        assert rel_err(out.detach().float().cpu().reshape(B, T_, heads * d).permute(0, 2, 1), out_r.detach()) < tol
        out.backward(dout.permute(0, 2, 1).reshape(B, 1, T_, heads * d).contiguous().to(DEV, tc.dtype))
        assert rel_err(qd.grad.detach().float().cpu().reshape(B, T_, heads * 3 * d).permute(0, 2, 1), qr.grad) < tol, (numerics, heads, d)


def test_sky_training_step_full_size_vs_autograd():
    """The configuration tools/ldm/video.py trains (configs/ldm/skytimelapse.yaml: the tri-plane UNet of model_channels 256, channel_mult
    [1, 2, 4, 6], 64-channel AttentionBlock heads, 16-head inter-plane attention of 16 .. 96 channels per head, [B, 64, 2048] latents)
    at B = 1: `p_losses` + backward through the C ABI against torch-CPU autograd of the oracle's forward (oracle.p_losses with
    unet_triplane_forward, pinned to the real reference's gradients by fixture T2).  fp32 mode: the loss and EVERY gradient (norm, sum,
    relative L2) within 1e-3; bf16 mode: loss within the bf16 tolerance, relative L2 <= 3e-2 per gradient; and a step's wall time."""
    import time
    from ddmi_amd import train as T
    from helpers import SKY_UNET, TOL_BF16, TOL_FP32
    from oracle import ddmi_oracle as O
    B, L = 1, 32 * 32 + 2 * 32 * 16
    x0 = torch.from_numpy(iw.normal("sky.train", "x_start", (B, 64, L)))
    nz = torch.from_numpy(iw.normal("sky.train", "noise", (B, 64, L)))
    t = torch.tensor([611], dtype=torch.int64)
    tr = T.LDMTrainStep(SKY_UNET, dict(timesteps=1000, channels=64, domain="video"), numerics="fp32", device=DEV)
    st = named_state("sky.unet3", [(k, tuple(p.shape)) for k, p in tr.model.state_dict().items()])
    logit_np = iw.normal("sky.train", "mixing_logit", (1, 64, 1), 0.0, 0.5)
    t0 = time.time()
    p = {k: torch.from_numpy(np.ascontiguousarray(v)).requires_grad_(True) for k, v in st.items()}
    logit = torch.from_numpy(logit_np.copy()).requires_grad_(True)
    loss_ref = O.p_losses(p, SKY_UNET, O.make_schedule(1000), logit, x0, t, nz, model=O.unet_triplane_forward)
    loss_ref.backward()
    ref = {"model." + k: v.grad for k, v in p.items()}
    ref["mixing_logit"] = logit.grad
    print("oracle: full-size tri-plane p_losses + autograd at B=1 in %.1f s" % (time.time() - t0))
    l2max = max(float(g.double().norm()) for g in ref.values())

    def run(tr):
        full = {"model." + k: v for k, v in st.items()}
        full["mixing_logit"] = logit_np
        tr.load_state_dict(full).finalize()
        loss = tr.loss_and_backward(x0, t, nz)
        return float(loss), {k: pp.grad.detach().float().cpu() for k, pp in tr.named_parameters()}

    loss32, g32 = run(tr)
    assert abs(loss32 - float(loss_ref)) < TOL_FP32 * float(loss_ref)
    assert set(g32) == set(ref)
    bad = {}
    for k, r in ref.items():
        g, rn = g32[k].double(), float(r.double().norm())
        e_norm = abs(float(g.norm()) - rn) / (rn + 1e-3 * l2max)
        e_sum = abs(float(g.sum()) - float(r.double().sum())) / (rn * np.sqrt(g.numel()) + 1e-3 * l2max)
        e_l2 = float((g - r.double()).norm()) / (rn + 1e-3 * l2max)
        if max(e_norm, e_sum, e_l2) >= TOL_FP32:
            bad[k] = (e_norm, e_sum, e_l2, rn)
    assert not bad, (len(bad), sorted(bad.items(), key=lambda kv: -kv[1][2])[:10])
    del tr, g32
    torch.cuda.empty_cache()
    tr16 = T.LDMTrainStep(SKY_UNET, dict(timesteps=1000, channels=64, domain="video"), numerics="bf16", device=DEV)
    loss16, g16 = run(tr16)
    assert abs(loss16 - float(loss_ref)) < TOL_BF16 * float(loss_ref)
    bad = {k: rel_l2(g16[k], r) for k, r in ref.items() if float(r.double().norm()) > 1e-4 * l2max and rel_l2(g16[k], r) > 3e-2}
    assert not bad, sorted(bad.items(), key=lambda kv: -kv[1])[:8]
    # wall time of a bf16 step at B = 8 (eager; the attention backward at head dims other than 32 runs on the plain kernels)
    z8 = torch.randn((8, 64, L), device=DEV)
    tr16.train_step(z8)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(3):
        tr16.train_step(z8)
    torch.cuda.synchronize()
    print("sky training step (bf16, B = 8, eager): %.1f ms per step = %.1f clips/s" % ((time.time() - t0) / 3 * 1e3, 8 / ((time.time() - t0) / 3)))


def test_main_video_train_then_gen_end_to_end(golden, tmp_path):
    """`python main.py --exp ldm --configs <yaml>` in the VIDEO domain: data.mode train (stage-1 checkpoint -> optimizer steps on synthetic
    [B, C, L] latents through the tri-plane UNet -> ldm-last.pt in the layout tools/ldm/video.py:94-108 writes, mixing_logit [1, C, 1]) and
    then data.mode gen from that checkpoint (VideoPipeline: sample -> VITAutoencoder.decode -> MLPVideo)."""
    import os
    import subprocess
    import sys
    import yaml
    from ddmi_amd.video import decoderv_manifest, mlpv_manifest
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfgs = golden.json("configs_video.json")
    save = str(tmp_path)
    tt = lambda d: {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in d.items()}
    torch.save({"model": tt(named_state("e2ev.vit", decoderv_manifest(cfgs["TOY_DDV"], cfgs["TOY_EMBED_V"]))),
                "mlp": tt(named_state("e2ev.mlpv", mlpv_manifest()))}, os.path.join(save, "model-last.pt"))
    uc = cfgs["TOY_UNET_V"]
    base = {"model": {"embed_dim": cfgs["TOY_EMBED_V"], "lr": 1e-4, "pretrained": False, "resume": False,
                      "params": {"unetconfig": uc, "ddconfig": cfgs["TOY_DDV"], "mlpconfig": cfgs["MLPV_CFG"],
                                 "ddpmconfig": {"timesteps": 1000, "sampling_timesteps": 4, "ddim_sampling_eta": 0.0, "channels": uc["in_channels"]},
                                 "lossconfig": {"gradient_accumulate_every": 2, "ema_decay": 0.9999, "ema_update_every": 1}}},
            "data": {"domain": "video", "mode": "train", "save_pth": save, "test_batch_size": 2, "batch_size": 4, "test_resolution": 64,
                     "frames": cfgs["TOY_FRAMES"], "train_steps": 6}}

    def run(cfg, name):
        path = os.path.join(save, name)
        with open(path, "w") as f:
            yaml.safe_dump(cfg, f)
        out = subprocess.run([sys.executable, os.path.join(ROOT, "main.py"), "--exp", "ldm", "--configs", path, "--numerics", "fp32"],
                             capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
        return out.stdout
    log = run(base, "train.yaml")
    assert "Dae loss" in log and "after 3 optimizer steps" in log            # 6 micro-steps / gradient_accumulate_every 2
    ck = torch.load(os.path.join(save, "ldm-last.pt"), map_location="cpu", weights_only=True)
    assert {"diffusion", "ema", "vaemodel", "mlp", "step"} <= set(ck)
    assert tuple(ck["diffusion"]["mixing_logit"].shape) == (1, uc["in_channels"], 1)
    assert "model.input_attns.1.q.weight" in ck["diffusion"] and "ema_model.model.mid_attn.proj_out.weight" in ck["ema"]
    gen = dict(base, data=dict(base["data"], mode="gen"), model=dict(base["model"], pretrained=True))
    log = run(gen, "gen.yaml")
    assert "Loading Pretrained Models!" in log and "Finished generating videos!" in log
    vid = np.load(os.path.join(save, "results", "generation_video.npy"))
    assert vid.shape == (2, cfgs["TOY_FRAMES"], 64, 64, 3) and vid.dtype == np.uint8 and vid.std() > 0
