"""SURVEY 8(f) rank 4 - the LDM training step on the GPU (include/ddmi_hip.h "Training", ddmi_amd/train.py):
  * every backward op against torch's CPU autograd on the same fp32 inputs (the floating-point reference of these kernels);
  * one whole step - p_losses, backward, AdamW - against fixture T1 captured from the REAL reference
    (oracle/gen_golden_train.py: loss, gradients, updated parameters on the toy UNet), parity mode < 1e-3;
  * the bf16 mode against the same fixture at bf16 tolerances (loss, gradient direction)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from ddmi_amd import _lib
from ddmi_amd import initweights as iw
from ddmi_amd import train as T
from helpers import TOL_BF16, TOL_FP32, named_state, rel_err
from oracle import ddmi_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))


def nhwc(x, tc):   # [B,C,H,W] fp32 cpu -> NHWC storage tensor on the device (leaf)
    return x.permute(0, 2, 3, 1).contiguous().to(DEV, tc.dtype).requires_grad_(True)


def nchw(y):
    return y.detach().float().permute(0, 3, 1, 2).cpu()


def g(tag, key, shape, std=1.0):
    return torch.from_numpy(iw.normal(tag, key, shape, 0.0, std))


@pytest.mark.parametrize("numerics,tol", [("fp32", 2e-5), ("bf16", TOL_BF16)])
@pytest.mark.parametrize("k,stride,ups,cin,cout,hw", [(3, 1, 0, 32, 64, 8), (1, 1, 0, 64, 96, 8), (3, 2, 0, 32, 32, 8), (3, 1, 1, 32, 32, 4),
                                                     (3, 1, 0, 224, 224, 16),
                                                     # the TN implicit-GEMM wgrad (bf16): several pixel tiles per sample, ragged
                                                     # channel tiles, W = 64 (two rows per tile), 1x1 at 448 channels
                                                     (3, 1, 0, 96, 160, 32), (3, 1, 0, 72, 40, 64), (1, 1, 0, 224, 448, 16),
                                                     (3, 1, 1, 64, 96, 16)])   # Upsample.conv: 16x16 source, 32x32 grid
def test_conv_forward_backward_vs_torch(numerics, tol, k, stride, ups, cin, cout, hw):
    tc = T.TrainContext(numerics, DEV)
    B = 2
    x, w, b = g("tc", "x", (B, cin, hw, hw)), g("tc", "w", (cout, cin, k, k), 1.0 / np.sqrt(cin * k * k)), g("tc", "b", (cout,), 0.1)
    emb = g("tc", "emb", (B, cout), 0.3)
    xr, wr, br, er = (t.clone().requires_grad_(True) for t in (x, w, b, emb))
    xin = F.interpolate(xr, scale_factor=2, mode="nearest") if ups else xr
    yr = F.conv2d(xin, wr, br, stride=stride, padding=k // 2) + er[:, :, None, None]
    res = g("tc", "res", tuple(yr.shape), 0.5)
    rr = res.clone().requires_grad_(True)
    yr = yr + rr
    dy = g("tc", "dy", tuple(yr.shape))
    yr.backward(dy)
    xd, rd = nhwc(x, tc), nhwc(res, tc)
    wd, bd, ed = (t.clone().to(DEV).requires_grad_(True) for t in (w, b, emb))
    y = T.ConvFn.apply(tc, xd, wd, bd, ed, rd, stride, bool(ups))
    assert rel_err(nchw(y), yr.detach()) < tol
    y.backward(dy.permute(0, 2, 3, 1).contiguous().to(DEV, tc.dtype))
    assert rel_err(nchw(xd.grad), xr.grad) < tol
    assert rel_err(wd.grad, wr.grad) < tol
    assert rel_err(bd.grad, br.grad) < tol
    assert rel_err(ed.grad, er.grad) < tol
    assert rel_err(nchw(rd.grad), rr.grad) < (1e-6 if numerics == "fp32" else 1e-2)


@pytest.mark.parametrize("numerics,tol", [("fp32", 2e-5), ("bf16", TOL_BF16)])
@pytest.mark.parametrize("silu", [True, False])
@pytest.mark.parametrize("C,hw", [(64, 8), (224, 16), (2688, 4)])
def test_groupnorm_backward_vs_torch(numerics, tol, silu, C, hw):
    tc = T.TrainContext(numerics, DEV)
    B = 2
    x, ga, be = g("tg", "x", (B, C, hw, hw)) * 1.7 + 0.3, g("tg", "g", (C,), 0.1) + 1.0, g("tg", "b", (C,), 0.1)
    if numerics == "bf16":
        x = x.to(torch.bfloat16).float()     # the same stored values on both sides
    xr, gr, br = (t.clone().requires_grad_(True) for t in (x, ga, be))
    yr = F.group_norm(xr, 32, gr, br, 1e-5)
    if silu:
        yr = F.silu(yr)
    dy = g("tg", "dy", (B, C, hw, hw))
    if numerics == "bf16":
        dy = dy.to(torch.bfloat16).float()
    yr.backward(dy)
    xd = nhwc(x, tc)
    gd, bd = ga.clone().to(DEV).requires_grad_(True), be.clone().to(DEV).requires_grad_(True)
    y = T.GroupNormFn.apply(tc, xd, gd, bd, 1e-5, silu)
    assert rel_err(nchw(y), yr.detach()) < tol
    y.backward(dy.permute(0, 2, 3, 1).contiguous().to(DEV, tc.dtype))
    assert rel_err(nchw(xd.grad), xr.grad) < tol
    assert rel_err(gd.grad, gr.grad) < tol
    assert rel_err(bd.grad, br.grad) < tol


@pytest.mark.parametrize("numerics,tol", [("fp32", 2e-5), ("bf16", TOL_BF16)])
@pytest.mark.parametrize("heads,Tside", [(2, 8), (14, 16)])
def test_attention_legacy_backward_vs_torch(numerics, tol, heads, Tside):
    tc = T.TrainContext(numerics, DEV)
    B, d = 2, 32
    qkv = g("ta", "qkv", (B, heads * 3 * d, Tside, Tside))
    if numerics == "bf16":
        qkv = qkv.to(torch.bfloat16).float()
    qr = qkv.clone().requires_grad_(True)
    out_r = O.qkv_attention_legacy(qr.reshape(B, heads * 3 * d, Tside * Tside), heads)      # [B, heads*d, T]
    dout = g("ta", "dout", tuple(out_r.shape))
    if numerics == "bf16":
        dout = dout.to(torch.bfloat16).float()
    out_r.backward(dout)
    qd = nhwc(qkv, tc)
    out = T.AttentionLegacyFn.apply(tc, qd, heads)
    assert rel_err(nchw(out).reshape(B, heads * d, -1), out_r.detach()) < tol
    out.backward(dout.reshape(B, heads * d, Tside, Tside).permute(0, 2, 3, 1).contiguous().to(DEV, tc.dtype))
    assert rel_err(nchw(qd.grad), qr.grad) < tol


def test_linear_loss_adamw_qsample_vs_torch():
    tc = T.TrainContext("fp32", DEV)
    B, K, N = 3, 96, 160
    for silu_in in (False, True):
        x, W, b = g("tl", "x", (B, K)), g("tl", "W", (N, K), 0.1), g("tl", "b", (N,), 0.1)
        xr, Wr, br = (t.clone().requires_grad_(True) for t in (x, W, b))
        yr = F.linear(F.silu(xr) if silu_in else xr, Wr, br)
        dy = g("tl", "dy", (B, N))
        yr.backward(dy)
        xd, Wd, bd = (t.clone().to(DEV).requires_grad_(True) for t in (x, W, b))
        y = T.LinearFn.apply(tc, xd, Wd, bd, silu_in)
        assert rel_err(y, yr.detach()) < 1e-5
        y.backward(dy.to(DEV))
        assert rel_err(xd.grad, xr.grad) < 1e-5 and rel_err(Wd.grad, Wr.grad) < 1e-5 and rel_err(bd.grad, br.grad) < 1e-5
    # q_sample + mixed-prediction loss: the oracle's schedule and mixing (ddpm.py:383-386, 469-477)
    B, Cc, H = 2, 8, 6
    sched = O.make_schedule(1000)
    t = torch.tensor([3, 900])
    x0, nz, mo = g("tq", "x0", (B, Cc, H, H)), g("tq", "nz", (B, Cc, H, H)), g("tq", "mo", (B, Cc, H, H))
    logit = g("tq", "logit", (1, Cc, 1, 1), 1.5)
    a = torch.from_numpy(sched["sqrt_alphas_cumprod"])[t]
    s = torch.from_numpy(sched["sqrt_one_minus_alphas_cumprod"])[t]
    xt_ref = a.view(-1, 1, 1, 1) * x0 + s.view(-1, 1, 1, 1) * nz
    xt = torch.empty_like(x0, device=DEV)
    ad, sd = a.to(DEV).contiguous(), s.to(DEV).contiguous()
    x0d, nzd = x0.to(DEV), nz.to(DEV)
    _lib.check(_lib.lib().ddmi_train_q_sample(x0d.data_ptr(), nzd.data_ptr(), ad.data_ptr(), sd.data_ptr(), xt.data_ptr(), B, Cc * H * H,
                                              _lib.stream_ptr()))
    assert rel_err(xt, xt_ref) < 1e-6
    mor, lr_ = mo.clone().requires_grad_(True), logit.clone().requires_grad_(True)
    coef = torch.sigmoid(lr_)
    eps = (1 - coef) * (s.view(-1, 1, 1, 1) * xt_ref) + coef * mor
    loss_r = F.mse_loss(nz, eps, reduction="none").mean(dim=[1, 2, 3]).mean()
    loss_r.backward()
    mod, ld = mo.clone().to(DEV).requires_grad_(True), logit.clone().to(DEV).requires_grad_(True)
    loss = T.EpsLossFn.apply(tc, mod, xt, nzd, sd, ld)
    loss.backward()
    assert abs(float(loss) - float(loss_r)) < 1e-6 * abs(float(loss_r))
    assert rel_err(mod.grad, mor.grad) < 1e-5 and rel_err(ld.grad, lr_.grad) < 1e-5
    # AdamW: three steps against torch.optim.AdamW
    p0, grads = g("to", "p", (1000,)), [g("to", f"g{i}", (1000,)) for i in range(3)]
    pr = p0.clone().requires_grad_(True)
    opt = torch.optim.AdamW([pr], lr=1e-3, weight_decay=0.01)
    pd, m, v = p0.clone().to(DEV), torch.zeros(1000, device=DEV), torch.zeros(1000, device=DEV)
    for i, gr in enumerate(grads):
        pr.grad = gr.clone()
        opt.step()
        gd = (gr * 4).to(DEV)   # grad_scale 0.25 undoes it: the all-reduce / accumulation factor
        _lib.check(_lib.lib().ddmi_train_adamw(pd.data_ptr(), gd.data_ptr(), m.data_ptr(), v.data_ptr(), 1000, 1e-3, 0.9, 0.999, 1e-8, 0.01, i + 1,
                                               0.25, _lib.stream_ptr()))
    assert rel_err(pd, pr.detach()) < 1e-6


def _toy_trainer(golden, numerics):
    cfg = golden.json("configs.json")["TOY_UNET"]
    tr = T.LDMTrainStep(cfg, dict(timesteps=1000, channels=64), numerics=numerics, lr=1e-4, weight_decay=0.0, device=DEV)
    st = {"model." + k: v for k, v in named_state("T1.unet", [(k, tuple(p.shape)) for k, p in tr.model.state_dict().items()]).items()}
    st["mixing_logit"] = iw.normal("T1", "mixing_logit", (1, 64, 1, 1), 0.0, 0.5)
    tr.load_state_dict(st).finalize()
    return tr


def test_training_step_matches_reference_fixture_t1(golden):
    """p_losses -> backward -> AdamW on the toy UNet == the real reference (oracle/gen_golden_train.py), parity mode."""
    t1 = golden.npz("T1_train_step.npz")
    tr = _toy_trainer(golden, "fp32")
    x0, nz = g("T1", "x_start", (2, 64, 8, 8)), g("T1", "noise", (2, 64, 8, 8))
    loss = tr.loss_and_backward(x0, torch.from_numpy(t1["t"]), nz)
    assert abs(float(loss) - float(t1["loss"])) < TOL_FP32 * float(t1["loss"])
    grads = {k: p.grad for k, p in tr.named_parameters()}
    # gradients that are analytically ZERO come back as fp32 rounding noise on both sides (the toy UNet's 32-channel blocks
    # have ONE channel per GroupNorm group, so a per-channel constant added before the norm - conv bias, time-embedding
    # addend - cannot change the output: |grad| ~ 1e-9 against 1e-2 elsewhere).  They are held to an absolute bound.
    floor = 1e-6 * float(t1["all_l2"].max())
    bad = {}
    for k in [k[5:] for k in t1 if k.startswith("grad.")]:
        ref = t1["grad." + k]
        if float(np.sqrt((ref.astype(np.float64) ** 2).sum())) < floor:
            if float(grads[k].abs().max()) > floor:
                bad[k] = "noise-level gradient came back large"
        elif not rel_err(grads[k], ref) < TOL_FP32:
            bad[k] = rel_err(grads[k], ref)
    assert not bad, bad
    for k, s_ref, l2_ref in zip(t1["all_keys"], t1["all_sum"], t1["all_l2"]):     # every one of the 129 gradients
        gk = grads[str(k)].double()
        assert abs(float(gk.pow(2).sum().sqrt()) - l2_ref) < TOL_FP32 * l2_ref + floor, k
        assert abs(float(gk.sum()) - s_ref) < TOL_FP32 * l2_ref * np.sqrt(gk.numel()) + floor, k
    tr.optimizer_step()
    params = dict(tr.named_parameters())
    for k in [k[6:] for k in t1 if k.startswith("adamw.")]:
        # Adam's first step is lr * sign(g): where the gradient is analytically zero (whole tensors, see above; the key third of
        # a qkv bias - softmax is invariant to a constant added to a row of logits) the sign of rounding noise decides, which is
        # not a property of the algorithm.  Those elements are masked; everything else must agree.
        gref = np.abs(t1["grad." + k])
        live = torch.from_numpy(gref > 1e-4 * gref.max() + floor * 1e-3).to(DEV)
        if not bool(live.any()):
            continue
        ref = torch.from_numpy(t1["adamw." + k]).to(DEV)
        err = float(((params[k].detach() - ref).abs() * live).max() / ref.abs().max())
        assert err < 1e-5, (k, err)
    assert float(tr.flat.grad.abs().max()) == 0.0      # zero_grad after the step


def test_training_step_bf16_tracks_the_fixture(golden):
    t1 = golden.npz("T1_train_step.npz")
    tr = _toy_trainer(golden, "bf16")
    x0, nz = g("T1", "x_start", (2, 64, 8, 8)), g("T1", "noise", (2, 64, 8, 8))
    loss = tr.loss_and_backward(x0, torch.from_numpy(t1["t"]), nz)
    assert abs(float(loss) - float(t1["loss"])) < TOL_BF16 * float(t1["loss"])
    grads = {k: p.grad for k, p in tr.named_parameters()}
    floor = 1e-6 * float(t1["all_l2"].max())
    for k in [k[5:] for k in t1 if k.startswith("grad.")]:
        a, b = grads[k].double().flatten().cpu(), torch.from_numpy(t1["grad." + k]).double().flatten()
        if float(b.norm()) < floor:     # analytically zero (see the fp32 test): bf16 noise must stay small
            assert float(a.norm()) < 1e-3 * float(t1["all_l2"].max()), k
            continue
        cos = float((a * b).sum() / (a.norm() * b.norm() + 1e-30))
        assert cos > 0.98, (k, cos)
    # a few steps reduce the loss on a fixed batch (the optimizer really descends)
    l0 = float(loss)
    tr.optimizer_step()
    for _ in range(5):
        l1 = float(tr.train_step(x0, torch.from_numpy(t1["t"]), nz))
    assert l1 < l0


@pytest.mark.parametrize("numerics", ["fp32", "bf16"])
def test_graph_replay_equals_eager_steps(golden, numerics):
    """LDMTrainStep(use_graph=True) - first step eager, second captured into a hipGraph, later ones replayed - walks through the
    same states as the eager trainer, bit for bit (same kernels, same order, same static gradient slab)."""
    t1 = golden.npz("T1_train_step.npz")
    runs = []
    for use_graph in (False, True):
        tr = _toy_trainer(golden, numerics)
        tr.use_graph = use_graph
        losses = []
        for step in range(5):
            x0, nz = g("T1g", f"x{step}", (2, 64, 8, 8)), g("T1g", f"n{step}", (2, 64, 8, 8))
            t = (torch.from_numpy(t1["t"]) + 37 * step) % 1000
            losses.append(float(tr.train_step(x0, t, nz)))
        runs.append((losses, tr.flat.data.clone(), tr.flat.exp_avg_sq.clone()))
        if use_graph:
            assert isinstance(tr._graphs[(2, 64, 8, 8)], tuple)      # a graph was really captured and replayed
    assert runs[0][0] == runs[1][0]
    assert torch.equal(runs[0][1], runs[1][1]) and torch.equal(runs[0][2], runs[1][2])


def test_fused_gradient_accumulation_equals_autograd_accumulation(golden):
    """Conv / GroupNorm parameter gradients added straight into the flat slab by the kernels' last stage (ddmi_train_accumulate_next)
    == the same gradients returned to autograd and added by AccumulateGrad; two micro-steps, so that a non-zero slab is added to."""
    t1 = golden.npz("T1_train_step.npz")
    slabs = []
    for fused in (True, False):
        T.FUSE_GRAD_ACCUMULATION = fused
        try:
            tr = _toy_trainer(golden, "bf16")
            for step in range(2):
                x0, nz = g("T1f", f"x{step}", (2, 64, 8, 8)), g("T1f", f"n{step}", (2, 64, 8, 8))
                tr.loss_and_backward(x0, (torch.from_numpy(t1["t"]) + 11 * step) % 1000, nz)
            slabs.append(tr.flat.grad.clone())
        finally:
            T.FUSE_GRAD_ACCUMULATION = True
    assert float(slabs[0].abs().max()) > 0
    assert torch.equal(slabs[0], slabs[1])


def test_trained_checkpoint_round_trips_into_the_sampling_path(golden, tmp_path):
    """A few steps -> LDMTrainStep.save (the dict LDMTrainer.save writes, tools/ldm/image.py:104-119) -> ImagePipeline.load_checkpoint:
    the inference UNet on the saved online weights equals the training-path forward, the EMA copy is what use_ema picks."""
    import os
    from ddmi_amd.pipeline import ImagePipeline
    cfgs = golden.json("configs.json")
    tr = _toy_trainer(golden, "fp32")
    tr.ema_every = 1
    tr.ema_update_after_step = 0   # ema_pytorch copies the online weights for its first update_after_step (100) calls; leave that phase at once
    x0, nz = g("T1", "x_start", (2, 64, 8, 8)), g("T1", "noise", (2, 64, 8, 8))
    for i in range(3):
        tr.train_step(x0, torch.tensor([3 + i, 900 - i]), nz)
    pipe = ImagePipeline(cfgs["TOY_UNET"], cfgs["TOY_DD"], cfgs["MLP_CFG"], dict(timesteps=1000, sampling_timesteps=10, channels=64, image_size=8),
                         cfgs["TOY_EMBED"], test_resolution=32, numerics="fp32", max_batch=2, device=DEV)
    tt = lambda d: {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in d.items()}
    path = os.path.join(tmp_path, "ldm-last.pt")
    tr.save(path, vaemodel_state=tt(named_state("ck.vae", pipe.vae.manifest())), mlp_state=tt(named_state("ck.mlp", pipe.mlp.manifest())))
    xt, t = g("T1", "probe", (2, 64, 8, 8)).to(DEV), torch.tensor([17, 640], device=DEV)
    with torch.no_grad():
        want = tr.model(xt, t).clone()
    pipe.load_checkpoint(path, use_ema=False)
    pipe.finalize()
    assert rel_err(pipe.unet(xt, t), want) < 1e-5
    assert rel_err(pipe.ddpm.mixing_logit, tr.mixing_logit.detach()) < 1e-7
    # EMA (ema_pytorch: copy at call 0, then lerps with the warm-up decays 1 - 2^(-2/3), 1 - 3^(-2/3)): close to, but not, the online weights
    pipe2 = ImagePipeline(cfgs["TOY_UNET"], cfgs["TOY_DD"], cfgs["MLP_CFG"], dict(timesteps=1000, sampling_timesteps=10, channels=64, image_size=8),
                          cfgs["TOY_EMBED"], test_resolution=32, numerics="fp32", max_batch=2, device=DEV)
    pipe2.load_checkpoint(path, use_ema=True)
    pipe2.finalize()
    e = rel_err(pipe2.unet(xt, t), want)
    assert 0 < e < 0.5       # three lr = 1e-4 Adam steps moved every weight by 3e-4: ~10 % in the output of this random net
    # and it is exactly the EMA slab: the training-path forward on those weights gives the same output
    with torch.no_grad():
        keep = tr.flat.data.clone()
        tr.flat.data.copy_(tr.ema)
        want_ema = tr.model(xt, t).clone()
        tr.flat.data.copy_(keep)
    assert rel_err(pipe2.unet(xt, t), want_ema) < 1e-5


def test_conv_wgrad_in_sample_chunks():
    """The wgrad GEMM runs over chunks of whole samples when its transposed im2col image would leave the 2 GiB descriptor range
    (B = 64 at 64x64 on the AFHQ UNet does); forced here with a tiny threshold in a child process: 5 samples -> chunks 2 + 2 + 1."""
    import os
    import subprocess
    import sys
    code = """
import sys, os
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, 'tests'))
import numpy as np, torch, torch.nn.functional as F
from ddmi_amd import train as T
from ddmi_amd import initweights as iw
from helpers import rel_err
g = lambda k, s, std=1.0: torch.from_numpy(iw.normal('tcw', k, s, 0.0, std))
for numerics, tol in (('fp32', 2e-5), ('bf16', 4e-2)):
    tc = T.TrainContext(numerics, 'cuda:0')
    B, cin, cout, hw = 5, 32, 64, 8
    x, w, b = g('x', (B, cin, hw, hw)), g('w', (cout, cin, 3, 3), 1 / np.sqrt(cin * 9)), g('b', (cout,), 0.1)
    xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
    yr = F.conv2d(xr, wr, br, padding=1)
    dy = g('dy', tuple(yr.shape))
    yr.backward(dy)
    xd = x.permute(0, 2, 3, 1).contiguous().to('cuda:0', tc.dtype).requires_grad_(True)
    wd, bd = w.clone().cuda().requires_grad_(True), b.clone().cuda().requires_grad_(True)
    y = T.ConvFn.apply(tc, xd, wd, bd, None, None, 1, False)
    y.backward(dy.permute(0, 2, 3, 1).contiguous().to('cuda:0', tc.dtype))
    assert rel_err(wd.grad, wr.grad) < tol, (numerics, rel_err(wd.grad, wr.grad))
    assert rel_err(xd.grad.float().permute(0, 3, 1, 2).cpu(), xr.grad) < tol
print('chunked wgrad ok')
""" % (ROOT, ROOT)
    # DDMI_WGRAD_GEMM: keep the bf16 leg on the im2col + GEMM formulation too (its default here is the TN kernel)
    env = dict(os.environ, DDMI_WGRAD_CHUNK_BYTES="50000", DDMI_WGRAD_GEMM="1")   # 2 samples x 64 pixels x 288 rows x 4 B = 147 KB > 50 KB -> chunks
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "chunked wgrad ok" in out.stdout, out.stdout + out.stderr


def test_main_train_then_gen_end_to_end(golden, tmp_path):
    """`python main.py --exp ldm --configs <yaml>` with data.mode train (stage-1 checkpoint -> 4 optimizer steps on synthetic
    latents -> ldm-last.pt in the reference's layout) and then data.mode gen from that checkpoint: the reference's two modes
    chained through its own file formats."""
    import os
    import subprocess
    import sys
    import yaml
    from ddmi_amd.autoencoder import decoder_manifest
    from ddmi_amd.mlp import mlp_manifest
    cfgs = golden.json("configs.json")
    save = str(tmp_path)
    tt = lambda d: {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in d.items()}
    torch.save({"model": tt(named_state("e2e.vae", decoder_manifest(cfgs["TOY_DD"], cfgs["TOY_EMBED"]))), "mlp": tt(named_state("e2e.mlp", mlp_manifest()))},
               os.path.join(save, "model-last.pt"))
    base = {"model": {"embed_dim": cfgs["TOY_EMBED"], "lr": 1e-4, "pretrained": False, "resume": False,
                      "params": {"unetconfig": cfgs["TOY_UNET"], "ddconfig": cfgs["TOY_DD"], "mlpconfig": cfgs["MLP_CFG"],
                                 "ddpmconfig": {"timesteps": 1000, "sampling_timesteps": 5, "ddim_sampling_eta": 0.0, "image_size": 8, "channels": 64},
                                 "lossconfig": {"gradient_accumulate_every": 2, "ema_decay": 0.9999, "ema_update_every": 1}}},
            "data": {"domain": "image", "mode": "train", "save_pth": save, "test_batch_size": 2, "batch_size": 4, "test_resolution": 32, "train_steps": 8}}
    def run(cfg, name):
        path = os.path.join(save, name)
        with open(path, "w") as f:
            yaml.safe_dump(cfg, f)
        out = subprocess.run([sys.executable, os.path.join(ROOT, "main.py"), "--exp", "ldm", "--configs", path, "--numerics", "fp32"],
                             capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
        return out.stdout
    log = run(base, "train.yaml")
    assert "Dae loss" in log and "after 4 optimizer steps" in log            # 8 micro-steps / gradient_accumulate_every 2
    ck = torch.load(os.path.join(save, "ldm-last.pt"), map_location="cpu", weights_only=True)
    assert {"diffusion", "ema", "vaemodel", "mlp", "step"} <= set(ck) and "ema_model.mixing_logit" in ck["ema"]
    gen = dict(base, data=dict(base["data"], mode="gen"), model=dict(base["model"], pretrained=True))
    log = run(gen, "gen.yaml")
    assert "Loading Pretrained Models!" in log and "Finished generating images!" in log
    img = np.load(os.path.join(save, "results", "generation.npy"))
    assert img.shape == (2, 32, 32, 3) and img.dtype == np.uint8 and img.std() > 0
