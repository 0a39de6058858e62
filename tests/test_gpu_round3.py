"""Round-3 parity tests on the GPU (through the C ABI): the ResBlock skip_connection folded into the 3x3 kernel's K loop,
the XCD-aware attention order at odd batch sizes, the trainer's EMA against the ema_pytorch restatement."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from ddmi_amd import _lib, initweights as iw
from helpers import TOL_BF16, TOL_FP32, assert_bf16_close, named_state, rel_err, rel_l2
from oracle import ddmi_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _dev(a):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32).to(DEV).contiguous()


def conv_skip_op(x, w, b, xs1, xs2, ws, bs, numerics, splitk=0):
    L = _lib.lib()
    _lib.init_device(0)
    B, Cin, H, W = x.shape
    Cout, C2, C3 = w.shape[0], xs1.shape[1], 0 if xs2 is None else xs2.shape[1]
    d = [_dev(t) for t in (x, w, b, xs1, ws, bs)]
    d2 = _dev(xs2) if xs2 is not None else None
    y = torch.empty((B, Cout, H, W), device=DEV, dtype=torch.float32)
    _lib.check(L.ddmi_op_conv2d_skip(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), d2.data_ptr() if d2 is not None else None,
                                     d[4].data_ptr(), d[5].data_ptr(), y.data_ptr(), B, Cin, H, W, Cout, C2, C3, _lib.numerics_code(numerics), splitk))
    return y.cpu()


SKIP_CASES = [
    # B, Cin (= Cout of the block), H, W, Cout, C2, C3, splitk
    (2, 224, 16, 16, 224, 448, 224, 0),   # output block: cat(h 448, skip 224); x2's 7 blocks are whole, x3's last one is half
    (1, 224, 64, 64, 224, 224, 224, 0),   # 64x64 level: TR = 4 rows per tile; C2 = 224 ends in a HALF block next to x3's columns
    (5, 128, 8, 8, 128, 96, 160, 0),      # 8x8 level: four samples per tile, batch tail (5 samples), halo<128>, ragged blocks on both sources
    (3, 448, 32, 32, 448, 224, 0, 0),     # input-side block (one skip source), 224 -> 448
    (2, 192, 16, 16, 224, 160, 96, 2),    # split-K 2: the skip blocks ride on split 0
    (1, 64, 32, 32, 224, 64, 0, 0),       # a single block of x: the skip halo is fetched behind the ONLY block's taps
]


@pytest.mark.parametrize("case", SKIP_CASES)
@pytest.mark.parametrize("numerics", ["fp32", "bf16"])
def test_conv3x3_with_folded_skip_connection(case, numerics):
    """`skip_connection(x) + out_layers(...)` (openaimodel.py:272-275) as one launch vs torch's two convolutions."""
    B, Cin, H, W, Cout, C2, C3, splitk = case
    tag = "skipfold%s" % (case,)
    x = iw.normal(tag, "x", (B, Cin, H, W))
    w = iw.normal(tag, "w", (Cout, Cin, 3, 3), 0.0, 1.0 / math.sqrt(Cin * 9))
    b = iw.normal(tag, "b", (Cout,), 0.0, 0.1)
    xs1 = iw.normal(tag, "xs1", (B, C2, H, W))
    xs2 = iw.normal(tag, "xs2", (B, C3, H, W)) if C3 else None
    ws = iw.normal(tag, "ws", (Cout, C2 + C3), 0.0, 1.0 / math.sqrt(C2 + C3))
    bs = iw.normal(tag, "bs", (Cout,), 0.0, 0.1)
    xs = torch.from_numpy(xs1) if xs2 is None else torch.cat([torch.from_numpy(xs1), torch.from_numpy(xs2)], dim=1)
    ref = F.conv2d(torch.from_numpy(x), torch.from_numpy(w), torch.from_numpy(b), padding=1) + \
        F.conv2d(xs, torch.from_numpy(ws)[:, :, None, None], torch.from_numpy(bs))
    got = conv_skip_op(x, w, b, xs1, xs2, ws, bs, numerics, splitk)
    if numerics == "fp32":
        assert rel_err(got, ref) < 2e-5
    else:
        assert rel_err(got, ref) < 1.5e-2 and rel_l2(got, ref) < 1e-2, (rel_err(got, ref), rel_l2(got, ref))
        # the fold leaves the arithmetic of each product unchanged: the skip term alone (zero 3x3 weights) and the 3x3 term alone
        # (zero skip weights) equal the two separate bf16 launches of the same library up to the order of the fp32 additions
        z = np.zeros_like(w)
        only_skip = conv_skip_op(x, z, b, xs1, xs2, ws, bs, numerics, splitk)
        ref_skip = F.conv2d(xs, torch.from_numpy(ws)[:, :, None, None], torch.from_numpy(bs + b))
        assert rel_err(only_skip, ref_skip) < 1.5e-2


@pytest.mark.parametrize("B", [1, 3, 5, 9, 66])
def test_attention_block_order_at_odd_batch_sizes(B):
    """The bf16 attention kernel deals workgroups to XCDs by sample (B % 8 == 0 or B >= 64), by (sample, head) or in plain order:
    every order computes the same function (QKVAttentionLegacy, openaimodel.py:397-422)."""
    heads, d, T = 3, 32, 256
    qkv = torch.from_numpy(iw.normal("attn_order", f"qkv{B}", (B, heads * 3 * d, T)))
    ref = O.qkv_attention_legacy(qkv, heads)
    out = torch.empty((B, heads * d, T), device=DEV, dtype=torch.float32)
    _lib.init_device(0)
    _lib.check(_lib.lib().ddmi_op_attention_legacy_hd(qkv.to(DEV).data_ptr(), out.data_ptr(), B, heads, d, T, _lib.numerics_code("bf16")))
    assert rel_err(out, ref) < 2e-2


# ---------------- training: EMA (ema_pytorch semantics) and resume ---------------------------------------------------
def _g(tag, key, shape, std=1.0):
    return torch.from_numpy(iw.normal(tag, key, shape, 0.0, std))


def _toy_trainer(golden, numerics="fp32", **kw):
    from ddmi_amd import train as T
    cfg = golden.json("configs.json")["TOY_UNET"]
    tr = T.LDMTrainStep(cfg, dict(timesteps=1000, channels=64), numerics=numerics, lr=1e-4, weight_decay=0.0, device=DEV, **kw)
    st = {"model." + k: v for k, v in named_state("T1.unet", [(k, tuple(p.shape)) for k, p in tr.model.state_dict().items()]).items()}
    st["mixing_logit"] = iw.normal("T1", "mixing_logit", (1, 64, 1, 1), 0.0, 0.5)
    return tr.load_state_dict(st)


def test_trainer_ema_follows_ema_pytorch(golden):
    """`ema.update()` after every optimizer step (tools/ldm/image.py:174) with ema_pytorch's rules: copies while the call count is
    <= update_after_step, then the warmed-up decay 1 - (1 + epoch)^(-2/3) capped at beta - against the oracle's restatement of
    that class driven with snapshots of the same online weights."""
    tr = _toy_trainer(golden, ema_decay=0.9, ema_update_every=2).finalize()
    tr.ema_update_after_step = 5          # the real default (100) with the same code path, kept short
    orc = O.EmaPytorchOracle([tr.flat.data.detach().cpu().clone()], beta=0.9, update_every=2, update_after_step=5)
    x0, nz = _g("T1", "x_start", (2, 64, 8, 8)), _g("T1", "noise", (2, 64, 8, 8))
    saw_lerp = False
    for i in range(24):
        tr.train_step(x0, torch.tensor([3 + 17 * i, 900 - 31 * i]), nz)
        orc.online[0].copy_(tr.flat.data.detach().cpu())
        orc.update()
        assert rel_err(tr.ema, orc.ema[0]) < 1e-6, i
        assert tr.ema_step == int(orc.step) and tr.ema_initted == bool(orc.initted)
        saw_lerp |= bool(orc.initted) and not torch.equal(orc.ema[0], orc.online[0])
    assert saw_lerp                       # the run left the copy phase: the EMA lags the online weights
    # the decay reached its cap: epoch = 24 - 1 - 5 = 18 -> 1 - 19^(-2/3) = 0.86 < 0.9; at the default beta it would still be warming up
    from ddmi_amd.train import ema_pytorch_plan
    assert ema_pytorch_plan(22, 0.9, 2, 5) == ("lerp", pytest.approx(1 - 18 ** (-2 / 3)))
    assert ema_pytorch_plan(100, 0.9999, 10)[0] == "copy" and ema_pytorch_plan(110, 0.9999, 10)[0] == "lerp"


def test_training_resume_restores_ema_optimizer_and_counters(golden, tmp_path):
    """LDMTrainer.save -> load (resume, tools/ldm/image.py:104-134): a trainer restored from the file continues EXACTLY like the
    one that wrote it - weights, EMA, AdamW moments and bias-correction step - and the file has the reference's layout:
    `diffusion` = DDPM.state_dict() incl. the schedule buffers, `dae_opt` loadable by torch.optim.AdamW, ema_pytorch's keys."""
    import os
    from ddmi_amd import checkpoint as CK
    x0, nz = _g("T1", "x_start", (2, 64, 8, 8)), _g("T1", "noise", (2, 64, 8, 8))
    ts = lambda i: torch.tensor([3 + 17 * i, 900 - 31 * i])
    a = _toy_trainer(golden, ema_update_every=1).finalize()
    a.ema_update_after_step = 1
    for i in range(4):
        a.train_step(x0, ts(i), nz)
    a.epoch = 2                                                            # the caller's epoch loop owns the reference's `step`
    path = os.path.join(tmp_path, "ldm-last.pt")
    a.save(path)
    ck = torch.load(path, map_location="cpu", weights_only=True)          # plain containers and tensors only
    assert {"args", "step", "current_iters", "vaemodel", "mlp", "diffusion", "dae_opt", "ema", "scaler"} <= set(ck)
    assert ck["diffusion"]["betas"].shape == (1000,) and "ema_model.betas" in ck["ema"] and "online_model.mixing_logit" in ck["ema"]
    # `step` = finished EPOCHS, `current_iters` = micro-steps (tools/ldm/image.py:127-128, 145, 166); optimizer steps live in dae_opt only
    assert int(ck["ema"]["step"]) == 4 and bool(ck["ema"]["initted"]) and ck["step"] == 2 and ck["current_iters"] == 4
    # the reference's `self.dae_opt.load_state_dict` on `AdamW(diffusion_process.parameters())` of a REAL nn.Module with the same
    # structure (a direct `mixing_logit` parameter + a `model` submodule): index 0 is mixing_logit, and every moment lands on a
    # parameter of its own shape
    class Holder(torch.nn.Module):
        def __init__(self, named):
            super().__init__()
            self.model = torch.nn.ParameterDict()      # registered first, like DDPM.model (ddpm.py:124)
            for k, p in named:
                if k.startswith("model."):
                    self.model[k[6:].replace(".", "__")] = torch.nn.Parameter(torch.zeros_like(p.detach().cpu()))
            self.mixing_logit = torch.nn.Parameter(torch.zeros_like(dict(named)["mixing_logit"].detach().cpu()))
    ref_mod = Holder(list(a.named_parameters()))
    params = list(ref_mod.parameters())
    assert params[0] is ref_mod.mixing_logit
    opt = torch.optim.AdamW(params, lr=1e-4, weight_decay=0.0)
    opt.load_state_dict(ck["dae_opt"])
    for p_ in params:
        assert opt.state[p_]["exp_avg"].shape == p_.shape and float(opt.state[p_]["step"]) == 4.0
    off = dict((k, o) for k, _, o in a.optimizer_order())["mixing_logit"]
    assert torch.equal(opt.state[params[0]]["exp_avg"].reshape(-1), a.flat.exp_avg[off:off + params[0].numel()].cpu())
    opt.step()                                                             # shapes line up: the first step does not raise
    assert CK.checkpoint_sampler_config(ck) == {"timesteps": 1000}
    b = _toy_trainer(golden, ema_update_every=1)
    b.load_state_dict({k: v for k, v in ck["diffusion"].items() if k.startswith("model.") or k == "mixing_logit"}, strict=True).finalize()
    b.ema_update_after_step = 1
    b.load_training_state(CK.read_checkpoint(path), resume=True)
    assert b.opt_step == 4 and b.ema_step == 4 and b.ema_initted and b.epoch == 2 and b.current_iters == 4 and b.micro == 0
    assert torch.equal(b.flat.data, a.flat.data) and torch.equal(b.ema, a.ema)
    assert torch.equal(b.flat.exp_avg, a.flat.exp_avg) and torch.equal(b.flat.exp_avg_sq, a.flat.exp_avg_sq)
    for i in range(4, 6):
        a.train_step(x0, ts(i), nz)
        b.train_step(x0, ts(i), nz)
    assert torch.equal(b.flat.data, a.flat.data) and torch.equal(b.ema, a.ema)
    # `pretrained` (tools/ldm/image.py:76-82) restores weights + EMA only: a fresh optimizer
    c = _toy_trainer(golden, ema_update_every=1)
    c.load_state_dict({k: v for k, v in ck["diffusion"].items() if k.startswith("model.") or k == "mixing_logit"}, strict=True).finalize()
    c.load_training_state(ck, resume=False)
    assert c.opt_step == 0 and float(c.flat.exp_avg.abs().max()) == 0.0 and torch.equal(c.ema.cpu(), torch.cat([ck["ema"]["ema_model." + k].reshape(-1) for k, _ in c.named_parameters()]))


# ---------------- training at FULL size ---------------------------------------------------------------------------------
AFHQ_FULL_GRADS = ["model.input_blocks.0.0.weight", "model.input_blocks.1.0.in_layers.2.weight", "model.input_blocks.1.0.emb_layers.1.weight",
                   "model.input_blocks.3.0.op.weight", "model.input_blocks.4.0.skip_connection.weight", "model.input_blocks.4.1.norm.weight",
                   "model.input_blocks.4.1.qkv.weight", "model.input_blocks.4.1.proj_out.weight", "model.middle_block.1.qkv.bias",
                   "model.output_blocks.2.2.conv.weight", "model.output_blocks.5.0.out_layers.0.bias", "model.output_blocks.11.0.skip_connection.weight",
                   "model.time_embed.0.weight", "model.out.2.weight", "mixing_logit"]


def test_afhq_training_step_full_size_vs_autograd():
    """The configuration `bench.py --workload train` times - the 519 M-parameter AFHQ UNet on [B,64,64,64] latents - at B = 2:
    `p_losses` + backward through the C ABI's backward kernels against torch-CPU autograd of the oracle's forward (oracle.p_losses,
    itself pinned to the real reference's gradients by fixture T1).  fp32 mode: loss and EVERY gradient (norm and sum; fifteen tensors
    element by element) within 1e-3; bf16 mode: loss within the bf16 tolerance and relative L2 <= 3e-2 per gradient."""
    import time
    from ddmi_amd import train as T
    from helpers import AFHQ_UNET
    B = 2
    x0, nz = _g("afhq.train", "x_start", (B, 64, 64, 64)), _g("afhq.train", "noise", (B, 64, 64, 64))
    t = torch.tensor([37, 811], dtype=torch.int64)
    tr = T.LDMTrainStep(AFHQ_UNET, dict(timesteps=1000, channels=64), numerics="fp32", device=DEV)
    st = named_state("afhq.unet", [(k, tuple(p.shape)) for k, p in tr.model.state_dict().items()])
    logit_np = iw.normal("afhq.train", "mixing_logit", (1, 64, 1, 1), 0.0, 0.5)
    # oracle side
    t0 = time.time()
    p = {k: torch.from_numpy(np.ascontiguousarray(v)).requires_grad_(True) for k, v in st.items()}
    logit = torch.from_numpy(logit_np.copy()).requires_grad_(True)
    loss_ref = O.p_losses(p, AFHQ_UNET, O.make_schedule(1000), logit, x0, t, nz)
    loss_ref.backward()
    ref = {"model." + k: v.grad for k, v in p.items()}
    ref["mixing_logit"] = logit.grad
    print("oracle: full-size p_losses + autograd at B=2 in %.1f s" % (time.time() - t0))
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
        g, rn = g32[k].double(), float(r.double().norm())   # (a float32 norm over 10 M elements is itself only good to ~1e-3)
        e_norm = abs(float(g.norm()) - rn) / (rn + 1e-3 * l2max)
        e_sum = abs(float(g.sum()) - float(r.double().sum())) / (rn * np.sqrt(g.numel()) + 1e-3 * l2max)
        e_l2 = float((g - r.double()).norm()) / (rn + 1e-3 * l2max)   # relative L2, floored for the (analytically) tiny gradients
        if max(e_norm, e_sum, e_l2) >= TOL_FP32:
            bad[k] = (e_norm, e_sum, e_l2, rn)
    assert not bad, (len(bad), sorted(bad.items(), key=lambda kv: -kv[1][2])[:10])
    for k in AFHQ_FULL_GRADS:
        assert float((g32[k] - ref[k]).abs().max()) < TOL_FP32 * float(ref[k].abs().max()) + 1e-6 * l2max, k
    del tr, g32
    torch.cuda.empty_cache()
    tr16 = T.LDMTrainStep(AFHQ_UNET, dict(timesteps=1000, channels=64), numerics="bf16", device=DEV)
    loss16, g16 = run(tr16)
    assert abs(loss16 - float(loss_ref)) < TOL_BF16 * float(loss_ref)
    bad = {k: rel_l2(g16[k], r) for k, r in ref.items() if float(r.double().norm()) > 1e-4 * l2max and rel_l2(g16[k], r) > 3e-2}
    assert not bad, sorted(bad.items(), key=lambda kv: -kv[1])[:8]


@pytest.mark.parametrize("case", [(224, 224, 64, 0), (1344, 1344, 8, 1), (448, 1344, 32, 2)])
def test_wgrad_rows_are_batch_invariant_at_the_bench_batch(case):
    """`wgrad_tn_kernel` at the training bench's batch (B = 128: pixel slices over > 2^31 bytes of im2col-equivalent input) against
    itself at B = 2: the batch is the B = 2 pair repeated 64 times, so dW(128) = 64 dW(2) and dX rows repeat - for a 3x3 conv at 64x64,
    the 1344-channel Upsample conv and a 1x1 qkv conv."""
    from ddmi_amd import train as T
    cin, cout, hw, kind = case
    tc = T.TrainContext("bf16", DEV)
    k = 1 if kind == 2 else 3
    x2 = _g("wgrow", f"x{case}", (2, cin, hw, hw))
    w = _g("wgrow", f"w{case}", (cout, cin, k, k), 1.0 / math.sqrt(cin * k * k))
    b = _g("wgrow", f"b{case}", (cout,), 0.1)
    ho = hw * 2 if kind == 1 else hw
    dy2 = _g("wgrow", f"dy{case}", (2, cout, ho, ho))
    outs = []
    for reps in (1, 64):
        xd = x2.repeat(reps, 1, 1, 1).permute(0, 2, 3, 1).contiguous().to(DEV, tc.dtype).requires_grad_(True)
        wd, bd = w.clone().to(DEV).requires_grad_(True), b.clone().to(DEV).requires_grad_(True)
        y = T.ConvFn.apply(tc, xd, wd, bd, None, None, 1, kind == 1)
        y.backward(dy2.repeat(reps, 1, 1, 1).permute(0, 2, 3, 1).contiguous().to(DEV, tc.dtype))
        outs.append((wd.grad.float().cpu() / reps, bd.grad.float().cpu() / reps, xd.grad.float().cpu()))
        del xd, y
        torch.cuda.empty_cache()
    (w2, b2, dx2), (w128, b128, dx128) = outs
    assert rel_l2(w128, w2) < 1e-3 and rel_err(w128, w2) < 2e-3, (rel_l2(w128, w2), rel_err(w128, w2))
    assert rel_l2(b128, b2) < 1e-3
    for r in (0, 63, 127):   # dX is stored in bf16 and the B = 128 launch runs other tiles / splits: equal up to the last bf16 bit
        assert rel_err(dx128[r], dx2[r % 2]) < 8e-3 and rel_l2(dx128[r], dx2[r % 2]) < 2e-3, r


# ---------------- bf16x3: the third numerics mode ------------------------------------------------------------------------
X3_CONV_CASES = [(2, 64, 8, 8, 224, 3, 1, 0), (1, 224, 16, 16, 224, 3, 1, 0), (2, 96, 8, 8, 192, 3, 2, 0), (2, 64, 8, 8, 64, 3, 1, 1),
                 (2, 448, 8, 8, 224, 1, 1, 0), (1, 128, 32, 32, 128, 3, 1, 0), (1, 512, 8, 8, 512, 3, 1, 0), (2, 672, 64, 64, 224, 3, 1, 0)]


@pytest.mark.parametrize("case", X3_CONV_CASES)
def test_conv2d_bf16x3_is_fp32_accurate(case):
    """DDMI_PARITY_BF16X3: x.w as xh.wh + xl.wh + xh.wl on the bf16 MFMA kernels (K widened to 3 Cin), fp32 accumulate and finish:
    within 1e-4 of torch's fp32 convolution where plain bf16 is at 1e-2 - stride 2, fused upsample, 1x1, split-K, the halo kernel."""
    B, Cin, H, W, Cout, k, stride, ups = case
    tag = "convx3%s" % (case,)
    x = iw.normal(tag, "x", (B, Cin, H, W))
    w = iw.normal(tag, "w", (Cout, Cin, k, k), 0.0, 1.0 / math.sqrt(Cin * k * k))
    b = iw.normal(tag, "b", (Cout,), 0.0, 0.1)
    xt = torch.from_numpy(x)
    if ups:
        xt = F.interpolate(xt, scale_factor=2, mode="nearest")
    ref = F.conv2d(xt.double(), torch.from_numpy(w).double(), torch.from_numpy(b).double(), stride=stride, padding=k // 2)
    L = _lib.lib()
    _lib.init_device(0)
    xd, wd, bd = _dev(x), _dev(w), _dev(b)
    y = torch.empty(tuple(ref.shape), device=DEV, dtype=torch.float32)
    for splitk in (0, 2):
        _lib.check(L.ddmi_op_conv2d(xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), y.data_ptr(), B, Cin, H, W, Cout, k, stride, ups,
                                    _lib.numerics_code("bf16x3"), splitk))
        assert rel_err(y, ref) < 1e-4 and rel_l2(y, ref) < 3e-5, (splitk, rel_err(y, ref), rel_l2(y, ref))


def test_afhq_networks_full_size_bf16x3_vs_oracle():
    """The full-size AFHQ UNet, decoder and fused MLP in the bf16x3 mode against the CPU oracle: the north star's 1e-3."""
    from ddmi_amd.autoencoder import Autoencoder
    from ddmi_amd.coords import generation_coords
    from ddmi_amd.mlp import MLP
    from ddmi_amd.unet import UNetModel
    from helpers import AFHQ_DD, AFHQ_EMBED, AFHQ_UNET, MLP_CFG
    unet = UNetModel(**AFHQ_UNET, numerics="bf16x3", max_batch=2)
    st = named_state("afhq.unet", unet.manifest())
    unet.load_state_dict(st)
    x = iw.normal("afhq", "x", (1, 64, 64, 64))
    t = torch.tensor([437], dtype=torch.int64)
    ref = O.unet_forward(O.to_state(st), AFHQ_UNET, torch.from_numpy(x), t)
    eps = unet(torch.from_numpy(x).to(DEV), t.to(DEV))
    assert rel_err(eps, ref) < TOL_FP32 and rel_l2(eps, ref) < TOL_FP32, (rel_err(eps, ref), rel_l2(eps, ref))
    xx = torch.from_numpy(np.concatenate([x, x])).to(DEV)        # B = 2 = max_batch: the autotuned plans
    eps2 = unet(xx, t.expand(2).contiguous().to(DEV))
    assert rel_err(eps2[1:2], ref) < TOL_FP32
    del unet
    torch.cuda.empty_cache()
    z = iw.normal("afhq", "z", (1, 64, 64, 64))
    vae = Autoencoder(AFHQ_DD, AFHQ_EMBED, numerics="bf16x3", max_batch=1)
    vst = named_state("afhq.vae", vae.manifest())
    vae.load_state_dict(vst)
    ref_planes = O.vae_decode(O.to_state(vst), AFHQ_DD, torch.from_numpy(z))
    planes = vae.decode(torch.from_numpy(z).to(DEV))
    for p, r in zip(planes, ref_planes):
        assert rel_err(p, r) < TOL_FP32 and rel_l2(p, r) < TOL_FP32
    mlp = MLP(**MLP_CFG, numerics="bf16x3")
    mst = named_state("afhq.mlp", mlp.manifest())
    mlp.load_state_dict(mst)
    ref_rgb = O.image_mlp(O.to_state(mst), O.generate_coords(256), ref_planes, 1.0)
    rgb = mlp(generation_coords(256), hdbf=planes, si=1.0)
    assert rel_err(rgb, ref_rgb) < TOL_FP32


# ---------------- the reference's compiled operator as a standalone HIP op -----------------------------------------------
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_fused_bias_act_matches_the_reference_kernel_semantics(dtype):
    """`fused.fused_bias_act(input, bias, refer, act, grad, alpha, scale)` (op/fused_bias_act.cpp:4-20, fused_bias_act_kernel.cu:19-49)
    on 2-D, 4-D and 5-D inputs (bias along dim 1), every (act, grad) pair of the kernel's switch, empty bias / refer - against the
    switch restated with torch ops (oracle.fused_lrelu is the (3, 0) case pinned by fixture M1)."""
    from ddmi_amd.fused_act import fused
    g = torch.Generator().manual_seed(11)
    tol = {torch.float32: 1e-6, torch.bfloat16: 8e-3, torch.float16: 1e-3}[dtype]
    for shape in [(3, 256), (2, 66, 5, 7), (1, 8, 3, 4, 5)]:
        x = torch.randn(shape, generator=g)
        b = torch.randn(shape[1], generator=g)
        ref = torch.randn(shape, generator=g)
        xd, bd, rd = (t.to(DEV, dtype) for t in (x, b, ref))
        x, b, ref = xd.float().cpu(), bd.float().cpu(), rd.float().cpu()          # the values the kernel sees
        z = x + b.view(1, -1, *([1] * (len(shape) - 2)))
        empty = xd.new_empty(0)
        want = {(3, 0): torch.where(z > 0, z, 0.2 * z), (3, 1): torch.where(ref > 0, z, 0.2 * z), (3, 2): torch.zeros_like(z),
                (1, 0): z, (1, 1): z, (1, 2): torch.zeros_like(z), (7, 0): z}
        for (act, grad), w in want.items():
            got = fused.fused_bias_act(xd, bd, rd if grad == 1 else empty, act, grad, 0.2, 2 ** 0.5)
            assert got.dtype == dtype and got.shape == xd.shape
            assert rel_err(got.float(), w * 2 ** 0.5) < tol, (shape, act, grad)
        nob = fused.fused_bias_act(xd, empty, empty, 3, 0, 0.2, 1.0)
        assert rel_err(nob.float(), torch.where(x > 0, x, 0.2 * x)) < tol
        if len(shape) == 2:   # the oracle's restatement is written for the MLP's [points, channels] activations
            assert rel_err(fused.fused_bias_act(xd, bd, empty, 3, 0, 0.2, 2 ** 0.5).float(), O.fused_lrelu(x, b)) < tol


def test_fused_leaky_relu_module_and_its_gradients():
    """`FusedLeakyReLU` / `fused_leaky_relu` (op/fused_act.py:61-99) on the HIP op: value, first derivatives w.r.t. input and bias and
    the double backward (gradient of a gradient penalty) against torch autograd of leaky_relu(x + b) * sqrt 2."""
    from ddmi_amd.fused_act import FusedLeakyReLU
    g = torch.Generator().manual_seed(3)
    x = torch.randn((2, 16, 6, 5), generator=g).to(DEV).requires_grad_(True)
    m = FusedLeakyReLU(16).to(DEV)
    with torch.no_grad():
        m.bias.copy_(torch.randn(16, generator=g))
    xr = x.detach().clone().requires_grad_(True)
    br = m.bias.detach().clone().requires_grad_(True)
    y = m(x)
    yr = F.leaky_relu(xr + br.view(1, -1, 1, 1), 0.2) * 2 ** 0.5
    assert rel_err(y, yr) < 1e-6
    w = torch.randn(tuple(y.shape), generator=g).to(DEV)
    (gx,) = torch.autograd.grad((y * w).sum(), x, create_graph=True)
    (gxr,) = torch.autograd.grad((yr * w).sum(), xr, create_graph=True)
    assert rel_err(gx, gxr) < 1e-6
    (gx ** 2).sum().backward()          # a gradient penalty: needs the op's double backward
    (gxr ** 2).sum().backward()
    assert rel_err(x.grad, xr.grad) < 1e-5
    y2 = m(x.detach())
    y2.sum().backward()
    F.leaky_relu(xr.detach() + br.view(1, -1, 1, 1), 0.2).mul(2 ** 0.5).sum().backward()
    assert rel_err(m.bias.grad, br.grad) < 1e-5


def test_occupancy_training_step_vs_autograd():
    """The occupancy LDM trainer's step (tools/ldm/occupancy.py:139-180) is the image one on z = cat(z_xy, z_xz, z_yz) [B,192,16,16]
    through the same UNetModel class (configs/ldm/shapenet.yaml: model_channels 256, attention at 16x16 / 8x8 / 4x4, down to 2x2):
    full size, B = 4, loss and every gradient against oracle autograd in fp32 mode, relative L2 <= 3e-2 per gradient in bf16."""
    import os
    from ddmi_amd import train as T
    from ddmi_amd.config import load_ldm_args
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    a = load_ldm_args(os.path.join(ROOT, "configs", "ldm", "shapenet.yaml"))
    ucfg = {k: v for k, v in dict(a.unetconfig).items() if k not in ("size1", "size2", "size3")}
    B = 4   # (the bf16 wgrad GEMM wants B * Ho * Wo to be a multiple of 8: at this net's 2x2 level an even batch)
    x0, nz = _g("occ.train", "x_start", (B, 192, 16, 16)), _g("occ.train", "noise", (B, 192, 16, 16))
    t = torch.tensor([5, 480, 993, 250], dtype=torch.int64)
    logit_np = iw.normal("occ.train", "mixing_logit", (1, 192, 1, 1), 0.0, 0.5)
    st = None
    ref = None
    for numerics in ("fp32", "bf16"):
        tr = T.LDMTrainStep(ucfg, dict(timesteps=1000, channels=192), numerics=numerics, device=DEV)
        if st is None:
            st = named_state("shapenet.unet", [(k, tuple(p.shape)) for k, p in tr.model.state_dict().items()])
            p = {k: torch.from_numpy(np.ascontiguousarray(v)).requires_grad_(True) for k, v in st.items()}
            logit = torch.from_numpy(logit_np.copy()).requires_grad_(True)
            loss_ref = O.p_losses(p, ucfg, O.make_schedule(1000), logit, x0, t, nz)
            loss_ref.backward()
            ref = {"model." + k: v.grad for k, v in p.items()}
            ref["mixing_logit"] = logit.grad
            l2max = max(float(g.double().norm()) for g in ref.values())
        full = {"model." + k: v for k, v in st.items()}
        full["mixing_logit"] = logit_np
        tr.load_state_dict(full).finalize()
        loss = float(tr.loss_and_backward(x0, t, nz))
        got = {k: pp.grad.detach().float().cpu() for k, pp in tr.named_parameters()}
        if numerics == "fp32":
            assert abs(loss - float(loss_ref)) < TOL_FP32 * float(loss_ref)
            bad = {k: float((got[k].double() - r.double()).norm()) / (float(r.double().norm()) + 1e-3 * l2max) for k, r in ref.items()}
            bad = {k: v for k, v in bad.items() if v >= TOL_FP32}
            assert not bad, sorted(bad.items(), key=lambda kv: -kv[1])[:8]
        else:
            assert abs(loss - float(loss_ref)) < TOL_BF16 * float(loss_ref)
            bad = {k: rel_l2(got[k], r) for k, r in ref.items() if float(r.double().norm()) > 1e-4 * l2max and rel_l2(got[k], r) > 3e-2}
            assert not bad, sorted(bad.items(), key=lambda kv: -kv[1])[:8]
        del tr
        torch.cuda.empty_cache()


@pytest.mark.parametrize("case", [(40, 448, 32, 32, 1344), (9, 1344, 8, 8, 1344), (3, 224, 64, 64, 224), (1, 96, 16, 16, 448),
                                  (6, 256, 32, 32, 768, 25), (2, 512, 16, 64, 512, 25)])
def test_persistent_1x1_kernel(case):
    """Configuration 24 (`conv_gemm1x1_kernel`: one workgroup per CU walks XCD-ordered tiles, the next tile's stages land under the
    previous tile's epilogue, bias from LDS) on grids larger and smaller than the CU count, a ragged last M tile (9 x 64 rows), a
    K that ends in a half block (224, 96) and both tile orders (m-major / n-major by operand size) - against torch."""
    B, Cin, H, W, Cout = case[:5]
    cfg = case[5] if len(case) > 5 else 24          # 25: the 256 x 128 instantiation (channel counts that are multiples of 128)
    tag = "gemm1x1%s" % (case,)
    x = iw.normal(tag, "x", (B, Cin, H, W))
    w = iw.normal(tag, "w", (Cout, Cin, 1, 1), 0.0, 1.0 / math.sqrt(Cin))
    b = iw.normal(tag, "b", (Cout,), 0.0, 0.1)
    ref = F.conv2d(torch.from_numpy(x), torch.from_numpy(w), torch.from_numpy(b))
    L = _lib.lib()
    _lib.init_device(0)
    xd, wd, bd = _dev(x), _dev(w), _dev(b)
    y = torch.empty(tuple(ref.shape), device=DEV, dtype=torch.float32)
    try:
        _lib.check(L.ddmi_debug_conv_config(cfg))
        _lib.check(L.ddmi_op_conv2d(xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), y.data_ptr(), B, Cin, H, W, Cout, 1, 1, 0, _lib.numerics_code("bf16"), 1))
    finally:
        L.ddmi_debug_conv_config(-1)
    assert rel_err(y, ref) < 1.5e-2 and rel_l2(y, ref) < 1e-2, (rel_err(y, ref), rel_l2(y, ref))
    assert float((y.cpu() - ref).abs().max()) < 0.03 * float(ref.abs().max())


@pytest.mark.parametrize("B,heads,side", [(3, 2, 8), (8, 3, 32), (5, 14, 16), (16, 2, 16)])
def test_attention_backward_from_the_forward_lse(B, heads, side, monkeypatch):
    """The one-pass attention backward (k_attn_bwd.hip): the forward hands over the row log-sum-exp, dQ / dK / dV come from two
    kernels with the query / the key on the MFMA lane.  Against the oracle's QKVAttentionLegacy under autograd (fp32), against the
    three-pass kernels it replaces, bit-reproducible; batches in plain order (B = 3, 5) and dealt to the XCDs by sample (8, 16);
    one tile (T = 64: 64 rows per workgroup), 256 and 1024 tokens (128 rows per workgroup)."""
    from ddmi_amd import train as T
    tc = T.TrainContext("bf16", DEV)
    d, tok = 32, side * side
    tag = "attnbwd%s" % ((B, heads, side),)
    qkv = torch.from_numpy(iw.normal(tag, "qkv", (B, heads * 3 * d, tok))).to(torch.bfloat16).float()
    dout = torch.from_numpy(iw.normal(tag, "dout", (B, heads * d, tok))).to(torch.bfloat16).float()
    qr = qkv.clone().requires_grad_(True)
    out_r = O.qkv_attention_legacy(qr, heads)
    out_r.backward(dout)

    def run():
        qd = qkv.reshape(B, -1, side, side).permute(0, 2, 3, 1).contiguous().to(DEV, torch.bfloat16).requires_grad_(True)
        out = T.AttentionLegacyFn.apply(tc, qd, heads)
        out.backward(dout.reshape(B, -1, side, side).permute(0, 2, 3, 1).contiguous().to(DEV, torch.bfloat16))
        return (out.detach().float().permute(0, 3, 1, 2).reshape(B, heads * d, tok).cpu(),
                qd.grad.float().permute(0, 3, 1, 2).reshape(B, heads * 3 * d, tok).cpu())

    assert _lib.lib().ddmi_train_attn_lse_ok(tc.h, tok, d) == 1
    out, dq = run()
    out2, dq2 = run()
    assert torch.equal(dq, dq2) and torch.equal(out, out2)          # no atomics anywhere
    assert rel_l2(out, out_r.detach()) < 6e-3
    assert rel_l2(dq, qr.grad) < 1.2e-2 and rel_err(dq, qr.grad) < TOL_BF16
    # the q, k and v blocks separately (a wrong dK hides behind a large dV in a norm over all three)
    g = dq.reshape(B, heads, 3, d, tok)
    gr = qr.grad.reshape(B, heads, 3, d, tok)
    for part in range(3):
        assert rel_l2(g[:, :, part], gr[:, :, part]) < 1.5e-2, part
    monkeypatch.setenv("DDMI_ATTN_BWD_LEGACY", "1")
    out_l, dq_l = run()
    assert torch.equal(out_l, out)                                   # the forward kernel is the same, with or without the lse output
    assert rel_l2(dq, dq_l) < 1.2e-2
    # fp32 handles keep the three-pass kernels
    assert _lib.lib().ddmi_train_attn_lse_ok(T.TrainContext("fp32", DEV).h, tok, d) == 0


@pytest.mark.parametrize("B,cin,cout,hw", [(2, 64, 224, 16), (3, 448, 448, 8), (2, 224, 96, 32)])
def test_training_upsample_conv_parity_form(B, cin, cout, hw):
    """Upsample.conv of the training path (bias only): the forward runs the parity form of nearest x2 + 3x3 - four 2x2
    convolutions on the source grid, weights pre-summed in fp32 from the master weights each step (2x2 halo kernel where the shape
    allows, Cout = 96 falls back) - the backward stays on the 3x3 weights.  Output and all gradients against torch."""
    from ddmi_amd import train as T
    tc = T.TrainContext("bf16", DEV)
    tag = "upt%s" % ((B, cin, cout, hw),)
    x = torch.from_numpy(iw.normal(tag, "x", (B, cin, hw, hw))).to(torch.bfloat16).float()
    w = torch.from_numpy(iw.normal(tag, "w", (cout, cin, 3, 3), 0.0, 1.0 / math.sqrt(cin * 9)))
    b = torch.from_numpy(iw.normal(tag, "b", (cout,), 0.0, 0.1))
    dy = torch.from_numpy(iw.normal(tag, "dy", (B, cout, 2 * hw, 2 * hw))).to(torch.bfloat16).float()
    xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
    yr = F.conv2d(F.interpolate(xr, scale_factor=2, mode="nearest"), wr, br, padding=1)
    yr.backward(dy)
    xd = x.permute(0, 2, 3, 1).contiguous().to(DEV, torch.bfloat16).requires_grad_(True)
    wd, bd = w.clone().to(DEV).requires_grad_(True), b.clone().to(DEV).requires_grad_(True)
    y = T.ConvFn.apply(tc, xd, wd, bd, None, None, 1, True)
    got = y.detach().float().permute(0, 3, 1, 2).cpu()
    assert got.shape == yr.shape
    assert rel_l2(got, yr.detach()) < 6e-3 and rel_err(got, yr.detach()) < TOL_BF16
    y.backward(dy.permute(0, 2, 3, 1).contiguous().to(DEV, torch.bfloat16))
    assert rel_l2(xd.grad.float().permute(0, 3, 1, 2).cpu(), xr.grad) < 1e-2
    assert rel_l2(wd.grad.cpu(), wr.grad) < 1e-2 and rel_l2(bd.grad.cpu(), br.grad) < 1e-2


@pytest.mark.parametrize("numerics", ["fp32", "bf16"])
def test_block_input_gradients_summed_in_the_groupnorm_backward(numerics, monkeypatch):
    """ResBlock / AttentionBlock hand their input to the skip path THROUGH the GroupNorm op (GroupNormFn passthrough), whose backward
    adds the skip path's gradient in its last pass (ddmi_train_gn_bwd_res).  Same loss and gradients as with autograd adding the two
    (DDMI_TRAIN_FUSE_RES=0 semantics: FUSE_RESIDUAL_GRAD False): a ResBlock with a 1x1 skip, one with the identity, an AttentionBlock."""
    from ddmi_amd import train as T
    tc = T.TrainContext(numerics, DEV)
    B, hw = 2, 8
    blocks = [T.ResBlock(64, 96, 96), T.ResBlock(96, 96, 96), T.AttentionBlock(96, 32)]
    for i, m in enumerate(blocks):
        for k, p in m.named_parameters():
            p.data = torch.from_numpy(iw.normal("fuse_res%d" % i, k, tuple(p.shape), 0.0, 0.05) + (1.0 if k.endswith("0.weight") and p.dim() == 1 else 0.0)).float()
        m.to(DEV)
    x0 = torch.from_numpy(iw.normal("fuse_res", "x", (B, hw, hw, 64))).to(DEV, tc.dtype)
    embs = [torch.from_numpy(iw.normal("fuse_res", "e%d" % i, (B, 96), 0.0, 0.3)).to(DEV) for i in range(2)]
    dy = torch.from_numpy(iw.normal("fuse_res", "dy", (B, hw, hw, 96))).to(DEV, tc.dtype)

    def run(fused):
        monkeypatch.setattr(T, "FUSE_RESIDUAL_GRAD", fused)
        for m in blocks:
            m.zero_grad(set_to_none=True)
        x = x0.clone().requires_grad_(True)
        h = blocks[0].run(tc, x * 1.0, embs[0])      # (x * 1.0: the block input is a non-leaf, as inside the UNet)
        h = blocks[1].run(tc, h, embs[1])
        h = blocks[2].run(tc, h)
        h.backward(dy)
        return h.detach().float().cpu(), x.grad.float().cpu(), {f"{i}.{k}": p.grad.float().cpu() for i, m in enumerate(blocks) for k, p in m.named_parameters() if p.grad is not None}   # (emb_layers run outside the block)

    y1, gx1, gp1 = run(True)
    y0, gx0, gp0 = run(False)
    assert torch.equal(y1, y0)
    tol = 2e-6 if numerics == "fp32" else 1e-2      # bf16: the fused sum is rounded once, autograd's twice
    assert rel_l2(gx1, gx0) < tol
    assert len(gp0) >= 14 and set(gp0) == set(gp1)
    for k in gp0:
        assert rel_l2(gp1[k], gp0[k]) < tol, k


def test_attention_lse_entry_points_refuse_what_they_do_not_serve():
    """ddmi_train_attn_fwd_lse / _bwd_lse exist for bf16 handles, head dim 32, token counts that are multiples of 64; everything else is
    refused with DDMI_ERR_INVALID and a message (the plain pair serves it) - no silent fallback inside the library."""
    from ddmi_amd import train as T
    L = _lib.lib()
    tc16, tc32 = T.TrainContext("bf16", DEV), T.TrainContext("fp32", DEV)
    assert L.ddmi_train_attn_lse_ok(tc16.h, 256, 32) == 1
    assert L.ddmi_train_attn_lse_ok(tc16.h, 100, 32) == 0 and L.ddmi_train_attn_lse_ok(tc16.h, 256, 64) == 0
    assert L.ddmi_train_attn_lse_ok(tc32.h, 256, 32) == 0
    B, heads, tok, d = 1, 2, 100, 32
    qkv = torch.zeros((B, tok, heads * 3 * d), dtype=torch.bfloat16, device=DEV)
    out = torch.zeros((B, tok, heads * d), dtype=torch.bfloat16, device=DEV)
    lse = torch.zeros((B, heads, tok), dtype=torch.float32, device=DEV)
    rc = L.ddmi_train_attn_fwd_lse(tc16.h, qkv.data_ptr(), out.data_ptr(), lse.data_ptr(), B, tok, heads, d, None)
    assert rc != 0 and b"multiple of 64" in L.ddmi_last_error()
    rc = L.ddmi_train_attn_bwd_lse(tc16.h, qkv.data_ptr(), out.data_ptr(), out.data_ptr(), lse.data_ptr(), qkv.data_ptr(), B, tok, heads, d, None)
    assert rc != 0
    rc = L.ddmi_train_attn_fwd_lse(tc16.h, qkv.data_ptr(), out.data_ptr(), None, B, 128, heads, d, None)
    assert rc != 0 and b"null" in L.ddmi_last_error()


@pytest.mark.parametrize("numerics", ["fp32", "bf16"])
def test_decoder_concat_op_and_its_gradient(numerics):
    """CatFn (ddmi_train_cat2 / ddmi_train_split2): the decoder's channel concatenation on NHWC tensors and its gradient are exact
    copies - equal to torch.cat and to slicing, bit for bit, with unequal channel counts and a row count that is not a block multiple."""
    from ddmi_amd import train as T
    tc = T.TrainContext(numerics, DEV)
    a = torch.from_numpy(iw.normal("cat2", "a", (3, 5, 7, 224))).to(DEV, tc.dtype).requires_grad_(True)
    b = torch.from_numpy(iw.normal("cat2", "b", (3, 5, 7, 448))).to(DEV, tc.dtype).requires_grad_(True)
    out = T.CatFn.apply(tc, a, b)
    assert torch.equal(out, torch.cat([a, b], dim=-1))
    g = torch.from_numpy(iw.normal("cat2", "g", (3, 5, 7, 672))).to(DEV, tc.dtype)
    out.backward(g)
    assert torch.equal(a.grad, g[..., :224]) and torch.equal(b.grad, g[..., 224:])
    rc = _lib.lib().ddmi_train_cat2(tc.h, a.data_ptr(), b.data_ptr(), out.data_ptr(), 105, 223, 448, None)
    assert rc != 0 and b"16-byte" in _lib.lib().ddmi_last_error()
