"""Round-5 GPU parity tests (through the C ABI)."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from ddmi_amd import _lib, initweights as iw
from helpers import rel_err, rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _dev(a):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32).to(DEV).contiguous()


def conv_op(x, w, b, numerics="bf16", splitk=1, cfg=-1, half=-1):
    L = _lib.lib()
    _lib.init_device(0)
    B, Cin, H, W = x.shape
    Cout, k = w.shape[0], w.shape[2]
    xd, wd, bd = _dev(x), _dev(w), _dev(b)
    y = torch.empty((B, Cout, H, W), device=DEV, dtype=torch.float32)
    _lib.check(L.ddmi_debug_conv_config(cfg))
    _lib.check(L.ddmi_debug_halo_half(half))
    try:
        _lib.check(L.ddmi_op_conv2d(xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), y.data_ptr(), B, Cin, H, W, Cout, k, 1, 0,
                                    _lib.numerics_code(numerics), splitk))
    finally:
        L.ddmi_debug_conv_config(-1)
        L.ddmi_debug_halo_half(-1)
    return y.cpu()


# (B, Cin, H, W, Cout): Cin = 64 k + r with r in 1..32 - the AFHQ UNet's 224 and 672, the ragged block as the ONLY block (32, 16), a
# 24-channel tail, 96 = 64 + 32; every W the halo kernel serves, the 8x8 level's four samples per tile past the batch end, a
# non-square plane, a Cout tail on both halo widths
HALF_CASES = [(2, 224, 64, 64, 224), (1, 672, 32, 32, 448), (3, 224, 16, 16, 224), (5, 96, 8, 8, 224), (1, 32, 32, 32, 128),
              (2, 16, 16, 16, 64), (1, 88, 32, 32, 160), (3, 96, 8, 16, 96)]


@pytest.mark.parametrize("case", HALF_CASES)
def test_halo_two_taps_per_stage_tail_is_bit_identical_to_the_plain_form(case):
    """k_conv.hip HALF instantiation (openaimodel.py's 224-wide convolutions: Cin = 3.5 blocks of 64): the ragged last channel block
    runs two taps per stage.  Same products, same accumulation order: the output equals the plain halo kernel's BIT FOR BIT, and
    torch's fp32 convolution within the bf16 bound."""
    B, Cin, H, W, Cout = case
    tag = "half%s" % (case,)
    x = iw.normal(tag, "x", (B, Cin, H, W))
    w = iw.normal(tag, "w", (Cout, Cin, 3, 3), 0.0, 1.0 / math.sqrt(Cin * 9))
    b = iw.normal(tag, "b", (Cout,), 0.0, 0.1)
    ref = F.conv2d(torch.from_numpy(x), torch.from_numpy(w), torch.from_numpy(b), padding=1)
    for cfg in (19, 20):   # halo<224> / halo<128>
        on = conv_op(x, w, b, cfg=cfg, half=1)
        off = conv_op(x, w, b, cfg=cfg, half=0)
        assert torch.equal(on, off), (cfg, case, float((on - off).abs().max()))
        assert rel_err(on, ref) < 1.5e-2, (cfg, case)
        assert float((on - ref).abs().max()) < 0.03 * float(ref.abs().max()), (cfg, case)
    # split-K launches keep the plain kernel (the tail form is a no-split-K instantiation): still right with the switch on
    if Cin >= 128:
        assert rel_err(conv_op(x, w, b, cfg=19, half=1, splitk=2), ref) < 1.5e-2


# ---- producer-side GroupNorm (k_conv.hip conv_epilogue_fgn) --------------------------------------------------------------------------
def conv_gn_op(x, w, b, emb, gamma, beta, eps=1e-5, silu=1, numerics="bf16", fuse=-1, iters=1):
    import ctypes as C
    L = _lib.lib()
    _lib.init_device(0)
    B, Cin, H, W = x.shape
    Cout = w.shape[0]
    args = [_dev(x), _dev(w), _dev(b), _dev(emb) if emb is not None else None, _dev(gamma), _dev(beta)]
    y = torch.empty((B, Cout, H, W), device=DEV, dtype=torch.float32)
    used = C.c_int(-1)
    _lib.check(L.ddmi_debug_gn_fuse(fuse))
    try:
        _lib.check(L.ddmi_op_conv_gn(*[a.data_ptr() if a is not None else None for a in args], y.data_ptr(), B, Cin, H, W, Cout, eps, silu,
                                     _lib.numerics_code(numerics), iters, C.byref(used)))
    finally:
        L.ddmi_debug_gn_fuse(-1)
    return y.cpu(), used.value


def _conv_gn_ref(x, w, b, emb, gamma, beta, eps, silu):
    h = F.conv2d(torch.from_numpy(x).double(), torch.from_numpy(w).double(), torch.from_numpy(b).double(), padding=1)
    if emb is not None:
        h = h + torch.from_numpy(emb).double()[:, :, None, None]
    y = F.group_norm(h, 32, torch.from_numpy(gamma).double(), torch.from_numpy(beta).double(), eps)
    return (F.silu(y) if silu else y).float()


# (B, Cin, H, W, Cout, emb, silu): the three fusable levels of the AFHQ UNet - 64x64 (16 tiles per sample exchange granules, 32 groups in
# the one channel tile; Cin 224 takes the two-taps-per-stage tail as well), 32x32 (4 tiles, 16 groups per tile, two channel tiles),
# 16x16 (one tile per sample: no exchange) - at batches that are and are not multiples of 8 (XCD-aligned tile ranges or not), with and
# without the embedding addend / SiLU; 1344 channels (42 per group: no whole groups per 224-wide tile) and 8x8 must fall back
FGN_CASES = [(2, 224, 64, 64, 224, True, 1), (8, 64, 64, 64, 224, False, 1), (3, 448, 32, 32, 448, True, 1), (16, 224, 32, 32, 448, True, 0),
             (5, 896, 16, 16, 896, True, 1), (9, 448, 16, 16, 896, False, 1),
             (2, 64, 32, 32, 256, True, 1), (3, 128, 16, 16, 512, True, 1)]   # 128- / 256-wide channel tiles (the model_channels = 256 networks)


@pytest.mark.parametrize("case", FGN_CASES)
def test_groupnorm_in_the_convolution_epilogue_matches_the_separate_pass_and_torch(case):
    """openaimodel.py:261-270 (h = in_layers(x) + emb_out; out_layers: GroupNorm32 -> SiLU): the pair as ONE launch.  Against torch
    (fp64 reference) within the bf16 bound, against the two-launch form (which rounds h to bf16 before normalising: the fused form is the
    closer one), and bit-identical from run to run (every tile reduces the sample's partials in the same order)."""
    B, Cin, H, W, Cout, use_emb, silu = case
    tag = "fgn%s" % (case,)
    x = iw.normal(tag, "x", (B, Cin, H, W))
    w = iw.normal(tag, "w", (Cout, Cin, 3, 3), 0.0, 1.0 / math.sqrt(Cin * 9))
    b = iw.normal(tag, "b", (Cout,), 0.0, 0.1)
    emb = iw.normal(tag, "e", (B, Cout), 0.0, 0.5) if use_emb else None
    gamma = iw.normal(tag, "g", (Cout,), 1.0, 0.2)
    beta = iw.normal(tag, "bt", (Cout,), 0.0, 0.2)
    ref = _conv_gn_ref(x, w, b, emb, gamma, beta, 1e-5, silu)
    fused, used = conv_gn_op(x, w, b, emb, gamma, beta, silu=silu, fuse=2, iters=3)
    assert used == 1, "the launch did not take the fused epilogue"
    plain, used0 = conv_gn_op(x, w, b, emb, gamma, beta, silu=silu, fuse=0)
    assert used0 == 0
    e_f, e_p = rel_err(fused, ref), rel_err(plain, ref)
    assert e_f < 1.5e-2 and e_p < 1.5e-2, (case, e_f, e_p)
    assert rel_l2(fused, ref) <= rel_l2(plain, ref) * 1.05 + 1e-4, (case, rel_l2(fused, ref), rel_l2(plain, ref))
    again, _ = conv_gn_op(x, w, b, emb, gamma, beta, silu=silu, fuse=2, iters=2)
    assert torch.equal(fused, again), case
    f32, used32 = conv_gn_op(x, w, b, emb, gamma, beta, silu=silu, numerics="fp32", fuse=2)
    assert used32 == 0 and rel_err(f32, ref) < 1e-4      # the parity mode keeps the reference's op order


# 42 channels per group (no whole groups per 224-wide tile), 8x8 (several samples per tile), and the decoder's 512 channels at 64x64: 16 tiles
# per sample x 4 channel tiles = 64 workgroups that would wait for each other - more than an XCD holds (conv_fgn_ok's progress bound)
@pytest.mark.parametrize("case", [(2, 448, 8, 8, 1344), (2, 64, 16, 16, 1344), (3, 64, 8, 8, 224), (1, 64, 64, 64, 512)])
def test_groupnorm_fusion_falls_back_where_a_tile_does_not_hold_whole_groups(case):
    B, Cin, H, W, Cout = case
    tag = "fgnfb%s" % (case,)
    x = iw.normal(tag, "x", (B, Cin, H, W))
    w = iw.normal(tag, "w", (Cout, Cin, 3, 3), 0.0, 1.0 / math.sqrt(Cin * 9))
    b = iw.normal(tag, "b", (Cout,), 0.0, 0.1)
    gamma = iw.normal(tag, "g", (Cout,), 1.0, 0.2)
    beta = iw.normal(tag, "bt", (Cout,), 0.0, 0.2)
    y, used = conv_gn_op(x, w, b, None, gamma, beta, fuse=2)
    assert used == 0
    assert rel_err(y, _conv_gn_ref(x, w, b, None, gamma, beta, 1e-5, 1)) < 1.5e-2


def test_groupnorm_fusion_hand_off_under_repetition_and_uneven_batches():
    """The granule hand-off between the tiles of a sample under what hides ordering bugs on an idle chip (MI355X_MICROARCH.md: test every
    hand-off under uneven load, checking every word): many back-to-back launches (stale tags of the previous launch in the same
    granules), batches that fill the chip several times over and do not align with the XCD tile ranges."""
    for B in (33, 70):
        tag = "fgnrep%d" % B
        x = iw.normal(tag, "x", (B, 64, 64, 64))
        w = iw.normal(tag, "w", (224, 64, 3, 3), 0.0, 1.0 / math.sqrt(64 * 9))
        b = iw.normal(tag, "b", (224,), 0.0, 0.1)
        emb = iw.normal(tag, "e", (B, 224), 0.0, 0.5)
        gamma = iw.normal(tag, "g", (224,), 1.0, 0.2)
        beta = iw.normal(tag, "bt", (224,), 0.0, 0.2)
        first, used = conv_gn_op(x, w, b, emb, gamma, beta, fuse=2, iters=1)
        assert used == 1
        many, _ = conv_gn_op(x, w, b, emb, gamma, beta, fuse=2, iters=25)
        assert torch.equal(first, many), B
        plain, _ = conv_gn_op(x, w, b, emb, gamma, beta, fuse=0)
        assert rel_err(first, plain) < 1.5e-2, B


# ---- the skip_connection on a concatenated input as one dual-source launch (k_conv.hip CAT) ------------------------------------------
# (B, C1, C2, H, W, Cout): the AFHQ output blocks' (448 | 224), (224 | 224) with the ragged first source (224 = 3.5 blocks of 64),
# deeper levels, a 128-wide network (tri-plane UNet), rows past the last whole tile, a shape the kernel does not serve (fallback)
CAT_CASES = [(2, 448, 224, 64, 64, 224, 1), (3, 224, 224, 32, 32, 224, 1), (2, 896, 448, 16, 16, 448, 1), (5, 1344, 1344, 8, 8, 1344, 1),
             (3, 512, 256, 8, 16, 256, 1), (1, 72, 88, 16, 16, 224, 1), (2, 64, 64, 8, 8, 96, 0)]


@pytest.mark.parametrize("case", CAT_CASES)
def test_skip_connection_over_a_concatenated_input_in_one_launch(case):
    import ctypes as C
    B, C1, C2, H, W, Cout, expect_cat = case
    tag = "cat%s" % (case,)
    x1 = iw.normal(tag, "x1", (B, C1, H, W))
    x2 = iw.normal(tag, "x2", (B, C2, H, W))
    w = iw.normal(tag, "w", (Cout, C1 + C2), 0.0, 1.0 / math.sqrt(C1 + C2))
    b = iw.normal(tag, "b", (Cout,), 0.0, 0.1)
    ref = F.conv2d(torch.cat([torch.from_numpy(x1), torch.from_numpy(x2)], 1), torch.from_numpy(w)[:, :, None, None], torch.from_numpy(b))
    L = _lib.lib()
    _lib.init_device(0)
    for numerics, tol in (("bf16", 1.5e-2), ("fp32", 2e-5)):
        y = torch.empty((B, Cout, H, W), device=DEV, dtype=torch.float32)
        used = C.c_int(-1)
        args = [_dev(x1), _dev(x2), _dev(w), _dev(b)]
        _lib.check(L.ddmi_op_conv1x1_cat(*[a.data_ptr() for a in args], y.data_ptr(), B, C1, C2, H, W, Cout, _lib.numerics_code(numerics),
                                         C.byref(used)))
        assert used.value == (expect_cat if numerics == "bf16" else 0), (case, numerics)
        assert rel_err(y.cpu(), ref) < tol, (case, numerics)
        assert float((y.cpu() - ref).abs().max()) < 2 * tol * float(ref.abs().max()), (case, numerics)


# ---- the 256-wide halo tile (configuration 26: per-wave 128 x 64) for the model_channels = 256 networks ---------------------------------
# (B, Cin, H, W, Cout): W = 32 / 16 whole-sample tiles, a non-square plane (video xt / yt: 16 x 32), a Cout tail (320 = 256 + 64), the
# ragged Cin the two-tap tail takes, a batch whose last tile is the only one of its XCD range; 8 x 8 and W = 64 must fall back
WIDE_CASES = [(2, 256, 32, 32, 256), (3, 128, 16, 16, 512), (2, 96, 16, 32, 1024), (1, 64, 32, 32, 320), (9, 32, 16, 16, 256)]


@pytest.mark.parametrize("case", WIDE_CASES)
def test_halo_256_wide_tile(case):
    B, Cin, H, W, Cout = case
    tag = "wide%s" % (case,)
    x = iw.normal(tag, "x", (B, Cin, H, W))
    w = iw.normal(tag, "w", (Cout, Cin, 3, 3), 0.0, 1.0 / math.sqrt(Cin * 9))
    b = iw.normal(tag, "b", (Cout,), 0.0, 0.1)
    ref = F.conv2d(torch.from_numpy(x), torch.from_numpy(w), torch.from_numpy(b), padding=1)
    got = conv_op(x, w, b, cfg=26)
    assert rel_err(got, ref) < 1.5e-2, case
    assert float((got - ref).abs().max()) < 0.03 * float(ref.abs().max()), case
    # the same products in the same order as the 128-wide tile: bit-identical
    assert torch.equal(got, conv_op(x, w, b, cfg=20)), case
    for other in ((2, 64, 8, 8, 256), (1, 64, 64, 64, 256)):     # shapes it does not serve fall back to the 128-wide tile
        Bo, Ci, Ho, Wo, Co = other
        xo = iw.normal(tag, "xo%d" % Ho, (Bo, Ci, Ho, Wo))
        wo = iw.normal(tag, "wo%d" % Ho, (Co, Ci, 3, 3), 0.0, 1.0 / math.sqrt(Ci * 9))
        bo = iw.normal(tag, "bo%d" % Ho, (Co,), 0.0, 0.1)
        ro = F.conv2d(torch.from_numpy(xo), torch.from_numpy(wo), torch.from_numpy(bo), padding=1)
        assert rel_err(conv_op(xo, wo, bo, cfg=26), ro) < 1.5e-2, other


@pytest.mark.parametrize("cfg", [26, 27])
@pytest.mark.parametrize("case", [(3, 64, 8, 8, 256), (2, 128, 16, 16, 512), (5, 64, 8, 16, 256)])
def test_wide_tile_epilogue_with_embedding_rows_and_channel_statistics(cfg, case):
    """The 128-row wave tile of the 256-wide configurations takes the fast epilogue as two 64-row halves: at 8x8 a wave's rows hold TWO
    samples (two embedding rows), and the consumer GroupNorm reads 64-row blocks of channel statistics.  conv + emb -> GroupNorm + SiLU
    through the network executor on the pinned configuration, against torch fp64 (cfg 26 falls back to the 128-wide halo tile at 8x8)."""
    B, Cin, H, W, Cout = case
    tag = "wideemb%s" % (case,)
    x = iw.normal(tag, "x", (B, Cin, H, W))
    w = iw.normal(tag, "w", (Cout, Cin, 3, 3), 0.0, 1.0 / math.sqrt(Cin * 9))
    b = iw.normal(tag, "b", (Cout,), 0.0, 0.1)
    emb = iw.normal(tag, "e", (B, Cout), 0.0, 0.7)
    gamma = iw.normal(tag, "g", (Cout,), 1.0, 0.2)
    beta = iw.normal(tag, "bt", (Cout,), 0.0, 0.2)
    ref = _conv_gn_ref(x, w, b, emb, gamma, beta, 1e-5, 1)
    L = _lib.lib()
    try:
        _lib.check(L.ddmi_debug_conv_config(cfg))
        y, used = conv_gn_op(x, w, b, emb, gamma, beta, fuse=0)
    finally:
        L.ddmi_debug_conv_config(-1)
    assert used == 0 and rel_err(y, ref) < 1.5e-2, (cfg, case, rel_err(y, ref))


def test_groupnorm_fusion_default_mode_fuses_exchanging_cohorts_only_at_batches_that_are_multiples_of_8():
    """conv_fgn_ok's speed rule: with B % 8 != 0 the XCD tile ranges cut through samples and the fused launch is slower than conv + pass
    (AFHQ UNet, B = 250: 65.9 -> 72.4 ms).  Default mode: B = 8 fuses, B = 9 keeps the pass; a 16x16 launch (one tile per sample: nothing to
    exchange) fuses at any batch.  Results right either way."""
    for B, Cin, H, Cout, expect in ((8, 64, 64, 224, 1), (9, 64, 64, 224, 0), (9, 64, 16, 896, 1)):
        tag = "fgnb8_%d_%d" % (B, H)
        x = iw.normal(tag, "x", (B, Cin, H, H))
        w = iw.normal(tag, "w", (Cout, Cin, 3, 3), 0.0, 1.0 / math.sqrt(Cin * 9))
        b = iw.normal(tag, "b", (Cout,), 0.0, 0.1)
        emb = iw.normal(tag, "e", (B, Cout), 0.0, 0.5)
        gamma = iw.normal(tag, "g", (Cout,), 1.0, 0.2)
        beta = iw.normal(tag, "bt", (Cout,), 0.0, 0.2)
        y, used = conv_gn_op(x, w, b, emb, gamma, beta, fuse=1)
        assert used == expect, (B, H, used)
        assert rel_err(y, _conv_gn_ref(x, w, b, emb, gamma, beta, 1e-5, 1)) < 1.5e-2, (B, H)


def test_unet_with_fused_groupnorm_handoff_eager_graph_and_unfused():
    """A UNet whose conv1 launches exchange GroupNorm statistics between the four tiles of a 32x32 sample (net.hip `attach_fused_groupnorm`,
    k_conv.hip `conv_epilogue_fgn`): the eager forward repeats bit for bit, the captured hipGraph - the epoch kernel that re-tags the
    granules is part of it - replays the same bits, and the result agrees with the network that keeps every GroupNorm a separate pass."""
    from ddmi_amd.unet import UNetModel
    from helpers import named_state
    cfg = dict(image_size=32, in_channels=16, model_channels=128, out_channels=16, attention_resolutions=[], num_res_blocks=1,
               channel_mult=[1, 2], num_head_channels=32)
    L = _lib.lib()
    _lib.init_device(0)
    x = torch.from_numpy(iw.normal("fgn_graph", "x", (8, 16, 32, 32))).to(DEV)
    t = torch.tensor([3, 50, 7, 120, 999, 0, 410, 64], dtype=torch.int64, device=DEV)
    outs = {}
    try:
        for mode in (0, 2):
            _lib.check(L.ddmi_debug_gn_fuse(mode))    # read when the network is planned and at every launch
            unet = UNetModel(**cfg, numerics="bf16", max_batch=8)
            unet.load_state_dict(named_state("fgn_graph.unet", unet.manifest()))
            side = torch.cuda.Stream()
            with torch.cuda.stream(side):
                ref = unet(x, t).clone()
                fused = unet.async_status()
                assert (fused > 0) == (mode == 2), (mode, fused)
                assert torch.equal(unet(x, t), ref)
                unet.set_graph(True)
                out = torch.empty_like(ref)
                for _ in range(4):   # capture, then three replays
                    unet(x, t, out=out)
                side.synchronize()
                assert torch.equal(out, ref), mode
                unet.async_status()      # no hand-off timed out
                unet.set_graph(False)
            outs[mode] = ref.float().cpu()
    finally:
        L.ddmi_debug_gn_fuse(-1)
    assert torch.isfinite(outs[2]).all()
    assert rel_l2(outs[2], outs[0]) < 1e-2   # the fused form normalises the fp32 accumulators, the pass their bf16 rounding
