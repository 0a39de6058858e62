"""Round-4 GPU parity tests (through the C ABI): the fused AttentionBlock launch (k_attnblock.hip) against the five-launch form
and the CPU oracle (openaimodel.py:279-325, 397-426 as restated in oracle.unet_attnblock)."""
import ctypes as C
import math

import numpy as np
import pytest
import torch

from ddmi_amd import _lib, initweights as iw
from helpers import TOL_FP32, assert_bf16_close, rel_err, rel_l2
from oracle import ddmi_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _dev(a):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32).to(DEV).contiguous()


def attention_block_op(x, p, head_channels=32, numerics="bf16", fused=-1):
    """x [B, C, H, W]; p: the block's state dict under 'blk.'; returns (y, fused_used)."""
    L = _lib.lib()
    _lib.init_device(0)
    B, Cc, H, W = x.shape
    args = [_dev(x), _dev(p["blk.norm.weight"]), _dev(p["blk.norm.bias"]), _dev(p["blk.qkv.weight"]), _dev(p["blk.qkv.bias"]),
            _dev(p["blk.proj_out.weight"]), _dev(p["blk.proj_out.bias"])]
    y = torch.empty((B, Cc, H, W), device=DEV, dtype=torch.float32)
    used = C.c_int(-1)
    _lib.check(L.ddmi_debug_attn_fused(fused))
    try:
        _lib.check(L.ddmi_op_attention_block(*[a.data_ptr() for a in args], y.data_ptr(), B, Cc, H, W, head_channels,
                                             _lib.numerics_code(numerics), C.byref(used)))
    finally:
        _lib.check(L.ddmi_debug_attn_fused(-1))
    return y.cpu(), used.value


def _block_params(tag, Cc):
    return {
        "blk.norm.weight": iw.normal(tag, "g", (Cc,), 1.0, 0.2),
        "blk.norm.bias": iw.normal(tag, "b", (Cc,), 0.0, 0.2),
        "blk.qkv.weight": iw.normal(tag, "wq", (3 * Cc, Cc, 1), 0.0, 1.6 / math.sqrt(Cc)),
        "blk.qkv.bias": iw.normal(tag, "bq", (3 * Cc,), 0.0, 0.2),
        "blk.proj_out.weight": iw.normal(tag, "wp", (Cc, Cc, 1), 0.0, 1.0 / math.sqrt(Cc)),
        "blk.proj_out.bias": iw.normal(tag, "bp", (Cc,), 0.0, 0.1),
    }


# (B, C, H): the three attention levels of the AFHQ UNet (1 / 4 / 16 samples per 1024-row workgroup), each with a batch that leaves
# the last group partly empty, one that takes the XCD-ordered grid (>= 8 groups, padded to a multiple of 8) and one below it
FUSED_CASES = [(3, 448, 32), (9, 448, 32), (5, 896, 16), (33, 896, 16), (17, 1344, 8), (130, 1344, 8), (2, 64, 8), (1, 128, 16)]


@pytest.mark.parametrize("case", FUSED_CASES)
def test_attention_block_fused_vs_five_launch_and_oracle(case):
    B, Cc, H = case
    tag = "attnblock%s" % (case,)
    p = _block_params(tag, Cc)
    x = iw.normal(tag, "x", (B, Cc, H, H), 0.3, 1.5)     # a non-zero mean: the GroupNorm in front matters
    ref = O.unet_attnblock(O.to_state(p), "blk.", torch.from_numpy(x), 32)
    y_f, used_f = attention_block_op(x, p, fused=1)
    y_u, used_u = attention_block_op(x, p, fused=0)
    assert used_f == 1 and used_u == 0
    # the residual x dominates the block output: compare the attention branch h = y - x
    h_ref = (ref - torch.from_numpy(x)).numpy()
    h_f, h_u = y_f.numpy() - x, y_u.numpy() - x
    assert_bf16_close(h_f, h_ref, what=tag + ".fused")
    assert_bf16_close(h_u, h_ref, what=tag + ".five_launch")
    # the two forms round the same bf16 tensors (x_hat, q, k, v, p) in different accumulation orders only
    assert rel_l2(h_f, h_u) < 1.0e-2, rel_l2(h_f, h_u)
    assert np.isfinite(y_f.numpy()).all()


def test_attention_block_fused_rows_are_batch_invariant():
    """Sample b of a batch = the same sample alone: groups of 4 samples per workgroup at 16x16 never mix keys across samples."""
    Cc, H = 896, 16
    tag = "attnblock_rows"
    p = _block_params(tag, Cc)
    x = iw.normal(tag, "x", (7, Cc, H, H), 0.0, 1.0)
    y_all, used = attention_block_op(x, p, fused=1)
    assert used == 1
    for b in (0, 3, 4, 6):
        y_one, _ = attention_block_op(x[b:b + 1], p, fused=1)
        # (the proj_out convolution may pick another tile / split at another batch: equal up to its accumulation order)
        assert rel_l2(y_all[b].numpy() - x[b], y_one[0].numpy() - x[b]) < 3e-3, b


def test_attention_block_unserved_shapes_fall_back():
    """fp32 numerics, other head dims and token counts keep the five-launch form (and stay within the parity tolerance)."""
    tag = "attnblock_fallback"
    p = _block_params(tag, 64)
    x = iw.normal(tag, "x", (2, 64, 8, 8))
    ref = O.unet_attnblock(O.to_state(p), "blk.", torch.from_numpy(x), 32)
    y, used = attention_block_op(x, p, numerics="fp32", fused=1)
    assert used == 0 and rel_err(y, ref) < TOL_FP32
    x2 = iw.normal(tag, "x2", (2, 64, 4, 8))          # 32 tokens
    y2, used2 = attention_block_op(x2, p, fused=1)
    assert used2 == 0
    ref2 = O.unet_attnblock(O.to_state(p), "blk.", torch.from_numpy(x2), 32)
    assert_bf16_close(y2.numpy() - x2, (ref2 - torch.from_numpy(x2)).numpy(), what=tag)
    y3, used3 = attention_block_op(x, p, head_channels=16, fused=1)
    assert used3 == 0


def test_autotuned_plans_agree_with_the_pinned_ones(tmp_path):
    """The parity suite runs plan-deterministic (DDMI_TUNE_STRICT=1, tests/conftest.py); the timing-based autotuner that bench.py and
    production finalizes use is exercised here in a child process: the toy UNet finalized with DDMI_TUNE_STRICT=0 writes a plan file,
    its bf16 output agrees with the pinned-plan run within the bf16 tolerance (other tiles / split-K: another rounding order, same
    network), and a second child that READS that plan file reproduces the first child's output bit for bit."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = r'''
import json, os, sys
import numpy as np, torch
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
from ddmi_amd.unet import UNetModel
from helpers import named_state
cfg = dict(image_size=16, in_channels=64, model_channels=64, out_channels=64, attention_resolutions=[2, 1], num_res_blocks=1,
           channel_mult=[1, 2], num_head_channels=32)
net = UNetModel(**cfg, numerics="bf16", max_batch=2, device="cuda:0")
net.load_state_dict(named_state("autotune.unet", net.manifest()))
net.finalize()
g = torch.Generator().manual_seed(3)
x = torch.randn((2, 64, 16, 16), generator=g).to("cuda:0")
y = net(x, torch.tensor([7, 400], device="cuda:0"))
np.save(sys.argv[1], y.float().cpu().numpy())
''' % (root, root)
    outs = {}
    for tag, env in (("strict", {"DDMI_TUNE_STRICT": "1"}),
                     ("tuned", {"DDMI_TUNE_STRICT": "0", "DDMI_TUNE_FILE": str(tmp_path / "plans.txt")}),
                     ("replayed", {"DDMI_TUNE_STRICT": "1", "DDMI_TUNE_FILE": str(tmp_path / "plans.txt")})):
        out = str(tmp_path / (tag + ".npy"))
        r = subprocess.run([sys.executable, "-c", script, out], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[tag] = np.load(out)
    plans = open(tmp_path / "plans.txt").read().splitlines()
    assert plans[0].startswith("v ") and len(plans) > 5                     # the autotuner ran and wrote one line per shape
    assert np.array_equal(outs["tuned"], outs["replayed"])                  # a plan file pins the result bit for bit
    assert_bf16_close(outs["tuned"], outs["strict"], what="autotuned vs pinned plans (toy UNet)", l2=2e-2)


def test_attention_block_fused_with_scores_beyond_the_bound():
    """The fused kernel drops the running-maximum bookkeeping of a round when |q| max|k| (Cauchy-Schwarz on the row norms its projection
    leaves behind) is within 2^100 of the first tile's maximum, and keeps it otherwise.  A qkv projection scaled up until the logits
    span several hundred (softmax close to one-hot, bound far above the threshold) must take the tracked path and still agree with the
    five-launch form and the oracle; the same block at ordinary scale takes the bounded path (covered by the cases above)."""
    B, Cc, H = 2, 448, 32
    tag = "attnblock_loose"
    p = _block_params(tag, Cc)
    p["blk.qkv.weight"] = (p["blk.qkv.weight"] * 6.0).astype(np.float32)      # logits x 36: |s| log2 e / sqrt(d) reaches several hundred
    x = iw.normal(tag, "x", (B, Cc, H, H), 0.0, 1.0)
    ref = O.unet_attnblock(O.to_state(p), "blk.", torch.from_numpy(x), 32)
    y_f, used = attention_block_op(x, p, fused=1)
    y_u, _ = attention_block_op(x, p, fused=0)
    assert used == 1 and np.isfinite(y_f.numpy()).all()
    h_ref = (ref - torch.from_numpy(x)).numpy()
    # near-one-hot softmax: a bf16 rounding of q or k can flip which key wins for a query, so the comparison is on the bulk - relative L2
    # of the attention branch against the oracle no worse than the five-launch form's, and the two device forms close to each other
    l2_f, l2_u = rel_l2(y_f.numpy() - x, h_ref), rel_l2(y_u.numpy() - x, h_ref)
    assert l2_f < max(2.0 * l2_u, 5e-2), (l2_f, l2_u)
    assert rel_l2(y_f.numpy() - x, y_u.numpy() - x) < max(2.0 * l2_u, 5e-2)


def test_block_gradients_are_final_when_the_block_input_hook_fires(golden):
    """What the overlapped gradient all-reduce (ddmi_amd.train.OverlappedGradReducer) relies on: the hook on a block's INPUT tensor
    fires after that block's backward, in slab-tail order (mixing_logit, out, output blocks reversed, middle, input blocks reversed),
    and at that moment the block's slab range already holds its FINAL gradient - except the `emb_layers` views, which the stacked
    embedding projection writes last.  Checked on the toy trainer by snapshotting the slab inside the callbacks of an eager step."""
    from test_gpu_round3 import _g, _toy_trainer
    from ddmi_amd.train import EmbProjFn
    tr = _toy_trainer(golden).finalize()
    order, ranges, late = tr.reduce_plan()
    assert order[0] == "mixing_logit" and order[1] == "out" and order[-1] == "input_blocks.1"
    assert all(b == c for (a, b), (c, d) in zip(ranges[1:], ranges[:-1]))      # contiguous, descending
    assert ranges[0][1] == tr.flat.n and len(late) == 2 * len(tr.model._resblocks())
    index = {k: i for i, k in enumerate(order)}
    seen, snaps = [], {}

    def cb(key):
        if key in index and key not in snaps:
            seen.append(key)
            snaps[key] = tr.flat.grad.detach().clone()
    late_mark = {}
    tr.model._block_grad_cb = cb
    EmbProjFn.pre_backward = lambda: late_mark.setdefault("n_seen", len(seen))
    try:
        x0, nz = _g("T1", "x_start", (2, 64, 8, 8)), _g("T1", "noise", (2, 64, 8, 8))
        tr.loss_and_backward(x0, torch.tensor([3, 900]), nz)
    finally:
        tr.model._block_grad_cb = None
        EmbProjFn.pre_backward = None
    torch.cuda.synchronize()
    final = tr.flat.grad.detach().clone()
    assert seen == [k for k in order if k != "mixing_logit"]          # every hooked block, in completion order
    assert late_mark["n_seen"] == len(seen)                           # the embedding projection's backward runs after the last hooked block
    keep = torch.ones(tr.flat.n, dtype=torch.bool, device=final.device)
    for a, b in late:
        keep[a:b] = False
    for key in seen:
        lo = ranges[index[key]][0]                                    # everything from this block to the end of the slab is final ...
        m = keep[lo:]
        assert torch.equal(snaps[key][lo:][m], final[lo:][m]), key
    first = snaps[seen[0]]
    assert all(float(first[a:b].abs().max()) == 0.0 for a, b in late)  # ... and the late views are still zero while the ranges go out
    assert any(float(final[a:b].abs().max()) > 0.0 for a, b in late)


@pytest.mark.parametrize("case", [(2, 448, 32), (3, 64, 8), (2, 128, 16)])
def test_attention_block_bf16x3_mode_keeps_the_parity_tolerance(case):
    """DDMI_PARITY_BF16X3 handles run the attention's two matrix products as three bf16 MFMA products each (attention_x3_kernel), like
    their convolutions: the AttentionBlock stays within the 1e-3 of the fp32 parity mode against the CPU oracle."""
    B, Cc, H = case
    tag = "attnblock_x3%s" % (case,)
    p = _block_params(tag, Cc)
    x = iw.normal(tag, "x", (B, Cc, H, H), 0.2, 1.0)
    ref = O.unet_attnblock(O.to_state(p), "blk.", torch.from_numpy(x), 32)
    y, used = attention_block_op(x, p, numerics="bf16x3", fused=1)
    assert used == 0                                   # the fused launch is a bf16-handle arrangement
    assert rel_err(y, ref) < TOL_FP32, rel_err(y, ref)
    h_ref = (ref - torch.from_numpy(x)).numpy()
    assert rel_l2(y.numpy() - x, h_ref) < TOL_FP32     # and on the attention branch alone (the residual does not hide it)


def _attn_mh_prescaled(q, k, v, heads, seg0, seg1):
    """q, k, v [B, heads*d, N] fp32 -> the bf16 network arrangement: q carries d^-1/2 log2(e) before it is rounded to bf16."""
    L = _lib.lib()
    _lib.init_device(0)
    B, Cc, N = q.shape
    d = Cc // heads
    qs = (torch.from_numpy(q) * (math.log2(math.e) / math.sqrt(d))).numpy()
    out = torch.empty((B, Cc, N), device=DEV, dtype=torch.float32)
    qd, kd, vd = _dev(qs), _dev(k), _dev(v)      # (held: a temporary's memory would be handed to the next one)
    _lib.check(L.ddmi_op_attention_mh_prescaled(qd.data_ptr(), kd.data_ptr(), vd.data_ptr(), out.data_ptr(), B, heads, d, N, seg0, seg1))
    return out.cpu()


# (B, heads, d, N, seg0, seg1): the tri-plane UNet's two resident levels (d = 16 over 1024 | 512 | 512 tokens, d = 32 over 256 | 128 | 128),
# one segment, a batch that does not fill the 8 XCD classes, and shapes the resident kernel must REFUSE (N not a multiple of 512, a segment
# that is not a multiple of 64, d = 64): those run the tiled kernel through the same entry point
ATTN_RES_CASES = [(2, 16, 16, 2048, 1024, 512), (3, 16, 32, 512, 256, 128), (9, 2, 16, 1024, 1024, 0), (1, 3, 32, 1024, 1024, 0),
                  (1, 2, 16, 1536, 1024, 256),   # d = 16, N not a multiple of 1024: the 16 x 16 x 32 resident form (the others: 32 x 32 x 16)
                  (2, 2, 32, 1024, 512, 256), (1, 2, 16, 1024, 448, 320), (1, 2, 32, 384, 128, 128), (2, 2, 64, 512, 256, 128),
                  # the STREAMING kernel (d = 64 / 128, N >= 2048): three segments, one segment, a batch of 8 (sample-per-XCD order),
                  # fewer than 8 (sample, head) pairs (plain order), and N = 2304 that d = 64 must refuse (not a multiple of 512)
                  (1, 8, 64, 4096, 2048, 1024), (2, 1, 128, 2048, 1024, 512), (8, 1, 64, 2048, 2048, 0), (1, 2, 128, 2560, 1280, 640),
                  (1, 2, 64, 2304, 1024, 640),
                  # d = 256 (32-key tiles, three stages) and d = 512 (two 64 KB stages), QF = 1
                  (1, 2, 256, 2048, 1024, 512), (1, 1, 512, 2048, 2048, 0), (2, 1, 512, 2176, 1152, 512),
                  # d = 32 beyond the resident kernel's 1024 tokens: streaming, waves 0-3 fetch K and waves 4-7 V
                  (2, 3, 32, 2048, 1024, 512), (1, 2, 32, 2560, 2560, 0)]


@pytest.mark.parametrize("case", ATTN_RES_CASES)
def test_resident_attention_matches_the_oracle_and_the_tiled_kernel(case, monkeypatch):
    """`attention_res_kernel` (one workgroup per (sample, head), K / V in LDS once, bounded tiles, the d = 16 two-block K fragments) and
    `attention_stream_kernel` (512 / 256 queries per workgroup, K / V tiles through a three-stage LDS-DMA ring) against the CPU oracle's
    softmax(q k^T / sqrt(d)) v and against the tiled kernel on the same pre-scaled bf16 inputs."""
    B, heads, d, N, seg0, seg1 = case
    tag = "attres%s" % (case,)
    q = iw.normal(tag, "q", (B, heads * d, N), 0.0, 1.2)
    k = iw.normal(tag, "k", (B, heads * d, N), 0.0, 1.2)
    v = iw.normal(tag, "v", (B, heads * d, N))
    q[0, 0:d, N // 3] *= 3.0                     # one query far above the first tile's reference
    k[B - 1, d:2 * d, N - 5] *= 6.0              # one key whose norm sets the Cauchy-Schwarz bound (and wins late in the sequence)
    ref = O.sdpa_heads(torch.from_numpy(q), torch.from_numpy(k), torch.from_numpy(v), heads)
    y = _attn_mh_prescaled(q, k, v, heads, seg0, seg1)
    assert rel_err(y, ref) < 2e-2, rel_err(y, ref)
    assert rel_l2(y.numpy(), ref.numpy()) < 1e-2
    import subprocess, sys, os
    # the tiled kernel in a fresh process (DDMI_ATTN_RES is read once per process)
    code = ("import sys, numpy as np, torch; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import test_gpu_round4 as T\n"
            "from ddmi_amd import initweights as iw\n"
            "case = %r; B, heads, d, N, seg0, seg1 = case; tag = 'attres%%s' %% (case,)\n"
            "q = iw.normal(tag, 'q', (B, heads * d, N), 0.0, 1.2); k = iw.normal(tag, 'k', (B, heads * d, N), 0.0, 1.2); v = iw.normal(tag, 'v', (B, heads * d, N))\n"
            "q[0, 0:d, N // 3] *= 3.0; k[B - 1, d:2 * d, N - 5] *= 6.0\n"
            "np.save(sys.argv[1], T._attn_mh_prescaled(q, k, v, heads, seg0, seg1).numpy())\n"
            % (os.path.dirname(os.path.abspath(__file__)), os.path.dirname(os.path.dirname(os.path.abspath(__file__))), case))
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "tiled.npy")
        env = dict(os.environ, DDMI_ATTN_RES="0", DDMI_ATTN_STREAM="0")   # (d >= 256: the generic kernel)
        r = subprocess.run([sys.executable, "-c", code, path], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        tiled = np.load(path)
    assert rel_l2(y.numpy(), tiled) < 6e-3, rel_l2(y.numpy(), tiled)   # same bf16 inputs, another reference maximum / summation order


@pytest.mark.parametrize("case", [(1, 2, 64, 2048, 1024, 512), (1, 1, 128, 2048, 2048, 0)])
def test_streaming_attention_second_pass_when_a_late_key_outgrows_the_first_tile(case):
    """The streaming kernel's first pass keeps the FIRST tile's maximum as the reference and never looks at the scores again; here the
    keys of the first tiles are tiny and a few late keys score ~2^500 above them (p = exp2 overflows, the row sum is inf): the workgroup
    must notice at the end and run the tracked pass, and the result is the oracle's (the late keys take all the weight)."""
    B, heads, d, N, seg0, seg1 = case
    tag = "attredo%s" % (case,)
    q = iw.normal(tag, "q", (B, heads * d, N), 0.0, 1.0)
    k = iw.normal(tag, "k", (B, heads * d, N), 0.0, 0.05)
    v = iw.normal(tag, "v", (B, heads * d, N))
    q[:, :d, :] = 6.0 + 0.1 * q[:, :d, :]            # head 0: every query points along +1 ...
    k[:, :d, N - 70:N - 60] = 5.0                     # ... and ten keys of the LAST tiles align with it: q.k / sqrt(d) ~ 240 (d = 64)
    ref = O.sdpa_heads(torch.from_numpy(q), torch.from_numpy(k), torch.from_numpy(v), heads)
    y = _attn_mh_prescaled(q, k, v, heads, seg0, seg1)
    assert bool(torch.isfinite(y).all())
    assert rel_l2(y.numpy(), ref.numpy()) < 1e-2, rel_l2(y.numpy(), ref.numpy())
    # head 0's output is the mean of the ten late keys' values (to bf16 accuracy), not of the early ones
    want = torch.from_numpy(v[:, :d, N - 70:N - 60]).mean(dim=2, keepdim=True)
    assert float((y[:, :d, :] - want).abs().max()) < 0.05


@pytest.mark.parametrize("case", [(2, 3, 16, 2048, 1024, 512), (1, 2, 16, 1024, 1024, 0)])
def test_resident_d16_attention_tracked_rounds_when_the_scores_are_large(case):
    """`attention_res16_kernel` takes p = exp2(s) with no reference maximum while |q| max|k| <= 100 (base-2 units); with q and k five times
    larger the Cauchy-Schwarz bound is in the thousands and every round runs the tracked loop (p = exp2(s - m), m moved lazily): softmax
    rows close to one-hot, compared with the oracle."""
    B, heads, d, N, seg0, seg1 = case
    tag = "attres16big%s" % (case,)
    q = iw.normal(tag, "q", (B, heads * d, N), 0.0, 6.0)
    k = iw.normal(tag, "k", (B, heads * d, N), 0.0, 6.0)
    v = iw.normal(tag, "v", (B, heads * d, N))
    ref = O.sdpa_heads(torch.from_numpy(q), torch.from_numpy(k), torch.from_numpy(v), heads)
    y = _attn_mh_prescaled(q, k, v, heads, seg0, seg1)
    assert bool(torch.isfinite(y).all())
    assert rel_l2(y.numpy(), ref.numpy()) < 2.5e-2, rel_l2(y.numpy(), ref.numpy())   # (bf16 q, k at |s| ~ 100: the logits themselves carry ~0.4 of rounding)


SHAPENET_UNET = dict(image_size=16, in_channels=192, model_channels=256, out_channels=192, attention_resolutions=[8, 4, 2],
                     num_res_blocks=2, channel_mult=[1, 2, 4, 6], num_head_channels=32)


def _shapenet_unet_bf16(path=None):
    from ddmi_amd.unet import UNetModel
    from helpers import named_state
    unet = UNetModel(**SHAPENET_UNET, numerics="bf16", max_batch=3)
    ust = named_state("shapenet.unet", unet.manifest())
    unet.load_state_dict(ust)
    x = iw.normal("shapenet", "x3", (3, 192, 16, 16))
    t = torch.tensor([777, 5, 400], dtype=torch.int64)
    y = unet(torch.from_numpy(x).to(DEV), t.to(DEV)).float().cpu()
    if path:
        np.save(path, y.numpy())
    return unet, ust, x, t, y


def test_dense_2x2_convolutions_of_the_shapenet_unet_match_the_oracle_and_the_3x3_form():
    """bf16 networks run a 3x3 convolution on a 2 x 2 image (the lowest level of configs/ldm/shapenet.yaml's UNet: 1536 channels) as the
    1x1 convolution [B][4 Cin] -> [B][4 Cout] it is (Net::pack_px4_weights, embedding rows repeated per pixel): the full-size UNet
    against the CPU oracle, the dense launches visible in the profiler record, and against the 3x3 kernels (DDMI_CONV_PX4=0, a fresh
    process) on the same inputs."""
    import os, subprocess, sys, tempfile
    os.environ["DDMI_PROF_SHAPES"] = "1"
    try:
        _lib.profile_enable(1)
        unet, ust, x, t, y = _shapenet_unet_bf16()
        prof = _lib.profile_collect()
    finally:
        _lib.profile_enable(0)
        os.environ.pop("DDMI_PROF_SHAPES", None)
    dense = [k for k in prof if "[M3,1x1,Cin6144,Cout6144,k1" in k or "[M3,1x1,Cin12288,Cout6144,k1" in k]
    assert dense and sum(prof[k]["launches"] for k in dense) >= 12, sorted(prof)[:20]        # 2 + 2 + 3 ResBlocks x 2 convolutions
    assert not [k for k in prof if ",2x2,Cin1536,Cout1536,k3" in k]                          # none of them runs the 3x3 kernel any more
    ref = O.unet_forward(O.to_state(ust), SHAPENET_UNET, torch.from_numpy(x), t)
    assert_bf16_close(y, ref, "shapenet UNet full size bf16 (dense 2x2 level)")
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import test_gpu_round4 as T\nT._shapenet_unet_bf16(sys.argv[1])\n"
            % (os.path.dirname(os.path.abspath(__file__)), os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "y3x3.npy")
        r = subprocess.run([sys.executable, "-c", code, path], env=dict(os.environ, DDMI_CONV_PX4="0"), capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        y3 = np.load(path)
    # same bf16 weights and inputs, another summation order in the 14 convolutions of the level: two bf16 evaluations of one network sit as far
    # from each other as each sits from the oracle (measured 0.0080; the oracle distance of either is ~0.011, limit 2.0e-2)
    assert rel_l2(y.numpy(), y3) < 1.5e-2, rel_l2(y.numpy(), y3)
