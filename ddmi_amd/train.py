"""LDM training step - host side: the inner step of `LDMTrainer.train` (tools/ldm/image.py:136-178):

    z -> DDPM.forward(z): t ~ randint, p_losses (diffusion/ddpm.py:409-454) -> backward -> [all-reduce] -> AdamW -> EMA

Every FLOP runs in libddmi_hip (include/ddmi_hip.h "Training": forward / backward ops for convolution, GroupNorm32+SiLU,
QKVAttentionLegacy, the time-embedding linears, q_sample, the mixed-prediction MSE loss, AdamW, EMA).  torch is the tape:
`torch.autograd.Function` records which op produced which tensor and replays the ops' backward entry points in reverse;
`torch.nn.Module` / `Parameter` hold the fp32 master weights under the reference's state-dict names (so `state_dict()` is
what `LDMTrainer.save` writes, tools/ldm/image.py:104-119), as views into ONE flat slab - the gradient slab is what RCCL
all-reduces (accelerate's DDP in the reference, tools/ldm/image.py:96) and what the fused AdamW kernel reads.
Activations are NHWC tensors in the storage dtype of the numerics mode (fp32 parity / bf16 perf).
"""
from __future__ import annotations

import ctypes as C
import os
import math
from typing import Dict, Optional, Sequence

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib
from .ddpm import make_beta_schedule


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


class TrainContext:
    """One `ddmi_train` handle (scratch arena) per device and numerics mode."""

    def __init__(self, numerics="bf16", device="cuda:0"):
        self.numerics = _lib.numerics_code(numerics)
        self.dtype = torch.float32 if _lib.is_fp32_storage(self.numerics) else torch.bfloat16
        self.device = torch.device(device)
        _lib.init_device(self.device.index or 0)
        h = C.c_void_p()
        _lib.check(_lib.lib().ddmi_train_create(self.numerics, C.byref(h)))
        self.h = h

    def __del__(self):
        try:
            if getattr(self, "h", None):
                _lib.lib().ddmi_train_destroy(self.h)
                self.h = None
        except Exception:
            pass


def _stream():
    return _lib.stream_ptr()


# ---------------------------------------------------------------------------------------------------------------------
# ops: forward and backward are entry points of the C ABI; autograd only sequences them
# ---------------------------------------------------------------------------------------------------------------------
class ToNHWC(torch.autograd.Function):
    @staticmethod
    def forward(ctx, tc, x):   # x [B,C,H,W] fp32 -> [B,H,W,C] storage dtype
        B, Cc, H, W = x.shape
        y = torch.empty((B, H, W, Cc), dtype=tc.dtype, device=x.device)
        _lib.check(_lib.lib().ddmi_train_to_nhwc(tc.h, x.contiguous().data_ptr(), y.data_ptr(), B, Cc, H * W, _stream()))
        ctx.tc = tc
        return y

    @staticmethod
    def backward(ctx, dy):
        B, H, W, Cc = dy.shape
        dx = torch.empty((B, Cc, H, W), dtype=torch.float32, device=dy.device)
        _lib.check(_lib.lib().ddmi_train_to_nchw(ctx.tc.h, dy.contiguous().data_ptr(), dx.data_ptr(), B, Cc, H * W, _stream()))
        return None, dx


class ToNCHW(torch.autograd.Function):
    @staticmethod
    def forward(ctx, tc, x):   # [B,H,W,C] storage dtype -> [B,C,H,W] fp32
        B, H, W, Cc = x.shape
        y = torch.empty((B, Cc, H, W), dtype=torch.float32, device=x.device)
        _lib.check(_lib.lib().ddmi_train_to_nchw(tc.h, x.contiguous().data_ptr(), y.data_ptr(), B, Cc, H * W, _stream()))
        ctx.tc = tc
        return y

    @staticmethod
    def backward(ctx, dy):
        B, Cc, H, W = dy.shape
        dx = torch.empty((B, H, W, Cc), dtype=ctx.tc.dtype, device=dy.device)
        _lib.check(_lib.lib().ddmi_train_to_nhwc(ctx.tc.h, dy.contiguous().data_ptr(), dx.data_ptr(), B, Cc, H * W, _stream()))
        return None, dx


FUSE_GRAD_ACCUMULATION = True   # False: every gradient goes back through autograd's AccumulateGrad (tests compare both)
FUSE_RESIDUAL_GRAD = os.environ.get("DDMI_TRAIN_FUSE_RES", "1") != "0"   # the block input's two gradients summed in the GroupNorm backward


def _grad_slot(p):
    """The tensor a parameter gradient may be accumulated into in place: the parameter's existing fp32 contiguous `.grad`."""
    if not FUSE_GRAD_ACCUMULATION or p is None or not isinstance(p, torch.Tensor) or not p.is_leaf:
        return None
    g = p.grad
    if g is None or g.dtype != torch.float32 or not g.is_contiguous() or g.shape != p.shape:
        return None
    return g


class ConvFn(torch.autograd.Function):
    """y = conv(nearest_x2?(x), w) + bias + emb[:, :, None, None] + res (ddmi_train_conv_fwd / _bwd)."""

    @staticmethod
    def forward(ctx, tc, x, w, bias, emb, res, stride, ups):
        x = x.contiguous()
        B, H, W, Cin = x.shape
        Cout, k = w.shape[0], (w.shape[2] if w.dim() == 4 else 1)
        hin, win, pad = (2 * H if ups else H), (2 * W if ups else W), k // 2
        Ho, Wo = (hin + 2 * pad - k) // stride + 1, (win + 2 * pad - k) // stride + 1
        y = torch.empty((B, Ho, Wo, Cout), dtype=tc.dtype, device=x.device)
        # emb [B, Cout] fp32, rows may be pitched (a column block of the batched embedding projection, EmbSliceFn)
        emb_c = (emb if emb.stride(1) == 1 else emb.contiguous()) if emb is not None else None
        res_c = res.contiguous() if res is not None else None
        _lib.check(_lib.lib().ddmi_train_conv_fwd(tc.h, x.data_ptr(), w.data_ptr(), _p(bias), _p(emb_c), emb_c.stride(0) if emb_c is not None else 0,
                                                  _p(res_c), y.data_ptr(), B, H, W, Cin, Cout, k, stride, int(ups), _stream()))
        ctx.save_for_backward(x, w)
        ctx.tc, ctx.meta = tc, (B, H, W, Cin, Cout, k, stride, int(ups), bias is not None, emb is not None, res is not None)
        ctx.params = (w, bias)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        B, H, W, Cin, Cout, k, stride, ups, has_b, has_e, has_r = ctx.meta
        dy = dy.contiguous()
        dx = torch.empty_like(x) if ctx.needs_input_grad[1] else None
        # parameters that already own a gradient buffer (the flat slab of FlatParams, or a .grad from an earlier backward) get their
        # gradient ADDED there by the kernel's last stage - no temporary, no separate add kernel - and autograd is handed None
        gw, gb = _grad_slot(ctx.params[0]), (_grad_slot(ctx.params[1]) if has_b else None)
        dw = None if gw is not None else torch.empty_like(w)
        db = None if (gb is not None or not has_b) else torch.empty((Cout,), dtype=torch.float32, device=x.device)
        de = torch.empty((B, Cout), dtype=torch.float32, device=x.device) if has_e else None
        mask = (1 if gw is not None else 0) | (2 if gb is not None else 0)
        if mask:
            _lib.check(_lib.lib().ddmi_train_accumulate_next(ctx.tc.h, mask))
        _lib.check(_lib.lib().ddmi_train_conv_bwd(ctx.tc.h, x.data_ptr(), w.data_ptr(), dy.data_ptr(), _p(dx), _p(gw if gw is not None else dw),
                                                  _p(gb if gb is not None else db), _p(de), B, H, W, Cin, Cout, k, stride, ups, _stream()))
        return None, dx, dw, db, de, (dy if has_r else None), None, None


class GroupNormFn(torch.autograd.Function):
    """GroupNorm32 (+ SiLU).  `passthrough=True` also returns x itself as a second output: a block hands THAT to its residual / skip
    path, so both gradients of the block input arrive in this backward, and the kernel's last pass writes their sum
    (ddmi_train_gn_bwd_res) - autograd's own add of the two (38 read-read-write passes per AFHQ step) never runs."""

    @staticmethod
    def forward(ctx, tc, x, gamma, beta, eps, silu, passthrough=False):
        x = x.contiguous()
        B, H, W, Cc = x.shape
        y = torch.empty_like(x)
        mr = torch.empty((B, 32, 2), dtype=torch.float32, device=x.device)
        _lib.check(_lib.lib().ddmi_train_gn_fwd(tc.h, x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), mr.data_ptr(), B, H * W,
                                                Cc, float(eps), int(silu), _stream()))
        ctx.save_for_backward(x, gamma, beta, mr)
        ctx.tc, ctx.silu = tc, int(silu)
        ctx.params = (gamma, beta)
        if passthrough:
            ctx.set_materialize_grads(False)      # an unused pass-through arrives as None, not as a tensor of zeros
            return y, x.view_as(x)
        return y

    @staticmethod
    def backward(ctx, dy, dres=None):
        x, gamma, beta, mr = ctx.saved_tensors
        B, H, W, Cc = x.shape
        if dy is None:                            # only the pass-through was used
            return None, dres, None, None, None, None, None
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        gg, gb = _grad_slot(ctx.params[0]), _grad_slot(ctx.params[1])
        fused = gg is not None and gb is not None          # see ConvFn.backward
        dg, db = (None, None) if fused else (torch.empty_like(gamma), torch.empty_like(beta))
        if fused:
            _lib.check(_lib.lib().ddmi_train_accumulate_next(ctx.tc.h, 4))
        if dres is not None:
            _lib.check(_lib.lib().ddmi_train_gn_bwd_res(ctx.tc.h, x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), mr.data_ptr(), dy.data_ptr(),
                                                        dres.contiguous().data_ptr(), dx.data_ptr(), (gg if fused else dg).data_ptr(),
                                                        (gb if fused else db).data_ptr(), B, H * W, Cc, ctx.silu, _stream()))
        else:
            _lib.check(_lib.lib().ddmi_train_gn_bwd(ctx.tc.h, x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), mr.data_ptr(), dy.data_ptr(),
                                                    dx.data_ptr(), (gg if fused else dg).data_ptr(), (gb if fused else db).data_ptr(), B, H * W, Cc,
                                                    ctx.silu, _stream()))
        return None, dx, dg, db, None, None, None


class CatFn(torch.autograd.Function):
    """th.cat([a, b], dim=1) of the decoder (openaimodel.py:739) on NHWC tensors: one 16-byte-vector copy kernel each way
    (torch's channel-last concatenation and the two strided slice copies of its backward ran at half the bandwidth)."""

    @staticmethod
    def forward(ctx, tc, a, b):
        a, b = a.contiguous(), b.contiguous()
        Ca, Cb = a.shape[-1], b.shape[-1]
        out = torch.empty(a.shape[:-1] + (Ca + Cb,), dtype=a.dtype, device=a.device)
        rows = a.numel() // Ca
        _lib.check(_lib.lib().ddmi_train_cat2(tc.h, a.data_ptr(), b.data_ptr(), out.data_ptr(), rows, Ca, Cb, _stream()))
        ctx.tc, ctx.meta = tc, (tuple(a.shape), tuple(b.shape), rows, Ca, Cb)
        return out

    @staticmethod
    def backward(ctx, dout):
        sa, sb, rows, Ca, Cb = ctx.meta
        dout = dout.contiguous()
        da = torch.empty(sa, dtype=dout.dtype, device=dout.device)
        db = torch.empty(sb, dtype=dout.dtype, device=dout.device)
        _lib.check(_lib.lib().ddmi_train_split2(ctx.tc.h, dout.data_ptr(), da.data_ptr(), db.data_ptr(), rows, Ca, Cb, _stream()))
        return None, da, db


PAD_NARROW_HEADS = os.environ.get("DDMI_TRAIN_ATTN_PAD", "1") != "0"   # A/B: heads of fewer than 32 channels on the plain backward kernels


class AttentionLegacyFn(torch.autograd.Function):
    """QKVAttentionLegacy on qkv [B,H,W,heads*3*d] -> [B,H,W,heads*d].

    Heads of d < 32 channels (bf16, T % 64 == 0) are ZERO-PADDED to 32 and run on the d = 32 MFMA kernels with the softmax scale of the d
    real channels (ddmi_train_attn_*_lse_scaled): the scores q . k, the softmax and the first d channels of every output and gradient are
    unchanged by the zero channels.  The tri-plane UNet's 16-head inter-plane attention at 256 channels - d = 16 over 2048 tokens - is
    two thirds of the video training step on the plain (one thread per row) backward kernels; padded it costs twice its own FLOPs at
    the MFMA rate."""

    @staticmethod
    def forward(ctx, tc, qkv, heads):
        qkv = qkv.contiguous()
        B, H, W, C3 = qkv.shape
        d = C3 // (3 * heads)
        L = _lib.lib()
        T_ = H * W
        ctx.pad_d = 0
        if PAD_NARROW_HEADS and d < 32 and L.ddmi_train_attn_lse_ok(tc.h, T_, 32) and os.environ.get("DDMI_ATTN_BWD_LEGACY") != "1":
            qp = torch.zeros((B, H, W, heads, 3, 32), dtype=qkv.dtype, device=qkv.device)
            qp[..., :d] = qkv.view(B, H, W, heads, 3, d)
            qp = qp.view(B, H, W, heads * 96)
            outp = torch.empty((B, H, W, heads * 32), dtype=tc.dtype, device=qkv.device)
            lse = torch.empty((B, heads, T_), dtype=torch.float32, device=qkv.device)
            _lib.check(L.ddmi_train_attn_fwd_lse_scaled(tc.h, qp.data_ptr(), outp.data_ptr(), lse.data_ptr(), B, T_, heads, float(d) ** -0.5, _stream()))
            ctx.save_for_backward(qp, outp, lse)
            ctx.tc, ctx.heads, ctx.pad_d = tc, heads, d
            return outp.view(B, H, W, heads, 32)[..., :d].reshape(B, H, W, heads * d)
        out = torch.empty((B, H, W, heads * d), dtype=tc.dtype, device=qkv.device)
        # the forward's row log-sum-exp rides along when the one-pass backward exists for this shape (bf16, d = 32, T % 64 == 0)
        if L.ddmi_train_attn_lse_ok(tc.h, H * W, d) and os.environ.get("DDMI_ATTN_BWD_LEGACY") != "1":
            lse = torch.empty((B, heads, H * W), dtype=torch.float32, device=qkv.device)
            _lib.check(L.ddmi_train_attn_fwd_lse(tc.h, qkv.data_ptr(), out.data_ptr(), lse.data_ptr(), B, H * W, heads, d, _stream()))
            ctx.save_for_backward(qkv, out, lse)
        else:
            _lib.check(L.ddmi_train_attn_fwd(tc.h, qkv.data_ptr(), out.data_ptr(), B, H * W, heads, d, _stream()))
            ctx.save_for_backward(qkv, out)
        ctx.tc, ctx.heads = tc, heads
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, out = ctx.saved_tensors[:2]
        B, H, W, C3 = qkv.shape
        if ctx.pad_d:
            d, heads = ctx.pad_d, ctx.heads
            dop = torch.zeros((B, H, W, heads, 32), dtype=qkv.dtype, device=qkv.device)
            dop[..., :d] = dout.reshape(B, H, W, heads, d)
            dqp = torch.empty_like(qkv)
            _lib.check(_lib.lib().ddmi_train_attn_bwd_lse_scaled(ctx.tc.h, qkv.data_ptr(), out.data_ptr(), dop.data_ptr(), ctx.saved_tensors[2].data_ptr(),
                                                                 dqp.data_ptr(), B, H * W, heads, float(d) ** -0.5, _stream()))
            return None, dqp.view(B, H, W, heads, 3, 32)[..., :d].reshape(B, H, W, heads * 3 * d), None
        d = C3 // (3 * ctx.heads)
        dqkv = torch.empty_like(qkv)
        if len(ctx.saved_tensors) == 3:
            _lib.check(_lib.lib().ddmi_train_attn_bwd_lse(ctx.tc.h, qkv.data_ptr(), out.data_ptr(), dout.contiguous().data_ptr(),
                                                          ctx.saved_tensors[2].data_ptr(), dqkv.data_ptr(), B, H * W, ctx.heads, d, _stream()))
        else:
            _lib.check(_lib.lib().ddmi_train_attn_bwd(ctx.tc.h, qkv.data_ptr(), out.data_ptr(), dout.contiguous().data_ptr(), dqkv.data_ptr(), B,
                                                      H * W, ctx.heads, d, _stream()))
        return None, dqkv, None


def _matmul(tc, A, sam, sar, Bm, sbr, sbn, M, N, R, bias=None):
    """C[m][n] = bias[n] + sum_r A[m sam + r sar] Bm[r sbr + n sbn] on fp32 device tensors (ddmi_train_matmul)."""
    out = torch.empty((M, N), dtype=torch.float32, device=A.device)
    _lib.check(_lib.lib().ddmi_train_matmul(tc.h, A.data_ptr(), sam, sar, Bm.data_ptr(), sbr, sbn, out.data_ptr(), _p(bias), M, N, R, _stream()))
    return out


def _silu_grad(x):
    sg = torch.sigmoid(x)
    return sg * (1.0 + x * (1.0 - sg))


class LinearFn(torch.autograd.Function):
    """y = act(x) W^T + b on fp32 [B,K]; act = SiLU when silu_in (time_embed; the ResBlock emb_layers run batched, EmbProjFn)."""

    @staticmethod
    def forward(ctx, tc, x, W, b, silu_in):
        x = x.contiguous()
        B, K = x.shape
        N = W.shape[0]
        xa = F.silu(x) if silu_in else x
        y = _matmul(tc, xa, K, 1, W, 1, K, B, N, K, b)                      # A = act(x) [B,K], B(r = k, n) = W[n][k]
        ctx.save_for_backward(x, W)
        ctx.tc, ctx.silu_in, ctx.has_b = tc, bool(silu_in), b is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, W = ctx.saved_tensors
        B, K = x.shape
        N = W.shape[0]
        dy = dy.contiguous()
        xa = F.silu(x) if ctx.silu_in else x
        dx = None
        if ctx.needs_input_grad[1]:
            dx = _matmul(ctx.tc, dy, N, 1, W, K, 1, B, K, N)                  # A = dy [B,N], B(r = n, k) = W[n][k]
            if ctx.silu_in:
                dx = dx * _silu_grad(x)
        dW = _matmul(ctx.tc, dy, 1, N, xa, K, 1, N, K, B)                      # A(m = n, r = b) = dy[b][n], B(r = b, k) = act(x)[b][k]
        db = dy.sum(0) if ctx.has_b else None
        return None, dx, dW, db, None


class EmbProjFn(torch.autograd.Function):
    """Every ResBlock's `emb_layers` (SiLU -> Linear(E, cout), openaimodel.py:207-213) of one UNet evaluation as ONE projection:
    the 22 weight matrices are stacked to [sum cout, E], SiLU(emb) is taken once, and the blocks read their columns of the
    [B, sum cout] result in place (the conv epilogue takes a row pitch).  Three GEMMs per step instead of 66 small launches whose
    latency, not arithmetic, filled 10 ms of a B = 64 step."""

    @staticmethod
    def forward(ctx, tc, emb, n_blocks, *wb):
        ws, bs = wb[:n_blocks], wb[n_blocks:]
        emb = emb.contiguous()
        B, E = emb.shape
        Wall, ball = torch.cat(ws, 0), torch.cat(bs, 0)
        total = Wall.shape[0]
        semb = F.silu(emb)
        y = _matmul(tc, semb, E, 1, Wall, 1, E, B, total, E, ball)
        ctx.save_for_backward(emb, Wall)
        ctx.tc, ctx.sizes = tc, [int(w.shape[0]) for w in ws]
        return y

    @staticmethod
    def backward(ctx, dy):
        if EmbProjFn.pre_backward is not None:   # the overlapped gradient all-reduce: see OverlappedGradReducer.before_late_writes
            EmbProjFn.pre_backward()
        emb, Wall = ctx.saved_tensors
        B, E = emb.shape
        total = Wall.shape[0]
        dy = dy.contiguous()
        semb = F.silu(emb)
        demb = _matmul(ctx.tc, dy, total, 1, Wall, E, 1, B, E, total) * _silu_grad(emb) if ctx.needs_input_grad[1] else None
        dW = _matmul(ctx.tc, dy, 1, total, semb, E, 1, total, E, B)
        db = dy.sum(0)
        return (None, demb, None) + tuple(dW.split(ctx.sizes, 0)) + tuple(db.split(ctx.sizes, 0))


EmbProjFn.pre_backward = None


class EmbSliceFn(torch.autograd.Function):
    """Column block [a, b) of the batched projection as a (row-pitched) view; the gradient of all blocks gathers in ONE
    [B, total] buffer shared through `bucket` instead of 22 zero-padded full-size tensors."""

    @staticmethod
    def forward(ctx, y_all, a, b, bucket):
        ctx.a, ctx.b, ctx.bucket, ctx.shape = a, b, bucket, y_all.shape
        return y_all[:, a:b]

    @staticmethod
    def backward(ctx, g):
        bk = ctx.bucket
        if bk.get("buf") is None:
            bk["buf"] = torch.zeros(ctx.shape, dtype=g.dtype, device=g.device)
            bk["pending"] = bk["n"]
        bk["buf"][:, ctx.a:ctx.b] = g
        bk["pending"] -= 1
        if bk["pending"] == 0:       # the last block to report hands the full gradient on; the others contribute nothing more
            out, bk["buf"] = bk["buf"], None
            return out, None, None, None
        return None, None, None, None


class EpsLossFn(torch.autograd.Function):
    """p_losses (ddpm.py:409-447): mixed prediction + MSE against the noise; gradients come out of the same kernel."""

    @staticmethod
    def forward(ctx, tc, model_out, x_t, noise, sqrt_1mac, logit):
        B, Cc = model_out.shape[:2]                      # [B, C, H, W] (image) or [B, C, L] (video latents, ddpm.py:152-159)
        HW = model_out[0, 0].numel()
        loss = torch.empty((), dtype=torch.float32, device=model_out.device)
        dmo = torch.empty_like(model_out)
        dlogit = torch.empty((Cc,), dtype=torch.float32, device=model_out.device) if logit is not None else None
        lg = logit.reshape(-1).contiguous() if logit is not None else None
        _lib.check(_lib.lib().ddmi_train_eps_loss(tc.h, model_out.contiguous().data_ptr(), x_t.data_ptr(), noise.data_ptr(), sqrt_1mac.data_ptr(),
                                                  _p(lg), loss.data_ptr(), dmo.data_ptr(), _p(dlogit), B, Cc, HW, _stream()))
        ctx.save_for_backward(dmo, dlogit if dlogit is not None else torch.empty(0))
        ctx.logit_shape = tuple(logit.shape) if logit is not None else None
        return loss

    @staticmethod
    def backward(ctx, g):
        dmo, dlogit = ctx.saved_tensors
        return None, dmo * g, None, None, None, (dlogit.reshape(ctx.logit_shape) * g if ctx.logit_shape else None)


# ---------------------------------------------------------------------------------------------------------------------
# UNetModel as a tree of parameter holders with the reference's state-dict names (openaimodel.py:463-793)
# ---------------------------------------------------------------------------------------------------------------------
class _Conv(nn.Module):
    def __init__(self, cin, cout, k, dims=2):
        super().__init__()
        self.weight = nn.Parameter(torch.empty((cout, cin, k, k) if dims == 2 else (cout, cin, k)))
        self.bias = nn.Parameter(torch.empty((cout,)))

    def w4(self):
        return self.weight if self.weight.dim() == 4 else self.weight.unsqueeze(-1)

    def run(self, tc, x, emb=None, res=None, stride=1, ups=False):
        # the parameter itself (a Conv1d's [Cout, Cin, 1] has the bytes of [Cout, Cin, 1, 1]): ConvFn can then add into its .grad
        return ConvFn.apply(tc, x, self.weight, self.bias, emb, res, stride, ups)


class _Norm(nn.Module):   # GroupNorm32(32, C); eps 1e-6 for the d2c_vae-style blocks (autoencoder_unet.py:46-47)
    def __init__(self, c, eps=1e-5):
        super().__init__()
        self.weight = nn.Parameter(torch.empty((c,)))
        self.bias = nn.Parameter(torch.empty((c,)))
        self.eps = float(eps)

    def run(self, tc, x, silu, passthrough=False):
        return GroupNormFn.apply(tc, x, self.weight, self.bias, self.eps, silu, passthrough and FUSE_RESIDUAL_GRAD)


class _Linear(nn.Module):
    def __init__(self, k, n):
        super().__init__()
        self.weight = nn.Parameter(torch.empty((n, k)))
        self.bias = nn.Parameter(torch.empty((n,)))

    def run(self, tc, x, silu_in):
        return LinearFn.apply(tc, x, self.weight, self.bias, silu_in)


def _seq(*mods):
    """nn.Sequential whose parameterless slots (SiLU, Dropout, Identity) keep the reference's numeric keys."""
    return nn.Sequential(*[m if m is not None else nn.Identity() for m in mods])


class ResBlock(nn.Module):
    def __init__(self, cin, emb_ch, cout):
        super().__init__()
        self.in_layers = _seq(_Norm(cin), None, _Conv(cin, cout, 3))
        self.emb_layers = _seq(None, _Linear(emb_ch, cout))
        self.out_layers = _seq(_Norm(cout), None, None, _Conv(cout, cout, 3))
        self.skip_connection = _Conv(cin, cout, 1) if cin != cout else nn.Identity()

    def run(self, tc, x, emb):
        """`emb` is this block's emb_layers OUTPUT [B, cout]: TrainUNetModel.forward projects all blocks at once (EmbProjFn)."""
        h = self.in_layers[0].run(tc, x, silu=True, passthrough=True)
        if isinstance(h, tuple):
            h, x = h                                                                             # x for the skip path: see GroupNormFn
        h = self.in_layers[2].run(tc, h, emb=emb)                                                # h + emb_out[..., None, None]
        h = self.out_layers[0].run(tc, h, silu=True)
        skip = x if isinstance(self.skip_connection, nn.Identity) else self.skip_connection.run(tc, x)
        return self.out_layers[3].run(tc, h, res=skip)                                          # skip_connection(x) + h


class AttentionBlock(nn.Module):
    def __init__(self, ch, head_ch):
        super().__init__()
        self.heads = ch // head_ch
        self.norm = _Norm(ch)
        self.qkv = _Conv(ch, 3 * ch, 1, dims=1)
        self.proj_out = _Conv(ch, ch, 1, dims=1)

    def run(self, tc, x, emb=None):
        hn = self.norm.run(tc, x, silu=False, passthrough=True)
        if isinstance(hn, tuple):
            hn, x = hn
        qkv = self.qkv.run(tc, hn)
        a = AttentionLegacyFn.apply(tc, qkv, self.heads)
        return self.proj_out.run(tc, a, res=x)


class Downsample(nn.Module):
    def __init__(self, ch):
        super().__init__()
        self.op = _Conv(ch, ch, 3)

    def run(self, tc, x, emb=None):
        return self.op.run(tc, x, stride=2)


class Upsample(nn.Module):
    def __init__(self, ch):
        super().__init__()
        self.conv = _Conv(ch, ch, 3)

    def run(self, tc, x, emb=None):
        return self.conv.run(tc, x, ups=True)


class _InConv(nn.Module):
    pass


class TrainUNetModel(nn.Module):
    """The training-path UNetModel: same constructor kwargs and state-dict keys as the reference (openaimodel.py:493-743)
    for the variant the DDMI configs build (2-D, legacy attention, conv_resample)."""

    def __init__(self, image_size, in_channels, model_channels, out_channels, num_res_blocks, attention_resolutions,
                 channel_mult=(1, 2, 4, 8), num_head_channels=32, numerics="bf16", device="cuda:0", **ignored):
        super().__init__()
        mc, E = model_channels, 4 * model_channels
        self.model_channels, self.in_channels, self.out_channels = mc, in_channels, out_channels
        self._numerics, self._device, self._tc = numerics, device, None   # the native handle is created on first use
        self._rb = None
        self._block_grad_cb = None   # set by LDMTrainStep for a step whose gradient all-reduce overlaps the backward
        self.time_embed = _seq(_Linear(mc, E), None, _Linear(E, E))
        self.input_blocks = nn.ModuleList([nn.Sequential(_Conv(in_channels, mc, 3))])
        chans, ch, ds = [mc], mc, 1
        attn = set(int(a) for a in attention_resolutions)
        mult = list(channel_mult)
        for level, m in enumerate(mult):
            for _ in range(num_res_blocks):
                layers = [ResBlock(ch, E, m * mc)]
                ch = m * mc
                if ds in attn:
                    layers.append(AttentionBlock(ch, num_head_channels))
                self.input_blocks.append(nn.Sequential(*layers))
                chans.append(ch)
            if level != len(mult) - 1:
                self.input_blocks.append(nn.Sequential(Downsample(ch)))
                chans.append(ch)
                ds *= 2
        self.middle_block = nn.Sequential(ResBlock(ch, E, ch), AttentionBlock(ch, num_head_channels), ResBlock(ch, E, ch))
        self.output_blocks = nn.ModuleList([])
        for level, m in list(enumerate(mult))[::-1]:
            for i in range(num_res_blocks + 1):
                layers = [ResBlock(ch + chans.pop(), E, mc * m)]
                ch = mc * m
                if ds in attn:
                    layers.append(AttentionBlock(ch, num_head_channels))
                if level and i == num_res_blocks:
                    layers.append(Upsample(ch))
                    ds //= 2
                self.output_blocks.append(nn.Sequential(*layers))
        self.out = _seq(_Norm(ch), None, _Conv(mc, out_channels, 3))
        self.to(device)

    @property
    def tc(self) -> TrainContext:
        if self._tc is None:
            self._tc = TrainContext(self._numerics, self._device)
        return self._tc

    @staticmethod
    def _run_block(tc, block, h, embs):
        for layer in block:
            if isinstance(layer, ResBlock):
                h = layer.run(tc, h, embs[id(layer)])
            else:
                h = layer.run(tc, h, None) if not isinstance(layer, _Conv) else layer.run(tc, h)
        return h

    def _resblocks(self):
        if self._rb is None:
            blocks = list(self.input_blocks) + [self.middle_block] + list(self.output_blocks)
            self._rb = [layer for blk in blocks for layer in blk if isinstance(layer, ResBlock)]
        return self._rb

    def _project_embeddings(self, tc, emb):
        """emb_layers of every ResBlock in one GEMM (EmbProjFn); returns {id(block): [B, cout] column view}."""
        rbs = self._resblocks()
        lin = [rb.emb_layers[1] for rb in rbs]
        y_all = EmbProjFn.apply(tc, emb, len(rbs), *[m.weight for m in lin], *[m.bias for m in lin])
        bucket, out, a = {"n": len(rbs), "buf": None}, {}, 0
        for rb, m in zip(rbs, lin):
            n = m.weight.shape[0]
            out[id(rb)] = EmbSliceFn.apply(y_all, a, a + n, bucket)
            a += n
        return out

    def forward(self, x, timesteps, context=None, **kw):
        """x [B,C,H,W] fp32, timesteps [B] int64 -> eps [B,C,H,W] fp32 (openaimodel.py:761-793)."""
        tc = self.tc
        B = x.shape[0]
        temb = torch.empty((B, self.model_channels), dtype=torch.float32, device=x.device)
        t64 = timesteps.to(device=x.device, dtype=torch.int64).contiguous()
        _lib.check(_lib.lib().ddmi_train_timestep_embedding(tc.h, t64.data_ptr(), temb.data_ptr(), B, self.model_channels, _stream()))
        emb = self.time_embed[2].run(tc, self.time_embed[0].run(tc, temb, silu_in=False), silu_in=True)
        embs = self._project_embeddings(tc, emb)
        h = ToNHWC.apply(tc, x)
        hs = []
        cb = self._block_grad_cb

        def mark(t, key):
            # the gradient of a block's INPUT exists once the block (and every other consumer of that tensor) has run its backward:
            # the block's parameter gradients are final then (OverlappedGradReducer; the hook leaves the gradient alone)
            if cb is not None and t.requires_grad:
                t.register_hook(lambda g, k=key: cb(k))

        for i, block in enumerate(self.input_blocks):
            mark(h, f"input_blocks.{i}")
            h = self._run_block(tc, block, h, embs)
            hs.append(h)
        mark(h, "middle_block")
        h = self._run_block(tc, self.middle_block, h, embs)
        for j, block in enumerate(self.output_blocks):
            skip = hs.pop()                              # th.cat([h, hs.pop()], dim=1) in NHWC
            vec = 16 // h.element_size()
            h = CatFn.apply(tc, h, skip) if h.shape[-1] % vec == 0 and skip.shape[-1] % vec == 0 else torch.cat([h, skip], dim=-1)
            mark(h, f"output_blocks.{j}")
            h = self._run_block(tc, block, h, embs)
        mark(h, "out")
        h = self.out[2].run(tc, self.out[0].run(tc, h, silu=True))
        return ToNCHW.apply(tc, h)


class InterAttention(nn.Module):
    """The block UNetModel_Triplane puts after every input block but the first, after the middle block and after every output block
    (openaimodel.py:912, 955-957, 984-986, 1023, 1082): make_attn('vanilla-1d-multihead') = MemoryEfficientAttnBlock1D
    (autoencoder_unet.py:404-468) over the tokens xy | xt | yt of a sample - GroupNorm(32, C, eps 1e-6) with statistics over ALL of
    them, q / k / v = Conv1d(C, C, 1), 16 heads of C / 16 channels, softmax(q k^T / sqrt(d)) v, proj_out, + x.
    Tokens arrive as an NHWC "image" [B, 1, N, C]."""

    def __init__(self, ch, heads=16):
        super().__init__()
        self.heads = heads
        self.norm = _Norm(ch, eps=1e-6)
        self.q, self.k, self.v = _Conv(ch, ch, 1, dims=1), _Conv(ch, ch, 1, dims=1), _Conv(ch, ch, 1, dims=1)
        self.proj_out = _Conv(ch, ch, 1, dims=1)

    def run(self, tc, x):
        hn = self.norm.run(tc, x, silu=False, passthrough=True)
        if isinstance(hn, tuple):
            hn, x = hn
        B, _, N, Cc = x.shape
        d = Cc // self.heads
        # three projections (the reference's three Conv1d modules keep their own state-dict entries), interleaved per head into
        # QKVAttentionLegacy's [q | k | v] channel blocks - the layout the attention kernels and their backward take
        q, k, v = (m.run(tc, hn).reshape(B, 1, N, self.heads, 1, d) for m in (self.q, self.k, self.v))
        qkv = torch.cat([q, k, v], dim=4).reshape(B, 1, N, 3 * Cc)
        a = AttentionLegacyFn.apply(tc, qkv, self.heads)
        return self.proj_out.run(tc, a, res=x)


class TrainUNetModel_Triplane(TrainUNetModel):
    """The training-path `UNetModel_Triplane` (openaimodel.py:800-1221; tools/ldm/video.py:150-167 trains it): the UNetModel's blocks
    applied to the three planes xy [s1, s2], xt [s3, s1], yt [s3, s2] of a latent [B, C, s1 s2 + s1 s3 + s2 s3] with SHARED weights
    (their gradients accumulate in the same slab entries), and a 16-head attention over the concatenated tokens after every block.
    Same constructor kwargs and state-dict keys as the reference.  Composed of the image model's ops (ConvFn, GroupNormFn,
    AttentionLegacyFn - forward and backward entry points of the C ABI) plus torch views for the plane / token bookkeeping."""

    def __init__(self, in_channels, model_channels, out_channels, num_res_blocks, attention_resolutions, channel_mult=(1, 2, 4, 8),
                 num_head_channels=32, size1=64, size2=64, size3=16, attn_type="vanilla-1d-multihead", numerics="bf16", device="cuda:0",
                 **ignored):
        if attn_type != "vanilla-1d-multihead":
            raise NotImplementedError("UNetModel_Triplane: only attn_type='vanilla-1d-multihead' (the default every config uses)")
        super().__init__(image_size=size1, in_channels=in_channels, model_channels=model_channels, out_channels=out_channels,
                         num_res_blocks=num_res_blocks, attention_resolutions=attention_resolutions, channel_mult=channel_mult,
                         num_head_channels=num_head_channels, numerics=numerics, device=device)
        self.size1, self.size2, self.size3 = int(size1), int(size2), int(size3)
        self.seq_len = self.size1 * self.size2 + self.size1 * self.size3 + self.size2 * self.size3
        mc, mult = model_channels, list(channel_mult)
        attns, ch = [nn.Identity()], mc                      # input_attns[0] is nn.Identity (openaimodel.py:912)
        for level, m in enumerate(mult):
            for _ in range(num_res_blocks):
                ch = m * mc
                attns.append(InterAttention(ch))
            if level != len(mult) - 1:
                attns.append(InterAttention(ch))               # after the Downsample block
        self.input_attns = nn.ModuleList(attns)
        self.mid_attn = InterAttention(ch)
        self.output_attns = nn.ModuleList([InterAttention(mc * m) for level, m in list(enumerate(mult))[::-1]
                                           for _ in range(num_res_blocks + 1)])
        assert len(self.input_attns) == len(self.input_blocks) and len(self.output_attns) == len(self.output_blocks)
        self.to(device)

    @staticmethod
    def _inter(tc, attn, hs):
        if isinstance(attn, nn.Identity):
            return hs
        B, Cc = hs[0].shape[0], hs[0].shape[-1]
        shapes = [tuple(h.shape) for h in hs]
        tok = torch.cat([h.reshape(B, 1, -1, Cc) for h in hs], dim=2)          # xy | xt | yt (openaimodel.py:1148-1152)
        out = attn.run(tc, tok)
        parts = torch.split(out, [s[1] * s[2] for s in shapes], dim=2)
        return [p_.reshape(s) for p_, s in zip(parts, shapes)]

    def forward(self, x, timesteps, context=None, **kw):
        """x [B, C, L] fp32, timesteps [B] int64 -> eps [B, C, L] fp32 (openaimodel.py:1113-1221)."""
        tc = self.tc
        B, Cc, L = x.shape
        assert L == self.seq_len, (L, self.seq_len)
        s1, s2, s3 = self.size1, self.size2, self.size3
        temb = torch.empty((B, self.model_channels), dtype=torch.float32, device=x.device)
        t64 = timesteps.to(device=x.device, dtype=torch.int64).contiguous()
        _lib.check(_lib.lib().ddmi_train_timestep_embedding(tc.h, t64.data_ptr(), temb.data_ptr(), B, self.model_channels, _stream()))
        emb = self.time_embed[2].run(tc, self.time_embed[0].run(tc, temb, silu_in=False), silu_in=True)
        embs = self._project_embeddings(tc, emb)
        n_xy, n_xt = s1 * s2, s1 * s3
        planes = (x[:, :, :n_xy].reshape(B, Cc, s1, s2), x[:, :, n_xy:n_xy + n_xt].reshape(B, Cc, s3, s1),
                  x[:, :, n_xy + n_xt:].reshape(B, Cc, s3, s2))
        hs = [ToNHWC.apply(tc, p_.contiguous()) for p_ in planes]
        skips = []
        for i, block in enumerate(self.input_blocks):
            hs = [self._run_block(tc, block, h, embs) for h in hs]
            hs = self._inter(tc, self.input_attns[i], hs)
            skips.append(hs)
        hs = [self._run_block(tc, self.middle_block, h, embs) for h in hs]
        hs = self._inter(tc, self.mid_attn, hs)
        for j, block in enumerate(self.output_blocks):
            sk = skips.pop()
            cat = []
            for h, s_ in zip(hs, sk):
                vec = 16 // h.element_size()
                cat.append(CatFn.apply(tc, h, s_) if h.shape[-1] % vec == 0 and s_.shape[-1] % vec == 0 else torch.cat([h, s_], dim=-1))
            hs = [self._run_block(tc, block, h, embs) for h in cat]
            hs = self._inter(tc, self.output_attns[j], hs)
        outs = [ToNCHW.apply(tc, self.out[2].run(tc, self.out[0].run(tc, h, silu=True))) for h in hs]
        return torch.cat([o.reshape(B, o.shape[1], -1) for o in outs], dim=-1)


# ---------------------------------------------------------------------------------------------------------------------
# the training step
# ---------------------------------------------------------------------------------------------------------------------
class FlatParams:
    """All parameters of a module as views into one fp32 slab (+ a gradient slab of the same layout, + AdamW moments)."""

    def __init__(self, params: Sequence[nn.Parameter], device):
        self.params = list(params)
        n = sum(p.numel() for p in self.params)
        self.data = torch.zeros(n, dtype=torch.float32, device=device)
        self.grad = torch.zeros(n, dtype=torch.float32, device=device)
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=device)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=device)
        off = 0
        with torch.no_grad():
            for p in self.params:
                k = p.numel()
                self.data[off:off + k].copy_(p.data.reshape(-1))
                p.data = self.data[off:off + k].view(p.shape)
                p.grad = self.grad[off:off + k].view(p.shape)
                off += k
        self.n = n

    def zero_grad(self):
        self.grad.zero_()


def allreduce_gradients(flat_grad: torch.Tensor, world_size: int, bucket_bytes: int = 1 << 30) -> torch.Tensor:
    """Sum of the flat gradient slab over the data-parallel ranks (RCCL all-reduce over xGMI with the `nccl` backend; `gloo` in
    the CPU tests) - what accelerate's DDP wrapper does for the reference (tools/ldm/image.py:96, 165).  The slab is reduced
    in place in buckets of `bucket_bytes` (1 GiB: two collectives for the AFHQ UNet's 2.08 GB - a ring all-reduce over 7 xGMI
    links is bandwidth-bound per link, so few large buckets; issued back to back on the collective stream they also overlap
    their own reduce-scatter / all-gather phases).  The 1 / world_size of DDP's mean is folded into the AdamW kernel."""
    import torch.distributed as dist
    if world_size <= 1 or not dist.is_initialized():
        return flat_grad
    n_per = max(1, bucket_bytes // flat_grad.element_size())
    works = [dist.all_reduce(flat_grad[a:a + n_per], async_op=True) for a in range(0, flat_grad.numel(), n_per)]
    for w in works:
        w.wait()
    return flat_grad


class OverlappedGradReducer:
    """The gradient all-reduce of `allreduce_gradients`, issued WHILE the backward still runs (what DDP's bucketed reducer does for the
    reference, tools/ldm/image.py:96, 165).  The flat slab makes it simple: the backward completes the slab from its END (the loss's
    `mixing_logit`, `out`, the output blocks in reverse, the middle block, the input blocks in reverse, `time_embed`), so the completed part
    is always one contiguous tail; every time another `bucket_bytes` of it are complete they go out as ONE asynchronous all-reduce (few,
    large collectives: a ring over 7 xGMI links is bandwidth-bound per link) while the remaining blocks' backward kernels keep the GPU busy.

    `ranges`: (lo, hi) element ranges in COMPLETION order - contiguous, descending.  `late`: (lo, hi) views inside those ranges that are
    only written at the very end of the backward (the 22 `emb_layers` weights / biases: ONE stacked projection computes them, `EmbProjFn`,
    and its backward runs after the last ResBlock's); they must be ZERO while their range goes out early (one micro-step per optimizer step)
    and are reduced again, gathered into one buffer, in `finish()`.  Sums commute, so the result is the blocking all-reduce's - bit for bit
    on two ranks (a + b), up to the ring's association order on more."""

    def __init__(self, flat_grad: torch.Tensor, ranges, late=(), world_size: int = 1, bucket_bytes: int = 256 << 20):
        self.g, self.world = flat_grad, int(world_size)
        self.ranges = [(int(a), int(b)) for a, b in ranges]
        for (a, b), (c, d) in zip(self.ranges, self.ranges[1:]):
            assert d == a and c < d, "ranges must be contiguous and in descending (completion) order"
        self.late = [(int(a), int(b)) for a, b in late]
        self.bucket = max(1, int(bucket_bytes) // flat_grad.element_size())
        self.top = self.ranges[0][1] if self.ranges else flat_grad.numel()
        self.done_lo = self.top      # [done_lo, top) is complete
        self.sent_lo = self.top      # [sent_lo, top) has been issued
        self.next = 0
        self.works = []
        self.collectives = 0
        self.late_written = False    # set by before_late_writes(): no bucket goes out after it (see range_done)

    def _active(self):
        import torch.distributed as dist
        return self.world > 1 and dist.is_initialized()

    def _issue(self, lo, hi):
        import torch.distributed as dist
        if hi > lo:
            self.works.append(dist.all_reduce(self.g[lo:hi], async_op=True))
            self.collectives += 1

    def range_done(self, idx: int):
        """Range `idx` (and, the backward being ordered, every range before it) holds its final gradient - except the late views."""
        if idx < self.next:
            return
        self.next = idx + 1
        self.done_lo = self.ranges[idx][0]
        # after before_late_writes() the late views hold FINAL values: a bucket issued now would reduce them, and finish() - which
        # redoes every late view of an early bucket - would sum them over the ranks a second time.  From then on range_done() only
        # records progress; finish() sends the rest in one piece (whose late views it does not redo).
        if self._active() and not self.late_written and self.sent_lo - self.done_lo >= self.bucket:
            self._issue(self.done_lo, self.sent_lo)
            self.sent_lo = self.done_lo

    def before_late_writes(self):
        """Called right before the backward writes the late views (EmbProjFn.backward): the collectives already issued cover them and
        run in place - one still in flight would overwrite what the backward is about to add there.  Waiting costs nothing by then:
        only the head of the slab is left to do."""
        for w in self.works:
            w.wait()
        self.works = []
        self.late_written = True

    def abort(self):
        """The backward raised: wait for what is in flight (the collectives run in place on the slab) and issue nothing more."""
        for w in self.works:
            w.wait()
        self.works = []
        self.late_written = True

    def finish(self):
        """After the backward: whatever is left of the slab (its head - `time_embed`, the first block - and anything below a bucket), then
        the late views of the ranges that went out early, as one gathered collective."""
        if not self._active():
            return self.g
        import torch.distributed as dist
        early_lo = self.sent_lo
        self._issue(0, self.sent_lo)                      # includes the late views that lie in it: final values, nothing to redo
        self.sent_lo = 0
        redo = [(a, b) for a, b in self.late if a >= early_lo]
        for w in self.works:
            w.wait()
        self.works = []
        if redo:
            # the early collective summed zeros there; the values written since are local: reduce them now
            buf = torch.cat([self.g[a:b] for a, b in redo])
            dist.all_reduce(buf)
            self.collectives += 1
            off = 0
            for a, b in redo:
                self.g[a:b].copy_(buf[off:off + b - a])
                off += b - a
        return self.g


def ema_pytorch_plan(step: int, beta: float, update_every: int, update_after_step: int = 100, inv_gamma: float = 1.0,
                     power: float = 2.0 / 3.0, min_value: float = 0.0):
    """What the `step`-th call (0-based: the value of ema_pytorch's `step` buffer BEFORE its increment) of `EMA.update()` does:
    ("skip", None) | ("copy", None) | ("lerp", decay) with ema = decay * ema + (1 - decay) * online."""
    if step % update_every != 0:
        return "skip", None
    if step <= update_after_step:
        return "copy", None
    epoch = max((step + 1) - update_after_step - 1, 0)     # get_current_decay reads the incremented buffer
    if epoch <= 0:
        return "lerp", 0.0
    value = 1.0 - (1.0 + epoch / inv_gamma) ** (-power)
    return "lerp", float(min(max(value, min_value), beta))


def module_parameters_order(names):
    """The order `nn.Module.parameters()` yields parameters named `names` (given in any order): the root module's OWN parameters
    (no dot in the name) first, then the submodules' - `named_parameters()` walks `named_modules()`, root first.  For the reference's
    `DDPM` that puts `mixing_logit` (ddpm.py:156) at index 0 of `AdamW(diffusion_process.parameters())` (tools/ldm/image.py:62) and
    `model.*` after it, whatever the registration order of the two was."""
    names = list(names)
    return [k for k in names if "." not in k] + [k for k in names if "." in k]


class LDMTrainStep:
    """`DDPM.forward` + backward + optimizer step of the reference's LDM trainer on one GPU of a data-parallel job.

    diffusion = (TrainUNetModel, mixing_logit); schedule from the `ddpmconfig` block; AdamW(lr, weight_decay 0) as
    tools/ldm/image.py:62; gradients are summed over ranks with one all-reduce of the flat slab and scaled by
    1 / (world_size * gradient_accumulate_every) inside the fused AdamW kernel."""

    def __init__(self, unetconfig, ddpmconfig, numerics="bf16", lr=1e-4, weight_decay=0.0, betas=(0.9, 0.999), eps=1e-8,
                 gradient_accumulate_every=1, ema_decay=0.9999, ema_update_every=10, device="cuda:0", use_graph=False,
                 overlap_allreduce=True, allreduce_bucket_bytes=256 << 20):
        cfg = dict(ddpmconfig)
        self.device = torch.device(device)
        # the video domain (tools/ldm/video.py): a tri-plane UNet on [B, C, L] latents, mixing_logit [1, C, 1] (ddpm.py:152-159)
        self.video = "size3" in dict(unetconfig) or cfg.get("domain") == "video"
        self.model = (TrainUNetModel_Triplane if self.video else TrainUNetModel)(**dict(unetconfig), numerics=numerics, device=device)
        self.channels = int(cfg.get("channels", unetconfig["in_channels"]))
        self.num_timesteps = int(cfg.get("timesteps", 1000))
        betas_np = make_beta_schedule(cfg.get("beta_schedule", "linear"), self.num_timesteps, cfg.get("linear_start", 0.0015),
                                      cfg.get("linear_end", 0.0195), cfg.get("cosine_s", 8e-3))
        ac = np.cumprod(1.0 - betas_np, axis=0)
        self.sqrt_ac = torch.from_numpy(np.sqrt(ac).astype(np.float32)).to(self.device)
        self.sqrt_1mac = torch.from_numpy(np.sqrt(1.0 - ac).astype(np.float32)).to(self.device)
        self.mixed_prediction = bool(cfg.get("mixed_prediction", True))
        self.mixing_logit = nn.Parameter(torch.full((1, self.channels, 1) if self.video else (1, self.channels, 1, 1),
                                                    float(cfg.get("mixed_init", -6)), device=self.device)) if self.mixed_prediction else None
        self.lr, self.wd, self.betas, self.eps = float(lr), float(weight_decay), betas, float(eps)
        self.accum = int(gradient_accumulate_every)
        self.ema_decay, self.ema_every = float(ema_decay), int(ema_update_every)
        # ema_pytorch.EMA defaults the reference relies on (tools/ldm/image.py:64 passes beta and update_every only)
        self.ema_update_after_step, self.ema_inv_gamma, self.ema_power, self.ema_min_value = 100, 1.0, 2.0 / 3.0, 0.0
        self.ema_step = 0            # ema_pytorch's `step` buffer: number of update() calls so far
        self.ema_initted = False     # its `initted` buffer
        self._ddpmconfig = {k: v for k, v in cfg.items() if isinstance(v, (int, float, str, bool, type(None)))}
        self.flat: Optional[FlatParams] = None
        self.ema: Optional[torch.Tensor] = None
        self.opt_step = 0
        self.micro = 0
        self.epoch = 0               # the reference trainer's `step`: finished EPOCHS (tools/ldm/image.py:145, 205), set by the caller's loop
        self.current_iters = 0       # the reference's `current_iters`: micro-steps so far
        self.use_graph = bool(use_graph)     # replay the micro-step as a hipGraph (see _graphed_micro)
        # gradient all-reduce overlapped with the backward (eager steps, one micro-step per optimizer step, more than one rank)
        # (the tri-plane model's inter-plane attention parameters sit behind `out` in the slab and finish with their blocks: the bucket
        # plan below is the image model's; the video step takes the blocking all-reduce)
        self.overlap_allreduce, self.allreduce_bucket_bytes = bool(overlap_allreduce) and not self.video, int(allreduce_bucket_bytes)
        self._reducer: Optional[OverlappedGradReducer] = None
        self._reduce_plan = None
        self._graphs: Dict[tuple, object] = {}
        self._graph_generation = 0

    # ---- parameters ----------------------------------------------------------------------------------------------------
    def named_parameters(self):
        """The reference's `diffusion_process.state_dict()` naming: 'model.*' + 'mixing_logit' (tools/ldm/image.py:79)."""
        for k, p in self.model.named_parameters():
            yield "model." + k, p
        if self.mixing_logit is not None:
            yield "mixing_logit", self.mixing_logit

    def optimizer_order(self):
        """(name, parameter, offset in the flat slab) in the order `torch.optim.AdamW(diffusion_process.parameters())` indexes them
        (tools/ldm/image.py:62): `nn.Module.parameters()` yields the ROOT module's own parameters first - `mixing_logit` (ddpm.py:156)
        is index 0 when present - and the `model` submodule's after it.  The flat slab keeps the state-dict order of
        named_parameters(); only `dae_opt` is indexed this way."""
        named, off, offs = list(self.named_parameters()), 0, {}
        for k, p in named:
            offs[k] = off
            off += p.numel()
        by_name = dict(named)
        return [(k, by_name[k], offs[k]) for k in module_parameters_order(k for k, _ in named)]

    def load_state_dict(self, state, strict=True):
        own = dict(self.named_parameters())
        missing = [k for k in own if k not in state]
        if strict and missing:
            raise KeyError(f"missing keys: {missing[:5]}")
        with torch.no_grad():
            for k, v in state.items():
                if k in own:
                    t = torch.as_tensor(np.asarray(v) if not hasattr(v, "detach") else v).to(self.device, torch.float32)
                    own[k].data.copy_(t.reshape(own[k].shape))
        return self

    def state_dict(self):
        return {k: p.detach().clone() for k, p in self.named_parameters()}

    def finalize(self):
        """Lays the parameters out in one slab (after load_state_dict); the EMA copy starts as a clone of the online weights
        (ema_pytorch deep-copies the model at construction)."""
        self.flat = FlatParams([p for _, p in self.named_parameters()], self.device)
        self.ema = self.flat.data.clone()
        return self

    # ---- overlapped gradient all-reduce -----------------------------------------------------------------------------------
    def reduce_plan(self):
        """(block keys in the order the backward completes them, their slab ranges, the late views) for OverlappedGradReducer: the slab is
        laid out in named_parameters() order - time_embed, input_blocks.*, middle_block, output_blocks.*, out, mixing_logit - and the
        backward finishes it from the end.  `input_blocks.0` and `time_embed` (the head) are left to finish()."""
        if self._reduce_plan is None:
            off, spans, late = 0, {}, []
            for k, p in self.named_parameters():
                n = p.numel()
                parts = k.split(".")
                key = "mixing_logit" if k == "mixing_logit" else (".".join(parts[1:3]) if parts[1] in ("input_blocks", "output_blocks") else parts[1])
                lo, hi = spans.get(key, (off, off))
                assert hi == off, "a block's parameters are contiguous in the slab"
                spans[key] = (lo, off + n)
                if ".emb_layers." in k:
                    late.append((off, off + n))
                off += n
            n_in = len(self.model.input_blocks)
            n_out = len(self.model.output_blocks)
            order = (["mixing_logit"] if "mixing_logit" in spans else []) + ["out"] + [f"output_blocks.{j}" for j in reversed(range(n_out))] + \
                ["middle_block"] + [f"input_blocks.{i}" for i in reversed(range(1, n_in))]
            # a block without parameters (e.g. a Downsample with conv_resample=False) has no span of its own: it completes nothing,
            # its neighbour's range_done() covers the slab up to there
            order = [k for k in order if k in spans and spans[k][1] > spans[k][0]]
            self._reduce_plan = (order, [spans[k] for k in order], late)
        return self._reduce_plan

    def _begin_overlapped_reduce(self, world_size):
        import torch.distributed as dist
        ok = (self.overlap_allreduce and world_size > 1 and dist.is_initialized() and self.accum == 1 and not self.use_graph)
        if not ok:
            self._reducer = None
            return
        order, ranges, late = self.reduce_plan()
        red = OverlappedGradReducer(self.flat.grad, ranges, late, world_size, self.allreduce_bucket_bytes)
        index = {k: i for i, k in enumerate(order)}
        self.model._block_grad_cb = lambda key: red.range_done(index[key]) if key in index else None
        EmbProjFn.pre_backward = red.before_late_writes
        self._reducer = red

    def _end_overlapped_reduce(self):
        self.model._block_grad_cb = None
        EmbProjFn.pre_backward = None

    # ---- one micro-step -------------------------------------------------------------------------------------------------
    def _micro_body(self, z0, t, noise):
        """q_sample -> UNet -> mixed-prediction MSE -> backward, every launch on the current stream, no host synchronisation
        (so that the whole body can also be recorded once in a hipGraph and replayed)."""
        B = z0.shape[0]
        a, s = self.sqrt_ac[t].contiguous(), self.sqrt_1mac[t].contiguous()
        x_t = torch.empty_like(z0)
        _lib.check(_lib.lib().ddmi_train_q_sample(z0.data_ptr(), noise.data_ptr(), a.data_ptr(), s.data_ptr(), x_t.data_ptr(), B,
                                                  z0[0].numel(), _stream()))
        model_out = self.model(x_t, t)
        loss = EpsLossFn.apply(self.model.tc, model_out, x_t, noise, s, self.mixing_logit)
        loss.backward()
        return loss.detach()

    def _graphed_micro(self, z0, t, noise):
        """The micro-step as a hipGraph replay.  ~3000 launches per step leave the host (Python tape + ctypes) as the slower
        side below B = 64 per GPU; a replay costs one launch.  The first call for a batch shape runs eagerly (lazy plans, scratch
        growth, kernel attributes: none of that is legal inside a capture); the second records the body on static input
        buffers - the gradients already land in the static flat slab - and replays it; later calls copy the inputs and replay."""
        key = tuple(z0.shape)
        gen = C.c_int(0)
        _lib.check(_lib.lib().ddmi_train_arena_generation(self.model.tc.h, C.byref(gen)))
        if gen.value != self._graph_generation:   # the scratch arena moved (a larger shape ran since): every recorded graph points
            self._graphs = {k: "warm" for k in self._graphs}   # into freed memory - re-record on the next call of each shape
            self._graph_generation = gen.value
        g = self._graphs.get(key)
        if g is None:
            self._graphs[key] = "warm"
            return self._micro_body(z0, t, noise)
        if g == "warm":
            bufs = (torch.empty_like(z0), torch.empty_like(t), torch.empty_like(noise))
            for dst, src in zip(bufs, (z0, t, noise)):
                dst.copy_(src)
            graph = torch.cuda.CUDAGraph()
            try:
                with torch.cuda.graph(graph):
                    loss = self._micro_body(*bufs)
            except _lib.DdmiError:
                # e.g. the scratch arena had to grow for this shape inside the capture (train_capi.hip reserve() refuses that): the
                # context manager has ended the capture; drop the half-built graph, run this step eagerly (which grows the arena) and
                # record again on the next call of the shape
                del graph
                self._graphs[key] = "warm"
                return self._micro_body(z0, t, noise)
            g = self._graphs[key] = (graph, bufs, loss)
        else:
            for dst, src in zip(g[1], (z0, t, noise)):
                dst.copy_(src)
        g[0].replay()
        return g[2].clone()

    def loss_and_backward(self, z0, t=None, noise=None):
        """p_losses on a latent batch z0 [B,C,H,W] fp32 and the backward pass; gradients ACCUMULATE in the flat slab."""
        if self.flat is None:
            self.finalize()
        z0 = z0.to(self.device, torch.float32).contiguous()
        B = z0.shape[0]
        if t is None:
            t = torch.randint(0, self.num_timesteps, (B,), device=self.device)      # ddpm.py:452
        t = t.to(self.device, torch.int64)
        if noise is None:
            noise = torch.randn_like(z0)                                                 # ddpm.py:410
        noise = noise.to(self.device, torch.float32).contiguous()
        loss = self._graphed_micro(z0, t, noise) if self.use_graph else self._micro_body(z0, t, noise)
        self.micro += 1
        self.current_iters += 1       # tools/ldm/image.py:166
        return loss

    def optimizer_step(self, world_size: int = 1):
        """All-reduce (sum) of the gradient slab over the ranks, fused AdamW, then `ema.update()` (tools/ldm/image.py:171-174)."""
        if self._reducer is not None:      # most of the slab went out during the backward: the head and the late views are left
            self._reducer.finish()
            self._reducer = None
        else:
            allreduce_gradients(self.flat.grad, world_size)
        self.opt_step += 1
        f = self.flat
        _lib.check(_lib.lib().ddmi_train_adamw(f.data.data_ptr(), f.grad.data_ptr(), f.exp_avg.data_ptr(), f.exp_avg_sq.data_ptr(), f.n,
                                               self.lr, self.betas[0], self.betas[1], self.eps, self.wd, self.opt_step,
                                               1.0 / (world_size * self.accum), _stream()))
        f.zero_grad()
        self.micro = 0
        self.ema_update()

    # ---- EMA: ema_pytorch.EMA.update (the reference's dependency, unpinned in README.md:33; absent from /root/reference, restated
    # from its published algorithm): act on calls whose PRE-increment count is a multiple of update_every; copy the online weights
    # while that count <= update_after_step (100); afterwards lerp with the warmed-up decay
    #   1 - (1 + epoch / inv_gamma)^-power,  epoch = count - update_after_step  (the post-increment step - update_after_step - 1),
    # clamped to [min_value, beta] (0.0 for epoch <= 0): ~0.95 at epoch 100, 0.99 at 1000, beta = 0.9999 from epoch 1e6 on.
    def ema_update(self):
        action, decay = ema_pytorch_plan(self.ema_step, self.ema_decay, self.ema_every, self.ema_update_after_step, self.ema_inv_gamma,
                                         self.ema_power, self.ema_min_value)
        self.ema_step += 1
        f = self.flat
        if action == "copy":
            self.ema.copy_(f.data)
        elif action == "lerp":
            if not self.ema_initted:
                self.ema.copy_(f.data)
                self.ema_initted = True
            _lib.check(_lib.lib().ddmi_train_ema(self.ema.data_ptr(), f.data.data_ptr(), f.n, decay, _stream()))

    # ---- checkpoints: the dict LDMTrainer.save writes (tools/ldm/image.py:104-119) ----------------------------------------------
    def _schedule_buffers(self) -> Dict[str, torch.Tensor]:
        """The 12 persistent DDPM buffers (ddpm.py:182-203) that `diffusion_process.state_dict()` carries next to the weights."""
        from .ddpm import DDPM, SCHEDULE_BUFFERS
        d = DDPM(model=None, **self._ddpmconfig)
        return {k: torch.from_numpy(np.ascontiguousarray(getattr(d, k))) for k in SCHEDULE_BUFFERS}

    def checkpoint(self, vaemodel_state=None, mlp_state=None, args=None) -> Dict[str, object]:
        """`diffusion` = DDPM.state_dict() (model.*, mixing_logit, the schedule buffers); `ema` = ema_pytorch's state dict
        (initted, step, online_model.*, ema_model.*); `dae_opt` = torch.optim.AdamW.state_dict() indexed in `parameters()` order
        (optimizer_order: mixing_logit first, then model.*), so the reference's strict loads and its resume accept it.  `step` is
        the reference's EPOCH counter (`while self.step < self.epochs`, tools/ldm/image.py:145) and `current_iters` its micro-step
        counter; the optimizer-step count lives only inside dae_opt's per-parameter `step`."""
        named = list(self.named_parameters())
        bufs = self._schedule_buffers()
        sd = {k: p.detach().clone() for k, p in named}
        sd.update(bufs)
        ema, opt_state, off = {"initted": torch.tensor(bool(self.ema_initted)), "step": torch.tensor(int(self.ema_step))}, {}, 0
        for k, p in named:
            n = p.numel()
            ema["online_model." + k] = p.detach().clone()
            ema["ema_model." + k] = self.ema[off:off + n].view(p.shape).clone()
            off += n
        for i, (k, p, o) in enumerate(self.optimizer_order()):
            n = p.numel()
            opt_state[i] = {"step": torch.tensor(float(self.opt_step)), "exp_avg": self.flat.exp_avg[o:o + n].view(p.shape).clone(),
                            "exp_avg_sq": self.flat.exp_avg_sq[o:o + n].view(p.shape).clone()}
        for k, v in bufs.items():
            ema["online_model." + k] = v.clone()
            ema["ema_model." + k] = v.clone()
        group = {"lr": self.lr, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": self.wd, "amsgrad": False, "maximize": False,
                 "foreach": None, "capturable": False, "differentiable": False, "fused": None, "params": list(range(len(named)))}
        return {"args": dict(args) if args else {"ddpmconfig": dict(self._ddpmconfig)}, "step": int(self.epoch),
                "current_iters": int(self.current_iters), "vaemodel": vaemodel_state or {}, "mlp": mlp_state or {},
                "diffusion": sd, "ema": ema, "dae_opt": {"state": opt_state if self.opt_step > 0 else {}, "param_groups": [group]},
                "scaler": None}

    def save(self, path, vaemodel_state=None, mlp_state=None, args=None):
        torch.save(self.checkpoint(vaemodel_state, mlp_state, args), path)

    def load_training_state(self, ck, resume: bool = True):
        """After finalize(): what `LDMTrainer.load` (resume, tools/ldm/image.py:121-134) or the `pretrained` branch (:76-82) restore
        besides the online weights - the EMA copy (+ its `initted` / `step` counters) for both, AdamW's moments and step plus the
        trainer's `step` for resume.  `dae_opt` is read in torch's {'state', 'param_groups'} layout (by parameter index) or this
        repo's round-2 flat layout ({'exp_avg', 'exp_avg_sq', 'step'})."""
        assert self.flat is not None, "load_training_state comes after finalize()"
        named = list(self.named_parameters())
        ema = ck.get("ema") or {}
        with torch.no_grad():
            if any(k.startswith("ema_model.") for k in ema):
                off = 0
                for k, p in named:
                    n = p.numel()
                    v = ema.get("ema_model." + k)
                    if v is None:
                        raise KeyError(f"checkpoint EMA lacks ema_model.{k}")
                    self.ema[off:off + n].copy_(torch.as_tensor(v).to(self.device, torch.float32).reshape(-1))
                    off += n
                self.ema_initted = bool(ema["initted"]) if "initted" in ema else True
                self.ema_step = int(ema["step"]) if "step" in ema else 0
            if not resume:
                return self
            opt = ck.get("dae_opt") or {}
            if "state" in opt:
                steps = []
                order = self.optimizer_order()
                extra = [i for i in opt["state"] if not (isinstance(i, int) and 0 <= i < len(order))]
                if extra:
                    raise ValueError(f"dae_opt holds state for parameter indices {extra[:5]} but the model has {len(order)} parameters")
                for i, (k, p, o) in enumerate(order):   # index i of AdamW(parameters()): mixing_logit first, then model.*
                    n = p.numel()
                    st = opt["state"].get(i)
                    if st is not None:
                        for key in ("exp_avg", "exp_avg_sq"):
                            if tuple(torch.as_tensor(st[key]).shape) != tuple(p.shape):
                                raise ValueError(f"dae_opt state {i} ({key}) has shape {tuple(torch.as_tensor(st[key]).shape)} but parameter "
                                                 f"{i} of parameters() is '{k}' {tuple(p.shape)}: the checkpoint was written for another "
                                                 "parameter order or model")
                        self.flat.exp_avg[o:o + n].copy_(torch.as_tensor(st["exp_avg"]).to(self.device, torch.float32).reshape(-1))
                        self.flat.exp_avg_sq[o:o + n].copy_(torch.as_tensor(st["exp_avg_sq"]).to(self.device, torch.float32).reshape(-1))
                        steps.append(int(float(st["step"])))
                if steps:
                    assert min(steps) == max(steps), "the fused AdamW keeps ONE step count for every parameter"
                    self.opt_step = steps[0]
                if opt.get("param_groups"):
                    g = opt["param_groups"][0]
                    self.lr, self.betas, self.eps, self.wd = float(g["lr"]), tuple(g["betas"]), float(g["eps"]), float(g["weight_decay"])
            elif "exp_avg" in opt:
                self.flat.exp_avg.copy_(torch.as_tensor(opt["exp_avg"]).to(self.device, torch.float32))
                self.flat.exp_avg_sq.copy_(torch.as_tensor(opt["exp_avg_sq"]).to(self.device, torch.float32))
                self.opt_step = int(opt.get("step", 0))
            # the trainer's own counters: `step` = finished epochs, `current_iters` = micro-steps (tools/ldm/image.py:127-128); the
            # accumulation phase follows from the latter.  Neither is the AdamW bias-correction step (that is dae_opt's).
            self.epoch = int(ck.get("step", 0) or 0)
            self.current_iters = int(ck.get("current_iters", 0) or 0)
            self.micro = self.current_iters % self.accum
        return self

    def train_step(self, z0, t=None, noise=None, world_size: int = 1):
        """The body of the reference's loop for one batch (tools/ldm/image.py:158-175)."""
        if self.flat is None:         # the reducer views the flat gradient slab: lay it out before the hooks are installed
            self.finalize()
        self._begin_overlapped_reduce(world_size)
        try:
            loss = self.loss_and_backward(z0, t, noise)
        except BaseException:
            # a backward that raised leaves buckets in flight over a half-written slab: wait them out and forget the reducer, so that
            # a later optimizer_step() cannot finish() a partial reduction (it falls back to the blocking pass over what is there)
            if self._reducer is not None:
                self._reducer.abort()
                self._reducer = None
            raise
        finally:
            self._end_overlapped_reduce()
        if self.micro % self.accum == 0:
            self.optimizer_step(world_size)
        return loss
