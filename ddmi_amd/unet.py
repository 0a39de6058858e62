"""`UNetModel` - host-side mirror of models/ldm/modules/diffusionmodules/openaimodel.py:463-793.

Same constructor kwargs (the YAML `unetconfig` block is splatted in, exp/stage.py:202), same state-dict key
names, same call contract `model(x, timesteps, context=None, y=None, **kw) -> eps`; the forward itself runs in
libddmi_hip.so (implicit-GEMM conv / GroupNorm / attention kernels for gfx950).
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Sequence, Tuple

import numpy as np

from . import _lib


def unet_manifest(in_channels, model_channels, out_channels, num_res_blocks, attention_resolutions,
                  channel_mult=(1, 2, 4, 8), **_ignored) -> List[Tuple[str, Tuple[int, ...]]]:
    """(key, shape) list of the reference UNetModel.state_dict() for this configuration (openaimodel.py:557-743)."""
    mc, E = model_channels, 4 * model_channels
    out: List[Tuple[str, Tuple[int, ...]]] = []

    def lin(p, o, i):
        out.extend([(p + "weight", (o, i)), (p + "bias", (o,))])

    def conv(p, o, i, k):
        out.extend([(p + "weight", (o, i, k, k)), (p + "bias", (o,))])

    def norm(p, c):
        out.extend([(p + "weight", (c,)), (p + "bias", (c,))])

    def res(p, cin, cout):
        norm(p + "in_layers.0.", cin)
        conv(p + "in_layers.2.", cout, cin, 3)
        lin(p + "emb_layers.1.", cout, E)
        norm(p + "out_layers.0.", cout)
        conv(p + "out_layers.3.", cout, cout, 3)
        if cin != cout:
            conv(p + "skip_connection.", cout, cin, 1)

    def attn(p, c):
        norm(p + "norm.", c)
        out.extend([(p + "qkv.weight", (3 * c, c, 1)), (p + "qkv.bias", (3 * c,)),
                    (p + "proj_out.weight", (c, c, 1)), (p + "proj_out.bias", (c,))])

    lin("time_embed.0.", E, mc)
    lin("time_embed.2.", E, E)
    conv("input_blocks.0.0.", mc, in_channels, 3)
    chans = [mc]
    ch, ds, blk = mc, 1, 1
    mult = list(channel_mult)
    attn_res = set(attention_resolutions)
    for level, m in enumerate(mult):
        for _ in range(num_res_blocks):
            res(f"input_blocks.{blk}.0.", ch, m * mc)
            ch = m * mc
            if ds in attn_res:
                attn(f"input_blocks.{blk}.1.", ch)
            chans.append(ch)
            blk += 1
        if level != len(mult) - 1:
            conv(f"input_blocks.{blk}.0.op.", ch, ch, 3)
            chans.append(ch)
            ds *= 2
            blk += 1
    res("middle_block.0.", ch, ch)
    attn("middle_block.1.", ch)
    res("middle_block.2.", ch, ch)
    blk = 0
    for level, m in list(enumerate(mult))[::-1]:
        for i in range(num_res_blocks + 1):
            ich = chans.pop()
            res(f"output_blocks.{blk}.0.", ch + ich, mc * m)
            ch = mc * m
            sub = 1
            if ds in attn_res:
                attn(f"output_blocks.{blk}.{sub}.", ch)
                sub += 1
            if level and i == num_res_blocks:
                conv(f"output_blocks.{blk}.{sub}.conv.", ch, ch, 3)
                ds //= 2
            blk += 1
    norm("out.0.", ch)
    conv("out.2.", out_channels, mc, 3)
    return out


class UNetModel:
    """Drop-in for the reference `UNetModel` on the sampling path (2-D, AttentionBlock + QKVAttentionLegacy,
    conv_resample=True; the only variant the DDMI configs instantiate)."""

    def __init__(self, image_size, in_channels, model_channels, out_channels, num_res_blocks, attention_resolutions,
                 dropout=0, channel_mult=(1, 2, 4, 8), conv_resample=True, dims=2, num_classes=None,
                 use_checkpoint=False, use_fp16=False, num_heads=-1, num_head_channels=-1, num_heads_upsample=-1,
                 use_scale_shift_norm=False, resblock_updown=False, use_new_attention_order=False,
                 use_spatial_transformer=False, transformer_depth=1, context_dim=None, n_embed=None, legacy=True,
                 numerics="bf16", max_batch=8, **ignorekwargs):
        unsupported = dict(dims=(dims, 2), num_classes=(num_classes, None), use_scale_shift_norm=(use_scale_shift_norm, False),
                           resblock_updown=(resblock_updown, False), use_new_attention_order=(use_new_attention_order, False),
                           use_spatial_transformer=(use_spatial_transformer, False), conv_resample=(conv_resample, True),
                           n_embed=(n_embed, None))
        for k, (v, want) in unsupported.items():
            if v != want:
                raise NotImplementedError(f"UNetModel({k}={v!r}) is not on the DDMI sampling path (configs use {want!r})")
        if num_head_channels == -1:
            raise NotImplementedError("set num_head_channels (all DDMI configs use 32)")
        self.image_size, self.in_channels, self.model_channels = int(image_size), int(in_channels), int(model_channels)
        self.out_channels, self.num_res_blocks = int(out_channels), int(num_res_blocks)
        self.attention_resolutions = [int(a) for a in attention_resolutions]
        self.channel_mult = [int(m) for m in channel_mult]
        self.num_head_channels = int(num_head_channels)
        self.numerics = _lib.numerics_code(numerics)
        self.max_batch = int(max_batch)
        self._state: Dict[str, np.ndarray] = {}
        self._handle = None

    # ---- state dict ---------------------------------------------------------------------------------
    def manifest(self):
        return unet_manifest(self.in_channels, self.model_channels, self.out_channels, self.num_res_blocks,
                             self.attention_resolutions, self.channel_mult)

    def load_state_dict(self, state, strict: bool = True):
        want = dict(self.manifest())
        got = {}
        for k, v in state.items():
            arr = v.detach().float().cpu().numpy() if hasattr(v, "detach") else np.asarray(v, dtype=np.float32)
            if k in want:
                if tuple(arr.shape) != want[k]:
                    raise ValueError(f"size mismatch for {k}: {tuple(arr.shape)} vs {want[k]}")
                got[k] = arr
            elif strict:
                raise KeyError(f"unexpected key {k}")
        missing = [k for k in want if k not in got]
        if missing and strict:
            raise KeyError(f"missing keys: {missing[:5]}{'...' if len(missing) > 5 else ''}")
        self._state = got
        self._destroy()
        return self

    def eval(self):
        return self

    def to(self, *a, **k):
        return self

    # ---- native handle --------------------------------------------------------------------------------
    def _destroy(self):
        if self._handle is not None:
            _lib.lib().ddmi_unet_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self._destroy()
        except Exception:
            pass

    def finalize(self, device_index: int = 0):
        if self._handle is not None:
            return self
        L = _lib.lib()
        _lib.init_device(device_index)
        cfg = _lib.UnetCfg()
        cfg.image_size, cfg.in_channels, cfg.model_channels = self.image_size, self.in_channels, self.model_channels
        cfg.out_channels, cfg.num_res_blocks = self.out_channels, self.num_res_blocks
        cfg.n_mult = len(self.channel_mult)
        for i, m in enumerate(self.channel_mult):
            cfg.channel_mult[i] = m
        cfg.n_attn = len(self.attention_resolutions)
        for i, a in enumerate(self.attention_resolutions):
            cfg.attention_resolutions[i] = a
        cfg.num_head_channels, cfg.max_batch, cfg.numerics = self.num_head_channels, self.max_batch, self.numerics
        h = C.c_void_p()
        _lib.check(self._create(L, cfg, h))
        try:
            _lib.load_weights(L.ddmi_unet_load_weight, h, self._state)
            _lib.check(L.ddmi_unet_finalize(h))
        except Exception:
            L.ddmi_unet_destroy(h)
            raise
        self._handle = h
        self._state = {}  # packed on device now
        if getattr(self, "_uniform_t", False):
            _lib.check(L.ddmi_unet_set_uniform_t(h, 1))
        if getattr(self, "_use_graph", False):
            _lib.check(L.ddmi_unet_set_graph(h, 1))
        return self

    def _create(self, L, cfg, h):
        return L.ddmi_unet_create(C.byref(cfg), C.byref(h))

    def memory(self):
        out = (C.c_uint64 * 3)()
        _lib.check(_lib.lib().ddmi_unet_memory(self._handle, C.byref(out)))
        return {"weights": int(out[0]), "arena": int(out[1]), "splitk": int(out[2])}

    def async_status(self, device_wide: bool = False):
        """Deferred errors of the asynchronous forwards (ddmi_unet_async_status_on: synchronises torch's CURRENT stream - the one `forward()`
        issues on; `device_wide=True`: ddmi_unet_async_status, the whole device): raises when a fused GroupNorm
        launch of a bf16 network timed out waiting for a partner tile - the forwards since the last call are invalid then, the library has
        switched the fusion off for this handle and the caller runs them again (`DDPM.sample` does, once); returns how many convolutions
        of the last forward carried their GroupNorm in the epilogue.  The sampling loops call it once per chain, at the point where they
        synchronise anyway; a caller of `forward()` / `__call__` on its own calls it wherever it copies results to the host (if it never
        does, the first `forward()` after a timeout raises the same error instead of running)."""
        n = C.c_int(0)
        if self._handle is not None:
            if device_wide:
                _lib.check(_lib.lib().ddmi_unet_async_status(self._handle, C.byref(n)))
            else:
                _lib.check(_lib.lib().ddmi_unet_async_status_on(self._handle, _lib.stream_ptr(), C.byref(n)))
        return int(n.value)

    # ---- forward ----------------------------------------------------------------------------------------
    def set_uniform_t(self, enabled: bool):
        """Sampling loops feed the same timestep to every chain of the batch; tell the library so."""
        self._uniform_t = bool(enabled)
        if self._handle is not None:
            _lib.check(_lib.lib().ddmi_unet_set_uniform_t(self._handle, int(enabled)))
        return self

    def set_graph(self, enabled: bool):
        """Replay the forward as a hipGraph for repeated (batch, buffers) - small-batch latency; see ddmi_unet_set_graph."""
        self._use_graph = bool(enabled)
        if self._handle is not None:
            _lib.check(_lib.lib().ddmi_unet_set_graph(self._handle, int(enabled)))
        return self

    def forward(self, x, timesteps=None, context=None, y=None, out=None, **kwargs):
        import torch
        assert context is None and y is None, "unconditional sampling only (ddpm.py:304-308)"
        assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 4
        if x.shape[2] != self.image_size or x.shape[3] != self.image_size:
            raise ValueError(f"this UNetModel handle was planned for {self.image_size}x{self.image_size} inputs, got "
                             f"{tuple(x.shape[2:])}: construct it with image_size = the latent size")
        self.finalize(x.device.index or 0)
        x = x.contiguous()
        t = timesteps.to(device=x.device, dtype=torch.int64).contiguous()
        B = x.shape[0]
        if out is None:
            out = torch.empty((B, self.out_channels, x.shape[2], x.shape[3]), device=x.device, dtype=torch.float32)
        _lib.check(_lib.lib().ddmi_unet_forward(self._handle, x.data_ptr(), t.data_ptr(), out.data_ptr(), B, _lib.stream_ptr()))
        return out

    __call__ = forward
