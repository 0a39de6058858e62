"""`Autoencoder` (decode only) - host-side mirror of models/d2c_vae/autoencoder_unet.py:834-858.

`decode(z)` returns the HDBF plane list coarse -> fine exactly like the reference, as torch tensors of logical shape
[B, out_ch, P, P].  By default they are *channels-last* tensors in the handle's storage dtype - the layout the fused
plane-sampling kernel reads with coalesced 16-byte loads; `decode(z, reference_layout=True)` returns contiguous NCHW
fp32 like the reference does.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Tuple

import numpy as np

from . import _lib


def decoder_manifest(ddconfig, embed_dim) -> List[Tuple[str, Tuple[int, ...]]]:
    """(key, shape) of the `post_quant_conv.*` and `decoder.*` entries of Autoencoder.state_dict()
    (autoencoder_unet.py:702-788, 846)."""
    ch, out_ch, zc = ddconfig["ch"], ddconfig["out_ch"], ddconfig["z_channels"]
    ch_mult = list(ddconfig["ch_mult"])
    nrb, nres = ddconfig["num_res_blocks"], len(ch_mult)
    attn_res = set(ddconfig.get("attn_resolutions", []))
    hdbf_res = set(ddconfig["hdbf_resolutions"])
    out: List[Tuple[str, Tuple[int, ...]]] = []

    def conv(p, o, i, k):
        out.extend([(p + "weight", (o, i, k, k)), (p + "bias", (o,))])

    def norm(p, c):
        out.extend([(p + "weight", (c,)), (p + "bias", (c,))])

    def res(p, cin, cout):
        norm(p + "norm1.", cin)
        conv(p + "conv1.", cout, cin, 3)
        norm(p + "norm2.", cout)
        conv(p + "conv2.", cout, cout, 3)
        if cin != cout:
            conv(p + "nin_shortcut.", cout, cin, 1)

    def attn(p, c):
        norm(p + "norm.", c)
        for n in ("q.", "k.", "v.", "proj_out."):
            conv(p + n, c, c, 1)

    conv("post_quant_conv.", zc, embed_dim, 1)
    d = "decoder."
    block_in = ch * ch_mult[-1]
    curr = ddconfig["resolution"] // 2 ** (nres - 1)
    conv(d + "conv_in.", block_in, zc, 3)
    res(d + "mid.block_1.", block_in, block_in)
    attn(d + "mid.attn_1.", block_in)
    res(d + "mid.block_2.", block_in, block_in)
    for lvl in reversed(range(nres)):
        block_out = ch * ch_mult[lvl]
        for ib in range(nrb + 1):
            res(f"{d}up.{lvl}.block.{ib}.", block_in, block_out)
            block_in = block_out
            if curr in attn_res:
                attn(f"{d}up.{lvl}.attn.{ib}.", block_in)
        if curr in hdbf_res:
            conv(f"{d}up.{lvl}.hdbf.0.", out_ch, block_in, 1)
        if lvl != 0:
            conv(f"{d}up.{lvl}.upsample.conv.", block_in, block_in, 3)
            curr *= 2
    norm(d + "norm_out.", block_in)
    conv(d + "conv_out.", out_ch, block_in, 3)
    return out


class Autoencoder:
    def __init__(self, ddconfig, embed_dim, image_key="image", numerics="bf16", max_batch=8, latent_size=None):
        self.ddconfig = {k: (list(v) if isinstance(v, (list, tuple)) or type(v).__name__ == "ListConfig" else v)
                         for k, v in dict(ddconfig).items()}
        if self.ddconfig.get("attn_type", "vanilla") != "vanilla":
            raise NotImplementedError("only attn_type 'vanilla' (single head) is on the image path")
        self.embed_dim = int(embed_dim)
        self.numerics = _lib.numerics_code(numerics)
        self.max_batch = int(max_batch)
        # the reference decoder is fully convolutional; the native handle plans its arena for one latent size: by default
        # resolution / 2^(levels-1) (autoencoder_unet.py:745-747), or the size the sampler produces (config 1: 16x16)
        self.latent_size = int(latent_size) if latent_size else self.ddconfig["resolution"] // 2 ** (len(self.ddconfig["ch_mult"]) - 1)
        self._state: Dict[str, np.ndarray] = {}
        self._handle = None
        self.plane_hw: List[int] = []

    def manifest(self):
        return decoder_manifest(self.ddconfig, self.embed_dim)

    def load_state_dict(self, state, strict: bool = True):
        """Accepts a full reference `Autoencoder` state dict; encoder / quant_conv entries (training only) are ignored."""
        want = dict(self.manifest())
        got = {}
        for k, v in state.items():
            if k in want:
                arr = v.detach().float().cpu().numpy() if hasattr(v, "detach") else np.asarray(v, dtype=np.float32)
                if tuple(arr.shape) != want[k]:
                    raise ValueError(f"size mismatch for {k}: {tuple(arr.shape)} vs {want[k]}")
                got[k] = arr
            elif strict and not (k.startswith("encoder.") or k.startswith("quant_conv.")):
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

    def _destroy(self):
        if self._handle is not None:
            _lib.lib().ddmi_decoder2d_destroy(self._handle)
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
        dd = self.ddconfig
        cfg = _lib.Decoder2dCfg()
        cfg.embed_dim, cfg.z_channels, cfg.resolution = self.embed_dim, dd["z_channels"], dd["resolution"]
        cfg.out_ch, cfg.ch, cfg.num_res_blocks = dd["out_ch"], dd["ch"], dd["num_res_blocks"]
        cfg.n_mult = len(dd["ch_mult"])
        for i, m in enumerate(dd["ch_mult"]):
            cfg.ch_mult[i] = int(m)
        ar = list(dd.get("attn_resolutions", []))
        cfg.n_attn = len(ar)
        for i, a in enumerate(ar):
            cfg.attn_resolutions[i] = int(a)
        hr = list(dd["hdbf_resolutions"])
        cfg.n_hdbf = len(hr)
        for i, a in enumerate(hr):
            cfg.hdbf_resolutions[i] = int(a)
        cfg.latent_size = self.latent_size
        cfg.max_batch, cfg.numerics = self.max_batch, self.numerics
        h = C.c_void_p()
        _lib.check(L.ddmi_decoder2d_create(C.byref(cfg), C.byref(h)))
        try:
            _lib.load_weights(L.ddmi_decoder2d_load_weight, h, self._state)
            _lib.check(L.ddmi_decoder2d_finalize(h))
            n = C.c_int()
            hw = (C.c_int * 8)()
            _lib.check(L.ddmi_decoder2d_num_planes(h, C.byref(n), C.byref(hw)))
            self.plane_hw = [int(hw[i]) for i in range(n.value)]
        except Exception:
            L.ddmi_decoder2d_destroy(h)
            raise
        self._handle = h
        self._state = {}
        return self

    def decode(self, z, reference_layout: bool = False):
        import torch
        assert z.is_cuda and z.dtype == torch.float32 and z.dim() == 4
        if z.shape[2] != self.latent_size or z.shape[3] != self.latent_size:
            raise ValueError(f"this decoder handle was planned for {self.latent_size}x{self.latent_size} latents, got {tuple(z.shape[2:])}")
        self.finalize(z.device.index or 0)
        z = z.contiguous()
        B, oc = z.shape[0], self.ddconfig["out_ch"]
        if reference_layout:
            planes = [torch.empty((B, oc, p, p), device=z.device, dtype=torch.float32) for p in self.plane_hw]
            layout = _lib.LAYOUT_NCHW_F32
        else:
            dt = torch.float32 if _lib.is_fp32_storage(self.numerics) else torch.bfloat16
            planes = [torch.empty((B, oc, p, p), device=z.device, dtype=dt, memory_format=torch.channels_last)
                      for p in self.plane_hw]
            layout = _lib.LAYOUT_NHWC_NATIVE
        ptrs = (C.c_void_p * len(planes))(*[p.data_ptr() for p in planes])
        _lib.check(_lib.lib().ddmi_decoder2d_forward(self._handle, z.data_ptr(), ptrs, layout, B, _lib.stream_ptr()))
        return planes
