"""`MLP` (image coordinate MLP) - host-side mirror of models/d2c_vae/mlp.py:12-66.

`mlp(coords, hdbf=planes, si=si)` like the reference; plane sampling + all 16 dense layers run as ONE fused HIP kernel.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Tuple

import numpy as np

from . import _lib


def mlp_manifest(in_ch=2, latent_dim=64, out_ch=3, ch=256) -> List[Tuple[str, Tuple[int, ...]]]:
    """(key, shape) of MLP.state_dict() (mlp.py:13-32, blocks.py:187-234, 312-356, 390-400, 604-624)."""
    out: List[Tuple[str, Tuple[int, ...]]] = []
    dim = ch // 4
    out += [("time_mlp.1.weight", (ch, dim)), ("time_mlp.1.bias", (ch,)),
            ("time_mlp.3.weight", (ch, ch)), ("time_mlp.3.bias", (ch,))]

    def styled(p, cin, cout):
        out.extend([(p + "conv.weight", (1, cout, cin, 1, 1)), (p + "conv.modulation.weight", (cin, ch)),
                    (p + "conv.modulation.bias", (cin,)), (p + "noise.weight", (1,)), (p + "activate.bias", (cout,))])

    def block(p, cin, cout):
        styled(p + "conv1.", cin, cout)
        styled(p + "conv2.", cout, cout)
        styled(p + "conv3.", cout, cout)
        if cin != cout:
            out.append((p + "skip.0.weight", (cout, cin, 1, 1)))

    block("net_res1.", in_ch + latent_dim, ch)
    block("net_res2.", ch + in_ch + latent_dim, ch)
    block("net_res3.", ch + in_ch + latent_dim, ch)
    block("net_res4.", ch, ch)
    out += [("torgb.bias", (1, out_ch, 1, 1)), ("torgb.conv.weight", (1, out_ch, ch, 1, 1)),
            ("torgb.conv.modulation.weight", (ch, ch)), ("torgb.conv.modulation.bias", (ch,))]
    return out


class MLP:
    def __init__(self, *, in_ch=2, latent_dim=64, out_ch=3, ch=256, numerics="bf16"):
        self.in_ch, self.latent_dim, self.out_ch, self.ch = int(in_ch), int(latent_dim), int(out_ch), int(ch)
        self.numerics = _lib.numerics_code(numerics)
        self._state: Dict[str, np.ndarray] = {}
        self._handle = None
        self._coords_cache = {}
        self.noise_nonzero = False

    def manifest(self):
        return mlp_manifest(self.in_ch, self.latent_dim, self.out_ch, self.ch)

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
        # NoiseInjection (blocks.py:286-297): weight * N(0,1) per StyledConv; zero at init, possibly not in a checkpoint
        self.noise_nonzero = any(bool(np.any(v != 0)) for k, v in got.items() if k.endswith("noise.weight"))
        self._destroy()
        return self

    def eval(self):
        return self

    def to(self, *a, **k):
        return self

    def _destroy(self):
        if self._handle is not None:
            _lib.lib().ddmi_mlp2d_destroy(self._handle)
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
        cfg = _lib.Mlp2dCfg(self.in_ch, self.out_ch, self.ch, self.latent_dim, self.numerics)
        h = C.c_void_p()
        _lib.check(L.ddmi_mlp2d_create(C.byref(cfg), C.byref(h)))
        try:
            _lib.load_weights(L.ddmi_mlp2d_load_weight, h, self._state)
            _lib.check(L.ddmi_mlp2d_finalize(h))
        except Exception:
            L.ddmi_mlp2d_destroy(h)
            raise
        self._handle = h
        return self

    def forward(self, coords, hdbf, si=1, noise=None, seeds=None):
        """coords [1,2,R,R] (channel 0 = x, 1 = y; convert_to_coord_format_2d), hdbf = 3 planes coarse -> fine,
        si = scale injection.  NoiseInjection (blocks.py:293-295, only relevant for checkpoints whose `noise.weight` are
        non-zero): `noise` [12,B,R*R] = the 12 draws given explicitly, or `seeds` [B] int64 = per-sample Philox seeds, drawn
        inside the kernel; with neither, a per-call torch seed is drawn (the reference draws fresh noise on every call) -
        the term is never dropped.  Returns [B,3,R,R] fp32."""
        import torch
        assert hdbf is not None and len(hdbf) == 3  # mlp.py:42
        dev = hdbf[0].device
        self.finalize(dev.index or 0)
        B = hdbf[0].shape[0]
        R_h, R_w = coords.shape[2], coords.shape[3]
        N = R_h * R_w
        pts = coords[0].permute(1, 2, 0).reshape(N, 2).to(device=dev, dtype=torch.float32).contiguous()
        native = torch.float32 if _lib.is_fp32_storage(self.numerics) else torch.bfloat16
        planes, layouts = [], set()
        for p in hdbf:
            if p.dtype == native and p.is_contiguous(memory_format=torch.channels_last):
                layouts.add(_lib.LAYOUT_NHWC_NATIVE)
                planes.append(p)
            else:
                layouts.add(_lib.LAYOUT_NCHW_F32)
                planes.append(p.float().contiguous())
        if len(layouts) != 1:
            planes = [p.float().contiguous() for p in hdbf]
            layouts = {_lib.LAYOUT_NCHW_F32}
        layout = layouts.pop()
        out = torch.empty((B, self.out_ch, R_h, R_w), device=dev, dtype=torch.float32)
        ptrs = (C.c_void_p * 3)(*[p.data_ptr() for p in planes])
        hw = (C.c_int * 3)(*[int(p.shape[2]) for p in planes])
        if noise is not None:
            nz = torch.as_tensor(noise).to(device=dev, dtype=torch.float32).reshape(12, B, N).contiguous()
            _lib.check(_lib.lib().ddmi_hdpe_mlp2d(self._handle, ptrs, hw, layout, pts.data_ptr(), float(si),
                                                  nz.data_ptr(), out.data_ptr(), B, N, _lib.stream_ptr()))
        elif self.noise_nonzero:
            if seeds is None:  # fresh noise per call, like `new_empty(...).normal_()` on the global generator
                seeds = torch.randint(0, 2 ** 62, (B,), dtype=torch.int64)
            sd = torch.as_tensor(seeds).to(device=dev, dtype=torch.int64).contiguous()
            assert sd.numel() == B, "one seed per sample"
            _lib.check(_lib.lib().ddmi_hdpe_mlp2d_rng(self._handle, ptrs, hw, layout, pts.data_ptr(), float(si),
                                                      sd.data_ptr(), out.data_ptr(), B, N, _lib.stream_ptr()))
        else:
            _lib.check(_lib.lib().ddmi_hdpe_mlp2d(self._handle, ptrs, hw, layout, pts.data_ptr(), float(si), None,
                                                  out.data_ptr(), B, N, _lib.stream_ptr()))
        return out

    __call__ = forward
