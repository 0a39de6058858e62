"""Video (SkyTimelapse) path mirrors: `UNetModel_Triplane` (models/ldm/modules/diffusionmodules/openaimodel.py:796-1221),
`VITAutoencoder.decode` (models/d2c_vae/autoencoder_vit.py:188-202) over `VideoDecoder_light`
(models/d2c_vae/autoencoder_unet.py:925-1144), `MLPVideo` (models/d2c_vae/mlp.py:114-157) and the coordinate grids of
`convert_to_coord_format_3d` (utils/general_utils.py:38-52).  All compute runs in libddmi_hip.so.

Attention semantics are those of the reference WITH xformers (which its video configuration requires):
softmax(q k^T / sqrt(head_dim)) v - see SURVEY.md 8a rows a14 / a17."""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Sequence, Tuple

import numpy as np

from . import _lib
from .unet import UNetModel, unet_manifest
from .occupancy import mlp3d_manifest


def _attn1d_keys(p, c, cq):
    return [(p + "norm.weight", (c,)), (p + "norm.bias", (c,)),
            (p + "q.weight", (cq, c, 1)), (p + "q.bias", (cq,)), (p + "k.weight", (cq, c, 1)), (p + "k.bias", (cq,)),
            (p + "v.weight", (cq, c, 1)), (p + "v.bias", (cq,)),
            (p + "proj_out.weight", (c, cq, 1)), (p + "proj_out.bias", (c,))]


def unet3_manifest(in_channels, model_channels, out_channels, num_res_blocks, attention_resolutions,
                   channel_mult=(1, 2, 4, 8), **_ignored) -> List[Tuple[str, Tuple[int, ...]]]:
    """(key, shape) list of UNetModel_Triplane.state_dict(): the UNetModel entries plus one 1-D attention block after
    every input block but the first (`input_attns.N`), after the middle block (`mid_attn`) and after every output block
    (`output_attns.N`) - openaimodel.py:912, 955-957, 984-986, 1023, 1082."""
    out = list(unet_manifest(in_channels, model_channels, out_channels, num_res_blocks, attention_resolutions, channel_mult))
    mc, mult = model_channels, list(channel_mult)
    blk, ch = 1, mc
    for level, m in enumerate(mult):
        for _ in range(num_res_blocks):
            ch = m * mc
            out += _attn1d_keys(f"input_attns.{blk}.", ch, ch)
            blk += 1
        if level != len(mult) - 1:
            out += _attn1d_keys(f"input_attns.{blk}.", ch, ch)
            blk += 1
    out += _attn1d_keys("mid_attn.", ch, ch)
    blk = 0
    for level, m in list(enumerate(mult))[::-1]:
        for _ in range(num_res_blocks + 1):
            out += _attn1d_keys(f"output_attns.{blk}.", mc * m, mc * m)
            blk += 1
    return out


class UNetModel_Triplane(UNetModel):
    """Drop-in for the reference `UNetModel_Triplane` on the sampling path: `model(x [B,C,L], timesteps) -> [B,C,L]`
    with L = size1*size2 + size1*size3 + size2*size3 (xy | xt | yt)."""

    def __init__(self, in_channels, model_channels, out_channels, num_res_blocks, attention_resolutions,
                 attn_type="vanilla-1d-multihead", size1=64, size2=64, size3=16, **kw):
        if attn_type != "vanilla-1d-multihead":
            raise NotImplementedError("UNetModel_Triplane: only attn_type='vanilla-1d-multihead' (the default every config uses)")
        super().__init__(image_size=size1, in_channels=in_channels, model_channels=model_channels, out_channels=out_channels,
                         num_res_blocks=num_res_blocks, attention_resolutions=attention_resolutions, **kw)
        self.size1, self.size2, self.size3 = int(size1), int(size2), int(size3)
        if self.size1 != self.size2:
            raise NotImplementedError("size1 must equal size2 (true for every reference configuration)")
        self.seq_len = self.size1 * self.size2 + self.size1 * self.size3 + self.size2 * self.size3

    def manifest(self):
        return unet3_manifest(self.in_channels, self.model_channels, self.out_channels, self.num_res_blocks,
                              self.attention_resolutions, self.channel_mult)

    def _create(self, L, cfg, h):
        return L.ddmi_unet3_create(C.byref(cfg), self.size1, self.size2, self.size3, C.byref(h))

    def forward(self, x, timesteps=None, cond=None, context=None, y=None, out=None, **kwargs):
        import torch
        assert cond is None and context is None and y is None, "unconditional sampling only"
        assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 3 and x.shape[2] == self.seq_len
        self.finalize(x.device.index or 0)
        x = x.contiguous()
        t = timesteps.to(device=x.device, dtype=torch.int64).contiguous()
        B = x.shape[0]
        if out is None:
            out = torch.empty((B, self.out_channels, self.seq_len), device=x.device, dtype=torch.float32)
        _lib.check(_lib.lib().ddmi_unet_forward(self._handle, x.data_ptr(), t.data_ptr(), out.data_ptr(), B, _lib.stream_ptr()))
        return out

    __call__ = forward


# --------------------------------------------------------------------------------------------------------------
def decoderv_manifest(ddconfig, embed_dim) -> List[Tuple[str, Tuple[int, ...]]]:
    """(key, shape) of the `post_*` and `decoder.*` entries of VITAutoencoder.state_dict()
    (autoencoder_vit.py:142-146, autoencoder_unet.py:925-1038)."""
    dd = ddconfig
    out: List[Tuple[str, Tuple[int, ...]]] = []
    zc, ch, oc = dd["z_channels"], dd["ch"], dd["out_ch"]
    ch_mult, nrb = list(dd["ch_mult"]), dd["num_res_blocks"]

    def norm(p, c):
        out.extend([(p + "weight", (c,)), (p + "bias", (c,))])

    def conv(p, o, i, k):
        out.extend([(p + "weight", (o, i, k, k)), (p + "bias", (o,))])

    def res(p, cin, cout):
        norm(p + "norm1.", cin); conv(p + "conv1.", cout, cin, 3); norm(p + "norm2.", cout); conv(p + "conv2.", cout, cout, 3)
        if cin != cout:
            conv(p + "nin_shortcut.", cout, cin, 1)

    def attn2d(p, c):
        norm(p + "norm.", c)
        for n in ("q.", "k.", "v.", "proj_out."):
            conv(p + n, c, c, 1)

    for n in ("xy", "xt", "yt"):
        conv(f"post_{n}.", zc, embed_dim, 1)
    block_in = ch * ch_mult[-1]
    conv("decoder.conv_in.", block_in, zc, 3)
    res("decoder.mid.block_1.", block_in, block_in)
    attn2d("decoder.mid.attn_1.", block_in)
    res("decoder.mid.block_2.", block_in, block_in)
    out.extend(_attn1d_keys("decoder.mid_attn.", block_in, 8 * block_in))
    curr = dd["resolution"] // 2 ** (len(ch_mult) - 1)
    attn_res, hdbf_res = set(dd.get("attn_resolutions", [])), set(dd["hdbf_resolutions"])
    inter_res = set(dd["inter_attn_resolutions"])
    for lvl in reversed(range(len(ch_mult))):
        block_out = ch * ch_mult[lvl]
        for ib in range(nrb + 1):
            res(f"decoder.up.{lvl}.block.{ib}.", block_in, block_out)
            block_in = block_out
            if curr in attn_res:
                attn2d(f"decoder.up.{lvl}.attn.{ib}.", block_in)
        if curr in inter_res:
            out.extend(_attn1d_keys(f"decoder.up.{lvl}.inter_attn.0.", block_in, 8 * block_in))
        if curr in hdbf_res:
            conv(f"decoder.up.{lvl}.hdbf.0.", oc, block_in, 1)
        if lvl != 0:
            conv(f"decoder.up.{lvl}.upsample.conv.", block_in, block_in, 3)
            curr *= 2
    norm("decoder.norm_out.", block_in)
    conv("decoder.conv_out.", oc, block_in, 3)
    return out


class VITAutoencoder:
    """decode(z [B, embed_dim, L]) -> (xy, yt, xt), each a list of HDBF planes coarse -> fine, like the reference
    (only the decode side is on the sampling path; `encoder.*` / TimeSformer keys of a checkpoint are ignored)."""

    def __init__(self, ddconfig, embed_dim, frames, numerics="bf16", max_batch=2, **ignore):
        self.ddconfig = {k: (list(v) if isinstance(v, (list, tuple)) or type(v).__name__ == "ListConfig" else v)
                         for k, v in dict(ddconfig).items()}
        self.embed_dim = int(embed_dim)
        self.frames = int(frames) // int(self.ddconfig.get("splits", 1))
        self.res = int(self.ddconfig["resolution"])
        self.numerics = _lib.numerics_code(numerics)
        self.max_batch = int(max_batch)
        self._state: Dict[str, np.ndarray] = {}
        self._handle = None
        self.plane_hw: List[int] = []
        self.out_ch = int(self.ddconfig["out_ch"])
        s = self.res // 8
        self.seq_len = s * s + 2 * s * self.frames

    def manifest(self):
        return decoderv_manifest(self.ddconfig, self.embed_dim)

    def load_state_dict(self, state, strict: bool = True):
        want = dict(self.manifest())
        got = {}
        for k, v in state.items():
            if k in want:
                arr = v.detach().float().cpu().numpy() if hasattr(v, "detach") else np.asarray(v, dtype=np.float32)
                if tuple(arr.shape) != want[k]:
                    raise ValueError(f"size mismatch for {k}: {tuple(arr.shape)} vs {want[k]}")
                got[k] = arr
            elif strict and (k.startswith("decoder.") or k.startswith("post_")):
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
            _lib.lib().ddmi_decoderv_destroy(self._handle)
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
        heads = {"vanilla": 1, "vanilla-multihead": 16}.get(dd.get("attn_type", "vanilla"))
        if heads is None:
            raise NotImplementedError(f"attn_type {dd.get('attn_type')!r} is not used by any video configuration")
        cfg = _lib.DecoderVCfg()
        cfg.embed_dim, cfg.z_channels, cfg.resolution, cfg.frames = self.embed_dim, dd["z_channels"], self.res, self.frames
        cfg.ch, cfg.num_res_blocks = dd["ch"], dd["num_res_blocks"]
        for name, key in (("ch_mult", "ch_mult"), ("attn_resolutions", "attn_resolutions"),
                          ("hdbf_resolutions", "hdbf_resolutions"), ("inter_attn_resolutions", "inter_attn_resolutions")):
            vals = [int(v) for v in dd.get(key, [])]
            setattr(cfg, {"ch_mult": "n_mult", "attn_resolutions": "n_attn", "hdbf_resolutions": "n_hdbf",
                          "inter_attn_resolutions": "n_inter"}[name], len(vals))
            arr = getattr(cfg, name)
            for i, v in enumerate(vals):
                arr[i] = v
        cfg.attn_heads, cfg.max_batch, cfg.numerics = heads, self.max_batch, self.numerics
        h = C.c_void_p()
        _lib.check(L.ddmi_decoderv_create(C.byref(cfg), C.byref(h)))
        try:
            _lib.load_weights(L.ddmi_decoderv_load_weight, h, self._state)
            _lib.check(L.ddmi_decoderv_finalize(h))
            n, oc = C.c_int(), C.c_int()
            hw = (C.c_int * 4)()
            _lib.check(L.ddmi_decoderv_num_scales(h, C.byref(n), C.byref(hw), C.byref(oc)))
            self.plane_hw = [int(hw[i]) for i in range(n.value)]
            self.out_ch = int(oc.value)
        except Exception:
            L.ddmi_decoderv_destroy(h)
            raise
        self._handle = h
        self._state = {}
        return self

    def decode(self, x, reference_layout: bool = False):
        import torch
        assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 3 and x.shape[2] == self.seq_len
        self.finalize(x.device.index or 0)
        x = x.contiguous()
        B, oc, T = x.shape[0], self.out_ch, self.frames
        if reference_layout:
            mk = lambda hh, ww: torch.empty((B, oc, hh, ww), device=x.device, dtype=torch.float32)
            layout = _lib.LAYOUT_NCHW_F32
        else:
            dt = torch.float32 if _lib.is_fp32_storage(self.numerics) else torch.bfloat16
            mk = lambda hh, ww: torch.empty((B, oc, hh, ww), device=x.device, dtype=dt, memory_format=torch.channels_last)
            layout = _lib.LAYOUT_NHWC_NATIVE
        planes = [[mk(p, p) for p in self.plane_hw], [mk(T, p) for p in self.plane_hw], [mk(T, p) for p in self.plane_hw]]
        flat = [t for lst in planes for t in lst]
        ptrs = (C.c_void_p * len(flat))(*[t.data_ptr() for t in flat])
        _lib.check(_lib.lib().ddmi_decoderv_forward(self._handle, x.data_ptr(), ptrs, layout, B, _lib.stream_ptr()))
        return tuple(planes)  # (xy, yt, xt)


# --------------------------------------------------------------------------------------------------------------
def mlpv_manifest(in_ch=2, latent_dim=64, out_ch=3, ch=256) -> List[Tuple[str, Tuple[int, ...]]]:
    """(key, shape) of MLPVideo.state_dict() (mlp.py:114-126): the MLP3D layout without net_p, 3*latent_dim features."""
    return [(k, v) for k, v in mlp3d_manifest(in_ch, 3 * latent_dim, out_ch, ch) if not k.startswith("net_p.")]


def convert_to_coord_format_3d(b, h, w, t, device="cpu", hstart=-1, hend=1, wstart=-1, wend=1, tstart=-1, tend=1):
    """utils/general_utils.py:38-52: {'xy': [b,2,h,w] (x, y), 'xt': [b,2,t,w] (t, x), 'yt': [b,2,t,h] (t, y)}."""
    import torch
    xs = torch.linspace(wstart, wend, w, device=device)
    ys = torch.linspace(hstart, hend, h, device=device)
    ts = torch.linspace(tstart, tend, t, device=device)
    rep = lambda a: a[None].repeat(b, 1, 1, 1)
    return {"xy": rep(torch.stack((xs.view(1, w).expand(h, w), ys.view(h, 1).expand(h, w)))),
            "xt": rep(torch.stack((ts.view(t, 1).expand(t, w), xs.view(1, w).expand(t, w)))),
            "yt": rep(torch.stack((ts.view(t, 1).expand(t, h), ys.view(1, h).expand(t, h))))}


def video_generation_coords(res: int, frames: int = 16, device="cpu"):
    """The grid `generate()` builds (tools/ldm/video.py:216-217), literal `wend = 255/res` included."""
    e = (res - 1) / res
    return convert_to_coord_format_3d(1, res, res, frames, device=device, hstart=-e, hend=e, wstart=-e, wend=255 / res,
                                      tstart=-15 / 16, tend=15 / 16)


class MLPVideo:
    def __init__(self, *, in_ch=2, latent_dim=64, out_ch=3, ch=256, numerics="bf16", **ignore_kwargs):
        self.in_ch, self.latent_dim, self.out_ch, self.ch = int(in_ch), int(latent_dim), int(out_ch), int(ch)
        self.numerics = _lib.numerics_code(numerics)
        self._state: Dict[str, np.ndarray] = {}
        self._handle = None

    def manifest(self):
        return mlpv_manifest(self.in_ch, self.latent_dim, self.out_ch, self.ch)

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

    def _destroy(self):
        if self._handle is not None:
            _lib.lib().ddmi_mlpv_destroy(self._handle)
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
        _lib.check(L.ddmi_mlpv_create(C.byref(cfg), C.byref(h)))
        try:
            _lib.load_weights(L.ddmi_mlpv_load_weight, h, self._state)
            _lib.check(L.ddmi_mlpv_finalize(h))
        except Exception:
            L.ddmi_mlpv_destroy(h)
            raise
        self._handle = h
        return self

    def forward(self, coords, hdbf):
        """coords: the dict of convert_to_coord_format_3d (batch 1); hdbf = (xy, yt, xt) plane lists as returned by
        VITAutoencoder.decode -> [B, out_ch, t, h, w]."""
        import torch
        assert len(hdbf) == 3 and all(len(p) == 3 for p in hdbf)  # mlp.py:129-133
        xy, yt, xt = hdbf
        dev = xy[0].device
        self.finalize(dev.index or 0)
        B = xy[0].shape[0]
        _, _, H, W = coords["xy"].shape
        T = coords["yt"].shape[2]
        frames = yt[0].shape[2]
        grids = [coords[k][0].to(device=dev, dtype=torch.float32).contiguous() for k in ("xy", "yt", "xt")]
        native = xy[0].dtype != torch.float32 or xy[0].is_contiguous(memory_format=torch.channels_last) and not xy[0].is_contiguous()
        flat = []
        for lst in (xy, yt, xt):
            for pl in lst:
                if native:
                    assert pl.is_contiguous(memory_format=torch.channels_last)
                    flat.append(pl)
                else:
                    flat.append(pl.contiguous())
        layout = _lib.LAYOUT_NHWC_NATIVE if native else _lib.LAYOUT_NCHW_F32
        ptrs = (C.c_void_p * 9)(*[t.data_ptr() for t in flat])
        hw = (C.c_int * 3)(*[int(p.shape[3]) for p in xy])
        out = torch.empty((B, self.out_ch, T, H, W), device=dev, dtype=torch.float32)
        _lib.check(_lib.lib().ddmi_hdpe_mlpv(self._handle, ptrs, hw, int(frames), layout, grids[0].data_ptr(),
                                             grids[1].data_ptr(), grids[2].data_ptr(), out.data_ptr(), B, T, H, W,
                                             _lib.stream_ptr()))
        return out

    __call__ = forward
