"""Occupancy (ShapeNet) path mirrors: `Autoencoder3D.decode` (models/d2c_vae/autoencoder_unet.py:877-916),
`MLP3D` (models/d2c_vae/mlp.py:69-111) and the dense query grid of `Generator3D.generate_mesh_fromdiffusion`
(convocc/src/conv_onet/generation.py:87-97, convocc/src/common.py:145-164)."""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Sequence, Tuple

import numpy as np

from . import _lib
from .autoencoder import decoder_manifest


def decoder3d_manifest(ddconfig, embed_dim) -> List[Tuple[str, Tuple[int, ...]]]:
    """(key, shape) of the `post_quant_conv_*` and `decoder.*` entries of Autoencoder3D.state_dict()
    (autoencoder_unet.py:1348-1459, 891-893)."""
    base = dict(decoder_manifest(ddconfig, embed_dim))
    out: List[Tuple[str, Tuple[int, ...]]] = []
    zc, ch = ddconfig["z_channels"], ddconfig["ch"]
    ch_mult = list(ddconfig["ch_mult"])
    for n in ("xy", "yz", "xz"):
        out += [(f"post_quant_conv_{n}.weight", (zc, embed_dim, 1, 1)), (f"post_quant_conv_{n}.bias", (zc,))]
    out += [(k, v) for k, v in base.items() if k.startswith("decoder.")]

    def norm(p, c):
        out.extend([(p + "weight", (c,)), (p + "bias", (c,))])

    def conv(p, o, i, k):
        out.extend([(p + "weight", (o, i, k, k)), (p + "bias", (o,))])

    def res(p, c):
        norm(p + "norm1.", c); conv(p + "conv1.", c, c, 3); norm(p + "norm2.", c); conv(p + "conv2.", c, c, 3)

    def attn(p, c):
        norm(p + "norm.", c)
        for n in ("q.", "k.", "v.", "proj_out."):
            conv(p + n, c, c, 1)

    top = 3 * ch * ch_mult[-1]
    res("decoder.mid.block_3.", top); attn("decoder.mid_attn.", top); res("decoder.mid.block_4.", top)
    curr = ddconfig["resolution"] // 2 ** (len(ch_mult) - 1)
    inter = set(ddconfig["inter_attn_resolutions"])
    for lvl in reversed(range(len(ch_mult))):
        c3 = 3 * ch * ch_mult[lvl]
        if curr in inter:
            res(f"decoder.up.{lvl}.inter_attn.0.", c3); attn(f"decoder.up.{lvl}.inter_attn.1.", c3); res(f"decoder.up.{lvl}.inter_attn.2.", c3)
        if lvl != 0:
            curr *= 2
    return out


class Autoencoder3D:
    """decode(z) -> (xy, yz, xz), each a list of HDBF planes coarse -> fine, like the reference."""

    def __init__(self, ddconfig, embed_dim, image_key="image", numerics="bf16", max_batch=4):
        self.ddconfig = {k: (list(v) if isinstance(v, (list, tuple)) or type(v).__name__ == "ListConfig" else v)
                         for k, v in dict(ddconfig).items()}
        self.embed_dim = int(embed_dim)
        self.numerics = _lib.numerics_code(numerics)
        self.max_batch = int(max_batch)
        self._state: Dict[str, np.ndarray] = {}
        self._handle = None
        self.plane_hw: List[int] = []

    def manifest(self):
        return decoder3d_manifest(self.ddconfig, self.embed_dim)

    def load_state_dict(self, state, strict: bool = True):
        want = dict(self.manifest())
        got = {}
        for k, v in state.items():
            if k in want:
                arr = v.detach().float().cpu().numpy() if hasattr(v, "detach") else np.asarray(v, dtype=np.float32)
                if tuple(arr.shape) != want[k]:
                    raise ValueError(f"size mismatch for {k}: {tuple(arr.shape)} vs {want[k]}")
                got[k] = arr
            elif strict and not (k.startswith("encoder.") or k.startswith("quant_conv")):
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
            _lib.lib().ddmi_decoder3d_destroy(self._handle)
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
        cfg.latent_size = dd["resolution"] // 2 ** (len(dd["ch_mult"]) - 1)
        cfg.max_batch, cfg.numerics = self.max_batch, self.numerics
        inter = [int(v) for v in dd["inter_attn_resolutions"]]
        inter_arr = (C.c_int * max(1, len(inter)))(*inter)
        h = C.c_void_p()
        _lib.check(L.ddmi_decoder3d_create(C.byref(cfg), inter_arr, len(inter), C.byref(h)))
        try:
            _lib.load_weights(L.ddmi_decoder3d_load_weight, h, self._state)
            _lib.check(L.ddmi_decoder3d_finalize(h))
            n = C.c_int()
            hw = (C.c_int * 4)()
            _lib.check(L.ddmi_decoder3d_num_scales(h, C.byref(n), C.byref(hw)))
            self.plane_hw = [int(hw[i]) for i in range(n.value)]
        except Exception:
            L.ddmi_decoder3d_destroy(h)
            raise
        self._handle = h
        self._state = {}
        return self

    def decode(self, x, reference_layout: bool = False):
        import torch
        assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 4
        self.finalize(x.device.index or 0)
        x = x.contiguous()
        B, oc = x.shape[0], self.ddconfig["out_ch"]
        if reference_layout:
            mk = lambda p: torch.empty((B, oc, p, p), device=x.device, dtype=torch.float32)
            layout = _lib.LAYOUT_NCHW_F32
        else:
            dt = torch.float32 if _lib.is_fp32_storage(self.numerics) else torch.bfloat16
            mk = lambda p: torch.empty((B, oc, p, p), device=x.device, dtype=dt, memory_format=torch.channels_last)
            layout = _lib.LAYOUT_NHWC_NATIVE
        planes = [[mk(p) for p in self.plane_hw] for _ in range(3)]
        flat = [t for lst in planes for t in lst]
        ptrs = (C.c_void_p * len(flat))(*[t.data_ptr() for t in flat])
        _lib.check(_lib.lib().ddmi_decoder3d_forward(self._handle, x.data_ptr(), ptrs, layout, B, _lib.stream_ptr()))
        return tuple(planes)  # (xy, yz, xz)


def mlp3d_manifest(in_ch=3, latent_dim=64, out_ch=1, ch=256) -> List[Tuple[str, Tuple[int, ...]]]:
    """(key, shape) of MLP3D.state_dict() (mlp.py:70-80, blocks.py:682-705)."""
    out: List[Tuple[str, Tuple[int, ...]]] = [("net_p.weight", (ch, in_ch)), ("net_p.bias", (ch,))]

    def rb(p, cin, cout):
        h = min(cin, cout)
        out.extend([(p + "fc_0.weight", (h, cin)), (p + "fc_0.bias", (h,)), (p + "fc_1.weight", (cout, h)), (p + "fc_1.bias", (cout,))])
        if cin != cout:
            out.append((p + "shortcut.weight", (cout, cin)))

    rb("net_res1.", latent_dim, ch)
    rb("net_res2.", ch + latent_dim, ch)
    rb("net_res3.", ch + latent_dim, ch)
    rb("net_res4.", ch, ch)
    out += [("net_out.weight", (out_ch, ch)), ("net_out.bias", (out_ch,))]
    return out


class OccupancyLogits:
    """What the reference returns is `dist.Bernoulli(logits=...)` (mlp.py:111); callers only read `.logits`
    (generation.py:139)."""

    def __init__(self, logits):
        self.logits = logits

    @property
    def probs(self):
        return self.logits.sigmoid()


class MLP3D:
    def __init__(self, *, in_ch=3, latent_dim=64, out_ch=1, ch=256, numerics="bf16"):
        self.in_ch, self.latent_dim, self.out_ch, self.ch = int(in_ch), int(latent_dim), int(out_ch), int(ch)
        self.numerics = _lib.numerics_code(numerics)
        self._state: Dict[str, np.ndarray] = {}
        self._handle = None

    def manifest(self):
        return mlp3d_manifest(self.in_ch, self.latent_dim, self.out_ch, self.ch)

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
            _lib.lib().ddmi_mlp3d_destroy(self._handle)
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
        _lib.check(L.ddmi_mlp3d_create(C.byref(cfg), C.byref(h)))
        try:
            _lib.load_weights(L.ddmi_mlp3d_load_weight, h, self._state)
            _lib.check(L.ddmi_mlp3d_finalize(h))
        except Exception:
            L.ddmi_mlp3d_destroy(h)
            raise
        self._handle = h
        return self

    def forward(self, coords, hdbf) -> OccupancyLogits:
        """coords [B,N,3]; hdbf = (xy, yz, xz) plane lists (3 scales each) as returned by Autoencoder3D.decode."""
        import torch
        assert len(hdbf) == 3 and all(len(p) == 3 for p in hdbf)  # mlp.py:83-87
        dev = hdbf[0][0].device
        self.finalize(dev.index or 0)
        pts = coords.to(device=dev, dtype=torch.float32).contiguous()
        B, N = pts.shape[0], pts.shape[1]
        native = torch.float32 if _lib.is_fp32_storage(self.numerics) else torch.bfloat16
        flat = [t for lst in hdbf for t in lst]
        if all(t.dtype == native and t.is_contiguous(memory_format=torch.channels_last) for t in flat):
            layout = _lib.LAYOUT_NHWC_NATIVE
        else:
            flat = [t.float().contiguous() for t in flat]
            layout = _lib.LAYOUT_NCHW_F32
        out = torch.empty((B, N), device=dev, dtype=torch.float32)
        ptrs = (C.c_void_p * 9)(*[t.data_ptr() for t in flat])
        hw = (C.c_int * 3)(*[int(t.shape[2]) for t in hdbf[0]])
        _lib.check(_lib.lib().ddmi_hdpe_mlp3d(self._handle, ptrs, hw, layout, pts.data_ptr(), out.data_ptr(), B, N,
                                              _lib.stream_ptr()))
        return OccupancyLogits(out)

    __call__ = forward


def make_3d_grid(bb_min, bb_max, shape):
    """convocc/src/common.py:145-164 (x slowest, z fastest)."""
    import torch
    size = shape[0] * shape[1] * shape[2]
    pxs = torch.linspace(bb_min[0], bb_max[0], shape[0]).view(-1, 1, 1).expand(*shape).contiguous().view(size)
    pys = torch.linspace(bb_min[1], bb_max[1], shape[1]).view(1, -1, 1).expand(*shape).contiguous().view(size)
    pzs = torch.linspace(bb_min[2], bb_max[2], shape[2]).view(1, 1, -1).expand(*shape).contiguous().view(size)
    return torch.stack([pxs, pys, pzs], dim=1)


_GRID_CACHE: Dict[Tuple[int, float, str], "object"] = {}


def occupancy_grid_logits(mlp: MLP3D, planes, nx: int, padding: float = 0.1):
    """generate_mesh_fromdiffusion's dense branch (generation.py:87-97) in ONE launch (the reference chunks the grid into
    100 000-point pieces, generation.py:41,130): logits on the (1+padding)-scaled nx^3 grid -> [B, nx, nx, nx].
    The query grid is a constant of (nx, padding): it is built once per device and kept there (building 128^3 points on
    the host and copying them took 26 ms of a 33 ms call; the fused kernel itself takes 6.6 ms)."""
    dev = planes[0][0].device
    B = planes[0][0].shape[0]
    key = (int(nx), float(padding), str(dev))
    pts = _GRID_CACHE.get(key)
    if pts is None:
        pts = ((1 + padding) * make_3d_grid((-0.5,) * 3, (0.5,) * 3, (nx,) * 3)).to(dev).contiguous()
        _GRID_CACHE[key] = pts
    logits = mlp(pts[None].expand(B, -1, -1).contiguous(), planes).logits
    return logits.reshape(B, nx, nx, nx)


class TriMesh:
    """What Generator3D hands back in place of trimesh.Trimesh(vertices, triangles, process=False) (generation.py:195-197;
    trimesh is not a dependency here): the arrays, and `export` for the Wavefront .obj the sampling loops write
    (tools/ldm/occupancy.py:218,228)."""

    def __init__(self, vertices, faces):
        self.vertices, self.faces = vertices, faces

    def export(self, path: str):
        import numpy as np
        with open(path, "w") as f:
            np.savetxt(f, self.vertices, fmt="v %.8f %.8f %.8f")
            np.savetxt(f, np.asarray(self.faces, dtype=np.int64) + 1, fmt="f %d %d %d")
        return path


class Generator3D:
    """Generator3D.generate_mesh_fromdiffusion (convocc/src/conv_onet/generation.py:66-121, 146-207) for the branch the
    shipped configs take (upsampling_steps 0, vol_bound None, no refinement / simplification / normals): decode the latent
    to tri-planes, evaluate the fused plane sampling + MLP3D on the dense (1+padding) grid in one launch, and extract the
    iso-surface ON THE DEVICE (csrc/k_mcubes.hip) - the resolution0^3 logit grid is never copied to the host.
    Defaults follow get_generator (convocc/src/conv_onet/config.py:117-128) over convocc/configs/default.yaml."""

    def __init__(self, threshold: float = 0.5, resolution0: int = 256, upsampling_steps: int = 0, padding: float = 0.1,
                 refinement_step: int = 0, simplify_nfaces=None, with_normals: bool = False, **_ignored):
        if upsampling_steps != 0:
            raise NotImplementedError("MISE refinement (upsampling_steps > 0) is outside the path: the dense grid costs "
                                      "one fused launch here")
        if refinement_step or simplify_nfaces is not None or with_normals:
            raise NotImplementedError("mesh refinement / simplification / normals are outside the path")
        self.threshold, self.resolution0, self.padding = threshold, resolution0, padding

    def generate_mesh_fromdiffusion(self, latent, vaemodel, mlp, device=None):
        """-> (mesh, mesh2): mesh = TriMesh(vertices float64 [nv,3], faces int64 [nf,3]) on the host, mesh2 = the
        device-resident pair pytorch3d's Meshes is built from (verts float32 [1,nv,3], faces int64 [1,nf,3])."""
        return self.generate_mesh_fromplanes(vaemodel.decode(latent), mlp)

    def generate_mesh_fromplanes(self, planes, mlp):
        """The same from already decoded planes of ONE latent ((xy, yz, xz) lists of [1, C, H, W] tensors): the decoder may have run on
        a batch of latents (OccupancyPipeline.decode_batch)."""
        return self.mesh_from_grid(occupancy_grid_logits(mlp, planes, self.resolution0, self.padding))

    def mesh_from_grid(self, grid):
        """Marching cubes on the dense logit grid [1, n, n, n] of ONE latent (generation.py:146-183), on the current stream."""
        from .mcubes import extract_mesh
        assert grid.shape[0] == 1, "the reference extracts one mesh per call (tools/ldm/occupancy.py:216-217)"
        v, f = extract_mesh(grid[0], self.threshold, self.padding)
        mesh2 = {"verts": v.float().unsqueeze(0), "faces": f.unsqueeze(0)}
        return TriMesh(*self._to_host(v, f)), mesh2

    def _to_host(self, v, f):
        """The mesh's two arrays to numpy.  These are the `__amd_rocclr_copyBuffer` dispatches of the shapenet leg's profiles (two PCIe
        transfers of a few MB per 128^3 mesh, run by ROCclr's copy kernel on the side stream; 3.6 % of the leg's kernel time in round 4).
        Round 5 measured what they cost the pipeline: with the copies SKIPPED (`DDMI_DIAG_SKIP_MESH_D2H=1`, meshes come back empty) the leg
        runs at 122.3 shapes/s against 122.4 with them - they are hidden under the next shape's MLP3D launch.  Staging them through pinned
        buffers (tried) kept the copy kernel and made it 4x as many dispatches; the plain `.cpu()` stays."""
        import os
        import numpy as np
        if os.environ.get("DDMI_DIAG_SKIP_MESH_D2H"):   # timing-only diagnostics (see above)
            return np.zeros((0, 3)), np.zeros((0, 3), np.int64)
        return v.cpu().numpy(), f.cpu().numpy()
