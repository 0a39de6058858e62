"""CPU oracle for the DDMI generation path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A plain, functional, fp32 restatement (torch CPU ops + numpy) of the reference
algorithms on the hot path.  Only `tests/`, `__graft_entry__.smoke()` and
`bench.py`'s `cpu_baseline` leg may import this module; `ddmi_amd/` never does.

Pinning: every function here is checked against golden vectors captured by
importing the reference itself in the build container (`oracle/gen_golden.py`
-> `tests/golden/*.npz`, `tests/test_oracle_golden.py`).  The reference ships no
known-answer tests of its own, so those captured vectors are the pin.

All citations are relative to the reference tree (mlvlab/DDMI):
  ddpm      = diffusion/ddpm.py
  openai    = models/ldm/modules/diffusionmodules/openaimodel.py
  util      = models/ldm/modules/diffusionmodules/util.py
  ae        = models/d2c_vae/autoencoder_unet.py
  mlp       = models/d2c_vae/mlp.py
  blocks    = models/d2c_vae/blocks.py
  gu        = utils/general_utils.py
  fba       = models/d2c_vae/op/fused_bias_act_kernel.cu

Third-party arithmetic that the reference delegates to torch (conv2d, group_norm,
softmax, linear) is delegated to the same torch CPU ops here; everything the
reference itself spells out (schedules, mixing, posterior, attention layouts,
bilinear plane sampling, weight modulation) is restated explicitly.
"""
from __future__ import annotations

import math
import time
from typing import Dict, List, Mapping, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor
State = Mapping[str, Tensor]


def _t(x) -> Tensor:
    if isinstance(x, Tensor):
        return x.detach().to(torch.float32)
    return torch.from_numpy(np.ascontiguousarray(x)).to(torch.float32)


def to_state(d: Mapping[str, "np.ndarray"], prefix: str = "") -> Dict[str, Tensor]:
    """numpy/torch mapping -> fp32 torch mapping, optionally keeping only `prefix`ed keys (prefix stripped)."""
    out = {}
    for k, v in d.items():
        if prefix and not k.startswith(prefix):
            continue
        out[k[len(prefix):]] = _t(v)
    return out


# ======================================================================================
# a1/a2  schedules
# ======================================================================================

def linspace_f64(start: float, end: float, steps: int) -> np.ndarray:
    """torch.linspace(dtype=float64) as ATen computes it (symmetric around the midpoint):
    i < steps//2 : start + step*i ; else end - step*(steps-1-i)."""
    if steps == 1:
        return np.array([start], dtype=np.float64)
    step = (np.float64(end) - np.float64(start)) / np.float64(steps - 1)
    i = np.arange(steps, dtype=np.float64)
    lo = np.float64(start) + step * i
    hi = np.float64(end) - step * (np.float64(steps - 1) - i)
    return np.where(np.arange(steps) < steps // 2, lo, hi)


def linspace_f32(start: float, end: float, steps: int) -> np.ndarray:
    """Same formula evaluated in float32 (what `torch.linspace(-1, T-1, S+1)` does, ddpm:338)."""
    if steps == 1:
        return np.array([start], dtype=np.float32)
    s, e = np.float32(start), np.float32(end)
    step = np.float32((e - s) / np.float32(steps - 1))
    i = np.arange(steps, dtype=np.float32)
    lo = (s + step * i).astype(np.float32)
    hi = (e - step * (np.float32(steps - 1) - i)).astype(np.float32)
    return np.where(np.arange(steps) < steps // 2, lo, hi).astype(np.float32)


def make_schedule(timesteps: int = 1000, linear_start: float = 0.0015, linear_end: float = 0.0195,
                  v_posterior: float = 0.0) -> Dict[str, np.ndarray]:
    """ddpm:37-41 (linear betas) + ddpm:163-203 (register_schedule): fp64 math, fp32 buffers."""
    betas = linspace_f64(linear_start ** 0.5, linear_end ** 0.5, timesteps) ** 2
    alphas = 1.0 - betas
    ac = np.cumprod(alphas, axis=0)
    ac_prev = np.append(1.0, ac[:-1])
    post_var = (1 - v_posterior) * betas * (1.0 - ac_prev) / (1.0 - ac) + v_posterior * betas
    f = lambda a: np.asarray(a, dtype=np.float64).astype(np.float32)
    return {
        "betas": f(betas),
        "alphas_cumprod": f(ac),
        "alphas_cumprod_prev": f(ac_prev),
        "sqrt_alphas_cumprod": f(np.sqrt(ac)),
        "sqrt_one_minus_alphas_cumprod": f(np.sqrt(1.0 - ac)),
        "log_one_minus_alphas_cumprod": f(np.log(1.0 - ac)),
        "sqrt_recip_alphas_cumprod": f(np.sqrt(1.0 / ac)),
        "sqrt_recipm1_alphas_cumprod": f(np.sqrt(1.0 / ac - 1)),
        "posterior_variance": f(post_var),
        "posterior_log_variance_clipped": f(np.log(np.maximum(post_var, 1e-20))),
        "posterior_mean_coef1": f(betas * np.sqrt(ac_prev) / (1.0 - ac)),
        "posterior_mean_coef2": f((1.0 - ac_prev) * np.sqrt(alphas) / (1.0 - ac)),
    }


def ddim_time_pairs(timesteps: int, sampling_timesteps: int) -> List[Tuple[int, int]]:
    """ddpm:338-340: fp32 linspace(-1, T-1, S+1) -> trunc to int -> reversed -> consecutive pairs."""
    times = linspace_f32(-1.0, float(timesteps - 1), sampling_timesteps + 1)
    times = [int(v) for v in np.trunc(times).astype(np.int64)][::-1]
    return list(zip(times[:-1], times[1:]))


# ======================================================================================
# a3-a6  per-step arithmetic (all fp32, same operation order as the reference)
# ======================================================================================

def mixed_eps(sched, x: Tensor, model_out: Tensor, t: int, mixing_logit: Optional[Tensor]) -> Tensor:
    """ddpm:469-477: eps = (1-c)*sqrt(1-abar_t)*x + c*model_out, c = sigmoid(mixing_logit)."""
    if mixing_logit is None:
        return model_out
    mixing_component = float(sched["sqrt_one_minus_alphas_cumprod"][t]) * x
    coeff = torch.sigmoid(_t(mixing_logit))
    return (1 - coeff) * mixing_component + coeff * model_out


def predict_x0(sched, x: Tensor, t: int, eps: Tensor) -> Tensor:
    """ddpm:236-240."""
    return float(sched["sqrt_recip_alphas_cumprod"][t]) * x - float(sched["sqrt_recipm1_alphas_cumprod"][t]) * eps


def p_sample_step(sched, x: Tensor, model_out: Tensor, t: int, noise: Tensor,
                  mixing_logit: Optional[Tensor], clip_denoised: bool = False) -> Tensor:
    """One ancestral step given the UNet output: ddpm:257-281."""
    eps = mixed_eps(sched, x, model_out, t, mixing_logit)
    x0 = predict_x0(sched, x, t, eps)
    if clip_denoised:
        x0 = x0.clamp(-1.0, 1.0)
    mean = float(sched["posterior_mean_coef1"][t]) * x0 + float(sched["posterior_mean_coef2"][t]) * x
    logvar = torch.tensor(sched["posterior_log_variance_clipped"][t])
    nonzero = 0.0 if t == 0 else 1.0
    return mean + nonzero * (0.5 * logvar).exp() * noise


def ddim_coefs(sched, t: int, t_next: int, eta: float) -> Tuple[float, float, float]:
    """ddpm:357-361 evaluated with fp32 scalars exactly like the 0-dim tensors there.
    Returns (sqrt(alpha_next), c, sigma) as python floats holding fp32 values."""
    alpha = torch.tensor(sched["alphas_cumprod"][t])
    alpha_next = torch.tensor(sched["alphas_cumprod"][t_next])
    sigma = eta * ((1 - alpha / alpha_next) * (1 - alpha_next) / (1 - alpha)).sqrt()
    c = (1 - alpha_next - sigma ** 2).sqrt()
    return float(alpha_next.sqrt()), float(c), float(sigma)


def ddim_step(sched, x: Tensor, model_out: Tensor, t: int, t_next: int, eta: float, noise: Optional[Tensor],
              mixing_logit: Optional[Tensor], clip_denoised: bool = False) -> Tensor:
    """ddpm:303-332 (cond=None branch) + ddpm:353-368."""
    eps = mixed_eps(sched, x, model_out, t, mixing_logit)
    x0 = predict_x0(sched, x, t, eps)
    if clip_denoised:
        x0 = x0.clamp(-1.0, 1.0)
    if t_next < 0:
        return x0
    san, c, sigma = ddim_coefs(sched, t, t_next, eta)
    return x0 * san + c * eps + sigma * noise


def sample_chain(unet_fn, sched, x_T: Tensor, noises: Sequence[Tensor], timesteps: int, sampling_timesteps: int,
                 eta: float, mixing_logit: Optional[Tensor], clip_denoised: bool = False) -> Tensor:
    """ddpm:371-381 dispatch + ddpm:283-300 (p_sample_loop) / ddpm:334-369 (ddim_sample).
    `noises[i]` is the i-th `randn` the reference would draw after x_T: one per step for the
    ancestral loop (including t=0, ddpm:278) and one per non-final step for DDIM (ddpm:363)."""
    x = _t(x_T)
    B = x.shape[0]
    if sampling_timesteps < timesteps:
        for i, (t, t_next) in enumerate(ddim_time_pairs(timesteps, sampling_timesteps)):
            out = unet_fn(x, torch.full((B,), t, dtype=torch.long))
            nz = None if t_next < 0 else _t(noises[i])
            x = ddim_step(sched, x, out, t, t_next, eta, nz, mixing_logit, clip_denoised)
    else:
        for i, t in enumerate(reversed(range(timesteps))):
            out = unet_fn(x, torch.full((B,), t, dtype=torch.long))
            x = p_sample_step(sched, x, out, t, _t(noises[i]), mixing_logit, clip_denoised)
    return x


# ======================================================================================
# a8-a13  UNet
# ======================================================================================

def timestep_embedding(t: Tensor, dim: int, max_period: float = 10000.0) -> Tensor:
    """util:150-170: cat(cos, sin) of t * exp(-ln(max_period) * k / half)."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


def silu(x: Tensor) -> Tensor:
    return x * torch.sigmoid(x)


def group_norm(x: Tensor, w: Tensor, b: Tensor, eps: float, groups: int = 32) -> Tensor:
    """util:198-215 (eps 1e-5) and ae:46-47 (eps 1e-6); fp32 statistics over (C/groups, spatial)."""
    return F.group_norm(x.float(), groups, w, b, eps)


def group_norm_explicit(x: Tensor, w: Tensor, b: Tensor, eps: float, groups: int = 32) -> Tensor:
    """Spelled-out GroupNorm (biased variance), used to cross-check the torch op in the tests."""
    B, C = x.shape[:2]
    xg = x.reshape(B, groups, -1).double()
    mean = xg.mean(-1, keepdim=True)
    var = ((xg - mean) ** 2).mean(-1, keepdim=True)
    y = ((xg - mean) / torch.sqrt(var + eps)).reshape(x.shape).float()
    shape = (1, C) + (1,) * (x.dim() - 2)
    return y * w.reshape(shape) + b.reshape(shape)


def qkv_attention_legacy(qkv: Tensor, n_heads: int) -> Tensor:
    """openai:397-422. qkv [B, H*3*ch, T] with per-head interleaved [q;k;v] channels."""
    bs, width, length = qkv.shape
    ch = width // (3 * n_heads)
    q, k, v = qkv.reshape(bs * n_heads, ch * 3, length).split(ch, dim=1)
    scale = 1 / math.sqrt(math.sqrt(ch))
    w = torch.einsum("bct,bcs->bts", q * scale, k * scale)
    w = torch.softmax(w.float(), dim=-1)
    a = torch.einsum("bts,bcs->bct", w, v)
    return a.reshape(bs, -1, length)


def unet_resblock(p: State, pre: str, x: Tensor, emb: Tensor) -> Tensor:
    """openai:244-276 (use_scale_shift_norm=False, no up/down, dropout 0)."""
    h = F.conv2d(silu(group_norm(x, p[pre + "in_layers.0.weight"], p[pre + "in_layers.0.bias"], 1e-5)),
                 p[pre + "in_layers.2.weight"], p[pre + "in_layers.2.bias"], padding=1)
    e = F.linear(silu(emb), p[pre + "emb_layers.1.weight"], p[pre + "emb_layers.1.bias"])
    h = h + e[:, :, None, None]
    h = F.conv2d(silu(group_norm(h, p[pre + "out_layers.0.weight"], p[pre + "out_layers.0.bias"], 1e-5)),
                 p[pre + "out_layers.3.weight"], p[pre + "out_layers.3.bias"], padding=1)
    if (pre + "skip_connection.weight") in p:
        x = F.conv2d(x, p[pre + "skip_connection.weight"], p[pre + "skip_connection.bias"])
    return x + h


def unet_attnblock(p: State, pre: str, x: Tensor, num_head_channels: int) -> Tensor:
    """openai:319-325."""
    b, c = x.shape[:2]
    xf = x.reshape(b, c, -1)
    qkv = F.conv1d(group_norm(xf, p[pre + "norm.weight"], p[pre + "norm.bias"], 1e-5),
                   p[pre + "qkv.weight"], p[pre + "qkv.bias"])
    h = qkv_attention_legacy(qkv, c // num_head_channels)
    h = F.conv1d(h, p[pre + "proj_out.weight"], p[pre + "proj_out.bias"])
    return (xf + h).reshape(x.shape)


def unet_topology(cfg: Mapping) -> Tuple[List[List[Tuple[str, int, int]]], List[List[Tuple[str, int, int]]], int]:
    """Block lists that openai:567-743 constructs: for each input / output block a list of
    (kind, cin, cout) with kind in {conv_in, res, attn, down, up}.  Returns (inputs, outputs, mid_ch)."""
    mc = cfg["model_channels"]
    mult = list(cfg.get("channel_mult", (1, 2, 4, 8)))
    nrb = cfg["num_res_blocks"]
    attn_res = set(cfg["attention_resolutions"])
    inputs = [[("conv_in", cfg["in_channels"], mc)]]
    chans = [mc]
    ch, ds = mc, 1
    for level, m in enumerate(mult):
        for _ in range(nrb):
            layers = [("res", ch, m * mc)]
            ch = m * mc
            if ds in attn_res:
                layers.append(("attn", ch, ch))
            inputs.append(layers)
            chans.append(ch)
        if level != len(mult) - 1:
            inputs.append([("down", ch, ch)])
            chans.append(ch)
            ds *= 2
    mid_ch = ch
    outputs = []
    for level, m in list(enumerate(mult))[::-1]:
        for i in range(nrb + 1):
            ich = chans.pop()
            layers = [("res", ch + ich, mc * m)]
            ch = mc * m
            if ds in attn_res:
                layers.append(("attn", ch, ch))
            if level and i == nrb:
                layers.append(("up", ch, ch))
                ds //= 2
            outputs.append(layers)
    return inputs, outputs, mid_ch


def unet_forward(p: State, cfg: Mapping, x: Tensor, t: Tensor) -> Tensor:
    """openai:761-793 for the configurations the DDMI yaml files use: 2-D, AttentionBlock with
    QKVAttentionLegacy and num_head_channels, conv_resample=True, resblock_updown=False,
    no class / context conditioning."""
    x = _t(x)
    nhc = cfg["num_head_channels"]
    inputs, outputs, _ = unet_topology(cfg)
    emb = timestep_embedding(t, cfg["model_channels"])
    emb = F.linear(silu(F.linear(emb, p["time_embed.0.weight"], p["time_embed.0.bias"])),
                   p["time_embed.2.weight"], p["time_embed.2.bias"])

    def run(layers, pre, h):
        for j, (kind, cin, cout) in enumerate(layers):
            q = f"{pre}{j}."
            if kind == "conv_in":
                h = F.conv2d(h, p[q + "weight"], p[q + "bias"], padding=1)
            elif kind == "res":
                h = unet_resblock(p, q, h, emb)
            elif kind == "attn":
                h = unet_attnblock(p, q, h, nhc)
            elif kind == "down":  # openai:135-161
                h = F.conv2d(h, p[q + "op.weight"], p[q + "op.bias"], stride=2, padding=1)
            elif kind == "up":  # openai:110-120
                h = F.interpolate(h, scale_factor=2, mode="nearest")
                h = F.conv2d(h, p[q + "conv.weight"], p[q + "conv.bias"], padding=1)
        return h

    hs = []
    h = x
    for i, layers in enumerate(inputs):
        h = run(layers, f"input_blocks.{i}.", h)
        hs.append(h)
    h = run([("res", 0, 0), ("attn", 0, 0), ("res", 0, 0)], "middle_block.", h)
    for i, layers in enumerate(outputs):
        h = torch.cat([h, hs.pop()], dim=1)
        h = run(layers, f"output_blocks.{i}.", h)
    h = silu(group_norm(h, p["out.0.weight"], p["out.0.bias"], 1e-5))
    return F.conv2d(h, p["out.2.weight"], p["out.2.bias"], padding=1)


# ======================================================================================
# a15  D2C-VAE image decoder
# ======================================================================================

def vae_resnet_block(p: State, pre: str, x: Tensor) -> Tensor:
    """ae:129-149 with temb=None, dropout 0."""
    h = F.conv2d(silu(group_norm(x, p[pre + "norm1.weight"], p[pre + "norm1.bias"], 1e-6)),
                 p[pre + "conv1.weight"], p[pre + "conv1.bias"], padding=1)
    h = F.conv2d(silu(group_norm(h, p[pre + "norm2.weight"], p[pre + "norm2.bias"], 1e-6)),
                 p[pre + "conv2.weight"], p[pre + "conv2.bias"], padding=1)
    if (pre + "nin_shortcut.weight") in p:
        x = F.conv2d(x, p[pre + "nin_shortcut.weight"], p[pre + "nin_shortcut.bias"])
    return x + h


def vae_attn_block(p: State, pre: str, x: Tensor) -> Tensor:
    """ae:187-215 with num_heads=1 (make_attn 'vanilla', ae:561-567): softmax over keys of q.k / sqrt(C)."""
    b, c, hh, ww = x.shape
    hn = group_norm(x, p[pre + "norm.weight"], p[pre + "norm.bias"], 1e-6)
    q = F.conv2d(hn, p[pre + "q.weight"], p[pre + "q.bias"]).reshape(b, c, -1)
    k = F.conv2d(hn, p[pre + "k.weight"], p[pre + "k.bias"]).reshape(b, c, -1)
    v = F.conv2d(hn, p[pre + "v.weight"], p[pre + "v.bias"]).reshape(b, c, -1)
    w = torch.bmm(q.permute(0, 2, 1), k) * (int(c) ** (-0.5))  # [b, query, key]
    w = torch.softmax(w, dim=2)
    h = torch.bmm(v, w.permute(0, 2, 1)).reshape(b, c, hh, ww)  # h[c, query] = sum_key v[c,key] w[query,key]
    h = F.conv2d(h, p[pre + "proj_out.weight"], p[pre + "proj_out.bias"])
    return x + h


def vae_decode(p: State, ddconfig: Mapping, z: Tensor) -> List[Tensor]:
    """Autoencoder.decode ae:855-858 + Decoder.forward ae:790-829 (attn_resolutions empty or not,
    tanh_out False).  `p` holds the `Autoencoder` state dict (keys `post_quant_conv.*`, `decoder.*`).
    Returns the HDBF plane list, coarse -> fine."""
    z = _t(z)
    ch_mult = list(ddconfig["ch_mult"])
    nres = len(ch_mult)
    nrb = ddconfig["num_res_blocks"]
    attn_resolutions = set(ddconfig.get("attn_resolutions", []))
    hdbf_resolutions = set(ddconfig["hdbf_resolutions"])
    curr_res = ddconfig["resolution"] // 2 ** (nres - 1)
    d = "decoder."
    h = F.conv2d(z, p["post_quant_conv.weight"], p["post_quant_conv.bias"])
    h = F.conv2d(h, p[d + "conv_in.weight"], p[d + "conv_in.bias"], padding=1)
    h = vae_resnet_block(p, d + "mid.block_1.", h)
    h = vae_attn_block(p, d + "mid.attn_1.", h)
    h = vae_resnet_block(p, d + "mid.block_2.", h)
    planes = []
    for lvl in reversed(range(nres)):
        for ib in range(nrb + 1):
            h = vae_resnet_block(p, f"{d}up.{lvl}.block.{ib}.", h)
            if curr_res in attn_resolutions:
                h = vae_attn_block(p, f"{d}up.{lvl}.attn.{ib}.", h)
        if curr_res in hdbf_resolutions:
            planes.append(F.conv2d(h, p[f"{d}up.{lvl}.hdbf.0.weight"], p[f"{d}up.{lvl}.hdbf.0.bias"]))
        if lvl != 0:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = F.conv2d(h, p[f"{d}up.{lvl}.upsample.conv.weight"], p[f"{d}up.{lvl}.upsample.conv.bias"], padding=1)
            curr_res *= 2
    h = silu(group_norm(h, p[d + "norm_out.weight"], p[d + "norm_out.bias"], 1e-6))
    planes.append(F.conv2d(h, p[d + "conv_out.weight"], p[d + "conv_out.bias"], padding=1))
    return planes


# ======================================================================================
# a18-a20  coordinates, plane sampling, image coordinate-MLP
# ======================================================================================

def coord_grid_2d(h: int, w: int, hstart=-1.0, hend=1.0, wstart=-1.0, wend=1.0) -> Tensor:
    """gu:27-35 for b=1 (square grids): channel 0 = x (varies along W), channel 1 = y."""
    xs = torch.from_numpy(linspace_f32(wstart, wend, w)).view(1, 1, 1, -1).repeat(1, 1, w, 1)
    ys = torch.from_numpy(linspace_f32(hstart, hend, h)).view(1, 1, -1, 1).repeat(1, 1, 1, h)
    return torch.cat((xs, ys), dim=1)


def generate_coords(res: int) -> Tensor:
    """The grid `generate()` builds: tools/ldm/image.py:227-233."""
    e = (res - 1) / res
    return coord_grid_2d(res, res, -e, e, -e, e)


def scale_injection(res: int, anchor: int = 256) -> float:
    """gu:204-206."""
    return anchor / res


def bilinear_border(plane: Tensor, gx: Tensor, gy: Tensor, align_corners: bool) -> Tensor:
    """F.grid_sample(plane, grid, mode='bilinear', padding_mode='border') restated.
    plane [B,C,H,W]; gx, gy [B,N] normalised coords in [-1,1] -> [B,C,N]."""
    B, C, H, W = plane.shape

    def unnorm(g, size):
        if align_corners:
            return (g + 1) / 2 * (size - 1)
        return ((g + 1) * size - 1) / 2

    ix = unnorm(gx.float(), W).clamp(0, W - 1)
    iy = unnorm(gy.float(), H).clamp(0, H - 1)
    x0 = torch.floor(ix)
    y0 = torch.floor(iy)
    wx1 = ix - x0
    wy1 = iy - y0
    wx0 = 1 - wx1
    wy0 = 1 - wy1
    x0i = x0.long()
    y0i = y0.long()
    x1i = (x0i + 1).clamp(max=W - 1)
    y1i = (y0i + 1).clamp(max=H - 1)
    flat = plane.reshape(B, C, H * W)

    def gather(yy, xx):
        idx = (yy * W + xx)[:, None, :].expand(B, C, -1)
        return torch.gather(flat, 2, idx)

    return (gather(y0i, x0i) * (wy0 * wx0)[:, None] + gather(y0i, x1i) * (wy0 * wx1)[:, None]
            + gather(y1i, x0i) * (wy1 * wx0)[:, None] + gather(y1i, x1i) * (wy1 * wx1)[:, None])


def sinusoidal_pos_emb(x: Tensor, dim: int) -> Tensor:
    """blocks:11-23: cat(sin, cos), divisor half-1."""
    half = dim // 2
    e = math.log(10000) / (half - 1)
    e = torch.exp(torch.arange(half) * -e)
    e = x[:, None] * e[None, :]
    return torch.cat((e.sin(), e.cos()), dim=-1)


def mlp_style(p: State, si: float, ch: int = 256) -> Tensor:
    """mlp:46-47 + mlp:18-25 for one sample: style = Linear(GELU(Linear(sinusoid(si))))."""
    s = sinusoidal_pos_emb(torch.ones(1) * si, ch // 4)
    s = F.gelu(F.linear(s, p["time_mlp.1.weight"], p["time_mlp.1.bias"]))
    return F.linear(s, p["time_mlp.3.weight"], p["time_mlp.3.bias"])  # [1, ch]


def modulated_weight(p: State, pre: str, style: Tensor, demodulate: bool) -> Tensor:
    """blocks:242-250 for kernel_size 1 and one sample: returns W' [out, in]."""
    w = p[pre + "weight"][0, :, :, 0, 0]  # [out, in]
    out_ch, in_ch = w.shape
    mw, mb = p[pre + "modulation.weight"], p[pre + "modulation.bias"]
    s = F.linear(style, mw * (1 / math.sqrt(mw.shape[1])), mb)  # EqualLinear, lr_mul 1 (blocks:158-166)
    wm = (1 / math.sqrt(in_ch)) * w * s.reshape(1, in_ch)
    if demodulate:
        wm = wm * torch.rsqrt(wm.pow(2).sum(1, keepdim=True) + 1e-8)
    return wm


def fused_lrelu(x: Tensor, bias: Tensor) -> Tensor:
    """fba:26-47 with act=3 (leaky relu), grad=0, alpha 0.2, scale sqrt(2)."""
    return F.leaky_relu(x + bias[None, :], 0.2) * math.sqrt(2)


def styled_res_block(p: State, pre: str, x: Tensor, style: Tensor, noise: Optional[Sequence[Tensor]] = None) -> Tensor:
    """blocks:626-638 on per-point rows x [N, in]; noise[i] is the [N] NoiseInjection draw of conv i."""
    out = x
    for i, name in enumerate(("conv1.", "conv2.", "conv3.")):
        w = modulated_weight(p, pre + name + "conv.", style, True)
        out = out @ w.t()
        nw = p[pre + name + "noise.weight"]
        if noise is not None:
            out = out + nw * noise[i][:, None]
        out = fused_lrelu(out, p[pre + name + "activate.bias"])
    if (pre + "skip.0.weight") in p:
        ws = p[pre + "skip.0.weight"][:, :, 0, 0]
        skip = x @ (ws * (1 / math.sqrt(ws.shape[1]))).t()  # EqualConv2d, bias=False (blocks:122-130, 544-546)
    else:
        skip = x
    return (out + skip) / math.sqrt(2)


def image_mlp(p: State, coords: Tensor, planes: Sequence[Tensor], si: float,
              noise: Optional[Tensor] = None) -> Tensor:
    """MLP.forward mlp:34-66 restated per query point with pre-folded modulated weights.
    coords [1,2,R,R]; planes: 3 x [B,64,P,P]; noise optional [12,B,R*R] (the NoiseInjection draws in
    call order, only matters when `noise.weight` != 0).  Returns [B,3,R,R]."""
    coords = _t(coords)
    B = planes[0].shape[0]
    R_h, R_w = coords.shape[2:]
    N = R_h * R_w
    gx = coords[0, 0].reshape(1, N).expand(B, N)
    gy = coords[0, 1].reshape(1, N).expand(B, N)
    style = mlp_style(p, si)
    feats = []
    for pl in planes:
        f = bilinear_border(_t(pl), gx, gy, align_corners=False)  # gu:122-123
        f = f.permute(0, 2, 1).reshape(B * N, -1)
        feats.append(torch.cat((f, torch.full((B * N, 2), float(si))), dim=1))  # mlp:44,51
    nz = (lambda k: None) if noise is None else (lambda k: [_t(noise[3 * k + i]).reshape(B * N) for i in range(3)])
    x = styled_res_block(p, "net_res1.", feats[0], style, nz(0))
    x = styled_res_block(p, "net_res2.", torch.cat((x, feats[1]), dim=1), style, nz(1))
    x = styled_res_block(p, "net_res3.", torch.cat((x, feats[2]), dim=1), style, nz(2))
    x = styled_res_block(p, "net_res4.", x, style, nz(3))
    w = modulated_weight(p, "torgb.conv.", style, False)  # blocks:398 demodulate=False
    x = x @ w.t() + p["torgb.bias"].reshape(1, -1)
    return x.reshape(B, R_h, R_w, -1).permute(0, 3, 1, 2).contiguous()


# ======================================================================================
# generate(): tools/ldm/image.py:225-249
# ======================================================================================

def generate(unet_p: State, unet_cfg: Mapping, vae_p: State, ddconfig: Mapping, mlp_p: State, sched,
             mixing_logit, x_T, noises, timesteps: int, sampling_timesteps: int, eta: float, res: int) -> Dict[str, Tensor]:
    z = sample_chain(lambda x, t: unet_forward(unet_p, unet_cfg, x, t), sched, x_T, noises,
                     timesteps, sampling_timesteps, eta, mixing_logit)
    planes = vae_decode(vae_p, ddconfig, z)
    img = image_mlp(mlp_p, generate_coords(res), planes, scale_injection(res))
    return {"z": z, "planes": planes, "img": img.clamp(-1.0, 1.0)}


# ======================================================================================
# a16, a21, a23  occupancy path: tri-plane decoder, tri-plane plane sampling, MLP3D, query grid
# ======================================================================================

def make_3d_grid(bb_min, bb_max, shape) -> Tensor:
    """convocc/src/common.py:145-164: x slowest, z fastest; fp32 linspace per axis."""
    size = shape[0] * shape[1] * shape[2]
    pxs = torch.from_numpy(linspace_f32(bb_min[0], bb_max[0], shape[0]))
    pys = torch.from_numpy(linspace_f32(bb_min[1], bb_max[1], shape[1]))
    pzs = torch.from_numpy(linspace_f32(bb_min[2], bb_max[2], shape[2]))
    pxs = pxs.view(-1, 1, 1).expand(*shape).contiguous().view(size)
    pys = pys.view(1, -1, 1).expand(*shape).contiguous().view(size)
    pzs = pzs.view(1, 1, -1).expand(*shape).contiguous().view(size)
    return torch.stack([pxs, pys, pzs], dim=1)


def occupancy_query_points(nx: int, padding: float = 0.1) -> Tensor:
    """convocc/src/conv_onet/generation.py:87-94: (1 + padding) * make_3d_grid((-.5,)*3, (.5,)*3, (nx,)*3)."""
    return (1 + padding) * make_3d_grid((-0.5,) * 3, (0.5,) * 3, (nx,) * 3)


_PLANE_AXES = {"xz": (0, 2), "xy": (0, 1), "yz": (1, 2)}


def normalize_coordinate(p: Tensor, plane: str, padding: float = 0.1) -> Tensor:
    """gu:71-94: u = p[axes] / (1 + padding + 10e-6) + 0.5; u >= 1 -> 1 - 10e-6; u < 0 -> 0.  p [B,N,3] -> [B,N,2]."""
    a0, a1 = _PLANE_AXES[plane]
    xy = torch.stack((p[..., a0], p[..., a1]), dim=-1)
    xy_new = xy / (1 + padding + 10e-6) + 0.5
    xy_new = torch.where(xy_new >= 1, torch.full_like(xy_new, 1 - 10e-6), xy_new)
    xy_new = torch.where(xy_new < 0, torch.zeros_like(xy_new), xy_new)
    return xy_new


def triplane_features_add(planes_xy: Tensor, planes_yz: Tensor, planes_xz: Tensor, p: Tensor) -> Tensor:
    """gu:115-119 (sample_plane_feature) + gu:126-131 (mode 'add'): sum over the three planes of bilinear samples
    (align_corners=True, border) at grid 2u-1.  planes [B,C,S,S], p [B,N,3] -> [B,N,C]."""
    out = None
    for pl, name in ((planes_xy, "xy"), (planes_yz, "yz"), (planes_xz, "xz")):
        u = normalize_coordinate(p.clone(), name)
        g = 2.0 * u.float() - 1.0
        f = bilinear_border(_t(pl), g[..., 0], g[..., 1], align_corners=True)  # [B,C,N]
        out = f if out is None else out + f
    return out.transpose(1, 2)


def resnet_block_fc(p: State, pre: str, x: Tensor) -> Tensor:
    """blocks:707-716: x_s + fc_1(relu(fc_0(relu(x)))), bias-free shortcut when size_in != size_out."""
    net = F.linear(F.relu(x), p[pre + "fc_0.weight"], p[pre + "fc_0.bias"])
    dx = F.linear(F.relu(net), p[pre + "fc_1.weight"], p[pre + "fc_1.bias"])
    xs = F.linear(x, p[pre + "shortcut.weight"]) if (pre + "shortcut.weight") in p else x
    return xs + dx


def mlp3d(p: State, coords: Tensor, hdbf: Sequence[Sequence[Tensor]]) -> Tensor:
    """MLP3D.forward mlp.py:82-111.  coords [B,N,3]; hdbf = (xy, yz, xz), each a list of 3 planes coarse -> fine.
    Returns the Bernoulli logits [B,N]."""
    coords = _t(coords)
    xy, yz, xz = hdbf
    feats = [triplane_features_add(xy[i], yz[i], xz[i], coords) for i in range(3)]
    x = F.linear(coords, p["net_p.weight"], p["net_p.bias"]) + resnet_block_fc(p, "net_res1.", feats[0])
    x = resnet_block_fc(p, "net_res2.", torch.cat((x, feats[1]), dim=-1))
    x = resnet_block_fc(p, "net_res3.", torch.cat((x, feats[2]), dim=-1))
    x = resnet_block_fc(p, "net_res4.", x)
    return F.linear(x, p["net_out.weight"], p["net_out.bias"]).squeeze(-1)


def vae3d_decode(p: State, ddconfig: Mapping, embed_dim: int, z: Tensor):
    """Autoencoder3D.decode ae:907-916 + Decoder_triplane.forward ae:1461-1564.  z [B, 3*embed_dim, S, S] with channel
    blocks (xy | xz | yz); per-plane post_quant convs; returns (xy, yz, xz), each a list of planes coarse -> fine."""
    z = _t(z)
    e = embed_dim
    xy, xz, yz = z[:, :e], z[:, e:2 * e], z[:, 2 * e:]
    hs = [F.conv2d(xy, p["post_quant_conv_xy.weight"], p["post_quant_conv_xy.bias"]),
          F.conv2d(yz, p["post_quant_conv_yz.weight"], p["post_quant_conv_yz.bias"]),
          F.conv2d(xz, p["post_quant_conv_xz.weight"], p["post_quant_conv_xz.bias"])]  # decoder((xy, yz, xz))
    d = "decoder."
    ch_mult = list(ddconfig["ch_mult"])
    nres, nrb = len(ch_mult), ddconfig["num_res_blocks"]
    attn_res = set(ddconfig.get("attn_resolutions", []))
    hdbf_res = set(ddconfig["hdbf_resolutions"])
    inter_res = set(ddconfig["inter_attn_resolutions"])
    curr = ddconfig["resolution"] // 2 ** (nres - 1)

    def per_plane(fn):
        return [fn(h) for h in hs]

    def inter(hs, names):
        ch = hs[0].shape[1]
        h = torch.cat(hs, dim=1)
        h = vae_resnet_block(p, names[0], h)
        h = vae_attn_block(p, names[1], h)
        h = vae_resnet_block(p, names[2], h)
        return [h[:, :ch], h[:, ch:2 * ch], h[:, 2 * ch:]]

    hs = per_plane(lambda h: F.conv2d(h, p[d + "conv_in.weight"], p[d + "conv_in.bias"], padding=1))
    hs = per_plane(lambda h: vae_resnet_block(p, d + "mid.block_2.",
                                              vae_attn_block(p, d + "mid.attn_1.", vae_resnet_block(p, d + "mid.block_1.", h))))
    hs = inter(hs, (d + "mid.block_3.", d + "mid_attn.", d + "mid.block_4."))
    outs = [[], [], []]
    for lvl in reversed(range(nres)):
        for ib in range(nrb + 1):
            hs = per_plane(lambda h: vae_resnet_block(p, f"{d}up.{lvl}.block.{ib}.", h))
            if curr in attn_res:
                hs = per_plane(lambda h: vae_attn_block(p, f"{d}up.{lvl}.attn.{ib}.", h))
        if curr in inter_res:
            hs = inter(hs, (f"{d}up.{lvl}.inter_attn.0.", f"{d}up.{lvl}.inter_attn.1.", f"{d}up.{lvl}.inter_attn.2."))
        if curr in hdbf_res:
            for i, h in enumerate(hs):
                outs[i].append(F.conv2d(h, p[f"{d}up.{lvl}.hdbf.0.weight"], p[f"{d}up.{lvl}.hdbf.0.bias"]))
        if lvl != 0:
            hs = per_plane(lambda h: F.conv2d(F.interpolate(h, scale_factor=2.0, mode="nearest"),
                                              p[f"{d}up.{lvl}.upsample.conv.weight"], p[f"{d}up.{lvl}.upsample.conv.bias"], padding=1))
            curr *= 2
    for i, h in enumerate(hs):
        h = silu(group_norm(h, p[d + "norm_out.weight"], p[d + "norm_out.bias"], 1e-6))
        outs[i].append(F.conv2d(h, p[d + "conv_out.weight"], p[d + "conv_out.bias"], padding=1))
    return tuple(outs)  # (xy, yz, xz)


# ======================================================================================
# a14, a17, a22  video path: tri-plane UNet, VideoDecoder_light, MLPVideo
#
# Third-party arithmetic absent from the reference tree: `xformers.ops.memory_efficient_attention`
# (unpinned, README.md:33).  Its published semantics for 3-D inputs [batch*heads, tokens, dim] with
# attn_bias=None are softmax(q k^T / sqrt(dim)) v, restated in `sdpa_heads` below.  The reference needs
# xformers for the video configuration anyway ('vanilla-1d-multihead-expand' raises without it, ae:596-600),
# so these semantics - not the C^-1/2 scale of the no-xformers fallback AttnBlock1d (ae:259) - are canonical.
# ======================================================================================

SDPA_QUERY_CHUNK = 512   # queries per block once the score matrix of a head would pass SDPA_FULL_SCORES elements
SDPA_FULL_SCORES = 1 << 26


# TIMING ONLY (bench.py's bounded cpu_baseline for config 5): when "max_chunks" is set, a query-chunked evaluation stops after that many
# chunks (the remaining output rows are zeros - the RESULT IS THEN WRONG) and the time spent in / the number of chunks of every such
# loop are accumulated here, so that the caller can extrapolate the attention it sampled: rows are independent softmaxes, the cost is
# linear in the chunks evaluated.  Never set by the tests.
SDPA_SAMPLE = {"max_chunks": None, "seconds": 0.0, "done": 0, "total": 0}


def sdpa_heads(q: Tensor, k: Tensor, v: Tensor, heads: int, q_chunk: Optional[int] = None) -> Tensor:
    """q, k, v [B, heads*d, N] with channel = head*d + j (ae:447-456): per head softmax(q k^T / sqrt(d)) v -> [B, heads*d, N].
    Every query row is its own softmax, so the rows are evaluated in blocks of `q_chunk` queries against ALL keys when the
    full [N, N] matrix would not fit (the video decoder's last inter-plane attention has N = 73 728: 21.7 GB per head in fp32,
    151 MB per head and block of 512) - the same function, only the order of evaluation differs."""
    B, HD, N = q.shape
    d = HD // heads
    qh = q.reshape(B, heads, d, N).permute(0, 1, 3, 2)  # [B,H,N,d]
    kh = k.reshape(B, heads, d, N)
    vh = v.reshape(B, heads, d, N).permute(0, 1, 3, 2)
    if q_chunk is None:
        q_chunk = N if N * N <= SDPA_FULL_SCORES else SDPA_QUERY_CHUNK
    outs = []
    limit = SDPA_SAMPLE["max_chunks"] if q_chunk < N else None
    t_loop = time.perf_counter() if limit else 0.0
    for i, a in enumerate(range(0, N, q_chunk)):
        if limit and i >= limit:   # timing sample: the rest of the rows are not evaluated
            outs.append(torch.zeros((B, heads, N - a, d), dtype=q.dtype))
            break
        w = torch.softmax(torch.matmul(qh[:, :, a:a + q_chunk], kh) * (d ** -0.5), dim=-1)  # [B,H,q,Nk]
        outs.append(torch.matmul(w, vh))  # [B,H,q,d]
    if limit:
        total = -(-N // q_chunk)
        SDPA_SAMPLE["seconds"] += time.perf_counter() - t_loop
        SDPA_SAMPLE["done"] += min(limit, total)
        SDPA_SAMPLE["total"] += total
    o = outs[0] if len(outs) == 1 else torch.cat(outs, dim=2)
    return o.permute(0, 1, 3, 2).reshape(B, HD, N)


def attn_block_1d(p: State, pre: str, x: Tensor, heads: int, expand: bool) -> Tensor:
    """MemoryEfficientAttnBlock1D (ae:404-468, heads of C/heads channels) and ..._expand (ae:470-534, q/k/v project
    C -> heads*C, head dim C) over a token sequence x [B,C,N]; GroupNorm eps 1e-6 (ae:46-47)."""
    hn = group_norm(x, p[pre + "norm.weight"], p[pre + "norm.bias"], 1e-6)
    q = F.conv1d(hn, p[pre + "q.weight"], p[pre + "q.bias"])
    k = F.conv1d(hn, p[pre + "k.weight"], p[pre + "k.bias"])
    v = F.conv1d(hn, p[pre + "v.weight"], p[pre + "v.bias"])
    assert q.shape[1] == (x.shape[1] * heads if expand else x.shape[1])
    o = sdpa_heads(q, k, v, heads)
    return x + F.conv1d(o, p[pre + "proj_out.weight"], p[pre + "proj_out.bias"])


def cat_planes(h_xy: Tensor, h_xt: Tensor, h_yt: Tensor) -> Tensor:
    """openai:1148-1152: token order xy | xt | yt."""
    b, c = h_xy.shape[:2]
    return torch.cat([h_xy.reshape(b, c, -1), h_xt.reshape(b, c, -1), h_yt.reshape(b, c, -1)], dim=-1)


def split_planes(h: Tensor, res_h: int, res_w: int, t: int) -> Tuple[Tensor, Tensor, Tensor]:
    """openai:1155-1157 / ae:1073-1075: xy [res_h,res_w], xt [t,res_w], yt [t,res_w] views of the token axis."""
    b, c = h.shape[:2]
    n_xy, n_t = res_h * res_w, t * res_w
    return (h[:, :, :n_xy].reshape(b, c, res_h, res_w), h[:, :, n_xy:n_xy + n_t].reshape(b, c, t, res_w),
            h[:, :, n_xy + n_t:n_xy + 2 * n_t].reshape(b, c, t, res_w))


def unet_triplane_forward(p: State, cfg: Mapping, x: Tensor, t: Tensor) -> Tensor:
    """UNetModel_Triplane.forward openai:1113-1221: the UNetModel blocks applied to the three planes with shared
    weights, and a 16-head 1-D attention over the concatenated tokens after every block
    (make_attn('vanilla-1d-multihead'), openai:955-957,1023,1082).  x [B, C, s1*s2 + s1*s3 + s2*s3]."""
    x = _t(x)
    nhc = cfg["num_head_channels"]
    s1, s2, s3 = cfg["size1"], cfg["size2"], cfg["size3"]
    inputs, outputs, _ = unet_topology(cfg)
    emb = timestep_embedding(t, cfg["model_channels"])
    emb = F.linear(silu(F.linear(emb, p["time_embed.0.weight"], p["time_embed.0.bias"])),
                   p["time_embed.2.weight"], p["time_embed.2.bias"])

    def run(layers, pre, h):
        for j, (kind, cin, cout) in enumerate(layers):
            q = f"{pre}{j}."
            if kind == "conv_in":
                h = F.conv2d(h, p[q + "weight"], p[q + "bias"], padding=1)
            elif kind == "res":
                h = unet_resblock(p, q, h, emb)
            elif kind == "attn":
                h = unet_attnblock(p, q, h, nhc)
            elif kind == "down":
                h = F.conv2d(h, p[q + "op.weight"], p[q + "op.bias"], stride=2, padding=1)
            elif kind == "up":
                h = F.interpolate(h, scale_factor=2, mode="nearest")
                h = F.conv2d(h, p[q + "conv.weight"], p[q + "conv.bias"], padding=1)
        return h

    def inter(pre, hs):
        if (pre + "norm.weight") not in p:  # input_attns[0] is nn.Identity (openai:912)
            return hs
        h_xy, h_xt, h_yt = hs
        res_h, res_w, tt = h_xy.shape[-2], h_xy.shape[-1], h_xt.shape[-2]
        return split_planes(attn_block_1d(p, pre, cat_planes(h_xy, h_xt, h_yt), 16, False), res_h, res_w, tt)

    n_xy = s1 * s2
    hs = (x[:, :, :n_xy].reshape(x.shape[0], x.shape[1], s1, s2),
          x[:, :, n_xy:n_xy + s1 * s3].reshape(x.shape[0], x.shape[1], s3, s1),
          x[:, :, n_xy + s1 * s3:n_xy + s1 * s3 + s2 * s3].reshape(x.shape[0], x.shape[1], s3, s2))
    skips = []
    for i, layers in enumerate(inputs):
        hs = tuple(run(layers, f"input_blocks.{i}.", h) for h in hs)
        hs = inter(f"input_attns.{i}.", hs)
        skips.append(hs)
    hs = tuple(run([("res", 0, 0), ("attn", 0, 0), ("res", 0, 0)], "middle_block.", h) for h in hs)
    hs = inter("mid_attn.", hs)
    for i, layers in enumerate(outputs):
        sk = skips.pop()
        hs = tuple(run(layers, f"output_blocks.{i}.", torch.cat([h, s], dim=1)) for h, s in zip(hs, sk))
        hs = inter(f"output_attns.{i}.", hs)
    outs = [F.conv2d(silu(group_norm(h, p["out.0.weight"], p["out.0.bias"], 1e-5)), p["out.2.weight"], p["out.2.bias"],
                     padding=1) for h in hs]
    return cat_planes(*outs)


def vae_attn_block_mh(p: State, pre: str, x: Tensor, heads: int) -> Tensor:
    """MemoryEfficientAttnBlock (ae:273-335) on one plane: `heads` heads of C/heads channels (make_attn
    'vanilla-multihead' -> 16 heads, ae:568-573)."""
    b, c, hh, ww = x.shape
    hn = group_norm(x, p[pre + "norm.weight"], p[pre + "norm.bias"], 1e-6)
    q = F.conv2d(hn, p[pre + "q.weight"], p[pre + "q.bias"]).reshape(b, c, -1)
    k = F.conv2d(hn, p[pre + "k.weight"], p[pre + "k.bias"]).reshape(b, c, -1)
    v = F.conv2d(hn, p[pre + "v.weight"], p[pre + "v.bias"]).reshape(b, c, -1)
    o = sdpa_heads(q, k, v, heads).reshape(b, c, hh, ww)
    return x + F.conv2d(o, p[pre + "proj_out.weight"], p[pre + "proj_out.bias"])


_ATTN_HEADS = {"vanilla": 1, "vanilla-multihead": 16}


def video_decoder(p: State, ddconfig: Mapping, z_xy: Tensor, z_yt: Tensor, z_xt: Tensor, pre: str = "decoder."):
    """VideoDecoder_light.forward ae:1040-1144.  Returns (xy, yt, xt), each the list of HDBF planes coarse -> fine;
    xy planes [res,res], yt / xt planes [t,res] (t is never upsampled, ae:1115-1118)."""
    ch_mult = list(ddconfig["ch_mult"])
    nres, nrb = len(ch_mult), ddconfig["num_res_blocks"]
    attn_res = set(ddconfig.get("attn_resolutions", []))
    hdbf_res = set(ddconfig["hdbf_resolutions"])
    inter_res = set(ddconfig["inter_attn_resolutions"])
    heads = _ATTN_HEADS[ddconfig.get("attn_type", "vanilla")]
    curr = ddconfig["resolution"] // 2 ** (nres - 1)
    d = pre
    hs = [z_xy, z_yt, z_xt]  # the reference's processing order (ae:1051-1069)
    hs = [F.conv2d(h, p[d + "conv_in.weight"], p[d + "conv_in.bias"], padding=1) for h in hs]
    hs = [vae_resnet_block(p, d + "mid.block_2.",
                           vae_attn_block_mh(p, d + "mid.attn_1.", vae_resnet_block(p, d + "mid.block_1.", h), heads))
          for h in hs]

    def inter(name, hs):
        h_xy, h_yt, h_xt = hs
        res_h, res_w, tt = h_xy.shape[-2], h_xy.shape[-1], h_xt.shape[-2]
        o_xy, o_xt, o_yt = split_planes(attn_block_1d(p, name, cat_planes(h_xy, h_xt, h_yt), 8, True), res_h, res_w, tt)
        return [o_xy, o_yt, o_xt]

    hs = inter(d + "mid_attn.", hs)
    out = ([], [], [])
    for lvl in reversed(range(nres)):
        for ib in range(nrb + 1):
            hs = [vae_resnet_block(p, f"{d}up.{lvl}.block.{ib}.", h) for h in hs]
            if curr in attn_res:
                hs = [vae_attn_block_mh(p, f"{d}up.{lvl}.attn.{ib}.", h, heads) for h in hs]
        if curr in inter_res:
            hs = inter(f"{d}up.{lvl}.inter_attn.0.", hs)
        if curr in hdbf_res:
            for o, h in zip(out, hs):
                o.append(F.conv2d(h, p[f"{d}up.{lvl}.hdbf.0.weight"], p[f"{d}up.{lvl}.hdbf.0.bias"]))
        if lvl != 0:
            w_, b_ = p[f"{d}up.{lvl}.upsample.conv.weight"], p[f"{d}up.{lvl}.upsample.conv.bias"]
            hs = [F.conv2d(F.interpolate(hs[0], scale_factor=2.0, mode="nearest"), w_, b_, padding=1),
                  F.conv2d(F.interpolate(hs[1], scale_factor=(1.0, 2.0), mode="nearest"), w_, b_, padding=1),
                  F.conv2d(F.interpolate(hs[2], scale_factor=(1.0, 2.0), mode="nearest"), w_, b_, padding=1)]
            curr *= 2
    for o, h in zip(out, hs):
        h = silu(group_norm(h, p[d + "norm_out.weight"], p[d + "norm_out.bias"], 1e-6))
        o.append(F.conv2d(h, p[d + "conv_out.weight"], p[d + "conv_out.bias"], padding=1))
    return out


def vit_decode(p: State, ddconfig: Mapping, frames: int, z: Tensor):
    """VITAutoencoder.decode autoencoder_vit.py:188-202: z [B, C, s*s + 2*s*frames] with s = resolution / 8;
    xy = [s,s], xt = [frames,s], yt = [frames,s]; per-plane 1x1 post convs; decoder((xy, yt, xt))."""
    z = _t(z)
    s = ddconfig["resolution"] // 8
    xy, xt, yt = split_planes(z, s, s, frames)
    xy = F.conv2d(xy, p["post_xy.weight"], p["post_xy.bias"])
    yt = F.conv2d(yt, p["post_yt.weight"], p["post_yt.bias"])
    xt = F.conv2d(xt, p["post_xt.weight"], p["post_xt.bias"])
    return video_decoder(p, ddconfig, xy, yt, xt)


def coord_grid_3d(h: int, w: int, t: int, hstart=-1.0, hend=1.0, wstart=-1.0, wend=1.0, tstart=-1.0, tend=1.0):
    """gu:38-52 for b=1: 'xy' = (x along W, y along H) on [h,w]; 'xt' = (t, x) on [t,w]; 'yt' = (t, y) on [t,h]."""
    xs = torch.from_numpy(linspace_f32(wstart, wend, w))
    ys = torch.from_numpy(linspace_f32(hstart, hend, h))
    ts = torch.from_numpy(linspace_f32(tstart, tend, t))
    return {"xy": torch.stack((xs.view(1, w).expand(h, w), ys.view(h, 1).expand(h, w)))[None],
            "xt": torch.stack((ts.view(t, 1).expand(t, w), xs.view(1, w).expand(t, w)))[None],
            "yt": torch.stack((ts.view(t, 1).expand(t, h), ys.view(1, h).expand(t, h)))[None]}


def video_coords(res: int, frames: int = 16):
    """tools/ldm/video.py:216-217 as written there: hstart/hend/wstart = -+(res-1)/res, wend = 255/res, t = -+15/16."""
    e = (res - 1) / res
    return coord_grid_3d(res, res, frames, -e, e, -e, 255 / res, -15 / 16, 15 / 16)


def triplane_features_concat(xy: Tensor, yt: Tensor, xt: Tensor, coords: Mapping[str, Tensor]) -> Tensor:
    """gu:134-145 (mode 'concat'): three 2-D bilinear samplings (align_corners=True, border), broadcast to (t,h,w),
    channel-concatenated; rows ordered (b,t,h,w).  Note the grid's channel 0 is fed to grid_sample's x (plane width)
    axis, exactly as the reference does.  -> [B*t*h*w, 3C]."""
    b = xy.shape[0]

    def sample(plane, c):
        _, _, gh, gw = c.shape
        g = c.expand(b, -1, -1, -1).reshape(b, 2, gh * gw)
        return bilinear_border(_t(plane), g[:, 0], g[:, 1], align_corners=True).reshape(b, plane.shape[1], gh, gw)

    x1 = sample(xy, coords["xy"])  # [b,c,h,w]
    x2 = sample(yt, coords["yt"])  # [b,c,t,h]
    x3 = sample(xt, coords["xt"])  # [b,c,t,w]
    c, h, w = x1.shape[1:]
    t = x2.shape[2]
    x1 = x1[:, :, None].expand(b, c, t, h, w)
    x2 = x2[:, :, :, :, None].expand(b, c, t, h, w)
    x3 = x3[:, :, :, None, :].expand(b, c, t, h, w)
    x = torch.cat((x1, x2, x3), dim=1).reshape(b, 3 * c, -1)
    return x.permute(0, 2, 1).reshape(-1, 3 * c)


def mlp_video_rows(p: State, coords: Mapping[str, Tensor], hdbf) -> Tensor:
    """The per-point part of MLPVideo.forward (mlp.py:128-155) -> [B * t * h * w (of `coords`), out_ch]; `coords` may cover fewer frames
    than the planes do (bench.py times two frames' worth of points)."""
    xy, yt, xt = hdbf
    f = [triplane_features_concat(xy[i], yt[i], xt[i], coords) for i in range(3)]
    x = resnet_block_fc(p, "net_res1.", f[0])
    x = resnet_block_fc(p, "net_res2.", torch.cat((x, f[1]), dim=1))
    x = resnet_block_fc(p, "net_res3.", torch.cat((x, f[2]), dim=1))
    x = resnet_block_fc(p, "net_res4.", x)
    return F.linear(F.leaky_relu(x, 0.2), p["net_out.weight"], p["net_out.bias"])


def mlp_video(p: State, coords: Mapping[str, Tensor], hdbf) -> Tensor:
    """MLPVideo.forward mlp.py:128-157.  hdbf = (xy, yt, xt), each 3 planes coarse -> fine -> [B, out_ch, t, h, w]."""
    xy, yt, xt = hdbf
    b, _, h, w = xy[-1].shape
    t = yt[-1].shape[2]
    x = mlp_video_rows(p, coords, hdbf)
    oc = x.shape[-1]
    return x.reshape(b, -1, oc).permute(0, 2, 1).reshape(b, oc, t, h, w)


def generate_video(unet_p: State, unet_cfg: Mapping, vae_p: State, ddconfig: Mapping, frames: int, mlp_p: State, sched,
                   mixing_logit, x_T, noises, timesteps: int, sampling_timesteps: int, eta: float, coords) -> Dict[str, Tensor]:
    """tools/ldm/video.py:214-230: sample -> VITAutoencoder.decode -> MLPVideo -> clamp."""
    z = sample_chain(lambda x, t: unet_triplane_forward(unet_p, unet_cfg, x, t), sched, x_T, noises,
                     timesteps, sampling_timesteps, eta, mixing_logit)
    hdbf = vit_decode(vae_p, ddconfig, frames, z)
    vid = mlp_video(mlp_p, coords, hdbf)
    return {"z": z, "hdbf": hdbf, "video": vid.clamp(-1.0, 1.0)}


# ======================================================================================
# f4  training loss: DDPM.p_losses (ddpm:383-386 q_sample, :409-447) for the configuration every DDMI yaml uses:
#     parameterization "eps", loss_type "l2", l_simple_weight 1, original_elbo_weight 0, mixed prediction.
#     Differentiable: torch autograd of this function w.r.t. the parameter tensors IS the reference's backward pass
#     (pinned against fixture T1, the real reference's loss and gradients, in tests/test_oracle_golden.py).
# ======================================================================================

def p_losses(unet_p: State, unet_cfg: Mapping, sched, mixing_logit: Optional[Tensor], x_start: Tensor, t: Tensor, noise: Tensor,
             model=None) -> Tensor:
    """loss = mean((noise - mixed(model(q_sample(x_start, t, noise), t)))^2); `unet_p` / `mixing_logit` may require grad.
    `model(p, cfg, x, t)` defaults to unet_forward (the tri-plane UNet takes unet_triplane_forward)."""
    x_start, noise = _t(x_start), _t(noise)
    shape = (-1,) + (1,) * (x_start.ndim - 1)
    sa = torch.from_numpy(np.asarray(sched["sqrt_alphas_cumprod"], dtype=np.float32))[t].reshape(shape)
    s1 = torch.from_numpy(np.asarray(sched["sqrt_one_minus_alphas_cumprod"], dtype=np.float32))[t].reshape(shape)
    x_noisy = sa * x_start + s1 * noise                                              # ddpm:383-386
    out = (model or unet_forward)(unet_p, unet_cfg, x_noisy, t)
    if mixing_logit is not None:                                                     # ddpm:469-477 with a per-sample t
        coeff = torch.sigmoid(mixing_logit)
        out = (1 - coeff) * (s1 * x_noisy) + coeff * out
    per_sample = ((noise - out) ** 2).mean(dim=tuple(range(1, x_start.ndim)))       # mse_loss(reduction='none').mean(dim=[1,2,3])
    return per_sample.mean()                                                         # l_simple_weight 1, original_elbo_weight 0


# ======================================================================================
# f4  EMA of the trainer: ema_pytorch.EMA (third-party, `pip install ema-pytorch`, README.md:33, unpinned and absent from
# the reference tree).  Restated from its published algorithm; the reference's call sites are tools/ldm/image.py:64
# (EMA(diffusion_process, beta=ema_decay, update_every=ema_update_every): every other argument at its default) and :174
# (ema.update() after each optimizer step); its state dict ('initted', 'step', 'online_model.*', 'ema_model.*') is what
# tools/ldm/image.py:115 saves and :80/:130 load.  No golden vector exists for it (the package cannot be imported here):
# parity for this function is "unpinned" and anchored on those call sites only.
# ======================================================================================

class EmaPytorchOracle:
    """ema_pytorch.EMA on a flat list of tensors: update_after_step = 100, inv_gamma = 1, power = 2/3, min_value = 0."""

    def __init__(self, online: Sequence[Tensor], beta: float = 0.9999, update_every: int = 10, update_after_step: int = 100,
                 inv_gamma: float = 1.0, power: float = 2.0 / 3.0, min_value: float = 0.0):
        self.online = list(online)
        self.ema = [t.detach().clone() for t in self.online]     # deepcopy(model) at construction
        self.beta, self.update_every, self.update_after_step = beta, update_every, update_after_step
        self.inv_gamma, self.power, self.min_value = inv_gamma, power, min_value
        self.step = torch.tensor(0)
        self.initted = torch.tensor(False)

    def get_current_decay(self) -> float:
        epoch = (self.step - self.update_after_step - 1).clamp(min=0.0)
        value = 1 - (1 + epoch / self.inv_gamma) ** -self.power
        if epoch.item() <= 0:
            return 0.0
        return value.clamp(min=self.min_value, max=self.beta).item()

    def copy_params_from_model_to_ema(self):
        for e, o in zip(self.ema, self.online):
            e.copy_(o)

    def update(self):
        step = self.step.item()
        self.step += 1
        if (step % self.update_every) != 0:
            return
        if step <= self.update_after_step:
            self.copy_params_from_model_to_ema()
            return
        if not self.initted.item():
            self.copy_params_from_model_to_ema()
            self.initted.data.copy_(torch.tensor(True))
        decay = self.get_current_decay()
        for e, o in zip(self.ema, self.online):
            e.lerp_(o, 1.0 - decay)
