"""`DDPM` - host-side mirror of diffusion/ddpm.py (sampling half: :76-217 schedule, :257-381 samplers).

The outer loop stays in Python on PyTorch-ROCm (as the north star asks); each iteration is one call of the native
denoiser (`self.model`, libddmi_hip's UNet) plus ONE fused HIP kernel for eps-mix -> x0 -> x_{t-1}
(`ddmi_ddpm_step`), instead of the reference's ~15 elementwise torch ops per step.
"""
from __future__ import annotations

import ctypes as C
from typing import Callable, Dict, List, Optional, Sequence, Tuple, Union

import numpy as np

from . import _lib


# ---------------------------------------------------------------------------------------------------------
# schedules (fp64 math -> fp32 buffers, exactly the formulae of ddpm.py:37-59 and :163-203)
# ---------------------------------------------------------------------------------------------------------
def _linspace(start, end, steps, dtype):
    """torch.linspace as ATen evaluates it: start + step*i below the midpoint, end - step*(steps-1-i) above."""
    start, end = dtype(start), dtype(end)
    if steps == 1:
        return np.array([start], dtype=dtype)
    step = dtype((end - start) / dtype(steps - 1))
    i = np.arange(steps).astype(dtype)
    lo = (start + step * i).astype(dtype)
    hi = (end - step * (dtype(steps - 1) - i)).astype(dtype)
    return np.where(np.arange(steps) < steps // 2, lo, hi).astype(dtype)


def make_beta_schedule(schedule, n_timestep, linear_start=1e-4, linear_end=2e-2, cosine_s=8e-3):
    if schedule == "linear":
        return _linspace(linear_start ** 0.5, linear_end ** 0.5, n_timestep, np.float64) ** 2
    if schedule == "cosine":
        ts = np.arange(n_timestep + 1, dtype=np.float64) / n_timestep + cosine_s
        alphas = np.cos(ts / (1 + cosine_s) * np.pi / 2) ** 2
        alphas = alphas / alphas[0]
        return np.clip(1 - alphas[1:] / alphas[:-1], 0, 0.999)
    if schedule == "sqrt_linear":
        return _linspace(linear_start, linear_end, n_timestep, np.float64)
    if schedule == "sqrt":
        return _linspace(linear_start, linear_end, n_timestep, np.float64) ** 0.5
    raise ValueError(f"schedule '{schedule}' unknown.")


def ddim_time_pairs(total_timesteps: int, sampling_timesteps: int) -> List[Tuple[int, int]]:
    """ddpm.py:338-340 - integer schedule, bit-exact: fp32 linspace(-1, T-1, S+1), truncation, reversed, paired."""
    times = _linspace(-1.0, float(total_timesteps - 1), sampling_timesteps + 1, np.float32)
    times = [int(v) for v in np.trunc(times).astype(np.int64)][::-1]
    return list(zip(times[:-1], times[1:]))


SCHEDULE_BUFFERS = ("betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod",
                    "sqrt_one_minus_alphas_cumprod", "log_one_minus_alphas_cumprod", "sqrt_recip_alphas_cumprod",
                    "sqrt_recipm1_alphas_cumprod", "posterior_variance", "posterior_log_variance_clipped",
                    "posterior_mean_coef1", "posterior_mean_coef2")


class DDPM:
    """Same constructor keywords as the reference (the YAML `ddpmconfig` block is splatted in, exp/stage.py:203)."""

    def __init__(self, model, timesteps=1000, beta_schedule="linear", loss_type="l2", image_size=256, channels=3,
                 clip_denoised=False, linear_start=0.0015, linear_end=0.0195, cosine_s=8e-3, given_betas=None,
                 v_posterior=0.0, parameterization="eps", sampling_timesteps=1000, ddim_sampling_eta=1.0, w=1.0,
                 mixed_prediction=True, mixed_init=-6, domain="image", **ignorekwargs):
        assert parameterization in ("eps", "x0", "v")
        self.model = model
        self.parameterization = parameterization
        self.clip_denoised = bool(clip_denoised)
        self.image_size, self.channels = image_size, int(channels)
        self.v_posterior = v_posterior
        self.domain = domain
        self.register_schedule(given_betas, beta_schedule, timesteps, linear_start, linear_end, cosine_s)
        self.sampling_timesteps = timesteps if sampling_timesteps is None else int(sampling_timesteps)
        assert self.sampling_timesteps <= timesteps
        self.is_ddim_sampling = self.sampling_timesteps < timesteps
        self.ddim_sampling_eta = float(ddim_sampling_eta)
        self.w = w
        self.mixed_prediction = bool(mixed_prediction)
        # [1,C,1,1] for image latents, [1,C,1] for the video token sequence [B,C,L] (ddpm.py:152-159)
        self._logit_shape = (1, self.channels, 1) if domain == "video" else (1, self.channels, 1, 1)
        self.mixing_logit = (np.full(self._logit_shape, mixed_init, dtype=np.float32)
                             if self.mixed_prediction else None)
        self._logit_dev = None

    # ---- schedule (ddpm.py:163-203) -----------------------------------------------------------------------
    def register_schedule(self, given_betas=None, beta_schedule="linear", timesteps=1000, linear_start=1e-4,
                          linear_end=2e-2, cosine_s=8e-3):
        betas = (np.asarray(given_betas, dtype=np.float64) if given_betas is not None else
                 make_beta_schedule(beta_schedule, timesteps, linear_start, linear_end, cosine_s))
        alphas = 1.0 - betas
        ac = np.cumprod(alphas, axis=0)
        ac_prev = np.append(1.0, ac[:-1])
        self.num_timesteps = int(betas.shape[0])
        f32 = lambda a: np.asarray(a, dtype=np.float64).astype(np.float32)
        post_var = (1 - self.v_posterior) * betas * (1.0 - ac_prev) / (1.0 - ac) + self.v_posterior * betas
        self.betas = f32(betas)
        self.alphas_cumprod = f32(ac)
        self.alphas_cumprod_prev = f32(ac_prev)
        self.sqrt_alphas_cumprod = f32(np.sqrt(ac))
        self.sqrt_one_minus_alphas_cumprod = f32(np.sqrt(1.0 - ac))
        self.log_one_minus_alphas_cumprod = f32(np.log(1.0 - ac))
        self.sqrt_recip_alphas_cumprod = f32(np.sqrt(1.0 / ac))
        self.sqrt_recipm1_alphas_cumprod = f32(np.sqrt(1.0 / ac - 1))
        self.posterior_variance = f32(post_var)
        self.posterior_log_variance_clipped = f32(np.log(np.maximum(post_var, 1e-20)))
        self.posterior_mean_coef1 = f32(betas * np.sqrt(ac_prev) / (1.0 - ac))
        self.posterior_mean_coef2 = f32((1.0 - ac_prev) * np.sqrt(alphas) / (1.0 - ac))

    # ---- state dict: 'model.*' + 12 schedule buffers + 'mixing_logit' (tools/ldm/image.py:79) -----------------
    def load_state_dict(self, state, strict: bool = True):
        model_state = {k[len("model."):]: v for k, v in state.items() if k.startswith("model.")}
        if model_state or strict:
            self.model.load_state_dict(model_state, strict=strict)
        if "mixing_logit" in state and self.mixed_prediction:
            v = state["mixing_logit"]
            v = v.detach().float().cpu().numpy() if hasattr(v, "detach") else np.asarray(v, dtype=np.float32)
            self.mixing_logit = v.reshape(self._logit_shape).astype(np.float32)
            self._logit_dev = None
        for name in SCHEDULE_BUFFERS:  # buffers are derived, a checkpoint's copy must agree with ours
            if name in state:
                v = state[name]
                v = v.detach().float().cpu().numpy() if hasattr(v, "detach") else np.asarray(v, dtype=np.float32)
                if v.shape != getattr(self, name).shape or not np.allclose(v, getattr(self, name), rtol=1e-5, atol=1e-8):
                    raise ValueError(f"schedule buffer {name} in the checkpoint disagrees with this ddpmconfig")
        return self

    def eval(self):
        return self

    def to(self, *a, **k):
        return self

    # ---- per-step coefficient blocks ---------------------------------------------------------------------------
    def _base_coefs(self, t: int) -> _lib.StepCoefs:
        k = _lib.StepCoefs()
        k.sqrt_one_minus_ac = float(self.sqrt_one_minus_alphas_cumprod[t])
        k.sqrt_recip_ac = float(self.sqrt_recip_alphas_cumprod[t])
        k.sqrt_recipm1_ac = float(self.sqrt_recipm1_alphas_cumprod[t])
        return k

    def ancestral_coefs(self, t: int) -> _lib.StepCoefs:
        """ddpm.py:248-255, 280-281."""
        k = self._base_coefs(t)
        k.coef1 = float(self.posterior_mean_coef1[t])
        k.coef2 = float(self.posterior_mean_coef2[t])
        lv = np.float32(self.posterior_log_variance_clipped[t])
        k.noise_scale = 0.0 if t == 0 else float(np.exp(np.float32(0.5) * lv, dtype=np.float32))
        return k

    def ddim_coefs(self, t: int, t_next: int) -> _lib.StepCoefs:
        """ddpm.py:357-361 with fp32 scalars in the same operation order."""
        k = self._base_coefs(t)
        one = np.float32(1.0)
        alpha, alpha_next = np.float32(self.alphas_cumprod[t]), np.float32(self.alphas_cumprod[t_next])
        eta = np.float32(self.ddim_sampling_eta)
        sigma = np.float32(eta * np.sqrt(np.float32(np.float32((one - alpha / alpha_next) * (one - alpha_next)) / (one - alpha))))
        c = np.sqrt(np.float32(one - alpha_next - np.float32(sigma * sigma)))
        k.sqrt_ac_next = float(np.sqrt(alpha_next))
        k.c_dir = float(c)
        k.sigma = float(sigma)
        return k

    # ---- fused step launcher ------------------------------------------------------------------------------------
    def _logit_ptr(self, device):
        import torch
        if not self.mixed_prediction:
            return None
        if self._logit_dev is None or self._logit_dev.device != device:
            self._logit_dev = torch.from_numpy(self.mixing_logit.reshape(-1).copy()).to(device)
        return self._logit_dev.data_ptr()

    def _step(self, x, model_out, noise, coefs, mode, out, clip, seeds=None, tag=0):
        B, Cc = x.shape[0], x.shape[1]
        hw = int(np.prod(x.shape[2:]))
        if seeds is not None:  # noise drawn inside the kernel from the per-chain Philox streams
            _lib.check(_lib.lib().ddmi_ddpm_step_rng(x.data_ptr(), model_out.data_ptr(), seeds.data_ptr(), int(tag),
                                                     self._logit_ptr(x.device), C.byref(coefs), mode, int(clip),
                                                     out.data_ptr(), None, B, Cc, hw, _lib.stream_ptr()))
            return out
        _lib.check(_lib.lib().ddmi_ddpm_step(x.data_ptr(), model_out.data_ptr(),
                                             noise.data_ptr() if noise is not None else None,
                                             self._logit_ptr(x.device), C.byref(coefs), mode, int(clip),
                                             out.data_ptr(), None, B, Cc, hw, _lib.stream_ptr()))
        return out

    X_T_TAG = 0xFFFFFFFF  # Philox stream of a chain's initial noise; step n of the loop uses stream n

    @staticmethod
    def randn_from_seeds(shape, seeds, tag):
        """[B, ...] standard normals from the per-chain streams: row b = normal(seeds[b], tag, .) - what
        `ddmi_ddpm_step_rng` draws in registers for that tag."""
        import torch
        out = torch.empty(tuple(shape), device=seeds.device, dtype=torch.float32)
        n = int(np.prod(shape[1:]))
        _lib.check(_lib.lib().ddmi_randn_fill(out.data_ptr(), seeds.data_ptr(), int(shape[0]), n, int(tag), _lib.stream_ptr()))
        return out

    @staticmethod
    def _noise_source(step_noise, shape, device, generator):
        import torch
        if step_noise is None:
            return lambda i: torch.randn(shape, device=device, generator=generator)
        if callable(step_noise):
            return lambda i: step_noise(i).to(device=device, dtype=torch.float32).contiguous()
        return lambda i: torch.as_tensor(step_noise[i]).to(device=device, dtype=torch.float32).contiguous()

    # ---- samplers ---------------------------------------------------------------------------------------------------
    def p_sample_loop(self, shape, cond=None, return_intermediates=False, noise=None, enable_mask=False,
                      step_noise=None, device=None, generator=None, seeds=None):
        """ddpm.py:283-300.  One `randn` per step including t=0 (ddpm.py:278), multiplied by 0 there.
        `seeds` ([B] int64 on the device): per-chain Philox streams - x_T and every step's noise are drawn in the
        library (no noise tensors, no per-sample generator launches) and do not depend on the batch composition."""
        import torch
        assert cond is None
        device = torch.device(device or "cuda")
        if seeds is not None:
            assert step_noise is None and generator is None
            seeds = seeds.to(device=device, dtype=torch.int64).contiguous()
        if noise is not None:
            img = torch.as_tensor(noise).to(device=device, dtype=torch.float32).contiguous().clone()
        elif seeds is not None:
            img = self.randn_from_seeds(shape, seeds, self.X_T_TAG)
        else:
            img = torch.randn(tuple(shape), device=device, generator=generator)
        draw = self._noise_source(step_noise, tuple(shape), device, generator)
        B = shape[0]
        nxt = torch.empty_like(img)
        eps = torch.empty_like(img)
        tt = torch.empty((B,), device=device, dtype=torch.int64)
        intermediates = [img.clone()] if return_intermediates else None
        if hasattr(self.model, "set_uniform_t"):
            self.model.set_uniform_t(True)  # torch.full((b,), i): every chain shares the timestep (ddpm.py:294)
        for n, i in enumerate(reversed(range(self.num_timesteps))):
            tt.fill_(i)
            self.model(img, tt, cond, out=eps, enable_mask=False)
            nz = draw(n) if seeds is None else None
            self._step(img, eps, nz, self.ancestral_coefs(i), _lib.STEP_ANCESTRAL, nxt, self.clip_denoised, seeds, n)
            img, nxt = nxt, img
            if return_intermediates:
                intermediates.append(img.clone())
        if hasattr(self.model, "set_uniform_t"):
            self.model.set_uniform_t(False)
        return (img, intermediates) if return_intermediates else img

    def ddim_sample(self, shape, cond=None, clip_denoised=False, noise=None, enable_mask=False, step_noise=None,
                    device=None, generator=None, seeds=None):
        """ddpm.py:334-369.  A `randn_like` is drawn on every non-final step even when eta == 0 (ddpm.py:363).
        `seeds`: as in p_sample_loop."""
        import torch
        assert cond is None
        device = torch.device(device or "cuda")
        if seeds is not None:
            assert step_noise is None and generator is None
            seeds = seeds.to(device=device, dtype=torch.int64).contiguous()
        if noise is not None:
            img = torch.as_tensor(noise).to(device=device, dtype=torch.float32).contiguous().clone()
        elif seeds is not None:
            img = self.randn_from_seeds(shape, seeds, self.X_T_TAG)
        else:
            img = torch.randn(tuple(shape), device=device, generator=generator)
        draw = self._noise_source(step_noise, tuple(shape), device, generator)
        B = shape[0]
        nxt = torch.empty_like(img)
        eps = torch.empty_like(img)
        tt = torch.empty((B,), device=device, dtype=torch.int64)
        if hasattr(self.model, "set_uniform_t"):
            self.model.set_uniform_t(True)  # ddpm.py:349
        for n, (t, t_next) in enumerate(ddim_time_pairs(self.num_timesteps, self.sampling_timesteps)):
            tt.fill_(t)
            self.model(img, tt, cond, out=eps, enable_mask=enable_mask)
            if t_next < 0:
                self._step(img, eps, None, self._base_coefs(t), _lib.STEP_DDIM_LAST, nxt, clip_denoised)
            else:
                k = self.ddim_coefs(t, t_next)
                use_seeds = seeds if (seeds is not None and k.sigma != 0.0) else None
                nz = draw(n) if (seeds is None and (k.sigma != 0.0 or step_noise is not None)) else None
                self._step(img, eps, nz, k, _lib.STEP_DDIM, nxt, clip_denoised, use_seeds, n)
            img, nxt = nxt, img
        if hasattr(self.model, "set_uniform_t"):
            self.model.set_uniform_t(False)
        return img

    def sample(self, shape, cond=None, return_intermediates=False, noise=None, enable_mask=False, **kw):
        """ddpm.py:371-381: DDIM when sampling_timesteps < timesteps, ancestral loop otherwise.

        Deferred errors of the asynchronous UNet forwards are looked at once per chain batch (`model.async_status()`, at the point where
        the chain is complete - it synchronises).  The one such error that exists - a fused GroupNorm hand-off of the bf16 UNet that timed
        out waiting for a partner tile, flagged instead of hung - leaves the library with the fusion switched OFF for that handle (and
        its captured graphs dropped); the chain is then sampled AGAIN, once, on the separate GroupNorm passes, so no batch is lost
        (`gn_fuse_fallbacks` counts these; with `seeds`, `noise` / `step_noise` the re-run is the same sample, with a torch `generator` it
        is a fresh draw)."""
        def run():
            if self.is_ddim_sampling:
                out = self.ddim_sample(shape, cond, noise=noise, enable_mask=enable_mask, **kw)
            else:
                out = self.p_sample_loop(shape, cond, return_intermediates=return_intermediates, noise=noise,
                                         enable_mask=enable_mask, **kw)
            status = getattr(self.model, "async_status", None)
            if status is not None:
                self.fused_gn_launches = status()
            return out
        try:
            return run()
        except _lib.DdmiError as e:
            if "fused GroupNorm" not in str(e):
                raise
            self.gn_fuse_fallbacks = getattr(self, "gn_fuse_fallbacks", 0) + 1
            self.gn_fuse_fallback_message = str(e)
            if hasattr(self.model, "set_uniform_t"):
                self.model.set_uniform_t(False)   # (the loop was left in the middle)
        return run()   # a second failure is the caller's
