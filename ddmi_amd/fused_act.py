"""`models/d2c_vae/op/fused_act.py` without nvcc: the module object `fused` the reference builds with torch's JIT loader
(fused_act.py:10-16 -> fused_bias_act.cpp / fused_bias_act_kernel.cu) and the three names the rest of the reference imports from
that package - `FusedLeakyReLU`, `fused_leaky_relu` (blocks.py) - on the HIP kernel `ddmi_fused_bias_act`.

Drop-in: in the reference's `models/d2c_vae/op/fused_act.py` replace the `fused = load(...)` block by
`from ddmi_amd.fused_act import fused`; or import `FusedLeakyReLU` / `fused_leaky_relu` from here directly.
On the generation path this operator is fused into the MLP kernels (k_mlp2d.hip); this module serves the reference's own
module code, e.g. training the D2C-VAE's MLP on ROCm."""
from __future__ import annotations

import torch
from torch import nn

from . import _lib

_DTYPES = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}


class _Fused:
    """What `torch.utils.cpp_extension.load('fused', ...)` returns for the reference: one function."""

    @staticmethod
    def fused_bias_act(input, bias, refer, act, grad, alpha, scale):
        if not input.is_cuda:
            raise RuntimeError("input must be a CUDA (HIP) tensor")        # CHECK_CUDA(input) of fused_bias_act.cpp:14
        if input.dtype not in _DTYPES:
            raise TypeError(f"fused_bias_act: unsupported dtype {input.dtype}")
        x = input.contiguous()
        b = bias.contiguous().to(x.dtype) if bias is not None and bias.numel() else None
        r = refer.contiguous().to(x.dtype) if refer is not None and refer.numel() else None
        step_b = 1
        for d in x.shape[2:]:
            step_b *= int(d)
        y = torch.empty_like(x)
        _lib.check(_lib.lib().ddmi_fused_bias_act(x.data_ptr(), b.data_ptr() if b is not None else None,
                                                  r.data_ptr() if r is not None else None, y.data_ptr(), x.numel(), step_b,
                                                  b.numel() if b is not None else 0, int(act), int(grad), float(alpha), float(scale),
                                                  _DTYPES[x.dtype], _lib.stream_ptr()))
        return y


fused = _Fused()


class _LeakyGrad(torch.autograd.Function):
    """d/dx and d/dbias of the forward below, itself differentiable (the reference's R1-style double backward)."""

    @staticmethod
    def forward(ctx, grad_out, out, slope, scale):
        ctx.save_for_backward(out)
        ctx.slope, ctx.scale = slope, scale
        gx = fused.fused_bias_act(grad_out, None, out, 3, 1, slope, scale)
        dims = [0] + list(range(2, gx.ndim))
        return gx, gx.sum(dims).detach()

    @staticmethod
    def backward(ctx, ggx, ggb):
        (out,) = ctx.saved_tensors
        return fused.fused_bias_act(ggx, ggb, out, 3, 1, ctx.slope, ctx.scale), None, None, None


class _Leaky(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, bias, slope, scale):
        out = fused.fused_bias_act(x, bias, None, 3, 0, slope, scale)
        ctx.save_for_backward(out)
        ctx.slope, ctx.scale = slope, scale
        return out

    @staticmethod
    def backward(ctx, grad_out):
        (out,) = ctx.saved_tensors
        gx, gb = _LeakyGrad.apply(grad_out, out, ctx.slope, ctx.scale)
        return gx, gb, None, None


def fused_leaky_relu(input, bias, negative_slope=0.2, scale=2 ** 0.5):
    return _Leaky.apply(input, bias, negative_slope, scale)


class FusedLeakyReLU(nn.Module):
    def __init__(self, channel, negative_slope=0.2, scale=2 ** 0.5):
        super().__init__()
        self.bias = nn.Parameter(torch.zeros(channel))
        self.negative_slope, self.scale = negative_slope, scale

    def forward(self, input):
        return fused_leaky_relu(input, self.bias, self.negative_slope, self.scale)
