"""Import-only stand-in for `models/d2c_vae/op/upfirdn2d.py`.

The reference's `op/__init__.py:2` imports `upfirdn2d` next to the fused activation, and `op/upfirdn2d.py:9-15` JIT-builds
`upfirdn2d.cpp` + `upfirdn2d_kernel.cu` with nvcc AT IMPORT, so `models.d2c_vae.blocks` (blocks.py:8) cannot be imported on a ROCm box
even though nothing on the generation path ever calls the operator: its only users are `Blur`, `Upsample` and `Downsample`
(blocks.py:44-101), which the image MLP of mlp.py:34-66 never builds.  This module gives the package the names it imports -
`upfirdn2d`, `upfirdn2d_native` and the module object `upfirdn2d_op` the loader would have returned - and fails loudly when one of
them is CALLED (there is no CPU fallback and no silent identity).

Reference-side change: in `models/d2c_vae/op/upfirdn2d.py` replace the `upfirdn2d_op = load(...)` block by
`from ddmi_amd.upfirdn2d import upfirdn2d_op`, or import `upfirdn2d` from here in `op/__init__.py`.
"""
from __future__ import annotations

_WHY = ("upfirdn2d is not on the DDMI generation path (only Blur / Upsample / Downsample of models/d2c_vae/blocks.py:44-101 "
        "call it, and mlp.py's MLP builds none of them); ddmi_amd ships it as an import-only stand-in")


class _UpFirDn2dOp:
    """What `torch.utils.cpp_extension.load('upfirdn2d', ...)` returns for the reference: one function (upfirdn2d.cpp)."""

    @staticmethod
    def upfirdn2d(input, kernel, up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1):
        raise NotImplementedError(_WHY)


upfirdn2d_op = _UpFirDn2dOp()


def upfirdn2d(input, kernel, up=1, down=1, pad=(0, 0)):
    """Signature of op/upfirdn2d.py:144-151."""
    raise NotImplementedError(_WHY)


def upfirdn2d_native(input, kernel, up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1):
    """Signature of op/upfirdn2d.py:153-187."""
    raise NotImplementedError(_WHY)
