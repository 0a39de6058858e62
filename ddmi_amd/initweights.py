"""Named-seed synthetic weights and noise (numpy only, bit-reproducible everywhere).

There is no network for checkpoints, so benchmarks and parity tests run on
random-init weights.  A freshly constructed reference UNet outputs exactly 0
(`zero_module`, models/ldm/modules/diffusionmodules/util.py:173-179 zeroes the
last conv of every ResBlock, every attention `proj_out` and the output conv),
which would make parity vacuous, so every tensor - including the zero-initialised
ones - is filled from a generator keyed on (tag, state-dict key).

The generator uses only integer arithmetic plus one IEEE division, so the bytes
are identical on every machine, numpy version and SIMD path:
    z_i   = splitmix64(seed64 + i * GOLDEN)            (uint64)
    n_i   = (sum of the eight 16-bit fields of z_2i, z_2i+1  -  8*32767.5) / std8
which is an Irwin-Hall(8) approximation of N(0, 1) (exact variance, |n| < 4.9).
"""
from __future__ import annotations

import zlib
from typing import Dict, Iterable, Mapping, Sequence, Tuple

import numpy as np

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
# std of a sum of 8 iid uniform integers on [0, 65535]: sqrt(8 * (65536**2 - 1) / 12)
_STD8 = float(np.sqrt(8.0 * (65536.0 ** 2 - 1.0) / 12.0))


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = x + _GOLDEN
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def seed_of(tag: str, key: str) -> int:
    """64-bit seed from two crc32s of the (tag, key) pair."""
    lo = zlib.crc32((tag + ":" + key).encode())
    hi = zlib.crc32((key + "#" + tag).encode())
    return (hi << 32) | lo


_CHUNK = 1 << 17  # elements per block: the ~10 temporaries of a block stay cache-resident (a 16 M-element conv weight
                  # otherwise walks 2.5 GB of fresh pages; same values either way - every element is a pure function of i)


def _std_normal_block(seed: int, start: int, n: int) -> np.ndarray:
    with np.errstate(over="ignore"):
        idx = np.arange(2 * start, 2 * (start + n), dtype=np.uint64) * _GOLDEN + np.uint64(seed & 0xFFFFFFFFFFFFFFFF)
    z = _splitmix64(idx)
    s = np.zeros(2 * n, dtype=np.int64)
    for sh in (0, 16, 32, 48):
        s += ((z >> np.uint64(sh)) & np.uint64(0xFFFF)).astype(np.int64)
    tot = s[0::2] + s[1::2]  # exact integers
    return (tot.astype(np.float64) - 8.0 * 32767.5) / _STD8


def std_normal(seed: int, n: int) -> np.ndarray:
    """`n` approximately-N(0,1) float64 values, bit-reproducible (see module doc)."""
    if n <= _CHUNK:
        return _std_normal_block(seed, 0, n)
    out = np.empty(n, dtype=np.float64)
    for a in range(0, n, _CHUNK):
        b = min(n, a + _CHUNK)
        out[a:b] = _std_normal_block(seed, a, b - a)
    return out


def normal(tag: str, key: str, shape: Sequence[int], mean: float = 0.0, std: float = 1.0) -> np.ndarray:
    n = int(np.prod(shape)) if len(shape) else 1
    seed = seed_of(tag, key)
    if n <= _CHUNK:
        return (std_normal(seed, n) * std + mean).astype(np.float32).reshape(tuple(shape))
    out = np.empty(n, dtype=np.float32)      # fp64 -> scale -> shift -> fp32 per block: the same per-element operations
    for a in range(0, n, _CHUNK):
        b = min(n, a + _CHUNK)
        out[a:b] = (_std_normal_block(seed, a, b - a) * std + mean).astype(np.float32)
    return out.reshape(tuple(shape))


# --------------------------------------------------------------------------------------
# fill rules


def _rule(key: str, shape: Tuple[int, ...]) -> Tuple[float, float]:
    """(mean, std) for one state-dict entry.  Patterns follow the reference's own
    initialisers where they are not degenerate (blocks.py:108-117,145-148,228-232)."""
    leaf = key.rsplit(".", 1)[-1]
    if key.endswith("mixing_logit"):
        # reference init is -6 (ddpm.py:155); centred on 0 here so that the UNet output
        # carries about half of the mixed prediction and parity on chains is not vacuous
        return 0.0, 0.5
    if key.endswith("noise.weight"):
        return 0.0, 0.0  # NoiseInjection (blocks.py:290) - kept at its init
    if leaf == "weight" and len(shape) >= 2:
        if ".modulation." in key or ".conv.weight" in key and len(shape) == 5 or key.endswith("skip.0.weight"):
            return 0.0, 1.0  # StyleGAN equalised-lr tensors are N(0,1) + runtime scale
        fan_in = int(np.prod(shape[1:]))
        return 0.0, 1.0 / np.sqrt(fan_in)
    if leaf == "weight":  # GroupNorm / 1-D affine scale
        return 1.0, 0.1
    if ".modulation.bias" in key:
        return 1.0, 0.02  # bias_init=1 (blocks.py:232)
    return 0.0, 0.02  # biases and everything else


def fill_state(tag: str, manifest: Iterable[Tuple[str, Sequence[int]]]) -> Dict[str, np.ndarray]:
    """Deterministic fp32 tensors for every `(key, shape)` of a state-dict manifest.  Tensors are independent (each has its
    own seed), so large manifests are filled by a few threads (numpy releases the GIL inside its loops)."""
    items = [(key, tuple(int(s) for s in shape)) for key, shape in manifest]

    def one(item):
        key, shape = item
        mean, std = _rule(key, shape)
        if std == 0.0:
            return np.full(shape, mean, dtype=np.float32)
        return normal(tag, key, shape, mean, std)

    total = sum(int(np.prod(sh)) if len(sh) else 1 for _, sh in items)
    if total < (1 << 22):
        return {it[0]: one(it) for it in items}
    import os
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 4)) as ex:
        vals = list(ex.map(one, items))
    return {it[0]: v for it, v in zip(items, vals)}


def fill_like(tag: str, state: Mapping[str, "np.ndarray"]) -> Dict[str, np.ndarray]:
    return fill_state(tag, [(k, tuple(v.shape)) for k, v in state.items()])
