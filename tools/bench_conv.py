#!/usr/bin/env python3
"""Per-shape micro-benchmark of the implicit-GEMM convolution (AFHQ UNet / decoder shapes)."""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ddmi_amd import _lib  # noqa: E402

# (name, Cin, HW, Cout, k, stride, ups, count-per-UNet-step)
UNET_SHAPES = [
    ("224->224 k3 @64", 224, 64, 224, 3, 1, 0, 7), ("448->448 k3 @32", 448, 32, 448, 3, 1, 0, 6),
    ("896->896 k3 @16", 896, 16, 896, 3, 1, 0, 6), ("1344->1344 k3 @8", 1344, 8, 1344, 3, 1, 0, 10),
    ("448->448 k3 up @32->64", 448, 32, 448, 3, 1, 1, 1), ("896->896 k3 up @16->32", 896, 16, 896, 3, 1, 1, 1),
    ("1344->1344 k3 up @8->16", 1344, 8, 1344, 3, 1, 1, 1),
    ("448->224 k3 @64", 448, 64, 224, 3, 1, 0, 2), ("672->224 k3 @64", 672, 64, 224, 3, 1, 0, 1),
    ("1344->448 k3 @32", 1344, 32, 448, 3, 1, 0, 1), ("2240->896 k3 @16", 2240, 16, 896, 3, 1, 0, 1),
    ("2688->1344 k3 @8", 2688, 8, 1344, 3, 1, 0, 2), ("448->1344 k1 @32 (qkv)", 448, 32, 1344, 1, 1, 0, 5),
    ("1344->4032 k1 @8 (qkv)", 1344, 8, 4032, 1, 1, 0, 6), ("224->224 k3 s2 @64", 224, 64, 224, 3, 2, 0, 1),
    ("64->224 k3 @64 (in)", 64, 64, 224, 3, 1, 0, 1), ("224->64 k3 @64 (out)", 224, 64, 64, 3, 1, 0, 1),
    ("896->2688 k1 @16 (qkv)", 896, 16, 2688, 1, 1, 0, 5), ("448->448 k1 @32 (proj)", 448, 32, 448, 1, 1, 0, 5),
    ("896->896 k1 @16 (proj)", 896, 16, 896, 1, 1, 0, 5), ("1344->1344 k1 @8 (proj)", 1344, 8, 1344, 1, 1, 0, 6),
    ("224->224 k1 @64 (skip)", 224, 64, 224, 1, 1, 0, 5),
]
DEC_SHAPES = [("512->512 k3 @64", 512, 64, 512, 3, 1, 0, 10), ("256->256 k3 @128", 256, 128, 256, 3, 1, 0, 5),
              ("128->128 k3 @256", 128, 256, 128, 3, 1, 0, 5), ("512->512 k3 up @64->128", 512, 64, 512, 3, 1, 1, 1)]


# the model_channels = 256 networks (configs 4 / 5): the 3x3 shapes the 256-wide halo tile (configuration 26) serves
WIDE_SHAPES = [("256->256 k3 @32", 256, 32, 256, 3, 1, 0, 4), ("512->256 k3 @32", 512, 32, 256, 3, 1, 0, 2), ("768->256 k3 @32", 768, 32, 256, 3, 1, 0, 1),
               ("256->256 k3 @16", 256, 16, 256, 3, 1, 0, 4), ("512->512 k3 @16", 512, 16, 512, 3, 1, 0, 6), ("1024->512 k3 @16", 1024, 16, 512, 3, 1, 0, 2),
               ("1536->512 k3 @16", 1536, 16, 512, 3, 1, 0, 1)]


def run(B, shape, numerics, bm=0, bn=0, splitk=0, iters=20):
    name, cin, hw, cout, k, stride, ups, cnt = shape
    ms = C.c_double()
    sk = C.c_int()
    _lib.check(_lib.lib().ddmi_bench_conv(B, cin, hw, hw, cout, k, stride, ups, _lib.numerics_code(numerics), splitk, bm, bn,
                                          iters, C.byref(ms), C.byref(sk)))
    ho = (hw * (2 if ups else 1) + 2 * (k // 2) - k) // stride + 1
    flops = 2.0 * B * ho * ho * cout * cin * k * k
    return ms.value, flops / (ms.value * 1e-3) / 1e12, sk.value


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, nargs="+", default=[4, 8])
    ap.add_argument("--numerics", default="bf16")
    ap.add_argument("--decoder", action="store_true")
    ap.add_argument("--wide", action="store_true", help="the 3x3 shapes of the model_channels = 256 networks at W = 16 / 32")
    ap.add_argument("--only", default="", help="substring filter on shape names")
    ap.add_argument("--tiles", default="", help="comma list of BMxBN to force, e.g. 128x112,128x128")
    ap.add_argument("--cfgs", default="", help="comma list of bf16 ring config ids to force (k_conv.hip kRing)")
    args = ap.parse_args()
    _lib.init_device(0)
    tiles = [tuple(map(int, t.split("x"))) for t in args.tiles.split(",") if t] or [(0, 0)]
    if args.cfgs:
        tiles = [(int(c), -1) for c in args.cfgs.split(",")]
    shapes = DEC_SHAPES if args.decoder else WIDE_SHAPES if args.wide else UNET_SHAPES
    if args.only:
        shapes = [s for s in shapes if args.only in s[0]]
    for B in args.batch:
        tot_ms = tot_fl = 0.0
        for sh in shapes:
            for bm, bn in tiles:
                try:
                    ms, tf, sk = run(B, sh, args.numerics, bm, bn)
                except Exception as e:  # a forced tile may not exist
                    print(f"B={B} {sh[0]:28s} tile {bm}x{bn}: {e}")
                    continue
                print(f"B={B} {sh[0]:28s} tile {bm or 'auto'}x{bn or 'auto'} splitk {sk:2d}: {ms * 1e3:9.1f} us  {tf:7.1f} TFLOP/s")
                if (bm, bn) == tiles[0]:
                    tot_ms += ms * sh[7]
                    tot_fl += tf * ms * sh[7]
        print(f"B={B} weighted total {tot_ms:.2f} ms per UNet step (listed shapes), {tot_fl / tot_ms:.1f} TFLOP/s average")
