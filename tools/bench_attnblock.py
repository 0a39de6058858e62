"""Times one AttentionBlock of the AFHQ UNet per level (bf16, device-resident random data): fused qkv + attention launch against
the five-launch form.  python tools/bench_attnblock.py [batch] [iters]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ddmi_amd import _lib  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    L = _lib.lib()
    _lib.init_device(0)
    for Cc, H in ((448, 32), (896, 16), (1344, 8)):
        row = []
        for fused in (0, 1):
            _lib.check(L.ddmi_debug_attn_fused(fused))
            us, used = C.c_double(0), C.c_int(-1)
            _lib.profile_enable(1)
            _lib.check(L.ddmi_bench_attention_block(B, Cc, H, H, iters, C.byref(us), C.byref(used)))
            row.append((us.value, used.value))
            prof = _lib.profile_collect()
            print("   ", "fused" if fused else "five-launch", {k: round(v["total_ms"] * 1e3, 1) for k, v in prof.items()})
        T = B * H * H
        gflop = (2.0 * T * Cc * 3 * Cc + 4.0 * T * H * H * Cc + 2.0 * T * Cc * Cc) / 1e9
        print(f"C={Cc} {H}x{H} B={B}: five-launch {row[0][0]:9.1f} us (fused={row[0][1]})   fused {row[1][0]:9.1f} us "
              f"(fused={row[1][1]})   {gflop / row[1][0]:.3f} PFLOP/s fused, x{row[0][0] / row[1][0]:.2f}", flush=True)
    _lib.check(L.ddmi_debug_attn_fused(-1))


if __name__ == "__main__":
    main()
