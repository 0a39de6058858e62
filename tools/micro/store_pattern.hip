// Micro-benchmark: issue rate of 16-byte-per-lane global stores from 8 waves of a CU for three lane->address maps.
//   0: lane l writes bytes [16 l, 16 l + 16) of a 1 KiB block                      (fully contiguous per instruction)
//   1: the conv epilogue's map: lane (lr = l & 15, lg = l >> 4) -> row lr, 16-byte chunk lg of a 64-byte row piece
//   2: lane l -> row l (every lane its own cache line)
//   3: same bytes as 1, lanes transposed: lane l -> row l >> 2, chunk l & 3 (4 adjacent lanes = one 64-byte piece)
//   4: 8 rows x 128 bytes per instruction: lane l -> row l >> 3, chunk l & 7
// hipcc --offload-arch=gfx950 -O3 tools/micro/store_pattern.hip -o tools/micro/store_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) float f32x4;
template <int MODE>
__global__ __launch_bounds__(512) void k(f32x4* out, int iters, long row_stride16, long wg_stride16) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x4 v = {1.f * lane, 2.f, 3.f, 4.f};
  for (int it = 0; it < iters; ++it) {
    f32x4* base = out + ((long)blockIdx.x * iters + it) % 4096 * wg_stride16 + wave * 16 * row_stride16 * 0 + wave * 4096;
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      long off;
      if (MODE == 0) off = s * 64 + lane;
      else if (MODE == 1) off = (long)(lane & 15) * row_stride16 + s * 4 + (lane >> 4);
      else if (MODE == 2) off = (long)lane * row_stride16 + s;
      else if (MODE == 3) off = (long)(lane >> 2) * row_stride16 + s * 4 + (lane & 3);
      else off = (long)((lane >> 3) + 8 * (s & 1)) * row_stride16 + (s >> 1) * 8 + (lane & 7);
      base[off] = v;
    }
  }
}
int main() {
  f32x4* buf;
  size_t bytes = (size_t)3 << 30;
  hipMalloc(&buf, bytes);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 64;
  const long row = 2688 / 16, wg = (long)(bytes / 16) / 4200;
  for (int mode = 0; mode < 5; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      if (mode == 0) k<0><<<256, 512>>>(buf, iters, row, wg);
      if (mode == 1) k<1><<<256, 512>>>(buf, iters, row, wg);
      if (mode == 2) k<2><<<256, 512>>>(buf, iters, row, wg);
      if (mode == 3) k<3><<<256, 512>>>(buf, iters, row, wg);
      if (mode == 4) k<4><<<256, 512>>>(buf, iters, row, wg);
      hipEventRecord(e1);
      hipDeviceSynchronize();
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      double gb = 256.0 * iters * 8 * 16 * 1024 / 1e9;
      if (rep) printf("mode %d: %.3f ms, %.1f GB/s, %.2f us per 128-store tile\n", mode, ms, gb / ms * 1e3, ms * 1e3 / iters);
    }
  }
  return 0;
}
