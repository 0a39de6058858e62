// On-box peak microbenchmarks ("a STREAM-like and a pure-MFMA microbench on the box, report against both"):
//   ddmi_bench_peak_mfma  - nothing but v_mfma_f32_16x16x32_bf16 on pseudo-random operands held in registers, every CU, two waves
//                           per SIMD, independent accumulators: the bf16 MFMA rate the chip sustains at the clock it holds under that
//                           load (MI355X_MICROARCH.md, DVFS give-back: well under the 2.4 GHz the 2.5 PFLOP/s figure is quoted at)
//   ddmi_bench_peak_copy  - a 16-byte-per-lane copy of >= 2 GiB: read + written bytes per second
// bench.py runs both before its timed region and reports them beside the nominal peaks (roofline.measured_peak).
#include "kernels.h"
#include <vector>

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_mb;

__device__ __forceinline__ unsigned mb_hash(unsigned h) {
  h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
  return h;
}

__global__ void __launch_bounds__(512) peak_mfma_kernel(float* out, int iters) {
  // bf16 values in +-[0.5, 2): sign + exponent 126 / 127 + 7 random mantissa bits per half-word
  bf16x8 a[2], b[2];
  const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    u32x4_mb ua, ub;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const unsigned h1 = mb_hash(tid * 16u + r * 8u + e), h2 = mb_hash(tid * 16u + r * 8u + 4u + e + 0x9E3779B9u);
      ua[e] = 0x3F003F00u | (h1 & 0x80FF80FFu);
      ub[e] = 0x3F003F00u | (h2 & 0x80FF80FFu);
    }
    a[r] = __builtin_bit_cast(bf16x8, ua);
    b[r] = __builtin_bit_cast(bf16x8, ub);
  }
  f32x4 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[(u + i) & 1], b[u & 1], acc[i], 0, 0, 0);
  }
  f32x4 s = acc[0];
#pragma unroll
  for (int i = 1; i < 8; ++i) s += acc[i];
  if (s[0] + s[1] + s[2] + s[3] == 12345.678f) out[tid] = s[0];   // keeps the loop alive; never true in practice
}

template <int U, bool NT>
__global__ void __launch_bounds__(256) peak_copy_kernel(const u32x4_mb* __restrict__ src, u32x4_mb* __restrict__ dst, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + (U - 1) * stride < n; i += U * stride) {   // U loads in flight per thread
    u32x4_mb v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(src + i + u * stride) : src[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (NT) __builtin_nontemporal_store(v[u], dst + i + u * stride);
      else dst[i + u * stride] = v[u];
    }
  }
  for (; i < n; i += stride) dst[i] = src[i];
}

// contiguous chunk per workgroup (the layout of the library's streaming passes: a workgroup walks its own 16-byte-vector range)
template <int U>
__global__ void __launch_bounds__(256) peak_copy_chunk_kernel(const u32x4_mb* __restrict__ src, u32x4_mb* __restrict__ dst, int64_t n) {
  const int64_t per = (n + gridDim.x - 1) / gridDim.x, i0 = (int64_t)blockIdx.x * per, i1 = i0 + per < n ? i0 + per : n;
  int64_t i = i0 + threadIdx.x;
  for (; i + (U - 1) * 256 < i1; i += U * 256) {
    u32x4_mb v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(src + i + u * 256);
#pragma unroll
    for (int u = 0; u < U; ++u) __builtin_nontemporal_store(v[u], dst + i + u * 256);
  }
  for (; i < i1; i += 256) dst[i] = src[i];
}

// average over `launches` launches of `iters` x 32 MFMAs per wave; 256 workgroups x 8 waves x ... = every CU at two waves per SIMD
int bench_peak_mfma(double seconds_target, double* tflops, double* ms_per_launch) {
  int dev = 0, cus = 256;
  HIP_TRY(hipGetDevice(&dev));
  HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  float* out = nullptr;
  HIP_TRY(hipMalloc((void**)&out, (size_t)cus * 4 * 512 * sizeof(float)));
  const int blocks = cus * 4;          // four rounds of one 512-thread workgroup per CU (a workgroup's 8 waves = two per SIMD);
  const int lds = 96 * 1024;           // 96 KB of (unused) dynamic LDS per workgroup keeps it at ONE workgroup per CU
  const int iters = 4096;              // 131 072 MFMAs per wave and launch: ~1 ms at the nominal rate
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(peak_mfma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(peak_mfma_kernel, dim3(blocks), dim3(512), lds, 0, out, iters);   // warm-up (clock ramp, code load)
  hipError_t err = hipDeviceSynchronize();
  int launches = 0;
  float ms = 0.f;
  if (err == hipSuccess) {
    // first a short timed burst to size the run, then the run itself
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(peak_mfma_kernel, dim3(blocks), dim3(512), lds, 0, out, iters);
    (void)hipEventRecord(e1, 0);
    err = hipDeviceSynchronize();
    float one = 1.f;
    (void)hipEventElapsedTime(&one, e0, e1);
    launches = (int)std::max(3.0, std::min(400.0, seconds_target * 1e3 / std::max(one, 0.05f)));
    (void)hipEventRecord(e0, 0);
    for (int i = 0; i < launches; ++i) hipLaunchKernelGGL(peak_mfma_kernel, dim3(blocks), dim3(512), lds, 0, out, iters);
    (void)hipEventRecord(e1, 0);
    if (err == hipSuccess) err = hipDeviceSynchronize();
    (void)hipEventElapsedTime(&ms, e0, e1);
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipFree(out);
  if (err != hipSuccess) return ddmi_fail(DDMI_ERR_HIP, std::string("peak_mfma: ") + hipGetErrorString(err));
  const double flop = (double)launches * blocks * 8 /*waves*/ * iters * 32.0 * (2.0 * 16 * 16 * 32);
  *tflops = flop / (ms * 1e-3) / 1e12;
  if (ms_per_launch) *ms_per_launch = ms / launches;
  return DDMI_OK;
}

int bench_peak_copy(double gib, double* tb_per_s) {
  const int64_t n = (int64_t)(gib * (1ull << 30)) / 16;
  u32x4_mb *src = nullptr, *dst = nullptr;
  HIP_TRY(hipMalloc((void**)&src, (size_t)n * 16));
  if (hipMalloc((void**)&dst, (size_t)n * 16) != hipSuccess) {
    (void)hipFree(src);
    return ddmi_fail(DDMI_ERR_HIP, "peak_copy: out of device memory");
  }
  (void)hipMemsetAsync(src, 0x3C, (size_t)n * 16, 0);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  const int reps = 5;
  // a few launch shapes (loads in flight per thread, cache policy, grid size); the best one is "what the box sustains"
  struct Variant { void (*k)(const u32x4_mb*, u32x4_mb*, int64_t); int blocks; };
  const Variant variants[] = {{peak_copy_kernel<4, false>, 256 * 16}, {peak_copy_kernel<8, false>, 256 * 8}, {peak_copy_kernel<8, true>, 256 * 8},
                              {peak_copy_kernel<8, true>, 256 * 32}, {peak_copy_kernel<4, true>, 256 * 16},
                              {peak_copy_chunk_kernel<8>, 256 * 8}, {peak_copy_chunk_kernel<8>, 256 * 64}, {peak_copy_chunk_kernel<4>, 256 * 256}};
  hipError_t err = hipSuccess;
  float best_ms = 1e30f;
  for (const Variant& v : variants) {
    hipLaunchKernelGGL(v.k, dim3(v.blocks), dim3(256), 0, 0, src, dst, n);
    (void)hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(v.k, dim3(v.blocks), dim3(256), 0, 0, src, dst, n);
    (void)hipEventRecord(e1, 0);
    err = hipDeviceSynchronize();
    if (err != hipSuccess) break;
    float t = 0.f;
    (void)hipEventElapsedTime(&t, e0, e1);
    best_ms = std::min(best_ms, t);
  }
  const float ms = best_ms;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipFree(src);
  (void)hipFree(dst);
  if (err != hipSuccess) return ddmi_fail(DDMI_ERR_HIP, std::string("peak_copy: ") + hipGetErrorString(err));
  *tb_per_s = 2.0 * (double)n * 16 * reps / (ms * 1e-3) / 1e12;
  return DDMI_OK;
}
