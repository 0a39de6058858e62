// HBM-bound elementwise / layout kernels: fused DDPM-DDIM step, NCHW<->NHWC, concat, weight packing,
// timestep embedding and the small fp32 dense layers of the time-embedding path.
#include "kernels.h"
#include "prof.h"
#include "philox.h"

// =====================================================================================================
// Fused eps-mix -> x0 -> x_{t-1} step.  Replaces diffusion/ddpm.py:257-281 (p_mean_variance + p_sample) and
// ddpm.py:303-332,353-368 (model_predictions + DDIM update) with one pass: reads x, model_out, noise (16 B/elem
// algorithmic traffic with the write), keeps everything in registers.
// Layout NCHW fp32 (the sampler state keeps the reference's layout); mixing_logit is [C].
// =====================================================================================================
// out[b][e] = normal(seeds[b], tag, e), e < n (n a multiple of 4)
__global__ void __launch_bounds__(256) randn_fill_kernel(float* __restrict__ out, const uint64_t* __restrict__ seeds,
                                                         int64_t nquad_per_sample, int64_t total_quads, unsigned tag) {
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total_quads; i += stride) {
    int64_t b = i / nquad_per_sample;
    unsigned q = (unsigned)(i - b * nquad_per_sample);
    float z[4];
    normal4(seeds[b], tag, q, z);
    *reinterpret_cast<float4*>(out + i * 4) = make_float4(z[0], z[1], z[2], z[3]);
  }
}

int launch_randn_fill(float* out, const uint64_t* seeds, int B, int64_t n_per_sample, unsigned tag, hipStream_t stream) {
  DDMI_REQUIRE(out && seeds && B > 0 && n_per_sample > 0, "randn_fill: bad arguments");
  DDMI_REQUIRE(n_per_sample % 4 == 0 && n_per_sample / 4 <= 0xFFFFFFFFll, "randn_fill: elements per sample must be a multiple of 4");
  int64_t nq = n_per_sample / 4, total = nq * B;
  int blocks = (int)std::min<int64_t>(cdiv64(total, 256), 256 * 8);
  hipLaunchKernelGGL(randn_fill_kernel, dim3(blocks), dim3(256), 0, stream, out, seeds, nq, total, tag);
  HIP_TRY(hipGetLastError());
  return DDMI_OK;
}

template <int VEC>
__global__ void __launch_bounds__(256) ddpm_step_kernel(const float* __restrict__ x, const float* __restrict__ mo,
                                                        const float* __restrict__ noise,
                                                        const uint64_t* __restrict__ seeds, unsigned tag,
                                                        const float* __restrict__ logit, StepCoefs k, int mode, int clip,
                                                        float* __restrict__ out, float* __restrict__ x0_out, int C, int HW,
                                                        int64_t nvec) {
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t quads_per_sample = (int64_t)C * HW / 4;  // only used on the VEC == 4 path
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) {
    int64_t e0 = i * VEC;
    float xv[VEC], mv[VEC], nv[VEC], ov[VEC], zv[VEC];
    if (VEC == 4) {
      float4 a = *reinterpret_cast<const float4*>(x + e0);
      float4 b = *reinterpret_cast<const float4*>(mo + e0);
      xv[0] = a.x; xv[1] = a.y; xv[2] = a.z; xv[3] = a.w;
      mv[0] = b.x; mv[1] = b.y; mv[2] = b.z; mv[3] = b.w;
      if (mode == STEP_DDIM_LAST) {
        nv[0] = nv[1] = nv[2] = nv[3] = 0.f;
      } else if (seeds != nullptr) {  // drawn in registers: same values as randn_fill(seeds, tag)
        int64_t bs = i / quads_per_sample;
        float z[4];
        normal4(seeds[bs], tag, (unsigned)(i - bs * quads_per_sample), z);
        nv[0] = z[0]; nv[1] = z[1]; nv[2] = z[2]; nv[3] = z[3];
      } else if (noise != nullptr) {
        float4 c = *reinterpret_cast<const float4*>(noise + e0);
        nv[0] = c.x; nv[1] = c.y; nv[2] = c.z; nv[3] = c.w;
      } else {
        nv[0] = nv[1] = nv[2] = nv[3] = 0.f;
      }
    } else {
      xv[0] = x[e0];
      mv[0] = mo[e0];
      nv[0] = (noise != nullptr && mode != STEP_DDIM_LAST) ? noise[e0] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      float eps = mv[j];
      if (logit != nullptr) {
        int c = (int)(((e0 + j) / HW) % C);
        float coeff = 1.0f / (1.0f + expf(-logit[c]));
        float mix = k.sqrt_one_minus_ac * xv[j];
        eps = (1.0f - coeff) * mix + coeff * mv[j];
      }
      float x0 = k.sqrt_recip_ac * xv[j] - k.sqrt_recipm1_ac * eps;
      if (clip) x0 = fminf(fmaxf(x0, -1.0f), 1.0f);
      float r;
      if (mode == STEP_ANCESTRAL) {
        float mean = k.coef1 * x0 + k.coef2 * xv[j];
        r = mean + k.noise_scale * nv[j];
      } else if (mode == STEP_DDIM) {
        r = x0 * k.sqrt_ac_next + k.c_dir * eps + k.sigma * nv[j];
      } else {
        r = x0;
      }
      ov[j] = r;
      zv[j] = x0;
    }
    if (VEC == 4) {
      *reinterpret_cast<float4*>(out + e0) = make_float4(ov[0], ov[1], ov[2], ov[3]);
      if (x0_out) *reinterpret_cast<float4*>(x0_out + e0) = make_float4(zv[0], zv[1], zv[2], zv[3]);
    } else {
      out[e0] = ov[0];
      if (x0_out) x0_out[e0] = zv[0];
    }
  }
}

// The sampling loop's form of the step (float4 vectors, < 2^31 quads, <= 1024 channels): the per-channel mixing
// coefficient sigmoid(logit[c]) is computed once per block into LDS (the general kernel evaluated expf and two 64-bit
// divisions per ELEMENT), sample / channel of a quad come from 32-bit divisions, and the noise is drawn with the
// hardware transcendentals (philox.h).  HBM-bound: 12 B per element with in-kernel noise (x, eps in; x out).
__global__ void __launch_bounds__(256) ddpm_step_fast_kernel(const float* __restrict__ x, const float* __restrict__ mo,
                                                             const float* __restrict__ noise, const uint64_t* __restrict__ seeds,
                                                             unsigned tag, const float* __restrict__ logit, StepCoefs k, int mode,
                                                             int clip, float* __restrict__ out, float* __restrict__ x0_out, unsigned C,
                                                             unsigned hw4, unsigned nquad) {
  __shared__ float coef[1024];
  for (unsigned c = threadIdx.x; c < C; c += 256) coef[c] = logit ? 1.0f / (1.0f + expf(-logit[c])) : 1.0f;
  __syncthreads();
  const unsigned qps = C * hw4;   // quads per sample
  const unsigned stride = gridDim.x * 256;
  auto one = [&](unsigned i, const float4& a, const float4& m, const float4& zn) {
    const unsigned b = i / qps, r = i - b * qps, c = r / hw4;
    float nv[4] = {0.f, 0.f, 0.f, 0.f};
    if (mode != STEP_DDIM_LAST) {
      if (seeds != nullptr) {
        normal4(seeds[b], tag, r, nv);
      } else if (noise != nullptr) {
        nv[0] = zn.x; nv[1] = zn.y; nv[2] = zn.z; nv[3] = zn.w;
      }
    }
    const float co = coef[c];
    const float xv[4] = {a.x, a.y, a.z, a.w}, mv[4] = {m.x, m.y, m.z, m.w};
    float ov[4], zv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float eps = mv[j];
      if (logit != nullptr) {   // same operation order as the general kernel: results are bit-identical
        const float mix = k.sqrt_one_minus_ac * xv[j];
        eps = (1.0f - co) * mix + co * mv[j];
      }
      float x0 = k.sqrt_recip_ac * xv[j] - k.sqrt_recipm1_ac * eps;
      if (clip) x0 = fminf(fmaxf(x0, -1.0f), 1.0f);
      float rr;
      if (mode == STEP_ANCESTRAL) rr = (k.coef1 * x0 + k.coef2 * xv[j]) + k.noise_scale * nv[j];
      else if (mode == STEP_DDIM) rr = x0 * k.sqrt_ac_next + k.c_dir * eps + k.sigma * nv[j];
      else rr = x0;
      ov[j] = rr;
      zv[j] = x0;
    }
    reinterpret_cast<float4*>(out)[i] = make_float4(ov[0], ov[1], ov[2], ov[3]);
    if (x0_out) reinterpret_cast<float4*>(x0_out)[i] = make_float4(zv[0], zv[1], zv[2], zv[3]);
  };
  const bool rd_noise = noise != nullptr && seeds == nullptr && mode != STEP_DDIM_LAST;
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  unsigned i = blockIdx.x * 256 + threadIdx.x;
  for (; i + stride < nquad; i += 2 * stride) {   // two quads per trip: four (six) 16-byte loads in flight per thread
    const float4 a0 = reinterpret_cast<const float4*>(x)[i], a1 = reinterpret_cast<const float4*>(x)[i + stride];
    const float4 m0 = reinterpret_cast<const float4*>(mo)[i], m1 = reinterpret_cast<const float4*>(mo)[i + stride];
    const float4 z0 = rd_noise ? reinterpret_cast<const float4*>(noise)[i] : zero4;
    const float4 z1 = rd_noise ? reinterpret_cast<const float4*>(noise)[i + stride] : zero4;
    one(i, a0, m0, z0);
    one(i + stride, a1, m1, z1);
  }
  if (i < nquad)
    one(i, reinterpret_cast<const float4*>(x)[i], reinterpret_cast<const float4*>(mo)[i],
        rd_noise ? reinterpret_cast<const float4*>(noise)[i] : zero4);
}

int launch_ddpm_step(const float* x, const float* model_out, const float* noise, const float* mixing_logit,
                     StepCoefs k, int mode, int clip, float* x_out, float* x0_out, int B, int C, int HW,
                     hipStream_t stream, const uint64_t* seeds, unsigned tag) {
  DDMI_REQUIRE(x && model_out && x_out, "ddpm_step: null tensor");
  DDMI_REQUIRE(mode >= 0 && mode <= 2, "ddpm_step: bad mode");
  DDMI_REQUIRE(B > 0 && C > 0 && HW > 0, "ddpm_step: bad shape");
  int64_t n = (int64_t)B * C * HW;
  bool vec4 = (HW % 4 == 0) && (((uintptr_t)x | (uintptr_t)model_out | (uintptr_t)noise | (uintptr_t)x_out |
                                 (uintptr_t)x0_out) % 16 == 0);
  DDMI_REQUIRE(seeds == nullptr || vec4, "ddpm_step: in-kernel noise needs HW % 4 == 0 and 16-byte aligned tensors");
  int64_t nvec = vec4 ? n / 4 : n;
  int blocks = (int)std::min<int64_t>(cdiv64(nvec, 256), 256 * 8);
  PROF_SCOPE("ddpm_step", 0, (noise && mode != STEP_DDIM_LAST ? 16.0 : 12.0) * (double)n, stream);
  if (vec4 && nvec < (int64_t)1 << 31 && C <= 1024)
    hipLaunchKernelGGL(ddpm_step_fast_kernel, dim3(blocks), dim3(256), 0, stream, x, model_out, noise, seeds, tag, mixing_logit, k,
                       mode, clip, x_out, x0_out, (unsigned)C, (unsigned)(HW / 4), (unsigned)nvec);
  else if (vec4)
    hipLaunchKernelGGL(ddpm_step_kernel<4>, dim3(blocks), dim3(256), 0, stream, x, model_out, noise, seeds, tag,
                       mixing_logit, k, mode, clip, x_out, x0_out, C, HW, nvec);
  else
    hipLaunchKernelGGL(ddpm_step_kernel<1>, dim3(blocks), dim3(256), 0, stream, x, model_out, noise, seeds, tag,
                       mixing_logit, k, mode, clip, x_out, x0_out, C, HW, nvec);
  HIP_TRY(hipGetLastError());
  return DDMI_OK;
}

// =====================================================================================================
// NCHW fp32 <-> NHWC T through a 32x32 LDS tile (both sides coalesced)
// =====================================================================================================
template <typename T>
__global__ void __launch_bounds__(256) nchw_to_nhwc_kernel(const float* __restrict__ src, T* __restrict__ dst, int C,
                                                           int HW, int ld, int off) {
  // source rows are `ld` long and the HW positions start at `off` (a token range of a [B, C, L] sequence); ld = HW,
  // off = 0 for a whole tensor
  __shared__ float tile[32][33];
  int b = blockIdx.z;
  int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const float* s = src + (int64_t)b * C * ld + off;
  T* d = dst + (int64_t)b * C * HW;
  for (int i = ty; i < 32; i += 8) {
    int c = c0 + i, p = p0 + tx;
    tile[i][tx] = (c < C && p < HW) ? s[(int64_t)c * ld + p] : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    int p = p0 + i, c = c0 + tx;
    if (p < HW && c < C) d[(int64_t)p * C + c] = from_f32<T>(tile[tx][i]);
  }
}

template <typename T>
__global__ void __launch_bounds__(256) nhwc_to_nchw_kernel(const T* __restrict__ src, float* __restrict__ dst, int C,
                                                           int HW, int ld, int off) {
  __shared__ float tile[32][33];
  int b = blockIdx.z;
  int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const T* s = src + (int64_t)b * C * HW;
  float* d = dst + (int64_t)b * C * ld + off;
  for (int i = ty; i < 32; i += 8) {
    int p = p0 + i, c = c0 + tx;
    tile[i][tx] = (c < C && p < HW) ? to_f32(s[(int64_t)p * C + c]) : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    int c = c0 + i, p = p0 + tx;
    if (p < HW && c < C) d[(int64_t)c * ld + p] = tile[tx][i];
  }
}

int launch_nchw_to_nhwc(const float* src, void* dst, int dt, int B, int C, int HW, hipStream_t s, int ld, int off) {
  if (ld <= 0) ld = HW;
  dim3 g(cdiv(HW, 32), cdiv(C, 32), B);
  if (dt == DT_F32)
    hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, g, dim3(256), 0, s, src, (float*)dst, C, HW, ld, off);
  else
    hipLaunchKernelGGL(nchw_to_nhwc_kernel<bf16_t>, g, dim3(256), 0, s, src, (bf16_t*)dst, C, HW, ld, off);
  HIP_TRY(hipGetLastError());
  return DDMI_OK;
}

int launch_nhwc_to_nchw(const void* src, int dt, float* dst, int B, int C, int HW, hipStream_t s, int ld, int off) {
  if (ld <= 0) ld = HW;
  dim3 g(cdiv(HW, 32), cdiv(C, 32), B);
  if (dt == DT_F32)
    hipLaunchKernelGGL(nhwc_to_nchw_kernel<float>, g, dim3(256), 0, s, (const float*)src, dst, C, HW, ld, off);
  else
    hipLaunchKernelGGL(nhwc_to_nchw_kernel<bf16_t>, g, dim3(256), 0, s, (const bf16_t*)src, dst, C, HW, ld, off);
  HIP_TRY(hipGetLastError());
  return DDMI_OK;
}

// =====================================================================================================
// channel concat of two NHWC tensors: out[p][0:Ca] = a[p], out[p][Ca:Ca+Cb] = b[p]   (16-byte vectors)
// =====================================================================================================
__global__ void __launch_bounds__(256) concat_kernel(const uint4* __restrict__ a, int va, const uint4* __restrict__ b,
                                                     int vb, uint4* __restrict__ out, int64_t nvec) {
  int vo = va + vb;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) {
    int64_t p = i / vo;
    int v = (int)(i - p * vo);
    out[i] = v < va ? a[p * va + v] : b[p * vb + (v - va)];
  }
}

int launch_concat(const void* a, int Ca, const void* b, int Cb, void* out, int dt, int64_t npix, hipStream_t s) {
  int per = dt == DT_F32 ? 4 : 8;
  DDMI_REQUIRE(Ca % per == 0 && Cb % per == 0, "concat: channels must be multiples of the 16-byte vector");
  int va = Ca / per, vb = Cb / per;
  int64_t nvec = npix * (va + vb);
  PROF_SCOPE("concat", 0, 32.0 * (double)nvec, s);
  int blocks = (int)std::min<int64_t>(cdiv64(nvec, 256), 256 * 16);
  hipLaunchKernelGGL(concat_kernel, dim3(blocks), dim3(256), 0, s, (const uint4*)a, va, (const uint4*)b, vb,
                     (uint4*)out, nvec);
  HIP_TRY(hipGetLastError());
  return DDMI_OK;
}

// =====================================================================================================
// conv weight packing: [Cout][Cin][KH][KW] fp32 -> [Cout][(kh*KW+kw)*Cin + ci] T   (done once at finalize)
// =====================================================================================================
template <typename T>
__global__ void pack_conv_weight_kernel(const float* __restrict__ w, T* __restrict__ out, int Cout, int Cin, int KH,
                                        int KW) {
  int64_t n = (int64_t)Cout * Cin * KH * KW;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int taps = KH * KW;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    int ci = (int)(i % Cin);
    int64_t r = i / Cin;
    int tap = (int)(r % taps);
    int co = (int)(r / taps);
    out[i] = from_f32<T>(w[((int64_t)co * Cin + ci) * taps + tap]);
  }
}

int launch_pack_conv_weight(const float* w, void* packed, int dt, int Cout, int Cin, int KH, int KW, hipStream_t s) {
  int64_t n = (int64_t)Cout * Cin * KH * KW;
  int blocks = (int)std::min<int64_t>(cdiv64(n, 256), 256 * 16);
  if (dt == DT_F32)
    hipLaunchKernelGGL(pack_conv_weight_kernel<float>, dim3(blocks), dim3(256), 0, s, w, (float*)packed, Cout, Cin, KH, KW);
  else
    hipLaunchKernelGGL(pack_conv_weight_kernel<bf16_t>, dim3(blocks), dim3(256), 0, s, w, (bf16_t*)packed, Cout, Cin, KH, KW);
  HIP_TRY(hipGetLastError());
  return DDMI_OK;
}

// =====================================================================================================
// bf16x3 numerics (DDMI_PARITY_BF16X3): an fp32 GEMM as THREE bf16 MFMA products with fp32 accumulation.
//   x = xh + xl,  w = wh + wl   (xh = bf16(x), xl = bf16(x - xh): 16 mantissa bits between them)
//   x . w  ~  xh.wh + xl.wh + xh.wl          (the dropped xl.wl is 2^-16 of the product)
// realised with the UNCHANGED bf16 convolution kernels by widening K: activations [xh | xl | xh] (3 C channels per pixel),
// weights [wh | wh | wl] per tap.  1/16 of the fp32-MFMA cost per product becomes 3/16.
// =====================================================================================================
__global__ void __launch_bounds__(256) split_x3_kernel(const float* __restrict__ x, bf16_t* __restrict__ out, int64_t npix, int C) {
  const int cv = C / 4;                                   // float4 vectors per pixel
  const int64_t n = npix * cv, stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int64_t p = i / cv;
    const int c = (int)(i - p * cv) * 4;
    const float4 v = *reinterpret_cast<const float4*>(x + p * C + c);
    const float f[4] = {v.x, v.y, v.z, v.w};
    bf16x4 hi, lo;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      hi[e] = (bf16_t)f[e];
      lo[e] = (bf16_t)(f[e] - (float)hi[e]);
    }
    bf16_t* o = out + p * 3 * C + c;
    *reinterpret_cast<bf16x4*>(o) = hi;
    *reinterpret_cast<bf16x4*>(o + C) = lo;
    *reinterpret_cast<bf16x4*>(o + 2 * C) = hi;
  }
}
int launch_split_x3(const float* x, void* out, int64_t npix, int C, hipStream_t s) {
  DDMI_REQUIRE(C % 4 == 0, "split_x3: C must be a multiple of 4");
  const int64_t n = npix * (C / 4);
  const int blocks = (int)std::min<int64_t>(cdiv64(n, 256), 256 * 32);
  hipLaunchKernelGGL(split_x3_kernel, dim3(blocks), dim3(256), 0, s, x, (bf16_t*)out, npix, C);
  HIP_TRY(hipGetLastError());
  return DDMI_OK;
}
// [Cout][Cin][KH][KW] fp32 -> [Cout][tap][wh(Cin) | wh(Cin) | wl(Cin)] bf16
__global__ void pack_conv_weight_x3_kernel(const float* __restrict__ w, bf16_t* __restrict__ out, int Cout, int Cin, int taps) {
  const int64_t n = (int64_t)Cout * Cin * taps, stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int ci = (int)(i % Cin);
    const int64_t r = i / Cin;
    const int tap = (int)(r % taps), co = (int)(r / taps);
    const float v = w[((int64_t)co * Cin + ci) * taps + tap];
    const bf16_t hi = (bf16_t)v, lo = (bf16_t)(v - (float)hi);
    bf16_t* o = out + (r * 3) * Cin + ci;
    o[0] = hi; o[Cin] = hi; o[2 * Cin] = lo;
  }
}
int launch_pack_conv_weight_x3(const float* w, void* packed, int Cout, int Cin, int KH, int KW, hipStream_t s) {
  const int64_t n = (int64_t)Cout * Cin * KH * KW;
  const int blocks = (int)std::min<int64_t>(cdiv64(n, 256), 256 * 16);
  hipLaunchKernelGGL(pack_conv_weight_x3_kernel, dim3(blocks), dim3(256), 0, s, w, (bf16_t*)packed, Cout, Cin, KH * KW);
  HIP_TRY(hipGetLastError());
  return DDMI_OK;
}

// parity form of (nearest x2 upsample -> 3x3 conv), see k_conv.hip conv_out_row: packed [Cout][9][Cin] bf16 ->
// [4][Cout][2*2][Cin] bf16, parity p = 2 dy + dx, tap (a, b) of the 2x2 source neighbourhood = the sum of the 3x3 taps
// that land on that source pixel: rows dy = 0: a0 <- {kh 0}, a1 <- {1, 2};  dy = 1: a0 <- {0, 1}, a1 <- {2} (same for columns)
// SRC = float: w9 is the raw [Cout][Cin][3][3] fp32 tensor (the sums are rounded to bf16 ONCE, so the parity form carries
// less weight-rounding error than the plain form); SRC = bf16_t: the packed [Cout][9][Cin] weights
template <typename SRC>
__global__ void conv_pack_up4_kernel(const SRC* __restrict__ w9, bf16_t* __restrict__ out, int Cout, int Cin) {
  const int64_t n = (int64_t)4 * Cout * 4 * Cin;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int ci = (int)(i % Cin);
    int64_t r = i / Cin;
    const int tap = (int)(r & 3);
    r >>= 2;
    const int co = (int)(r % Cout), par = (int)(r / Cout);
    const int dy = par >> 1, dx = par & 1, ta = tap >> 1, tb = tap & 1;
    const int kh0 = dy == 0 ? (ta == 0 ? 0 : 1) : (ta == 0 ? 0 : 2), kh1 = dy == 0 ? (ta == 0 ? 0 : 2) : (ta == 0 ? 1 : 2);
    const int kw0 = dx == 0 ? (tb == 0 ? 0 : 1) : (tb == 0 ? 0 : 2), kw1 = dx == 0 ? (tb == 0 ? 0 : 2) : (tb == 0 ? 1 : 2);
    float acc = 0.f;
    for (int kh = kh0; kh <= kh1; ++kh)
      for (int kw = kw0; kw <= kw1; ++kw)
        acc += sizeof(SRC) == 4 ? (float)w9[((int64_t)co * Cin + ci) * 9 + kh * 3 + kw]
                                : (float)w9[((int64_t)co * 9 + kh * 3 + kw) * Cin + ci];
    out[i] = (bf16_t)acc;
  }
}
int launch_pack_conv_up4(const void* packed9, void* out, int Cout, int Cin, hipStream_t s) {
  const int64_t n = (int64_t)16 * Cout * Cin;
  const int blocks = (int)std::min<int64_t>(cdiv64(n, 256), 256 * 16);
  hipLaunchKernelGGL(conv_pack_up4_kernel<bf16_t>, dim3(blocks), dim3(256), 0, s, (const bf16_t*)packed9, (bf16_t*)out, Cout, Cin);
  HIP_TRY(hipGetLastError());
  return DDMI_OK;
}
int launch_pack_conv_up4_f32(const float* w_oihw, void* out, int Cout, int Cin, hipStream_t s) {
  const int64_t n = (int64_t)16 * Cout * Cin;
  const int blocks = (int)std::min<int64_t>(cdiv64(n, 256), 256 * 16);
  hipLaunchKernelGGL(conv_pack_up4_kernel<float>, dim3(blocks), dim3(256), 0, s, w_oihw, (bf16_t*)out, Cout, Cin);
  HIP_TRY(hipGetLastError());
  return DDMI_OK;
}

template <typename T>
__global__ void cast_kernel(const float* __restrict__ src, T* __restrict__ dst, int64_t n) {
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = from_f32<T>(src[i]);
}

int launch_cast_f32(const float* src, void* dst, int dt, int64_t n, hipStream_t s) {
  int blocks = (int)std::min<int64_t>(cdiv64(n, 256), 256 * 16);
  if (dt == DT_F32)
    hipLaunchKernelGGL(cast_kernel<float>, dim3(blocks), dim3(256), 0, s, src, (float*)dst, n);
  else
    hipLaunchKernelGGL(cast_kernel<bf16_t>, dim3(blocks), dim3(256), 0, s, src, (bf16_t*)dst, n);
  HIP_TRY(hipGetLastError());
  return DDMI_OK;
}

// =====================================================================================================
// timestep embedding (util.py:150-170): out[b] = [cos(t*f_k) | sin(t*f_k)], f_k precomputed on the host
// =====================================================================================================
__global__ void timestep_embedding_kernel(const int64_t* __restrict__ t, const float* __restrict__ freqs,
                                          float* __restrict__ out, int dim) {
  int b = blockIdx.x;
  int half = dim / 2;
  float tv = (float)t[b];
  for (int i = threadIdx.x; i < half; i += blockDim.x) {
    float a = tv * freqs[i];
    out[(int64_t)b * dim + i] = cosf(a);
    out[(int64_t)b * dim + half + i] = sinf(a);
  }
  if ((dim & 1) && threadIdx.x == 0) out[(int64_t)b * dim + dim - 1] = 0.f;
}

int launch_timestep_embedding(const int64_t* t, const float* freqs, float* out, int B, int dim, hipStream_t s) {
  hipLaunchKernelGGL(timestep_embedding_kernel, dim3(B), dim3(128), 0, s, t, freqs, out, dim);
  HIP_TRY(hipGetLastError());
  return DDMI_OK;
}

// =====================================================================================================
// small dense layer, M = batch rows (<= 64): one wave per output feature streams W[n][:] once (coalesced,
// weight-read bound) and keeps the B activations rows in LDS.  fp32 throughout.
// =====================================================================================================
#define LIN_MAXB 8
__global__ void __launch_bounds__(256) linear_small_kernel(const float* __restrict__ x, int x_stride,
                                                           const float* __restrict__ W, const float* __restrict__ bias,
                                                           float* __restrict__ y, int y_stride, int B, int K, int N,
                                                           int silu_in) {
  extern __shared__ float xs[];  // [bb][K]
  int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int b0 = 0; b0 < B; b0 += LIN_MAXB) {
    int nb = min(LIN_MAXB, B - b0);
    __syncthreads();
    for (int i = threadIdx.x; i < nb * K; i += blockDim.x) {
      int bb = i / K, kk = i - bb * K;
      float v = x[(int64_t)(b0 + bb) * x_stride + kk];
      xs[i] = silu_in ? silu_precise(v) : v;
    }
    __syncthreads();
    for (int n = blockIdx.x * 4 + wave; n < N; n += gridDim.x * 4) {
      float acc[LIN_MAXB];
#pragma unroll
      for (int j = 0; j < LIN_MAXB; ++j) acc[j] = 0.f;
      const float* wr = W + (int64_t)n * K;
      for (int kk = lane; kk < K; kk += 64) {
        float wv = wr[kk];
#pragma unroll
        for (int j = 0; j < LIN_MAXB; ++j)
          if (j < nb) acc[j] += wv * xs[j * K + kk];
      }
#pragma unroll
      for (int j = 0; j < LIN_MAXB; ++j) {
        float v = acc[j];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
        if (lane == 0 && j < nb) y[(int64_t)(b0 + j) * y_stride + n] = v + (bias ? bias[n] : 0.f);
      }
    }
  }
}

int launch_linear_small(const float* x, int x_stride, const float* W, const float* bias, float* y, int y_stride, int B,
                        int K, int N, int silu_in, hipStream_t s) {
  DDMI_REQUIRE(K * LIN_MAXB * 4 <= 64 * 1024, "linear_small: K too large for LDS staging");
  int blocks = std::min(cdiv(N, 4), 2048);
  size_t lds = (size_t)std::min(B, LIN_MAXB) * K * sizeof(float);
  hipLaunchKernelGGL(linear_small_kernel, dim3(blocks), dim3(256), lds, s, x, x_stride, W, bias, y, y_stride, B, K, N,
                     silu_in);
  HIP_TRY(hipGetLastError());
  return DDMI_OK;
}

// =====================================================================================================
// nearest x2 upsample (only used by the diagnostic op API; the networks fuse it into the conv gather)
// =====================================================================================================
__global__ void upsample2x_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, int B, int H, int W, int vc) {
  int64_t n = (int64_t)B * 2 * H * 2 * W * vc;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    int v = (int)(i % vc);
    int64_t p = i / vc;
    int ow = (int)(p % (2 * W));
    int64_t q = p / (2 * W);
    int oh = (int)(q % (2 * H));
    int b = (int)(q / (2 * H));
    dst[i] = src[(((int64_t)b * H + (oh >> 1)) * W + (ow >> 1)) * vc + v];
  }
}

int launch_upsample2x(const void* src, void* dst, int dt, int B, int H, int W, int C, hipStream_t s) {
  int per = dt == DT_F32 ? 4 : 8;
  DDMI_REQUIRE(C % per == 0, "upsample2x: channels must be a multiple of the 16-byte vector");
  int64_t n = (int64_t)B * 4 * H * W * (C / per);
  int blocks = (int)std::min<int64_t>(cdiv64(n, 256), 256 * 16);
  hipLaunchKernelGGL(upsample2x_kernel, dim3(blocks), dim3(256), 0, s, (const uint4*)src, (uint4*)dst, B, H, W, C / per);
  HIP_TRY(hipGetLastError());
  return DDMI_OK;
}

// =====================================================================================================
// tri-plane channel concat / split: planes [P][Bn][HW][C] <-> [Bn][HW][P*C]  (autoencoder_unet.py:1512, 1517-1519)
// =====================================================================================================
__global__ void __launch_bounds__(256) planes_channels_kernel(uint4* __restrict__ planes, uint4* __restrict__ chans,
                                                              int64_t npix, int vc, int P, int reverse) {
  int64_t n = npix * vc * P;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    int v = (int)(i % vc);
    int64_t r = i / vc;
    int pl = (int)(r % P);
    int64_t pix = r / P;  // b * HW + hw
    int64_t pi = ((int64_t)pl * npix + pix) * vc + v;
    if (reverse) planes[pi] = chans[i];
    else chans[i] = planes[pi];
  }
}

int launch_planes_channels(const void* planes, void* chans, int dt, int Bn, int HW, int C, int P, int reverse, hipStream_t s) {
  int per = dt == DT_F32 ? 4 : 8;
  DDMI_REQUIRE(C % per == 0, "planes_channels: channels must be a multiple of the 16-byte vector");
  int64_t npix = (int64_t)Bn * HW;
  int64_t n = npix * (C / per) * P;
  int blocks = (int)std::min<int64_t>(cdiv64(n, 256), 256 * 16);
  hipLaunchKernelGGL(planes_channels_kernel, dim3(blocks), dim3(256), 0, s, (uint4*)planes, (uint4*)chans, npix, C / per, P, reverse);
  HIP_TRY(hipGetLastError());
  return DDMI_OK;
}

// =====================================================================================================
// fused_bias_act: the reference's one compiled operator (models/d2c_vae/op/fused_bias_act.cpp:4-20, kernel
// fused_bias_act_kernel.cu:19-49: StyleGAN2's bias + activation + gain with its gradient modes), as a standalone op so
// that models/d2c_vae/op imports on ROCm without nvcc.  On the generation path it is fused into the MLP kernels; this
// entry point serves the reference's own module code (FusedLeakyReLU and its double-backward).
//   z = x + bias[(i / step_b) % size_b]          (bias optional)
//   act 1 (linear): grad 0, 1 -> z;  grad 2 -> 0
//   act 3 (lrelu) : grad 0 -> z > 0 ? z : alpha z;   grad 1 -> ref > 0 ? z : alpha z  (ref = the forward's output);   grad 2 -> 0
//   out = y * scale.     Any other (act, grad) pair behaves as (1, 0), as the reference's `default:` label does.
// Arithmetic in the tensor's own type rounded per operation like the reference's scalar_t code: fp32 in fp32; bf16 / fp16
// through fp32 with one rounding at the end (the reference computes half in half: differences are one ulp of the narrow type).
// =====================================================================================================
template <typename T>
__global__ void __launch_bounds__(256) fused_bias_act_kernel(const T* __restrict__ x, const T* __restrict__ bias, const T* __restrict__ ref,
                                                             T* __restrict__ y, int64_t n, int64_t step_b, int size_b, int mode, float alpha,
                                                             float scale) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    float z = (float)x[i];
    if (bias) z += (float)bias[(i / step_b) % size_b];
    float v;
    switch (mode) {
      case 12: case 32: v = 0.f; break;
      case 30: v = z > 0.f ? z : z * alpha; break;
      case 31: v = (ref ? (float)ref[i] : 0.f) > 0.f ? z : z * alpha; break;
      default: v = z; break;
    }
    y[i] = (T)(v * scale);
  }
}

extern "C" int ddmi_fused_bias_act(const void* x, const void* bias, const void* refer, void* y, int64_t size_x, int64_t step_b, int size_b,
                                   int act, int grad, float alpha, float scale, int dtype, void* stream) {
  DDMI_REQUIRE(x && y && size_x >= 0 && step_b >= 1, "fused_bias_act: bad arguments");
  DDMI_REQUIRE(bias == nullptr || size_b >= 1, "fused_bias_act: bias without a size");
  if (size_x == 0) return DDMI_OK;
  const int blocks = (int)std::min<int64_t>(cdiv64(size_x, 256), 256 * 32);
  const int mode = act * 10 + grad;
  hipStream_t s = (hipStream_t)stream;
  switch (dtype) {
    case 0: hipLaunchKernelGGL(fused_bias_act_kernel<float>, dim3(blocks), dim3(256), 0, s, (const float*)x, (const float*)bias, (const float*)refer,
                               (float*)y, size_x, step_b, size_b, mode, alpha, scale); break;
    case 1: hipLaunchKernelGGL(fused_bias_act_kernel<bf16_t>, dim3(blocks), dim3(256), 0, s, (const bf16_t*)x, (const bf16_t*)bias, (const bf16_t*)refer,
                               (bf16_t*)y, size_x, step_b, size_b, mode, alpha, scale); break;
    case 2: hipLaunchKernelGGL(fused_bias_act_kernel<_Float16>, dim3(blocks), dim3(256), 0, s, (const _Float16*)x, (const _Float16*)bias,
                               (const _Float16*)refer, (_Float16*)y, size_x, step_b, size_b, mode, alpha, scale); break;
    default: return ddmi_fail(DDMI_ERR_INVALID, "fused_bias_act: dtype must be 0 (fp32), 1 (bf16) or 2 (fp16)");
  }
  HIP_TRY(hipGetLastError());
  return DDMI_OK;
}
