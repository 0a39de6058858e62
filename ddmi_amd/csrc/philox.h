// Counter-based per-sample noise streams shared by the sampler-step kernel and the fused image MLP.
#pragma once
#include "common.h"

// ---- counter-based noise: Philox4x32-10 (Salmon et al., SC'11) keyed by the SAMPLE's 64-bit seed ----------------------
// normal(seed, tag, e) for element e of a sample depends on nothing else, so a chain's noise is the same whatever batch
// or GPU it runs in (the sharding invariant of ddmi_amd/dist.py), and the sampler step can draw it in registers instead
// of reading a [B, C, HW] tensor that 256 per-sample generator launches filled.  Counter = (e / 4, tag, 0x44444d49, 0);
// `tag` = step index (0xFFFFFFFF for x_T).  Four uniforms -> four normals by Box-Muller.
__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1,
                                              unsigned (&out)[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const unsigned hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    const unsigned n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
__device__ __forceinline__ void normal4(uint64_t seed, unsigned tag, unsigned quad, float (&z)[4]) {
  unsigned r[4];
  philox4x32_10(quad, tag, 0x44444d49u, 0u, (unsigned)seed, (unsigned)(seed >> 32), r);
  // u in (0, 1]: 24-bit mantissas, never 0 so the logarithm is finite
  const float u0 = ((r[0] >> 8) + 1) * (1.0f / 16777216.0f), u1 = (r[1] >> 8) * (1.0f / 16777216.0f);
  const float u2 = ((r[2] >> 8) + 1) * (1.0f / 16777216.0f), u3 = (r[3] >> 8) * (1.0f / 16777216.0f);
  // Box-Muller on the hardware transcendentals: v_log_f32 (log2, 1 ulp), v_sqrt_f32, v_sin_f32 / v_cos_f32 (argument in
  // revolutions - exactly what the uniform is).  The libm forms (logf, sincospif: ~300 instructions per quad) made the
  // sampler-step kernel VALU-bound at 2.9 TB/s; every consumer of the streams goes through this one function.
  const float ra = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u0));   // -2 ln u = -2 ln2 log2 u
  const float rb = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u2));
  z[0] = ra * __builtin_amdgcn_cosf(u1); z[1] = ra * __builtin_amdgcn_sinf(u1);
  z[2] = rb * __builtin_amdgcn_cosf(u3); z[3] = rb * __builtin_amdgcn_sinf(u3);
}

// stream tags: step n of a sampling loop uses tag n, x_T 0xFFFFFFFF; NoiseInjection draw of StyledConv c (0..11, call
// order, blocks.py:293-295) of the image MLP uses tag PHILOX_TAG_INJECT + c, element = query point index of the sample
#define PHILOX_TAG_INJECT 0xFFFE0000u
