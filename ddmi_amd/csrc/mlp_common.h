// Pieces shared by the fused point-MLP kernels (k_mlp3d.hip, k_mlpv.hip): one workgroup = 4 waves = a tile of P
// points that never leaves the CU; weights stream L2 -> registers, activations live in LDS.
#pragma once
#include "common.h"

// acc[mf][nf] += W[nbase + 16 nf + r][koff + k] * SRC[16 (m_first + mf) + r][k],  k in [0, K)
template <typename T, int MF, int NF, int K>
__device__ __forceinline__ void m3_gemm(const T* __restrict__ src, int pitch, const T* __restrict__ W, int ldw, int koff,
                                        int nbase, int m_first, f32x4 (&acc)[MF][NF]) {
  const int lane = threadIdx.x & 63, lr = lane & 15, lg = lane >> 4;
#pragma unroll 2
  for (int k0 = 0; k0 < K; k0 += 32) {
    if constexpr (sizeof(T) == 2) {
      bf16x8 wf[NF], xf[MF];
#pragma unroll
      for (int nf = 0; nf < NF; ++nf)
        wf[nf] = *reinterpret_cast<const bf16x8*>(W + (int64_t)(nbase + 16 * nf + lr) * ldw + koff + k0 + 8 * lg);
#pragma unroll
      for (int mf = 0; mf < MF; ++mf)
        xf[mf] = *reinterpret_cast<const bf16x8*>(src + (16 * (m_first + mf) + lr) * pitch + k0 + 8 * lg);
#pragma unroll
      for (int mf = 0; mf < MF; ++mf)
#pragma unroll
        for (int nf = 0; nf < NF; ++nf)
          acc[mf][nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nf], xf[mf], acc[mf][nf], 0, 0, 0);
    } else {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        f32x4 wf[NF], xf[MF];
#pragma unroll
        for (int nf = 0; nf < NF; ++nf)
          wf[nf] = *reinterpret_cast<const f32x4*>(W + (int64_t)(nbase + 16 * nf + lr) * ldw + koff + k0 + 16 * h + 4 * lg);
#pragma unroll
        for (int mf = 0; mf < MF; ++mf)
          xf[mf] = *reinterpret_cast<const f32x4*>(src + (16 * (m_first + mf) + lr) * pitch + k0 + 16 * h + 4 * lg);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int mf = 0; mf < MF; ++mf)
#pragma unroll
            for (int nf = 0; nf < NF; ++nf)
              acc[mf][nf] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[nf][e], xf[mf][e], acc[mf][nf], 0, 0, 0);
      }
    }
  }
}

template <typename T> __device__ __forceinline__ void st4(T* dst, const float* v) {
  if constexpr (sizeof(T) == 2) {
    bf16x4 q;
#pragma unroll
    for (int e = 0; e < 4; ++e) q[e] = (bf16_t)v[e];
    *reinterpret_cast<bf16x4*>(dst) = q;
  } else {
    *reinterpret_cast<f32x4*>(dst) = f32x4{v[0], v[1], v[2], v[3]};
  }
}

