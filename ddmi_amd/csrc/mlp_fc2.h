// Second-generation pieces of the fused ResnetBlockFC point-MLP kernels (bf16: k_mlp3d.hip, k_mlpv.hip) - the design
// the image kernel (k_mlp2d.hip, mlp2d_v2_kernel) measured its way to:
//   * one workgroup = 8 waves = a tile of P points that never leaves the CU; wave w owns output channels [32 w, 32 w + 32)
//     of every 256-wide layer for ALL P points, so each weight byte is fetched once per workgroup (L2 -> registers, MFMA
//     fragment order: one wave load = 1 KiB contiguous), while the activations are shared through LDS;
//   * LDS holds the RAW block input / output (bf16) in two [P][256+16] buffers + the raw features of the current scale:
//     relu() for fc_0 is applied on the fragment after the LDS read (v_pk_max_i16 on the bf16 bit patterns - the same
//     bits as rounding relu(x)), the shortcut / identity path continues on the accumulators of fc_1's pass, so neither a
//     rectified copy of x nor 64 more accumulators exist;
//   * paired fragments: a lane's two accumulator fragments are eight consecutive channels -> 16-byte epilogue stores.
#pragma once
#include "common.h"

// timing-only ablation builds (tools/build_variant.sh NAME k_mlp3d.hip -DFC2_ABL=<bits>; WRONG results on purpose): bit 0 no relu on the
// fragments, bit 1 no weight fetch inside the k-loop, bit 2 no activation reads inside the k-loop, bit 3 (k_mlp3d.hip) no feature gather
// after the first.  (Dropping the epilogue's LDS stores is not a usable ablation: the compiler then removes the passes that feed them.)
#ifndef FC2_ABL
#define FC2_ABL 0
#endif
#define FC2_XP 272   // activation row pitch in elements (544 B): (row * 136 + 4 lg) mod 64 conflict-free for ds_read_b128

typedef __attribute__((ext_vector_type(8))) short fc2_s16x8;

// element offset of the 1 KiB fragment block (16-row group g, k-step ks) of a matrix at `off` with KS k-steps
__device__ __forceinline__ int fc2_wfrag(int off, int KS, int g, int ks) { return off + ((g * KS + ks) << 9); }

template <int NF>
__device__ __forceinline__ void fc2_wload(const bf16_t* __restrict__ W, int off, int KS, int g0, int ks, int lane, bf16x8 (&dst)[NF]) {
#pragma unroll
  for (int nf = 0; nf < NF; ++nf) dst[nf] = *reinterpret_cast<const bf16x8*>(W + fc2_wfrag(off, KS, g0 + nf, ks) + 8 * lane);
}

// relu on eight packed bf16: negative values (sign bit set, incl. -0) are negative as int16 -> max with 0
__device__ __forceinline__ bf16x8 fc2_relu(bf16x8 v) {
  fc2_s16x8 s = __builtin_bit_cast(fc2_s16x8, v);
  s = __builtin_elementwise_max(s, (fc2_s16x8)(0));
  return __builtin_bit_cast(bf16x8, s);
}

// acc[mf][nf] += W[16 (g0 + nf) + r][k] * act(IN[16 mf + r][k]), k over KX columns of X (pitch FC2_XP) then KF columns of F
// (pitch FP); act = relu when RELU.  wq[0] must hold k-step 0 (fc2_wload, issued before the previous epilogue / barrier);
// weights and activations are fetched one k-step ahead.  Two k-steps per trip of a rolled loop (static ring slots).
template <int MF, int NF, int KX, int KF, int FP, bool RELU>
__device__ __forceinline__ void fc2_pass(const bf16_t* __restrict__ X, const bf16_t* __restrict__ F, const bf16_t* __restrict__ W,
                                         int off, int g0, int lane, bf16x8 (&wq)[2][NF], f32x4 (&acc)[MF][NF]) {
  constexpr int KS = (KX + KF) / 32;
  static_assert(KS % 2 == 0 && (KX % 32) == 0, "two k-steps per trip");
  const int lr = lane & 15, lg = lane >> 4;
  const bf16_t* xrow = X + lr * FC2_XP + 8 * lg;
  const bf16_t* frow = F + lr * FP + 8 * lg - KX;
  auto xload = [&](int ks, bf16x8(&xf)[MF]) {
    const int k0 = ks * 32;
    const bf16_t* p = (k0 >= KX) ? frow + k0 : xrow + k0;
    const int pitch = (k0 >= KX) ? 16 * FP : 16 * FC2_XP;
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) {
      xf[mf] = *reinterpret_cast<const bf16x8*>(p + mf * pitch);
      if (RELU && !(FC2_ABL & 1)) xf[mf] = fc2_relu(xf[mf]);
    }
  };
  auto mma = [&](const bf16x8(&w)[NF], const bf16x8(&x)[MF]) {
#pragma unroll
    for (int mf = 0; mf < MF; ++mf)
#pragma unroll
      for (int nf = 0; nf < NF; ++nf) acc[mf][nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[nf], x[mf], acc[mf][nf], 0, 0, 0);
  };
  bf16x8 xq[2][MF];
  xload(0, xq[0]);
#pragma unroll 1
  for (int ks = 0; ks < KS; ks += 2) {
    if (!(FC2_ABL & 2)) fc2_wload<NF>(W, off, KS, g0, ks + 1, lane, wq[1]);
    if (!(FC2_ABL & 4)) xload(ks + 1, xq[1]);
    mma(wq[0], xq[0]);
    if (ks + 2 < KS) {
      if (!(FC2_ABL & 2)) fc2_wload<NF>(W, off, KS, g0, ks + 2, lane, wq[0]);
      if (!(FC2_ABL & 4)) xload(ks + 2, xq[0]);
    }
    mma((FC2_ABL & 2) ? wq[0] : wq[1], (FC2_ABL & 4) ? xq[0] : xq[1]);
  }
}

// The same pass for a wave that owns FOUR channel fragments (NF = 4: k_mlp3d.hip's four-wave form): the activation fragments are
// fetched half a k-step ahead in two halves of MF / 2 (32 registers instead of 64 - with 128 accumulators the whole-k-step form
// leaves the register allocator shuffling accumulators through AGPRs inside the loop).  Same MFMA order per accumulator.
template <int MF, int NF, int KX, int KF, int FP, bool RELU>
__device__ __forceinline__ void fc2_pass_halves(const bf16_t* __restrict__ X, const bf16_t* __restrict__ F, const bf16_t* __restrict__ W,
                                                int off, int g0, int lane, bf16x8 (&wq)[2][NF], f32x4 (&acc)[MF][NF]) {
  constexpr int KS = (KX + KF) / 32, MH = MF / 2;
  static_assert(KS % 2 == 0 && (KX % 32) == 0 && MF % 2 == 0, "two k-steps per trip");
  const int lr = lane & 15, lg = lane >> 4;
  const bf16_t* xrow = X + lr * FC2_XP + 8 * lg;
  const bf16_t* frow = F + lr * FP + 8 * lg - KX;
  auto xload = [&](int ks, int half, bf16x8(&xf)[MH]) {
    const int k0 = ks * 32;
    const bf16_t* p = (k0 >= KX) ? frow + k0 : xrow + k0;
    const int pitch = (k0 >= KX) ? 16 * FP : 16 * FC2_XP;
#pragma unroll
    for (int m = 0; m < MH; ++m) {
      xf[m] = *reinterpret_cast<const bf16x8*>(p + (half * MH + m) * pitch);
      if (RELU) xf[m] = fc2_relu(xf[m]);
    }
  };
  auto mma = [&](const bf16x8(&w)[NF], int half, const bf16x8(&x)[MH]) {
#pragma unroll
    for (int m = 0; m < MH; ++m)
#pragma unroll
      for (int nf = 0; nf < NF; ++nf)
        acc[half * MH + m][nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[nf], x[m], acc[half * MH + m][nf], 0, 0, 0);
  };
  bf16x8 xa[MH], xb[MH];
  xload(0, 0, xa);
#pragma unroll 1
  for (int ks = 0; ks < KS; ks += 2) {
    fc2_wload<NF>(W, off, KS, g0, ks + 1, lane, wq[1]);
    xload(ks, 1, xb);
    mma(wq[0], 0, xa);
    xload(ks + 1, 0, xa);
    mma(wq[0], 1, xb);
    if (ks + 2 < KS) fc2_wload<NF>(W, off, KS, g0, ks + 2, lane, wq[0]);
    xload(ks + 1, 1, xb);
    mma(wq[1], 0, xa);
    if (ks + 2 < KS) xload(ks + 2, 0, xa);
    mma(wq[1], 1, xb);
  }
}

// host: one row-major [rows][K] fp32 matrix -> fragment order [rows/16][K/32][64 lanes][8]; rows % 32 == 0 matrices are
// PAIRED: row 4q + e of group 2p + h is output channel 32 p + 8 q + 4 h + e (a lane's two fragments = 8 consecutive channels)
static inline void fc2_pack_matrix(const float* src, float* dst, int rows, int K) {
  const int KS = K / 32;
  const bool paired = rows % 32 == 0;
  for (int g = 0; g < rows / 16; ++g)
    for (int ks = 0; ks < KS; ++ks)
      for (int l = 0; l < 64; ++l) {
        const int r = l & 15;
        int row = 16 * g + r;
        if (paired) row = 32 * (g >> 1) + 8 * (r >> 2) + 4 * (g & 1) + (r & 3);
        for (int e = 0; e < 8; ++e) dst[(((g * KS + ks) * 64 + l) << 3) + e] = src[row * K + 32 * ks + 8 * (l >> 4) + e];
      }
}
