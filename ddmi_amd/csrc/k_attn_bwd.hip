// Backward of QKVAttentionLegacy (openaimodel.py:397-422, head dim 32) for the LDM training step - second generation.
//
// What round 2's pair of kernels (k_train.hip: attn_bwd_dq_mfma_kernel / attn_bwd_dkv_mfma_kernel) spent: 9.7 % of the AFHQ
// training step at 6x the forward's time - a third pass over the scores only to rebuild the row log-sum-exp, 16 queries (keys)
// per wave against a 64-row tile staged behind two barriers, and the transposed operands written to LDS by eight 2-byte stores
// per vector.  Here
//   * the forward kernel hands over the row log-sum-exp (AttnArgs::lse), so each kernel makes ONE pass over the scores;
//   * a wave owns NF = 2 fragments of 16 queries (keys): every fragment read from LDS feeds two MFMAs;
//   * the K / V (Q / dO) tiles stay ROW-major in LDS under the forward's swizzle, which serves both the 16-byte row reads of
//     the S / dP products and ds_read_b64_tr_b16 (the hardware 4 x 16 transpose) for the K^T / Q^T / dO^T operands of the
//     accumulation products - no transposed image is ever written;
//   * the next tile travels global -> registers under the current tile's MFMAs and is parked in the other LDS buffer: one
//     barrier per tile;
//   * -D_q (dq kernel: a lane constant; dkv kernel: a row vector read from LDS) is the INITIAL VALUE of the dP accumulators,
//     so dS = P (dP - D) is one multiply; the softmax scale on dS is applied once, to the finished dQ / dK;
//   * workgroups are dealt to the XCDs by sample, like the forward's (all tiles and heads of a sample read through one L2).
// The loops are VALU-bound like the forward's (d = 32: one exp2 + 3 VALU slots per score against 0.75 MFMA).
// Two kernels, no atomics: dQ, dK and dV are bit-reproducible.
#include "kernels.h"
#include "prof.h"
#include <math.h>
#include <cstdlib>

namespace {

constexpr int BD = 32;            // head dim
constexpr int BT = 64;            // rows of a streamed tile
constexpr int BRB = BD * 2;       // bytes per tile row
constexpr int BTILE = BT * BRB;   // 4 KiB

// 16-byte slot swizzle of a [64][32] bf16 tile (k_attention.hip att_swz<32>): conflict-free for the 16-byte row reads, and - the
// mask of rows 4..7 / 12..15 differing from rows 0..3 / 8..11 in bit 1 - for the transposing reads of an 8-row half as well
__device__ __forceinline__ int bw_mask(int row) { return (0x78 >> (((row >> 2) & 3) * 2)) & 3; }

typedef short bw_s16x4 __attribute__((ext_vector_type(4)));
typedef short bw_s16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) bw_s16x4* bw_lds4;
__device__ __forceinline__ bf16x8 bw_tr_frag(const unsigned char* lo, const unsigned char* hi) {
  const bw_s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((bw_lds4)lo);
  const bw_s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((bw_lds4)hi);
  const bw_s16x8 both = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, both);   // whole-vector reinterpretation (see k_attention.hip on the element-wise trap)
}

struct AttnBwdK {
  const bf16_t* qkv;    // [B][T][heads * 96], per head [q | k | v]
  const bf16_t* out;    // [B][T][heads * 32]
  const bf16_t* dout;   // [B][T][heads * 32]
  bf16_t* dqkv;         // [B][T][heads * 96]
  const float* lse;     // [B][heads][T] natural-log row log-sum-exp of the scaled scores (forward)
  float* dsum;          // [B][heads][T] D_q = dO_q . O_q (written by the dq kernel, read by the dkv kernel)
  int B, T, heads, by_sample;
  float scale;          // softmax scale of the scores: 1 / sqrt(32), or 1 / sqrt(d) for a head of d < 32 channels zero-padded to 32
};

// linear workgroup id -> (sample, head, tile): XCD x (= id % 8) owns the samples b = x (mod 8) when by_sample, else plain order
__device__ __forceinline__ bool bw_decode(const AttnBwdK& a, int tiles, int* b, int* h, int* tile) {
  const int L = blockIdx.x;
  if (a.by_sample) {
    const int slot = L >> 3, per_sample = tiles * a.heads;
    const int bl = slot / per_sample, rem = slot - bl * per_sample;
    *h = rem / tiles;
    *tile = rem - *h * tiles;
    *b = bl * 8 + (L & 7);
  } else {
    const int bh = L / tiles;
    *tile = L - bh * tiles;
    *b = bh / a.heads;
    *h = bh - *b * a.heads;
  }
  return *b < a.B;
}

// ---------------------------------------------------------------------------------------------------------------------
// dQ (and D): the QUERY sits on the MFMA lane.  Workgroup = 4 waves = 64 NF queries of one (sample, head); keys / values stream.
//   S^T = K Q^T, dP^T = V dO^T - D   (A = key rows from LDS, B = the query's q / dO channels in registers)
//   dS^T = P^T o dP^T                (per lane: 16 NF scores of its own queries per tile)
//   dQ^T += K^T dS^T                 (A = K^T by transposing reads, B = dS straight from the accumulator registers)
// ---------------------------------------------------------------------------------------------------------------------
template <int NF>
__global__ void __launch_bounds__(256) attn_bwd2_dq_kernel(AttnBwdK a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char bw_smem[];
  unsigned char* const ks = bw_smem;               // [2][64][64 B]
  unsigned char* const vs = bw_smem + 2 * BTILE;   // [2][64][64 B]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lg = lane >> 4;
  const int T = a.T, ld = a.heads * 3 * BD, ldo = a.heads * BD;
  int b, h, q_tile;
  if (!bw_decode(a, T / (BT * NF), &b, &h, &q_tile)) return;
  const bf16_t* base = a.qkv + (int64_t)b * T * ld + h * 3 * BD;
  const float scale = a.scale;                            // 1 / sqrt(32) = (d^-1/4)^2 unless the head is a zero-padded narrower one
  const float sl2 = scale * 1.4426950408889634f;          // scores in the exp2 domain

  bf16x8 qb[NF], gob[NF];
  float negL2[NF], negD[NF];
#pragma unroll
  for (int f = 0; f < NF; ++f) {
    const int q = q_tile * (BT * NF) + (wave * NF + f) * 16 + lr;
    qb[f] = *reinterpret_cast<const bf16x8*>(base + (int64_t)q * ld + 8 * lg);
    gob[f] = *reinterpret_cast<const bf16x8*>(a.dout + ((int64_t)b * T + q) * ldo + h * BD + 8 * lg);
    const bf16x8 of = *reinterpret_cast<const bf16x8*>(a.out + ((int64_t)b * T + q) * ldo + h * BD + 8 * lg);
    float part = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) part += (float)gob[f][e] * (float)of[e];
    part += __shfl_xor(part, 16);
    part += __shfl_xor(part, 32);
    negD[f] = -part;
    const int64_t si = ((int64_t)b * a.heads + h) * T + q;
    negL2[f] = -a.lse[si] * 1.4426950408889634f;
    if (lg == 0) a.dsum[si] = part;
  }

  // staging: thread owns one 16-byte vector (key = tid >> 2, slot = tid & 3) of K and of V per tile
  const int skey = tid >> 2, sslot = tid & 3;
  const int sdst = skey * BRB + ((sslot ^ bw_mask(skey)) << 4);
  bf16x8 kreg, vreg;
  auto load_tile = [&](int kt) {
    const bf16_t* p = base + (int64_t)(kt * BT + skey) * ld + sslot * 8;
    kreg = *reinterpret_cast<const bf16x8*>(p + BD);
    vreg = *reinterpret_cast<const bf16x8*>(p + 2 * BD);
  };
  auto store_tile = [&](int buf) {
    *reinterpret_cast<bf16x8*>(ks + buf * BTILE + sdst) = kreg;
    *reinterpret_cast<bf16x8*>(vs + buf * BTILE + sdst) = vreg;
  };
  // transposed-read lane constants: lane 4q+p of a 16-lane group addresses key row 4 lg + q, 4 channels at 4p
  const int tq = lr >> 2, tp = lr & 3;
  const int tbase = (4 * lg + tq) * BRB + (tp & 1) * 8;
  const int tmask = bw_mask(4 * lg);
  // row-read lane constants: fragment f row 16 f + lr, slot lg
  const int rmask = bw_mask(lr);   // (16 f does not change (row >> 2) & 3)
  const int rbase = lr * BRB + ((lg ^ rmask) << 4);

  f32x4 dq[NF][2];
#pragma unroll
  for (int f = 0; f < NF; ++f) dq[f][0] = dq[f][1] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int ntiles = T / BT;
  load_tile(0);
  store_tile(0);
  if (ntiles > 1) load_tile(1);
  __syncthreads();
  for (int kt = 0; kt < ntiles; ++kt) {
    const int buf = kt & 1;
    const unsigned char* kb = ks + buf * BTILE;
    const unsigned char* vb = vs + buf * BTILE;
    f32x4 sc[NF][4], dp[NF][4];
#pragma unroll
    for (int f4 = 0; f4 < 4; ++f4) {
      const bf16x8 kf = *reinterpret_cast<const bf16x8*>(kb + rbase + f4 * 16 * BRB);
      const bf16x8 vf = *reinterpret_cast<const bf16x8*>(vb + rbase + f4 * 16 * BRB);
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        sc[f][f4] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qb[f], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        dp[f][f4] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, gob[f], f32x4{negD[f], negD[f], negD[f], negD[f]}, 0, 0, 0);
      }
    }
    if (kt + 1 < ntiles) {   // the other buffer was last read before the barrier that ended tile kt-1
      store_tile(buf ^ 1);
      if (kt + 2 < ntiles) load_tile(kt + 2);
    }
    bf16x8 dsb[NF][2];
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
      for (int f4 = 0; f4 < 4; ++f4)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[f][f4][j], sl2, negL2[f]));
          dsb[f][f4 >> 1][(f4 & 1) * 4 + j] = (bf16_t)(p * dp[f][f4][j]);
        }
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int df = 0; df < 2; ++df) {
        const int coff = ((2 * df + (tp >> 1)) ^ tmask) << 4;
        const bf16x8 kt_f = bw_tr_frag(kb + tbase + (32 * u) * BRB + coff, kb + tbase + (32 * u + 16) * BRB + coff);
#pragma unroll
        for (int f = 0; f < NF; ++f) dq[f][df] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kt_f, dsb[f][u], dq[f][df], 0, 0, 0);
      }
    __syncthreads();   // tile kt consumed by everyone; tile kt+1 visible
  }
#pragma unroll
  for (int f = 0; f < NF; ++f) {
    const int q = q_tile * (BT * NF) + (wave * NF + f) * 16 + lr;
    bf16_t* dst = a.dqkv + ((int64_t)b * T + q) * ld + h * 3 * BD;
#pragma unroll
    for (int df = 0; df < 2; ++df) {
      bf16x4 v;
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = (bf16_t)(dq[f][df][j] * scale);
      *reinterpret_cast<bf16x4*>(dst + 16 * df + 4 * lg) = v;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// dK, dV: the KEY sits on the MFMA lane.  Workgroup = 64 NF keys of one (sample, head); queries / dO rows stream with their
// -L (base 2) and -D.
//   S = Q K^T, dP = dO V^T - D       (A = query rows from LDS, B = the key's k / v channels in registers): a lane holds
//                                    rows (queries) 16 f + 4 lg + j of ITS key's column, so L and D are row vectors
//   dV^T += dO^T P, dK^T += Q^T dS   (A by transposing reads of the same tiles, B = P / dS from the accumulator registers)
// ---------------------------------------------------------------------------------------------------------------------
template <int NF>
__global__ void __launch_bounds__(256) attn_bwd2_dkv_kernel(AttnBwdK a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char bw_smem[];
  unsigned char* const qs = bw_smem;                      // [2][64][64 B]
  unsigned char* const gs = bw_smem + 2 * BTILE;          // [2][64][64 B]
  float* const ls = reinterpret_cast<float*>(bw_smem + 4 * BTILE);   // [2][64]  -L in the exp2 domain
  float* const dsm = ls + 2 * BT;                                    // [2][64]  -D
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lg = lane >> 4;
  const int T = a.T, ld = a.heads * 3 * BD, ldo = a.heads * BD;
  int b, h, k_tile;
  if (!bw_decode(a, T / (BT * NF), &b, &h, &k_tile)) return;
  const bf16_t* base = a.qkv + (int64_t)b * T * ld + h * 3 * BD;
  const bf16_t* gbase = a.dout + (int64_t)b * T * ldo + h * BD;
  const float* Lr = a.lse + ((int64_t)b * a.heads + h) * T;
  const float* Dr = a.dsum + ((int64_t)b * a.heads + h) * T;
  const float scale = a.scale, sl2 = scale * 1.4426950408889634f;

  bf16x8 kb_[NF], vb_[NF];
#pragma unroll
  for (int f = 0; f < NF; ++f) {
    const int key = k_tile * (BT * NF) + (wave * NF + f) * 16 + lr;
    kb_[f] = *reinterpret_cast<const bf16x8*>(base + (int64_t)key * ld + BD + 8 * lg);
    vb_[f] = *reinterpret_cast<const bf16x8*>(base + (int64_t)key * ld + 2 * BD + 8 * lg);
  }
  const int srow = tid >> 2, sslot = tid & 3;
  const int sdst = srow * BRB + ((sslot ^ bw_mask(srow)) << 4);
  bf16x8 qreg, greg;
  float sreg = 0.f;   // threads 0..63: -L of query tid; 64..127: -D of query tid - 64
  auto load_tile = [&](int qt) {
    qreg = *reinterpret_cast<const bf16x8*>(base + (int64_t)(qt * BT + srow) * ld + sslot * 8);
    greg = *reinterpret_cast<const bf16x8*>(gbase + (int64_t)(qt * BT + srow) * ldo + sslot * 8);
    if (tid < BT) sreg = -Lr[qt * BT + tid] * 1.4426950408889634f;
    else if (tid < 2 * BT) sreg = -Dr[qt * BT + tid - BT];
  };
  auto store_tile = [&](int buf) {
    *reinterpret_cast<bf16x8*>(qs + buf * BTILE + sdst) = qreg;
    *reinterpret_cast<bf16x8*>(gs + buf * BTILE + sdst) = greg;
    if (tid < BT) ls[buf * BT + tid] = sreg;
    else if (tid < 2 * BT) dsm[buf * BT + tid - BT] = sreg;
  };
  const int tq = lr >> 2, tp = lr & 3;
  const int tbase = (4 * lg + tq) * BRB + (tp & 1) * 8;
  const int tmask = bw_mask(4 * lg);
  const int rbase = lr * BRB + ((lg ^ bw_mask(lr)) << 4);

  f32x4 dk[NF][2], dv[NF][2];
#pragma unroll
  for (int f = 0; f < NF; ++f) dk[f][0] = dk[f][1] = dv[f][0] = dv[f][1] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int ntiles = T / BT;
  load_tile(0);
  store_tile(0);
  if (ntiles > 1) load_tile(1);
  __syncthreads();
  for (int qt = 0; qt < ntiles; ++qt) {
    const int buf = qt & 1;
    const unsigned char* qb = qs + buf * BTILE;
    const unsigned char* gb = gs + buf * BTILE;
    f32x4 sc[NF][4], dp[NF][4], nl[4];
#pragma unroll
    for (int f4 = 0; f4 < 4; ++f4) {
      const bf16x8 qf = *reinterpret_cast<const bf16x8*>(qb + rbase + f4 * 16 * BRB);
      const bf16x8 gf = *reinterpret_cast<const bf16x8*>(gb + rbase + f4 * 16 * BRB);
      nl[f4] = *reinterpret_cast<const f32x4*>(ls + buf * BT + 16 * f4 + 4 * lg);
      const f32x4 nd = *reinterpret_cast<const f32x4*>(dsm + buf * BT + 16 * f4 + 4 * lg);
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        sc[f][f4] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf, kb_[f], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);   // rows = queries, column = key
        dp[f][f4] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gf, vb_[f], nd, 0, 0, 0);
      }
    }
    if (qt + 1 < ntiles) {
      store_tile(buf ^ 1);
      if (qt + 2 < ntiles) load_tile(qt + 2);
    }
    bf16x8 pb[NF][2], dsb[NF][2];
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
      for (int f4 = 0; f4 < 4; ++f4)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[f][f4][j], sl2, nl[f4][j]));
          pb[f][f4 >> 1][(f4 & 1) * 4 + j] = (bf16_t)p;
          dsb[f][f4 >> 1][(f4 & 1) * 4 + j] = (bf16_t)(p * dp[f][f4][j]);
        }
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int df = 0; df < 2; ++df) {
        const int coff = ((2 * df + (tp >> 1)) ^ tmask) << 4;
        const bf16x8 gt = bw_tr_frag(gb + tbase + (32 * u) * BRB + coff, gb + tbase + (32 * u + 16) * BRB + coff);
        const bf16x8 qt_f = bw_tr_frag(qb + tbase + (32 * u) * BRB + coff, qb + tbase + (32 * u + 16) * BRB + coff);
#pragma unroll
        for (int f = 0; f < NF; ++f) {
          dv[f][df] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gt, pb[f][u], dv[f][df], 0, 0, 0);
          dk[f][df] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qt_f, dsb[f][u], dk[f][df], 0, 0, 0);
        }
      }
    __syncthreads();
  }
#pragma unroll
  for (int f = 0; f < NF; ++f) {
    const int key = k_tile * (BT * NF) + (wave * NF + f) * 16 + lr;
    bf16_t* dst = a.dqkv + ((int64_t)b * T + key) * ld + h * 3 * BD;
#pragma unroll
    for (int df = 0; df < 2; ++df) {
      bf16x4 k4, v4;
#pragma unroll
      for (int j = 0; j < 4; ++j) { k4[j] = (bf16_t)(dk[f][df][j] * scale); v4[j] = (bf16_t)dv[f][df][j]; }
      *reinterpret_cast<bf16x4*>(dst + BD + 16 * df + 4 * lg) = k4;
      *reinterpret_cast<bf16x4*>(dst + 2 * BD + 16 * df + 4 * lg) = v4;
    }
  }
}

}  // namespace

// lse [B][heads][T] from the forward (launch_attention_legacy_lse); dsum [B][heads][T] scratch.  bf16, d = 32, T a multiple of 64.
int launch_attention_legacy_backward_lse(const void* qkv, const void* out, const void* dout, void* dqkv, const float* lse, float* dsum,
                                         int B, int T, int heads, hipStream_t s, float scale) {
  DDMI_REQUIRE(qkv && out && dout && dqkv && lse && dsum, "attention backward: null argument");
  DDMI_REQUIRE(T % BT == 0 && B > 0 && heads > 0, "attention backward: the token count must be a multiple of 64");
  AttnBwdK a{(const bf16_t*)qkv, (const bf16_t*)out, (const bf16_t*)dout, (bf16_t*)dqkv, lse, dsum, B, T, heads, 0,
             scale > 0.f ? scale : 0.17677669529663687f};
  a.by_sample = (B % 8 == 0 || B >= 64) ? 1 : 0;
  const int nf = T % (2 * BT) == 0 ? 2 : 1;
  const int tiles = T / (BT * nf);
  const dim3 grid((unsigned)(tiles * heads * (a.by_sample ? cdiv(B, 8) * 8 : B)));
  PROF_SCOPE("attention_legacy_bwd", 2.0 * 5.0 * B * heads * (double)T * T * BD, 0, s);
  constexpr int lds_dq = 4 * BTILE, lds_dkv = 4 * BTILE + 4 * BT * (int)sizeof(float);
  if (nf == 2) {
    hipLaunchKernelGGL(attn_bwd2_dq_kernel<2>, grid, dim3(256), lds_dq, s, a);
    hipLaunchKernelGGL(attn_bwd2_dkv_kernel<2>, grid, dim3(256), lds_dkv, s, a);
  } else {
    hipLaunchKernelGGL(attn_bwd2_dq_kernel<1>, grid, dim3(256), lds_dq, s, a);
    hipLaunchKernelGGL(attn_bwd2_dkv_kernel<1>, grid, dim3(256), lds_dkv, s, a);
  }
  HIP_TRY(hipGetLastError());
  return DDMI_OK;
}
