// Fused multi-head self-attention (online softmax, keys / values streamed through LDS), token-major activations.
//
// One kernel serves every attention on the path:
//   * LDM UNet AttentionBlock + QKVAttentionLegacy (openaimodel.py:279-325, 397-422): per-head interleaved
//     [q | k | v] channels, scale d^-1/4 on q and k = d^-1/2 on the scores
//   * the video path's MemoryEfficientAttnBlock / ...1D / ...1D_expand (autoencoder_unet.py:273-335, 404-534; xformers
//     semantics softmax(q k^T / sqrt(d)) v): blocked [q heads | k heads | v heads] channels, and the token sequence
//     of one sample is the CONCATENATION of up to three row ranges (xy | xt | yt planes, openaimodel.py:1148-1157)
//     that live in different tensors - `seg_len` / the three base pointers describe that row map, nothing is copied.
//
// One workgroup = 4 waves = 64 queries of one (sample, head[, slice of the value channels]).  Everything is arranged
// so the QUERY sits on the MFMA lane (lane & 15):
//   S^T = K . Q^T     A-operand = K tile rows (keys) from LDS, B-operand = Q rows (registers, loaded once)
//                     -> lane holds scores of its query for keys 16f + 4*(lane>>4) + j  (row-max / row-sum need only
//                        in-register reductions plus two xor-shuffles over lane>>4)
//   O^T = V^T . P^T   A-operand = V^T rows (channels) from a transposed LDS image, B-operand = P straight from the
//                     score registers (key order inside an MFMA k-step is permuted identically on both operands)
//                     -> lane holds out channels 16*df + 4*(lane>>4) + j of its query: 8/16-byte vector stores.
// Template: D = head dim rounded up to {32, 64, 128, 256, 512} (channels >= d are zero-filled), DV = value / output
// channels handled by one workgroup (blockIdx.z walks the slices; S is recomputed per slice), KT = keys per LDS tile.
// bf16 path: v_mfma_f32_16x16x32_bf16, fp32 softmax/accumulate.  fp32 (parity) path: v_mfma_f32_16x16x4_f32.
//
// The kernels of this file, by who runs where (launch_attention, bf16 with q pre-scaled unless noted):
//   attention_mh_kernel      the generic form above: fp32 handles, odd head dims / token counts, d > 128 off the streaming shapes
//   attention_v2_kernel      d <= 128, 128 (QF = 2) or 64 queries per workgroup, K / V tiles re-staged through LDS; the only one with the
//                            log-sum-exp output (training forward) and the eager maximum (q not pre-scaled)
//   attention_res_kernel     d = 16 / 32, 1024 <= N <= 128 KB / 4d: K / V of a (sample, head) resident in LDS (round 4)
//   attention_res16_kernel   d = 16, N a multiple of 1024: the same on the 32 x 32 x 16 MFMA, denominator as a row of V^T P^T (round 4)
//   attention_stream_kernel  d = 32 .. 512, N >= 2048: 128 QF queries per workgroup, LDS-DMA ring (round 4)
//   attention_x3_kernel      bf16x3 parity handles: fp32 tensors, three bf16 MFMA products per matrix product (round 4)
#include "kernels.h"
#include "prof.h"
#include <math.h>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#define ATT_QB 64   // queries per workgroup
// A/B builds (tools/build_variant.sh): static priority of one half of the streaming kernel's waves (1: waves 4-7, 2: waves 0-3);
// 0 drops the priority hand-over of the d = 16 resident kernel's two waves per SIMD
#ifndef ATT_STREAM_PRIO
#define ATT_STREAM_PRIO 0
#endif
#ifndef ATT_RES16_PRIO
#define ATT_RES16_PRIO 1
#endif

// 16-byte slot swizzle of the bf16 K tile, conflict-free for the four ds_read_b128 lane groups at every row size:
// 64-byte rows (D = 32): slot ^ {0,2,3,1}[(row>>2)&3];  128-byte rows: slot ^ ((row>>1)&7);  >= 256-byte rows: every
// row starts on bank 0, so the low four slot bits are XORed with row & 15.
template <int D> __device__ __forceinline__ int att_swz(int row, int slot) {
  if constexpr (D == 32) {
    int r = (row >> 2) & 3;
    return slot ^ ((0x78 >> (r * 2)) & 3);
  } else if constexpr (D == 64) {
    return slot ^ ((row >> 1) & 7);
  } else {
    return slot ^ (row & 15);
  }
}

// row r (0 .. sum seg_len) of sample b -> (segment, row inside that segment's [B * seg_len][ld] matrix)
__device__ __forceinline__ int64_t att_row(const AttnArgs& a, int b, int n, int* seg) {
  int s = 0;
  if (n >= a.seg_len[0]) {
    n -= a.seg_len[0];
    s = 1;
    if (n >= a.seg_len[1]) {
      n -= a.seg_len[1];
      s = 2;
    }
  }
  *seg = s;
  return (int64_t)b * a.seg_len[s] + n;
}

// VN consecutive channels c0.. of one token row, zero beyond the head dim d; element-wise when the row is not
// 16-byte aligned (toy head dims 2 / 4)
template <typename T, int VN, typename V>
__device__ __forceinline__ V att_load(const T* p, int c0, int d, bool vec_ok) {
  V v;
  if (vec_ok && c0 + VN <= d) return *reinterpret_cast<const V*>(p + c0);
#pragma unroll
  for (int e = 0; e < VN; ++e) v[e] = c0 + e < d ? p[c0 + e] : (T)0;
  return v;
}

template <typename T, int D, int DV, int KT>
__global__ void __launch_bounds__(256) attention_mh_kernel(AttnArgs a) {
  constexpr bool IS_BF16 = sizeof(T) == 2;
  constexpr int KROWB = IS_BF16 ? D * 2 : D * 4 + 16;  // fp32 rows padded by one slot
  constexpr int VROWB = (KT + 4) * (int)sizeof(T);    // V^T row: KT keys + 4 pad
  constexpr int NF = KT / 16;                          // score fragments per tile
  constexpr int NDF = DV / 16;                         // output fragments
  extern __shared__ __attribute__((aligned(16))) unsigned char att_smem[];
  unsigned char* const ks = att_smem;
  unsigned char* const vs = att_smem + KT * KROWB;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lg = lane >> 4;
  const int bh = blockIdx.y, b = bh / a.heads, h = bh - b * a.heads;
  const int dv0 = blockIdx.z * DV;
  const int d = a.d;
  const int N = a.seg_len[0] + a.seg_len[1] + a.seg_len[2];
  const bool vec_ok = a.vec_ok != 0;
  const int q_idx = blockIdx.x * ATT_QB + wave * 16 + lr;
  const bool q_ok = q_idx < N;
  const float sscale = IS_BF16 ? a.scale * 1.44269504088896341f : a.scale;  // bf16 path exponentiates in base 2

  // ---- Q fragment(s) in registers --------------------------------------------------------------
  bf16x8 qb[IS_BF16 ? D / 32 : 1];
  f32x4 qf[IS_BF16 ? 1 : D / 16];
  {
    int seg = 0;
    int64_t r = att_row(a, b, q_ok ? q_idx : 0, &seg);
    const T* qp = (const T*)a.qkv[seg] + r * a.ld_qkv + a.q_off + h * a.hs_qkv;
    if constexpr (IS_BF16) {
#pragma unroll
      for (int kk = 0; kk < D / 32; ++kk)
        qb[kk] = q_ok && kk * 32 + 8 * lg < d ? att_load<T, 8, bf16x8>(qp, kk * 32 + 8 * lg, d, vec_ok) : bf16x8{};
    } else {
#pragma unroll
      for (int kk = 0; kk < D / 16; ++kk)
        qf[kk] = q_ok && kk * 16 + 4 * lg < d ? att_load<T, 4, f32x4>(qp, kk * 16 + 4 * lg, d, vec_ok) : f32x4{0, 0, 0, 0};
    }
  }

  float m_run = -INFINITY, l_run = 0.f;
  f32x4 o[NDF];
#pragma unroll
  for (int df = 0; df < NDF; ++df) o[df] = f32x4{0, 0, 0, 0};

  const int ntiles = (N + KT - 1) / KT;
  for (int kt = 0; kt < ntiles; ++kt) {
    __syncthreads();  // previous tile fully consumed
    // ---- stage K tile [key][D] and V^T tile [DV][key] ----------------------------------------------
    {
      constexpr int VN = IS_BF16 ? 8 : 4;
      constexpr int KSL = D / VN, VSL = DV / VN;  // 16-byte slots per key row
      for (int idx = tid; idx < KT * KSL; idx += 256) {
        int key = idx / KSL, slot = idx - key * KSL;
        int kg = kt * KT + key;
        int seg = 0;
        int64_t r = att_row(a, b, kg < N ? kg : 0, &seg);
        const T* kp = (const T*)a.qkv[seg] + r * a.ld_qkv + a.k_off + h * a.hs_qkv;
        if constexpr (IS_BF16) {
          bf16x8 kv = kg < N && slot * 8 < d ? att_load<T, 8, bf16x8>(kp, slot * 8, d, vec_ok) : bf16x8{};
          *reinterpret_cast<bf16x8*>(ks + key * KROWB + att_swz<D>(key, slot) * 16) = kv;
        } else {
          f32x4 kv = kg < N && slot * 4 < d ? att_load<T, 4, f32x4>(kp, slot * 4, d, vec_ok) : f32x4{0, 0, 0, 0};
          *reinterpret_cast<f32x4*>(ks + key * KROWB + slot * 16) = kv;
        }
      }
      for (int idx = tid; idx < KT * VSL; idx += 256) {
        int key = idx / VSL, slot = idx - key * VSL;
        int kg = kt * KT + key;
        int seg = 0;
        int64_t r = att_row(a, b, kg < N ? kg : 0, &seg);
        const T* vp = (const T*)a.qkv[seg] + r * a.ld_qkv + a.v_off + h * a.hs_qkv;
        int c0 = dv0 + slot * VN;
        if constexpr (IS_BF16) {
          bf16x8 vv = kg < N && c0 < d ? att_load<T, 8, bf16x8>(vp, c0, d, vec_ok) : bf16x8{};
#pragma unroll
          for (int e = 0; e < 8; ++e) *reinterpret_cast<bf16_t*>(vs + (8 * slot + e) * VROWB + key * 2) = vv[e];
        } else {
          f32x4 vv = kg < N && c0 < d ? att_load<T, 4, f32x4>(vp, c0, d, vec_ok) : f32x4{0, 0, 0, 0};
#pragma unroll
          for (int e = 0; e < 4; ++e) *reinterpret_cast<float*>(vs + (4 * slot + e) * VROWB + key * 4) = vv[e];
        }
      }
    }
    __syncthreads();

    // ---- S^T = K . Q^T : NF fragments of 16 keys ---------------------------------------------------
    f32x4 sc[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      sc[f] = f32x4{0, 0, 0, 0};
      int row = 16 * f + lr;
      if constexpr (IS_BF16) {
#pragma unroll
        for (int kk = 0; kk < D / 32; ++kk) {
          bf16x8 kf = *reinterpret_cast<const bf16x8*>(ks + row * KROWB + att_swz<D>(row, 4 * kk + lg) * 16);
          sc[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qb[kk], sc[f], 0, 0, 0);
        }
      } else {
#pragma unroll
        for (int kk = 0; kk < D / 16; ++kk) {
          f32x4 kf = *reinterpret_cast<const f32x4*>(ks + row * KROWB + (4 * kk + lg) * 16);
#pragma unroll
          for (int e = 0; e < 4; ++e) sc[f] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[e], qf[kk][e], sc[f], 0, 0, 0);
        }
      }
    }
    // ---- online softmax for this lane's query ------------------------------------------------------
    float tmax = -INFINITY;
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        int kg = kt * KT + 16 * f + 4 * lg + j;
        float v = kg < N ? sc[f][j] * sscale : -INFINITY;
        sc[f][j] = v;
        tmax = fmaxf(tmax, v);
      }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 16));
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
    float m_new = fmaxf(m_run, tmax);
    float alpha = IS_BF16 ? __builtin_amdgcn_exp2f(m_run - m_new) : expf(m_run - m_new);
    float psum = 0.f;
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float p = IS_BF16 ? __builtin_amdgcn_exp2f(sc[f][j] - m_new) : expf(sc[f][j] - m_new);
        sc[f][j] = p;
        psum += p;
      }
    psum += __shfl_xor(psum, 16);
    psum += __shfl_xor(psum, 32);
    l_run = l_run * alpha + psum;
    m_run = m_new;
#pragma unroll
    for (int df = 0; df < NDF; ++df) o[df] *= alpha;

    // ---- O^T += V^T . P^T ---------------------------------------------------------------------------
    if constexpr (IS_BF16) {
#pragma unroll
      for (int u = 0; u < KT / 32; ++u) {  // k-step u covers keys 32u .. 32u+31 (fragments 2u, 2u+1)
        bf16x8 pf;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          pf[j] = (bf16_t)sc[2 * u][j];
          pf[4 + j] = (bf16_t)sc[2 * u + 1][j];
        }
#pragma unroll
        for (int df = 0; df < NDF; ++df) {
          const unsigned char* vr = vs + (16 * df + lr) * VROWB;
          bf16x4 lo = *reinterpret_cast<const bf16x4*>(vr + (32 * u + 4 * lg) * 2);
          bf16x4 hi = *reinterpret_cast<const bf16x4*>(vr + (32 * u + 16 + 4 * lg) * 2);
          bf16x8 vf;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            vf[j] = lo[j];
            vf[4 + j] = hi[j];
          }
          o[df] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf, o[df], 0, 0, 0);
        }
      }
    } else {
#pragma unroll
      for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int df = 0; df < NDF; ++df) {
          f32x4 vf = *reinterpret_cast<const f32x4*>(vs + (16 * df + lr) * VROWB + (16 * f + 4 * lg) * 4);
#pragma unroll
          for (int j = 0; j < 4; ++j) o[df] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[j], sc[f][j], o[df], 0, 0, 0);
        }
    }
  }

  if (q_ok) {
    float inv = 1.0f / l_run;
    int seg = 0;
    int64_t r = att_row(a, b, q_idx, &seg);
    T* op = (T*)a.out[seg] + r * a.ld_out + h * a.hs_out;
#pragma unroll
    for (int df = 0; df < NDF; ++df) {
      int c0 = dv0 + 16 * df + 4 * lg;
      if (c0 >= d) continue;
      f32x4 v = o[df] * inv;
      if (vec_ok && c0 + 4 <= d) {
        if constexpr (IS_BF16) {
          bf16x4 t;
#pragma unroll
          for (int j = 0; j < 4; ++j) t[j] = (bf16_t)v[j];
          *reinterpret_cast<bf16x4*>(op + c0) = t;
        } else {
          *reinterpret_cast<f32x4*>(op + c0) = v;
        }
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (c0 + j < d) op[c0 + j] = (T)v[j];
      }
    }
  }
}

// =====================================================================================================
// bf16 production variant for head dims <= 128 (every attention of the image UNet and of the video UNet, and the
// decoder's 64- and 128-channel levels).  Same query-on-lane formulation as above, plus:
//   * QF fragments of 16 queries per wave (QF = 2: 128 queries per workgroup) - every K fragment read from LDS feeds
//     QF MFMAs, halving LDS traffic per FLOP;
//   * V stays ROW-major in LDS ([key][channel], written with one 16-byte store per thread like K) and the V^T
//     A-operand is fetched with ds_read_b64_tr_b16 (hardware 4 x 16 transpose per 16-lane group) - the generic
//     kernel's eight 2-byte transposed stores per vector are gone;
//   * the next K/V tile is prefetched global -> registers while the current one is consumed, and written to the other
//     LDS buffer afterwards: one barrier per tile, HBM/L2 latency off the critical path;
//   * the softmax scale is folded into the exp2 argument (one FMA per score).
// V swizzle (16-byte chunk index ^ 2*f(key)) keeps the tr reads of a 32-lane half (8 keys x 32 bytes) on distinct banks.
// =====================================================================================================
template <int D> __device__ __forceinline__ int att_vswz(int key) {
  if constexpr (D == 32) return 2 * ((key >> 2) & 1);
  else if constexpr (D == 64) return 2 * ((key >> 1) & 3);
  else return 2 * (key & 7);
}

// LAZY (q pre-scaled: the softmax scale x log2 e is folded into the q rows of the qkv projection at finalize, so a score IS the
// base-2 exponent): the loop is VALU-bound (d = 32: ~7 VALU slots per score against 0.6 MFMA), so the per-score work is cut to the
// exponential itself.  The score accumulators start at -m_ref, the query's REFERENCE maximum, so the MFMA result is already
// s - m_ref and p = exp2(that) needs no multiply and no subtract.  m_ref is only moved when some score of the wave exceeds it by
// more than 2^6 (wave-uniform branch, rare after the first tiles: p <= 64 stays well inside bf16 / fp32 range; the final division
// by the denominator - accumulated with the same p - makes any reference equally exact), and the cross-lane maximum, the
// rescale of the output accumulators and the extra subtract are paid only then.
template <int D, int QF, bool SEG3, bool LAZY = false>
__global__ void __launch_bounds__(256) attention_v2_kernel(AttnArgs a) {
  using T = bf16_t;
  constexpr int KT = 64, RB = D * 2, TILE = KT * RB;
  constexpr int SL = D / 8;                 // 16-byte slots per row
  constexpr int NV = KT * SL / 256;         // staging vectors per thread per operand
  constexpr int NKK = D / 32, NDF = D / 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char att_smem[];
  unsigned char* const ks = att_smem;               // [2][KT][RB]
  unsigned char* const vs = att_smem + 2 * TILE;    // [2][KT][RB]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lg = lane >> 4;
  const int d = a.d;
  const int N = a.seg_len[0] + a.seg_len[1] + a.seg_len[2];
  // Workgroups are dealt round-robin to the 8 XCDs (linear id % 8), each with its own L2.  With the query tile as the fastest
  // grid index the tiles of one (sample, head) landed on 8 different XCDs and every L2 fetched that head's K / V (PMC, round 2:
  // 6.5x the algorithmic reads at 1024 tokens).  Here XCD x owns the samples b = x (mod 8) whole: all query tiles of a head
  // and all heads of a sample (whose q | k | v blocks interleave inside the same token rows) read through ONE L2.
  int q_tile, b, h;
  {
    const int tiles = (N + 64 * QF - 1) / (64 * QF);
    const int L = blockIdx.x, slot = L >> 3;
    if (a.xcd_map == 0) {          // XCD x owns samples b = x (mod 8)
      const int per_sample = tiles * a.heads;
      const int bl = slot / per_sample, rem = slot - bl * per_sample;
      h = rem / tiles;
      q_tile = rem - h * tiles;
      b = bl * 8 + (L & 7);
    } else if (a.xcd_map == 1) {   // few samples: XCD x owns the (sample, head) pairs bh = x (mod 8)
      const int ul = slot / tiles;
      q_tile = slot - ul * tiles;
      const int bh = ul * 8 + (L & 7);
      b = bh / a.heads;
      h = bh - b * a.heads;
    } else {                       // fewer than 8 (sample, head) pairs: plain order, the query tile fastest
      const int bh = L / tiles;
      q_tile = L - bh * tiles;
      b = bh / a.heads;
      h = bh - b * a.heads;
    }
    if (b >= a.B) return;   // the grid is padded to whole groups of 8 (workgroup-uniform exit before any barrier)
  }
  const float sscale = a.scale * 1.44269504088896341f;
  // token n of sample b -> (segment, row); with one segment (every per-plane / image attention) this is a plain offset
  auto row_of = [&](int n, int* seg) -> int64_t {
    if constexpr (SEG3) return att_row(a, b, n, seg);
    *seg = 0;
    return (int64_t)b * a.seg_len[0] + n;
  };

  // ---- Q fragments ------------------------------------------------------------------------------------
  bf16x8 qb[QF][NKK];
  bool q_ok[QF];
#pragma unroll
  for (int qf = 0; qf < QF; ++qf) {
    const int q_idx = q_tile * (64 * QF) + (wave * QF + qf) * 16 + lr;
    q_ok[qf] = q_idx < N;
    int seg = 0;
    int64_t r = row_of(q_ok[qf] ? q_idx : 0, &seg);
    const T* qp = (const T*)(SEG3 ? a.qkv[seg] : a.qkv[0]) + r * a.ld_qkv + a.q_off + h * a.hs_qkv;
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk)
      qb[qf][kk] = q_ok[qf] && kk * 32 + 8 * lg < d ? *reinterpret_cast<const bf16x8*>(qp + kk * 32 + 8 * lg) : bf16x8{};
  }

  // ---- staging: thread owns NV (key, slot) vectors of every tile ------------------------------------------
  bf16x8 kreg[NV], vreg[NV];
  auto load_tile = [&](int kt) {
#pragma unroll
    for (int it = 0; it < NV; ++it) {
      const int idx = tid + it * 256;
      const int key = idx / SL, slot = idx - key * SL;
      int kg = kt * KT + key;
      kg = kg < N ? kg : N - 1;  // rows past the end repeat the last key: finite data, scores masked below
      int seg = 0;
      int64_t r = row_of(kg, &seg);
      const T* base = (const T*)(SEG3 ? a.qkv[seg] : a.qkv[0]) + r * a.ld_qkv + h * a.hs_qkv + slot * 8;
      const bool cok = slot * 8 < d;
      kreg[it] = cok ? *reinterpret_cast<const bf16x8*>(base + a.k_off) : bf16x8{};
      vreg[it] = cok ? *reinterpret_cast<const bf16x8*>(base + a.v_off) : bf16x8{};
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int it = 0; it < NV; ++it) {
      const int idx = tid + it * 256;
      const int key = idx / SL, slot = idx - key * SL;
      *reinterpret_cast<bf16x8*>(ks + buf * TILE + key * RB + att_swz<D>(key, slot) * 16) = kreg[it];
      *reinterpret_cast<bf16x8*>(vs + buf * TILE + key * RB + ((slot ^ att_vswz<D>(key)) << 4)) = vreg[it];
    }
  };

  // The softmax denominator rides on the matrix pipe: one more accumulator fragment per query fragment takes P against an
  // all-ones "V" fragment, so ol[qf][j] = sum over keys of the bf16 P values that also multiply V (every row of the fragment is
  // the same sum; lane lr reads its own query's).  The loop is VALU-bound at small head dims (d = 32: ~220 VALU issue slots per
  // 64-key tile and wave against 16 MFMAs), and this replaces 32 v_add + 4 ds_bpermute per tile by 4 MFMAs; the running
  // rescale `l = l alpha + psum` is the accumulators' own `o *= alpha`.
  float m_run[QF];
  f32x4 o[QF][NDF], ol[QF];
  bf16x8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (bf16_t)1.0f;
#pragma unroll
  for (int qf = 0; qf < QF; ++qf) {
    m_run[qf] = -INFINITY;
    ol[qf] = f32x4{0, 0, 0, 0};
#pragma unroll
    for (int df = 0; df < NDF; ++df) o[qf][df] = f32x4{0, 0, 0, 0};
  }
  // transposed-read lane constants: lane 4q+p of a 16-lane group addresses key row q, 4 channels at 4p
  const int tq = lr >> 2, tp = lr & 3;
  const int vkey = 4 * lg + tq;                       // key inside an aligned 16-key block
  const int vsw = att_vswz<D>(vkey);                  // (the swizzle depends on key & 7 / (key>>1)&3 / (key>>2)&1 only)
  const int vbase = vkey * RB + (tp & 1) * 8;

  const int ntiles = (N + KT - 1) / KT;
  load_tile(0);
  store_tile(0);
  if (ntiles > 1) load_tile(1);
  __syncthreads();

  if constexpr (LAZY) {
#pragma unroll
    for (int qf = 0; qf < QF; ++qf) m_run[qf] = 0.f;   // the reference before the first tile (which always takes the rescale path)
  }
  for (int kt = 0; kt < ntiles; ++kt) {
    const int buf = kt & 1;
    const unsigned char* kb = ks + buf * TILE;
    const unsigned char* vb = vs + buf * TILE;
    // ---- S^T = K . Q^T -----------------------------------------------------------------------------
    f32x4 sc[QF][4];
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      const int row = 16 * f + lr;
#pragma unroll
      for (int qf = 0; qf < QF; ++qf) {
        const float init = LAZY ? -m_run[qf] : 0.f;
        sc[qf][f] = f32x4{init, init, init, init};
      }
#pragma unroll
      for (int kk = 0; kk < NKK; ++kk) {
        bf16x8 kf = *reinterpret_cast<const bf16x8*>(kb + row * RB + att_swz<D>(row, 4 * kk + lg) * 16);
#pragma unroll
        for (int qf = 0; qf < QF; ++qf) sc[qf][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qb[qf][kk], sc[qf][f], 0, 0, 0);
      }
    }
    // ---- the other buffer is free (everyone passed the barrier that ended tile kt-1): park tile kt+1 there and
    //      start fetching tile kt+2 ----------------------------------------------------------------------
    if (kt + 1 < ntiles) {
      store_tile(buf ^ 1);
      if (kt + 2 < ntiles) load_tile(kt + 2);
    }
    // ---- online softmax (base 2, scale folded into the exponent) -----------------------------------------
    const bool tail = (kt + 1) * KT > N;
    bf16x8 pf[QF][2];
    if constexpr (LAZY) {
      if (tail) {  // wave-uniform: only the last tile pays for the key-range masks
#pragma unroll
        for (int qf = 0; qf < QF; ++qf)
#pragma unroll
          for (int f = 0; f < 4; ++f)
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (kt * KT + 16 * f + 4 * lg + j >= N) sc[qf][f][j] = -INFINITY;
      }
      float lmax[QF];
      bool grow = kt == 0;
#pragma unroll
      for (int qf = 0; qf < QF; ++qf) {
        float t = sc[qf][0][0];
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
          for (int j = 0; j < 4; ++j) t = fmaxf(t, sc[qf][f][j]);
        lmax[qf] = t;
        grow |= t > 6.0f;
      }
      if (__builtin_amdgcn_ballot_w64(grow) != 0) {   // move the reference of every query of the wave to its true running maximum
#pragma unroll
        for (int qf = 0; qf < QF; ++qf) {
          float t = lmax[qf];
          t = fmaxf(t, __shfl_xor(t, 16));
          t = fmaxf(t, __shfl_xor(t, 32));
          t = fmaxf(t, kt == 0 ? -1e30f : 0.f);       // never lower the reference (a masked / very negative tile keeps it)
          if (kt == 0) {
            // nothing accumulated yet: no rescale (and no exp2 of a possibly huge difference)
          } else {
            const float alpha = __builtin_amdgcn_exp2f(-t);
#pragma unroll
            for (int df = 0; df < NDF; ++df) o[qf][df] *= alpha;
            ol[qf][0] *= alpha;
          }
          m_run[qf] += t;
#pragma unroll
          for (int f = 0; f < 4; ++f)
#pragma unroll
            for (int j = 0; j < 4; ++j) sc[qf][f][j] -= t;
        }
      }
#pragma unroll
      for (int qf = 0; qf < QF; ++qf)
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
          for (int j = 0; j < 4; ++j) pf[qf][f >> 1][(f & 1) * 4 + j] = (bf16_t)__builtin_amdgcn_exp2f(sc[qf][f][j]);
    } else {
#pragma unroll
    for (int qf = 0; qf < QF; ++qf) {
      float tmax = -INFINITY;
      if (tail) {  // wave-uniform: only the last tile pays for the key-range masks
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (kt * KT + 16 * f + 4 * lg + j >= N) sc[qf][f][j] = -INFINITY;
      }
#pragma unroll
      for (int f = 0; f < 4; ++f)
#pragma unroll
        for (int j = 0; j < 4; ++j) tmax = fmaxf(tmax, sc[qf][f][j]);
      tmax = fmaxf(tmax, __shfl_xor(tmax, 16));
      tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
      const float m_new = fmaxf(m_run[qf], tmax * sscale);
      const float alpha = __builtin_amdgcn_exp2f(m_run[qf] - m_new);
#pragma unroll
      for (int f = 0; f < 4; ++f)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float p = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[qf][f][j], sscale, -m_new));
          pf[qf][f >> 1][(f & 1) * 4 + j] = (bf16_t)p;
        }
      m_run[qf] = m_new;
#pragma unroll
      for (int df = 0; df < NDF; ++df) o[qf][df] *= alpha;
      ol[qf][0] *= alpha;   // the other three rows of the fragment hold the same sum and are never read
    }
    }
    // ---- O^T += V^T . P^T, V^T fragments by transposed LDS reads -------------------------------------------
#pragma unroll
    for (int u = 0; u < 2; ++u) {
#pragma unroll
      for (int qf = 0; qf < QF; ++qf) ol[qf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, pf[qf][u], ol[qf], 0, 0, 0);
#pragma unroll
      for (int df = 0; df < NDF; ++df) {
        const int coff = ((2 * df + (tp >> 1)) ^ vsw) << 4;
        // whole-vector reinterpretation: an element-wise bit_cast short -> bf16 is folded by hipcc 7.2 into a splat
        // of element 0 (seen in the ISA as v_perm + v_mov of the low dword), which silently drops three of four keys
        typedef short s16x4 __attribute__((ext_vector_type(4)));
        typedef __attribute__((address_space(3))) s16x4* lds_s16x4;
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(vb + vbase + (32 * u) * RB + coff));
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(vb + vbase + (32 * u + 16) * RB + coff));
        typedef short s16x8 __attribute__((ext_vector_type(8)));
        s16x8 both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        bf16x8 vf = __builtin_bit_cast(bf16x8, both);
#pragma unroll
        for (int qf = 0; qf < QF; ++qf) o[qf][df] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[qf][u], o[qf][df], 0, 0, 0);
      }
    }
    __syncthreads();  // tile kt consumed by everyone; tile kt+1 visible
  }

#pragma unroll
  for (int qf = 0; qf < QF; ++qf) {
    const int q_idx = q_tile * (64 * QF) + (wave * QF + qf) * 16 + lr;
    if (!q_ok[qf]) continue;
    const float inv = 1.0f / ol[qf][0];
    // row log-sum-exp for the training backward: m_run is the (reference) maximum in the exp2 domain, ol the sum of the bf16 P
    // values that multiplied V - the normalisation the output actually carries
    if (a.lse != nullptr && lg == 0) a.lse[((int64_t)b * a.heads + h) * N + q_idx] = (m_run[qf] + __log2f(ol[qf][0])) * 0.6931471805599453f;
    int seg = 0;
    int64_t r = row_of(q_idx, &seg);
    T* op = (T*)(SEG3 ? a.out[seg] : a.out[0]) + r * a.ld_out + h * a.hs_out;
#pragma unroll
    for (int df = 0; df < NDF; ++df) {
      const int c0 = 16 * df + 4 * lg;
      if (c0 >= d) continue;
      f32x4 v = o[qf][df] * inv;
      bf16x4 t;
#pragma unroll
      for (int j = 0; j < 4; ++j) t[j] = (bf16_t)v[j];
      *reinterpret_cast<bf16x4*>(op + c0) = t;
    }
  }
}

// =====================================================================================================
// RESIDENT form (round 4; the tri-plane UNet's 16-head token attention, openaimodel.py:1148-1169: d = 16 over N = 2048 tokens and
// d = 32 over 512 - 16 % of config 5's kernel time on the tiled kernel above, whose every 128-query workgroup re-streams the head's
// whole K / V through LDS behind a barrier per 64 keys).  One workgroup = 8 waves = ONE (sample, head): K and V of the head are brought
// into LDS once by LDS-DMA (N x d x 2 x 2 B = 128 KB at d = 16, N = 2048) and every wave runs N / 8 queries against them, four 16-query
// fragments at a time - k_attnblock.hip's phase B: no global loads, no staging stores and no barrier inside the loop, the lazy
// reference maximum, the BOUNDED tiles (|q| max|k| within 2^100 of the first tile's maximum: no per-tile maximum at all) and the
// priority hand-over between the two waves of a SIMD.
// d = 16 on the K = 32 MFMA: a K fragment read serves TWO 16-key blocks - lanes 0-31 hold the 16 channels of keys 0-15, lanes 32-63
// those of keys 16-31 - against two images of the query fragment with the other half zeroed (q | 0 and 0 | q): the same MFMA count as
// zero-padding the head dim, half the LDS reads, and the 32-byte K rows need no swizzle (a ds_read_b128 lane group covers 512
// contiguous bytes, a transposed V read 256).
// Requirements (else the tiled kernel): bf16, q pre-scaled (LAZY), d == 16 or 32 exactly, N a multiple of 512 with K + V <= 128 KB,
// segments multiples of 64 tokens, no log-sum-exp output.
// =====================================================================================================
typedef __attribute__((address_space(3))) void att_lds_void;
typedef __attribute__((address_space(3))) unsigned char* att_lds_u8;
template <int D, bool SEG3>
__global__ void __launch_bounds__(512) attention_res_kernel(AttnArgs a) {
  using T = bf16_t;
  constexpr int RB = D * 2, KT = 64, QF = 4, NDF = D / 16, SL = D / 8;
  static_assert(D == 16 || D == 32, "resident attention: head dim 16 or 32");
  extern __shared__ __attribute__((aligned(16))) unsigned char att_smem[];
  const int N = a.seg_len[0] + a.seg_len[1] + a.seg_len[2];
  unsigned char* const Ks = att_smem;
  unsigned char* const Vs = att_smem + N * RB;
  float* const red = reinterpret_cast<float*>(att_smem + 2 * N * RB);   // [8] per-wave maxima of |k|^2

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lg = lane >> 4;
  // XCD x (= blockIdx.x % 8 under round-robin dispatch) owns the samples b = x (mod 8): the 16 heads of a sample interleave inside
  // the same token rows (a 128-byte line holds four heads' 32-byte K at d = 16) and read them through one L2
  const int L = blockIdx.x, slot = L >> 3;
  const int bl = slot / a.heads, h = slot - bl * a.heads, b = bl * 8 + (L & 7);
  if (b >= a.B) return;   // padded grid: workgroup-uniform, before any barrier

  // ---- K | V of (b, h) -> LDS by LDS-DMA: one wave-instruction = 64 consecutive 16-byte pieces = 64 / SL keys of one segment ----
  {
    const int kpw = 64 / SL;                                  // keys per wave-instruction
    const int key_l = lane / SL, ps = lane - key_l * SL;      // key inside the instruction's block, PHYSICAL slot of its LDS row
    for (int it = wave; it < N / kpw; it += 8) {
      int n = it * kpw, seg = 0;                              // wave-uniform: a block never straddles segments (multiples of 64)
      if (SEG3) {
        if (n >= a.seg_len[0]) { n -= a.seg_len[0]; seg = 1; if (n >= a.seg_len[1]) { n -= a.seg_len[1]; seg = 2; } }
      }
      const int sl = a.seg_len[seg];
      const T* base = (const T*)a.qkv[seg] + (int64_t)b * sl * a.ld_qkv + h * a.hs_qkv;
      auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (unsigned)(sl * a.ld_qkv * 2), 0x00020000);
      const int key = it * kpw + key_l;                       // LDS row
      int ksl = ps, vsl = ps;                                 // logical slot that lives in physical slot ps (the swizzles are XORs)
      if constexpr (D == 32) {
        ksl = att_swz<32>(key, ps);
        vsl = ps ^ att_vswz<32>(key);
      }
      const unsigned row = (unsigned)((n + key_l) * a.ld_qkv);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (att_lds_void*)(Ks + it * 1024), 16, (row + a.k_off + ksl * 8) * 2, 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (att_lds_void*)(Vs + it * 1024), 16, (row + a.v_off + vsl * 8) * 2, 0, 0, 0);
    }
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
  __syncthreads();
  // ---- max over the keys of |k|^2 (the bounded tiles' Cauchy-Schwarz factor): thread t squares rows t, t + 512, ... -------------
  {
    float km = 0.f;
    for (int key = tid; key < N; key += 512) {
      float s2 = 0.f;
#pragma unroll
      for (int sl = 0; sl < SL; ++sl) {
        const bf16x8 kv = *reinterpret_cast<const bf16x8*>(Ks + key * RB + sl * 16);
#pragma unroll
        for (int e = 0; e < 8; ++e) s2 = __builtin_fmaf((float)kv[e], (float)kv[e], s2);
      }
      km = fmaxf(km, s2);
    }
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) km = fmaxf(km, __shfl_xor(km, m));
    if (lane == 0) red[wave] = km;
  }
  __syncthreads();
  float Kmax2 = 0.f;
#pragma unroll
  for (int w = 0; w < 8; ++w) Kmax2 = fmaxf(Kmax2, red[w]);

  bf16x8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (bf16_t)1.0f;
  const int tq = lr >> 2, tp = lr & 3;
  const int vkey = 4 * lg + tq;
  const int vsw = D == 32 ? att_vswz<32>(vkey) : 0;
  const int vbase = vkey * RB + (tp & 1) * 8;
  const int ntiles = N / KT;
  const int qpw = N >> 3;               // queries per wave
  const int rounds = qpw / (16 * QF);
  auto row_of = [&](int n, int* seg) -> int64_t {
    if constexpr (SEG3) return att_row(a, b, n, seg);
    *seg = 0;
    return (int64_t)b * a.seg_len[0] + n;
  };

#pragma unroll 1
  for (int rnd = 0; rnd < rounds; ++rnd) {
    const int q0 = wave * qpw + rnd * QF * 16;
    // ---- Q fragments of the round (B operands); d = 16: the two half-zeroed images ------------------------------------------------
    bf16x8 qa[QF], qz[QF];   // D = 32: qa only.  D = 16: qa = (q | 0) for the keys in lanes 0-31, qz = (0 | q) for those in lanes 32-63
    float qmax2 = 0.f;
#pragma unroll
    for (int qf = 0; qf < QF; ++qf) {
      int seg = 0;
      const int64_t r = row_of(q0 + qf * 16 + lr, &seg);
      const T* qp = (const T*)(SEG3 ? a.qkv[seg] : a.qkv[0]) + r * a.ld_qkv + a.q_off + h * a.hs_qkv;
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(qp + (D == 32 ? 8 * lg : 8 * (lg & 1)));
      float s2 = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) s2 = __builtin_fmaf((float)v[e], (float)v[e], s2);
      if constexpr (D == 32) {
        qa[qf] = v;
      } else {
        qa[qf] = lg < 2 ? v : bf16x8{};
        qz[qf] = lg < 2 ? bf16x8{} : v;
        s2 = lg < 2 ? s2 : 0.f;
      }
      qmax2 = fmaxf(qmax2, s2);
    }
    // |q|^2 of a row = the sum of its lane groups' partials <= the sum over the groups of each group's maximum over rows
    qmax2 = fmaxf(qmax2, __shfl_xor(qmax2, 1)); qmax2 = fmaxf(qmax2, __shfl_xor(qmax2, 2));
    qmax2 = fmaxf(qmax2, __shfl_xor(qmax2, 4)); qmax2 = fmaxf(qmax2, __shfl_xor(qmax2, 8));
    qmax2 += __shfl_xor(qmax2, 16); qmax2 += __shfl_xor(qmax2, 32);

    float m_run[QF];
    f32x4 o[QF][NDF], ol[QF];
#pragma unroll
    for (int qf = 0; qf < QF; ++qf) {
      m_run[qf] = 0.f;
      ol[qf] = f32x4{0, 0, 0, 0};
#pragma unroll
      for (int df = 0; df < NDF; ++df) o[qf][df] = f32x4{0, 0, 0, 0};
    }
    // per-lane LDS cursors, advanced by one 64-key tile per trip
    const unsigned char* kp = D == 32 ? Ks + lr * RB + att_swz<32>(lr, lg) * 16      // (the swizzle looks at row bits 2-3 only)
                                      : Ks + lr * RB + (lg & 1) * 16 + (lg >> 1) * (16 * RB);
    const unsigned char* vp[NDF];
#pragma unroll
    for (int df = 0; df < NDF; ++df) vp[df] = Vs + vbase + (((2 * df + (tp >> 1)) ^ vsw) << 4);

    auto tile = [&](auto TRACK, int kt) {
      {   // the two waves of a SIMD (w, w + 4): the younger holds the priority for the first two thirds of its tiles (k_attnblock.hip)
        const int prog = rnd * ntiles + kt, sw = (2 * rounds * ntiles + 1) / 3;
        if (prog == 0 || prog == sw) {
          if ((((wave >> 2) ^ (prog == 0 ? 0 : 1)) & 1) != 0) __builtin_amdgcn_s_setprio(1);
          else __builtin_amdgcn_s_setprio(0);
        }
      }
      f32x4 sc[QF][4];
      if constexpr (D == 32) {
#pragma unroll
        for (int f = 0; f < 4; ++f) {
          const bf16x8 kf = *reinterpret_cast<const bf16x8*>(kp + 16 * f * RB);
#pragma unroll
          for (int qf = 0; qf < QF; ++qf) {
            const float init = -m_run[qf];
            sc[qf][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qa[qf], f32x4{init, init, init, init}, 0, 0, 0);
          }
        }
      } else {
#pragma unroll
        for (int f2 = 0; f2 < 2; ++f2) {
          const bf16x8 kf = *reinterpret_cast<const bf16x8*>(kp + 32 * f2 * RB);   // keys 32 f2 + (0-15 | 16-31)
#pragma unroll
          for (int qf = 0; qf < QF; ++qf) {
            const float init = -m_run[qf];
            sc[qf][2 * f2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qa[qf], f32x4{init, init, init, init}, 0, 0, 0);
            sc[qf][2 * f2 + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qz[qf], f32x4{init, init, init, init}, 0, 0, 0);
          }
        }
      }
      if constexpr (decltype(TRACK)::value) {
        float lmax[QF];
        bool grow = kt == 0;
#pragma unroll
        for (int qf = 0; qf < QF; ++qf) {
          float t = sc[qf][0][0];
#pragma unroll
          for (int f = 0; f < 4; ++f)
#pragma unroll
            for (int j = 0; j < 4; ++j) t = fmaxf(t, sc[qf][f][j]);
          lmax[qf] = t;
          grow |= t > 6.0f;
        }
        if (__builtin_amdgcn_ballot_w64(grow) != 0) {   // move the reference of every query of the wave to its running maximum
#pragma unroll
          for (int qf = 0; qf < QF; ++qf) {
            float t = lmax[qf];
            t = fmaxf(t, __shfl_xor(t, 16));
            t = fmaxf(t, __shfl_xor(t, 32));
            t = fmaxf(t, kt == 0 ? -1e30f : 0.f);
            if (kt != 0) {
              const float alpha = __builtin_amdgcn_exp2f(-t);
#pragma unroll
              for (int df = 0; df < NDF; ++df) o[qf][df] *= alpha;
              ol[qf][0] *= alpha;
            }
            m_run[qf] += t;
#pragma unroll
            for (int f = 0; f < 4; ++f)
#pragma unroll
              for (int j = 0; j < 4; ++j) sc[qf][f][j] -= t;
          }
        }
      }
      bf16x8 pf[QF][2];
#pragma unroll
      for (int qf = 0; qf < QF; ++qf)
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
          for (int j = 0; j < 4; ++j) pf[qf][f >> 1][(f & 1) * 4 + j] = (bf16_t)__builtin_amdgcn_exp2f(sc[qf][f][j]);
#pragma unroll
      for (int u = 0; u < 2; ++u) {
#pragma unroll
        for (int qf = 0; qf < QF; ++qf) ol[qf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, pf[qf][u], ol[qf], 0, 0, 0);
#pragma unroll
        for (int df = 0; df < NDF; ++df) {
          typedef short s16x4 __attribute__((ext_vector_type(4)));
          typedef __attribute__((address_space(3))) s16x4* lds_s16x4;
          s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(vp[df] + (32 * u) * RB));
          s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(vp[df] + (32 * u + 16) * RB));
          typedef short s16x8 __attribute__((ext_vector_type(8)));
          s16x8 both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
          const bf16x8 vf = __builtin_bit_cast(bf16x8, both);   // whole-vector reinterpretation (the element-wise one miscompiles, above)
#pragma unroll
          for (int qf = 0; qf < QF; ++qf) o[qf][df] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[qf][u], o[qf][df], 0, 0, 0);
        }
      }
      kp += KT * RB;
#pragma unroll
      for (int df = 0; df < NDF; ++df) vp[df] += KT * RB;
    };
    tile(std::true_type{}, 0);   // the first tile sets the reference to its true maximum
    bool loose = false;
    {
      const float bound = __builtin_sqrtf(qmax2 * Kmax2);
#pragma unroll
      for (int qf = 0; qf < QF; ++qf) loose |= !(bound - m_run[qf] <= 100.0f);
    }
    if (__builtin_amdgcn_ballot_w64(loose) == 0) {
#pragma unroll 1
      for (int kt = 1; kt < ntiles; ++kt) tile(std::false_type{}, kt);
    } else {
#pragma unroll 1
      for (int kt = 1; kt < ntiles; ++kt) tile(std::true_type{}, kt);
    }
#pragma unroll
    for (int qf = 0; qf < QF; ++qf) {
      const float inv = 1.0f / ol[qf][0];
      int seg = 0;
      const int64_t r = row_of(q0 + qf * 16 + lr, &seg);
      T* op = (T*)(SEG3 ? a.out[seg] : a.out[0]) + r * a.ld_out + h * a.hs_out;
#pragma unroll
      for (int df = 0; df < NDF; ++df) {
        const f32x4 v = o[qf][df] * inv;
        bf16x4 t;
#pragma unroll
        for (int j = 0; j < 4; ++j) t[j] = (bf16_t)v[j];
        *reinterpret_cast<bf16x4*>(op + 16 * df + 4 * lg) = t;
      }
    }
  }
}

// =====================================================================================================
// RESIDENT form for d = 16 on the 32 x 32 x 16 MFMA (the tri-plane UNet's first attention level, 11 % of config 5's step on the 16 x 16 x 32
// form above).  K = 16 IS the head dim, so nothing is padded, and the shape's 32 rows leave room for what the 16 x 16 form needed extra
// MFMAs for:
//   S^T = K . Q^T      ONE MFMA per 32 keys x 32 queries (A = 32 key rows of 32 B straight from LDS: lane (key l & 31, half l >> 5)
//                      reads its 16 B - a contiguous KiB, no swizzle; B = the query fragment, loaded once)
//   O^T = V^T . P^T    A = [16 channels of V^T | a ROW OF ONES | 15 zero rows] x 16 keys: the row of ones delivers the softmax
//                      denominator as row 16 of the same product - two MFMAs per 32 keys carry the output AND the row sums.  P^T is the
//                      score accumulator itself (registers 8 s .. 8 s + 7 -> k-step s, cdna_hip_programming.md "accumulator tile as
//                      the next MFMA's operand"); the V^T fragment comes by two transposed reads per k-step whose lanes 16-31 / 48-63
//                      (rows 16-31 of A) point at a constant LDS table [1, 0, ..., 0] per 32-byte row instead of V - no select.
// Per 1024 scores: 3 MFMAs of 32 cycles (8 of 16 before) that hold the vector issue for 24 cycles (64 before) beside the same 16
// exponentials per lane.  And with every score of a query within 2^100 of ZERO (|q| max|k| <= 100: the Cauchy-Schwarz bound the
// 16 x 16 form compares with its first tile) no reference maximum is needed at all: p = exp2(s), C = 0.  Rounds whose bound is larger
// run a tracked loop (p = exp2(s - m), m moved lazily).
// Requirements (else the 16 x 16 form): d == 16, N a multiple of 1024 and <= 2048, the resident form's others.
// =====================================================================================================
typedef __attribute__((ext_vector_type(16))) float f32x16;
template <bool SEG3>
__global__ void __launch_bounds__(512) attention_res16_kernel(AttnArgs a) {
  using T = bf16_t;
  constexpr int RB = 32, QF = 4;   // four 32-query fragments per wave and round
  extern __shared__ __attribute__((aligned(16))) unsigned char att_smem[];
  const int N = a.seg_len[0] + a.seg_len[1] + a.seg_len[2];
  unsigned char* const Ks = att_smem;
  unsigned char* const Vs = att_smem + N * RB;
  float* const red = reinterpret_cast<float*>(att_smem + 2 * N * RB);   // [8] per-wave maxima of |k|^2
  unsigned char* const tab = att_smem + 2 * N * RB + 256;               // 1280 B of 32-byte rows [1, 0 x 15] (256-byte aligned)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, hf = lane >> 5;
  const int L = blockIdx.x, slot = L >> 3;
  const int bl = slot / a.heads, h = slot - bl * a.heads, b = bl * 8 + (L & 7);
  if (b >= a.B) return;   // padded grid: workgroup-uniform, before any barrier

  for (int e = tid; e < 640; e += 512) reinterpret_cast<bf16_t*>(tab)[e] = (bf16_t)((e & 15) == 0 ? 1.0f : 0.0f);
  // ---- K | V of (b, h) -> LDS by LDS-DMA (attention_res_kernel's fill at d = 16: 32 keys per wave-instruction, natural rows) ----------
  {
    const int key_l = lane >> 1, ps = lane & 1;
    for (int it = wave; it < N / 32; it += 8) {
      int n = it * 32, seg = 0;
      if (SEG3) {
        if (n >= a.seg_len[0]) { n -= a.seg_len[0]; seg = 1; if (n >= a.seg_len[1]) { n -= a.seg_len[1]; seg = 2; } }
      }
      const int sl = a.seg_len[seg];
      const T* base = (const T*)a.qkv[seg] + (int64_t)b * sl * a.ld_qkv + h * a.hs_qkv;
      auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (unsigned)(sl * a.ld_qkv * 2), 0x00020000);
      const unsigned row = (unsigned)((n + key_l) * a.ld_qkv);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (att_lds_void*)(Ks + it * 1024), 16, (row + a.k_off + ps * 8) * 2, 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (att_lds_void*)(Vs + it * 1024), 16, (row + a.v_off + ps * 8) * 2, 0, 0, 0);
    }
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
  __syncthreads();
  {
    float km = 0.f;
    for (int key = tid; key < N; key += 512) {
      float s2 = 0.f;
#pragma unroll
      for (int sl = 0; sl < 2; ++sl) {
        const bf16x8 kv = *reinterpret_cast<const bf16x8*>(Ks + key * RB + sl * 16);
#pragma unroll
        for (int e = 0; e < 8; ++e) s2 = __builtin_fmaf((float)kv[e], (float)kv[e], s2);
      }
      km = fmaxf(km, s2);
    }
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) km = fmaxf(km, __shfl_xor(km, m));
    if (lane == 0) red[wave] = km;
  }
  __syncthreads();
  float Kmax2 = 0.f;
#pragma unroll
  for (int w = 0; w < 8; ++w) Kmax2 = fmaxf(Kmax2, red[w]);

  auto row_of = [&](int n, int* seg) -> int64_t {
    if constexpr (SEG3) return att_row(a, b, n, seg);
    *seg = 0;
    return (int64_t)b * a.seg_len[0] + n;
  };
  // transposed-read lane roles: 16-lane group g = lane >> 4; groups 0 / 2 (rows 0-15 of A, halves 0 / 1) read V, groups 1 / 3 the table.
  // A read of half hf touches V rows 4 hf + tq (+ 8 jj + 16 s + 32 tile): banks 0-31 for hf = 0, 32-63 for hf = 1 (128-byte windows) -
  // the table lanes of the same 32-lane half use the OTHER bank half (table + 128 for hf = 0).
  const int i16 = lane & 15, tq = i16 >> 2, tp = i16 & 3;
  const bool real = ((lane >> 4) & 1) == 0;
  const unsigned char* const v0 = real ? Vs + (4 * hf + tq) * RB + tp * 8 : tab + (hf == 0 ? 128 : 0) + tq * RB + tp * 8;
  const int vstep = real ? 32 * RB : 0;   // per 32-key tile
  const int ntiles = N / 32, qpw = N >> 3, rounds = qpw / (32 * QF);
  typedef short s16x4 __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(3))) s16x4* lds_s16x4;
  typedef short s16x8 __attribute__((ext_vector_type(8)));

#pragma unroll 1
  for (int rnd = 0; rnd < rounds; ++rnd) {
    const int q0 = wave * qpw + rnd * (32 * QF);
    bf16x8 qb[QF];
    float qmax2 = 0.f;
#pragma unroll
    for (int qf = 0; qf < QF; ++qf) {
      int seg = 0;
      const int64_t rr = row_of(q0 + qf * 32 + r, &seg);
      const T* qp = (const T*)(SEG3 ? a.qkv[seg] : a.qkv[0]) + rr * a.ld_qkv + a.q_off + h * a.hs_qkv;
      qb[qf] = *reinterpret_cast<const bf16x8*>(qp + 8 * hf);
      float s2 = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) s2 = __builtin_fmaf((float)qb[qf][e], (float)qb[qf][e], s2);
      qmax2 = fmaxf(qmax2, s2);
    }
    // |q|^2 of a row = its two halves' partials <= the sum over the halves of each half's maximum over rows
#pragma unroll
    for (int m = 1; m < 32; m <<= 1) qmax2 = fmaxf(qmax2, __shfl_xor(qmax2, m));
    qmax2 += __shfl_xor(qmax2, 32);
    const bool plain = __builtin_sqrtf(qmax2 * Kmax2) <= 100.0f;   // every score of the round within 2^100 of zero (false for NaN)

    f32x16 o[QF];
#pragma unroll
    for (int qf = 0; qf < QF; ++qf)
#pragma unroll
      for (int e = 0; e < 16; ++e) o[qf][e] = 0.f;
    const unsigned char* kp = Ks + r * RB + hf * 16;
    const unsigned char* vp = v0;
    {   // the two waves of a SIMD (w, w + 4): the younger holds the priority for the first two thirds of the rounds' tiles (k_attnblock.hip)
      if (ATT_RES16_PRIO && rnd == 0) { if ((wave >> 2) & 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
    }
    auto pv = [&](const bf16x8 (&pa)[QF], const bf16x8 (&pb)[QF]) {
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(vp + 512 * s2));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(vp + 512 * s2 + 256));
        const s16x8 both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        const bf16x8 vf = __builtin_bit_cast(bf16x8, both);
#pragma unroll
        for (int qf = 0; qf < QF; ++qf) o[qf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, s2 == 0 ? pa[qf] : pb[qf], o[qf], 0, 0, 0);
      }
    };
    if (__builtin_amdgcn_ballot_w64(!plain) == 0) {
#pragma unroll 1
      for (int kt = 0; kt < ntiles; ++kt) {
        if (ATT_RES16_PRIO && rnd * ntiles + kt == (2 * rounds * ntiles + 1) / 3) { if ((wave >> 2) & 1) __builtin_amdgcn_s_setprio(0); else __builtin_amdgcn_s_setprio(1); }
        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(kp);
        bf16x8 pa[QF], pb[QF];
#pragma unroll
        for (int qf = 0; qf < QF; ++qf) {
          f32x16 z;
#pragma unroll
          for (int e = 0; e < 16; ++e) z[e] = 0.f;
          const f32x16 sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qb[qf], z, 0, 0, 0);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            pa[qf][e] = (bf16_t)__builtin_amdgcn_exp2f(sc[e]);
            pb[qf][e] = (bf16_t)__builtin_amdgcn_exp2f(sc[8 + e]);
          }
        }
        pv(pa, pb);
        kp += 32 * RB;
        vp += vstep;
      }
    } else {
      // tracked: a per-query reference m (both halves of a query's lane pair hold the same), moved when a tile's maximum exceeds it by 2^6
      float m_run[QF];
#pragma unroll 1
      for (int kt = 0; kt < ntiles; ++kt) {
        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(kp);
        bf16x8 pa[QF], pb[QF];
#pragma unroll
        for (int qf = 0; qf < QF; ++qf) {
          f32x16 z;
#pragma unroll
          for (int e = 0; e < 16; ++e) z[e] = 0.f;
          const f32x16 sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qb[qf], z, 0, 0, 0);
          float t = sc[0];
#pragma unroll
          for (int e = 1; e < 16; ++e) t = fmaxf(t, sc[e]);
          t = fmaxf(t, __shfl_xor(t, 32));
          if (kt == 0) {
            m_run[qf] = t;
          } else if (__builtin_amdgcn_ballot_w64(t > m_run[qf] + 6.0f) != 0) {
            const float mn = fmaxf(m_run[qf], t);
            const float alpha = __builtin_amdgcn_exp2f(m_run[qf] - mn);
#pragma unroll
            for (int e = 0; e < 16; ++e) o[qf][e] *= alpha;
            m_run[qf] = mn;
          }
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            pa[qf][e] = (bf16_t)__builtin_amdgcn_exp2f(sc[e] - m_run[qf]);
            pb[qf][e] = (bf16_t)__builtin_amdgcn_exp2f(sc[8 + e] - m_run[qf]);
          }
        }
        pv(pa, pb);
        kp += 32 * RB;
        vp += vstep;
      }
    }
    // ---- output: row 16 of O^T (register 8 of the half-0 lane) is the denominator; the lane holds channels 4 hf + e and 8 + 4 hf + e ----
#pragma unroll
    for (int qf = 0; qf < QF; ++qf) {
      const float den = __shfl(o[qf][8], r);
      const float inv = 1.0f / den;
      int seg = 0;
      const int64_t rr = row_of(q0 + qf * 32 + r, &seg);
      T* op = (T*)(SEG3 ? a.out[seg] : a.out[0]) + rr * a.ld_out + h * a.hs_out;
      bf16x4 t0, t1;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        t0[e] = (bf16_t)(o[qf][e] * inv);
        t1[e] = (bf16_t)(o[qf][4 + e] * inv);
      }
      *reinterpret_cast<bf16x4*>(op + 4 * hf) = t0;
      *reinterpret_cast<bf16x4*>(op + 8 + 4 * hf) = t1;
    }
  }
}

// =====================================================================================================
// STREAMING form for long sequences (round 4; the video decoder's inter-plane attention, autoencoder_unet.py:404-534: 8 heads of
// d = 64 over 73 728 tokens and d = 128 over 20 480 - a fifth of config 5's step on the tiled kernel above, at 1.4 / 0.33 scores per
// clock and SIMD).  Same loop as the resident form, but K / V cannot stay: one workgroup = 8 waves = 128 QF queries of one (sample,
// head) - four times the tiled kernel's, so a head's K / V stream through L2 a quarter as often - and the 64-key tiles come through a
// THREE-stage LDS ring by LDS-DMA: no registers, no ds_write and no per-key row arithmetic in the loop (the tiled kernel's staging is
// two global loads + two ds_write_b128 + the three-segment row map per thread and tile, beside a VALU-bound softmax).  A tile never
// straddles a segment (segments are multiples of 64 tokens), so the row map is one scalar decision per tile; the per-lane source
// offsets (key row, swizzled slot) are two registers per operand for the whole kernel.  One barrier per tile: behind it the DMA of
// tile t + 2 is issued into the stage tile t - 1 was read from.
// d = 256 / 512 (the decoder's deepest inter-plane levels: 6 144 and 2 048 tokens; 64 - 160 TFLOP/s on the generic kernel, which
// recomputes S per 256-channel value slice): QF = 1, 32-key tiles, the V^T fragments read in groups of four; at d = 512 a stage is 64 KB
// and the ring has two (the DMA of tile t + 1 flies under tile t).
// Requirements (else the tiled kernels): bf16, q pre-scaled (LAZY), d == 64 (QF = 4), 128 (QF = 2), 256 or 512 (QF = 1), N a multiple of
// 128 QF and >= 2048, segments multiples of 64 tokens, no log-sum-exp output.
// =====================================================================================================
template <int D, int QF, int KT, int NS, bool SEG3>
__global__ void __launch_bounds__(512) attention_stream_kernel(AttnArgs a) {
  using T = bf16_t;
  constexpr int RB = D * 2, NDF = D / 16, NKK = D / 32, SL = D / 8;
  constexpr int NF = KT / 16, NU = KT / 32;   // 16-key score blocks / 32-key PV steps per tile
  constexpr int TILE = KT * RB, STAGE = 2 * TILE;
  constexpr int KPP = 1024 / RB;            // keys per DMA wave-instruction (64 lanes x 16 B)
  constexpr int PIECES = TILE / 1024;       // DMA wave-instructions per operand tile
  constexpr bool SPLIT = PIECES == 4;       // d = 32: a tile is 4 + 4 pieces - waves 0-3 fetch K, waves 4-7 fetch V, one instruction each
  constexpr int DPW = SPLIT ? 1 : PIECES / 8;   // DMA instructions per wave, tile and operand: 1 (d = 64), 2 (d = 128, 256), 4 (d = 512)
  constexpr int DW = SPLIT ? 1 : 2 * DPW;   // DMA instructions per wave and tile
  constexpr int PD = NS - 1;                // tiles in flight ahead of the one being consumed
  static_assert((D == 32 || D == 64 || D == 128 || D == 256 || D == 512) && (KT == 64 || KT == 32) && (SPLIT || (PIECES % 8 == 0 && DPW <= 4)) &&
                (NS == 2 || NS == 3), "streaming attention: unsupported tile");
  extern __shared__ __attribute__((aligned(16))) unsigned char att_smem[];   // [NS][K tile | V tile] | redo flag
  int* const redo = reinterpret_cast<int*>(att_smem + NS * STAGE);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lg = lane >> 4;
  const int N = a.seg_len[0] + a.seg_len[1] + a.seg_len[2];
  int q_tile, b, h;
  {   // attention_v2_kernel's XCD-aware order: all query tiles of a (sample, head) on one XCD, so its K / V stream is an L2 hit for
      // the workgroups that run beside each other
    const int tiles = N / (128 * QF);
    const int L = blockIdx.x, slot = L >> 3;
    if (a.xcd_map == 0) {
      const int per_sample = tiles * a.heads;
      const int bl = slot / per_sample, rem = slot - bl * per_sample;
      h = rem / tiles;
      q_tile = rem - h * tiles;
      b = bl * 8 + (L & 7);
    } else if (a.xcd_map == 1) {
      const int ul = slot / tiles;
      q_tile = slot - ul * tiles;
      const int bh = ul * 8 + (L & 7);
      b = bh / a.heads;
      h = bh - b * a.heads;
    } else {
      const int bh = L / tiles;
      q_tile = L - bh * tiles;
      b = bh / a.heads;
      h = bh - b * a.heads;
    }
    if (b >= a.B) return;   // padded grid: workgroup-uniform, before any barrier
  }
  if (tid == 0) *redo = 0;   // (visible behind the first tile's barrier)
  auto row_of = [&](int n, int* seg) -> int64_t {
    if constexpr (SEG3) return att_row(a, b, n, seg);
    *seg = 0;
    return (int64_t)b * a.seg_len[0] + n;
  };

  // ---- per-lane DMA source offsets inside a tile (bytes from the tile's first key row): lane -> (key, PHYSICAL slot) ------------
  // (scalars, not a [DPW] array: with the DMA's offset operand read from an array element the host pass of hipcc 7.2 silently drops the
  // kernel's handle - an undefined symbol when the library is loaded, no diagnostic)
  auto lane_off = [&](int piece, int op_off, bool is_v) -> unsigned {
    const int key = piece * KPP + lane / SL, ps = lane % SL;
    const int sl = is_v ? (ps ^ att_vswz<D>(key)) : att_swz<D>(key, ps);
    return (unsigned)((key * a.ld_qkv + op_off + sl * 8) * 2);
  };
  const unsigned lane_k0 = lane_off(SPLIT ? (wave & 3) : wave, a.k_off, false), lane_v0 = lane_off(SPLIT ? (wave & 3) : wave, a.v_off, true);
  const unsigned lane_k1 = DPW > 1 ? lane_off(wave + 8, a.k_off, false) : 0u, lane_v1 = DPW > 1 ? lane_off(wave + 8, a.v_off, true) : 0u;
  const unsigned lane_k2 = DPW > 2 ? lane_off(wave + 16, a.k_off, false) : 0u, lane_v2 = DPW > 2 ? lane_off(wave + 16, a.v_off, true) : 0u;
  const unsigned lane_k3 = DPW > 2 ? lane_off(wave + 24, a.k_off, false) : 0u, lane_v3 = DPW > 2 ? lane_off(wave + 24, a.v_off, true) : 0u;
  // one descriptor per segment, based at (sample b, head h)
  auto seg_rsrc = [&](int sg) -> const T* {
    return a.qkv[sg] ? (const T*)a.qkv[sg] + (int64_t)b * a.seg_len[sg] * a.ld_qkv + h * a.hs_qkv : nullptr;
  };
  const auto rs0 = __builtin_amdgcn_make_buffer_rsrc((void*)seg_rsrc(0), 0, (unsigned)(a.seg_len[0] * a.ld_qkv * 2), 0x00020000);
  const auto rs1 = __builtin_amdgcn_make_buffer_rsrc((void*)seg_rsrc(1), 0, (unsigned)(a.seg_len[1] * a.ld_qkv * 2), 0x00020000);
  const auto rs2 = __builtin_amdgcn_make_buffer_rsrc((void*)seg_rsrc(2), 0, (unsigned)(a.seg_len[2] * a.ld_qkv * 2), 0x00020000);
  const int ntiles = N / KT;
  // ---- Q fragments (B operands), loaded once - BEFORE the first DMAs: vmcnt retires in order, so the loop's "all but the youngest
  //      tile" wait covers them, and hipcc's wait-count pass sees them complete before the loop (issued after the DMAs it put a
  //      vmcnt(0) for the last fragment INSIDE the loop, behind the DMA of tile t + 2: every tile waited for its prefetch) ---------
  const int q0 = q_tile * (128 * QF) + wave * (16 * QF);
  bf16x8 qb[QF][NKK];
#pragma unroll
  for (int qf = 0; qf < QF; ++qf) {
    int seg = 0;
    const int64_t r = row_of(q0 + qf * 16 + lr, &seg);
    const T* qp = (const T*)(SEG3 ? a.qkv[seg] : a.qkv[0]) + r * a.ld_qkv + a.q_off + h * a.hs_qkv;
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk) qb[qf][kk] = *reinterpret_cast<const bf16x8*>(qp + kk * 32 + 8 * lg);
  }

#if ATT_STREAM_PRIO == 1
  if (wave >= 4) __builtin_amdgcn_s_setprio(1);
#elif ATT_STREAM_PRIO == 2
  if (wave < 4) __builtin_amdgcn_s_setprio(1);
#endif
  float m_run[QF];
  f32x4 o[QF][NDF], ol[QF];
  bf16x8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (bf16_t)1.0f;
  const int tq = lr >> 2, tp = lr & 3;
  const int vkey = 4 * lg + tq;
  const int vsw = att_vswz<D>(vkey);                  // (unchanged by + 16 / + 32 key rows)
  const int vbase = vkey * RB + (tp & 1) * 8;
  // K fragment (key block f, k-step kk): row 16 f + lr, logical slot 4 kk + lg.  The swizzle term depends on lr only (every template:
  // (row >> 1) & 7 or row & 15 of a row 16 f + lr), so  address = k0 ^ (kk << 6) + 16 f RB  with ONE register k0
  const int k0 = lr * RB + att_swz<D>(lr, lg) * 16;

  // OPTIMISTIC reference maximum.  The lazy loop moves a query's reference only when a score exceeds it by 2^6, but it still LOOKS at
  // every score: 32 v_max3 + compare + ballot per tile and wave beside a VALU-bound softmax.  Any reference within ~2^100 of the true
  // maximum is exact (p and the sums are floating point, the denominator accumulates the same p), so the first pass takes the FIRST
  // tile's maximum and never looks again; a score that outgrows it by more than 2^127 turns its p, and with it the row sum, into inf
  // (or the sum outgrows fp32): the row sums are checked once at the end, and if any wave of the workgroup saw one out of range the whole
  // workgroup runs the pass again with the tracked maximum (the K / V tiles are the workgroup's: the decision is made together).
  // (one copy of the loop, a run-time flag: as a generic lambda instantiated twice the kernel needed 40 more registers)
#pragma unroll 1
  for (int track_all = D >= 256 ? 1 : 0; track_all < 2; ++track_all) {   // d >= 256 is MFMA- / LDS-bound: tracked from the start
#pragma unroll
  for (int qf = 0; qf < QF; ++qf) {
    m_run[qf] = 0.f;
    ol[qf] = f32x4{0, 0, 0, 0};
#pragma unroll
    for (int df = 0; df < NDF; ++df) o[qf][df] = f32x4{0, 0, 0, 0};
  }
  // The loop starts PD trips early: those trips only issue the DMAs of the first PD tiles (ONE issue site, in the loop body)
  int stage = 0;
#pragma unroll 1
  for (int kt = -PD; kt < ntiles; ++kt) {
    if (kt >= 0) {
      // vmcnt retires in order: with the DMAs of the PD - 1 tiles behind tile kt the youngest, "at most (PD - 1) DW outstanding"
      // means tile kt (and Q) have landed
      if (PD > 1 && kt + 1 < ntiles) __builtin_amdgcn_s_waitcnt(0x0F70 | ((PD - 1) * DW));
      else __builtin_amdgcn_s_waitcnt(0x0F70);
      __builtin_amdgcn_s_barrier();   // everyone's pieces of tile kt are in LDS; everyone is done reading tile kt - 1
    }
    if (kt + PD < ntiles) {   // tile kt + PD -> the stage tile kt - 1 was read from
      int n = (kt + PD) * KT, seg = 0;   // wave-uniform
      if (SEG3) {
        if (n >= a.seg_len[0]) { n -= a.seg_len[0]; seg = 1; if (n >= a.seg_len[1]) { n -= a.seg_len[1]; seg = 2; } }
      }
      const int soff = n * a.ld_qkv * 2;   // the tile's first key row: the scalar offset of the load
      int s2 = kt < 0 ? kt + PD : stage + PD;
      s2 = s2 >= NS ? s2 - NS : s2;
      unsigned char* st = att_smem + s2 * STAGE;
      if constexpr (SPLIT) {   // one instruction per wave: K piece (waves 0-3) or V piece (waves 4-7)
        att_lds_void* dd = (att_lds_void*)(st + (wave >> 2) * TILE + (wave & 3) * 1024);
        const unsigned lo = wave < 4 ? lane_k0 : lane_v0;
        if (!SEG3 || seg == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs0, dd, 16, lo, soff, 0, 0);
        else if (seg == 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, dd, 16, lo, soff, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs2, dd, 16, lo, soff, 0, 0);
      } else {
#pragma unroll
      for (int i = 0; i < DPW; ++i) {
        att_lds_void* kd = (att_lds_void*)(st + (wave + 8 * i) * 1024);
        att_lds_void* vd = (att_lds_void*)(st + TILE + (wave + 8 * i) * 1024);
        const unsigned lk = i == 0 ? lane_k0 : i == 1 ? lane_k1 : i == 2 ? lane_k2 : lane_k3;
        const unsigned lv = i == 0 ? lane_v0 : i == 1 ? lane_v1 : i == 2 ? lane_v2 : lane_v3;
        if (!SEG3 || seg == 0) {
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs0, kd, 16, lk, soff, 0, 0);
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs0, vd, 16, lv, soff, 0, 0);
        } else if (seg == 1) {
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, kd, 16, lk, soff, 0, 0);
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, vd, 16, lv, soff, 0, 0);
        } else {
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs2, kd, 16, lk, soff, 0, 0);
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs2, vd, 16, lv, soff, 0, 0);
        }
      }
      }
    }
    if (kt < 0) continue;
    const unsigned char* kb = att_smem + stage * STAGE;
    const unsigned char* vb = kb + TILE;
    f32x4 sc[QF][NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) {
#pragma unroll
      for (int qf = 0; qf < QF; ++qf) {
        const float init = -m_run[qf];
        sc[qf][f] = f32x4{init, init, init, init};
      }
#pragma unroll
      for (int kk = 0; kk < NKK; ++kk) {
        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(kb + (k0 ^ (kk << 6)) + 16 * f * RB);
#pragma unroll
        for (int qf = 0; qf < QF; ++qf) sc[qf][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qb[qf][kk], sc[qf][f], 0, 0, 0);
      }
    }
    if (track_all != 0 || kt == 0) {   // lazy reference maximum (attention_v2_kernel LAZY)
      float lmax[QF];
      bool grow = kt == 0;
#pragma unroll
      for (int qf = 0; qf < QF; ++qf) {
        float t = sc[qf][0][0];
#pragma unroll
        for (int f = 0; f < NF; ++f)
#pragma unroll
          for (int j = 0; j < 4; ++j) t = fmaxf(t, sc[qf][f][j]);
        lmax[qf] = t;
        grow |= t > 6.0f;
      }
      if (__builtin_amdgcn_ballot_w64(grow) != 0) {
#pragma unroll
        for (int qf = 0; qf < QF; ++qf) {
          float t = lmax[qf];
          t = fmaxf(t, __shfl_xor(t, 16));
          t = fmaxf(t, __shfl_xor(t, 32));
          t = fmaxf(t, kt == 0 ? -1e30f : 0.f);
          if (kt != 0) {
            const float alpha = __builtin_amdgcn_exp2f(-t);
#pragma unroll
            for (int df = 0; df < NDF; ++df) o[qf][df] *= alpha;
            ol[qf][0] *= alpha;
          }
          m_run[qf] += t;
#pragma unroll
          for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int j = 0; j < 4; ++j) sc[qf][f][j] -= t;
        }
      }
    }
    bf16x8 pf[QF][NU];
#pragma unroll
    for (int qf = 0; qf < QF; ++qf)
#pragma unroll
      for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int j = 0; j < 4; ++j) pf[qf][f >> 1][(f & 1) * 4 + j] = (bf16_t)__builtin_amdgcn_exp2f(sc[qf][f][j]);
    // V^T fragments by transposed LDS reads - as INLINE ASM: for the builtin hipcc's wait-count pass assumes the read may alias the
    // LDS-DMA in flight (the tiles fetched ahead, issued above) and puts a vmcnt(0) in front of it, i.e. every tile waits for its own
    // prefetch.  The asm reads are invisible to that pass, so their lgkmcnt wait is written out too (a second asm that "produces" the
    // registers).  Four channel fragments (eight reads) per group: at d = 512 a tile's V^T is 32 fragments.
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const unsigned vaddr = (unsigned)(__SIZE_TYPE__)(att_lds_u8)(vb + vbase);   // 32-bit LDS address
#pragma unroll
    for (int u = 0; u < NU; ++u) {
#pragma unroll
      for (int qf = 0; qf < QF; ++qf) ol[qf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, pf[qf][u], ol[qf], 0, 0, 0);
#pragma unroll
      for (int c = 0; c < (NDF + 3) / 4; ++c) {
        s16x4 lo[4], hi[4];
#pragma unroll
        for (int d4 = 0; d4 < 4; ++d4) {
          const int df = NDF >= 4 ? 4 * c + d4 : (d4 & (NDF - 1));   // (d = 32: two fragments; the group's other two slots re-read them)
          const unsigned ad = vaddr + (unsigned)(((2 * df + (tp >> 1)) ^ vsw) << 4);
          asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\tds_read_b64_tr_b16 %1, %2 offset:%4"
                       : "=&v"(lo[d4]), "=&v"(hi[d4]) : "v"(ad), "n"((32 * u) * RB), "n"((32 * u + 16) * RB) : "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lo[0]), "+v"(hi[0]), "+v"(lo[1]), "+v"(hi[1]), "+v"(lo[2]), "+v"(hi[2]), "+v"(lo[3]), "+v"(hi[3]));
#pragma unroll
        for (int d4 = 0; d4 < (NDF >= 4 ? 4 : NDF); ++d4) {
          s16x8 both = __builtin_shufflevector(lo[d4], hi[d4], 0, 1, 2, 3, 4, 5, 6, 7);
          const bf16x8 vf = __builtin_bit_cast(bf16x8, both);   // whole-vector reinterpretation (the element-wise one miscompiles, above)
#pragma unroll
          for (int qf = 0; qf < QF; ++qf) o[qf][4 * c + d4] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[qf][u], o[qf][4 * c + d4], 0, 0, 0);
        }
      }
    }
    stage = stage + 1 == NS ? 0 : stage + 1;
  }
  if (track_all != 0) break;
  bool bad = false;
#pragma unroll
  for (int qf = 0; qf < QF; ++qf) bad |= !(ol[qf][0] < 1e30f);   // inf, NaN, or within 2^27 of the fp32 range (|v| headroom for o)
  if (__builtin_amdgcn_ballot_w64(bad) != 0 && lane == 0) *redo = 1;
  __syncthreads();   // also: every wave is done with the last tiles before a second pass refills the ring
  if (*redo == 0) break;
  }

#pragma unroll
  for (int qf = 0; qf < QF; ++qf) {
    const float inv = 1.0f / ol[qf][0];
    int seg = 0;
    const int64_t r = row_of(q0 + qf * 16 + lr, &seg);
    T* op = (T*)(SEG3 ? a.out[seg] : a.out[0]) + r * a.ld_out + h * a.hs_out;
#pragma unroll
    for (int df = 0; df < NDF; ++df) {
      const f32x4 v = o[qf][df] * inv;
      bf16x4 t;
#pragma unroll
      for (int j = 0; j < 4; ++j) t[j] = (bf16_t)v[j];
      *reinterpret_cast<bf16x4*>(op + 16 * df + 4 * lg) = t;
    }
  }
}

// =====================================================================================================
// bf16x3 numerics (DDMI_PARITY_BF16X3, k_elementwise.hip / net.hip): fp32 tensors in and out, but both matrix products of the
// attention as THREE bf16 MFMA products each - x = xh + xl with xh = bf16(x), xl = bf16(x - xh) (16 mantissa bits between them),
// q.k ~ qh.kh + ql.kh + qh.kl and p.v likewise, fp32 accumulation, fp32 softmax - instead of the fp32 MFMA (1/16 of the bf16 rate)
// the parity mode's attention ran on (16.5 of its 121 ms per UNet evaluation at B = 128).  The generic kernel's shape: 4 waves x 16
// queries, the query on the MFMA lane; K and V^T tiles in LDS, each as a hi and a lo bf16 image.  Head dims <= 128.
// =====================================================================================================
__device__ __forceinline__ void x3_split(float x, bf16_t& hi, bf16_t& lo) {
  hi = (bf16_t)x;
  lo = (bf16_t)(x - (float)hi);
}

template <int D, int KT>
__global__ void __launch_bounds__(256) attention_x3_kernel(AttnArgs a) {
  constexpr int KROWB = D * 2;                 // one bf16 image row of K
  constexpr int VROWB = (KT + 4) * 2;          // one bf16 image row of V^T: KT keys + 4 pad
  constexpr int NF = KT / 16, NDF = D / 16, NKK = D / 32;
  constexpr int KIMG = KT * KROWB, VIMG = D * VROWB;
  extern __shared__ __attribute__((aligned(16))) unsigned char att_smem[];
  unsigned char* const ksh = att_smem;
  unsigned char* const ksl = att_smem + KIMG;
  unsigned char* const vsh = att_smem + 2 * KIMG;
  unsigned char* const vsl = att_smem + 2 * KIMG + VIMG;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lg = lane >> 4;
  const int bh = blockIdx.y, b = bh / a.heads, h = bh - b * a.heads;
  const int d = a.d;
  const int N = a.seg_len[0] + a.seg_len[1] + a.seg_len[2];
  const int q_idx = blockIdx.x * ATT_QB + wave * 16 + lr;
  const bool q_ok = q_idx < N;
  const float sscale = a.scale * 1.44269504088896341f;

  bf16x8 qh[NKK], ql[NKK];
  {
    int seg = 0;
    int64_t r = att_row(a, b, q_ok ? q_idx : 0, &seg);
    const float* qp = (const float*)a.qkv[seg] + r * a.ld_qkv + a.q_off + h * a.hs_qkv;
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk) {
      const int c0 = kk * 32 + 8 * lg;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float v = q_ok && c0 + e < d ? qp[c0 + e] : 0.f;
        bf16_t hi, lo;
        x3_split(v, hi, lo);
        qh[kk][e] = hi;
        ql[kk][e] = lo;
      }
    }
  }
  float m_run = -INFINITY, l_run = 0.f;
  f32x4 o[NDF];
#pragma unroll
  for (int df = 0; df < NDF; ++df) o[df] = f32x4{0, 0, 0, 0};

  const int ntiles = (N + KT - 1) / KT;
  for (int kt = 0; kt < ntiles; ++kt) {
    __syncthreads();
    constexpr int SL = D / 8;   // 8-channel groups per key row
    for (int idx = tid; idx < KT * SL; idx += 256) {
      const int key = idx / SL, slot = idx - key * SL;
      const int kg = kt * KT + key;
      int seg = 0;
      int64_t r = att_row(a, b, kg < N ? kg : 0, &seg);
      const float* base = (const float*)a.qkv[seg] + r * a.ld_qkv + h * a.hs_qkv + slot * 8;
      const bool ok = kg < N && slot * 8 < d;
      f32x4 k0 = ok ? *reinterpret_cast<const f32x4*>(base + a.k_off) : f32x4{0, 0, 0, 0};
      f32x4 k1 = ok ? *reinterpret_cast<const f32x4*>(base + a.k_off + 4) : f32x4{0, 0, 0, 0};
      f32x4 v0 = ok ? *reinterpret_cast<const f32x4*>(base + a.v_off) : f32x4{0, 0, 0, 0};
      f32x4 v1 = ok ? *reinterpret_cast<const f32x4*>(base + a.v_off + 4) : f32x4{0, 0, 0, 0};
      bf16x8 kh8, kl8;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        bf16_t hi, lo;
        x3_split(e < 4 ? k0[e] : k1[e - 4], hi, lo);
        kh8[e] = hi;
        kl8[e] = lo;
        x3_split(e < 4 ? v0[e] : v1[e - 4], hi, lo);
        *reinterpret_cast<bf16_t*>(vsh + (8 * slot + e) * VROWB + key * 2) = hi;
        *reinterpret_cast<bf16_t*>(vsl + (8 * slot + e) * VROWB + key * 2) = lo;
      }
      const int sw = att_swz<D>(key, slot) * 16;
      *reinterpret_cast<bf16x8*>(ksh + key * KROWB + sw) = kh8;
      *reinterpret_cast<bf16x8*>(ksl + key * KROWB + sw) = kl8;
    }
    __syncthreads();

    f32x4 sc[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      sc[f] = f32x4{0, 0, 0, 0};
      const int row = 16 * f + lr;
#pragma unroll
      for (int kk = 0; kk < NKK; ++kk) {
        const int off = row * KROWB + att_swz<D>(row, 4 * kk + lg) * 16;
        const bf16x8 kfh = *reinterpret_cast<const bf16x8*>(ksh + off);
        const bf16x8 kfl = *reinterpret_cast<const bf16x8*>(ksl + off);
        sc[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfl, qh[kk], sc[f], 0, 0, 0);   // small terms first
        sc[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfh, ql[kk], sc[f], 0, 0, 0);
        sc[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfh, qh[kk], sc[f], 0, 0, 0);
      }
    }
    float tmax = -INFINITY;
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int kg = kt * KT + 16 * f + 4 * lg + j;
        const float v = kg < N ? sc[f][j] * sscale : -INFINITY;
        sc[f][j] = v;
        tmax = fmaxf(tmax, v);
      }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 16));
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
    const float m_new = fmaxf(m_run, tmax);
    const float alpha = exp2f(m_run - m_new);
    float psum = 0.f;
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float p = exp2f(sc[f][j] - m_new);
        sc[f][j] = p;
        psum += p;
      }
    psum += __shfl_xor(psum, 16);
    psum += __shfl_xor(psum, 32);
    l_run = l_run * alpha + psum;
    m_run = m_new;
#pragma unroll
    for (int df = 0; df < NDF; ++df) o[df] *= alpha;

#pragma unroll
    for (int u = 0; u < KT / 32; ++u) {
      bf16x8 ph, pl;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        bf16_t hi, lo;
        x3_split(sc[2 * u][j], hi, lo);
        ph[j] = hi; pl[j] = lo;
        x3_split(sc[2 * u + 1][j], hi, lo);
        ph[4 + j] = hi; pl[4 + j] = lo;
      }
#pragma unroll
      for (int df = 0; df < NDF; ++df) {
        const int voff = (16 * df + lr) * VROWB;
        bf16x8 vfh, vfl;
        {
          const bf16x4 a0 = *reinterpret_cast<const bf16x4*>(vsh + voff + (32 * u + 4 * lg) * 2);
          const bf16x4 a1 = *reinterpret_cast<const bf16x4*>(vsh + voff + (32 * u + 16 + 4 * lg) * 2);
          const bf16x4 b0 = *reinterpret_cast<const bf16x4*>(vsl + voff + (32 * u + 4 * lg) * 2);
          const bf16x4 b1 = *reinterpret_cast<const bf16x4*>(vsl + voff + (32 * u + 16 + 4 * lg) * 2);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            vfh[j] = a0[j]; vfh[4 + j] = a1[j];
            vfl[j] = b0[j]; vfl[4 + j] = b1[j];
          }
        }
        o[df] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vfl, ph, o[df], 0, 0, 0);
        o[df] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vfh, pl, o[df], 0, 0, 0);
        o[df] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vfh, ph, o[df], 0, 0, 0);
      }
    }
  }

  if (q_ok) {
    const float inv = 1.0f / l_run;
    int seg = 0;
    int64_t r = att_row(a, b, q_idx, &seg);
    float* op = (float*)a.out[seg] + r * a.ld_out + h * a.hs_out;
#pragma unroll
    for (int df = 0; df < NDF; ++df) {
      const int c0 = 16 * df + 4 * lg;
      if (c0 >= d) continue;
      *reinterpret_cast<f32x4*>(op + c0) = o[df] * inv;
    }
  }
}

template <int D>
static int launch_attention_x3(const AttnArgs& a, hipStream_t s) {
  constexpr int KT = 64;
  constexpr int lds = 2 * KT * D * 2 + 2 * D * (KT + 4) * 2;
  auto kern = attention_x3_kernel<D, KT>;
  static bool attr_set = false;
  if (!attr_set && lds > 48 * 1024) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    attr_set = true;
  }
  const int N = a.seg_len[0] + a.seg_len[1] + a.seg_len[2];
  hipLaunchKernelGGL(kern, dim3(cdiv(N, ATT_QB), a.B * a.heads), dim3(256), lds, s, a);
  HIP_TRY(hipGetLastError());
  return DDMI_OK;
}

template <int D, int QF>
static int launch_attention_v2(const AttnArgs& a_in, hipStream_t s) {
  constexpr int lds = 4 * 64 * D * 2;
  auto kern1 = attention_v2_kernel<D, QF, false>;
  auto kern3 = attention_v2_kernel<D, QF, true>;
  auto kern1l = attention_v2_kernel<D, QF, false, true>;
  auto kern3l = attention_v2_kernel<D, QF, true, true>;
  static bool attr_set = false;
  if (!attr_set && lds > 48 * 1024) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern1), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern3), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern1l), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern3l), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    attr_set = true;
  }
  const int N = a_in.seg_len[0] + a_in.seg_len[1] + a_in.seg_len[2];
  // 1-D grid, XCD-aware order (top of the kernel): by sample when the batch fills the 8 XCDs evenly, else by (sample, head)
  AttnArgs a = a_in;
  const int tiles = cdiv(N, 64 * QF), BH = a.B * a.heads;
  a.xcd_map = (a.B % 8 == 0 || a.B >= 64) ? 0 : BH >= 8 ? 1 : 2;
  if (const char* e = getenv("DDMI_ATTN_XCDMAP")) a.xcd_map = atoi(e);   // diagnostics (tools/bench_attn_afhq.py): 2 = plain order
  dim3 grid(a.xcd_map == 0 ? tiles * a.heads * (cdiv(a.B, 8) * 8) : a.xcd_map == 1 ? tiles * (cdiv(BH, 8) * 8) : tiles * BH);
  static const bool no_lazy = getenv("DDMI_ATTN_LAZY") && atoi(getenv("DDMI_ATTN_LAZY")) == 0;   // diagnostics: the eager-maximum loop
  const bool lazy = a.q_prescaled && !no_lazy;
  if (a.seg_len[1] + a.seg_len[2] == 0) hipLaunchKernelGGL(lazy ? kern1l : kern1, grid, dim3(256), lds, s, a);
  else hipLaunchKernelGGL(lazy ? kern3l : kern3, grid, dim3(256), lds, s, a);
  HIP_TRY(hipGetLastError());
  return DDMI_OK;
}

static bool attention_res_ok(const AttnArgs& a, int N) {
  static const bool off = getenv("DDMI_ATTN_RES") && atoi(getenv("DDMI_ATTN_RES")) == 0;   // diagnostics: the tiled kernel everywhere
  if (off || !a.q_prescaled || a.lse || !a.vec_ok || (a.d != 16 && a.d != 32)) return false;
  // N >= 1024: at 512 tokens a workgroup is one round of four fragments and the K / V fill is not amortised (d = 32, N = 512, B = 256 x 16
  // heads: 339 us against the tiled kernel's 319)
  if (N < 1024 || N % 512 != 0 || (int64_t)N * a.d * 4 > 128 * 1024) return false;
  for (int s = 0; s < 3; ++s)
    if (a.seg_len[s] % 64 != 0 || (int64_t)a.seg_len[s] * a.ld_qkv * 2 > 0x7FFFFFFFll) return false;
  return a.ld_qkv % 8 == 0 && a.hs_qkv % 8 == 0 && a.q_off % 8 == 0 && a.k_off % 8 == 0 && a.v_off % 8 == 0;
}

static int launch_attention_res16(const AttnArgs& a, hipStream_t s) {
  const int N = a.seg_len[0] + a.seg_len[1] + a.seg_len[2];
  const int lds = 2 * N * 32 + 256 + 1280;
  auto kern1 = attention_res16_kernel<false>;
  auto kern3 = attention_res16_kernel<true>;
  static bool attr_set = false;
  if (!attr_set) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern1), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024 + 256 + 1280));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern3), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024 + 256 + 1280));
    attr_set = true;
  }
  dim3 grid(a.heads * cdiv(a.B, 8) * 8);
  if (a.seg_len[1] + a.seg_len[2] == 0) hipLaunchKernelGGL(kern1, grid, dim3(512), lds, s, a);
  else hipLaunchKernelGGL(kern3, grid, dim3(512), lds, s, a);
  HIP_TRY(hipGetLastError());
  return DDMI_OK;
}

template <int D>
static int launch_attention_res(const AttnArgs& a, hipStream_t s) {
  const int N = a.seg_len[0] + a.seg_len[1] + a.seg_len[2];
  const int lds = 2 * N * D * 2 + 64;
  auto kern1 = attention_res_kernel<D, false>;
  auto kern3 = attention_res_kernel<D, true>;
  static bool attr_set = false;
  if (!attr_set) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern1), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024 + 64));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern3), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024 + 64));
    attr_set = true;
  }
  dim3 grid(a.heads * cdiv(a.B, 8) * 8);
  if (a.seg_len[1] + a.seg_len[2] == 0) hipLaunchKernelGGL(kern1, grid, dim3(512), lds, s, a);
  else hipLaunchKernelGGL(kern3, grid, dim3(512), lds, s, a);
  HIP_TRY(hipGetLastError());
  return DDMI_OK;
}

static int attention_stream_qf(int d) { return d <= 64 ? 4 : d == 128 ? 2 : 1; }
// the (head dim, token count) pairs the streaming kernel serves, switches included - ONE predicate for the launcher below and for the
// graph builders that choose the flash form only where this kernel will run (models.hip build_vae_attn)
bool attention_stream_shape_ok(int d, int N) {
  static const bool off = getenv("DDMI_ATTN_STREAM") && atoi(getenv("DDMI_ATTN_STREAM")) == 0;   // diagnostics: the tiled kernels
  static const bool big_off = getenv("DDMI_ATTN_STREAM_BIG") && atoi(getenv("DDMI_ATTN_STREAM_BIG")) == 0;   // ... for d = 256 / 512 only
  if (off || (d != 32 && d != 64 && d != 128 && d != 256 && d != 512)) return false;
  if (d >= 256 && big_off) return false;
  return N >= 2048 && N % (128 * attention_stream_qf(d)) == 0;
}
static bool attention_stream_ok(const AttnArgs& a, int N) {
  if (!a.q_prescaled || a.lse || !a.vec_ok || !attention_stream_shape_ok(a.d, N)) return false;
  for (int s = 0; s < 3; ++s)
    if (a.seg_len[s] % 64 != 0 || (int64_t)a.seg_len[s] * a.ld_qkv * 2 > 0x7FFFFFFFll) return false;
  return a.ld_qkv % 8 == 0 && a.hs_qkv % 8 == 0 && a.q_off % 8 == 0 && a.k_off % 8 == 0 && a.v_off % 8 == 0;
}

template <int D, int QF, int KT, int NS>
static int launch_attention_stream(const AttnArgs& a_in, hipStream_t s) {
  constexpr int lds = NS * 2 * KT * D * 2 + 64;
  static_assert(lds <= 160 * 1024, "streaming attention: the stage ring exceeds the LDS");
  auto kern1 = attention_stream_kernel<D, QF, KT, NS, false>;
  auto kern3 = attention_stream_kernel<D, QF, KT, NS, true>;
  static bool attr_set = false;
  if (!attr_set) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern1), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern3), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    attr_set = true;
  }
  const int N = a_in.seg_len[0] + a_in.seg_len[1] + a_in.seg_len[2];
  AttnArgs a = a_in;
  const int tiles = N / (128 * QF), BH = a.B * a.heads;
  a.xcd_map = (a.B % 8 == 0 || a.B >= 64) ? 0 : BH >= 8 ? 1 : 2;
  dim3 grid(a.xcd_map == 0 ? tiles * a.heads * (cdiv(a.B, 8) * 8) : a.xcd_map == 1 ? tiles * (cdiv(BH, 8) * 8) : tiles * BH);
  if (a.seg_len[1] + a.seg_len[2] == 0) hipLaunchKernelGGL(kern1, grid, dim3(512), lds, s, a);
  else hipLaunchKernelGGL(kern3, grid, dim3(512), lds, s, a);
  HIP_TRY(hipGetLastError());
  return DDMI_OK;
}

template <typename T, int D, int DV, int KT>
static int launch_attention_t(const AttnArgs& a, hipStream_t s) {
  constexpr bool IS_BF16 = sizeof(T) == 2;
  constexpr int lds = KT * (IS_BF16 ? D * 2 : D * 4 + 16) + DV * (KT + 4) * (int)sizeof(T);
  static_assert(lds <= 160 * 1024, "attention tiles exceed the LDS");
  auto kern = attention_mh_kernel<T, D, DV, KT>;
  static bool attr_set = false;
  if (!attr_set && lds > 48 * 1024) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    attr_set = true;
  }
  const int N = a.seg_len[0] + a.seg_len[1] + a.seg_len[2];
  dim3 grid(cdiv(N, ATT_QB), a.B * a.heads, cdiv(a.d, DV));
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, a);
  HIP_TRY(hipGetLastError());
  return DDMI_OK;
}

int launch_attention(const AttnArgs& a_in, int dt, hipStream_t s) {
  AttnArgs a = a_in;
  const int N = a.seg_len[0] + a.seg_len[1] + a.seg_len[2];
  DDMI_REQUIRE(a.B > 0 && N > 0 && a.heads > 0 && a.d > 0 && a.d <= 512, "attention: bad shape (head dim must be <= 512)");
  if (a.q_prescaled) {   // the producer of q folded scale x log2 e into it: every bf16 kernel's base-2 exponent is the score itself
    DDMI_REQUIRE(dt == DT_BF16, "attention: pre-scaled q is a bf16-mode arrangement");
    a.scale = 0.69314718055994531f;   // x log2 e = 1
  }
  DDMI_REQUIRE(a.qkv[0] && a.out[0], "attention: null tensor");
  const int vn = dt == DT_BF16 ? 8 : 4;
  // 16-byte vector loads / 8-16-byte stores need every row, head and q/k/v block to start on a vector boundary
  a.vec_ok = (a.d % vn == 0 && a.ld_qkv % vn == 0 && a.hs_qkv % vn == 0 && a.q_off % vn == 0 && a.k_off % vn == 0 &&
              a.v_off % vn == 0 && a.ld_out % 4 == 0 && a.hs_out % 4 == 0) ? 1 : 0;
  // DDMI_PROF_SHAPES >= 1 tags the profiler scope with the launch's shape (tools/prof_attn_shapes.py; bench.py's sky leg, whose
  // "attention_mh" is five kernels on shapes three orders of magnitude apart: its roofline entry names ONE of them)
  // (read per profiled launch, not once: bench.py switches it on for one workload leg of a longer process; 2 = attention only)
  char pname[96];
  snprintf(pname, sizeof pname, "%s", a.legacy_name ? "attention_legacy" : "attention_mh");
  if (prof_active() && getenv("DDMI_PROF_SHAPES") && atoi(getenv("DDMI_PROF_SHAPES")) >= 1)
    snprintf(pname, sizeof pname, "%s[B%d,h%d,d%d,N%d%s]", a.legacy_name ? "attention_legacy" : "attention_mh", a.B, a.heads, a.d, N,
             a.seg_len[1] + a.seg_len[2] ? ",seg3" : "");
  PROF_SCOPE(pname, 4.0 * a.B * a.heads * (double)N * N * a.d, 4.0 * a.B * a.heads * N * a.d * dt_size(dt), s);
  static const int use_v2 = getenv("DDMI_ATTN_V1") ? 0 : 1;  // diagnostics: DDMI_ATTN_V1=1 pins the generic kernel
  if (dt == DT_BF16 && use_v2 && attention_stream_ok(a, N)) {
    if (a.d == 32) return launch_attention_stream<32, 4, 64, 3>(a, s);   // (N >= 2048 only: the resident kernel below serves d = 32 up to 1024 tokens)
    if (a.d == 64) return launch_attention_stream<64, 4, 64, 3>(a, s);
    if (a.d == 128) return launch_attention_stream<128, 2, 64, 3>(a, s);
    if (a.d == 256) return launch_attention_stream<256, 1, 32, 3>(a, s);
    return launch_attention_stream<512, 1, 32, 2>(a, s);
  }
  if (dt == DT_BF16 && a.vec_ok && a.d % 4 == 0 && a.d <= 128 && use_v2) {
    const bool big = N >= 128;
    // small-batch regime (net.h): the resident kernel is one workgroup per (sample, head) - 14 workgroups for a 32 x 32 AttentionBlock at
    // B = 1, 40 us each; the tiled kernel below splits the queries of a head over workgroups
    const bool few = net_small_batch_enabled() && a.B * a.heads < 128 && !a.lse;
    if (attention_res_ok(a, N) && !few) {
      static const bool res16_off = getenv("DDMI_ATTN_RES16") && atoi(getenv("DDMI_ATTN_RES16")) == 0;   // diagnostics: the 16 x 16 x 32 form at d = 16
      if (a.d == 16 && N % 1024 == 0 && !res16_off) return launch_attention_res16(a, s);
      return a.d == 16 ? launch_attention_res<16>(a, s) : launch_attention_res<32>(a, s);
    }
    if (a.d <= 32) return big ? launch_attention_v2<32, 2>(a, s) : launch_attention_v2<32, 1>(a, s);
    if (a.d <= 64) return big ? launch_attention_v2<64, 2>(a, s) : launch_attention_v2<64, 1>(a, s);
    return big ? launch_attention_v2<128, 2>(a, s) : launch_attention_v2<128, 1>(a, s);
  }
  DDMI_REQUIRE(!a.lse, "attention: the log-sum-exp output exists on the bf16 production kernel only");
  static const bool x3_attn_off = getenv("DDMI_X3_ATTN") && atoi(getenv("DDMI_X3_ATTN")) == 0;   // diagnostics: the fp32-MFMA attention in bf16x3 handles
  if (dt == DT_F32 && a.x3 && !x3_attn_off && a.d % 8 == 0 && a.d <= 128 && a.ld_qkv % 4 == 0 && a.hs_qkv % 4 == 0 && a.q_off % 4 == 0 &&
      a.k_off % 4 == 0 && a.v_off % 4 == 0 && a.ld_out % 4 == 0 && a.hs_out % 4 == 0) {
    if (a.d <= 32) return launch_attention_x3<32>(a, s);
    if (a.d <= 64) return launch_attention_x3<64>(a, s);
    return launch_attention_x3<128>(a, s);
  }
  if (dt == DT_BF16) {
    if (a.d <= 32) return launch_attention_t<bf16_t, 32, 32, 64>(a, s);
    if (a.d <= 64) return launch_attention_t<bf16_t, 64, 64, 64>(a, s);
    if (a.d <= 128) return launch_attention_t<bf16_t, 128, 128, 64>(a, s);
    if (a.d <= 256) return launch_attention_t<bf16_t, 256, 256, 64>(a, s);
    return launch_attention_t<bf16_t, 512, 256, 64>(a, s);
  }
  if (a.d <= 32) return launch_attention_t<float, 32, 32, 64>(a, s);
  if (a.d <= 64) return launch_attention_t<float, 64, 64, 64>(a, s);
  if (a.d <= 128) return launch_attention_t<float, 128, 128, 64>(a, s);
  if (a.d <= 256) return launch_attention_t<float, 256, 128, 32>(a, s);
  return launch_attention_t<float, 512, 128, 32>(a, s);
}

// QKVAttentionLegacy layout: qkv [B, T, heads * 3d] with per-head [q | k | v] channel blocks; out [B, T, heads * d]
int launch_attention_legacy(const void* qkv, void* out, int dt, int B, int T, int heads, int d, hipStream_t s, float* lse, float scale) {
  AttnArgs a{};
  a.lse = lse;
  a.qkv[0] = qkv; a.out[0] = out;
  a.seg_len[0] = T;
  a.ld_qkv = heads * 3 * d; a.ld_out = heads * d;
  a.q_off = 0; a.k_off = d; a.v_off = 2 * d;
  a.hs_qkv = 3 * d; a.hs_out = d;
  a.d = d; a.heads = heads; a.B = B;
  a.scale = scale > 0.f ? scale : 1.0f / sqrtf((float)d);  // (d^-1/4 q) . (d^-1/4 k), openaimodel.py:413-416; `scale`: a zero-padded narrower head
  a.legacy_name = 1;
  return launch_attention(a, dt, s);
}

// =====================================================================================================
// row softmax (in place), one wave per row, fp32 math.  Used by the decoder's single-head 512-channel attention
// (autoencoder_unet.py:187-215), which runs as GEMM -> softmax -> GEMM once per sample.
// =====================================================================================================
template <typename T>
__global__ void __launch_bounds__(256) softmax_rows_kernel(T* __restrict__ x, int64_t rows, int cols) {
  int lane = threadIdx.x & 63;
  int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  T* p = x + row * cols;
  float m = -INFINITY;
  for (int c = lane; c < cols; c += 64) m = fmaxf(m, to_f32(p[c]));
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
  float sum = 0.f;
  for (int c = lane; c < cols; c += 64) sum += expf(to_f32(p[c]) - m);
  for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
  float inv = 1.0f / sum;
  for (int c = lane; c < cols; c += 64) p[c] = from_f32<T>(expf(to_f32(p[c]) - m) * inv);
}

int launch_softmax_rows(void* x, int dt, int64_t rows, int cols, hipStream_t s) {
  dim3 grid((unsigned)cdiv64(rows, 4));
  if (dt == DT_BF16)
    hipLaunchKernelGGL(softmax_rows_kernel<bf16_t>, grid, dim3(256), 0, s, (bf16_t*)x, rows, cols);
  else
    hipLaunchKernelGGL(softmax_rows_kernel<float>, grid, dim3(256), 0, s, (float*)x, rows, cols);
  HIP_TRY(hipGetLastError());
  return DDMI_OK;
}

// dst[b][c][r] = src[b][r][c]
template <typename T>
__global__ void __launch_bounds__(256) transpose_2d_kernel(const T* __restrict__ src, T* __restrict__ dst, int R, int Cc) {
  __shared__ float tile[32][33];
  int b = blockIdx.z;
  int r0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const T* s = src + (int64_t)b * R * Cc;
  T* d = dst + (int64_t)b * R * Cc;
  for (int i = ty; i < 32; i += 8) {
    int r = r0 + i, c = c0 + tx;
    tile[i][tx] = (r < R && c < Cc) ? to_f32(s[(int64_t)r * Cc + c]) : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    int c = c0 + i, r = r0 + tx;
    if (r < R && c < Cc) d[(int64_t)c * R + r] = from_f32<T>(tile[tx][i]);
  }
}

int launch_transpose_2d(const void* src, void* dst, int dt, int batch, int R, int Cc, hipStream_t s) {
  dim3 grid(cdiv(R, 32), cdiv(Cc, 32), batch);
  if (dt == DT_BF16)
    hipLaunchKernelGGL(transpose_2d_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)src, (bf16_t*)dst, R, Cc);
  else
    hipLaunchKernelGGL(transpose_2d_kernel<float>, grid, dim3(256), 0, s, (const float*)src, (float*)dst, R, Cc);
  HIP_TRY(hipGetLastError());
  return DDMI_OK;
}
