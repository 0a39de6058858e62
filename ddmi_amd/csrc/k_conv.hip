// Implicit-GEMM convolution on MFMA for gfx950 (the UNet / decoder hot op).
//
//   y[m][n] = sum_{tap, c} x[pixel(m) + tap][c] * w[n][tap][c]        m = (b, oh, ow),  n = output channel
//
// * activations NHWC, weights packed [Cout][tap][Cin]: both GEMM operands are K-contiguous, so one 16-byte load
//   per lane feeds an MFMA fragment directly (no transposes anywhere).
// * K is walked in chunks of 32 channels of one tap, KPS chunks per pipeline stage; the im2col gather (padding,
//   stride 2, fused nearest x2 upsample) is folded into the per-lane SOURCE offset of the loads and never
//   materialised.  Loads go through buffer descriptors: an out-of-image tap, a channel tail or a row past M is an
//   out-of-range offset, which the hardware answers with zeros - no branches around loads, no select afterwards.
// * bf16 path: `buffer_load_dwordx4 ... lds` (LDS-DMA): tiles land in LDS without touching VGPRs.  One DMA
//   wave-instruction writes 1 KiB lane-linearly = 16 rows x 64 B; the 16-byte-slot XOR swizzle that makes the
//   ds_read_b128 fragment reads conflict-free is applied to the per-lane source address instead (swz()).
//   Two LDS stages: DMA of stage s+1 is in flight while the MFMAs of stage s run; one barrier per stage.
//   fp32 (parity) path: register staging (buffer_load -> ds_write), v_mfma_f32_16x16x4_f32 = exact fp32 FMA chains.
// * the (tap, channel) walk is kept in wave-uniform scalar state, so per load only one add + range compares remain.
// * operands are swapped (A-operand = weights, B-operand = pixels) so each lane ends up with 4 consecutive output
//   channels of one pixel: bias / time-embedding / residual / store are 8- or 16-byte vector accesses.
// * epilogue fuses + bias[n] + emb[b][n] + residual, * out_scale, optional fp32-NCHW store, optional split-K
//   (fp32 partial slabs [z][M][Cout] written with plain 16-byte stores, summed by splitk_finish_kernel).
#include <cstdlib>
#include "kernels.h"
#include "prof.h"

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((address_space(3))) void lds_void;

struct ConvK {
  const void* x;
  const void* w;
  const float* bias;
  const float* emb;
  const void* res;
  void* y;
  float* y_nchw;
  float* ws;
  float* chstat;   // optional [M / (16 FM)][Cout][2] per-channel (sum, sum of squares) of the STORED values per wave row block
  int B, Hin, Win, Cin, Hout, Wout, Cout;
  int KH, KW, stride, pad, ups;
  int emb_stride, emb_mod;
  int M;        // rows per batch entry
  int CPT;      // chunks per tap = ceil(Cin / 32)
  int NK;       // total chunks
  int NS;       // total stages
  int SPS;      // stages per split
  int tiles_n;
  int up4;                                  // x2 nearest upsample + 3x3 as four 2x2 parity convolutions (blockIdx.y = 2 dy + dx)
  int tile_map;                             // 0 row-major, 1 XCD-contiguous m-major, 2 XCD-contiguous n-major
  int paired;                               // pipe kernel: paired-fragment instantiation (order 3 / 4)
  int chan_major;                           // K walk order (ring kernel): taps innermost when set
  int dbg;                                  // diagnostics (DDMI_CONV_DBG): 1 no DMA, 2 no ds_read/MFMA, 4 no barrier, 8 no epilogue, 16 general epilogue only
  float out_scale;
  unsigned long long* ts;                   // diagnostics (DDMI_CONV_TS): per-workgroup phase timestamps (ring and halo kernels)
  uint64_t x_bytes;                         // activation extent in bytes (per batch entry); may exceed 2 GiB
  unsigned w_bytes;                         // weight extent (per batch entry) for the range check
  int64_t x_bstride, w_bstride, y_bstride;  // element strides for blockIdx.y (batched GEMM mode)
  // folded 1x1 skip connection (halo kernel, SKIP instantiation): see ConvArgs
  const void* x2;
  const void* x3;
  const void* w2;
  int C2, C3;
  unsigned w2_bytes;
  // bf16x3 numerics: the bf16 kernels finish in fp32 themselves - fp32 NHWC output and fp32 residual (general epilogue)
  float* y32;
  const float* res32;
  // 1x1 convolution over the channel concatenation [x | xcat] read in place (persistent 1x1 kernel, CAT instantiation): x has C1
  // channels per pixel, xcat Cin - C1; w is [Cout][Cin]
  const void* xcat;
  int C1;
  // producer-side GroupNorm (halo kernel, FGN instantiation; see conv_epilogue_fgn): y = [SiLU](GroupNorm32(conv + bias + emb))
  const float* gn_gamma;
  const float* gn_beta;
  float gn_eps;
  int gn_silu, gn_cpg, gn_tps;              // channels per group, 256-row tiles per sample
  float* gn_part;                           // [B * tps][32 groups] 16-byte granules {mean_t, M2_t, tag, tag}
  unsigned* gn_sync;                        // [0] launch epoch (tag = epoch), [1] error flag (a poll timed out)
  unsigned* gn_flag_host;                   // host-mapped copy of the error flag (the host peeks it without a synchronisation), or null
};

template <typename T> struct Tile;
template <> struct Tile<bf16_t> {
  static constexpr int EPS = 8;     // elements per 16-byte slot
  static constexpr int SLOTS = 4;   // slots per 32-element chunk row
  static constexpr int ROWB = 64;   // LDS bytes per row (DMA needs rows contiguous)
};
template <> struct Tile<float> {
  static constexpr int EPS = 4;
  static constexpr int SLOTS = 8;
  static constexpr int ROWB = 144;  // 128 + 16 pad (register staging)
};

// 16-byte slot swizzle for 64-byte rows: slot ^ {0,2,3,1}[(row>>2)&3]; with it the four ds_read_b128 lane groups
// {0-3,12-15,20-27}, {4-11,16-19,28-31}, ... each touch 16 distinct slots of the 256-byte bank row.
__device__ __forceinline__ int swz(int row, int slot) {
  int r = (row >> 2) & 3;
  return slot ^ ((0x78 >> (r * 2)) & 3);
}

#define OOB_OFFSET 0x80000000u

// ---- blockIdx.x -> (tile_m, tile_n), XCD-aware ---------------------------------------------------------------
// Workgroups are dealt round-robin to the 8 XCDs (blockIdx % 8), each with its own 4 MiB L2.  `tile_map` 1/2 gives the
// workgroups of one XCD a CONTIGUOUS range of logical tile ids, walked m-major (1: neighbours share activation rows,
// each XCD streams all weights) or n-major (2: neighbours share weights - the 8x8 / 16x16 levels whose weights
// outweigh their activations), so an operand is pulled into one L2 instead of several.  0 = plain row-major.
struct TileMN { int m, n; };
__device__ __forceinline__ TileMN conv_tile_of_block(int tile_map, int tiles_n) {
  int id = blockIdx.x;
  if (tile_map) {
    const int nb = gridDim.x, q = nb >> 3, r = nb & 7, x = id & 7;
    id = x * q + min(x, r) + (id >> 3);
  }
  TileMN t;
  if (tile_map == 2) {
    const int tiles_m = gridDim.x / tiles_n;
    t.n = id / tiles_m;
    t.m = id - t.n * tiles_m;
  } else {
    t.m = id / tiles_n;
    t.n = id - t.m * tiles_n;
  }
  return t;
}

// ---- x2 nearest upsample fused into a 3x3 convolution, computed on the SOURCE grid ---------------------------------
// Output pixel (2y+dy, 2x+dx) of conv3x3(upsample2(x)) only ever sees the 2x2 source neighbourhood rows {y-1+dy, y+dy},
// columns {x-1+dx, x+dx}: the three taps of each axis collapse onto two source pixels.  So the op is four 2x2
// convolutions over the source image - one per output parity, weights pre-summed (conv_pack_up4_kernel) - with
// 16 tap-products per 2x2 output block instead of 36: 2.25x fewer FLOPs, same result up to the rounding of the summed
// weights.  blockIdx.y is the parity; M counts source pixels; the epilogue scatters rows to (2y+dy, 2x+dx).
__device__ __forceinline__ int conv_pad_h(const ConvK& a) { return a.up4 ? a.pad - (int)(blockIdx.y >> 1) : a.pad; }
__device__ __forceinline__ int conv_pad_w(const ConvK& a) { return a.up4 ? a.pad - (int)(blockIdx.y & 1) : a.pad; }
__device__ __forceinline__ int64_t conv_out_row(const ConvK& a, int m, int b, int hw) {
  if (!a.up4) return m;
  const int r = m - b * hw, y = r / a.Wout, x = r - y * a.Wout;
  return ((int64_t)b * (2 * a.Hout) + 2 * y + (int)(blockIdx.y >> 1)) * (2 * a.Wout) + 2 * x + (int)(blockIdx.y & 1);
}

// ---- fragment pairing ---------------------------------------------------------------------------------------------
// An MFMA 16x16 result leaves lane (lr, lg) with rows 4lg..4lg+3 = four output channels.  With PAIRED the weight rows
// fed to fragments 2p and 2p+1 are interleaved in groups of four, so those two fragments hand the lane EIGHT consecutive
// channels: every epilogue access (bias, embedding, residual, store) becomes 16 bytes wide and there are half as many
// of them - the store tail of a tile is issue-bound (MI355X_MICROARCH.md, attention epilogue row).
//   weight row of (fragment j, MFMA row r): 32 (j>>1) + 8 (r>>2) + 4 (j&1) + (r&3);  an unpaired last fragment keeps 16 j + r
template <int FN, bool PAIRED> __device__ __forceinline__ int frag_row(int j, int r) {
  if (!PAIRED || ((FN & 1) && j == FN - 1)) return 16 * j + r;
  return 32 * (j >> 1) + 8 * (r >> 2) + 4 * (j & 1) + (r & 3);
}
// The pairing is realised where the weight tile is LOADED: LDS row R of the tile holds weight row lds_row_chan(R), so
// the fragment reads (16 consecutive LDS rows, conflict-free swizzle) stay exactly as they are.  R = wn*TN + l with l
// inside the wave's slab: rows of the paired part are permuted inside 32-row blocks, 32p + 16h + 4q + t -> 32p + 8q + 4h + t.
template <int TN> __device__ __forceinline__ int lds_row_chan(int R) {
  constexpr int FN = TN / 16;
  const int wn = R / TN, l = R - wn * TN;
  if (l >= 32 * (FN / 2)) return R;
  const int r32 = l & 31;
  return R - r32 + 8 * ((r32 >> 2) & 3) + 4 * (r32 >> 4) + (r32 & 3);
}
// first channel (relative to the wave's slab) of the lane's four values in fragment j
template <int FN, bool PAIRED> __device__ __forceinline__ int frag_chan(int j, int lg) {
  if (!PAIRED || ((FN & 1) && j == FN - 1)) return 16 * j + 4 * lg;
  return 32 * (j >> 1) + 8 * lg + 4 * (j & 1);
}

// ---- epilogue shared by all kernels --------------------------------------------------------------------------------
template <typename T, int NV>
__device__ __forceinline__ void conv_epilogue_vec(const ConvK& a, float (&v)[NV], int m, int n, int b, int be, int hw,
                                                  T* __restrict__ yg, const T* __restrict__ rg) {
  constexpr bool IS_BF16 = sizeof(T) == 2;
  if (a.ws != nullptr) {  // split-K partial slab [z][M][Cout], summed by splitk_finish_kernel
    float* p = a.ws + ((int64_t)blockIdx.z * a.M + m) * a.Cout + n;
#pragma unroll
    for (int q = 0; q < NV; q += 4) *reinterpret_cast<f32x4*>(p + q) = f32x4{v[q], v[q + 1], v[q + 2], v[q + 3]};
    return;
  }
  if (a.bias) {
#pragma unroll
    for (int q = 0; q < NV; q += 4) {
      f32x4 t = *reinterpret_cast<const f32x4*>(a.bias + n + q);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[q + e] += t[e];
    }
  }
  if (a.emb) {
    const float* pe = a.emb + (int64_t)be * a.emb_stride + n;
#pragma unroll
    for (int q = 0; q < NV; q += 4) {
      f32x4 t = *reinterpret_cast<const f32x4*>(pe + q);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[q + e] += t[e];
    }
  }
  if (rg) {
    const T* p = rg + (int64_t)m * a.Cout + n;
    if constexpr (IS_BF16 && NV == 8) {
      bf16x8 t = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += (float)t[e];
    } else if constexpr (IS_BF16) {
      bf16x4 t = *reinterpret_cast<const bf16x4*>(p);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] += (float)t[e];
    } else {
#pragma unroll
      for (int q = 0; q < NV; q += 4) {
        f32x4 t = *reinterpret_cast<const f32x4*>(p + q);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[q + e] += t[e];
      }
    }
  }
  if (a.res32) {
    const float* p = a.res32 + (int64_t)m * a.Cout + n;
#pragma unroll
    for (int q = 0; q < NV; q += 4) {
      f32x4 t = *reinterpret_cast<const f32x4*>(p + q);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[q + e] += t[e];
    }
  }
#pragma unroll
  for (int e = 0; e < NV; ++e) v[e] *= a.out_scale;
  if (a.y_nchw) {
    int r = m - b * hw;
#pragma unroll
    for (int e = 0; e < NV; ++e) a.y_nchw[((int64_t)b * a.Cout + n + e) * hw + r] = v[e];
    return;
  }
  if (a.y32) {
    float* p = a.y32 + conv_out_row(a, m, b, hw) * a.Cout + n;
#pragma unroll
    for (int q = 0; q < NV; q += 4) *reinterpret_cast<f32x4*>(p + q) = f32x4{v[q], v[q + 1], v[q + 2], v[q + 3]};
    return;
  }
  T* p = yg + conv_out_row(a, m, b, hw) * a.Cout + n;
  if constexpr (IS_BF16 && NV == 8) {
    bf16x8 t;
#pragma unroll
    for (int e = 0; e < 8; ++e) t[e] = (bf16_t)v[e];
    *reinterpret_cast<bf16x8*>(p) = t;
  } else if constexpr (IS_BF16) {
    bf16x4 t;
#pragma unroll
    for (int e = 0; e < 4; ++e) t[e] = (bf16_t)v[e];
    *reinterpret_cast<bf16x4*>(p) = t;
  } else {
#pragma unroll
    for (int q = 0; q < NV; q += 4) *reinterpret_cast<f32x4*>(p + q) = f32x4{v[q], v[q + 1], v[q + 2], v[q + 3]};
  }
}

// The common case of the bf16 kernels - plain NHWC output, the wave's whole channel range exists, all of the wave's rows
// in one sample - without the per-element generality of the path below.  That path waits for every bias / embedding /
// residual load right where it is issued (48 dependent round trips per wave) and re-derives every address: measured with
// DDMI_CONV_TS on the 448->1344 1x1 conv, its epilogue took 4.2 us of a 24 us tile BEFORE any store was issued.  Here the
// per-channel constant (bias + the sample's embedding row) is loaded once per wave, the residual rows of fragment row
// i+1 are in flight while row i is converted and stored, and there is one 64-bit address per fragment row.
// BIAS_LDS (persistent 1x1 kernel): the bias vector sits in LDS (`bias_lds`, indexed by output channel) - a global load in the
// epilogue would queue behind the next tile's LDS-DMA (vmcnt retires in order) and stall it for an L2 round trip per tile
template <int FM, int FN, bool STATS = true, bool BIAS_LDS = false>
__device__ __forceinline__ void conv_epilogue_fast(const ConvK& a, f32x4 (&acc)[FM][FN], int m_base, int n_base, int lr,
                                                   int lg, int hw, const float* bias_lds = nullptr) {
  using T = bf16_t;
  constexpr int NPAIR = FN / 2;
  constexpr bool ODD = (FN & 1) != 0;
  if (m_base >= a.M) return;
  T* __restrict__ yg = (T*)a.y + (int64_t)blockIdx.y * a.y_bstride;
  const T* __restrict__ rg = (const T*)a.res;
  const int n_pair = n_base + 8 * lg;                             // + 32 p: eight consecutive channels of pair p
  const int n_odd = n_base + frag_chan<FN, true>(FN - 1, lg);     // the unpaired last fragment: four channels

  f32x4 cb[FN];
#pragma unroll
  for (int j = 0; j < FN; ++j) cb[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  if constexpr (BIAS_LDS) {
#pragma unroll
    for (int p = 0; p < NPAIR; ++p) {
      cb[2 * p] = *reinterpret_cast<const f32x4*>(bias_lds + n_pair + 32 * p);
      cb[2 * p + 1] = *reinterpret_cast<const f32x4*>(bias_lds + n_pair + 32 * p + 4);
    }
    if constexpr (ODD) cb[FN - 1] = *reinterpret_cast<const f32x4*>(bias_lds + n_odd);
  } else if (a.bias) {
#pragma unroll
    for (int p = 0; p < NPAIR; ++p) {
      cb[2 * p] = *reinterpret_cast<const f32x4*>(a.bias + n_pair + 32 * p);
      cb[2 * p + 1] = *reinterpret_cast<const f32x4*>(a.bias + n_pair + 32 * p + 4);
    }
    if constexpr (ODD) cb[FN - 1] = *reinterpret_cast<const f32x4*>(a.bias + n_odd);
  }
  if (a.emb) {
    int be = m_base / hw;   // wave-uniform: the caller checked that the wave's rows share a sample
    if (a.emb_mod) {
      be -= be >= a.emb_mod ? a.emb_mod : 0;
      be -= be >= a.emb_mod ? a.emb_mod : 0;
    }
    const float* pe = a.emb + (int64_t)be * a.emb_stride;
#pragma unroll
    for (int p = 0; p < NPAIR; ++p) {
      cb[2 * p] += *reinterpret_cast<const f32x4*>(pe + n_pair + 32 * p);
      cb[2 * p + 1] += *reinterpret_cast<const f32x4*>(pe + n_pair + 32 * p + 4);
    }
    if constexpr (ODD) cb[FN - 1] += *reinterpret_cast<const f32x4*>(pe + n_odd);
  }

  // GroupNorm statistics of the consumer, gathered here so that no separate pass has to read the tensor back: per channel the
  // sum and the sum of squares of the values AS STORED (bf16-rounded) over this wave's 16 FM rows, fp32 (<= 64 terms each),
  // written per row block; launch_gn_chstat_finish adds the blocks of a sample in fp64 in a fixed order (deterministic).
  const bool st = STATS && a.chstat != nullptr;
  float s1p[NPAIR > 0 ? NPAIR : 1][8], s2p[NPAIR > 0 ? NPAIR : 1][8], s1o[4], s2o[4];
#pragma unroll
  for (int p = 0; p < (NPAIR > 0 ? NPAIR : 1); ++p)
#pragma unroll
    for (int e = 0; e < 8; ++e) s1p[p][e] = s2p[p][e] = 0.f;
#pragma unroll
  for (int e = 0; e < 4; ++e) s1o[e] = s2o[e] = 0.f;

  bf16x8 rp[2][NPAIR > 0 ? NPAIR : 1];
  bf16x4 ro[2];
  auto load_res = [&](int i, int slot) {
    const int m = m_base + i * 16 + lr;
    if (m < a.M) {
      const T* row = rg + (int64_t)m * a.Cout;
#pragma unroll
      for (int p = 0; p < NPAIR; ++p) rp[slot][p] = *reinterpret_cast<const bf16x8*>(row + n_pair + 32 * p);
      if constexpr (ODD) ro[slot] = *reinterpret_cast<const bf16x4*>(row + n_odd);
    }
  };
  if (rg) load_res(0, 0);
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    if (rg && i + 1 < FM) load_res(i + 1, (i + 1) & 1);
    const int m = m_base + i * 16 + lr;
    if (m >= a.M) continue;
    T* row = yg + (a.up4 ? conv_out_row(a, m, m / hw, hw) : (int64_t)m) * a.Cout;
#pragma unroll
    for (int p = 0; p < NPAIR; ++p) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[e] = acc[i][2 * p][e] + cb[2 * p][e];
        v[4 + e] = acc[i][2 * p + 1][e] + cb[2 * p + 1][e];
      }
      if (rg) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += (float)rp[i & 1][p][e];
      }
#ifdef GN_ABL   // TIMING-ONLY build (tools/build_variant.sh ... -DGN_ABL): the VALU work of a producer-side GroupNorm + SiLU with made-up
      // scale / shift on the convolutions that carry a time embedding (the out_layers producers); results are wrong on purpose
      if ((a.dbg & 32) && a.emb) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float q = fmaf(v[e], cb[2 * p][e & 3], cb[2 * p + 1][e & 3]);
          v[e] = silu_fast(q);
        }
      }
#endif
      bf16x8 t;
#pragma unroll
      for (int e = 0; e < 8; ++e) t[e] = (bf16_t)(v[e] * a.out_scale);
      *reinterpret_cast<bf16x8*>(row + n_pair + 32 * p) = t;
      if (st) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float q = (float)t[e];
          s1p[p][e] += q;
          s2p[p][e] = fmaf(q, q, s2p[p][e]);
        }
      }
    }
    if constexpr (ODD) {
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = acc[i][FN - 1][e] + cb[FN - 1][e];
      if (rg) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += (float)ro[i & 1][e];
      }
#ifdef GN_ABL
      if ((a.dbg & 32) && a.emb) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float q = fmaf(v[e], cb[FN - 1][e], cb[0][e]);
          v[e] = silu_fast(q);
        }
      }
#endif
      bf16x4 t;
#pragma unroll
      for (int e = 0; e < 4; ++e) t[e] = (bf16_t)(v[e] * a.out_scale);
      *reinterpret_cast<bf16x4*>(row + n_odd) = t;
      if (st) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float q = (float)t[e];
          s1o[e] += q;
          s2o[e] = fmaf(q, q, s2o[e]);
        }
      }
    }
  }
  if (st) {   // lanes lr = 0..15 of a lane group hold the 16 rows of a fragment: butterfly over the group, lane lr = 0 writes
    // DPP row rotations (a row = the 16 lanes of a lane group): VALU-rate, where __shfl_xor's ds_bpermute sends 1 800 requests per
    // tile through the LDS pipe (measured: + 58 us on the 1 660 us dominant launch)
    auto red16 = [](float v) {
      v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xf, 0xf, false));   // row_ror:8
      v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xf, 0xf, false));   // row_ror:4
      v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x122, 0xf, 0xf, false));   // row_ror:2
      v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x121, 0xf, 0xf, false));   // row_ror:1
      return v;
    };
    // block index: row blocks in output order; the parity form of an upsample conv (up4) computes SOURCE rows per parity, and a
    // sample's 4 hw / rows blocks are laid out [sample][parity][source row block] - any order inside a sample serves the sums
    int64_t blk = m_base / (FM * 16);
    if (a.up4) {
      const int bs = m_base / hw, per = hw / (FM * 16);
      blk = ((int64_t)bs * 4 + blockIdx.y) * per + (m_base - bs * hw) / (FM * 16);
    }
    float* d = a.chstat + blk * a.Cout * 2;
#pragma unroll
    for (int p = 0; p < NPAIR; ++p) {
      float o[16];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        o[2 * e] = red16(s1p[p][e]);
        o[2 * e + 1] = red16(s2p[p][e]);
      }
      if (lr == 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *reinterpret_cast<f32x4*>(d + (n_pair + 32 * p) * 2 + 4 * q) = f32x4{o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]};
      }
    }
    if constexpr (ODD) {
      float o[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        o[2 * e] = red16(s1o[e]);
        o[2 * e + 1] = red16(s2o[e]);
      }
      if (lr == 0) {
        *reinterpret_cast<f32x4*>(d + n_odd * 2) = f32x4{o[0], o[1], o[2], o[3]};
        *reinterpret_cast<f32x4*>(d + n_odd * 2 + 4) = f32x4{o[4], o[5], o[6], o[7]};
      }
    }
  }
}

// A 128-row wave tile (FM = 8: the 256-wide configurations 26-28) takes the fast epilogue as two 64-row halves: each half has its own
// embedding row (an 8x8 plane puts two samples into the wave's rows) and its own 64-row block of channel statistics
template <int FM, int FN, bool STATS = true, bool BIAS_LDS = false>
__device__ __forceinline__ void conv_epilogue_fast_rows(const ConvK& a, f32x4 (&acc)[FM][FN], int m_base, int n_base, int lr, int lg, int hw,
                                                        const float* bias_lds = nullptr) {
  if constexpr (FM == 8) {
    conv_epilogue_fast<4, FN, STATS, BIAS_LDS>(a, *reinterpret_cast<f32x4(*)[4][FN]>(&acc[0]), m_base, n_base, lr, lg, hw, bias_lds);
    conv_epilogue_fast<4, FN, STATS, BIAS_LDS>(a, *reinterpret_cast<f32x4(*)[4][FN]>(&acc[4]), m_base + 64, n_base, lr, lg, hw, bias_lds);
  } else {
    conv_epilogue_fast<FM, FN, STATS, BIAS_LDS>(a, acc, m_base, n_base, lr, lg, hw, bias_lds);
  }
}

// =====================================================================================================
// Producer-side GroupNorm (round 5): y = [SiLU](GroupNorm32(acc + bias + emb)) leaves the convolution's epilogue as bf16 - the raw
// tensor is never stored, the gn_apply pass over it (a read + a write of the tensor at the HBM roof: 9.6 % of a UNet step in
// round 4) and its launch are gone.  Serves the ResBlock's `out_layers` norm (openaimodel.py:207-215, 261-270: h = in_layers(x) +
// emb_out; out_layers = GroupNorm32 -> SiLU -> conv): nothing but that norm reads h.
//
// A GroupNorm group = cpg consecutive channels x ALL pixels of a sample.  A 256 x BN tile holds BN / cpg whole groups of 256 pixels;
// a sample is tps = H W / 256 tiles.  Per tile: per-channel (sum, sumsq) of the RAW fp32 accumulators - before bias and embedding are
// added, so that a large per-channel addend never sits inside an fp32 sum of squares (DPP row reduction over the 16 rows of a fragment,
// as the chstat path) -> LDS -> per-group tile partials in fp64 with the addend re-attached exactly, kept as (mean_t, M2_t): the pair
// that combines across tiles without the E[x^2] - E[x]^2 cancellation (round-5 verdict, weak #2a).  tps = 1 (16x16): done, no traffic
// at all.  tps > 1: the partial goes out as ONE 16-byte granule {mean_t, M2_t, tag, tag} per (tile, group) through a write-through
// (`sc1`) store; every tile of the sample polls the sample's tps x groups granules with `sc1` loads until their tag is this launch's
// epoch, and combines them in tile order (fp64; every tile computes the same mean / rstd bit for bit: deterministic).  No counter, no
// atomic, no fence: the tag travels inside the granule's own 16 bytes (MI355X_MICROARCH.md, hand-offs: data-tagged granules, one `sc1`
// store each).  The epoch is a device word that a one-thread kernel bumps before every such launch (hipGraph-replayable; a kernel
// argument would be frozen into a captured graph).
// Progress: the tiles that wait for each other are tps consecutive m-tiles of one n-tile.  Workgroups are dispatched in
// blockIdx order (round-robin over the XCDs, in order inside each), conv_tile_of_block hands an XCD a contiguous range of tiles,
// and tps * tiles_n <= 16 = half the CUs of an XCD: whenever the oldest waiting tile's partners are not all resident, a CU is free
// and the dispatcher brings the next one in - also with a SECOND stream's fused launch interleaved on the same XCD (each launch has at
// most one incomplete cohort, < 16 waiting workgroups each; tests/test_gpu_round6.py runs two handles on two streams).  HIP does not
// promise that order, so the poll is BOUNDED in time: on a timeout the workgroup sets gn_sync[1] (and the host-mapped copy) and
// finishes with what it has; every later fused launch sees the flag and stops waiting at once; the host finds it at the next forward
// or at its next synchronisation (Net::async_status), switches the fusion off FOR THAT NETWORK and reports the run as invalid - the
// samplers (ddmi_amd/ddpm.py) run the chain again on the separate pass.
__device__ __forceinline__ void store_granule_sc1(float* p, f32x4 v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ f32x4 load_granule_sc1(const float* p) {
  f32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}

__device__ __forceinline__ unsigned load_word_sc1(const unsigned* p) {
  unsigned v;
  asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}
// bound of a fused-GroupNorm tile's wait for its partners, in ticks of the constant 100 MHz clock (wall_clock64): 50 ms - a launch of the
// largest shape lasts 2-5 ms and a partner normally arrives within one tile's duration (tens of microseconds)
#define CONV_FGN_TIMEOUT_TICKS 5000000ull

template <int FM, int FN, int BN, int WM>
__device__ __forceinline__ void conv_epilogue_fgn(const ConvK& a, f32x4 (&acc)[FM][FN], unsigned char* lds, int m0, int n0, int wm,
                                                  int wn, int lane, int tid, int hw) {
  using T = bf16_t;
  constexpr int TM = FM * 16, TN = FN * 16, NPAIR = FN / 2;
  constexpr bool ODD = (FN & 1) != 0;
  const int lr = lane & 15, lg = lane >> 4;
  const int m_base = m0 + wm * TM, n_base = n0 + wn * TN;
  // scratch in the halo buffers (nobody reads or fills them after the stage loop's last barrier)
  float* const cs = reinterpret_cast<float*>(lds);     // [WM][BN][2]   per (wave row, channel): sum, sumsq of the RAW accumulators over the wave's TM rows
  float* const gp = cs + WM * BN * 2;                   // [16][32][2]   the sample's tile partials per group: mean, M2
  float* const mr = gp + 16 * 32 * 2;                   // [32][2]       mean, rstd per group of this n-tile
  float* const ss = mr + 64;                            // [BN][2]       scale, shift per channel
  float* const cbs = ss + BN * 2;                       // [BN]          bias + embedding addend per channel (the statistics' shift)
  const int cpg = a.gn_cpg, tps = a.gn_tps, NG = BN / cpg;
  const int b = m0 / hw, t_own = (m0 - b * hw) >> 8;    // sample, tile inside it (BM = 256)
  const int n_pair = wn * TN + 8 * lg;                  // tile-local channel of pair p: + 32 p
  const int n_odd = wn * TN + frag_chan<FN, true>(FN - 1, lg);

  // ---- 1. the per-channel addend cb = bias + emb (fp32); the sums of step 2 are taken BEFORE it is added: a large embedding row or bias
  //         (|mean| >> sigma of the normalised tensor) then never enters an fp32 sum of squares - step 3 re-attaches it exactly in fp64
  f32x4 cb[FN];
  {
#pragma unroll
    for (int j = 0; j < FN; ++j) cb[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (a.bias) {
#pragma unroll
      for (int p = 0; p < NPAIR; ++p) {
        cb[2 * p] = *reinterpret_cast<const f32x4*>(a.bias + n0 + n_pair + 32 * p);
        cb[2 * p + 1] = *reinterpret_cast<const f32x4*>(a.bias + n0 + n_pair + 32 * p + 4);
      }
      if constexpr (ODD) cb[FN - 1] = *reinterpret_cast<const f32x4*>(a.bias + n0 + n_odd);
    }
    if (a.emb) {
      int be = b;
      if (a.emb_mod) {
        be -= be >= a.emb_mod ? a.emb_mod : 0;
        be -= be >= a.emb_mod ? a.emb_mod : 0;
      }
      const float* pe = a.emb + (int64_t)be * a.emb_stride + n0;
#pragma unroll
      for (int p = 0; p < NPAIR; ++p) {
        cb[2 * p] += *reinterpret_cast<const f32x4*>(pe + n_pair + 32 * p);
        cb[2 * p + 1] += *reinterpret_cast<const f32x4*>(pe + n_pair + 32 * p + 4);
      }
      if constexpr (ODD) cb[FN - 1] += *reinterpret_cast<const f32x4*>(pe + n_odd);
    }
  }
  // ---- 2. per-channel sums of the raw accumulators over the wave's TM rows -> LDS; then v = acc + cb in place
  {
    auto red16 = [](float v) {   // sum over the 16 lanes of a DPP row (= the 16 pixel rows of a fragment)
      v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xf, 0xf, false));   // row_ror:8
      v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xf, 0xf, false));   // row_ror:4
      v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x122, 0xf, 0xf, false));   // row_ror:2
      v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x121, 0xf, 0xf, false));   // row_ror:1
      return v;
    };
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      float o[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < FM; ++i) {
          const float q = acc[i][j][e];
          s1 += q;
          s2 = fmaf(q, q, s2);
        }
        o[2 * e] = red16(s1);
        o[2 * e + 1] = red16(s2);
      }
      if (lr == 0) {
        const int ch = wn * TN + frag_chan<FN, true>(j, lg);
        float* d = cs + ((wm * BN) + ch) * 2;
        *reinterpret_cast<f32x4*>(d) = f32x4{o[0], o[1], o[2], o[3]};
        *reinterpret_cast<f32x4*>(d + 4) = f32x4{o[4], o[5], o[6], o[7]};
        if (wm == 0) *reinterpret_cast<f32x4*>(cbs + ch) = cb[j];
      }
    }
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) acc[i][j] += cb[j];
  }
  __syncthreads();
  // (the affine parameters of step 5, fetched here so that their latency lies under the hand-off)
  float gam = 0.f, bet = 0.f;
  if (tid < BN) { gam = a.gn_gamma[n0 + tid]; bet = a.gn_beta[n0 + tid]; }
  // ---- 3. tile partial per group, in fp64 from here on: with k = cb[c] and (a1, a2) the raw sums of channel c over the tile's 256 rows,
  //         sum v = a1 + 256 k and sum v^2 = a2 + 2 k a1 + 256 k^2.  The partial leaves the tile as (mean_t, M2_t = sum (v - mean_t)^2) -
  //         the pair that combines across tiles without cancellation (Chan et al.) - rounded to fp32 ONCE.  tps == 1: complete here.
  const unsigned tag = tps > 1 ? a.gn_sync[0] : 0u;
  const double cnt_t = 256.0 * cpg;
  if (tid < NG) {
    double s1 = 0.0, s2 = 0.0;
    for (int c = tid * cpg; c < (tid + 1) * cpg; ++c) {
      double a1 = 0.0, a2 = 0.0;
      for (int w = 0; w < WM; ++w) {
        a1 += (double)cs[(w * BN + c) * 2];
        a2 += (double)cs[(w * BN + c) * 2 + 1];
      }
      const double k = (double)cbs[c];
      s1 += a1 + 256.0 * k;
      s2 += a2 + 2.0 * k * a1 + 256.0 * k * k;
    }
    const double m = s1 / cnt_t;
    double m2 = s2 - s1 * m;
    if (m2 < 0) m2 = 0;
    if (tps == 1) {
      mr[tid * 2] = (float)m;
      mr[tid * 2 + 1] = (float)(1.0 / sqrt(m2 / cnt_t + (double)a.gn_eps));
    } else {
      const int gg = (n0 / BN) * NG + tid;   // group index in the sample
      store_granule_sc1(a.gn_part + ((int64_t)(b * tps + t_own) * 32 + gg) * 4, f32x4{(float)m, (float)m2, __uint_as_float(tag), __uint_as_float(tag)});
    }
  }
  if (tps > 1) {
    // ---- 4. the sample's tps x NG granules: one lane each, polled until they carry this launch's tag.  BOUNDED in time (the constant
    //         100 MHz clock): a partner that has not arrived after CONV_FGN_TIMEOUT_TICKS - tens of launch durations - is not coming (the
    //         dispatch order is not the assumed one); and once ANY workgroup has flagged that, nobody waits any longer: every later fused
    //         launch of the chain falls through at once instead of sitting out its own timeout (round-5 advisor)
    const int idx = tid, want = tps * NG;
    const int t = idx / NG, g = idx - t * NG;
    const float* src = a.gn_part + ((int64_t)(b * tps + (idx < want ? t : 0)) * 32 + (n0 / BN) * NG + (idx < want ? g : 0)) * 4;
    bool done = idx >= want;
    f32x4 v{0.f, 0.f, 0.f, 0.f};
    const unsigned long long t_start = wall_clock64();
    for (int it = 0;; ++it) {
      if (!done) {
        v = load_granule_sc1(src);
        done = __float_as_uint(v[2]) == tag && __float_as_uint(v[3]) == tag;
      }
      if (__builtin_amdgcn_ballot_w64(!done) == 0) break;
      if ((it & 15) == 15) {   // wave-uniform
        const unsigned flagged = __builtin_amdgcn_readfirstlane(load_word_sc1(a.gn_sync + 1));
        if (flagged != 0u || wall_clock64() - t_start > CONV_FGN_TIMEOUT_TICKS) break;
      }
      __builtin_amdgcn_s_sleep(8);
    }
    if (!done) {   // a partner tile never arrived: flagged (device word for the kernels, host-mapped word for the host), not hung
      atomicOr(a.gn_sync + 1, 1u);
      if (a.gn_flag_host) __hip_atomic_store(a.gn_flag_host, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (idx < want) {
      gp[(t * 32 + g) * 2] = v[0];
      gp[(t * 32 + g) * 2 + 1] = v[1];
    }
  }
  __syncthreads();
  // ---- 5. per-channel scale / shift of this n-tile.  tps > 1: every channel's thread combines its group's tile partials itself, in tile
  //         order (the same bits in every thread and every tile of the sample: deterministic) - one barrier and one LDS round trip less
  //         than reducing on 32 threads first.  Equal counts per tile: mean = average of the tile means, M2 = sum M2_t + n_t sum (mean_t - mean)^2
  if (tid < BN) {
    const int g = tid / cpg;
    float mean, rstd;
    if (tps == 1) {
      mean = mr[g * 2];
      rstd = mr[g * 2 + 1];
    } else {
      double sm = 0.0;
      for (int tt = 0; tt < tps; ++tt) sm += (double)gp[(tt * 32 + g) * 2];
      const double m = sm / tps;
      double m2 = 0.0;
      for (int tt = 0; tt < tps; ++tt) {
        const double d = (double)gp[(tt * 32 + g) * 2] - m;
        m2 += (double)gp[(tt * 32 + g) * 2 + 1] + cnt_t * d * d;
      }
      mean = (float)m;
      rstd = (float)(1.0 / sqrt(m2 / (cnt_t * tps) + (double)a.gn_eps));
    }
    const float sc = rstd * gam;
    ss[tid * 2] = sc;
    ss[tid * 2 + 1] = bet - mean * sc;
  }
  __syncthreads();
  // ---- 6. normalise, SiLU, round once, store
  T* __restrict__ yg = (T*)a.y;
  const bool silu = a.gn_silu != 0;
  auto act = [&](float x, float sc, float sh) {
    float q = fmaf(x, sc, sh);
    if (silu) q = silu_fast(q);
    return q;
  };
#pragma unroll
  for (int p = 0; p < NPAIR; ++p) {
    float sc[8], sh[8];
    {
      const f32x4* q = reinterpret_cast<const f32x4*>(ss + (n_pair + 32 * p) * 2);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const f32x4 w = q[u];
        sc[2 * u] = w[0]; sh[2 * u] = w[1]; sc[2 * u + 1] = w[2]; sh[2 * u + 1] = w[3];
      }
    }
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      const int m = m_base + i * 16 + lr;
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        o[e] = (bf16_t)act(acc[i][2 * p][e], sc[e], sh[e]);
        o[4 + e] = (bf16_t)act(acc[i][2 * p + 1][e], sc[4 + e], sh[4 + e]);
      }
      if (m < a.M) *reinterpret_cast<bf16x8*>(yg + (int64_t)m * a.Cout + n0 + n_pair + 32 * p) = o;
    }
  }
  if constexpr (ODD) {
    float sc[4], sh[4];
    {
      const f32x4* q = reinterpret_cast<const f32x4*>(ss + n_odd * 2);
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const f32x4 w = q[u];
        sc[2 * u] = w[0]; sh[2 * u] = w[1]; sc[2 * u + 1] = w[2]; sh[2 * u + 1] = w[3];
      }
    }
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      const int m = m_base + i * 16 + lr;
      bf16x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (bf16_t)act(acc[i][FN - 1][e], sc[e], sh[e]);
      if (m < a.M) *reinterpret_cast<bf16x4*>(yg + (int64_t)m * a.Cout + n0 + n_odd) = o;
    }
  }
}
// one thread: the launch epoch the granules of the next FGN convolution are tagged with
__global__ void conv_fgn_epoch_kernel(unsigned* sync) { sync[0] += 1u; }

template <typename T, int FM, int FN, bool PAIRED = false>
__device__ __forceinline__ void conv_epilogue(const ConvK& a, f32x4 (&acc)[FM][FN], int m_base, int n_base, int lr,
                                              int lg, int hw) {
  if (a.dbg & 8) return;
  if constexpr (PAIRED && sizeof(T) == 2 && (FM == 2 || FM == 4 || FM == 8)) {
    // wave-uniform test (m_base, n_base and the kernel arguments are): FM * 16 is 32 or 64, a power of two
    constexpr int RG = FM == 8 ? 64 : FM * 16;   // rows that must share a sample (conv_epilogue_fast_rows)
    if (!(a.dbg & 16) && a.ws == nullptr && a.y_nchw == nullptr && a.y32 == nullptr && n_base + FN * 16 <= a.Cout &&
        (a.emb == nullptr || (hw & (RG - 1)) == 0)) {
      conv_epilogue_fast_rows<FM, FN>(a, acc, m_base, n_base, lr, lg, hw);
      return;
    }
  }
  T* __restrict__ yg = a.y ? (T*)a.y + (int64_t)blockIdx.y * a.y_bstride : nullptr;
  const T* __restrict__ rg = (const T*)a.res;
  constexpr int NPAIR = PAIRED ? FN / 2 : 0;
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    int m = m_base + i * 16 + lr;
    if (m >= a.M) continue;
    int b = m / hw;
    int be = b;  // embedding row: sample index modulo emb_mod (planes stacked <= 3x along the batch axis), no division
    if (a.emb_mod) {
      be -= be >= a.emb_mod ? a.emb_mod : 0;
      be -= be >= a.emb_mod ? a.emb_mod : 0;
    }
#pragma unroll
    for (int p = 0; p < NPAIR; ++p) {  // fragments 2p, 2p+1: eight consecutive channels
      int n = n_base + 32 * p + 8 * lg;
      if (n + 8 <= a.Cout) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] = acc[i][2 * p][e]; v[4 + e] = acc[i][2 * p + 1][e]; }
        conv_epilogue_vec<T, 8>(a, v, m, n, b, be, hw, yg, rg);
      } else if (n < a.Cout) {  // Cout is a multiple of 4: the first half exists, the second does not
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[i][2 * p][e];
        conv_epilogue_vec<T, 4>(a, v, m, n, b, be, hw, yg, rg);
      }
    }
#pragma unroll
    for (int j = 2 * NPAIR; j < FN; ++j) {
      int n = n_base + frag_chan<FN, PAIRED>(j, lg);
      if (n >= a.Cout) continue;
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = acc[i][j][e];
      conv_epilogue_vec<T, 4>(a, v, m, n, b, be, hw, yg, rg);
    }
  }
}

// =====================================================================================================
// bf16 production kernel: NST-deep LDS ring filled by LDS-DMA, counted vmcnt, one raw barrier per stage.
//
//   prologue : issue the DMA of stages 0 .. NST-2
//   stage s  : s_waitcnt vmcnt(L * ahead)   - this wave's DMA of stage s has landed (later stages stay in flight)
//              s_barrier                    - everybody's has, and everybody is done reading buffer (s-1) % NST
//              issue the DMA of stage s+NST-1 into buffer (s-1) % NST
//              ds_read fragments of buffer s % NST, MFMA
//   Every wave issues exactly L = KPS * (ceil(BM/16/NW) + ceil(BN/16/NW)) DMA instructions per stage (row groups a
//   wave does not own become out-of-range loads into a scratch KiB), so the vmcnt immediates are wave-independent.
// =====================================================================================================
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// ILV: the refill's DMA instructions are issued BETWEEN the MFMA groups of the stage instead of in one burst behind the barrier.
// A stage of a 1x1 convolution fills 30 KB of LDS for 28 MFMAs per wave: in a burst every wave of the CU queues on the address
// unit at once and no wave feeds the matrix pipe meanwhile (the 448 -> 1344 qkv conv: MFMA alone 286 us, DMA alone 203, together
// 489).  The ring is NST deep with counted vmcnt, so a piece issued anywhere inside stage s is in time for stage s + NST - 1.
template <int BM, int BN, int WM, int WN, int KPS, int NST, bool ILV = false>
__global__ void __launch_bounds__(WM * WN * 64) conv_ring_kernel(ConvK a) {
  using T = bf16_t;
  constexpr int NT = WM * WN * 64, NW = WM * WN;
  constexpr int EPS = 8, SLOTS = 4, ROWB = 64, SZ = 2;
  constexpr int TM = BM / WM, TN = BN / WN, FM = TM / 16, FN = TN / 16;
  static_assert(TM % 16 == 0 && TN % 16 == 0 && BM % 16 == 0 && BN % 16 == 0, "tiles must be multiples of 16");
  constexpr int GA = BM / 16, GB = BN / 16;                  // 16-row DMA groups per chunk
  constexpr int A_PER = (GA + NW - 1) / NW, B_PER = (GB + NW - 1) / NW;
  constexpr int L = KPS * (A_PER + B_PER);                   // DMA instructions per wave per stage
  static_assert(L * (NST - 2) <= 63, "vmcnt immediate overflow");
  constexpr int A_BYTES = KPS * BM * ROWB, B_BYTES = KPS * BN * ROWB;
  constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
  constexpr int SCRATCH = NST * STAGE_BYTES;                 // + wave * 1024: landing zone of dummy DMAs

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  unsigned long long* const ts = a.ts && tid == 0 && blockIdx.y == 0 && blockIdx.z == 0 ? a.ts + 8 * blockIdx.x : nullptr;
  if (ts) {
    ts[0] = wall_clock64();
    ts[5] = ((unsigned long long)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) << 32) |   // XCC_ID
            __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);                               // HW_ID
  }
  const TileMN tmn = conv_tile_of_block(a.tile_map, a.tiles_n);
  const int tile_m = __builtin_amdgcn_readfirstlane(tmn.m), tile_n = __builtin_amdgcn_readfirstlane(tmn.n);
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int Ktot = a.KH * a.KW * a.Cin;
  const int ush = a.ups == 1, usw = a.ups != 0;  // fused nearest upsample: 1 = x2 both axes, 2 = x2 along W only
  const int Hlim = a.Hin << ush, Wlim = a.Win << usw;
  const int hw = a.Hout * a.Wout;

  // descriptors address at most 2 GiB: rebase the activation descriptor at the first sample this tile touches
  const int b_first = m0 / hw;
  const int64_t x_sample = (int64_t)a.Hin * a.Win * a.Cin;
  const T* xg = (const T*)a.x + (int64_t)blockIdx.y * a.x_bstride + (int64_t)b_first * x_sample;
  const uint64_t x_rem = a.x_bytes - (uint64_t)b_first * x_sample * sizeof(T);
  const unsigned x_extent = x_rem > 0x7FFFFFFFull ? 0x7FFFFFFFu : (unsigned)x_rem;
  const T* wg = (const T*)a.w + (int64_t)blockIdx.y * a.w_bstride;
  auto x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)xg, 0, x_extent, 0x00020000);
  auto w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)wg, 0, a.w_bytes, 0x00020000);

  // ---- per-lane gather descriptors (fixed over the K loop); lane owns PHYSICAL slot lane&3 of row lane>>2 ----
  int a_ih0[A_PER], a_iw0[A_PER], a_base[A_PER], a_coff[A_PER], a_pix[A_PER];
  int a_ok[A_PER];
#pragma unroll
  for (int i = 0; i < A_PER; ++i) {
    int row = (i * NW + wave) * 16 + (lane >> 2);
    int slot = swz(row, lane & 3);
    int m = m0 + row;
    int ok = (row < BM) & (m < a.M);
    int b = ok ? m / hw : 0;
    int r = ok ? m - b * hw : 0;
    int oh = r / a.Wout, ow = r - oh * a.Wout;
    a_ok[i] = ok;
    a_ih0[i] = oh * a.stride - conv_pad_h(a);
    a_iw0[i] = ow * a.stride - conv_pad_w(a);
    a_coff[i] = slot * EPS;
    a_pix[i] = (ok ? b - b_first : 0) * a.Hin * a.Win;
    a_base[i] = ((a_pix[i] + a_ih0[i] * a.Win + a_iw0[i]) * a.Cin + a_coff[i]) * SZ;
  }
  int b_base[B_PER], b_coff[B_PER], b_ok[B_PER];
#pragma unroll
  for (int i = 0; i < B_PER; ++i) {
    int row = (i * NW + wave) * 16 + (lane >> 2);
    int slot = swz(row, lane & 3);
    int n = n0 + lds_row_chan<TN>(row);  // paired fragments: LDS row -> weight row (see frag_row)
    b_ok[i] = (row < BN) & (n < a.Cout);
    b_coff[i] = slot * EPS;
    b_base[i] = (n * Ktot + b_coff[i]) * SZ;
  }

  // ---- wave-uniform walk over (tap, channel-chunk) ------------------------------------------------------------
  const int s_begin = blockIdx.z * a.SPS;
  const int s_end = min(a.NS, s_begin + a.SPS);
  int kc_next = s_begin * KPS;
  const int taps = a.KH * a.KW;
  // two walk orders over (tap, channel chunk): channel-major keeps the 9 shifted reads of one channel chunk adjacent
  // in time (L1/L2 reuse of the activation rows); tap-major streams each tap's weights contiguously.
  // mixed-radix counter: channel-major (kw, kh, ci) radices (KW, KH); tap-major (ci, kw, kh) radices (CPT, KW)
  const int r0 = a.chan_major ? a.KW : a.CPT, r1 = a.chan_major ? a.KH : a.KW;
  int d0 = kc_next % r0, d1 = (kc_next / r0) % r1, d2 = kc_next / (r0 * r1);

  // one DMA instruction of chunk c of a stage: piece p < A_PER = activation row group p, else weight row group p - A_PER
  auto load_piece = [&](unsigned char* sbase, int c, int p) {
    const int kvalid = kc_next < a.NK;
    const int q_kw = a.chan_major ? d0 : d1, q_kh = a.chan_major ? d1 : d2, q_ci = a.chan_major ? d2 : d0;
    const int c0 = q_ci * 32;
#pragma unroll
    for (int i = 0; i < A_PER; ++i) {
      if (i != p) continue;
      const int a_delta = ((q_kh * a.Win + q_kw) * a.Cin + c0) * SZ;
      int ih = a_ih0[i] + q_kh, iw = a_iw0[i] + q_kw;
      int ok = a_ok[i] & kvalid & (c0 + a_coff[i] < a.Cin) & ((unsigned)ih < (unsigned)Hlim) &
               ((unsigned)iw < (unsigned)Wlim);
      int off = a_base[i] + a_delta;
      if (a.ups) off = ((a_pix[i] + (ih >> ush) * a.Win + (iw >> usw)) * a.Cin + c0 + a_coff[i]) * SZ;
      unsigned voff = ok ? (unsigned)off : OOB_OFFSET;
      const int g = i * NW + wave;  // wave-uniform
      unsigned char* dst = g < GA ? sbase + c * (BM * ROWB) + g * 1024 : smem + SCRATCH + wave * 1024;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rsrc, (lds_void*)dst, 16, voff, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < B_PER; ++i) {
      if (i + A_PER != p) continue;
      const int b_delta = ((q_kh * a.KW + q_kw) * a.Cin + c0) * SZ;
      int ok = b_ok[i] & kvalid & (c0 + b_coff[i] < a.Cin);
      unsigned voff = ok ? (unsigned)(b_base[i] + b_delta) : OOB_OFFSET;
      const int g = i * NW + wave;
      unsigned char* dst = g < GB ? sbase + A_BYTES + c * (BN * ROWB) + g * 1024 : smem + SCRATCH + wave * 1024;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_void*)dst, 16, voff, 0, 0, 0);
    }
  };
  auto advance_walk = [&]() {
    ++kc_next;
    if (++d0 == r0) {
      d0 = 0;
      if (++d1 == r1) {
        d1 = 0;
        ++d2;
      }
    }
  };
  auto load_stage = [&](int buf) {
    unsigned char* sbase = smem + buf * STAGE_BYTES;
#pragma unroll
    for (int c = 0; c < KPS; ++c) {
      const int kvalid = kc_next < a.NK;
      const int q_kw = a.chan_major ? d0 : d1, q_kh = a.chan_major ? d1 : d2, q_ci = a.chan_major ? d2 : d0;
      const int c0 = q_ci * 32;
      const int a_delta = ((q_kh * a.Win + q_kw) * a.Cin + c0) * SZ;
      const int b_delta = ((q_kh * a.KW + q_kw) * a.Cin + c0) * SZ;
#pragma unroll
      for (int i = 0; i < A_PER; ++i) {
        int ih = a_ih0[i] + q_kh, iw = a_iw0[i] + q_kw;
        int ok = a_ok[i] & kvalid & (c0 + a_coff[i] < a.Cin) & ((unsigned)ih < (unsigned)Hlim) &
                 ((unsigned)iw < (unsigned)Wlim);
        int off = a_base[i] + a_delta;
        if (a.ups) off = ((a_pix[i] + (ih >> ush) * a.Win + (iw >> usw)) * a.Cin + c0 + a_coff[i]) * SZ;
        unsigned voff = ok ? (unsigned)off : OOB_OFFSET;
        const int g = i * NW + wave;  // wave-uniform
        unsigned char* dst = g < GA ? sbase + c * (BM * ROWB) + g * 1024 : smem + SCRATCH + wave * 1024;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rsrc, (lds_void*)dst, 16, voff, 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < B_PER; ++i) {
        int ok = b_ok[i] & kvalid & (c0 + b_coff[i] < a.Cin);
        unsigned voff = ok ? (unsigned)(b_base[i] + b_delta) : OOB_OFFSET;
        const int g = i * NW + wave;
        unsigned char* dst = g < GB ? sbase + A_BYTES + c * (BN * ROWB) + g * 1024 : smem + SCRATCH + wave * 1024;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_void*)dst, 16, voff, 0, 0, 0);
      }
      advance_walk();
    }
  };

  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int lr = lane & 15, lg = lane >> 4;

  if (ts) ts[1] = wall_clock64();
#pragma unroll
  for (int p = 0; p < NST - 1; ++p)
    if (s_begin + p < s_end && !(a.dbg & 1)) load_stage(p);

  int buf = 0;  // buffer of stage s
  for (int s = s_begin; s < s_end; ++s) {
    // stages issued after s and still allowed in flight
    const int ahead = min(NST - 2, s_end - 1 - s);
    if (ahead >= NST - 2) wait_vmcnt<L*(NST - 2)>();
    else if (NST > 3 && ahead == NST - 3) wait_vmcnt<(NST > 3 ? L * (NST - 3) : 0)>();
    else if (NST > 4 && ahead == NST - 4) wait_vmcnt<(NST > 4 ? L * (NST - 4) : 0)>();
    else wait_vmcnt<0>();
    if (!(a.dbg & 4)) __builtin_amdgcn_s_barrier();
    if (ts && s == s_begin) ts[2] = wall_clock64();
    const bool refill = s + NST - 1 < s_end && !(a.dbg & 1);
    unsigned char* const rbase = smem + (buf == 0 ? NST - 1 : buf - 1) * STAGE_BYTES;   // the released buffer
    if (!ILV && refill) load_stage(buf == 0 ? NST - 1 : buf - 1);

    const unsigned char* sa = smem + buf * STAGE_BYTES;
    const unsigned char* sb = sa + A_BYTES;
    if (!(a.dbg & 2))
#pragma unroll
    for (int c = 0; c < KPS; ++c) {
      bf16x8 xf[FM], wf[FN];
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        int row = wm * TM + i * 16 + lr;
        xf[i] = *reinterpret_cast<const bf16x8*>(sa + c * (BM * ROWB) + row * ROWB + swz(row, lg) * 16);
      }
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        int row = wn * TN + j * 16 + lr;
        wf[j] = *reinterpret_cast<const bf16x8*>(sb + c * (BN * ROWB) + row * ROWB + swz(row, lg) * 16);
      }
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        if constexpr (ILV) {   // this fragment row's share of the chunk's DMA pieces, then its MFMAs
          constexpr int NP = A_PER + B_PER;
          if (refill) {
#pragma unroll
            for (int p = (i * NP) / FM; p < ((i + 1) * NP) / FM; ++p) load_piece(rbase, c, p);
          }
          asm volatile("" ::: "memory");
        }
#pragma unroll
        for (int j = 0; j < FN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xf[i], acc[i][j], 0, 0, 0);
        if constexpr (ILV) asm volatile("" ::: "memory");
      }
      if (ILV && refill) advance_walk();
    }
    buf = buf + 1 == NST ? 0 : buf + 1;
  }
  if (ts) ts[3] = wall_clock64();
  conv_epilogue<T, FM, FN, true>(a, acc, m0 + wm * TM, n0 + wn * TN, lr, lg, hw);
  if (ts) {
    ts[4] = wall_clock64();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ts[6] = wall_clock64();   // this wave's stores acknowledged
  }
}

// =====================================================================================================
// bf16 production kernel v2 ("pipe"): 64-channel stages with 128-byte LDS rows, software-pipelined fragments.
//
// What the ablation of the ring kernel showed (224->224 3x3 @64x64, B=32, 256x224 tile): DMA alone 122 us, LDS reads +
// MFMA alone 101 us, both 159 us - (a) every DMA wave-instruction fetched 16 half cache lines (64-byte row pieces),
// (b) all waves read their fragments and then all ran MFMAs, so the LDS read phase was never hidden.  Hence:
//   * a stage = 64 channels of one tap; one DMA instruction moves 8 rows x 128 B = 8 whole cache lines;
//     LDS rows are 128 B with the slot swizzle  pslot = slot ^ ((row >> 1) & 7)  (slot = 16-byte unit, 0..7 = two
//     32-channel chunks x 4 k-groups), conflict-free for every ds_read_b128 lane group; applied on the DMA source side.
//   * fragments of chunk q+1 are read into a second register set while the MFMAs of chunk q run; the single barrier
//     of a stage sits between its two chunks:
//         [A] read frags(2s+1) || MFMA(2s)        wait vmcnt(0); s_barrier   -> stage s+1 landed, buffer s%2 is free
//             issue DMA(stage s+2 -> buffer s%2)
//         [B] read frags(2s+2) || MFMA(2s+1)
//     Two LDS buffers; the DMA of stage s+2 has one full stage of MFMA time to land.
// =====================================================================================================
template <int BM, int BN, int WM, int WN, bool PAIRED>
__global__ void __launch_bounds__(WM * WN * 64) conv_pipe_kernel(ConvK a) {
  using T = bf16_t;
  constexpr int NW = WM * WN;
  constexpr int TM = BM / WM, TN = BN / WN, FM = TM / 16, FN = TN / 16;
  static_assert(TM % 16 == 0 && TN % 16 == 0 && BM % 8 == 0 && BN % 8 == 0, "bad tile");
  constexpr int GA = BM / 8, GB = BN / 8;  // 8-row DMA groups per stage
  constexpr int A_PER = (GA + NW - 1) / NW, B_PER = (GB + NW - 1) / NW;
  constexpr int ROWB = 128, SZ = 2;
  constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB, STAGE_BYTES = A_BYTES + B_BYTES;
  constexpr int SCRATCH = 2 * STAGE_BYTES;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const TileMN tmn = conv_tile_of_block(a.tile_map, a.tiles_n);
  const int tile_m = __builtin_amdgcn_readfirstlane(tmn.m), tile_n = __builtin_amdgcn_readfirstlane(tmn.n);
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int Ktot = a.KH * a.KW * a.Cin;
  const int ush = a.ups == 1, usw = a.ups != 0;  // fused nearest upsample: 1 = x2 both axes, 2 = x2 along W only
  const int Hlim = a.Hin << ush, Wlim = a.Win << usw;
  const int hw = a.Hout * a.Wout;
  const int taps = a.KH * a.KW;
  const int NCB = (a.Cin + 63) >> 6;  // 64-channel blocks per tap

  // descriptors address at most 2 GiB: rebase the activation descriptor at the first sample this tile touches
  const int b_first = m0 / hw;
  const int64_t x_sample = (int64_t)a.Hin * a.Win * a.Cin;
  const T* xg = (const T*)a.x + (int64_t)blockIdx.y * a.x_bstride + (int64_t)b_first * x_sample;
  const uint64_t x_rem = a.x_bytes - (uint64_t)b_first * x_sample * sizeof(T);
  const unsigned x_extent = x_rem > 0x7FFFFFFFull ? 0x7FFFFFFFu : (unsigned)x_rem;
  const T* wg = (const T*)a.w + (int64_t)blockIdx.y * a.w_bstride;
  auto x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)xg, 0, x_extent, 0x00020000);
  auto w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)wg, 0, a.w_bytes, 0x00020000);

  // ---- per-lane gather descriptors: lane owns physical slot lane&7 of row (lane>>3) of its 8-row group --------
  // (row >> 1) & 7 == (4*(group & 1) + (lane >> 4)) & 7 and group parity == wave parity (NW is even), so the logical
  // slot - hence the channel offset - is ONE per-lane constant for every load of both operands.
  static_assert(NW % 2 == 0, "slot constant needs an even wave count");
  const int coff = (((lane & 7) ^ ((4 * (wave & 1) + (lane >> 4)) & 7))) * 8;
  // one register per A row group (the kernel sits at the 256-VGPR limit): sample index relative to the tile's first
  // sample << 24 | (ih0 + 0x400) << 12 | (iw0 + 0x400); invalid rows get an ih0 field of 0 = -0x400 -> bounds test fails
  int a_row[A_PER];
#pragma unroll
  for (int i = 0; i < A_PER; ++i) {
    int row = (i * NW + wave) * 8 + (lane >> 3);
    int m = m0 + row;
    bool ok = (row < BM) && (m < a.M);
    int b = ok ? m / hw : 0;
    int r = ok ? m - b * hw : 0;
    int oh = r / a.Wout, ow = r - oh * a.Wout;
    int ih0 = ok ? oh * a.stride - conv_pad_h(a) + 0x400 : 0;
    a_row[i] = ((ok ? b - b_first : 0) << 24) | (ih0 << 12) | (ow * a.stride - conv_pad_w(a) + 0x400);
  }
  const int sample_bytes = a.Hin * a.Win * a.Cin * SZ, coff_b = coff * SZ;
  // weight rows: group i of a wave sits 8*NW rows after group 0 - a scalar step, so one base register serves all
  const int b_row0 = wave * 8 + (lane >> 3);
  const int b_base0 = (n0 * Ktot + coff) * SZ;   // + weight row * Ktot * SZ per load
  // paired fragments: LDS row R holds weight row lds_row_chan(R); the B_PER row shifts (|d| <= 12) packed 8 bits each
  int b_perm[(B_PER + 3) / 4];
#pragma unroll
  for (int i = 0; i < (B_PER + 3) / 4; ++i) b_perm[i] = 0;
#pragma unroll
  for (int i = 0; i < B_PER; ++i) {
    const int R = b_row0 + i * 8 * NW;
    b_perm[i >> 2] |= (((PAIRED ? lds_row_chan<TN>(R) : R) - R + 16) & 0xFF) << (8 * (i & 3));
  }

  // ---- wave-uniform walk over stages: (tap, 64-channel block) in tap-major or channel-major order ----------------
  const int s_begin = blockIdx.z * a.SPS;
  const int s_end = min(a.NS, s_begin + a.SPS);
  int st_next = s_begin;
  // mixed-radix counter d0 (fastest) / d1 / d2 with radices r0, r1:
  //   channel-major: (kw, kh, cb) radices (KW, KH);   tap-major: (cb, kw, kh) radices (NCB, KW)
  const int r0 = a.chan_major ? a.KW : NCB, r1 = a.chan_major ? a.KH : a.KW;
  int d0 = st_next % r0, d1 = (st_next / r0) % r1, d2 = st_next / (r0 * r1);

  auto load_stage = [&](int buf) {
    unsigned char* sbase = smem + buf * STAGE_BYTES;
    const int kvalid = st_next < a.NS;
    const int q_kw = a.chan_major ? d0 : d1, q_kh = a.chan_major ? d1 : d2, q_cb = a.chan_major ? d2 : d0;
    const int c0 = q_cb * 64;
    const int cvalid = kvalid & (c0 + coff < a.Cin);
    const int b_delta = ((q_kh * a.KW + q_kw) * a.Cin + c0) * SZ;
    const int rowb = a.Cin * SZ, c0b = c0 * SZ;
#pragma unroll
    for (int i = 0; i < A_PER; ++i) {
      int ih = ((a_row[i] >> 12) & 0xFFF) + (q_kh - 0x400), iw = (a_row[i] & 0xFFF) + (q_kw - 0x400);
      int ok = cvalid & ((unsigned)ih < (unsigned)Hlim) & ((unsigned)iw < (unsigned)Wlim);
      ih >>= ush; iw >>= usw;
      int off = (int)((unsigned)a_row[i] >> 24) * sample_bytes + coff_b + (ih * a.Win + iw) * rowb + c0b;
      unsigned voff = ok ? (unsigned)off : OOB_OFFSET;
      const int g = i * NW + wave;
      unsigned char* dst = g < GA ? sbase + g * 1024 : smem + SCRATCH + wave * 1024;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rsrc, (lds_void*)dst, 16, voff, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < B_PER; ++i) {
      int brow = b_row0 + i * 8 * NW;
      if constexpr (PAIRED) {  // LDS row -> weight row of the paired layout
        int pk = b_perm[i >> 2];
        asm volatile("" : "+v"(pk));  // keep the unpack (2 VALU) in the loop: hoisted it would cost B_PER registers
        brow += (int)((pk >> (8 * (i & 3))) & 0xFF) - 16;
      }
      int ok = cvalid & (b_row0 + i * 8 * NW < BN) & (n0 + brow < a.Cout);
      unsigned voff = ok ? (unsigned)(b_base0 + brow * Ktot * SZ + b_delta) : OOB_OFFSET;
      const int g = i * NW + wave;
      unsigned char* dst = g < GB ? sbase + A_BYTES + g * 1024 : smem + SCRATCH + wave * 1024;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_void*)dst, 16, voff, 0, 0, 0);
    }
    ++st_next;
    if (++d0 == r0) {
      d0 = 0;
      if (++d1 == r1) {
        d1 = 0;
        ++d2;
      }
    }
  };

  const int lr = lane & 15, lg = lane >> 4;
  // LDS byte offset of this lane's first fragment row per operand; further fragments are +16 rows = +2048 B (the
  // swizzle term ((row >> 1) & 7) is the same for rows 16 apart, and TM/2, TN/2 are multiples of 8); the second
  // 32-channel chunk is the same address with bit 6 flipped (logical slot + 4).
  static_assert((TM / 2) % 8 == 0 && (TN / 2) % 8 == 0, "fragment offsets must be swizzle-periodic");
  const int xoff0 = (wm * TM + lr) * ROWB + ((lg ^ ((lr >> 1) & 7)) << 4);
  const int woff0 = A_BYTES + (wn * TN + lr) * ROWB + ((lg ^ ((lr >> 1) & 7)) << 4);
  auto read_frags = [&](int buf, int c, bf16x8(&xf)[FM], bf16x8(&wf)[FN]) {
    const unsigned char* px = smem + buf * STAGE_BYTES + (xoff0 ^ (c << 6));
    const unsigned char* pw = smem + buf * STAGE_BYTES + (woff0 ^ (c << 6));
#pragma unroll
    for (int i = 0; i < FM; ++i) xf[i] = *reinterpret_cast<const bf16x8*>(px + i * 2048);
#pragma unroll
    for (int j = 0; j < FN; ++j) wf[j] = *reinterpret_cast<const bf16x8*>(pw + j * 2048);
  };

  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto mma = [&](bf16x8(&xf)[FM], bf16x8(&wf)[FN]) {
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xf[i], acc[i][j], 0, 0, 0);
  };

  auto mma_rows_impl = [&](int i0, int i1, bf16x8(&xf)[FM], bf16x8(&wf)[FN]) {
#pragma unroll
    for (int i = 0; i < FM; ++i)
      if (i >= i0 && i < i1) {
#pragma unroll
        for (int j = 0; j < FN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xf[i], acc[i][j], 0, 0, 0);
      }
  };
#define mma_rows_helper(I0, I1, X, W) mma_rows_impl((I0), (I1), (X), (W))
  bf16x8 xa[FM], wa[FN], xb[FM], wb[FN];  // even / odd chunk register sets
  constexpr int L = A_PER + B_PER;

  if (s_begin < s_end) {
    load_stage(0);
    if (s_begin + 1 < s_end) {
      load_stage(1);
      wait_vmcnt<L>();
    } else {
      wait_vmcnt<0>();
    }
    __builtin_amdgcn_s_barrier();
    read_frags(0, 0, xa, wa);
  }
  int buf = 0;
  for (int s = s_begin; s < s_end; ++s) {
    // is the second 32-channel chunk of this stage inside Cin?  (Cin % 64 == 32 leaves a half stage per tap)
    const int cb = a.chan_major ? s / taps : s % NCB;
    const bool c1 = cb * 64 + 32 < a.Cin;
    // [A] chunk 2s on set a, prefetch chunk 2s+1 into set b
    read_frags(buf, 1, xb, wb);
    mma(xa, wa);
    // lgkmcnt(0): this wave's ds_reads of `buf` are retired before it arrives at the barrier, so the DMA that any wave
    // issues into `buf` afterwards cannot overtake a read (WAR); vmcnt(0): this wave's share of stage s+1 has landed.
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (s + 2 < s_end) load_stage(buf);
    // [B] chunk 2s+1 on set b, prefetch chunk 0 of stage s+1 into set a
    if (s + 1 < s_end) read_frags(buf ^ 1, 0, xa, wa);
    if (c1) mma(xb, wb);
    buf ^= 1;
  }
  conv_epilogue<T, FM, FN, PAIRED>(a, acc, m0 + wm * TM, n0 + wn * TN, lr, lg, hw);
}

// =====================================================================================================
// bf16 production kernel v4 ("gemm"): PERSISTENT 1x1 convolution (qkv / proj_out / skip_connection: 19 % of a UNet step at
// 0.6-0.9 PFLOP/s on the kernels above).  A 256 x 224 tile of a K = 448 GEMM lives 21 us, of which only ~15 are its K loop:
// index setup 0.7 + first stages landing 3.2 + epilogue 2.8 + store acknowledgement 0.5 + workgroup turnover 0.55 (DDMI_CONV_TS,
// round 1) add, because a tile's prologue cannot start before the previous workgroup has left the CU.  The two persistent
// attempts of round 1 (on the general ring / pipe kernels) lost to the registers of the tile transition; a 1x1-ONLY kernel has
// almost none: no taps, no padding, no upsampling, one per-lane offset per operand (row * K + slot) and two scalars per tile.
//   * one workgroup per CU walks a contiguous range of tiles (n fastest: the six N tiles of a qkv M tile re-read its A rows
//     from L2 back to back);
//   * ONE flat stage loop over (tile, 64-channel block) with the pipe kernel's structure (two LDS buffers, software-pipelined
//     fragments, one barrier per stage): the DMA of stage g + 2 - which may belong to the NEXT tile - is issued behind the
//     barrier of stage g, so a tile's first stages land under the previous tile's last MFMAs and its epilogue;
//   * the epilogue (conv_epilogue_fast: bias / embedding / residual / GroupNorm statistics as before) runs on the accumulators
//     right after the tile's last stage; its stores drain under the next tile's first MFMA chunk.
// Requirements (else the launch falls back to the pipe kernel): bf16, 1x1, stride 1, no split-K, Cout % 224 == 0, NHWC output.
// =====================================================================================================
// NA = activation slots in LDS.  NA = 2: A and B of stage g + 2 are fetched together behind the barrier of stage g.  NA = 3 (when
// the bias vector leaves room): the ACTIVATION tile - the operand that comes from HBM - is fetched one stage further ahead than the
// weights (L2): behind barrier g go B(g + 2) and then A(g + 3), and the wait before barrier g + 1 is `vmcnt(A_PER)`: everything but
// the four youngest instructions, i.e. A(g + 3) stays in flight across the barrier (every wave issues exactly A_PER activation
// pieces per stage, so the immediate is wave-independent whatever its share of the weight rows).
// CAT instantiation (round 5): the A operand is the channel concatenation [x | xcat] of two NHWC tensors read in place - the ResBlock's
// skip_connection on th.cat([h, hs.pop()], 1) (openaimodel.py:787, 272-275) in ONE launch.  It ran as two accumulating 1x1 launches
// (conv(x1, W[:, :C1]) then conv(x2, W[:, C1:]) + the first result as residual): the intermediate was written and read back - a third
// of the traffic of a pair that is HBM-bound at 64x64.  Here the 64-channel K blocks of x are followed by those of xcat (a ragged last
// block of x is zero-filled like any channel tail; its weight columns stop at C1), the per-block source is wave-uniform scalar state.
template <int BN, int WM, int WN, bool STATS, int NA, bool CAT = false>
__global__ void __launch_bounds__(WM * WN * 64) conv_gemm1x1_kernel(ConvK a) {
  static_assert(!(CAT && STATS), "the dual-source 1x1 convolution delivers no channel statistics");
  using T = bf16_t;
  constexpr int BM = 256, NW = WM * WN;
  constexpr int TM = BM / WM, TN = BN / WN, FM = TM / 16, FN = TN / 16;
  constexpr int GA = BM / 8, GB = BN / 8;
  constexpr int A_PER = GA / NW, B_PER = (GB + NW - 1) / NW;
  static_assert(GA % NW == 0 && NW % 2 == 0 && (TM / 2) % 8 == 0 && (TN / 2) % 8 == 0 && (NA == 2 || NA == 3), "bad tile");
  constexpr int ROWB = 128, SZ = 2;
  constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB, B_BASE = NA * A_BYTES;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* const bias_lds = reinterpret_cast<float*>(smem + B_BASE + 2 * B_BYTES);   // [Cout] (zeros when the convolution has no bias)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int K = a.Cin;
  const int NS1 = CAT ? (a.C1 + 63) >> 6 : 0;                          // 64-channel blocks of the first source
  const int NS = CAT ? NS1 + ((K - a.C1 + 63) >> 6) : (K + 63) >> 6;
  for (int i = tid; i < a.Cout; i += NW * 64) bias_lds[i] = a.bias ? a.bias[i] : 0.f;   // visible after the first barrier below
  const int tiles_n = a.tiles_n, tiles_m = (a.M + BM - 1) / BM, T_all = tiles_m * tiles_n;
  // Tile order.  Workgroups are dealt to the 8 XCDs round-robin (w % 8); XCD x owns the contiguous eighth [r0, r1) of the tile list
  // and its workgroups take tiles r0 + q, r0 + q + Sx, ... (q = w / 8, Sx = workgroups on this XCD): at any time the XCD's CUs sit on
  // ~32 CONSECUTIVE tiles - the n tiles of a few m tiles (m-major list) or the m tiles of one n tile (n-major, tile_map 2: weights
  // larger than activations) - so the shared operand is fetched into that L2 once, as under the hardware's own dispatch order.
  // (A first version gave every workgroup a contiguous range, n fastest: each CU then re-read its own A tile 12 times with 31 other
  // CUs' tiles between the reads - 14.7 MB per XCD against a 4 MB L2 - and the 896 -> 2688 qkv ran 13 % SLOWER than the pipe kernel.)
  const int NX = gridDim.x >= 8 ? 8 : (int)gridDim.x;   // (a grid of fewer than 8 workgroups: as many groups as workgroups)
  const int xcd = blockIdx.x % NX, q = blockIdx.x / NX, Sx = ((int)gridDim.x - xcd + NX - 1) / NX;
  const int r0 = (int)((int64_t)xcd * T_all / NX), r1 = (int)((int64_t)(xcd + 1) * T_all / NX);
  const int t0 = r0 + q;
  const int ntile = t0 < r1 ? (r1 - t0 + Sx - 1) / Sx : 0;
  const int G = ntile * NS;   // stages of this workgroup
  const int hw = a.Hout * a.Wout;
  const bool n_major = a.tile_map == 2;
  auto tile_mn = [&](int t, int* tm, int* tn) {
    if (n_major) { *tn = t / tiles_m; *tm = t - *tn * tiles_m; }
    else { *tm = t / tiles_n; *tn = t - *tm * tiles_n; }
  };

  // per-lane constants (the pipe kernel's): physical slot lane&7 of row lane>>3 of an 8-row group; logical slot -> channel offset
  const int coff = (((lane & 7) ^ ((4 * (wave & 1) + (lane >> 4)) & 7))) * 8;
  const int row0 = wave * 8 + (lane >> 3);            // group i adds 8 * NW rows (a scalar step)
  const int a_lane = (row0 * K + coff) * SZ;          // + (i * 8 * NW) * K * SZ   (+ 128 per 64-channel block), tile-relative
  int b_lane[B_PER];                                  // weight row of the paired layout for this lane's row of group i
#pragma unroll
  for (int i = 0; i < B_PER; ++i) b_lane[i] = (lds_row_chan<TN>(row0 + i * 8 * NW) * K + coff) * SZ;
  auto w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, a.w_bytes, 0x00020000);

  // ---- load cursors: (tile, 64-channel block) of the next activation / weight stage to fetch -----------------------------
  int lta = t0, lka = 0, ltb = t0, lkb = 0;
  auto load_a = [&](int slot) {
    int tm, tn;
    tile_mn(lta, &tm, &tn);
    const int m0 = tm * BM;
    // the activation descriptor is rebased at the tile's first row (tensors may exceed 2 GiB); rows past M fall outside its
    // extent and read as zeros
    if constexpr (CAT) {   // block lka of [x | xcat]: source, its row pitch and the block inside it are scalars
      const bool second = lka >= NS1;
      const int Cs = second ? K - a.C1 : a.C1, kb = second ? lka - NS1 : lka;
      const T* src = (const T*)(second ? a.xcat : a.x) + (int64_t)m0 * Cs;
      const uint64_t rem = (uint64_t)(a.M - m0) * Cs * SZ;
      auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, rem > 0x7FFFFFFFull ? 0x7FFFFFFFu : (unsigned)rem, 0x00020000);
      const int cok = kb * 64 + coff < Cs;
      const int lane_off = (row0 * Cs + coff) * SZ + kb * 128;
#pragma unroll
      for (int i = 0; i < A_PER; ++i) {
        const unsigned voff = cok ? (unsigned)(lane_off + i * (8 * NW) * Cs * SZ) : OOB_OFFSET;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(smem + slot * A_BYTES + (i * NW + wave) * 1024), 16, voff, 0, 0, 0);
      }
      if (++lka == NS) { lka = 0; lta += Sx; }
      return;
    }
    const uint64_t rem = (uint64_t)(a.M - m0) * K * SZ;
    auto x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)((const T*)a.x + (int64_t)m0 * K), 0, rem > 0x7FFFFFFFull ? 0x7FFFFFFFu : (unsigned)rem,
                                                    0x00020000);
    const int cok = lka * 64 + coff < K;
    const int kbyte = lka * 128;
#pragma unroll
    for (int i = 0; i < A_PER; ++i) {
      const unsigned voff = cok ? (unsigned)(a_lane + i * (8 * NW) * K * SZ + kbyte) : OOB_OFFSET;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rsrc, (lds_void*)(smem + slot * A_BYTES + (i * NW + wave) * 1024), 16, voff, 0, 0, 0);
    }
    if (++lka == NS) { lka = 0; lta += Sx; }
  };
  auto load_b = [&](int slot) {
    int tm, tn;
    tile_mn(ltb, &tm, &tn);
    const int n0 = tn * BN;
    int cok = lkb * 64 + coff < K;
    int kbyte = lkb * 128;
    if constexpr (CAT) {   // weight columns of the block: [0, C1) for x, [C1, K) for xcat; a ragged last block of x stops at C1
      const bool second = lkb >= NS1;
      const int kb = second ? lkb - NS1 : lkb;
      cok = kb * 64 + coff < (second ? K - a.C1 : a.C1);
      kbyte = ((second ? a.C1 : 0) + kb * 64) * SZ;
    }
#pragma unroll
    for (int i = 0; i < B_PER; ++i) {
      const int g = i * NW + wave;
      if (g < GB) {
        int bl = b_lane[i];
        const unsigned voff = cok ? (unsigned)(bl + n0 * K * SZ + kbyte) : OOB_OFFSET;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_void*)(smem + B_BASE + slot * B_BYTES + g * 1024), 16, voff, 0, 0, 0);
      }
    }
    if (++lkb == NS) { lkb = 0; ltb += Sx; }
  };

  const int lr = lane & 15, lg = lane >> 4;
  const int xoff0 = (wm * TM + lr) * ROWB + ((lg ^ ((lr >> 1) & 7)) << 4);
  const int woff0 = B_BASE + (wn * TN + lr) * ROWB + ((lg ^ ((lr >> 1) & 7)) << 4);
  auto read_frags = [&](int sa, int sb, int c, bf16x8(&xf)[FM], bf16x8(&wf)[FN]) {
    const unsigned char* px = smem + sa * A_BYTES + (xoff0 ^ (c << 6));
    const unsigned char* pw = smem + sb * B_BYTES + (woff0 ^ (c << 6));
#pragma unroll
    for (int i = 0; i < FM; ++i) xf[i] = *reinterpret_cast<const bf16x8*>(px + i * 2048);
#pragma unroll
    for (int j = 0; j < FN; ++j) wf[j] = *reinterpret_cast<const bf16x8*>(pw + j * 2048);
  };
  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto mma = [&](bf16x8(&xf)[FM], bf16x8(&wf)[FN]) {
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xf[i], acc[i][j], 0, 0, 0);
  };
  bf16x8 xa[FM], wa[FN], xb[FM], wb[FN];

  if (G > 0) {
    load_b(0);
    if (G > 1) load_b(1);
#pragma unroll
    for (int p = 0; p < NA; ++p)
      if (p < G) load_a(p);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    read_frags(0, 0, 0, xa, wa);
  }
  int ct = t0, ckb = 0, sa = 0, sb = 0;   // compute cursor; LDS slots of the current stage
#pragma unroll 1
  for (int g = 0; g < G; ++g) {
    const bool c1 = CAT ? (ckb >= NS1 ? (ckb - NS1) * 64 + 32 < K - a.C1 : ckb * 64 + 32 < a.C1) : ckb * 64 + 32 < K;
    read_frags(sa, sb, 1, xb, wb);
    mma(xa, wa);
    // A(g + 2) - issued behind the previous barrier as the youngest A_PER instructions - may stay in flight (NA = 3)
    if (NA == 3 && g >= 1 && g + 2 < G) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(A_PER) : "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (g + 2 < G) load_b(sb);                 // the weight slot just consumed
    if (g + NA < G) load_a(sa);                // the activation slot just consumed
    const int sa1 = sa + 1 == NA ? 0 : sa + 1, sb1 = sb ^ 1;
    const bool last = ckb + 1 == NS;
    if (!last && g + 1 < G) read_frags(sa1, sb1, 0, xa, wa);
    if (c1) mma(xb, wb);
    if (last) {   // the tile is complete: epilogue on the accumulators, then start the next tile's first chunk
      int tm, tn;
      tile_mn(ct, &tm, &tn);
      conv_epilogue_fast_rows<FM, FN, STATS, true>(a, acc, tm * BM + wm * TM, tn * BN + wn * TN, lr, lg, hw, bias_lds);
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (g + 1 < G) read_frags(sa1, sb1, 0, xa, wa);
      ckb = 0; ct += Sx;
    } else {
      ++ckb;
    }
    sa = sa1; sb = sb1;
  }
}

// =====================================================================================================
// bf16 production kernel v3 ("halo") for 3x3 / stride 1 / pad 1 convolutions on W in {8, 16, 32, 64}.
//
// PMC counters of the pipe kernel: 456 MB fetched per launch against 70 MB of algorithmic reads - the nine taps re-read
// every activation row and L2 does not absorb it.  Here a 256-pixel tile = TR whole rows of one sample, and for each
// 64-channel block its (TR+2) x (W+2) HALO is staged in LDS once (zero-filled outside the image by out-of-range DMA
// offsets); the nine taps of that block then read their A fragments from the halo at shifted rows, so only the
// weight stage streams per tap.  Activation traffic per 64-channel block: 396 rows instead of 9 x 256.
//   LDS: 2 halo buffers (400 x 128 B) + 2 weight stages (BN x 128 B) = 156 KiB at BN = 224.
//   The halo of block cb+1 is fetched in eight pieces behind the barriers of taps 0..7 of block cb.
// Loop structure, fragment double-buffering and the swizzle are those of the pipe kernel.
// =====================================================================================================
// SKIP instantiation: the ResBlock's 1x1 skip_connection rides on the same accumulators.  The 64-channel blocks of the skip
// sources x2 (| x3) are appended to the block list as ONE-stage blocks: their A tile is the tile's 256 centre pixels as plain rows
// in a halo buffer (32 DMA groups instead of the halo's 50), the single stage takes its weight stage from w2 [Cout][C2 + C3].
// Same loop, same pipeline; skip tiles are fetched whole and two blocks ahead (see the refill).
// What it replaces: a separate 1x1 launch at 0.4-0.6 PFLOP/s that wrote the skip tensor (HBM-bound at 64x64) plus this kernel's
// epilogue reading it back as the residual.  Split z = 0 carries the skip blocks.
#define HALO_MAX_ROWS 400  // 6 x 66 = 396 rows at W = 64, rounded up to whole 8-row DMA groups (the tail lanes write zeros)
// KT = 2 instantiation: the parity form of upsample + 3x3 (four 2x2 convolutions on the source grid, blockIdx.y = parity):
// a (TR+1) x (W+1) halo whose border side depends on the parity, four taps per channel block, the up4 scatter in the epilogue.
// HALF instantiation (round 5): Cin = 64 k + 32 (the AFHQ UNet's 224 and 672) left the last 64-channel block half empty - nine
// stages that paid their barrier, their weight-stage DMA issue and their fragment reads for half the MFMAs (12.5 % of the stages of a
// 224-channel convolution).  Here the ragged block runs TWO TAPS per stage: chunk 0 = its 32 channels at tap 2j, chunk 1 = the same
// 32 channels at tap 2j + 1 - the weight stage holds [tap 2j | tap 2j + 1] halves (a per-lane column offset on the DMA source), the A
// fragments of chunk 1 come from the SAME halo chunk at the next tap's shift.  Five stages instead of nine, every MFMA useful but
// the last stage's second half; the products and their accumulation order are unchanged, so results are bit-identical.  A separate
// instantiation: the plain one keeps its registers and schedule.
template <int BN, int WM, int WN, bool SKIP, int KT = 3, bool HALF = false, bool FGN = false>
__global__ void __launch_bounds__(WM * WN * 64) conv_halo_kernel(ConvK a) {
  static_assert(!FGN || (!SKIP && KT == 3), "halo kernel: the GroupNorm epilogue exists for plain 3x3 convolutions");
  using T = bf16_t;
  constexpr int BM = 256, NW = WM * WN, TAPS = KT * KT;
  static_assert(KT == 3 || (KT == 2 && !SKIP), "halo kernel: 3x3, or the 2x2 parity convolutions of an upsample");
  static_assert(!HALF || (!SKIP && KT == 3), "halo kernel: the two-taps-per-stage tail exists for plain 3x3 convolutions");
  constexpr int TM = BM / WM, TN = BN / WN, FM = TM / 16, FN = TN / 16;
  static_assert(TM % 16 == 0 && TN % 16 == 0 && (TN / 2) % 8 == 0 && NW % 2 == 0, "bad tile");
  constexpr int GB = BN / 8, B_PER = (GB + NW - 1) / NW;
  constexpr int ROWB = 128, SZ = 2;
  // halo buffer rows: 400 serve every W (6 x 66 at W = 64, 4 x 100 at 8 x 8); the 256-wide tile's two 32 KB weight stages leave room for
  // 344 (10 x 34 at W = 32, 18 x 18 at W = 16: the launcher keeps it off the other shapes)
  constexpr int HROWS = BN > 224 ? 344 : HALO_MAX_ROWS;
  constexpr int H_BYTES = HROWS * ROWB, B_BYTES = BN * ROWB;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const HB = smem;                 // [2][H_BYTES]
  unsigned char* const BB = smem + 2 * H_BYTES;   // [2][B_BYTES]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  unsigned long long* const ts = !SKIP && a.ts && tid == 0 && blockIdx.z == 0 ? a.ts + 8 * blockIdx.x : nullptr;
  if (ts) {
    ts[0] = wall_clock64();
    ts[5] = ((unsigned long long)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) << 32) |   // XCC_ID
            __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);                               // HW_ID
  }
  const TileMN tmn = conv_tile_of_block(a.tile_map, a.tiles_n);
  const int tile_m = __builtin_amdgcn_readfirstlane(tmn.m), tile_n = __builtin_amdgcn_readfirstlane(tmn.n);
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int W = a.Win, H = a.Hin, hw = H * W;
  const int Ktot = TAPS * a.Cin;
  const int NCB = (a.Cin + 63) >> 6;
  // a tile is TR whole rows of ONE sample (hw >= 256), or SPT whole samples of RS = H rows each (hw = 64 / 128: the
  // 8x8 level); every sample block carries its own one-pixel border, so the halo has SPT * (RS + 2) rows of W + 2
  const int HWd = W + KT - 1;
  const int SPT = hw >= BM ? 1 : BM / hw, RS = hw >= BM ? BM / W : H;
  const int HRS = (RS + KT - 1) * HWd, NH = SPT * HRS, GH = (NH + 7) >> 3;
  const int b_first = m0 / hw;
  const int ty0 = hw >= BM ? (m0 - b_first * hw) / W : 0;

  const T* xg = (const T*)a.x + (int64_t)b_first * hw * a.Cin;
  const T* wg = (const T*)a.w + (KT == 2 ? (int64_t)blockIdx.y * a.w_bstride : 0);
  const uint64_t x_rem = a.x_bytes - (uint64_t)b_first * hw * a.Cin * sizeof(T);
  const unsigned x_extent = x_rem > 0x7FFFFFFFull ? 0x7FFFFFFFu : (unsigned)x_rem;
  auto x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)xg, 0, x_extent, 0x00020000);
  auto w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)wg, 0, a.w_bytes, 0x00020000);

  // ---- the block list of this workgroup: its split's 64-channel blocks of x (nine stages each), then - SKIP, split 0 - the blocks
  // of the skip sources (one stage each).  split-K: whole blocks per split (SPS is a multiple of 9)
  const int cb_begin = (blockIdx.z * a.SPS) / TAPS;
  const int cb_end = min(NCB, cb_begin + a.SPS / TAPS);
  const int nmain = max(cb_end - cb_begin, 0);
  // The skip arguments are read from the kernarg segment WHERE THEY ARE USED (laundered pointer, scalar loads that hit the constant
  // cache): as ordinary kernel arguments they would occupy a dozen SGPRs for the whole 3x3 loop, which has none to spare.
  auto skip_arg = [&](auto proto, size_t off) {
    const char* kp = (const char*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kp));
    return *reinterpret_cast<const decltype(proto)*>(kp + off);
  };
  int NCB2 = 0, nsk = 0;
  if constexpr (SKIP) {
    if (blockIdx.z == 0) {
      NCB2 = (skip_arg(int(0), offsetof(ConvK, C2)) + 63) >> 6;
      nsk = NCB2 + ((skip_arg(int(0), offsetof(ConvK, C3)) + 63) >> 6);
    }
  }
  // HALF (launched without split-K only): the ragged block is the LAST block and runs in a loop of its own behind the stage loop,
  // HT = (TAPS + 1) / 2 two-tap stages (the first version decided "half or not" inside the ONE stage loop: 5-9 % slower on every shape -
  // the loop's scalar state spilled); the stage loop only knows that its last two refills fetch that block's first two stages
  constexpr int HT = (TAPS + 1) / 2;
  const int nblk = nmain + nsk, nstage = HALF ? (nmain - 1) * TAPS : nmain * TAPS + nsk;
  // block -> (64-channel block inside its source, the source's channel count); scalar
  auto blk_cbl = [&](int blk) {
    if (!SKIP || blk < nmain) return cb_begin + blk;
    const int j = blk - nmain;
    return j >= NCB2 ? j - NCB2 : j;
  };
  auto blk_C = [&](int blk) {
    if (!SKIP || blk < nmain) return a.Cin;
    return skip_arg(int(0), blk - nmain >= NCB2 ? offsetof(ConvK, C3) : offsetof(ConvK, C2));
  };

  // weight rows: lane owns physical slot lane&7 of row (lane>>3) of its 8-row group; one channel offset per lane
  const int coff = (((lane & 7) ^ ((4 * (wave & 1) + (lane >> 4)) & 7))) * 8;
  const int b_row0 = wave * 8 + (lane >> 3);             // group i adds 8*NW rows (a scalar step)
  const int b_base0 = (n0 * Ktot + coff) * SZ;   // + weight row * Ktot * SZ per load
  // halo group g: 8 halo rows hr = 8g + (lane>>3); the slot swizzle depends on hr, so the channel offset does too
  auto load_halo_group = [&](int g, int blk, int hbuf) {
    const int hr = g * 8 + (lane >> 3);
    const int C = blk_C(blk);
    const int ch = blk_cbl(blk) * 64 + (((lane & 7) ^ ((hr >> 1) & 7)) << 3);
    if constexpr (SKIP) {
      if (blk >= nmain) {
        // a skip block is read at the centre tap only: its A tile is the tile's 256 pixels as plain rows (row = tile-local pixel,
        // 32 groups instead of the halo's 50), the tile's pixels being contiguous in memory.  The source's descriptor is built
        // here from scalars (rebased like x's at the tile's first sample).
        const T* base = (const T*)skip_arg((const void*)nullptr, blk - nmain >= NCB2 ? offsetof(ConvK, x3) : offsetof(ConvK, x2)) +
                        (int64_t)b_first * hw * C;
        const uint64_t r = (uint64_t)(a.B - b_first) * hw * C * sizeof(T);
        auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, r > 0x7FFFFFFFull ? 0x7FFFFFFFu : (unsigned)r, 0x00020000);
        const int ok = (hr < BM) & (m0 + hr < a.M) & (ch < C);
        const unsigned voff = ok ? (unsigned)(((m0 - b_first * hw + hr) * C + ch) * SZ) : OOB_OFFSET;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(HB + hbuf * H_BYTES + g * 1024), 16, voff, 0, 0, 0);
        return;
      }
    }
    int sb = 0, rem = hr;                           // sample block of the tile, halo row inside it
    if (SPT > 1) { sb = hr / HRS; rem = hr - sb * HRS; }
    const int hy = rem / HWd, hx = rem - hy * HWd;
    // 2x2: the border is on the parity's side (recomputed from blockIdx.y here: two SALU ops instead of two loop-long SGPRs)
    const int iy = ty0 - (KT == 2 ? 1 - (int)(blockIdx.y >> 1) : 1) + hy, ix = hx - (KT == 2 ? 1 - (int)(blockIdx.y & 1) : 1);
    const int ok = (hr < NH) & ((unsigned)iy < (unsigned)H) & ((unsigned)ix < (unsigned)W) & (ch < C);
    // samples past the batch end fall outside the descriptor's extent and read as zeros
    const unsigned voff = ok ? (unsigned)(((sb * hw + iy * W + ix) * C + ch) * SZ) : OOB_OFFSET;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rsrc, (lds_void*)(HB + hbuf * H_BYTES + g * 1024), 16, voff, 0, 0, 0);
  };
  auto load_b_stage = [&](int blk, int tap, int bbuf, bool valid) {
    int cvalid, rstride, col, base = b_base0;
    auto rs = w_rsrc;
    if (!SKIP || blk < nmain) {
      const int c0 = (cb_begin + blk) * 64;
      cvalid = valid & (c0 + coff < a.Cin);
      rstride = Ktot * SZ;
      col = (tap * a.Cin + c0) * SZ;
    } else {   // w2 [Cout][C2 + C3]: columns of block j; a ragged last block of x2 must not read x3's columns
      const int C2 = skip_arg(int(0), offsetof(ConvK, C2));
      const int j = blk - nmain, Cs = C2 + skip_arg(int(0), offsetof(ConvK, C3));
      const bool second = j >= NCB2;
      const int col0 = second ? C2 + (j - NCB2) * 64 : j * 64;
      cvalid = valid & (col0 + coff < (second ? Cs : C2));
      rstride = Cs * SZ;
      col = col0 * SZ;
      base = (n0 * Cs + coff) * SZ;
      rs = __builtin_amdgcn_make_buffer_rsrc((void*)skip_arg((const void*)nullptr, offsetof(ConvK, w2)), 0,
                                             skip_arg(unsigned(0), offsetof(ConvK, w2_bytes)), 0x00020000);
    }
#pragma unroll
    for (int i = 0; i < B_PER; ++i) {
      const int g = i * NW + wave;
      if (g < GB) {
        const int brow = lds_row_chan<TN>(b_row0 + i * 8 * NW);  // LDS row -> weight row of the paired layout
        int ok = cvalid & (b_row0 + i * 8 * NW < BN) & (n0 + brow < a.Cout);
        unsigned voff = ok ? (unsigned)(base + brow * rstride + col) : OOB_OFFSET;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(BB + bbuf * B_BYTES + g * 1024), 16, voff, 0, 0, 0);
      }
    }
  };

  // HALF: a two-tap weight stage of the ragged (last) block: [tap | tap + 1] halves of its <= 32 channels - lanes of the logical
  // slots 4..7 read the next tap's columns (a per-lane column offset on the DMA source; everything else as load_b_stage)
  auto load_b_half = [&](int tap, int bbuf) {
    const int c0 = (NCB - 1) * 64, sel = coff >> 5;
    const int cvalid = (c0 + (coff & 31) < a.Cin) & (tap + sel < TAPS);
    const int col = (tap * a.Cin + c0 + sel * (a.Cin - 32)) * SZ;
#pragma unroll
    for (int i = 0; i < B_PER; ++i) {
      const int g = i * NW + wave;
      if (g < GB) {
        const int brow = lds_row_chan<TN>(b_row0 + i * 8 * NW);
        int ok = cvalid & (b_row0 + i * 8 * NW < BN) & (n0 + brow < a.Cout);
        unsigned voff = ok ? (unsigned)(b_base0 + brow * Ktot * SZ + col) : OOB_OFFSET;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_void*)(BB + bbuf * B_BYTES + g * 1024), 16, voff, 0, 0, 0);
      }
    }
  };

  const int lr = lane & 15, lg = lane >> 4;
  // A fragments: tile-local pixel m = wm*TM + 16 i + lr -> halo row of tap (0,0); 16 consecutive pixels share a row
  // fragment i sits 16 i pixels further: + 16 i halo rows and + 2 per image row crossed (scalar, W >= 16)
  int hr00;
  {
    int m = wm * TM + lr;
    int ty = m / W, tx = m - ty * W;
    int sb = SPT > 1 ? ty / RS : 0;  // a wave's TM = 64 pixels never straddle two samples (hw is a multiple of 64)
    hr00 = sb * HRS + (ty - sb * RS) * HWd + tx;
  }
  const int woff0 = (wn * TN + lr) * ROWB + ((lg ^ ((lr >> 1) & 7)) << 4);
  // `shift` = kh*(W+2) + kw moves every A fragment to the tap's halo row; it stays a loop-carried scalar so the
  // per-fragment swizzle (5 VALU) is recomputed per stage instead of being hoisted into ~70 loop-invariant VGPRs
  // `plain` (a skip block's tile): row = tile-local pixel wm*TM + 16 i + lr, no border; lr is recovered from woff0 (no new register)
  // c: 32-channel chunk of the weight stage; ca: chunk of the halo rows (= c but in the two-tap stages of HALF, which read chunk 0 twice)
  auto read_frags = [&](int hbuf, int bbuf, int shift, bool plain, int c, int ca, bf16x8(&xf)[FM], bf16x8(&wf)[FN]) {
    const unsigned char* ph = HB + hbuf * H_BYTES;
    const int vb = (SKIP && plain) ? wm * TM - wn * TN + (woff0 >> 7) : hr00 + shift;
    const int rj = (SKIP && plain) ? 0 : KT - 1;
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      int hr = vb + i * 16 + ((i * 16) / W) * rj;
      xf[i] = *reinterpret_cast<const bf16x8*>(ph + ((hr * ROWB + ((lg ^ ((hr >> 1) & 7)) << 4)) ^ (ca << 6)));
    }
    const unsigned char* pw = BB + bbuf * B_BYTES + (woff0 ^ (c << 6));
#pragma unroll
    for (int j = 0; j < FN; ++j) wf[j] = *reinterpret_cast<const bf16x8*>(pw + j * 2048);
  };

  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto mma = [&](bf16x8(&xf)[FM], bf16x8(&wf)[FN]) {
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xf[i], acc[i][j], 0, 0, 0);
  };

  bf16x8 xa[FM], wa[FN], xb[FM], wb[FN];

  if (ts) ts[1] = wall_clock64();
  if (nstage > 0 || HALF) {   // (split 0 always has blocks of x, so stages 0 and 1 are taps 0 and 1 of block 0)
    for (int g = wave; g < GH; g += NW) load_halo_group(g, 0, 0);
    if (HALF && nmain == 1) {   // the ragged block is the only one: its stages 0 and 1 are taps (0, 1) and (2, 3)
      load_b_half(0, 0);
      load_b_half(2, 1);
    } else {
      load_b_stage(0, 0, 0, true);
      load_b_stage(0, 1, 1, true);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (ts) ts[2] = wall_clock64();
    read_frags(0, 0, 0, false, 0, 0, xa, wa);
  }
  // one flat stage loop; (blk, tap, kw, shift, hbuf) are scalar state so nothing per-tap gets unrolled or hoisted
  int blk = 0, tap = 0, kw = 0, shift = 0, hbuf = 0;
#pragma unroll 1
  for (int st = 0; st < nstage; ++st) {
    const int bbuf = st & 1;
    const bool c1 = HALF || blk_cbl(blk) * 64 + 32 < blk_C(blk);   // (HALF: every block of this loop is whole)
    const bool one_stage = SKIP && blk >= nmain;   // a skip block: centre tap only
    // [A] chunk 0 of this stage on set a, prefetch chunk 1 into set b
    read_frags(hbuf, bbuf, shift, one_stage, 1, 1, xb, wb);
    mma(xa, wa);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // next weight stage + halo pieces issued a stage ago have landed; BB[bbuf] is free
    {  // refill right behind the barrier (issuing later - e.g. staggering the two waves of a SIMD - measured 10 % slower:
       // the pieces then land after the next barrier's vmcnt(0)): weight stage st+2 into BB[bbuf], one halo piece
      int t2 = tap + 2, b2 = blk;   // the stage two ahead: taps of this block, then of the next (a skip block is one stage long)
      if (one_stage) {
        t2 = 0; b2 = blk + 2;
      } else if (t2 >= TAPS) {
        t2 -= TAPS; ++b2;
        if (SKIP && b2 >= nmain) { b2 += t2; t2 = 0; }
      }
      if (HALF && b2 + 1 == nblk) load_b_half(2 * t2, bbuf);   // entering the ragged block: its stage t2 = taps 2 t2, 2 t2 + 1
      else load_b_stage(b2, t2, bbuf, b2 < nblk);
      // halo prefetch.  A nine-stage block fetches its successor's halo in eight pieces behind taps 0..7.  A one-stage block has
      // no such lead for ITS successor (the stage after the barrier already reads it), so one-stage blocks are fetched TWO blocks
      // ahead: whole, into the halo buffer this barrier has just released (the current block's chunk-1 reads are retired) - behind
      // the last barrier of the last nine-stage block and behind the barrier of every one-stage block.
      if (!one_stage && tap < TAPS - 1) {
        // (2x2: three stages carry the next halo, so a stage issues up to PPS groups per wave)
        constexpr int PPS = KT == 3 ? 1 : (HROWS / 8 + NW * (TAPS - 1) - 1) / (NW * (TAPS - 1));
#pragma unroll 1
        for (int pp = 0; pp < PPS; ++pp) {
          const int q = (tap * PPS + pp) * NW + wave;
          if (blk + 1 < nblk && q < ((SKIP && blk + 1 >= nmain) ? BM / 8 : GH)) load_halo_group(q, blk + 1, hbuf ^ 1);
        }
      } else if (SKIP && blk + 1 >= nmain && blk + 2 < nblk) {
        for (int g = wave; g < BM / 8; g += NW) load_halo_group(g, blk + 2, hbuf);
      }
    }
    // advance the scalar state to the next stage
    if (!one_stage) {
      ++tap;
      if (++kw == KT) { kw = 0; shift += HWd - (KT - 1); } else { ++shift; }
      if (tap == TAPS) { tap = 0; shift = 0; ++blk; hbuf ^= 1; }
    } else {
      ++blk; hbuf ^= 1;
    }
    // [B] chunk 1 on set b, prefetch chunk 0 of the next stage into set a
    if (HALF || st + 1 < nstage) read_frags(hbuf, bbuf ^ 1, shift, SKIP && blk >= nmain, 0, 0, xa, wa);   // (HALF: the tail's first chunk)
    if (c1) mma(xb, wb);
  }
  if constexpr (HALF) {
    // ---- the ragged block, two taps per stage (unrolled: taps, shifts and buffer parities are constants) --------------------------
    // chunk 0 = its channels at tap 2 j (halo chunk 0, weight half 0), chunk 1 = the SAME halo chunk at tap 2 j + 1 (weight half 1)
    const int par = nstage & 1;   // LDS weight buffer of its first stage
#pragma unroll
    for (int j = 0; j < HT; ++j) {
      const int t0 = 2 * j, t1 = 2 * j + 1;
      const int bbuf = par ^ (j & 1);
      const int sh1 = (t1 / KT) * HWd + (t1 % KT);
      if (t1 < TAPS) read_frags(hbuf, bbuf, sh1, false, 1, 0, xb, wb);
      mma(xa, wa);
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (t0 + 4 < TAPS) load_b_half(t0 + 4, bbuf);   // stage j + 2
      if (j + 1 < HT) {
        const int sh0 = ((t0 + 2) / KT) * HWd + ((t0 + 2) % KT);
        read_frags(hbuf, bbuf ^ 1, sh0, false, 0, 0, xa, wa);
      }
      if (t1 < TAPS) mma(xb, wb);
    }
  }
  if (ts) ts[3] = wall_clock64();
  if constexpr (FGN) conv_epilogue_fgn<FM, FN, BN, WM>(a, acc, HB, m0, n0, wm, wn, lane, tid, hw);
  else conv_epilogue<T, FM, FN, true>(a, acc, m0 + wm * TM, n0 + wn * TN, lr, lg, hw);
  if (ts) {
    ts[4] = wall_clock64();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ts[6] = wall_clock64();
  }
}

template <typename T, int BM, int BN, int WM, int WN, int KPS>
__global__ void __launch_bounds__(WM * WN * 64) conv_igemm_kernel(ConvK a) {
  using TL = Tile<T>;
  constexpr int NT = WM * WN * 64, NW = WM * WN;
  constexpr int EPS = TL::EPS, SLOTS = TL::SLOTS, ROWB = TL::ROWB;
  constexpr int TM = BM / WM, TN = BN / WN, FM = TM / 16, FN = TN / 16;
  static_assert(TM % 16 == 0 && TN % 16 == 0, "wave tile must be a multiple of 16");
  static_assert(BM % 16 == 0 && BN % 16 == 0, "block tile must be a multiple of 16");
  constexpr bool IS_BF16 = sizeof(T) == 2;
  constexpr int SZ = (int)sizeof(T);
  // per chunk: rows x SLOTS 16-byte items; item j of a chunk -> (row = j / SLOTS, slot = j % SLOTS).
  // bf16: item j is written by DMA lane (j % 64) of row-group (j / 64); fp32: by thread (j % NT), iteration j / NT.
  constexpr int A_CH_ITEMS = BM * SLOTS, B_CH_ITEMS = BN * SLOTS;
  constexpr int A_PER = IS_BF16 ? (BM / 16 + NW - 1) / NW : (A_CH_ITEMS + NT - 1) / NT;  // loads per thread per chunk
  constexpr int B_PER = IS_BF16 ? (BN / 16 + NW - 1) / NW : (B_CH_ITEMS + NT - 1) / NT;
  constexpr int A_BYTES = KPS * BM * ROWB, B_BYTES = KPS * BN * ROWB;
  constexpr int STAGE_BYTES = A_BYTES + B_BYTES;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const TileMN tmn = conv_tile_of_block(a.tile_map, a.tiles_n);
  const int tile_m = __builtin_amdgcn_readfirstlane(tmn.m), tile_n = __builtin_amdgcn_readfirstlane(tmn.n);
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int Ktot = a.KH * a.KW * a.Cin;
  const int ush = a.ups == 1, usw = a.ups != 0;  // fused nearest upsample: 1 = x2 both axes, 2 = x2 along W only
  const int Hlim = a.Hin << ush, Wlim = a.Win << usw;
  const int hw = a.Hout * a.Wout;

  // descriptors address at most 2 GiB: rebase the activation descriptor at the first sample this tile touches
  const int b_first = m0 / hw;
  const int64_t x_sample = (int64_t)a.Hin * a.Win * a.Cin;
  const T* xg = (const T*)a.x + (int64_t)blockIdx.y * a.x_bstride + (int64_t)b_first * x_sample;
  const uint64_t x_rem = a.x_bytes - (uint64_t)b_first * x_sample * sizeof(T);
  const unsigned x_extent = x_rem > 0x7FFFFFFFull ? 0x7FFFFFFFu : (unsigned)x_rem;
  const T* wg = (const T*)a.w + (int64_t)blockIdx.y * a.w_bstride;
  auto x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)xg, 0, x_extent, 0x00020000);
  auto w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)wg, 0, a.w_bytes, 0x00020000);

  // ---- per-thread gather descriptors (fixed over the K loop) ---------------------------------------
  // A rows (pixels)
  int a_ih0[A_PER], a_iw0[A_PER], a_base[A_PER], a_coff[A_PER], a_pix[A_PER];
  bool a_ok[A_PER];
  int a_lds[A_PER];  // fp32 path: LDS byte offset inside a chunk
#pragma unroll
  for (int i = 0; i < A_PER; ++i) {
    int j = IS_BF16 ? (i * NW + wave) * 64 + lane : i * NT + tid;
    int row = j / SLOTS, pslot = j % SLOTS;
    int slot = IS_BF16 ? swz(row, pslot) : pslot;  // DMA: lane owns PHYSICAL slot pslot, fetch the logical one
    int m = m0 + row;
    bool ok = (j < A_CH_ITEMS) && (m < a.M);
    int b = ok ? m / hw : 0;
    int r = ok ? m - b * hw : 0;
    int oh = r / a.Wout, ow = r - oh * a.Wout;
    a_ok[i] = ok;
    a_ih0[i] = oh * a.stride - conv_pad_h(a);
    a_iw0[i] = ow * a.stride - conv_pad_w(a);
    a_coff[i] = slot * EPS;
    a_pix[i] = (ok ? b - b_first : 0) * a.Hin * a.Win;  // pixel index of the sample's first pixel
    a_base[i] = ((a_pix[i] + a_ih0[i] * a.Win + a_iw0[i]) * a.Cin + a_coff[i]) * SZ;
    a_lds[i] = row * ROWB + pslot * 16;
  }
  // B rows (output channels)
  int b_base[B_PER], b_coff[B_PER], b_lds[B_PER];
  bool b_ok[B_PER];
#pragma unroll
  for (int i = 0; i < B_PER; ++i) {
    int j = IS_BF16 ? (i * NW + wave) * 64 + lane : i * NT + tid;
    int row = j / SLOTS, pslot = j % SLOTS;
    int slot = IS_BF16 ? swz(row, pslot) : pslot;
    int n = n0 + row;
    b_ok[i] = (j < B_CH_ITEMS) && (n < a.Cout);
    b_coff[i] = slot * EPS;
    b_base[i] = (n * Ktot + b_coff[i]) * SZ;
    b_lds[i] = row * ROWB + pslot * 16;
  }

  // ---- wave-uniform walk over (tap, channel-chunk) -------------------------------------------------------
  const int s_begin = blockIdx.z * a.SPS;
  const int s_end = min(a.NS, s_begin + a.SPS);
  int kc_next = s_begin * KPS;  // first chunk of the next stage to load
  int q_tap = kc_next / a.CPT;
  int q_ci = kc_next - q_tap * a.CPT;
  int q_kh = q_tap / a.KW, q_kw = q_tap - q_kh * a.KW;

  u32x4 ra[KPS][A_PER], rb[KPS][B_PER];  // fp32 register staging only

  // issue the loads of one stage into LDS buffer `buf` (bf16: DMA) or into registers (fp32)
  auto load_stage = [&](int buf) {
#pragma unroll
    for (int c = 0; c < KPS; ++c) {
      const bool kvalid = kc_next < a.NK;
      const int c0 = q_ci * 32;
      const int a_delta = ((q_kh * a.Win + q_kw) * a.Cin + c0) * SZ;
      const int b_delta = ((q_kh * a.KW + q_kw) * a.Cin + c0) * SZ;
#pragma unroll
      for (int i = 0; i < A_PER; ++i) {
        int ih = a_ih0[i] + q_kh, iw = a_iw0[i] + q_kw;
        bool ok = a_ok[i] && kvalid && (c0 + a_coff[i] < a.Cin) && (unsigned)ih < (unsigned)Hlim &&
                  (unsigned)iw < (unsigned)Wlim;
        int off = a_base[i] + a_delta;
        if (a.ups) off = ((a_pix[i] + (ih >> ush) * a.Win + (iw >> usw)) * a.Cin + c0 + a_coff[i]) * SZ;
        unsigned voff = ok ? (unsigned)off : OOB_OFFSET;
        if constexpr (IS_BF16) {
          if ((i * NW + wave) * 16 < BM) {  // wave-uniform: this row group exists
            unsigned char* dst = smem + buf * STAGE_BYTES + c * (BM * ROWB) + (i * NW + wave) * 1024;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rsrc, (lds_void*)dst, 16, voff, 0, 0, 0);
          }
        } else {
          ra[c][i] = __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, voff, 0, 0);
        }
      }
#pragma unroll
      for (int i = 0; i < B_PER; ++i) {
        bool ok = b_ok[i] && kvalid && (c0 + b_coff[i] < a.Cin);
        unsigned voff = ok ? (unsigned)(b_base[i] + b_delta) : OOB_OFFSET;
        if constexpr (IS_BF16) {
          if ((i * NW + wave) * 16 < BN) {
            unsigned char* dst = smem + buf * STAGE_BYTES + A_BYTES + c * (BN * ROWB) + (i * NW + wave) * 1024;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_void*)dst, 16, voff, 0, 0, 0);
          }
        } else {
          rb[c][i] = __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, voff, 0, 0);
        }
      }
      // advance the uniform walk by one chunk
      ++kc_next;
      if (++q_ci == a.CPT) {
        q_ci = 0;
        if (++q_kw == a.KW) {
          q_kw = 0;
          ++q_kh;
        }
      }
    }
  };
  auto store_stage = [&](int buf) {  // fp32 path: registers -> LDS
    if constexpr (!IS_BF16) {
      unsigned char* base = smem + buf * STAGE_BYTES;
#pragma unroll
      for (int c = 0; c < KPS; ++c) {
#pragma unroll
        for (int i = 0; i < A_PER; ++i)
          if (i * NT + tid < A_CH_ITEMS) *reinterpret_cast<u32x4*>(base + c * (BM * ROWB) + a_lds[i]) = ra[c][i];
#pragma unroll
        for (int i = 0; i < B_PER; ++i)
          if (i * NT + tid < B_CH_ITEMS)
            *reinterpret_cast<u32x4*>(base + A_BYTES + c * (BN * ROWB) + b_lds[i]) = rb[c][i];
      }
    }
  };

  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int lr = lane & 15, lg = lane >> 4;

  if (s_begin < s_end) {
    load_stage(0);
    store_stage(0);
  }
  __syncthreads();  // (waits for the DMA: an LDS-DMA in flight makes the barrier's fence a vmcnt(0))

  for (int s = s_begin; s < s_end; ++s) {
    const int buf = (s - s_begin) & 1;
    if (s + 1 < s_end) load_stage(buf ^ 1);  // bf16: DMA straight into the other buffer, overlapping the MFMAs below
    const unsigned char* sa = smem + buf * STAGE_BYTES;
    const unsigned char* sb = sa + A_BYTES;
#pragma unroll
    for (int c = 0; c < KPS; ++c) {
      if constexpr (IS_BF16) {
        bf16x8 xf[FM], wf[FN];
#pragma unroll
        for (int i = 0; i < FM; ++i) {
          int row = wm * TM + i * 16 + lr;
          xf[i] = *reinterpret_cast<const bf16x8*>(sa + c * (BM * ROWB) + row * ROWB + swz(row, lg) * 16);
        }
#pragma unroll
        for (int j = 0; j < FN; ++j) {
          int row = wn * TN + j * 16 + lr;
          wf[j] = *reinterpret_cast<const bf16x8*>(sb + c * (BN * ROWB) + row * ROWB + swz(row, lg) * 16);
        }
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int j = 0; j < FN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xf[i], acc[i][j], 0, 0, 0);
      } else {
#pragma unroll
        for (int h = 0; h < 2; ++h) {  // two 16-deep sub-blocks of the 32-channel chunk
          f32x4 xf[FM], wf[FN];
#pragma unroll
          for (int i = 0; i < FM; ++i) {
            int row = wm * TM + i * 16 + lr;
            xf[i] = *reinterpret_cast<const f32x4*>(sa + c * (BM * ROWB) + row * ROWB + (4 * h + lg) * 16);
          }
#pragma unroll
          for (int j = 0; j < FN; ++j) {
            int row = wn * TN + j * 16 + lr;
            wf[j] = *reinterpret_cast<const f32x4*>(sb + c * (BN * ROWB) + row * ROWB + (4 * h + lg) * 16);
          }
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
              for (int j = 0; j < FN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[j][e], xf[i][e], acc[i][j], 0, 0, 0);
        }
      }
    }
    if (s + 1 < s_end) store_stage(buf ^ 1);
    __syncthreads();
  }

  conv_epilogue<T, FM, FN>(a, acc, m0 + wm * TM, n0 + wn * TN, lr, lg, hw);
}

// y = (sum_z ws[z] + bias + emb + res) * scale, one thread per 4 channels
template <typename T>
__global__ void __launch_bounds__(256) splitk_finish_kernel(ConvK a) {
  int64_t nvec = (int64_t)a.M * a.Cout / 4;
  int hw = a.Hout * a.Wout;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) {
    int64_t e0 = i * 4;
    int m = (int)(e0 / a.Cout);
    int n = (int)(e0 - (int64_t)m * a.Cout);
    int b = m / hw;
    f32x4 v = *reinterpret_cast<const f32x4*>(a.ws + e0);
    for (int z = 1; z < a.SPS; ++z) v += *reinterpret_cast<const f32x4*>(a.ws + (int64_t)z * a.M * a.Cout + e0);
    if (a.bias) v += *reinterpret_cast<const f32x4*>(a.bias + n);
    if (a.emb) v += *reinterpret_cast<const f32x4*>(a.emb + (int64_t)(a.emb_mod ? b % a.emb_mod : b) * a.emb_stride + n);
    if (a.res) {
      const T* p = (const T*)a.res + e0;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] += to_f32(p[e]);
    }
    v *= a.out_scale;
    if (a.y_nchw) {
      int r = m - b * hw;
#pragma unroll
      for (int e = 0; e < 4; ++e) a.y_nchw[((int64_t)b * a.Cout + n + e) * hw + r] = v[e];
    } else {
      T* p = (T*)a.y + e0;
#pragma unroll
      for (int e = 0; e < 4; ++e) p[e] = from_f32<T>(v[e]);
    }
  }
}

// ------------------------------------------------------------------------------------------------------
// host side: tile selection and launch
// ------------------------------------------------------------------------------------------------------
// ---- configurations ------------------------------------------------------------------------------------
// bf16 ring kernel: id -> <BM, BN, WM, WN, KPS, NST>
struct RingCfg { int bm, bn, kps; };
static const RingCfg kRing[] = {
    {128, 112, 1},  // 0: <128,112,4,1,1,5>  77 KB LDS -> 2 workgroups / CU
    {128, 112, 2},  // 1: <128,112,4,1,2,4> 123 KB
    {256, 112, 2},  // 2: <256,112,8,1,2,3> 146 KB, 8 waves
    {128, 224, 2},  // 3: <128,224,4,2,2,3> 140 KB, 8 waves
    {128, 128, 2},  // 4: <128,128,2,2,2,4> 132 KB
    {128, 64, 2},   // 5: <128, 64,4,1,2,4>  100 KB
    {64, 128, 2},   // 6: < 64,128,2,2,2,4>  100 KB
    {64, 64, 2},    // 7: < 64, 64,2,2,2,4>   68 KB
    {256, 128, 2},  // 8: <256,128,4,2,2,3> 152 KB, 8 waves
    {128, 224, 1},  // 9: <128,224,2,2,1,3>  72 KB, per-wave 64x112
    {256, 112, 1},  // 10: <256,112,4,1,1,3> 75 KB, per-wave 64x112
    {256, 224, 1},  // 11: <256,224,4,2,1,4> 131 KB, 8 waves, per-wave 64x112
    {128, 224, 1},  // 12: <128,224,2,2,1,4>  94 KB
    {256, 224, -1},  // 13: pipe<256,224,4,2> 8 waves, per-wave 64x112, 124 KB   (kps -1: pipe kernel, 64-channel stages)
    {128, 224, -1},  // 14: pipe<128,224,2,2> 4 waves, per-wave 64x112,  92 KB
    {256, 112, -1},  // 15: pipe<256,112,4,1> 4 waves, per-wave 64x112,  96 KB
    {256, 128, -1},  // 16: pipe<256,128,4,2> 8 waves, per-wave 64x64,  104 KB
    {128, 128, -1},  // 17: pipe<128,128,2,2> 4 waves, per-wave 64x64,   68 KB
    {128, 64, -1},   // 18: pipe<128, 64,2,1> 2 waves, per-wave 64x64,   50 KB
    {256, 224, -2},  // 19: halo<224,4,2> 8 waves, per-wave 64x112, 155 KB  (kps -2: halo kernel, 3x3 s1 p1, W in 16..64)
    {256, 128, -2},  // 20: halo<128,4,2> 8 waves, per-wave 64x64,  131 KB
    {256, 224, 1},   // 21: ring <256,224,4,2,1,4> of 11 with the refill's DMA instructions interleaved with the MFMA groups
    {128, 224, 1},   // 22: ring <128,224,2,2,1,4> of 12, interleaved
    {256, 224, 1},   // 23: ring <256,224,4,2,1,5> interleaved, one stage deeper (158 KB)
    {256, 224, -3},  // 24: gemm1x1<224,4,2> PERSISTENT 1x1 kernel, 8 waves, per-wave 64x112, 120 KB  (kps -3; other shapes fall back to 13)
    {256, 128, -3},  // 25: gemm1x1<128,4,2> the same for channel counts that are multiples of 128 (tri-plane UNet, decoders), per-wave 64x64
    {256, 256, -2},  // 26: halo<256,2,4> 8 waves, per-wave 128x64, 150 KB: the 256-wide tile for the model_channels = 256 networks (W = 16 / 32)
    {256, 256, -1},  // 27: pipe<256,256,2,4> the same tile on the pipe kernel (their levels below 16 pixels, stride 2, upsampling), 136 KB
    {256, 256, -3},  // 28: gemm1x1<256,2,4> ... and on the persistent 1x1 kernel (two activation slots), 128 KB + bias
};
bool conv_halo_shape_ok(int H, int W) {
  if (!(W == 8 || W == 16 || W == 32 || W == 64)) return false;
  const int hw = H * W;
  if (hw % 256 == 0) return true;                       // TR whole rows of one sample
  if (hw != 64 && hw != 128) return false;              // whole samples per tile, 64 pixels per wave inside one sample
  return (256 / hw) * (H + 2) * (W + 2) <= HALO_MAX_ROWS;
}
// the 256-wide halo tile (configuration 26) keeps 344 halo rows: whole image rows of one sample at W = 16 / 32
static bool halo256_shape_ok(int H, int W) { return (W == 16 || W == 32) && (H * W) % 256 == 0; }
static bool halo_ok(const ConvArgs& a) {
  // the parity form of upsample + 3x3: 2x2 taps on the source grid (the KT = 2 instantiation)
  if (a.ups == 3) return a.KH == 2 && a.KW == 2 && a.pad == 1 && !a.x2 && conv_halo_shape_ok(a.Hin, a.Win);
  if (!(a.KH == 3 && a.KW == 3 && a.stride == 1 && a.pad == 1 && !a.ups && a.Hout == a.Hin && a.Wout == a.Win)) return false;
  return conv_halo_shape_ok(a.Hin, a.Win);
}
// diagnostics (tools/prof_attn_shapes.py): DDMI_PROF_SHAPES=1 appends the convolution's shape to the profiler scope's name
static std::string conv_prof_name(const std::string& base, const ConvK& k) {
  if (!prof_active() || !getenv("DDMI_PROF_SHAPES") || atoi(getenv("DDMI_PROF_SHAPES")) != 1) return base;
  char buf[128];
  snprintf(buf, sizeof buf, "[M%d,%dx%d,Cin%d,Cout%d,k%d,s%d,u%d]", k.M, k.Hout, k.Wout, k.Cin, k.Cout, k.KH, k.stride, k.ups);
  return base + buf;
}

// a launch with a folded skip connection exists on the halo kernel only: the tuned / heuristic tile when it is one, else the
// halo configuration that fits Cout
static int skip_fold_cfg(const ConvArgs& a) { return a.Cout > 128 ? 19 : 20; }
static const int kHaloFallback = 13;
static const int kNumRing = sizeof(kRing) / sizeof(kRing[0]);
struct TileCfg { int bm, bn, kps, id; };  // id < 0: fp32 kernel tile
static int g_force_bm = 0, g_force_bn = 0;  // diagnostics (ddmi_bench_conv): bn == -1 -> bm is a ring config id
void conv_force_tile(int bm, int bn) { g_force_bm = bm; g_force_bn = bn; }
static int g_halo_half = -1;   // ddmi_debug_halo_half: -1 = the DDMI_HALO_HALF environment decides (default on)
void conv_halo_half_override(int mode) { g_halo_half = mode; }

static TileCfg pick_tile(int M, int Cout, int dt, int cfg = 0) {
  if (dt == DT_BF16) {
    int id;
    static const int env_cfg = getenv("DDMI_CONV_CFG") ? atoi(getenv("DDMI_CONV_CFG")) : -1;  // diagnostics
    if (cfg > 0 && cfg <= kNumRing) id = cfg - 1;
    else if (env_cfg >= 0 && env_cfg < kNumRing) id = env_cfg;
    else if (g_force_bn == -1 && g_force_bm >= 0 && g_force_bm < kNumRing) id = g_force_bm;
    else if (Cout % 112 == 0) id = cdiv(M, 128) * cdiv(Cout, 224) >= 384 ? 11 : 3;
    else if (Cout <= 64) id = cdiv(M, 128) >= 64 ? 5 : 7;
    else id = cdiv(M, 128) * cdiv(Cout, 128) >= 128 ? 4 : 6;
    return {kRing[id].bm, kRing[id].bn, kRing[id].kps, id};
  }
  int bn;
  if (Cout % 112 == 0) bn = 112;
  else if (Cout % 128 == 0) bn = 128;
  else if (Cout <= 64) bn = 64;
  else {
    int best = 128, waste = cdiv(Cout, 128) * 128 - Cout;  // least padding among the compiled widths
    int w112 = cdiv(Cout, 112) * 112 - Cout, w64 = cdiv(Cout, 64) * 64 - Cout;
    if (w112 < waste) { best = 112; waste = w112; }
    if (w64 < waste) { best = 64; waste = w64; }
    bn = best;
  }
  int tiles128 = cdiv(M, 128) * cdiv(Cout, bn);
  int bm = (tiles128 >= 128 || bn == 112) ? 128 : 64;
  return {bm, bn, 1, -1};
}

static int make_k(const ConvArgs& a, int dt, int kps, int splitk, ConvK* out) {
  ConvK k{};
  k.x = a.x; k.w = a.w; k.bias = a.bias; k.emb = a.emb; k.res = a.res; k.y = a.y; k.y_nchw = a.y_nchw;
  k.ws = (splitk > 1 || a.force_ws) ? a.splitk_ws : nullptr;
  k.chstat = nullptr;   // launch_conv sets it after checking that every wave takes the fast epilogue
  k.B = a.B; k.Hin = a.Hin; k.Win = a.Win; k.Cin = a.Cin; k.Hout = a.Hout; k.Wout = a.Wout; k.Cout = a.Cout;
  k.KH = a.KH; k.KW = a.KW; k.stride = a.stride; k.pad = a.pad; k.ups = a.ups;
  k.emb_stride = a.emb_stride; k.emb_mod = a.emb_mod;
  k.M = a.B * a.Hout * a.Wout;
  if (a.ups == 3) {   // four parity convolutions on the source grid (conv_out_row scatters the rows)
    k.up4 = 1; k.ups = 0; k.Hout = a.Hin; k.Wout = a.Win; k.M = a.B * a.Hin * a.Win;
    k.w_bstride = (int64_t)a.Cout * a.KH * a.KW * a.Cin;
  }
  k.CPT = cdiv(a.Cin, 32);
  k.NK = k.CPT * a.KH * a.KW;
  k.NS = kps < 0 ? cdiv(a.Cin, 64) * a.KH * a.KW : cdiv(k.NK, kps);
  k.SPS = kps == -2 ? a.KH * a.KW * cdiv(cdiv(a.Cin, 64), splitk) : cdiv(k.NS, splitk);  // halo: whole channel blocks per split
  k.out_scale = a.out_scale;
  static const int dbg = getenv("DDMI_CONV_DBG") ? atoi(getenv("DDMI_CONV_DBG")) : 0;
  k.dbg = dbg;
  uint64_t xb = (uint64_t)a.B * a.Hin * a.Win * a.Cin * dt_size(dt);
  uint64_t wb = (uint64_t)a.Cout * a.KH * a.KW * a.Cin * dt_size(dt);
  // weights and the few samples one tile can touch must fit a 2 GiB descriptor; the whole activation need not
  uint64_t per_sample = (uint64_t)a.Hin * a.Win * a.Cin * dt_size(dt);
  uint64_t span = (uint64_t)(256 / std::max(1, k.Hout * k.Wout) + 2) * per_sample;
  DDMI_REQUIRE(wb < 0x80000000ull && (span < 0x7FFFFFFFull || xb < 0x7FFFFFFFull), "conv: tensor exceeds the 2 GiB buffer-descriptor range");
  k.x_bytes = xb;
  k.w_bytes = (unsigned)wb;
  k.x2 = a.x2; k.x3 = a.C3 > 0 ? a.x3 : nullptr; k.w2 = a.w2; k.C2 = a.x2 ? a.C2 : 0; k.C3 = a.x2 && a.x3 ? a.C3 : 0;
  k.w2_bytes = (unsigned)((uint64_t)a.Cout * (k.C2 + k.C3) * dt_size(dt));
  k.y32 = a.y32; k.res32 = a.res32;
  if (a.xcat) { k.xcat = a.xcat; k.C1 = a.Cin - a.Ccat; }
  if (a.gn_gamma) {
    k.gn_gamma = a.gn_gamma; k.gn_beta = a.gn_beta; k.gn_eps = a.gn_eps; k.gn_silu = a.gn_silu;
    k.gn_cpg = a.Cout / 32; k.gn_tps = (a.Hout * a.Wout) / 256;
    k.gn_part = a.gn_part; k.gn_sync = a.gn_sync; k.gn_flag_host = a.gn_flag_host;
  }
  *out = k;
  return DDMI_OK;
}

// GEMM rows of one launch slice: output pixels, or SOURCE pixels per parity for the parity form of upsample + 3x3
static int conv_rows(const ConvArgs& a) { return a.ups == 3 ? a.B * a.Hin * a.Win : a.B * a.Hout * a.Wout; }

int conv_pick_splitk(const ConvArgs& a, int dt) {
  if (a.ups == 3) return 1;
  int M = a.B * a.Hout * a.Wout;
  TileCfg t = pick_tile(M, a.Cout, dt, a.cfg);
  if (a.x2 && t.kps != -2) t = pick_tile(M, a.Cout, dt, skip_fold_cfg(a) + 1);
  if (t.kps == -3) return 1;   // the persistent 1x1 kernel has no split-K
  int tiles = cdiv(M, t.bm) * cdiv(a.Cout, t.bn);
  int NS = t.kps < 0 ? cdiv(a.Cin, 64) * a.KH * a.KW : cdiv(cdiv(a.Cin, 32) * a.KH * a.KW, t.kps);
  if (tiles >= 192 || NS < 16) return 1;
  int want = cdiv(384, tiles);
  int maxs = std::max(1, NS / 8);
  return std::max(1, std::min(std::min(want, maxs), 16));
}

static void conv_work(const ConvK& k, int batch, int esz, double* flops, double* bytes) {
  // algorithmic work of one launch: 2*M*N*K flops; operands + output once
  const double Kdim = (double)k.KH * k.KW * k.Cin, Mtot = (double)k.M * batch;
  const double Ks = k.x2 ? (double)(k.C2 + k.C3) : 0.0;   // folded skip connection: Ks more K columns, its sources read once
  *flops = 2.0 * Mtot * k.Cout * (Kdim + Ks);
  *bytes = esz * ((double)batch * k.B * k.Hin * k.Win * k.Cin + (double)k.Cout * (Kdim + Ks) * (k.w_bstride ? batch : 1) + Mtot * (k.Cout + Ks));
}

template <typename T, int BM, int BN, int WM, int WN, int KPS>
static int launch_cfg(ConvK k, int splitk, int batch, hipStream_t s) {
  constexpr int stage = KPS * (BM + BN) * Tile<T>::ROWB;
  k.tiles_n = cdiv(k.Cout, BN);
  dim3 grid(cdiv(k.M, BM) * k.tiles_n, batch, splitk);
  auto kern = conv_igemm_kernel<T, BM, BN, WM, WN, KPS>;
  static bool attr_set = false;
  if (!attr_set) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                2 * stage));
    attr_set = true;
  }
  {
    double flops, bytes;
    conv_work(k, batch, sizeof(T), &flops, &bytes);
    static const std::string name = std::string("conv_igemm<") + (sizeof(T) == 2 ? "bf16" : "f32") + "," +
                                    std::to_string(BM) + "x" + std::to_string(BN) + ">";
    PROF_SCOPE(conv_prof_name(name, k).c_str(), flops, bytes, s);
    hipLaunchKernelGGL(kern, grid, dim3(WM * WN * 64), 2 * stage, s, k);
  }
  HIP_TRY(hipGetLastError());
  return DDMI_OK;
}

// DDMI_CONV_TS=1: the 6th ring launch of the process prints, for a few CUs, what each workgroup that ran there did when
// (100 MHz wall clock, us since the kernel's first workgroup): index setup, first stage landed, K loop, epilogue issue,
// store acknowledgement, and the gap to the next workgroup of the same slot.
static void conv_dump_timeline(const unsigned long long* ts_dev, int n, hipStream_t s) {
  std::vector<unsigned long long> h((size_t)n * 8);
  if (hipStreamSynchronize(s) != hipSuccess) return;
  if (hipMemcpy(h.data(), ts_dev, h.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) return;
  unsigned long long t_min = ~0ull, t_max = 0;
  for (int i = 0; i < n; ++i) { t_min = std::min(t_min, h[8 * i]); t_max = std::max(t_max, h[8 * i + 6]); }
  struct Rec { unsigned key; int wg; };
  std::vector<Rec> order(n);
  for (int i = 0; i < n; ++i) {
    unsigned hw = (unsigned)h[8 * i + 5], xcc = (unsigned)(h[8 * i + 5] >> 32) & 15;
    unsigned cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    order[i] = {(xcc << 12) | (se << 8) | (sh << 4) | cu, i};
  }
  std::sort(order.begin(), order.end(), [&](const Rec& a, const Rec& b) {
    return a.key != b.key ? a.key < b.key : h[8 * a.wg] < h[8 * b.wg];
  });
  fprintf(stderr, "[conv timeline] %d workgroups, kernel span %.1f us\n", n, (t_max - t_min) / 100.0);
  int cus = 0;
  double sum[6] = {0, 0, 0, 0, 0, 0};
  for (int i = 0; i < n; ++i) {
    const unsigned long long* t = &h[8 * order[i].wg];
    for (int q = 0; q < 4; ++q) sum[q] += (t[q + 1] - t[q]) / 100.0;
    sum[4] += (t[6] - t[4]) / 100.0;
    sum[5] += (t[6] - t[0]) / 100.0;
  }
  fprintf(stderr, "  mean per workgroup: setup %.2f  first-stage %.2f  k-loop %.2f  epilogue-issue %.2f  store-ack %.2f  total %.2f us\n",
          sum[0] / n, sum[1] / n, sum[2] / n, sum[3] / n, sum[4] / n, sum[5] / n);
  for (int i = 0; i < n && cus < 2; ++i) {
    if (i && order[i].key != order[i - 1].key) ++cus;
    if (cus >= 2) break;
    const unsigned long long* t = &h[8 * order[i].wg];
    unsigned hw = (unsigned)t[5];
    fprintf(stderr, "  cu %05x tg %u wg %5d: start %8.2f  setup %5.2f  first-stage %5.2f  k-loop %6.2f  epilogue %5.2f  ack %5.2f  end %8.2f\n",
            order[i].key, (hw >> 16) & 15, order[i].wg, (t[0] - t_min) / 100.0, (t[1] - t[0]) / 100.0, (t[2] - t[1]) / 100.0,
            (t[3] - t[2]) / 100.0, (t[4] - t[3]) / 100.0, (t[6] - t[4]) / 100.0, (t[6] - t_min) / 100.0);
  }
}

static unsigned long long* conv_ts_buffer(int n_workgroups) {
  static const bool want = getenv("DDMI_CONV_TS") != nullptr;
  static unsigned long long* dev = nullptr;
  if (!want || n_workgroups > 65536) return nullptr;
  if (!dev && hipMalloc(&dev, sizeof(unsigned long long) * 8 * 65536) != hipSuccess) return nullptr;
  return dev;
}

template <int BM, int BN, int WM, int WN, int KPS, int NST, bool ILV = false>
static int launch_ring(ConvK k, int splitk, int batch, hipStream_t s) {
  constexpr int lds = NST * KPS * (BM + BN) * 64 + WM * WN * 1024;
  static_assert(lds <= 160 * 1024, "ring does not fit the 160 KiB LDS");
  k.tiles_n = cdiv(k.Cout, BN);
  dim3 grid(cdiv(k.M, BM) * k.tiles_n, batch, splitk);
  auto kern = conv_ring_kernel<BM, BN, WM, WN, KPS, NST, ILV>;
  static bool attr_set = false;
  if (!attr_set) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    attr_set = true;
  }
  static int ts_launch = 0;
  k.ts = conv_ts_buffer((int)grid.x);
  {
    double flops, bytes;
    conv_work(k, batch, 2, &flops, &bytes);
    static const std::string name = "conv_ring<bf16," + std::to_string(BM) + "x" + std::to_string(BN) + ",k" +
                                    std::to_string(KPS * 32) + "x" + std::to_string(NST) + (ILV ? ",ilv>" : ">");
    PROF_SCOPE(conv_prof_name(name, k).c_str(), flops, bytes, s);
    hipLaunchKernelGGL(kern, grid, dim3(WM * WN * 64), lds, s, k);
  }
  HIP_TRY(hipGetLastError());
  if (k.ts && ++ts_launch == 6) conv_dump_timeline(k.ts, (int)grid.x, s);
  return DDMI_OK;
}

template <int BM, int BN, int WM, int WN>
static int launch_pipe(ConvK k, int splitk, int batch, hipStream_t s) {
  constexpr int lds = 2 * (BM + BN) * 128 + WM * WN * 1024;
  static_assert(lds <= 160 * 1024, "pipe tile does not fit the 160 KiB LDS");
  k.tiles_n = cdiv(k.Cout, BN);
  dim3 grid(cdiv(k.M, BM) * k.tiles_n, batch, splitk);
  // two instantiations: plain, and PAIRED fragments (16-byte epilogue accesses: +7 % on the 1x1 convs whose tiles are
  // mostly prologue / epilogue, -4 % on long-K 3x3 tiles from two more VALU per weight load); the autotuner picks
  auto kern_plain = conv_pipe_kernel<BM, BN, WM, WN, false>;
  auto kern_paired = conv_pipe_kernel<BM, BN, WM, WN, true>;
  static bool attr_set = false;
  if (!attr_set) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern_plain), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern_paired), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    attr_set = true;
  }
  {
    double flops, bytes;
    conv_work(k, batch, 2, &flops, &bytes);
    static const std::string name = "conv_pipe<bf16," + std::to_string(BM) + "x" + std::to_string(BN) + ">";
    PROF_SCOPE(conv_prof_name(name, k).c_str(), flops, bytes, s);
    if (k.paired) hipLaunchKernelGGL(kern_paired, grid, dim3(WM * WN * 64), lds, s, k);
    else hipLaunchKernelGGL(kern_plain, grid, dim3(WM * WN * 64), lds, s, k);
  }
  HIP_TRY(hipGetLastError());
  return DDMI_OK;
}

bool conv_cat_ok(const ConvArgs& a, int dt);
// shapes the persistent 1x1 kernel (configuration 24) serves; anything else asked of it runs on the pipe kernel (13)
static bool gemm1x1_ok(const ConvArgs& a, int bn = 224) {
  return a.KH == 1 && a.KW == 1 && a.stride == 1 && !a.ups && a.splitk <= 1 && !a.force_ws && a.Cout % bn == 0 && a.y_nchw == nullptr &&
         a.y32 == nullptr && a.x2 == nullptr && a.Cin % 8 == 0 && a.y != nullptr && a.Cin > 128 &&
         (2 * 256 + 2 * bn) * 128 + a.Cout * 4 <= 160 * 1024 &&   // two activation slots, two weight stages and the bias vector in LDS
         (a.emb == nullptr || (a.Hout * a.Wout) % 64 == 0);   // (the fast epilogue's conditions; at least three stages per tile)
}
static const int kGemmFallback = 13;
// a 1x1 convolution over [x | xcat]: bf16, no split-K, plain NHWC output, whole 224- or 128-wide channel tiles, 16-byte channel vectors
bool conv_cat_ok(const ConvArgs& a, int dt) {
  if (dt != DT_BF16 || !a.xcat || a.Ccat <= 0 || a.Ccat >= a.Cin || a.Ccat % 8 || (a.Cin - a.Ccat) % 8) return false;
  if (a.splitk > 1 || a.chstat || a.x2 || a.emb || a.Cout > 2016 * 4) return false;
  return gemm1x1_ok(a, 224) || gemm1x1_ok(a, 128);
}
template <int BN, int WM, int WN>
static int launch_gemm1x1(ConvK k, hipStream_t s) {
  // three activation slots where the bias vector leaves room for them (Cout <= 2016), else two
  static const int force_na = getenv("DDMI_GEMM1X1_NA") ? atoi(getenv("DDMI_GEMM1X1_NA")) : 0;   // diagnostics
  const int lds3 = (3 * 256 + 2 * BN) * 128 + k.Cout * 4, lds2 = (2 * 256 + 2 * BN) * 128 + k.Cout * 4;
  const bool na3 = lds3 <= 160 * 1024 && force_na != 2;
  const int lds = na3 ? lds3 : lds2;
  DDMI_REQUIRE(lds <= 160 * 1024, "gemm1x1: Cout too large for the bias in LDS");
  k.tiles_n = cdiv(k.Cout, BN);
  const int tiles = cdiv(k.M, 256) * k.tiles_n;
  static int n_cu = 0;
  if (!n_cu) {
    int dev = 0;
    hipDeviceProp_t prop;
    n_cu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
  }
  dim3 grid(std::min(tiles, n_cu));   // one workgroup per CU walks its share of the tiles
  // STATS: + the consumer GroupNorm's statistics out of the epilogue
  void (*kerns[4])(ConvK) = {conv_gemm1x1_kernel<BN, WM, WN, false, 2>, conv_gemm1x1_kernel<BN, WM, WN, true, 2>,
                             conv_gemm1x1_kernel<BN, WM, WN, false, 3>, conv_gemm1x1_kernel<BN, WM, WN, true, 3>};
  static bool attr_set = false;
  if (!attr_set) {
    for (auto kp : kerns) HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kp), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  auto kern = kerns[(na3 ? 2 : 0) + (k.chstat ? 1 : 0)];
  if (k.xcat) {   // [x | xcat] read in place
    void (*kcat[2])(ConvK) = {conv_gemm1x1_kernel<BN, WM, WN, false, 2, true>, conv_gemm1x1_kernel<BN, WM, WN, false, 3, true>};
    static bool cat_attr_set = false;
    if (!cat_attr_set) {
      for (auto kp : kcat) HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kp), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      cat_attr_set = true;
    }
    DDMI_REQUIRE(!k.chstat, "gemm1x1: the dual-source form delivers no channel statistics");
    kern = kcat[na3 ? 1 : 0];
  }
  {
    double flops, bytes;
    conv_work(k, 1, 2, &flops, &bytes);
    static const std::string name = "conv_gemm1x1<bf16,256x" + std::to_string(BN) + ",persistent>";
    static const std::string name_cat = "conv_gemm1x1_cat<bf16,256x" + std::to_string(BN) + ",persistent>";
    PROF_SCOPE(conv_prof_name(k.xcat ? name_cat : name, k).c_str(), flops, bytes, s);
    hipLaunchKernelGGL(kern, grid, dim3(WM * WN * 64), lds, s, k);
  }
  HIP_TRY(hipGetLastError());
  return DDMI_OK;
}

template <int BN, int WM, int WN>
static int launch_halo(ConvK k, int splitk, int batch, hipStream_t s) {
  constexpr int lds = 2 * (BN > 224 ? 344 : HALO_MAX_ROWS) * 128 + 2 * BN * 128;
  static_assert(lds <= 160 * 1024, "halo tile does not fit the 160 KiB LDS");
  DDMI_REQUIRE(batch == 1 || (k.up4 && batch == 4 && splitk == 1), "halo kernel: no batched-GEMM mode");
  DDMI_REQUIRE(BN <= 224 || halo256_shape_ok(k.Hin, k.Win), "halo kernel: the 256-wide tile serves W = 16 / 32 with whole 256-pixel tiles per sample");
  k.tiles_n = cdiv(k.Cout, BN);
  dim3 grid(cdiv(k.M, 256) * k.tiles_n, batch, splitk);
  if (k.up4) {   // four 2x2 parity convolutions on the source grid
    auto kern2 = conv_halo_kernel<BN, WM, WN, false, 2>;
    static bool attr2_set = false;
    if (!attr2_set) {
      HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern2), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
      attr2_set = true;
    }
    double flops, bytes;
    conv_work(k, batch, 2, &flops, &bytes);
    static const std::string name2 = "conv_halo_up4<bf16,256x" + std::to_string(BN) + ">";
    PROF_SCOPE(conv_prof_name(name2, k).c_str(), flops, bytes, s);
    k.ts = nullptr;
    hipLaunchKernelGGL(kern2, grid, dim3(WM * WN * 64), lds, s, k);
    HIP_TRY(hipGetLastError());
    return DDMI_OK;
  }
  auto kern = conv_halo_kernel<BN, WM, WN, false>;
  auto kern_skip = conv_halo_kernel<BN, WM, WN, true>;   // + the folded 1x1 skip connection
  auto kern_half = conv_halo_kernel<BN, WM, WN, false, 3, true>;   // Cin = 64 k + (1..32): two taps per stage in the ragged block
  static bool attr_set = false;
  if (!attr_set) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern_skip), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern_half), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    attr_set = true;
  }
  auto kern_fgn = conv_halo_kernel<BN, WM, WN, false, 3, false, true>;        // + the producer-side GroupNorm epilogue
  auto kern_half_fgn = conv_halo_kernel<BN, WM, WN, false, 3, true, true>;
  static bool attr_fgn_set = false;
  if (!attr_fgn_set) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern_fgn), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern_half_fgn), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    attr_fgn_set = true;
  }
  static const bool half_off = getenv("DDMI_HALO_HALF") && atoi(getenv("DDMI_HALO_HALF")) == 0;   // diagnostics: nine half-empty stages
  const bool half = (g_halo_half < 0 ? !half_off : g_halo_half != 0) && !k.x2 && splitk == 1 && (k.Cin & 63) != 0 && (k.Cin & 63) <= 32;
  {
    double flops, bytes;
    conv_work(k, 1, 2, &flops, &bytes);
    static const std::string name = "conv_halo<bf16,256x" + std::to_string(BN) + ">";
    static const std::string name_skip = "conv_halo+skip<bf16,256x" + std::to_string(BN) + ">";
    static const std::string name_fgn = "conv_halo+gn<bf16,256x" + std::to_string(BN) + ">";
    // (one profiler scope per INSTANTIATION, so that a scope's average launch is one rocprofv3 kernel's: tail2 = the two-taps-per-stage tail)
    static const std::string name_half = "conv_halo<bf16,256x" + std::to_string(BN) + ",tail2>";
    static const std::string name_half_fgn = "conv_halo+gn<bf16,256x" + std::to_string(BN) + ",tail2>";
    if (k.gn_gamma && k.gn_tps > 1) hipLaunchKernelGGL(conv_fgn_epoch_kernel, dim3(1), dim3(1), 0, s, k.gn_sync);   // this launch's granule tag
    PROF_SCOPE(conv_prof_name(k.x2 ? name_skip : k.gn_gamma ? (half ? name_half_fgn : name_fgn) : half ? name_half : name, k).c_str(), flops, bytes, s);
    k.ts = conv_ts_buffer((int)grid.x);
    if (k.x2) hipLaunchKernelGGL(kern_skip, grid, dim3(WM * WN * 64), lds, s, k);
    else if (k.gn_gamma) hipLaunchKernelGGL(half ? kern_half_fgn : kern_fgn, grid, dim3(WM * WN * 64), lds, s, k);
    else if (half) hipLaunchKernelGGL(kern_half, grid, dim3(WM * WN * 64), lds, s, k);
    else hipLaunchKernelGGL(kern, grid, dim3(WM * WN * 64), lds, s, k);
  }
  HIP_TRY(hipGetLastError());
  static int ts_launch = 0;
  if (k.ts && ++ts_launch == 6) conv_dump_timeline(k.ts, (int)grid.x, s);
  return DDMI_OK;
}

static int dispatch(const ConvK& k, TileCfg t, int dt, int splitk, int batch, hipStream_t s) {
  if (dt == DT_BF16) {
    switch (t.id) {
      case 0: return launch_ring<128, 112, 4, 1, 1, 5>(k, splitk, batch, s);
      case 1: return launch_ring<128, 112, 4, 1, 2, 4>(k, splitk, batch, s);
      case 2: return launch_ring<256, 112, 8, 1, 2, 3>(k, splitk, batch, s);
      case 3: return launch_ring<128, 224, 4, 2, 2, 3>(k, splitk, batch, s);
      case 4: return launch_ring<128, 128, 2, 2, 2, 4>(k, splitk, batch, s);
      case 5: return launch_ring<128, 64, 4, 1, 2, 4>(k, splitk, batch, s);
      case 6: return launch_ring<64, 128, 2, 2, 2, 4>(k, splitk, batch, s);
      case 7: return launch_ring<64, 64, 2, 2, 2, 4>(k, splitk, batch, s);
      case 8: return launch_ring<256, 128, 4, 2, 2, 3>(k, splitk, batch, s);
      case 9: return launch_ring<128, 224, 2, 2, 1, 3>(k, splitk, batch, s);
      case 10: return launch_ring<256, 112, 4, 1, 1, 3>(k, splitk, batch, s);
      case 11: return launch_ring<256, 224, 4, 2, 1, 4>(k, splitk, batch, s);
      case 12: return launch_ring<128, 224, 2, 2, 1, 4>(k, splitk, batch, s);
      case 13: return launch_pipe<256, 224, 4, 2>(k, splitk, batch, s);
      case 14: return launch_pipe<128, 224, 2, 2>(k, splitk, batch, s);
      case 15: return launch_pipe<256, 112, 4, 1>(k, splitk, batch, s);
      case 16: return launch_pipe<256, 128, 4, 2>(k, splitk, batch, s);
      case 17: return launch_pipe<128, 128, 2, 2>(k, splitk, batch, s);
      case 18: return launch_pipe<128, 64, 2, 1>(k, splitk, batch, s);
      case 19: return launch_halo<224, 4, 2>(k, splitk, batch, s);
      case 20: return launch_halo<128, 4, 2>(k, splitk, batch, s);
      case 21: return launch_ring<256, 224, 4, 2, 1, 4, true>(k, splitk, batch, s);
      case 22: return launch_ring<128, 224, 2, 2, 1, 4, true>(k, splitk, batch, s);
      case 23: return launch_ring<256, 224, 4, 2, 1, 5, true>(k, splitk, batch, s);
      case 24: return launch_gemm1x1<224, 4, 2>(k, s);
      case 25: return launch_gemm1x1<128, 4, 2>(k, s);
      case 26: return launch_halo<256, 2, 4>(k, splitk, batch, s);
      case 27: return launch_pipe<256, 256, 2, 4>(k, splitk, batch, s);
      case 28: return launch_gemm1x1<256, 2, 4>(k, s);
    }
    return ddmi_fail(DDMI_ERR_INVALID, "conv: no ring configuration");
  }
  if (t.bm == 128 && t.bn == 112) return launch_cfg<float, 128, 112, 4, 1, 1>(k, splitk, batch, s);
  if (t.bm == 128 && t.bn == 128) return launch_cfg<float, 128, 128, 2, 2, 1>(k, splitk, batch, s);
  if (t.bm == 64 && t.bn == 128) return launch_cfg<float, 64, 128, 2, 2, 1>(k, splitk, batch, s);
  if (t.bm == 128 && t.bn == 64) return launch_cfg<float, 128, 64, 4, 1, 1>(k, splitk, batch, s);
  if (t.bm == 64 && t.bn == 64) return launch_cfg<float, 64, 64, 2, 2, 1>(k, splitk, batch, s);
  return ddmi_fail(DDMI_ERR_INVALID, "conv: no tile configuration");
}

static int check_conv(const ConvArgs& a, int dt) {
  int eps = dt == DT_F32 ? 4 : 8;
  DDMI_REQUIRE(a.x && a.w && (a.y || a.y_nchw || a.y32 || (a.force_ws && a.splitk_ws)), "conv: null tensor");
  DDMI_REQUIRE(!(a.y32 || a.res32) || (dt == DT_BF16 && a.splitk <= 1 && a.ups != 3 && !a.res), "conv: fp32 output is the bf16 kernels' one-launch (no split-K) finish");
  DDMI_REQUIRE(a.Cin % eps == 0, "conv: Cin must be a multiple of the 16-byte vector");
  DDMI_REQUIRE(a.Cout % 4 == 0, "conv: Cout must be a multiple of 4");
  DDMI_REQUIRE(!a.emb || a.emb_stride % 4 == 0, "conv: emb_stride must be a multiple of 4");
  DDMI_REQUIRE(a.stride == 1 || a.stride == 2, "conv: stride must be 1 or 2");
  DDMI_REQUIRE(!(a.ups && a.stride != 1), "conv: fused upsample needs stride 1");
  if (a.ups == 3) {   // parity form of upsample + 3x3: weights [4][Cout][2*2*Cin] from launch_pack_conv_up4
    DDMI_REQUIRE(dt == DT_BF16 && a.KH == 2 && a.KW == 2 && a.pad == 1, "conv: parity upsample is a bf16 2x2 form");
    DDMI_REQUIRE(a.Hout == 2 * a.Hin && a.Wout == 2 * a.Win, "conv: output size mismatch");
    DDMI_REQUIRE(!a.emb && !a.res && !a.y_nchw && a.splitk <= 1, "conv: parity upsample takes bias only, no split-K");
    return DDMI_OK;
  }
  DDMI_REQUIRE((a.KH == 1 && a.KW == 1) || (a.KH == 3 && a.KW == 3), "conv: kernel must be 1x1 or 3x3");
  if (a.x2) {
    DDMI_REQUIRE(dt == DT_BF16 && halo_ok(a), "conv: a folded skip connection needs a bf16 3x3 / stride 1 shape of the halo kernel");
    DDMI_REQUIRE(a.w2 && a.C2 > 0 && a.C2 % 8 == 0 && a.C3 >= 0 && a.C3 % 8 == 0 && (a.C3 == 0 || a.x3), "conv: bad skip sources");
    DDMI_REQUIRE(!a.y_nchw && !a.res, "conv: a folded skip connection replaces the residual");
  }
  int hin = a.ups == 1 ? 2 * a.Hin : a.Hin, win = a.ups ? 2 * a.Win : a.Win;
  DDMI_REQUIRE(a.Hout == (hin + 2 * a.pad - a.KH) / a.stride + 1 && a.Wout == (win + 2 * a.pad - a.KW) / a.stride + 1,
               "conv: output size mismatch");
  return DDMI_OK;
}

// wave tile (rows, channels) of the bf16 configurations kRing[id]
static const int kRingWave[][2] = {{32, 112}, {32, 112}, {32, 112}, {32, 112}, {64, 64}, {32, 64}, {32, 64}, {32, 32}, {64, 64}, {64, 112}, {64, 112},
                                   {64, 112}, {64, 112}, {64, 112}, {64, 112}, {64, 112}, {64, 64}, {64, 64}, {64, 64}, {64, 112}, {64, 64},
                                   {64, 112}, {64, 112}, {64, 112}, {64, 112}, {64, 64}, {64, 64}, {64, 64}, {64, 64}};   // (26-28: 128 x 64 as two 64-row halves)
static_assert(sizeof(kRingWave) / sizeof(kRingWave[0]) == kNumRing, "one wave tile per ring configuration");

// Rows per statistics block (the wave tile height) when a launch of `a` with a.chstat set would fill the WHOLE
// [M / rows][Cout][2] buffer - every wave on conv_epilogue_fast - else 0 (the caller then runs the statistics pass).
int conv_chstat_rows(const ConvArgs& a, int dt) {
  static const bool off = getenv("DDMI_GN_CHSTAT") && atoi(getenv("DDMI_GN_CHSTAT")) == 0;
  static const int dbg = getenv("DDMI_CONV_DBG") ? atoi(getenv("DDMI_CONV_DBG")) : 0;
  if (off || dbg || dt != DT_BF16 || a.splitk > 1 || a.y_nchw || !a.y || a.xcat) return 0;
  const int M = conv_rows(a);
  TileCfg t = pick_tile(M, a.Cout, dt, a.cfg);
  if (t.kps == -2 && !halo_ok(a)) t = pick_tile(M, a.Cout, dt, kHaloFallback + 1);
  if (t.kps == -2 && t.bn == 256 && (a.ups == 3 || a.x2 || !halo256_shape_ok(a.Hin, a.Win))) t = pick_tile(M, a.Cout, dt, 20 + 1);
  if (a.x2 && t.kps != -2) t = pick_tile(M, a.Cout, dt, skip_fold_cfg(a) + 1);
  if (t.kps == -3 && !gemm1x1_ok(a, t.bn)) t = pick_tile(M, a.Cout, dt, kGemmFallback + 1);
  if (t.id < 0) return 0;
  const bool paired = t.kps != -1 || a.order > 2 || (a.order == 0 && a.KH * a.KW == 1);   // ring and halo kernels always are
  const int TM = kRingWave[t.id][0], TN = kRingWave[t.id][1];
  const int hw = a.ups == 3 ? a.Hin * a.Win : a.Hout * a.Wout;   // parity-form upsample: the kernel's rows are source pixels
  if (!paired || a.Cout % TN || hw % TM || M % TM) return 0;
  return TM;
}

// Producer-side GroupNorm: the launch must be a plain bf16 3x3 / stride 1 convolution on the halo kernel with no split-K, whole
// BN-wide channel tiles that hold whole groups (BN % (Cout / 32) == 0), 256-row tiles inside one sample (H W % 256 == 0), at most
// 16 tiles per sample and one polling lane per (tile, group) granule, granule runs that fill whole 128-byte lines.
bool conv_fgn_ok(const ConvArgs& a, int dt) {
  static const bool off = getenv("DDMI_GN_FUSE") && atoi(getenv("DDMI_GN_FUSE")) == 0;
  static const int dbg = getenv("DDMI_CONV_DBG") ? atoi(getenv("DDMI_CONV_DBG")) : 0;
  if (off || dbg || dt != DT_BF16 || !a.gn_gamma || !a.gn_beta || !a.gn_part || !a.gn_sync) return false;
  if (a.splitk > 1 || a.force_ws || a.res || a.res32 || a.y32 || a.y_nchw || !a.y || a.x2 || a.ups || a.out_scale != 1.f) return false;
  if (!(a.KH == 3 && a.KW == 3 && a.stride == 1 && a.pad == 1) || !halo_ok(a)) return false;
  const int M = conv_rows(a), hw = a.Hout * a.Wout;
  TileCfg t = pick_tile(M, a.Cout, dt, a.cfg);
  if (t.kps != -2) return false;
  if (t.bn == 256 && !halo256_shape_ok(a.Hin, a.Win)) t = pick_tile(M, a.Cout, dt, 20 + 1);
  const int BN = t.bn, cpg = a.Cout / 32, tps = hw / 256;
  if (a.Cout % 32 || a.Cout % BN || BN % cpg || hw % 256 || tps > 16) return false;
  const int NG = BN / cpg, tiles_n = a.Cout / BN;
  if (tps > 1 && (a.gn_no_handoff || tps * NG > 512 || NG % 8 != 0)) return false;
  // progress (conv_epilogue_fgn): the tiles that wait for each other - tps m-tiles of one n-tile, tps * tiles_n consecutive ids in the
  // m-major order the launch is pinned to - must all fit the 32 CUs of an XCD with room to spare.  (The D2C-VAE decoder's 512-channel
  // convolutions at 64x64 - 16 x 4 = 64 - do not: every launch sat in its bounded poll until the timeout; found on the first bench run.)
  if (tps > 1 && tps * tiles_n > 16) return false;
  // ... and each cohort inside ONE XCD's contiguous tile range: with B % 8 != 0 the ranges cut through samples, the first workgroups of
  // an XCD then hold their CUs for the whole launch waiting for the last ones of its neighbour (measured on the AFHQ UNet: B = 250
  // 65.9 -> 72.4 ms per evaluation with the fusion on, B = 100 31.6 -> 32.2; no timeout, just slower) - a speed rule, not a correctness one
  if (tps > 1 && a.B % 8 != 0 && !a.gn_any_batch) return false;
  if (a.emb && a.emb_stride % 4) return false;
  return true;
}

int launch_conv(const ConvArgs& a, int dt, hipStream_t s) {
  DDMI_TRY(check_conv(a, dt));
  DDMI_REQUIRE(!a.gn_gamma || conv_fgn_ok(a, dt), "conv: a fused GroupNorm asked of a launch that cannot provide it");
  int splitk = std::max(1, a.splitk);
  DDMI_REQUIRE((splitk == 1 && !a.force_ws) || a.splitk_ws, "conv: split-K needs a workspace");
  int M = conv_rows(a);
  TileCfg t = pick_tile(M, a.Cout, dt, a.cfg);
  if (t.kps == -2 && !halo_ok(a)) t = pick_tile(M, a.Cout, dt, kHaloFallback + 1);
  if (t.kps == -2 && t.bn == 256 && (a.ups == 3 || a.x2 || !halo256_shape_ok(a.Hin, a.Win))) t = pick_tile(M, a.Cout, dt, 20 + 1);
  if (a.x2 && t.kps != -2) t = pick_tile(M, a.Cout, dt, skip_fold_cfg(a) + 1);
  if (t.kps == -3 && !gemm1x1_ok(a, t.bn)) t = pick_tile(M, a.Cout, dt, kGemmFallback + 1);
  if (a.xcat) {   // the dual-source form exists on the persistent 1x1 kernel only
    DDMI_REQUIRE(conv_cat_ok(a, dt), "conv: a channel-concatenated input needs a bf16 1x1 shape of the persistent kernel");
    if (t.kps != -3 || a.Cout % t.bn) t = pick_tile(M, a.Cout, dt, (a.Cout % 224 == 0 ? 24 : 25) + 1);
  }
  ConvK k;
  DDMI_TRY(make_k(a, dt, t.kps, splitk, &k));
  // order: 1 tap-major, 2 channel-major, +2 = paired-fragment instantiation of the pipe kernel; 0 = heuristic
  const int walk = a.order > 2 ? a.order - 2 : a.order;
  k.chan_major = walk == 2 || (a.order == 0 && a.KH * a.KW > 1 && splitk == 1);
  k.paired = a.order > 2 || (a.order == 0 && a.KH * a.KW == 1);
  {
    static const int env_map = getenv("DDMI_CONV_TILEMAP") ? atoi(getenv("DDMI_CONV_TILEMAP")) : 0;  // diagnostics
    int tm = env_map ? env_map : a.tile_map;
    if (tm == 0) {  // heuristic: keep the LARGER operand's reuse inside one XCD
      const double act = (double)a.B * a.Hin * a.Win * a.Cin, wgt = (double)a.Cout * a.KH * a.KW * a.Cin;
      tm = wgt > act ? 2 : 1;
    }
    k.tile_map = tm < 0 ? 0 : tm;
    if (a.gn_gamma) k.tile_map = 1;   // producer-side GroupNorm: the tiles of a sample on consecutive ids of one XCD (conv_fgn_ok)
  }
  if (a.chstat) {
    DDMI_REQUIRE(conv_chstat_rows(a, dt) > 0, "conv: channel statistics asked of a launch that cannot provide them");
    k.chstat = a.chstat;
  }
  DDMI_TRY(dispatch(k, t, dt, splitk, a.ups == 3 ? 4 : 1, s));
  if (splitk > 1 && !a.force_ws) DDMI_TRY(launch_splitk_finish(a, dt, s));   // force_ws: the caller finishes (in another dtype)
  return DDMI_OK;
}

int launch_splitk_finish(const ConvArgs& a, int dt, hipStream_t s) {
  ConvK k;
  DDMI_TRY(make_k(a, dt, 1, 2, &k));
  k.SPS = a.splitk;  // number of slabs to sum
  int64_t nvec = (int64_t)k.M * k.Cout / 4;
  int blocks = (int)std::min<int64_t>(cdiv64(nvec, 256), 256 * 8);
  PROF_SCOPE("splitk_finish", 0, (double)(a.splitk * 4 + dt_size(dt)) * k.M * k.Cout, s);
  if (dt == DT_BF16)
    hipLaunchKernelGGL(splitk_finish_kernel<bf16_t>, dim3(blocks), dim3(256), 0, s, k);
  else
    hipLaunchKernelGGL(splitk_finish_kernel<float>, dim3(blocks), dim3(256), 0, s, k);
  HIP_TRY(hipGetLastError());
  return DDMI_OK;
}

int launch_gemm_nt(const void* A, const void* Bm, void* C, int dt, int batch, int M, int N, int K, int64_t strideA,
                   int64_t strideB, int64_t strideC, float alpha, hipStream_t s) {
  ConvArgs a{};
  a.x = A; a.w = Bm; a.y = C;
  a.B = 1; a.Hin = 1; a.Win = M; a.Cin = K; a.Hout = 1; a.Wout = M; a.Cout = N;
  a.KH = a.KW = 1; a.stride = 1; a.pad = 0; a.ups = 0; a.out_scale = alpha; a.splitk = 1;
  DDMI_TRY(check_conv(a, dt));
  TileCfg t = pick_tile(M * batch, N, dt);
  ConvK k;
  DDMI_TRY(make_k(a, dt, t.kps, 1, &k));
  k.x_bstride = strideA; k.w_bstride = strideB; k.y_bstride = strideC;
  return dispatch(k, t, dt, 1, batch, s);
}

// ------------------------------------------------------------------------------------------------------
// autotune: the best (tile configuration, K order, split-K) differs per shape (SURVEY.md Appendix A has 46 distinct
// conv shapes in the AFHQ UNet alone) - time the candidates once at finalize on the real buffers.
// ------------------------------------------------------------------------------------------------------
int conv_autotune(const ConvArgs& a0, int dt, float* ws, size_t ws_bytes, ConvPlan* best) {
  *best = ConvPlan{};
  if (dt != DT_BF16) return DDMI_OK;
  DDMI_TRY(check_conv(a0, dt));
  const int M = conv_rows(a0);
  hipEvent_t e0, e1;
  HIP_TRY(hipEventCreate(&e0));
  HIP_TRY(hipEventCreate(&e1));
  float best_us = 1e30f;
  static const int kSplits[] = {1, 2, 3, 4, 6, 8, 12, 16};
  for (int id = 0; id < kNumRing; ++id) {
    const RingCfg& rc = kRing[id];
    if (rc.kps == -2 && !halo_ok(a0)) continue;
    if (rc.kps == -2 && rc.bn == 256 && (a0.ups == 3 || a0.x2 || !halo256_shape_ok(a0.Hin, a0.Win))) continue;
    if (a0.x2 && rc.kps != -2) continue;   // a folded skip connection runs on the halo kernel only
    if (a0.xcat && rc.kps != -3) continue;  // a channel-concatenated input on the persistent 1x1 kernel only
    if (rc.kps == -3) {
      static const bool no_gemm = getenv("DDMI_NO_GEMM1X1") != nullptr;   // diagnostics: A/B against the pipe / ring kernels
      if (no_gemm) continue;
      ConvArgs probe = a0;
      probe.splitk = 1;
      if (!gemm1x1_ok(probe, rc.bn)) continue;
    }
    // prune: tile width must fit Cout without more than ~15 % padding; tile height not larger than 2x the problem
    int padded = cdiv(a0.Cout, rc.bn) * rc.bn;
    if (padded * 100 > a0.Cout * 115 && !(a0.Cout <= 64 && rc.bn == 64)) continue;
    if (rc.bm > 2 * M && rc.bm > 64) continue;
    int tiles = cdiv(M, rc.bm) * cdiv(a0.Cout, rc.bn);
    int NS = rc.kps < 0 ? cdiv(a0.Cin, 64) * a0.KH * a0.KW : cdiv(cdiv(a0.Cin, 32) * a0.KH * a0.KW, rc.kps);
    for (int sk : kSplits) {
      if (sk > 1 && (a0.ups == 3 || tiles * (sk - 1) >= 512 || NS / sk < 4)) continue;
      if (rc.kps == -2 && sk > cdiv(a0.Cin, 64)) continue;
      if (rc.kps == -3 && sk > 1) continue;
      if ((size_t)sk * M * a0.Cout * sizeof(float) > ws_bytes && sk > 1) continue;
      const int n_walk = (a0.KH * a0.KW > 1 && rc.kps != -2) ? 2 : 1;
      for (int order = 1; order <= (rc.kps == -1 ? n_walk + 2 : n_walk); ++order) {  // pipe: + the paired instantiation
        if (order > n_walk && order <= 2) continue;
        ConvArgs a = a0;
        a.cfg = id + 1; a.order = order; a.splitk = sk; a.splitk_ws = ws;
        // a0.chstat set = the consumer GroupNorm wants this convolution's channel statistics: a candidate that can deliver them is
        // timed WITH that epilogue work, one that cannot is charged the statistics pass it forces (a read of the output at ~5 TB/s
        // + a launch) - round 5: the tuner had picked the unpaired 256-wide pipe tile for sky's convolutions and `gn_stats` doubled
        float penalty_us = 0.f;
        if (a0.chstat) {
          if (conv_chstat_rows(a, dt) == 0) {
            a.chstat = nullptr;
            penalty_us = 5.f + (float)((double)M * a0.Cout * 2.0 / 5.0e6);
          }
        }
        int rc2 = DDMI_OK;
        for (int i = 0; i < 2 && rc2 == DDMI_OK; ++i) rc2 = launch_conv(a, dt, 0);
        (void)hipEventRecord(e0, 0);
        for (int i = 0; i < 4 && rc2 == DDMI_OK; ++i) rc2 = launch_conv(a, dt, 0);
        (void)hipEventRecord(e1, 0);
        if (hipEventSynchronize(e1) != hipSuccess || rc2 != DDMI_OK) {
          (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
          return rc2 != DDMI_OK ? rc2 : ddmi_fail(DDMI_ERR_HIP, "conv_autotune: launch failed");
        }
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        float us = ms * 250.f + penalty_us;
        if (us < best_us) {
          best_us = us;
          *best = ConvPlan{id + 1, order, sk, us};
        }
      }
    }
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return DDMI_OK;
}
