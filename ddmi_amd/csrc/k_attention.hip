// Fused self-attention for the LDM UNet's AttentionBlock (openaimodel.py:279-325 + QKVAttentionLegacy :397-422),
// head dim 32, token-major activations.
//
//   qkv [B, T, H*96]  (per head: 32 q | 32 k | 32 v channels - the legacy "split heads before qkv" interleave)
//   out [B, T, H*32]   out[b, t, h*32 + c] = sum_s softmax_s(q_t . k_s / sqrt(32)) v_s[c]
//
// One workgroup = 4 waves = 64 queries of one (b, head); keys/values stream through LDS in tiles of 64 with an
// online softmax.  Everything is arranged so the QUERY sits on the MFMA lane (lane & 15):
//   S^T = K . Q^T     A-operand = K tile rows (keys) from LDS, B-operand = Q rows (registers, loaded once)
//                     -> lane holds scores of its query for keys 16f + 4*(lane>>4) + j  (row-max / row-sum need only
//                        in-register reductions plus two xor-shuffles over lane>>4)
//   O^T = V^T . P^T   A-operand = V^T rows (channels) from a transposed LDS image, B-operand = P straight from the
//                     score registers (key order inside an MFMA k-step is permuted identically on both operands)
//                     -> lane holds out channels 16*df + 4*(lane>>4) + j of its query: 8/16-byte vector stores.
// bf16 path: v_mfma_f32_16x16x32_bf16, fp32 softmax/accumulate.  fp32 (parity) path: v_mfma_f32_16x16x4_f32.
#include "kernels.h"
#include "prof.h"
#include <math.h>

#define ATT_D 32
#define ATT_KT 64   // keys per LDS tile
#define ATT_QB 64   // queries per workgroup

template <typename T> struct AttL;
template <> struct AttL<bf16_t> {
  static constexpr int KROWB = 64;              // K tile row bytes (32 bf16), XOR swizzled like the conv tiles
  static constexpr int VROWB = (ATT_KT + 4) * 2;  // V^T row bytes (64 keys + 4 pad)
};
template <> struct AttL<float> {
  static constexpr int KROWB = 144;
  static constexpr int VROWB = (ATT_KT + 4) * 4;
};

__device__ __forceinline__ int att_swz(int row, int slot) {
  int r = (row >> 2) & 3;
  return slot ^ ((0x78 >> (r * 2)) & 3);
}

template <typename T>
__global__ void __launch_bounds__(256) attention_legacy_kernel(const T* __restrict__ qkv, T* __restrict__ out, int T_len,
                                                               int heads, float scale) {
  constexpr bool IS_BF16 = sizeof(T) == 2;
  constexpr int KROWB = AttL<T>::KROWB, VROWB = AttL<T>::VROWB;
  __shared__ __attribute__((aligned(16))) unsigned char ks[ATT_KT * KROWB];
  __shared__ __attribute__((aligned(16))) unsigned char vs[ATT_D * VROWB];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lg = lane >> 4;
  const int bh = blockIdx.y, b = bh / heads, h = bh - b * heads;
  const int C3 = heads * 3 * ATT_D;
  const T* base = qkv + (int64_t)b * T_len * C3 + h * 3 * ATT_D;
  const int q_idx = blockIdx.x * ATT_QB + wave * 16 + lr;
  const bool q_ok = q_idx < T_len;

  // ---- Q fragment(s) in registers --------------------------------------------------------------
  bf16x8 qb;
  f32x4 qf[2];
  if constexpr (IS_BF16) {
    qb = q_ok ? *reinterpret_cast<const bf16x8*>(base + (int64_t)q_idx * C3 + 8 * lg) : bf16x8{};
  } else {
#pragma unroll
    for (int hh = 0; hh < 2; ++hh)
      qf[hh] = q_ok ? *reinterpret_cast<const f32x4*>(base + (int64_t)q_idx * C3 + 16 * hh + 4 * lg) : f32x4{0, 0, 0, 0};
  }

  float m_run = -INFINITY, l_run = 0.f;
  f32x4 o[2] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};

  const int ntiles = (T_len + ATT_KT - 1) / ATT_KT;
  for (int kt = 0; kt < ntiles; ++kt) {
    __syncthreads();  // previous tile fully consumed
    // ---- stage K tile [key][32] and V^T tile [32][key] ---------------------------------------------
    if constexpr (IS_BF16) {
      int key = tid >> 2, slot = tid & 3;  // 64 keys x 4 slots of 8 channels
      int kg = kt * ATT_KT + key;
      bf16x8 kv = bf16x8{}, vv = bf16x8{};
      if (kg < T_len) {
        const T* p = base + (int64_t)kg * C3;
        kv = *reinterpret_cast<const bf16x8*>(p + ATT_D + 8 * slot);
        vv = *reinterpret_cast<const bf16x8*>(p + 2 * ATT_D + 8 * slot);
      }
      *reinterpret_cast<bf16x8*>(ks + key * KROWB + att_swz(key, slot) * 16) = kv;
#pragma unroll
      for (int e = 0; e < 8; ++e) *reinterpret_cast<bf16_t*>(vs + (8 * slot + e) * VROWB + key * 2) = vv[e];
    } else {
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        int idx = tid + it * 256;
        int key = idx >> 3, slot = idx & 7;  // 64 keys x 8 slots of 4 channels
        int kg = kt * ATT_KT + key;
        f32x4 kv = f32x4{0, 0, 0, 0}, vv = f32x4{0, 0, 0, 0};
        if (kg < T_len) {
          const T* p = base + (int64_t)kg * C3;
          kv = *reinterpret_cast<const f32x4*>(p + ATT_D + 4 * slot);
          vv = *reinterpret_cast<const f32x4*>(p + 2 * ATT_D + 4 * slot);
        }
        *reinterpret_cast<f32x4*>(ks + key * KROWB + slot * 16) = kv;
#pragma unroll
        for (int e = 0; e < 4; ++e) *reinterpret_cast<float*>(vs + (4 * slot + e) * VROWB + key * 4) = vv[e];
      }
    }
    __syncthreads();

    // ---- S^T = K . Q^T : 4 fragments of 16 keys ----------------------------------------------------
    f32x4 sc[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      sc[f] = f32x4{0, 0, 0, 0};
      int row = 16 * f + lr;
      if constexpr (IS_BF16) {
        bf16x8 kf = *reinterpret_cast<const bf16x8*>(ks + row * KROWB + att_swz(row, lg) * 16);
        sc[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qb, sc[f], 0, 0, 0);
      } else {
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          f32x4 kf = *reinterpret_cast<const f32x4*>(ks + row * KROWB + (4 * hh + lg) * 16);
#pragma unroll
          for (int e = 0; e < 4; ++e) sc[f] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[e], qf[hh][e], sc[f], 0, 0, 0);
        }
      }
    }
    // ---- online softmax for this lane's query ------------------------------------------------------
    float tmax = -INFINITY;
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        int kg = kt * ATT_KT + 16 * f + 4 * lg + j;
        float v = kg < T_len ? sc[f][j] * (IS_BF16 ? scale * 1.44269504088896341f : scale) : -INFINITY;
        sc[f][j] = v;
        tmax = fmaxf(tmax, v);
      }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 16));
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
    float m_new = fmaxf(m_run, tmax);
    float alpha = IS_BF16 ? __builtin_amdgcn_exp2f(m_run - m_new) : expf(m_run - m_new);
    float psum = 0.f;
#pragma unroll
    for (int f = 0; f < 4; ++f)
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
    o[0] *= alpha;
    o[1] *= alpha;

    // ---- O^T += V^T . P^T ---------------------------------------------------------------------------
    if constexpr (IS_BF16) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {  // k-step u covers keys 32u .. 32u+31 (fragments 2u, 2u+1)
        bf16x8 pf;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          pf[j] = (bf16_t)sc[2 * u][j];
          pf[4 + j] = (bf16_t)sc[2 * u + 1][j];
        }
#pragma unroll
        for (int df = 0; df < 2; ++df) {
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
      for (int f = 0; f < 4; ++f)
#pragma unroll
        for (int df = 0; df < 2; ++df) {
          f32x4 vf = *reinterpret_cast<const f32x4*>(vs + (16 * df + lr) * VROWB + (16 * f + 4 * lg) * 4);
#pragma unroll
          for (int j = 0; j < 4; ++j) o[df] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[j], sc[f][j], o[df], 0, 0, 0);
        }
    }
  }

  if (q_ok) {
    float inv = 1.0f / l_run;
    T* op = out + ((int64_t)b * T_len + q_idx) * (heads * ATT_D) + h * ATT_D;
#pragma unroll
    for (int df = 0; df < 2; ++df) {
      f32x4 v = o[df] * inv;
      if constexpr (IS_BF16) {
        bf16x4 t;
#pragma unroll
        for (int j = 0; j < 4; ++j) t[j] = (bf16_t)v[j];
        *reinterpret_cast<bf16x4*>(op + 16 * df + 4 * lg) = t;
      } else {
        *reinterpret_cast<f32x4*>(op + 16 * df + 4 * lg) = v;
      }
    }
  }
}

int launch_attention_legacy(const void* qkv, void* out, int dt, int B, int T, int heads, hipStream_t s) {
  DDMI_REQUIRE(B > 0 && T > 0 && heads > 0, "attention: bad shape");
  dim3 grid(cdiv(T, ATT_QB), B * heads);
  float scale = 1.0f / sqrtf((float)ATT_D);
  PROF_SCOPE("attention_legacy", 4.0 * B * heads * (double)T * T * ATT_D, 4.0 * B * heads * T * ATT_D * dt_size(dt), s);
  if (dt == DT_BF16)
    hipLaunchKernelGGL(attention_legacy_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)qkv, (bf16_t*)out, T, heads, scale);
  else
    hipLaunchKernelGGL(attention_legacy_kernel<float>, grid, dim3(256), 0, s, (const float*)qkv, (float*)out, T, heads, scale);
  HIP_TRY(hipGetLastError());
  return DDMI_OK;
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
