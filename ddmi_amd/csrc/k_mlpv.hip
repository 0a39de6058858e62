// Fused tri-plane HDPE sampling + video MLP (replaces models/d2c_vae/mlp.py:128-157 `MLPVideo.forward` at inference):
//   triplane_positional_encoding(mode='concat') at three scales (utils/general_utils.py:134-145: three 2-D bilinear
//   grid_samples, align_corners=True, border; the xy sample is broadcast over t, the yt sample over w, the xt sample
//   over h; channel-concatenated -> 192 features per scale, rows ordered (b,t,h,w)),
//   four ResnetBlockFC (blocks.py:673-716) + Linear(leaky_relu(x, 0.2)).
//
// Same shape as the occupancy kernel (k_mlp3d.hip): a tile of P points per workgroup, wave w owns output channels
// [64w, 64w+64) of every 256-wide layer (48 of the 192-wide hidden layer of block 1).  0.857 MMAC per point.
#include "kernels.h"
#include "prof.h"
#include "mlp_common.h"
#include "mlp_fc2.h"

// packed weight matrices (row-major [N][K], storage dtype), in execution order
//  0 r1.fc_0 [192][192]  1 r1.fc_1 [256][192]  2 r1.shortcut [256][192]
//  3 r2.fc_0 [256][448]  4 r2.fc_1 [256][256]  5 r2.shortcut [256][448]
//  6 r3.fc_0 [256][448]  7 r3.fc_1 [256][256]  8 r3.shortcut [256][448]
//  9 r4.fc_0 [256][256] 10 r4.fc_1 [256][256] 11 net_out [16 (3 used)][256]
__host__ __device__ constexpr int mv_rows(int i) { return i == 0 ? 192 : i == 11 ? 16 : 256; }
__host__ __device__ constexpr int mv_k(int i) { return i <= 2 ? 192 : (i == 3 || i == 5 || i == 6 || i == 8) ? 448 : 256; }
__host__ __device__ constexpr int mv_off(int i) {
  int o = 0;
  for (int j = 0; j < i; ++j) o += mv_rows(j) * mv_k(j);
  return o;
}
size_t mlpv_weight_elems() { return (size_t)mv_off(12); }
// fp32 constants: biases r1.fc_0[192] | r1.fc_1 | r2.fc_0 | r2.fc_1 | r3.fc_0 | r3.fc_1 | r4.fc_0 | r4.fc_1 (256 each) | net_out[4]
#define MV_B_R1FC0 0
#define MV_B(i) (192 + (i) * 256)
#define MV_BOUT (192 + 7 * 256)
size_t mlpv_fb_elems() { return MV_BOUT + 4; }

template <typename T> struct MVL;
template <> struct MVL<bf16_t> { static constexpr int P = 64, XP = 272, FP = 200; };
template <> struct MVL<float> { static constexpr int P = 32, XP = 260, FP = 196; };

struct MlpVK {
  const void* planes[9];  // [type j = xy, yt, xt][scale i]: index j*3 + i, NHWC [B][Hp][Wp][64]
  int pw[3];              // P of each scale: xy planes [P][P], yt / xt planes [frames][P]
  int frames;
  const float *cxy, *cyt, *cxt;  // [2][H][W], [2][T][H], [2][T][W]
  const void* w;
  const float* fb;
  float* out;             // [B][3][T][H][W]
  int B, T, H, W;
  int64_t total;          // B*T*H*W
};

// features of scale `sc`: [xy(h,w) | yt(t,h) | xt(t,w)] bilinear samples -> FR (raw) and FA (rectified)
template <typename T>
__device__ __forceinline__ void mv_gather(const MlpVK& a, int sc, int64_t tile0, T* FR, T* FA) {
  constexpr int P = MVL<T>::P, FP = MVL<T>::FP;
  constexpr int EPS = sizeof(T) == 2 ? 8 : 4;
  constexpr int CGP = 64 / EPS, CG = 3 * CGP;  // channel groups per plane / per point
  const int HW = a.H * a.W, THW = a.T * HW;
  for (int it = threadIdx.x; it < P * CG; it += 256) {
    int pi = it / CG, cg = it - pi * CG;
    int j = cg / CGP, cgi = cg - j * CGP;
    int64_t gp = tile0 + pi;
    float v[EPS];
#pragma unroll
    for (int e = 0; e < EPS; ++e) v[e] = 0.f;
    if (gp < a.total) {
      int b = (int)(gp / THW);
      int idx = (int)(gp - (int64_t)b * THW);
      int t = idx / HW, r = idx - t * HW, h = r / a.W, w = r - h * a.W;
      // grid channel 0 -> grid_sample's x (plane width), channel 1 -> y (plane height)   (general_utils.py:27-52, 134-138)
      float gx, gy;
      int Hp, Wp = a.pw[sc];
      if (j == 0) { gx = a.cxy[h * a.W + w]; gy = a.cxy[HW + h * a.W + w]; Hp = Wp; }
      else if (j == 1) { gx = a.cyt[t * a.H + h]; gy = a.cyt[a.T * a.H + t * a.H + h]; Hp = a.frames; }
      else { gx = a.cxt[t * a.W + w]; gy = a.cxt[a.T * a.W + t * a.W + w]; Hp = a.frames; }
      float ix = fminf(fmaxf((gx + 1.0f) * 0.5f * (float)(Wp - 1), 0.f), (float)(Wp - 1));
      float iy = fminf(fmaxf((gy + 1.0f) * 0.5f * (float)(Hp - 1), 0.f), (float)(Hp - 1));
      float x0f = floorf(ix), y0f = floorf(iy);
      float wx1 = ix - x0f, wy1 = iy - y0f, wx0 = 1.f - wx1, wy0 = 1.f - wy1;
      int x0 = (int)x0f, y0 = (int)y0f;
      int x1 = min(x0 + 1, Wp - 1), y1 = min(y0 + 1, Hp - 1);
      const T* base = (const T*)a.planes[j * 3 + sc] + (int64_t)b * Hp * Wp * 64 + cgi * EPS;
      const T* p00 = base + ((int64_t)y0 * Wp + x0) * 64;
      const T* p01 = base + ((int64_t)y0 * Wp + x1) * 64;
      const T* p10 = base + ((int64_t)y1 * Wp + x0) * 64;
      const T* p11 = base + ((int64_t)y1 * Wp + x1) * 64;
      float w00 = wy0 * wx0, w01 = wy0 * wx1, w10 = wy1 * wx0, w11 = wy1 * wx1;
      if constexpr (sizeof(T) == 2) {
        bf16x8 t00 = *reinterpret_cast<const bf16x8*>(p00), t01 = *reinterpret_cast<const bf16x8*>(p01);
        bf16x8 t10 = *reinterpret_cast<const bf16x8*>(p10), t11 = *reinterpret_cast<const bf16x8*>(p11);
#pragma unroll
        for (int e = 0; e < 8; ++e)
          v[e] = (float)t00[e] * w00 + (float)t01[e] * w01 + (float)t10[e] * w10 + (float)t11[e] * w11;
      } else {
        f32x4 t00 = *reinterpret_cast<const f32x4*>(p00), t01 = *reinterpret_cast<const f32x4*>(p01);
        f32x4 t10 = *reinterpret_cast<const f32x4*>(p10), t11 = *reinterpret_cast<const f32x4*>(p11);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = t00[e] * w00 + t01[e] * w01 + t10[e] * w10 + t11[e] * w11;
      }
    }
    float r[EPS];
#pragma unroll
    for (int e = 0; e < EPS; ++e) r[e] = fmaxf(v[e], 0.f);
    if constexpr (sizeof(T) == 2) {
      bf16x8 o, q;
#pragma unroll
      for (int e = 0; e < 8; ++e) { o[e] = (bf16_t)v[e]; q[e] = (bf16_t)r[e]; }
      *reinterpret_cast<bf16x8*>(FR + pi * FP + cg * EPS) = o;
      *reinterpret_cast<bf16x8*>(FA + pi * FP + cg * EPS) = q;
    } else {
      *reinterpret_cast<f32x4*>(FR + pi * FP + cg * EPS) = f32x4{v[0], v[1], v[2], v[3]};
      *reinterpret_cast<f32x4*>(FA + pi * FP + cg * EPS) = f32x4{r[0], r[1], r[2], r[3]};
    }
  }
}

template <typename T>
__global__ void __launch_bounds__(256, 1) mlpv_kernel(MlpVK a) {
  constexpr int P = MVL<T>::P, XP = MVL<T>::XP, FP = MVL<T>::FP;
  constexpr int MF = P / 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  T* XR = (T*)smem;          // block output, raw        [P][XP]
  T* XA = XR + P * XP;       // block output, rectified (leaky-rectified after the last block)
  T* NET = XA + P * XP;      // hidden activation, rectified
  T* FR = NET + P * XP;      // features of the current scale, raw   [P][FP]
  T* FA = FR + P * FP;       // ... rectified

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lr = lane & 15, lg = lane >> 4;
  const int nbase = 64 * wave;
  const int64_t tile0 = (int64_t)blockIdx.x * P;
  const T* W = (const T*)a.w;
  const float* fb = a.fb;

  f32x4 acc[MF][4], xs[MF][4];
  auto zero = [&](f32x4(&r)[MF][4]) {
#pragma unroll
    for (int i = 0; i < MF; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) r[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  };
  auto store_hidden = [&](const float* bias) {  // NET <- relu(acc + bias)
#pragma unroll
    for (int mf = 0; mf < MF; ++mf)
#pragma unroll
      for (int nf = 0; nf < 4; ++nf) {
        int n = nbase + 16 * nf + 4 * lg;
        f32x4 b4 = *reinterpret_cast<const f32x4*>(bias + n);
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = fmaxf(acc[mf][nf][e] + b4[e], 0.f);
        st4<T>(NET + (16 * mf + lr) * XP + n, o);
      }
  };
  // x = xs + acc + bias -> XR, XA ; xs <- x.  `slope`: 0 = relu for the next block, 0.2 = leaky_relu for net_out
  auto store_block = [&](const float* bias, float slope) {
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) {
      int row = 16 * mf + lr;
#pragma unroll
      for (int nf = 0; nf < 4; ++nf) {
        int n = nbase + 16 * nf + 4 * lg;
        f32x4 b4 = *reinterpret_cast<const f32x4*>(bias + n);
        float o[4], r[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float v = xs[mf][nf][e] + acc[mf][nf][e] + b4[e];
          o[e] = v;
          r[e] = v > 0.f ? v : v * slope;
        }
        xs[mf][nf] = f32x4{o[0], o[1], o[2], o[3]};
        st4<T>(XR + row * XP + n, o);
        st4<T>(XA + row * XP + n, r);
      }
    }
  };

  // ================= block 1: x = RB(192 -> 256)(f0), hidden width 192 = 48 channels per wave =====================
  mv_gather<T>(a, 0, tile0, FR, FA);
  __syncthreads();
  {
    f32x4 h[MF][3];
#pragma unroll
    for (int i = 0; i < MF; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) h[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    m3_gemm<T, MF, 3, 192>(FA, FP, W + mv_off(0), 192, 0, 48 * wave, 0, h);
#pragma unroll
    for (int mf = 0; mf < MF; ++mf)
#pragma unroll
      for (int nf = 0; nf < 3; ++nf) {
        int n = 48 * wave + 16 * nf + 4 * lg;
        f32x4 b4 = *reinterpret_cast<const f32x4*>(fb + MV_B_R1FC0 + n);
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = fmaxf(h[mf][nf][e] + b4[e], 0.f);
        st4<T>(NET + (16 * mf + lr) * XP + n, o);
      }
  }
  zero(xs);
  m3_gemm<T, MF, 4, 192>(FR, FP, W + mv_off(2), 192, 0, nbase, 0, xs);     // shortcut(f0), bias-free
  __syncthreads();
  zero(acc);
  m3_gemm<T, MF, 4, 192>(NET, XP, W + mv_off(1), 192, 0, nbase, 0, acc);   // fc_1(relu(net))
  store_block(fb + MV_B(0), 0.f);
  mv_gather<T>(a, 1, tile0, FR, FA);  // f0 is dead: every wave is past the barrier that followed its last read
  __syncthreads();
  // ================= blocks 2, 3: x = RB(448 -> 256)([x | f_k]) ===================================================
#pragma unroll
  for (int blk = 0; blk < 2; ++blk) {
    const int m0 = 3 + 3 * blk;
    zero(acc);
    m3_gemm<T, MF, 4, 256>(XA, XP, W + mv_off(m0), 448, 0, nbase, 0, acc);
    m3_gemm<T, MF, 4, 192>(FA, FP, W + mv_off(m0), 448, 256, nbase, 0, acc);
    zero(xs);
    m3_gemm<T, MF, 4, 256>(XR, XP, W + mv_off(m0 + 2), 448, 0, nbase, 0, xs);
    m3_gemm<T, MF, 4, 192>(FR, FP, W + mv_off(m0 + 2), 448, 256, nbase, 0, xs);
    store_hidden(fb + MV_B(1 + 2 * blk));
    __syncthreads();  // NET complete; XR/XA/F no longer read
    zero(acc);
    m3_gemm<T, MF, 4, 256>(NET, XP, W + mv_off(m0 + 1), 256, 0, nbase, 0, acc);
    store_block(fb + MV_B(2 + 2 * blk), 0.f);
    if (blk == 0) mv_gather<T>(a, 2, tile0, FR, FA);
    __syncthreads();
  }
  // ================= block 4: x = x + fc_1(relu(fc_0(relu(x)))) ====================================================
  zero(acc);
  m3_gemm<T, MF, 4, 256>(XA, XP, W + mv_off(9), 256, 0, nbase, 0, acc);
  store_hidden(fb + MV_B(5));
  __syncthreads();
  zero(acc);
  m3_gemm<T, MF, 4, 256>(NET, XP, W + mv_off(10), 256, 0, nbase, 0, acc);
  store_block(fb + MV_B(6), 0.2f);  // XA <- leaky_relu(x, 0.2) for net_out (mlp.py:127, 153)
  __syncthreads();
  // ================= net_out: 256 -> 3 (rows padded to 16), points split over the waves ============================
  {
    constexpr int MFW = MF >= 4 ? MF / 4 : 1;
    if (wave * MFW < MF) {
      f32x4 r[MFW][1];
#pragma unroll
      for (int i = 0; i < MFW; ++i) r[i][0] = f32x4{0.f, 0.f, 0.f, 0.f};
      m3_gemm<T, MFW, 1, 256>(XA, XP, W + mv_off(11), 256, 0, 0, wave * MFW, r);
      if (lg == 0) {  // lanes 0..15 hold output channels 0..3 of their point
        const int64_t THW = (int64_t)a.T * a.H * a.W;
#pragma unroll
        for (int i = 0; i < MFW; ++i) {
          int64_t gp = tile0 + 16 * (wave * MFW + i) + lr;
          if (gp < a.total) {
            int64_t b = gp / THW, idx = gp - b * THW;
#pragma unroll
            for (int c = 0; c < 3; ++c) a.out[(b * 3 + c) * THW + idx] = r[i][0][c] + fb[MV_BOUT + c];
          }
        }
      }
    }
  }
}

// =====================================================================================================================
// bf16 perf kernel, second generation (mlp_fc2.h): 96 points per workgroup (two [96][272] activation buffers + the
// [96][208] feature tile = 141 KB of LDS), 8 waves x (32 channels x 96 points), fragment-ordered weights fetched once per
// workgroup, relu applied on the fragment.  First generation: 4.8 ms per 16x256x256 clip (0.37 PFLOP/s).
// =====================================================================================================================
#define MVV2_P 96
#define MVV2_FP 208   // 416-byte feature rows: (row * 104 + 4 lg) mod 64 conflict-free for the b128 lane groups

// raw features of scale `sc`: [xy(h,w) | yt(t,h) | xt(t,w)] bilinear samples -> F[P][FP]
__device__ __forceinline__ void mvv2_gather(const MlpVK& a, int sc, int64_t tile0, bf16_t* F) {
  const int HW = a.H * a.W, THW = a.T * HW;
#pragma unroll   // a thread's 4-5 items: up to 20 texel loads in flight
  for (int rnd = 0; rnd < (MVV2_P * 24 + 511) / 512; ++rnd) {
    const int it = threadIdx.x + 512 * rnd;
    if (it >= MVV2_P * 24) break;
    const int pi = it / 24, cg = it - pi * 24;
    const int j = cg >> 3, cgi = cg & 7;
    const int64_t gp = tile0 + pi;
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (bf16_t)0.f;
    if (gp < a.total) {
      const int b = (int)(gp / THW);
      const int idx = (int)(gp - (int64_t)b * THW);
      const int t = idx / HW, r = idx - t * HW, h = r / a.W, w = r - h * a.W;
      // grid channel 0 -> grid_sample's x (plane width), channel 1 -> y (plane height)   (general_utils.py:27-52, 134-138)
      float gx, gy;
      int Hp;
      const int Wp = a.pw[sc];
      if (j == 0) { gx = a.cxy[h * a.W + w]; gy = a.cxy[HW + h * a.W + w]; Hp = Wp; }
      else if (j == 1) { gx = a.cyt[t * a.H + h]; gy = a.cyt[a.T * a.H + t * a.H + h]; Hp = a.frames; }
      else { gx = a.cxt[t * a.W + w]; gy = a.cxt[a.T * a.W + t * a.W + w]; Hp = a.frames; }
      const float ix = fminf(fmaxf((gx + 1.0f) * 0.5f * (float)(Wp - 1), 0.f), (float)(Wp - 1));
      const float iy = fminf(fmaxf((gy + 1.0f) * 0.5f * (float)(Hp - 1), 0.f), (float)(Hp - 1));
      const float x0f = floorf(ix), y0f = floorf(iy);
      const float wx1 = ix - x0f, wy1 = iy - y0f, wx0 = 1.f - wx1, wy0 = 1.f - wy1;
      const int x0 = (int)x0f, y0 = (int)y0f;
      const int x1 = min(x0 + 1, Wp - 1), y1 = min(y0 + 1, Hp - 1);
      const bf16_t* base = (const bf16_t*)a.planes[j * 3 + sc] + (int64_t)b * Hp * Wp * 64 + cgi * 8;
      const bf16x8 t00 = *reinterpret_cast<const bf16x8*>(base + (y0 * Wp + x0) * 64), t01 = *reinterpret_cast<const bf16x8*>(base + (y0 * Wp + x1) * 64);
      const bf16x8 t10 = *reinterpret_cast<const bf16x8*>(base + (y1 * Wp + x0) * 64), t11 = *reinterpret_cast<const bf16x8*>(base + (y1 * Wp + x1) * 64);
      const float w00 = wy0 * wx0, w01 = wy0 * wx1, w10 = wy1 * wx0, w11 = wy1 * wx1;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (bf16_t)((float)t00[e] * w00 + (float)t01[e] * w01 + (float)t10[e] * w10 + (float)t11[e] * w11);
    }
    *reinterpret_cast<bf16x8*>(F + pi * MVV2_FP + cg * 8) = o;
  }
}

__global__ void __launch_bounds__(512, 1) mlpv_v2_kernel(MlpVK a) {
  constexpr int P = MVV2_P, MF = P / 16, NF = 2, FP = MVV2_FP;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16_t* X0 = (bf16_t*)smem;
  bf16_t* X1 = X0 + P * FC2_XP;
  bf16_t* F = X1 + P * FC2_XP;
  const int lane = threadIdx.x & 63, lr = lane & 15, lg = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g0 = 2 * wave, ch0 = 32 * wave + 8 * lg;   // this lane's eight channels: ch0 .. ch0 + 7
  const int64_t tile0 = (int64_t)blockIdx.x * P;
  const bf16_t* W = (const bf16_t*)a.w;
  const float* fb = a.fb;

  f32x4 acc[MF][NF], cst[NF];
  bf16x8 wq[2][NF];
  auto zero = [&]() {
#pragma unroll
    for (int i = 0; i < MF; ++i)
#pragma unroll
      for (int j = 0; j < NF; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  };
  auto wprime = [&](int off, int KS) { fc2_wload<NF>(W, off, KS, g0, 0, lane, wq[0]); };
  auto cload = [&](const float* table) {
    cst[0] = *reinterpret_cast<const f32x4*>(table + ch0);
    cst[1] = *reinterpret_cast<const f32x4*>(table + ch0 + 4);
  };
  auto store8 = [&](bf16_t* Xout, int row, const f32x4& lo, const f32x4& hi) {
    bf16x8 q;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      q[e] = (bf16_t)lo[e];
      q[4 + e] = (bf16_t)hi[e];
    }
    *reinterpret_cast<bf16x8*>(Xout + row * FC2_XP + ch0) = q;
  };
  auto store_hidden = [&](bf16_t* Xout) {   // relu(acc + bias) -> Xout (rectified: fc_1 reads it as is)
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) {
      f32x4 lo, hi;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        lo[e] = fmaxf(acc[mf][0][e] + cst[0][e], 0.f);
        hi[e] = fmaxf(acc[mf][1][e] + cst[1][e], 0.f);
      }
      store8(Xout, 16 * mf + lr, lo, hi);
    }
  };
  // block output: x = acc (fc_1 + shortcut) + bias [+ identity x_in] -> Xout; slope 1 = raw, 0.2 = leaky_relu for net_out
  auto store_block = [&](const bf16_t* Xid, bf16_t* Xout, float slope) {
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) {
      const int row = 16 * mf + lr;
      f32x4 lo = acc[mf][0], hi = acc[mf][1];
      if (Xid != nullptr) {
        const bf16x8 xi = *reinterpret_cast<const bf16x8*>(Xid + row * FC2_XP + ch0);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          lo[e] += (float)xi[e];
          hi[e] += (float)xi[4 + e];
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        lo[e] += cst[0][e];
        hi[e] += cst[1][e];
        lo[e] = lo[e] > 0.f ? lo[e] : lo[e] * slope;
        hi[e] = hi[e] > 0.f ? hi[e] : hi[e] * slope;
      }
      store8(Xout, row, lo, hi);
    }
  };

  // ================= block 1: x1 = RB(192 -> 256)(f0), hidden width 192 (waves 0..5) ===============================
  wprime(mv_off(0), 6);
  mvv2_gather(a, 0, tile0, F);
  __syncthreads();
  if (wave < 6) {
    zero();
    cload(fb + MV_B_R1FC0);
    fc2_pass<MF, NF, 0, 192, FP, true>(X1, F, W, mv_off(0), g0, lane, wq, acc);    // fc_0(relu(f0)): 192 -> 192
    store_hidden(X1);
  }
  wprime(mv_off(1), 6);
  __syncthreads();
  zero();
  fc2_pass<MF, NF, 192, 0, FP, false>(X1, F, W, mv_off(1), g0, lane, wq, acc);       // fc_1(relu(net)): 192 -> 256
  wprime(mv_off(2), 6);
  cload(fb + MV_B(0));
  fc2_pass<MF, NF, 0, 192, FP, false>(X0, F, W, mv_off(2), g0, lane, wq, acc);       // + shortcut(f0), bias-free
  wprime(mv_off(3), 14);
  store_block(nullptr, X0, 1.f);
  __syncthreads();   // every wave is done with f0 and the hidden tile
  mvv2_gather(a, 1, tile0, F);
  __syncthreads();

#define MVV2_BLOCK(IN, TMP, KF, m_fc0, m_fc1, m_sc, b_fc0, b_fc1, next_scale, next_off, next_KS, slope)               \
  {                                                                                                                   \
    zero();                                                                                                           \
    cload(fb + MV_B(b_fc0));                                                                                          \
    fc2_pass<MF, NF, 256, KF, FP, true>(IN, F, W, mv_off(m_fc0), g0, lane, wq, acc);      /* fc_0(relu([x | f])) */   \
    wprime(mv_off(m_fc1), 8);                                                                                         \
    store_hidden(TMP);                                                                                                \
    __syncthreads();                                                                                                  \
    zero();                                                                                                           \
    cload(fb + MV_B(b_fc1));                                                                                          \
    fc2_pass<MF, NF, 256, 0, FP, false>(TMP, F, W, mv_off(m_fc1), g0, lane, wq, acc);     /* fc_1(relu(net)) */       \
    if ((m_sc) >= 0) wprime(mv_off((m_sc) >= 0 ? (m_sc) : 0), (256 + (KF)) / 32);                                     \
    __syncthreads(); /* every wave has read the hidden tile: the block output goes there */                           \
    if ((m_sc) >= 0)                                                                                                  \
      fc2_pass<MF, NF, 256, KF, FP, false>(IN, F, W, mv_off((m_sc) >= 0 ? (m_sc) : 0), g0, lane, wq, acc);            \
    if ((next_KS) > 0) wprime((next_off), (next_KS));                                                                 \
    store_block((m_sc) >= 0 ? nullptr : IN, TMP, (slope));                                                            \
    __syncthreads();                                                                                                  \
    if ((next_scale) >= 0) {                                                                                          \
      mvv2_gather(a, (next_scale), tile0, F);                                                                         \
      __syncthreads();                                                                                                \
    }                                                                                                                 \
  }
  MVV2_BLOCK(X0, X1, 192, 3, 4, 5, 1, 2, 2, mv_off(6), 14, 1.f);     // block 2: [x1 | f1] -> X1
  MVV2_BLOCK(X1, X0, 192, 6, 7, 8, 3, 4, -1, mv_off(9), 8, 1.f);     // block 3: [x2 | f2] -> X0
  MVV2_BLOCK(X0, X1, 0, 9, 10, -1, 5, 6, -1, 0, 0, 0.2f);            // block 4: identity; leaky_relu(x, 0.2) for net_out (mlp.py:153)
#undef MVV2_BLOCK
  // ================= net_out: 256 -> 3 (rows padded to 16): one 16-point fragment per wave (waves 0..5) =============
  if (wave < MF) {
    f32x4 r = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const bf16x8 wf = *reinterpret_cast<const bf16x8*>(W + fc2_wfrag(mv_off(11), 8, 0, ks) + 8 * lane);
      const bf16x8 xf = *reinterpret_cast<const bf16x8*>(X1 + (16 * wave + lr) * FC2_XP + ks * 32 + 8 * lg);
      r = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, xf, r, 0, 0, 0);
    }
    if (lg == 0) {  // lanes 0..15 hold output channels 0..3 of their point
      const int64_t THW = (int64_t)a.T * a.H * a.W;
      const int64_t gp = tile0 + 16 * wave + lr;
      if (gp < a.total) {
        const int64_t b = gp / THW, idx = gp - b * THW;
#pragma unroll
        for (int c = 0; c < 3; ++c) a.out[(b * 3 + c) * THW + idx] = r[c] + fb[MV_BOUT + c];
      }
    }
  }
}

void mlpv_pack_fragment_order(const float* src, float* dst) {
  for (int i = 0; i < 12; ++i) fc2_pack_matrix(src + mv_off(i), dst + mv_off(i), mv_rows(i), mv_k(i));
}

int launch_mlpv(const MlpVArgs& a, int dt, hipStream_t s) {
  MlpVK k{};
  for (int i = 0; i < 9; ++i) k.planes[i] = a.planes[i];
  for (int i = 0; i < 3; ++i) k.pw[i] = a.plane_hw[i];
  k.frames = a.frames;
  k.cxy = a.cxy; k.cyt = a.cyt; k.cxt = a.cxt;
  k.w = a.w; k.fb = a.fb; k.out = a.out;
  k.B = a.B; k.T = a.T; k.H = a.H; k.W = a.W;
  k.total = (int64_t)a.B * a.T * a.H * a.W;
  double plane_bytes = 0;
  for (int i = 0; i < 3; ++i)
    plane_bytes += (double)a.B * ((double)a.plane_hw[i] * a.plane_hw[i] + 2.0 * a.frames * a.plane_hw[i]) * 64 * dt_size(dt);
  PROF_SCOPE("mlpv_fused", 2.0 * 857088.0 * (double)k.total, plane_bytes + 12.0 * (double)k.total, s);
  if (dt == DT_BF16) {  // weights in fragment order (mlpv_pack_fragment_order)
    const size_t lds = (size_t)(2 * MVV2_P * FC2_XP + MVV2_P * MVV2_FP) * sizeof(bf16_t);
    static bool set = false;
    if (!set) {
      HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(mlpv_v2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      set = true;
    }
    hipLaunchKernelGGL(mlpv_v2_kernel, dim3((unsigned)cdiv64(k.total, MVV2_P)), dim3(512), lds, s, k);
  } else {
    using L = MVL<float>;
    size_t lds = (size_t)(3 * L::P * L::XP + 2 * L::P * L::FP) * sizeof(float);
    static bool set = false;
    if (!set) {
      HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(mlpv_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      set = true;
    }
    hipLaunchKernelGGL(mlpv_kernel<float>, dim3((unsigned)cdiv64(k.total, L::P)), dim3(256), lds, s, k);
  }
  HIP_TRY(hipGetLastError());
  return DDMI_OK;
}
