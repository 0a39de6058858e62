// Fused tri-plane HDPE sampling + occupancy MLP (replaces models/d2c_vae/mlp.py:82-111 `MLP3D.forward` at inference):
//   sample_plane_feature x3 (utils/general_utils.py:115-119, normalize_coordinate :71-94 with both clamp branches),
//   triplane_positional_encoding(mode='add') at three scales (:126-131: bilinear, align_corners=True, border),
//   net_p + four ResnetBlockFC (blocks.py:673-716) + net_out  ->  Bernoulli logits.
//
// Same shape as the image kernel (k_mlp2d.hip): one workgroup = 4 waves = a tile of P query points that never leaves
// the CU; wave w owns output channels [64w, 64w+64) of every 256-wide layer, weights stream L2 -> registers, activations
// live in LDS.  ResnetBlockFC needs relu(x) for fc_0 and raw x for the shortcut, so block outputs are stored twice
// (raw + rectified); the hidden activation is stored rectified only.  0.628 MMAC per point; the nine planes of a shape
// are 2.1 MB bf16 (L2-resident), so the kernel is MFMA-bound (~1 B/pt of plane traffic).
#include "kernels.h"
#include "prof.h"
#include "mlp_common.h"
#include "mlp_fc2.h"

// packed weight matrices (row-major [N][K], storage dtype), in execution order
//  0 r1.fc_0 [64][64]   1 r1.fc_1 [256][64]   2 r1.shortcut [256][64]
//  3 r2.fc_0 [256][320] 4 r2.fc_1 [256][256]  5 r2.shortcut [256][320]
//  6 r3.fc_0 [256][320] 7 r3.fc_1 [256][256]  8 r3.shortcut [256][320]
//  9 r4.fc_0 [256][256] 10 r4.fc_1 [256][256] 11 net_out [16 (1 used)][256]
__host__ __device__ constexpr int m3_rows(int i) { return i == 0 ? 64 : i == 11 ? 16 : 256; }
__host__ __device__ constexpr int m3_k(int i) { return i <= 2 ? 64 : (i == 3 || i == 5 || i == 6 || i == 8) ? 320 : 256; }
__host__ __device__ constexpr int m3_off(int i) {
  int o = 0;
  for (int j = 0; j < i; ++j) o += m3_rows(j) * m3_k(j);
  return o;
}
size_t mlp3d_weight_elems() { return (size_t)m3_off(12); }
// fp32 constants: biases  r1.fc_0[64] | r1.fc_1[256] | r2.fc_0 | r2.fc_1 | r3.fc_0 | r3.fc_1 | r4.fc_0 | r4.fc_1 (256 each)
//                 net_p weight [256][4 (3 used)] | net_p bias [256] | net_out bias [4]
#define M3_B_R1FC0 0
#define M3_B(i) (64 + (i) * 256)  // i = 0: r1.fc_1, 1: r2.fc_0, 2: r2.fc_1, 3: r3.fc_0, 4: r3.fc_1, 5: r4.fc_0, 6: r4.fc_1
#define M3_WP (64 + 7 * 256)
#define M3_BP (M3_WP + 1024)
#define M3_BOUT (M3_BP + 256)
size_t mlp3d_fb_elems() { return M3_BOUT + 4; }

template <typename T> struct M3L;
template <> struct M3L<bf16_t> { static constexpr int P = 64, XP = 272, FP = 72; };
template <> struct M3L<float> { static constexpr int P = 32, XP = 260, FP = 68; };

struct Mlp3dK {
  const void* planes[9];  // [plane j = xy,yz,xz][scale i]: index j*3 + i, each [B][S_i][S_i][64] NHWC
  int hw[3];
  const float* pts;       // [B][N][3]
  const void* w;
  const float* fb;
  float* logits;          // [B][N]
  int B, N;
  int64_t total;
};

// normalize_coordinate (general_utils.py:86-93) followed by grid = 2u-1 and the align_corners=True un-normalisation
__device__ __forceinline__ float m3_texcoord(float p, int S) {
  float u = p / (1.0f + 0.1f + 10e-6f) + 0.5f;
  if (u >= 1.0f) u = 1.0f - 10e-6f;
  if (u < 0.0f) u = 0.0f;
  float g = 2.0f * u - 1.0f;
  float ix = (g + 1.0f) * 0.5f * (float)(S - 1);
  return fminf(fmaxf(ix, 0.f), (float)(S - 1));
}

// features of scale `sc`: sum over the three planes of bilinear samples -> FR (raw) and FA (rectified)
template <typename T>
__device__ __forceinline__ void m3_gather(const Mlp3dK& a, int sc, int64_t tile0, T* FR, T* FA) {
  constexpr int P = M3L<T>::P, FP = M3L<T>::FP;
  constexpr int EPS = sizeof(T) == 2 ? 8 : 4;
  constexpr int CG = 64 / EPS;
  const int S = a.hw[sc];
  for (int it = threadIdx.x; it < P * CG; it += 256) {
    int pi = it / CG, cg = it - pi * CG;
    int64_t gp = tile0 + pi;
    float v[EPS];
#pragma unroll
    for (int e = 0; e < EPS; ++e) v[e] = 0.f;
    if (gp < a.total) {
      int b = (int)(gp / a.N);
      const float* pt = a.pts + gp * 3;
      float c3[3] = {pt[0], pt[1], pt[2]};
      // plane j: xy -> axes (0,1), yz -> (1,2), xz -> (0,2)   (general_utils.py:79-84)
      const int ax0[3] = {0, 1, 0}, ax1[3] = {1, 2, 2};
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        float ix = m3_texcoord(c3[ax0[j]], S), iy = m3_texcoord(c3[ax1[j]], S);
        float x0f = floorf(ix), y0f = floorf(iy);
        float wx1 = ix - x0f, wy1 = iy - y0f, wx0 = 1.f - wx1, wy0 = 1.f - wy1;
        int x0 = (int)x0f, y0 = (int)y0f;
        int x1 = min(x0 + 1, S - 1), y1 = min(y0 + 1, S - 1);
        const T* base = (const T*)a.planes[j * 3 + sc] + (int64_t)b * S * S * 64 + cg * EPS;
        const T* p00 = base + ((int64_t)y0 * S + x0) * 64;
        const T* p01 = base + ((int64_t)y0 * S + x1) * 64;
        const T* p10 = base + ((int64_t)y1 * S + x0) * 64;
        const T* p11 = base + ((int64_t)y1 * S + x1) * 64;
        float w00 = wy0 * wx0, w01 = wy0 * wx1, w10 = wy1 * wx0, w11 = wy1 * wx1;
        if constexpr (sizeof(T) == 2) {
          bf16x8 t00 = *reinterpret_cast<const bf16x8*>(p00), t01 = *reinterpret_cast<const bf16x8*>(p01);
          bf16x8 t10 = *reinterpret_cast<const bf16x8*>(p10), t11 = *reinterpret_cast<const bf16x8*>(p11);
#pragma unroll
          for (int e = 0; e < 8; ++e)
            v[e] += (float)t00[e] * w00 + (float)t01[e] * w01 + (float)t10[e] * w10 + (float)t11[e] * w11;
        } else {
          f32x4 t00 = *reinterpret_cast<const f32x4*>(p00), t01 = *reinterpret_cast<const f32x4*>(p01);
          f32x4 t10 = *reinterpret_cast<const f32x4*>(p10), t11 = *reinterpret_cast<const f32x4*>(p11);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += t00[e] * w00 + t01[e] * w01 + t10[e] * w10 + t11[e] * w11;
        }
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
__global__ void __launch_bounds__(256, 1) mlp3d_kernel(Mlp3dK a) {
  constexpr int P = M3L<T>::P, XP = M3L<T>::XP, FP = M3L<T>::FP;
  constexpr int MF = P / 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  T* XR = (T*)smem;          // block output, raw        [P][XP]
  T* XA = XR + P * XP;       // block output, rectified
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
  // NET <- relu(acc + bias)
  auto store_hidden = [&](const float* bias) {
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
  // x = xs + acc + bias (+ extra) -> XR, XA ; xs <- x
  auto store_block = [&](const float* bias, bool with_netp) {
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) {
      int row = 16 * mf + lr;
      float px = 0.f, py = 0.f, pz = 0.f;
      if (with_netp) {
        int64_t gp = tile0 + row;
        if (gp < a.total) { px = a.pts[gp * 3]; py = a.pts[gp * 3 + 1]; pz = a.pts[gp * 3 + 2]; }
      }
#pragma unroll
      for (int nf = 0; nf < 4; ++nf) {
        int n = nbase + 16 * nf + 4 * lg;
        f32x4 b4 = *reinterpret_cast<const f32x4*>(bias + n);
        float o[4], r[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float v = xs[mf][nf][e] + acc[mf][nf][e] + b4[e];
          if (with_netp) {  // net_p(coords): Linear(3 -> 256)  (mlp.py:103)
            const float* wp = fb + M3_WP + (n + e) * 4;
            v += wp[0] * px + wp[1] * py + wp[2] * pz + fb[M3_BP + n + e];
          }
          o[e] = v;
          r[e] = fmaxf(v, 0.f);
        }
        xs[mf][nf] = f32x4{o[0], o[1], o[2], o[3]};
        st4<T>(XR + row * XP + n, o);
        st4<T>(XA + row * XP + n, r);
      }
    }
  };

  // ================= block 1: x = net_p(p) + RB(64 -> 256)(f0), hidden width 64 =================================
  m3_gather<T>(a, 0, tile0, FR, FA);
  __syncthreads();
  {
    f32x4 h[MF][1];
#pragma unroll
    for (int i = 0; i < MF; ++i) h[i][0] = f32x4{0.f, 0.f, 0.f, 0.f};
    m3_gemm<T, MF, 1, 64>(FA, FP, W + m3_off(0), 64, 0, 16 * wave, 0, h);  // fc_0: 64 -> 64, 16 channels per wave
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) {
      int n = 16 * wave + 4 * lg;
      f32x4 b4 = *reinterpret_cast<const f32x4*>(fb + M3_B_R1FC0 + n);
      float o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = fmaxf(h[mf][0][e] + b4[e], 0.f);
      st4<T>(NET + (16 * mf + lr) * XP + n, o);
    }
  }
  zero(xs);
  m3_gemm<T, MF, 4, 64>(FR, FP, W + m3_off(2), 64, 0, nbase, 0, xs);        // shortcut(f0), bias-free
  __syncthreads();
  zero(acc);
  m3_gemm<T, MF, 4, 64>(NET, XP, W + m3_off(1), 64, 0, nbase, 0, acc);      // fc_1(relu(net))
  store_block(fb + M3_B(0), true);
  m3_gather<T>(a, 1, tile0, FR, FA);  // f0 is dead: every wave is past the barrier that followed its last read
  __syncthreads();
  // ================= blocks 2, 3: x = RB(320 -> 256)([x | f_k]) ===================================================
#pragma unroll
  for (int blk = 0; blk < 2; ++blk) {
    const int m0 = 3 + 3 * blk;  // fc_0, fc_1, shortcut matrices of this block
    zero(acc);
    m3_gemm<T, MF, 4, 256>(XA, XP, W + m3_off(m0), 320, 0, nbase, 0, acc);
    m3_gemm<T, MF, 4, 64>(FA, FP, W + m3_off(m0), 320, 256, nbase, 0, acc);
    zero(xs);
    m3_gemm<T, MF, 4, 256>(XR, XP, W + m3_off(m0 + 2), 320, 0, nbase, 0, xs);
    m3_gemm<T, MF, 4, 64>(FR, FP, W + m3_off(m0 + 2), 320, 256, nbase, 0, xs);
    store_hidden(fb + M3_B(1 + 2 * blk));
    __syncthreads();  // NET complete; XR/XA/F no longer read
    zero(acc);
    m3_gemm<T, MF, 4, 256>(NET, XP, W + m3_off(m0 + 1), 256, 0, nbase, 0, acc);
    store_block(fb + M3_B(2 + 2 * blk), false);
    if (blk == 0) m3_gather<T>(a, 2, tile0, FR, FA);
    __syncthreads();
  }
  // ================= block 4: x = x + fc_1(relu(fc_0(relu(x)))) ====================================================
  zero(acc);
  m3_gemm<T, MF, 4, 256>(XA, XP, W + m3_off(9), 256, 0, nbase, 0, acc);
  store_hidden(fb + M3_B(5));
  __syncthreads();
  zero(acc);
  m3_gemm<T, MF, 4, 256>(NET, XP, W + m3_off(10), 256, 0, nbase, 0, acc);
  store_block(fb + M3_B(6), false);  // xs (= x3, fp32 in registers) + dx
  __syncthreads();
  // ================= net_out: 256 -> 1 (rows padded to 16), points split over the waves ============================
  {
    constexpr int MFW = MF >= 4 ? MF / 4 : 1;
    if (wave * MFW < MF) {
      f32x4 r[MFW][1];
#pragma unroll
      for (int i = 0; i < MFW; ++i) r[i][0] = f32x4{0.f, 0.f, 0.f, 0.f};
      m3_gemm<T, MFW, 1, 256>(XR, XP, W + m3_off(11), 256, 0, 0, wave * MFW, r);
      if (lg == 0) {
        float bo = fb[M3_BOUT];
#pragma unroll
        for (int i = 0; i < MFW; ++i) {
          int64_t gp = tile0 + 16 * (wave * MFW + i) + lr;
          if (gp < a.total) a.logits[gp] = r[i][0][0] + bo;
        }
      }
    }
  }
}

// =====================================================================================================================
// bf16 perf kernel, second generation (mlp_fc2.h): 128 points per workgroup, 8 waves x (32 channels x 128 points),
// fragment-ordered weights fetched once per workgroup, raw activations in two LDS buffers, relu applied on the fragment.
// Measured on the 128^3 grid: first generation 6.6 ms (0.40 PFLOP/s).
// =====================================================================================================================
#define M3V2_P 128
#define M3V2_FP 80   // feature row pitch (160 B): conflict-free for the b128 lane groups

// raw features of scale `sc` (sum over the three planes of bilinear samples) -> F[P][FP]
template <int NT>
__device__ __forceinline__ void m3v2_gather(const Mlp3dK& a, int sc, int64_t tile0, bf16_t* F) {
  const int S = a.hw[sc];
#pragma unroll   // all items of a thread: 24 (NT = 512) / 48 (NT = 256) texel loads in flight (the accumulators are dead at a block boundary)
  for (int rnd = 0; rnd < M3V2_P * 8 / NT; ++rnd) {
    const int it = threadIdx.x + NT * rnd;
    const int pi = it >> 3, cg = it & 7;
    const int64_t gp = tile0 + pi;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 0.f;
    if (gp < a.total) {
      const int b = (int)(gp / a.N);
      const float* pt = a.pts + gp * 3;
      const float c3[3] = {pt[0], pt[1], pt[2]};
      // plane j: xy -> axes (0,1), yz -> (1,2), xz -> (0,2)   (general_utils.py:79-84)
      const int ax0[3] = {0, 1, 0}, ax1[3] = {1, 2, 2};
      bf16x8 t[3][4];
      float w[3][4];
#pragma unroll
      for (int j = 0; j < 3; ++j) {   // all twelve texel loads of the point in flight before the first blend
        const float ix = m3_texcoord(c3[ax0[j]], S), iy = m3_texcoord(c3[ax1[j]], S);
        const float x0f = floorf(ix), y0f = floorf(iy);
        const float wx1 = ix - x0f, wy1 = iy - y0f, wx0 = 1.f - wx1, wy0 = 1.f - wy1;
        const int x0 = (int)x0f, y0 = (int)y0f;
        const int x1 = min(x0 + 1, S - 1), y1 = min(y0 + 1, S - 1);
        const bf16_t* base = (const bf16_t*)a.planes[j * 3 + sc] + (int64_t)b * S * S * 64 + cg * 8;
        t[j][0] = *reinterpret_cast<const bf16x8*>(base + (y0 * S + x0) * 64);
        t[j][1] = *reinterpret_cast<const bf16x8*>(base + (y0 * S + x1) * 64);
        t[j][2] = *reinterpret_cast<const bf16x8*>(base + (y1 * S + x0) * 64);
        t[j][3] = *reinterpret_cast<const bf16x8*>(base + (y1 * S + x1) * 64);
        w[j][0] = wy0 * wx0; w[j][1] = wy0 * wx1; w[j][2] = wy1 * wx0; w[j][3] = wy1 * wx1;
      }
#pragma unroll
      for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e)   // the first generation's (and the fp32 kernel's) operation order
          v[e] += (float)t[j][0][e] * w[j][0] + (float)t[j][1][e] * w[j][1] + (float)t[j][2][e] * w[j][2] + (float)t[j][3][e] * w[j][3];
    }
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (bf16_t)v[e];
    *reinterpret_cast<bf16x8*>(F + pi * M3V2_FP + cg * 8) = o;
  }
}

template <int NW, int MF, int NF, int KX, int KF, int FP, bool RELU>
__device__ __forceinline__ void m3_pass(const bf16_t* __restrict__ X, const bf16_t* __restrict__ F, const bf16_t* __restrict__ W, int off, int g0,
                                        int lane, bf16x8 (&wq)[2][NF], f32x4 (&acc)[MF][NF]) {
  if constexpr (NW == 4) fc2_pass_halves<MF, NF, KX, KF, FP, RELU>(X, F, W, off, g0, lane, wq, acc);
  else fc2_pass<MF, NF, KX, KF, FP, RELU>(X, F, W, off, g0, lane, wq, acc);
}

// NW = 8: wave w owns the channel pair-group [32 w, 32 w + 32) (the second generation as measured in rounds 2-4).
// NW = 4 (round 5): wave w owns TWO pair-groups, [64 w, 64 w + 64), still for all 128 points - one wave per SIMD.  An activation
// fragment read from LDS now feeds four MFMAs instead of two: per k-step the CU's waves read 32 KB of activations where eight waves
// read 64 KB (= the whole LDS bandwidth over the 512 clocks the k-step's MFMAs take), the relu on the fragments is done half as often,
// and every weight byte is still fetched once per workgroup (the 64 x 64 wave tile that also halves the LDS reads doubles them).
// NW = 8, PH = 2 (round 5, the split the round-4 verdict asked to see measured): wave = 64 channels x 64 points - channel group wave % 4,
// point half wave / 4.  Half the LDS reads per MFMA as well, at two waves per SIMD, but the two point halves each fetch the weights.
template <int NW, int PH>
__global__ void __launch_bounds__(64 * NW, 1) mlp3d_v2_kernel(Mlp3dK a) {
  constexpr int P = M3V2_P, MF = P / 16 / PH, PR = 8 * PH / NW, NF = 2 * PR, FP = M3V2_FP, CW = NW / PH, NOUT = P / 16 / NW;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16_t* X0 = (bf16_t*)smem;
  bf16_t* X1 = X0 + P * FC2_XP;
  bf16_t* F = X1 + P * FC2_XP;
  const int lane = threadIdx.x & 63, lr = lane & 15, lg = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int cgp = wave % CW, r0 = (wave / CW) * (P / PH);     // channel group; first point row of this wave's point range
  const int g0 = NF * cgp, ch0 = 32 * PR * cgp + 8 * lg;      // this lane's channels: ch0 + 32 pr .. + 7 of pair-group pr
  const int xo = r0 * FC2_XP, fo = r0 * FP;
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
#pragma unroll
    for (int pr = 0; pr < PR; ++pr) {
      cst[2 * pr] = *reinterpret_cast<const f32x4*>(table + ch0 + 32 * pr);
      cst[2 * pr + 1] = *reinterpret_cast<const f32x4*>(table + ch0 + 32 * pr + 4);
    }
  };
  auto store8 = [&](bf16_t* Xout, int row, int pr, const f32x4& lo, const f32x4& hi) {
    bf16x8 q;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      q[e] = (bf16_t)lo[e];
      q[4 + e] = (bf16_t)hi[e];
    }
    *reinterpret_cast<bf16x8*>(Xout + row * FC2_XP + ch0 + 32 * pr) = q;
  };
  // hidden activation: relu(acc + bias) -> Xout (rectified: fc_1 reads it as is)
  auto store_hidden = [&](bf16_t* Xout) {
#pragma unroll
    for (int mf = 0; mf < MF; ++mf)
#pragma unroll
      for (int pr = 0; pr < PR; ++pr) {
        f32x4 lo, hi;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          lo[e] = fmaxf(acc[mf][2 * pr][e] + cst[2 * pr][e], 0.f);
          hi[e] = fmaxf(acc[mf][2 * pr + 1][e] + cst[2 * pr + 1][e], 0.f);
        }
        store8(Xout, r0 + 16 * mf + lr, pr, lo, hi);
      }
  };
  // block output: x = acc (fc_1 + shortcut) + bias [+ identity x_in] -> Xout, raw
  auto store_block = [&](const bf16_t* Xid, bf16_t* Xout) {
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) {
      const int row = r0 + 16 * mf + lr;
#pragma unroll
      for (int pr = 0; pr < PR; ++pr) {
        f32x4 lo = acc[mf][2 * pr], hi = acc[mf][2 * pr + 1];
        if (Xid != nullptr) {
          const bf16x8 xi = *reinterpret_cast<const bf16x8*>(Xid + row * FC2_XP + ch0 + 32 * pr);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            lo[e] += (float)xi[e];
            hi[e] += (float)xi[4 + e];
          }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          lo[e] += cst[2 * pr][e];
          hi[e] += cst[2 * pr + 1][e];
        }
        store8(Xout, row, pr, lo, hi);
      }
    }
  };

  // ================= block 1: x1 = net_p(p) + RB(64 -> 256)(f0), hidden width 64 (waves 0, 1) ======================
  wprime(m3_off(0), 2);
  m3v2_gather<64 * NW>(a, 0, tile0, F);
  __syncthreads();
  if (cgp * PR < 2) {   // the 64 hidden channels = two pair-groups
    zero();
    cload(fb + M3_B_R1FC0);
    m3_pass<NW, MF, NF, 0, 64, FP, true>(X1 + xo, F + fo, W, m3_off(0), g0, lane, wq, acc);     // fc_0(relu(f0)): 64 -> 64
    store_hidden(X1);
  }
  wprime(m3_off(1), 2);
  __syncthreads();
  zero();
  m3_pass<NW, MF, NF, 64, 0, FP, false>(X1 + xo, F + fo, W, m3_off(1), g0, lane, wq, acc);        // fc_1(relu(net)): 64 -> 256
  wprime(m3_off(2), 2);
  cload(fb + M3_B(0));
  m3_pass<NW, MF, NF, 0, 64, FP, false>(X0 + xo, F + fo, W, m3_off(2), g0, lane, wq, acc);        // + shortcut(f0), bias-free
  wprime(m3_off(3), 10);
  {  // + net_p(coords): Linear(3 -> 256) on the fp32 coordinates (mlp.py:103)
#pragma unroll
    for (int pr = 0; pr < PR; ++pr) {
      float wp[8][3], bp[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const f32x4 w4 = *reinterpret_cast<const f32x4*>(fb + M3_WP + (ch0 + 32 * pr + c) * 4);
        wp[c][0] = w4[0]; wp[c][1] = w4[1]; wp[c][2] = w4[2];
        bp[c] = fb[M3_BP + ch0 + 32 * pr + c];
      }
#pragma unroll
      for (int mf = 0; mf < MF; ++mf) {
        const int64_t gp = tile0 + r0 + 16 * mf + lr;
        float px = 0.f, py = 0.f, pz = 0.f;
        if (gp < a.total) { px = a.pts[gp * 3]; py = a.pts[gp * 3 + 1]; pz = a.pts[gp * 3 + 2]; }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          acc[mf][2 * pr][e] += wp[e][0] * px + wp[e][1] * py + wp[e][2] * pz + bp[e];
          acc[mf][2 * pr + 1][e] += wp[4 + e][0] * px + wp[4 + e][1] * py + wp[4 + e][2] * pz + bp[4 + e];
        }
      }
    }
  }
  store_block(nullptr, X0);
  __syncthreads();   // every wave is done with f0 and the hidden tile
  if (!(FC2_ABL & 8)) m3v2_gather<64 * NW>(a, 1, tile0, F);
  __syncthreads();

  // ---- blocks 2-4.  IN = raw block input, TMP = the other buffer (hidden, then the block output).
#define M3V2_BLOCK(IN, TMP, KF, m_fc0, m_fc1, m_sc, b_fc0, b_fc1, next_scale, next_off, next_KS)                      \
  {                                                                                                                   \
    zero();                                                                                                           \
    cload(fb + M3_B(b_fc0));                                                                                          \
    m3_pass<NW, MF, NF, 256, KF, FP, true>(IN + xo, F + fo, W, m3_off(m_fc0), g0, lane, wq, acc);      /* fc_0(relu([x | f])) */   \
    wprime(m3_off(m_fc1), 8);                                                                                         \
    store_hidden(TMP);                                                                                                \
    __syncthreads();                                                                                                  \
    zero();                                                                                                           \
    cload(fb + M3_B(b_fc1));                                                                                          \
    m3_pass<NW, MF, NF, 256, 0, FP, false>(TMP + xo, F + fo, W, m3_off(m_fc1), g0, lane, wq, acc);     /* fc_1(relu(net)) */       \
    if ((m_sc) >= 0) wprime(m3_off((m_sc) >= 0 ? (m_sc) : 0), (256 + (KF)) / 32);                                     \
    __syncthreads(); /* every wave has read the hidden tile: the block output goes there */                           \
    if ((m_sc) >= 0)                                                                                                  \
      m3_pass<NW, MF, NF, 256, KF, FP, false>(IN + xo, F + fo, W, m3_off((m_sc) >= 0 ? (m_sc) : 0), g0, lane, wq, acc);            \
    if ((next_KS) > 0) wprime((next_off), (next_KS));                                                                 \
    store_block((m_sc) >= 0 ? nullptr : IN, TMP);                                                                     \
    __syncthreads();                                                                                                  \
    if ((next_scale) >= 0 && !(FC2_ABL & 8)) {                                                                       \
      m3v2_gather<64 * NW>(a, (next_scale), tile0, F);                                                                         \
      __syncthreads();                                                                                                \
    }                                                                                                                 \
  }
  M3V2_BLOCK(X0, X1, 64, 3, 4, 5, 1, 2, 2, m3_off(6), 10);     // block 2: [x1 | f1] -> X1
  M3V2_BLOCK(X1, X0, 64, 6, 7, 8, 3, 4, -1, m3_off(9), 8);     // block 3: [x2 | f2] -> X0
  M3V2_BLOCK(X0, X1, 0, 9, 10, -1, 5, 6, -1, 0, 0);            // block 4: identity shortcut -> X1
#undef M3V2_BLOCK
  // ================= net_out: 256 -> 1 (rows padded to 16): 16-point fragments, NOUT per wave ========================
#pragma unroll
  for (int i = 0; i < NOUT; ++i) {
    const int pf = NOUT * wave + i;
    f32x4 r = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const bf16x8 wf = *reinterpret_cast<const bf16x8*>(W + fc2_wfrag(m3_off(11), 8, 0, ks) + 8 * lane);
      const bf16x8 xf = *reinterpret_cast<const bf16x8*>(X1 + (16 * pf + lr) * FC2_XP + ks * 32 + 8 * lg);
      r = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, xf, r, 0, 0, 0);
    }
    if (lg == 0) {
      const int64_t gp = tile0 + 16 * pf + lr;
      if (gp < a.total) a.logits[gp] = r[0] + fb[M3_BOUT];
    }
  }
}

// host: the bf16 kernel's weight image (fragment order, paired rows) from the row-major pack
void mlp3d_pack_fragment_order(const float* src, float* dst) {
  for (int i = 0; i < 12; ++i) fc2_pack_matrix(src + m3_off(i), dst + m3_off(i), m3_rows(i), m3_k(i));
}

int launch_mlp3d(const Mlp3dArgs& a, int dt, hipStream_t s) {
  Mlp3dK k{};
  for (int i = 0; i < 9; ++i) k.planes[i] = a.planes[i];
  for (int i = 0; i < 3; ++i) k.hw[i] = a.plane_hw[i];
  k.pts = a.pts; k.w = a.w; k.fb = a.fb; k.logits = a.logits;
  k.B = a.B; k.N = a.N; k.total = (int64_t)a.B * a.N;
  double plane_bytes = 0;
  for (int i = 0; i < 3; ++i) plane_bytes += 3.0 * a.B * a.plane_hw[i] * a.plane_hw[i] * 64 * dt_size(dt);
  PROF_SCOPE("mlp3d_fused", 2.0 * 627712.0 * (double)k.total, plane_bytes + 16.0 * (double)k.total, s);
  if (dt == DT_BF16) {  // weights in fragment order (mlp3d_pack_fragment_order)
    const size_t lds = (size_t)(2 * M3V2_P * FC2_XP + M3V2_P * M3V2_FP) * sizeof(bf16_t);
    // DDMI_MLP3D_WAVES (read per launch: the tests A/B the forms) - 8: the shipped form; 4: four waves x (64 channels x 128 points);
    // 64: eight waves x (64 channels x 64 points).  Both alternatives are bit-identical and measured slower.
    const char* nwe = getenv("DDMI_MLP3D_WAVES");
    const int nw = nwe ? atoi(nwe) : 8;
    static bool set = false;
    if (!set) {
      HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(mlp3d_v2_kernel<8, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(mlp3d_v2_kernel<4, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(mlp3d_v2_kernel<8, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      set = true;
    }
    const dim3 grid((unsigned)cdiv64(k.total, M3V2_P));
    if (nw == 4) hipLaunchKernelGGL((mlp3d_v2_kernel<4, 1>), grid, dim3(256), lds, s, k);
    else if (nw == 64) hipLaunchKernelGGL((mlp3d_v2_kernel<8, 2>), grid, dim3(512), lds, s, k);
    else hipLaunchKernelGGL((mlp3d_v2_kernel<8, 1>), grid, dim3(512), lds, s, k);
  } else {
    using L = M3L<float>;
    size_t lds = (size_t)(3 * L::P * L::XP + 2 * L::P * L::FP) * sizeof(float);
    static bool set = false;
    if (!set) {
      HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(mlp3d_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      set = true;
    }
    hipLaunchKernelGGL(mlp3d_kernel<float>, dim3((unsigned)cdiv64(k.total, L::P)), dim3(256), lds, s, k);
  }
  HIP_TRY(hipGetLastError());
  return DDMI_OK;
}
