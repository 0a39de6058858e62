// Implicit-GEMM convolution on MFMA for gfx950 (the UNet / decoder hot op).
//
//   y[m][n] = sum_{tap, c} x[pixel(m) + tap][c] * w[n][tap][c]        m = (b, oh, ow),  n = output channel
//
// * activations NHWC, weights packed [Cout][tap][Cin]: both GEMM operands are K-contiguous, so one 16-byte load
//   per lane feeds an MFMA fragment directly (no transposes anywhere).
// * K is walked in chunks of 32 channels of one tap; KPS chunks per pipeline stage; the im2col gather (padding,
//   stride 2, fused nearest x2 upsample) happens in the global->register load, never materialised.
// * register-staged double buffering: loads of stage s+1 are issued before the MFMAs of stage s and written to the
//   other LDS buffer afterwards -> one barrier per stage.
// * bf16 path: v_mfma_f32_16x16x32_bf16; LDS rows are 64 B with a 16-byte-slot XOR swizzle that makes every
//   ds_read_b128 lane group hit 16 distinct slots (see swz()).
//   fp32 (parity) path: v_mfma_f32_16x16x4_f32 - exact fp32 FMA chains - with 144-byte padded rows.
// * operands are swapped (A-operand = weights, B-operand = pixels) so each lane ends up with 4 consecutive output
//   channels of one pixel: bias / time-embedding / residual / store are 8- or 16-byte vector accesses.
// * epilogue fuses + bias[n] + emb[b][n] + residual, * out_scale, optional fp32-NCHW store, optional split-K
//   (fp32 partial slabs [z][M][Cout] written with plain 16-byte stores, summed by splitk_finish_kernel).
#include "kernels.h"
#include "prof.h"

struct ConvK {
  const void* x;
  const void* w;
  const float* bias;
  const float* emb;
  const void* res;
  void* y;
  float* y_nchw;
  float* ws;
  int B, Hin, Win, Cin, Hout, Wout, Cout;
  int KH, KW, stride, pad, ups;
  int emb_stride;
  int M;        // rows per batch entry
  int CPT;      // chunks per tap = ceil(Cin / 32)
  int NK;       // total chunks
  int NS;       // total stages
  int SPS;      // stages per split
  int tiles_n;
  float out_scale;
  int64_t x_bstride, w_bstride, y_bstride;  // element strides for blockIdx.y (batched GEMM mode)
};

template <typename T> struct Tile;
template <> struct Tile<bf16_t> {
  static constexpr int EPS = 8;     // elements per 16-byte slot
  static constexpr int SLOTS = 4;   // slots per 32-element chunk row
  static constexpr int ROWB = 64;   // LDS bytes per row
};
template <> struct Tile<float> {
  static constexpr int EPS = 4;
  static constexpr int SLOTS = 8;
  static constexpr int ROWB = 144;  // 128 + 16 pad
};

// 16-byte slot swizzle for 64-byte rows: f = {0,2,3,1}[(row>>2)&3]; with it the four ds_read_b128 lane groups
// {0-3,12-15,20-27}, {4-11,16-19,28-31}, ... each touch 16 distinct slots of the 256-byte bank row.
__device__ __forceinline__ int swz(int row, int slot) {
  int r = (row >> 2) & 3;
  int f = (0x78 >> (r * 2)) & 3;  // table {0,2,3,1} packed two bits per entry
  return slot ^ f;
}

template <typename T, int BM, int BN, int WM, int WN, int KPS>
__global__ void __launch_bounds__(WM * WN * 64) conv_igemm_kernel(ConvK a) {
  using TL = Tile<T>;
  constexpr int NT = WM * WN * 64;
  constexpr int EPS = TL::EPS, SLOTS = TL::SLOTS, ROWB = TL::ROWB;
  constexpr int TM = BM / WM, TN = BN / WN, FM = TM / 16, FN = TN / 16;
  static_assert(TM % 16 == 0 && TN % 16 == 0, "wave tile must be a multiple of 16");
  constexpr int A_ITEMS = KPS * BM * SLOTS, B_ITEMS = KPS * BN * SLOTS;
  constexpr int A_IT = (A_ITEMS + NT - 1) / NT, B_IT = (B_ITEMS + NT - 1) / NT;
  constexpr int A_BYTES = KPS * BM * ROWB, B_BYTES = KPS * BN * ROWB;
  constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
  constexpr bool IS_BF16 = sizeof(T) == 2;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int tile_m = blockIdx.x / a.tiles_n, tile_n = blockIdx.x % a.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const T* __restrict__ xg = (const T*)a.x + (int64_t)blockIdx.y * a.x_bstride;
  const T* __restrict__ wg = (const T*)a.w + (int64_t)blockIdx.y * a.w_bstride;
  const int Ktot = a.KH * a.KW * a.Cin;
  const int Hlim = a.ups ? 2 * a.Hin : a.Hin, Wlim = a.ups ? 2 * a.Win : a.Win;

  // ---- per-thread gather descriptors (fixed over the K loop) ---------------------------------------
  int a_ih0[A_IT], a_iw0[A_IT], a_lds[A_IT], a_chunk[A_IT], a_coff[A_IT];
  int64_t a_base[A_IT];
  bool a_ok[A_IT];
#pragma unroll
  for (int i = 0; i < A_IT; ++i) {
    int idx = tid + i * NT;
    int chunk = idx / (BM * SLOTS);
    int rem = idx - chunk * (BM * SLOTS);
    int row = rem / SLOTS, slot = rem - row * SLOTS;
    int m = m0 + row;
    bool ok = (idx < A_ITEMS) && (m < a.M);
    int hw = a.Hout * a.Wout;
    int b = ok ? m / hw : 0;
    int r = ok ? m - b * hw : 0;
    int oh = r / a.Wout, ow = r - oh * a.Wout;
    a_ok[i] = ok;
    a_ih0[i] = oh * a.stride - a.pad;
    a_iw0[i] = ow * a.stride - a.pad;
    a_base[i] = (int64_t)b * a.Hin * a.Win * a.Cin;
    a_chunk[i] = chunk;
    a_coff[i] = slot * EPS;
    a_lds[i] = chunk * (BM * ROWB) + row * ROWB + (IS_BF16 ? swz(row, slot) : slot) * 16;
  }
  int b_lds[B_IT], b_chunk[B_IT], b_coff[B_IT];
  int64_t b_base[B_IT];
  bool b_ok[B_IT];
#pragma unroll
  for (int i = 0; i < B_IT; ++i) {
    int idx = tid + i * NT;
    int chunk = idx / (BN * SLOTS);
    int rem = idx - chunk * (BN * SLOTS);
    int row = rem / SLOTS, slot = rem - row * SLOTS;
    int n = n0 + row;
    b_ok[i] = (idx < B_ITEMS) && (n < a.Cout);
    b_base[i] = (int64_t)n * Ktot;
    b_chunk[i] = chunk;
    b_coff[i] = slot * EPS;
    b_lds[i] = A_BYTES + chunk * (BN * ROWB) + row * ROWB + (IS_BF16 ? swz(row, slot) : slot) * 16;
  }

  uint4 ra[A_IT], rb[B_IT];
  const uint4 zero4 = make_uint4(0, 0, 0, 0);

  auto load_stage = [&](int s) {
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      int kc = s * KPS + a_chunk[i];
      int tap = kc / a.CPT;
      int c = (kc - tap * a.CPT) * 32 + a_coff[i];
      int kh = tap / a.KW, kw = tap - kh * a.KW;
      int ih = a_ih0[i] + kh, iw = a_iw0[i] + kw;
      bool ok = a_ok[i] && kc < a.NK && c < a.Cin && ih >= 0 && ih < Hlim && iw >= 0 && iw < Wlim;
      if (a.ups) { ih >>= 1; iw >>= 1; }
      const T* p = xg + a_base[i] + ((int64_t)ih * a.Win + iw) * a.Cin + c;
      ra[i] = ok ? *reinterpret_cast<const uint4*>(p) : zero4;
    }
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
      int kc = s * KPS + b_chunk[i];
      int tap = kc / a.CPT;
      int c = (kc - tap * a.CPT) * 32 + b_coff[i];
      bool ok = b_ok[i] && kc < a.NK && c < a.Cin;
      const T* p = wg + b_base[i] + (int64_t)tap * a.Cin + c;
      rb[i] = ok ? *reinterpret_cast<const uint4*>(p) : zero4;
    }
  };
  auto store_stage = [&](int buf) {
    unsigned char* base = smem + buf * STAGE_BYTES;
#pragma unroll
    for (int i = 0; i < A_IT; ++i)
      if (tid + i * NT < A_ITEMS) *reinterpret_cast<uint4*>(base + a_lds[i]) = ra[i];
#pragma unroll
    for (int i = 0; i < B_IT; ++i)
      if (tid + i * NT < B_ITEMS) *reinterpret_cast<uint4*>(base + b_lds[i]) = rb[i];
  };

  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int s_begin = blockIdx.z * a.SPS;
  const int s_end = min(a.NS, s_begin + a.SPS);
  const int lr = lane & 15, lg = lane >> 4;

  if (s_begin < s_end) {
    load_stage(s_begin);
    store_stage(0);
  }
  __syncthreads();

  for (int s = s_begin; s < s_end; ++s) {
    const int buf = (s - s_begin) & 1;
    if (s + 1 < s_end) load_stage(s + 1);
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

  // ---- epilogue: lane holds channels n..n+3 of pixel m for every (i, j) fragment --------------------
  const int hw = a.Hout * a.Wout;
  T* __restrict__ yg = a.y ? (T*)a.y + (int64_t)blockIdx.y * a.y_bstride : nullptr;
  const T* __restrict__ rg = (const T*)a.res;
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    int m = m0 + wm * TM + i * 16 + lr;
    if (m >= a.M) continue;
    int b = m / hw;
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      int n = n0 + wn * TN + j * 16 + lg * 4;
      if (n >= a.Cout) continue;
      f32x4 v = acc[i][j];
      if (a.ws != nullptr) {  // split-K partial slab [z][M][Cout], summed by splitk_finish_kernel
        *reinterpret_cast<f32x4*>(a.ws + ((int64_t)blockIdx.z * a.M + m) * a.Cout + n) = v;
        continue;
      }
      if (a.bias) {
        f32x4 t = *reinterpret_cast<const f32x4*>(a.bias + n);
        v += t;
      }
      if (a.emb) {
        f32x4 t = *reinterpret_cast<const f32x4*>(a.emb + (int64_t)b * a.emb_stride + n);
        v += t;
      }
      if (rg) {
        const T* p = rg + (int64_t)m * a.Cout + n;
        if constexpr (IS_BF16) {
          bf16x4 t = *reinterpret_cast<const bf16x4*>(p);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += (float)t[e];
        } else {
          v += *reinterpret_cast<const f32x4*>(p);
        }
      }
      v *= a.out_scale;
      if (a.y_nchw) {
        int r = m - b * hw;
#pragma unroll
        for (int e = 0; e < 4; ++e) a.y_nchw[((int64_t)b * a.Cout + n + e) * hw + r] = v[e];
      } else {
        T* p = yg + (int64_t)m * a.Cout + n;
        if constexpr (IS_BF16) {
          bf16x4 t;
#pragma unroll
          for (int e = 0; e < 4; ++e) t[e] = (bf16_t)v[e];
          *reinterpret_cast<bf16x4*>(p) = t;
        } else {
          *reinterpret_cast<f32x4*>(p) = v;
        }
      }
    }
  }
}

// y = (ws + bias + emb + res) * scale, one thread per 4 channels
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
    if (a.emb) v += *reinterpret_cast<const f32x4*>(a.emb + (int64_t)b * a.emb_stride + n);
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
struct TileCfg { int bm, bn; };

static TileCfg pick_tile(int M, int Cout) {
  int bn;
  if (Cout % 112 == 0) bn = 112;
  else if (Cout % 128 == 0) bn = 128;
  else if (Cout <= 64) bn = 64;
  else {
    // least padding among the compiled widths
    int best = 128, waste = cdiv(Cout, 128) * 128 - Cout;
    int w112 = cdiv(Cout, 112) * 112 - Cout, w64 = cdiv(Cout, 64) * 64 - Cout;
    if (w112 < waste) { best = 112; waste = w112; }
    if (w64 < waste) { best = 64; waste = w64; }
    bn = best;
  }
  int tiles128 = cdiv(M, 128) * cdiv(Cout, bn);
  int bm = tiles128 >= 256 ? 128 : 64;
  return {bm, bn};
}

static ConvK make_k(const ConvArgs& a, int kps, int splitk) {
  ConvK k{};
  k.x = a.x; k.w = a.w; k.bias = a.bias; k.emb = a.emb; k.res = a.res; k.y = a.y; k.y_nchw = a.y_nchw;
  k.ws = splitk > 1 ? a.splitk_ws : nullptr;
  k.B = a.B; k.Hin = a.Hin; k.Win = a.Win; k.Cin = a.Cin; k.Hout = a.Hout; k.Wout = a.Wout; k.Cout = a.Cout;
  k.KH = a.KH; k.KW = a.KW; k.stride = a.stride; k.pad = a.pad; k.ups = a.ups;
  k.emb_stride = a.emb_stride;
  k.M = a.B * a.Hout * a.Wout;
  k.CPT = cdiv(a.Cin, 32);
  k.NK = k.CPT * a.KH * a.KW;
  k.NS = cdiv(k.NK, kps);
  k.SPS = cdiv(k.NS, splitk);
  k.out_scale = a.out_scale;
  return k;
}

int conv_pick_splitk(const ConvArgs& a, int dt) {
  int M = a.B * a.Hout * a.Wout;
  TileCfg t = pick_tile(M, a.Cout);
  int tiles = cdiv(M, t.bm) * cdiv(a.Cout, t.bn);
  int kps = dt == DT_BF16 ? 2 : 1;
  int NS = cdiv(cdiv(a.Cin, 32) * a.KH * a.KW, kps);
  if (tiles >= 192 || NS < 16) return 1;
  int want = cdiv(384, tiles);
  int maxs = std::max(1, NS / 8);
  return std::max(1, std::min(std::min(want, maxs), 16));
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
    // algorithmic work of this launch: 2*M*N*K flops; operands + output once
    const double Kdim = (double)k.KH * k.KW * k.Cin, Mtot = (double)k.M * batch;
    const double flops = 2.0 * Mtot * k.Cout * Kdim;
    const double bytes = sizeof(T) * ((double)batch * k.B * k.Hin * k.Win * k.Cin + (double)k.Cout * Kdim * (k.w_bstride ? batch : 1) + Mtot * k.Cout);
    static const std::string name = std::string("conv_igemm<") + (sizeof(T) == 2 ? "bf16" : "f32") + "," + std::to_string(BM) + "x" + std::to_string(BN) + ">";
    PROF_SCOPE(name.c_str(), flops, bytes, s);
    hipLaunchKernelGGL(kern, grid, dim3(WM * WN * 64), 2 * stage, s, k);
  }
  HIP_TRY(hipGetLastError());
  return DDMI_OK;
}

template <typename T, int KPS>
static int dispatch_tile(const ConvK& k, TileCfg t, int splitk, int batch, hipStream_t s) {
  if (t.bm == 128 && t.bn == 112) return launch_cfg<T, 128, 112, 4, 1, KPS>(k, splitk, batch, s);
  if (t.bm == 64 && t.bn == 112) return launch_cfg<T, 64, 112, 4, 1, KPS>(k, splitk, batch, s);
  if (t.bm == 128 && t.bn == 128) return launch_cfg<T, 128, 128, 2, 2, KPS>(k, splitk, batch, s);
  if (t.bm == 64 && t.bn == 128) return launch_cfg<T, 64, 128, 2, 2, KPS>(k, splitk, batch, s);
  if (t.bm == 128 && t.bn == 64) return launch_cfg<T, 128, 64, 4, 1, KPS>(k, splitk, batch, s);
  if (t.bm == 64 && t.bn == 64) return launch_cfg<T, 64, 64, 2, 2, KPS>(k, splitk, batch, s);
  return ddmi_fail(DDMI_ERR_INVALID, "conv: no tile configuration");
}

static int check_conv(const ConvArgs& a, int dt) {
  int eps = dt == DT_F32 ? 4 : 8;
  DDMI_REQUIRE(a.x && a.w && (a.y || a.y_nchw), "conv: null tensor");
  DDMI_REQUIRE(a.Cin % eps == 0, "conv: Cin must be a multiple of the 16-byte vector");
  DDMI_REQUIRE(a.Cout % 4 == 0, "conv: Cout must be a multiple of 4");
  DDMI_REQUIRE(!a.emb || a.emb_stride % 4 == 0, "conv: emb_stride must be a multiple of 4");
  DDMI_REQUIRE((a.KH == 1 && a.KW == 1) || (a.KH == 3 && a.KW == 3), "conv: kernel must be 1x1 or 3x3");
  DDMI_REQUIRE(a.stride == 1 || a.stride == 2, "conv: stride must be 1 or 2");
  DDMI_REQUIRE(!(a.ups && a.stride != 1), "conv: fused upsample needs stride 1");
  int hin = a.ups ? 2 * a.Hin : a.Hin, win = a.ups ? 2 * a.Win : a.Win;
  DDMI_REQUIRE(a.Hout == (hin + 2 * a.pad - a.KH) / a.stride + 1 && a.Wout == (win + 2 * a.pad - a.KW) / a.stride + 1,
               "conv: output size mismatch");
  return DDMI_OK;
}

int launch_conv(const ConvArgs& a, int dt, hipStream_t s) {
  DDMI_TRY(check_conv(a, dt));
  int splitk = std::max(1, a.splitk);
  DDMI_REQUIRE(splitk == 1 || a.splitk_ws, "conv: split-K needs a workspace");
  int M = a.B * a.Hout * a.Wout;
  TileCfg t = pick_tile(M, a.Cout);
  if (dt == DT_BF16) {
    ConvK k = make_k(a, 2, splitk);
    DDMI_TRY((dispatch_tile<bf16_t, 2>(k, t, splitk, 1, s)));
  } else {
    ConvK k = make_k(a, 1, splitk);
    DDMI_TRY((dispatch_tile<float, 1>(k, t, splitk, 1, s)));
  }
  if (splitk > 1) DDMI_TRY(launch_splitk_finish(a, dt, s));
  return DDMI_OK;
}

int launch_splitk_finish(const ConvArgs& a, int dt, hipStream_t s) {
  ConvK k = make_k(a, 1, 2);
  k.SPS = a.splitk;  // number of slabs to sum
  int64_t nvec = (int64_t)k.M * k.Cout / 4;
  int blocks = (int)std::min<int64_t>(cdiv64(nvec, 256), 256 * 8);
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
  TileCfg t = pick_tile(M * batch, N);
  if (dt == DT_BF16) {
    ConvK k = make_k(a, 2, 1);
    k.x_bstride = strideA; k.w_bstride = strideB; k.y_bstride = strideC;
    return dispatch_tile<bf16_t, 2>(k, t, 1, batch, s);
  }
  ConvK k = make_k(a, 1, 1);
  k.x_bstride = strideA; k.w_bstride = strideB; k.y_bstride = strideC;
  return dispatch_tile<float, 1>(k, t, 1, batch, s);
}
