// GroupNorm(32 groups) over NHWC activations, fp32 statistics (util.py:198-215 eps 1e-5; autoencoder_unet.py:46-47
// eps 1e-6), optionally fused with SiLU.  HBM-bound: stats pass reads x once, apply pass reads x and writes y.
//   gn_stats : per-(sample, group) sum / sum-of-squares; per-thread fp32 partials over a few dozen pixels, fixed-order
//              fp64 combine per block and across blocks (deterministic; E[x^2]-E[x]^2 keeps ~1e-7 relative accuracy)
//   gn_apply : y = silu?(x * (rstd*gamma) + (beta - mean*rstd*gamma)), per-channel scale/shift staged in LDS
#include <cstdlib>
#include "kernels.h"
#include "prof.h"

#define GN_GROUPS 32

template <typename T> struct Vec16;
template <> struct Vec16<float> {
  static constexpr int N = 4;
  __device__ static void load(const float* p, float* v) {
    float4 a = *reinterpret_cast<const float4*>(p);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
  }
  __device__ static void store(float* p, const float* v) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  }
};
template <> struct Vec16<bf16_t> {
  static constexpr int N = 8;
  __device__ static void load(const bf16_t* p, float* v) {
    bf16x8 a = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)a[i];
  }
  __device__ static void store(bf16_t* p, const float* v) {
    bf16x8 a;
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = (bf16_t)v[i];
    *reinterpret_cast<bf16x8*>(p) = a;
  }
};

// grid (nsplit, B); block = NV * R threads (NV = C / vec channel-vectors per pixel, R pixel lanes).
// Deterministic: per-thread fp32 partials -> LDS [R][C] -> 64 threads sum (r, channel-in-group) in a fixed order in
// fp64 -> part[b][split][group*2 + {sum, sumsq}].  No atomics, no zero-fill; bit-identical from run to run.
#define GN_MAXSPLIT 64
template <typename T>
__global__ void gn_stats_kernel(const T* __restrict__ x, const T* __restrict__ x2, int C1, double* __restrict__ part,
                                int HW, int C, int NV, int R, int seg, int nseg) {
  constexpr int VN = Vec16<T>::N;
  extern __shared__ float red[];  // s1[R][C] | s2[R][C]
  int b = blockIdx.y;
  int tid = threadIdx.x;
  int v = tid % NV, r = tid / NV;
  int per = cdiv(HW, gridDim.x);
  int p0 = blockIdx.x * per, p1 = min(HW, p0 + per);
  if (r < R) {
    float s1[VN], s2[VN];
#pragma unroll
    for (int i = 0; i < VN; ++i) s1[i] = s2[i] = 0.f;
    // the tensor may be the channel concatenation [x | x2] of two tensors read in place (th.cat([h, hs.pop()], 1),
    // openaimodel.py:787): a thread's 16-byte channel vector lies wholly in one of them (C1 is a vector multiple)
    const bool second = x2 != nullptr && v * VN >= C1;
    const int ldc = x2 == nullptr ? C : (second ? C - C1 : C1);
    const T* base = (second ? x2 + (int64_t)v * VN - C1 : x + (int64_t)v * VN) + (int64_t)b * HW * ldc;
    int p = p0 + r;
    // four independent 16-byte loads in flight per thread: the pass is a pure HBM stream and one load per thread per
    // iteration leaves the memory pipes half empty
    for (; p + 3 * R < p1; p += 4 * R) {
      float t[4][VN];
#pragma unroll
      for (int u = 0; u < 4; ++u) Vec16<T>::load(base + (int64_t)(p + u * R) * ldc, t[u]);
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < VN; ++i) {
          s1[i] += t[u][i];
          s2[i] += t[u][i] * t[u][i];
        }
    }
    for (; p < p1; p += R) {
      float t[VN];
      Vec16<T>::load(base + (int64_t)p * ldc, t);
#pragma unroll
      for (int i = 0; i < VN; ++i) {
        s1[i] += t[i];
        s2[i] += t[i] * t[i];
      }
    }
#pragma unroll
    for (int i = 0; i < VN; ++i) {
      red[r * C + v * VN + i] = s1[i];
      red[(R + r) * C + v * VN + i] = s2[i];
    }
  }
  __syncthreads();
  if (tid < GN_GROUPS * 2) {
    int g = tid >> 1, which = tid & 1;
    int cpg = C / GN_GROUPS;
    double acc = 0.0;
    for (int rr = 0; rr < R; ++rr)
      for (int c = g * cpg; c < (g + 1) * cpg; ++c) acc += (double)red[(which * R + rr) * C + c];
    part[(((int64_t)b * nseg + seg) * GN_MAXSPLIT + blockIdx.x) * (GN_GROUPS * 2) + tid] = acc;
  }
}

// grid (nsplit, B); block = NV * R threads like the stats kernel: every thread owns ONE 16-byte channel vector for all
// its pixels, so its scale/shift live in registers (no per-element index math, no LDS in the streaming loop).
struct GnSegs { int nseg, nsplit[3], total_hw; };
// Statistics straight from the producers' epilogues (ConvArgs::chstat: per row block of `rows` pixels and channel (sum, sumsq)
// in fp32).  Round 2 reduced them in a launch of their own (gn_chstat_finish_kernel: 61 launches of ~35 us per UNet evaluation =
// 1.5 % of a step, mostly launch latency: 59 MB at most); here every workgroup of the apply pass reduces its SAMPLE's blocks itself
// before it starts streaming (64 blocks x C channels = 115 KB at 64x64: ~3 us of a > 100 us workgroup; the splits of a sample
// repeat it - L2 hits).  Fixed order, fp64: deterministic.
struct GnChstat { const float* cs1; const float* cs2; int rows1, rows2; };
template <typename T, bool CHSTAT>
__global__ void gn_apply_kernel(const T* __restrict__ x, const T* __restrict__ x2, int C1, T* __restrict__ y,
                                const double* __restrict__ part,
                                GnSegs sg, const float* __restrict__ gamma, const float* __restrict__ beta,
                                float eps, int silu, int HW, int C, int NV, int R, GnChstat ch, int x3_out, int frag_out) {
  constexpr int VN = Vec16<T>::N;
  __shared__ float mr[GN_GROUPS * 2];  // mean, rstd per group
  int b = blockIdx.y;
  int tid = threadIdx.x;
  int cpg = C / GN_GROUPS;
  if constexpr (CHSTAT) {
    extern __shared__ double cred[];   // [2 C]: (sum, sumsq) per channel over the sample
    for (int c = tid; c < C; c += blockDim.x) {   // one thread per channel: (sum, sumsq) pairs, eight blocks in flight
      const bool second = ch.cs2 != nullptr && c >= C1;
      const int ldc = ch.cs2 == nullptr ? C : (second ? C - C1 : C1), nblk = HW / (second ? ch.rows2 : ch.rows1);
      const float2* p = reinterpret_cast<const float2*>((second ? ch.cs2 : ch.cs1) + ((int64_t)b * nblk * ldc) * 2) + (c - (second ? C1 : 0));
      double a1 = 0.0, a2 = 0.0;
      int k = 0;
      for (; k + 7 < nblk; k += 8) {   // the adds stay in block order (fixed order: deterministic)
        float2 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = p[(int64_t)(k + u) * ldc];
#pragma unroll
        for (int u = 0; u < 8; ++u) { a1 += (double)v[u].x; a2 += (double)v[u].y; }
      }
      for (; k < nblk; ++k) {
        const float2 v = p[(int64_t)k * ldc];
        a1 += (double)v.x; a2 += (double)v.y;
      }
      cred[2 * c] = a1;
      cred[2 * c + 1] = a2;
    }
    __syncthreads();
    if (tid < GN_GROUPS) {
      double s1 = 0.0, s2 = 0.0;
      for (int c = tid * cpg; c < (tid + 1) * cpg; ++c) { s1 += cred[2 * c]; s2 += cred[2 * c + 1]; }
      double cnt = (double)HW * cpg;
      double m = s1 / cnt;
      double var = s2 / cnt - m * m;
      if (var < 0) var = 0;
      mr[tid * 2] = (float)m;
      mr[tid * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
    }
  } else if (tid < GN_GROUPS) {
    double s1 = 0.0, s2 = 0.0;
    // statistics run over every segment of the sample (the video path normalises xy | xt | yt tokens jointly)
    for (int sgi = 0; sgi < sg.nseg; ++sgi) {
      const double* p = part + ((int64_t)b * sg.nseg + sgi) * GN_MAXSPLIT * (GN_GROUPS * 2) + tid * 2;
      for (int sidx = 0; sidx < sg.nsplit[sgi]; ++sidx) {
        s1 += p[(int64_t)sidx * (GN_GROUPS * 2)];
        s2 += p[(int64_t)sidx * (GN_GROUPS * 2) + 1];
      }
    }
    double cnt = (double)sg.total_hw * cpg;
    double m = s1 / cnt;
    double var = s2 / cnt - m * m;
    if (var < 0) var = 0;
    mr[tid * 2] = (float)m;
    mr[tid * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
  }
  __syncthreads();
  if constexpr (sizeof(T) == 2) {
    if (frag_out) {
      // Output in MFMA B-fragment order for the fused attention block (k_attnblock.hip): [16-token block][32-channel k-step][lane = 16 (c >> 3
      // & 3) + (token & 15)][8 channels], so that a wave of that kernel loads a whole fragment as ONE contiguous KiB (from the
      // row-major tensor every 16-lane group of its loads touched 16 cache lines: 12 B/clk per CU, measured).
      // A wave takes [16 tokens][64 channels] items: two loads of 8 whole 128-byte row pieces (lane = 8 (token & 7) + 16-byte piece), the
      // affine transform on the lane's fixed 8 channels, then a transpose THROUGH LDS (padded: the 8 lanes of a ds_write_b128 group
      // hit 32 different banks) into the two fragments, which leave as two contiguous-KiB stores.  (Round 4 first stored the
      // fragments' scattered 16-byte pieces straight from the row-major lane map - the pass took 2x - and then loaded with the
      // fragment's lane map, 16 cache lines per 16-lane group: +47 %.)
      extern __shared__ double gn_dyn[];
      constexpr int FRAG_PAD = 1088, ITEM_BYTES = 2 * FRAG_PAD, UF = 4;   // per k-step: 4 x (256 + 16) bytes
      const int lane = tid & 63, wave = tid >> 6, NW = blockDim.x >> 6;
      unsigned char* const scr = reinterpret_cast<unsigned char*>(gn_dyn) + (CHSTAT ? (size_t)2 * C * sizeof(double) : 0) + (size_t)wave * UF * ITEM_BYTES;
      const int KP = C >> 6, TB = HW >> 4;
      const int per_tb = cdiv(TB, gridDim.x), tb0 = blockIdx.x * per_tb, tb1 = min(TB, tb0 + per_tb);
      const int nchunk = (tb1 - tb0 + UF - 1) / UF;
      const int piece = lane & 7, trow = lane >> 3;
      // LDS write position of (token t, piece): fragment piece >> 2, lane group piece & 3, row t
      const int wr0 = (piece >> 2) * FRAG_PAD + (piece & 3) * 272 + trow * 16;
      const int rd0 = (lane >> 4) * 272 + (lane & 15) * 16;
      typedef __attribute__((ext_vector_type(4))) unsigned int raw16;
      for (int item = wave; item < nchunk * KP; item += NW) {   // item = (UF token blocks, one 64-channel pair of k-steps)
        const int kp = item % KP, t0 = tb0 + (item / KP) * UF;
        const int c0 = 64 * kp + 8 * piece;
        float sc[8], sh[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int g = (c0 + i) / cpg;
          const float s = mr[g * 2 + 1] * gamma[c0 + i];
          sc[i] = s;
          sh[i] = beta[c0 + i] - mr[g * 2] * s;
        }
        const T* xb = x + ((int64_t)b * HW + trow) * C + c0;
        raw16 raw[UF][2];
#pragma unroll
        for (int u = 0; u < UF; ++u)
          if (t0 + u < tb1) {
            raw[u][0] = __builtin_nontemporal_load(reinterpret_cast<const raw16*>(xb + (int64_t)((t0 + u) * 16) * C));
            raw[u][1] = __builtin_nontemporal_load(reinterpret_cast<const raw16*>(xb + (int64_t)((t0 + u) * 16 + 8) * C));
          }
#pragma unroll
        for (int u = 0; u < UF; ++u) {
          if (t0 + u >= tb1) continue;
#pragma unroll
          for (int hlf = 0; hlf < 2; ++hlf) {
            float t[8];
            Vec16<T>::load(reinterpret_cast<const T*>(&raw[u][hlf]), t);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              float q = t[j] * sc[j] + sh[j];
              if (silu) q = silu_fast(q);
              t[j] = q;
            }
            Vec16<T>::store(reinterpret_cast<T*>(scr + u * ITEM_BYTES + wr0 + hlf * 128), t);
          }
        }
        // same wave wrote and reads: LDS operations of a wave complete in order, no barrier
#pragma unroll
        for (int u = 0; u < UF; ++u) {
          if (t0 + u >= tb1) continue;
          T* yb = y + ((((int64_t)b * TB + t0 + u) * (C >> 5) + 2 * kp) * 64 + lane) * 8;
#pragma unroll
          for (int f = 0; f < 2; ++f) {
            const raw16 v = *reinterpret_cast<const raw16*>(scr + u * ITEM_BYTES + f * FRAG_PAD + rd0);
            *reinterpret_cast<raw16*>(yb + f * 512) = v;
          }
        }
      }
      return;
    }
  }
  int v = tid % NV, r = tid / NV;
  if (r >= R) return;
  float sc[VN], sh[VN];
#pragma unroll
  for (int i = 0; i < VN; ++i) {
    int c = v * VN + i;
    int g = c / cpg;
    float s = mr[g * 2 + 1] * gamma[c];
    sc[i] = s;
    sh[i] = beta[c] - mr[g * 2] * s;
  }
  int per = cdiv(HW, gridDim.x);
  int p0 = blockIdx.x * per, p1 = min(HW, p0 + per);
  const bool second = x2 != nullptr && v * VN >= C1;  // dual-source input, see gn_stats_kernel
  const int ldc = x2 == nullptr ? C : (second ? C - C1 : C1);
  const T* xb = (second ? x2 + (int64_t)v * VN - C1 : x + (int64_t)v * VN) + (int64_t)b * HW * ldc;
  T* yb = y + (int64_t)b * HW * C + (int64_t)v * VN;
  // x3_out (bf16x3 numerics, fp32 input): the output is written as the [xh | xl | xh] bf16 image the consumer convolution reads
  // (k_elementwise.hip split_x3_kernel) - 6 bytes per element instead of an fp32 tensor plus a split pass over it
  bf16_t* y3 = reinterpret_cast<bf16_t*>(y) + (int64_t)b * HW * 3 * C + (int64_t)v * VN;
  auto store_out = [&](int pix, const float* t) {
    if (sizeof(T) == 4 && x3_out) {
      bf16x4 hi, lo;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        hi[e] = (bf16_t)t[e];
        lo[e] = (bf16_t)(t[e] - (float)hi[e]);
      }
      bf16_t* o = y3 + (int64_t)pix * 3 * C;
      *reinterpret_cast<bf16x4*>(o) = hi;
      *reinterpret_cast<bf16x4*>(o + C) = lo;
      *reinterpret_cast<bf16x4*>(o + 2 * C) = hi;
    } else {
      Vec16<T>::store(yb + (int64_t)pix * C, t);
    }
  };
  int p = p0 + r;
  // eight 16-byte loads in flight per thread: at one workgroup of 256 threads per (split, sample) and 4 workgroups per CU, four
  // loads left ~64 KB in flight per CU - about one HBM latency x the CU's share of the bandwidth, no slack (5.0 TB/s measured)
  typedef __attribute__((ext_vector_type(4))) unsigned int raw16;
  for (; p + 7 * R < p1; p += 8 * R) {
    raw16 raw[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) raw[u] = __builtin_nontemporal_load(reinterpret_cast<const raw16*>(xb + (int64_t)(p + u * R) * ldc));
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      float t[VN];
      Vec16<T>::load(reinterpret_cast<const T*>(&raw[u]), t);
#pragma unroll
      for (int j = 0; j < VN; ++j) {
        float q = t[j] * sc[j] + sh[j];
        if (silu) q = sizeof(T) == 2 ? silu_fast(q) : silu_precise(q);
        t[j] = q;
      }
      store_out(p + u * R, t);
    }
  }
  for (; p + 3 * R < p1; p += 4 * R) {  // four loads in flight per thread (see gn_stats_kernel)
    float t[4][VN];
#pragma unroll
    for (int u = 0; u < 4; ++u) Vec16<T>::load(xb + (int64_t)(p + u * R) * ldc, t[u]);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int j = 0; j < VN; ++j) {
        float q = t[u][j] * sc[j] + sh[j];
        if (silu) q = sizeof(T) == 2 ? silu_fast(q) : silu_precise(q);
        t[u][j] = q;
      }
      store_out(p + u * R, t[u]);
    }
  }
  for (; p < p1; p += R) {
    float t[VN];
    Vec16<T>::load(xb + (int64_t)p * ldc, t);
#pragma unroll
    for (int j = 0; j < VN; ++j) {
      float q = t[j] * sc[j] + sh[j];
      if (silu) q = sizeof(T) == 2 ? silu_fast(q) : silu_precise(q);
      t[j] = q;
    }
    store_out(p, t);
  }
}

static int gn_splits_of(int B, int HW);
// Statistics from the producers' epilogues (ConvArgs::chstat): per row block of `rows` pixels, per channel (sum, sumsq) in fp32.
// grid (nsplit, B) like gn_stats_kernel: block (sidx, b) adds its share of the sample's row blocks per channel (fp64, block
// order), then the channels of each group, and writes split `sidx` of the layout gn_apply_kernel reads.  The input may be the
// channel concatenation of two tensors, each with its own buffer and block height.
__global__ void __launch_bounds__(256) gn_chstat_finish_kernel(const float* __restrict__ cs1, int rows1, const float* __restrict__ cs2, int rows2,
                                                               int C1, int C, int HW, double* __restrict__ part) {
  extern __shared__ double cred[];   // [2 C]: (sum, sumsq) per channel
  const int b = blockIdx.y, sidx = blockIdx.x, ns = gridDim.x, tid = threadIdx.x;
  for (int i = tid; i < 2 * C; i += 256) {
    const int c = i >> 1;
    const bool second = cs2 != nullptr && c >= C1;
    const int ldc = cs2 == nullptr ? C : (second ? C - C1 : C1), nblk = HW / (second ? rows2 : rows1);
    const int k0 = (int)((int64_t)sidx * nblk / ns), k1 = (int)((int64_t)(sidx + 1) * nblk / ns);
    const float* p = (second ? cs2 : cs1) + ((int64_t)b * nblk * ldc) * 2 + (i - (second ? 2 * C1 : 0));
    double acc = 0.0;
    for (int k = k0; k < k1; ++k) acc += (double)p[(int64_t)k * ldc * 2];
    cred[i] = acc;
  }
  __syncthreads();
  if (tid < GN_GROUPS * 2) {
    const int g = tid >> 1, which = tid & 1, cpg = C / GN_GROUPS;
    double acc = 0.0;
    for (int c = g * cpg; c < (g + 1) * cpg; ++c) acc += cred[2 * c + which];
    part[((int64_t)b * GN_MAXSPLIT + sidx) * (GN_GROUPS * 2) + tid] = acc;
  }
}

int launch_gn_chstat_finish(const float* cs1, int rows1, const float* cs2, int rows2, int C1, double* stats, int B, int HW, int C,
                            hipStream_t s) {
  DDMI_REQUIRE(cs1 && rows1 > 0 && HW % rows1 == 0 && (cs2 == nullptr || (rows2 > 0 && HW % rows2 == 0 && C1 > 0 && C1 < C)),
               "groupnorm: bad channel-statistics buffers");
  DDMI_REQUIRE(C % GN_GROUPS == 0, "groupnorm: C must be a multiple of 32");
  PROF_SCOPE("gn_stats", 0, 8.0 * B * C * (HW / rows1), s);
  hipLaunchKernelGGL(gn_chstat_finish_kernel, dim3(gn_splits_of(B, HW), B), dim3(256), (size_t)2 * C * sizeof(double), s, cs1, rows1, cs2, rows2,
                     C1, C, HW, stats);
  HIP_TRY(hipGetLastError());
  return DDMI_OK;
}

static int gn_splits(int B, int HW) {
  // enough blocks to fill 256 CUs a few times over, at least 16 pixels per block
  static const int target = getenv("DDMI_GN_BLOCKS") ? atoi(getenv("DDMI_GN_BLOCKS")) : 256 * 4;   // diagnostics: workgroups per launch to aim for
  int want = std::max(1, target / std::max(B, 1));
  return std::max(1, std::min(std::min(want, HW / 16), GN_MAXSPLIT));
}

int gn_nsplit(int B, int HW) { return gn_splits(B, HW); }
static int gn_splits_of(int B, int HW) { return gn_splits(B, HW); }

int launch_gn_stats(const void* x, int dt, double* stats, int B, int HW, int C, hipStream_t s, int seg, int nseg,
                    const void* x2, int C1) {
  DDMI_REQUIRE(nseg >= 1 && nseg <= 3 && seg >= 0 && seg < nseg, "groupnorm: bad segment index");
  int vn = dt == DT_F32 ? 4 : 8;
  DDMI_REQUIRE(C % GN_GROUPS == 0, "groupnorm: C must be a multiple of 32");
  DDMI_REQUIRE(C % vn == 0, "groupnorm: C must be a multiple of the 16-byte vector");
  DDMI_REQUIRE(x2 == nullptr || (C1 > 0 && C1 < C && C1 % vn == 0), "groupnorm: the concat boundary must fall on a 16-byte vector");
  int NV = C / vn;
  DDMI_REQUIRE(NV <= 1024, "groupnorm: too many channels");
  int R = std::max(1, 256 / NV);
  int threads = std::max(64, cdiv(NV * R, 64) * 64);
  dim3 g(gn_splits(B, HW), B);
  size_t lds = (size_t)2 * R * C * sizeof(float);
  PROF_SCOPE("gn_stats", 0, (double)B * HW * C * dt_size(dt), s);
  if (dt == DT_F32)
    hipLaunchKernelGGL(gn_stats_kernel<float>, g, dim3(threads), lds, s, (const float*)x, (const float*)x2, C1, stats, HW, C,
                       NV, R, seg, nseg);
  else
    hipLaunchKernelGGL(gn_stats_kernel<bf16_t>, g, dim3(threads), lds, s, (const bf16_t*)x, (const bf16_t*)x2, C1, stats, HW,
                       C, NV, R, seg, nseg);
  HIP_TRY(hipGetLastError());
  return DDMI_OK;
}

static bool gn_frag_out_ok(int dt, int B, int HW, int C) { return dt == DT_BF16 && C % 64 == 0 && HW % 16 == 0; }
static size_t gn_frag_scratch_bytes(int threads) { return (size_t)(threads / 64) * 4 * 2 * 1088; }   // per wave: UF items of two padded fragments

int launch_gn_apply(const void* x, void* y, int dt, const double* stats, const float* gamma, const float* beta,
                    float eps, int silu, int B, int HW, int C, hipStream_t s, const int* seg_hw, int nseg, const void* x2,
                    int C1, int x3_out, int frag_out) {
  DDMI_REQUIRE(!x3_out || dt == DT_F32, "groupnorm: the [xh | xl | xh] output belongs to the fp32 (bf16x3) networks");
  DDMI_REQUIRE(!frag_out || (gn_frag_out_ok(dt, B, HW, C) && x2 == nullptr && nseg <= 1),
               "groupnorm: fragment-order output needs bf16, one source, C % 32 == 0 and whole 16-token blocks");
  int vn = dt == DT_F32 ? 4 : 8;
  int NV = C / vn;
  int R = std::max(1, 256 / NV);
  int threads = std::max(64, cdiv(NV * R, 64) * 64);
  int ns = gn_splits(B, HW);
  dim3 g(ns, B);
  GnSegs sg{};
  if (seg_hw == nullptr || nseg <= 1) {
    sg.nseg = 1; sg.nsplit[0] = ns; sg.total_hw = HW;
  } else {
    DDMI_REQUIRE(nseg <= 3, "groupnorm: at most three segments");
    sg.nseg = nseg;
    for (int i = 0; i < nseg; ++i) {
      sg.nsplit[i] = gn_splits(B, seg_hw[i]);  // what launch_gn_stats used for that segment (same B)
      sg.total_hw += seg_hw[i];
    }
  }
  PROF_SCOPE("gn_apply", 0, 2.0 * B * HW * C * dt_size(dt), s);
  GnChstat none{};
  if (dt == DT_F32)
    hipLaunchKernelGGL((gn_apply_kernel<float, false>), g, dim3(threads), 0, s, (const float*)x, (const float*)x2, C1, (float*)y, stats, sg, gamma,
                       beta, eps, silu, HW, C, NV, R, none, x3_out, 0);
  else
    hipLaunchKernelGGL((gn_apply_kernel<bf16_t, false>), g, dim3(threads), frag_out ? gn_frag_scratch_bytes(threads) : 0, s, (const bf16_t*)x,
                       (const bf16_t*)x2, C1, (bf16_t*)y, stats, sg, gamma, beta, eps, silu, HW, C, NV, R, none, 0, frag_out);
  HIP_TRY(hipGetLastError());
  return DDMI_OK;
}

// GroupNorm apply with the statistics taken from the producers' channel-statistics buffers (bf16 networks; see gn_apply_kernel)
int launch_gn_apply_chstat(const void* x, void* y, const float* cs1, int rows1, const float* cs2, int rows2, const float* gamma,
                           const float* beta, float eps, int silu, int B, int HW, int C, hipStream_t s, const void* x2, int C1, int frag_out) {
  DDMI_REQUIRE(!frag_out || (gn_frag_out_ok(DT_BF16, B, HW, C) && x2 == nullptr),
               "groupnorm: fragment-order output needs one source, C % 32 == 0 and whole 16-token blocks");
  DDMI_REQUIRE(cs1 && rows1 > 0 && HW % rows1 == 0 && (cs2 == nullptr || (rows2 > 0 && HW % rows2 == 0 && C1 > 0 && C1 < C)),
               "groupnorm: bad channel-statistics buffers");
  DDMI_REQUIRE(C % GN_GROUPS == 0 && C % 8 == 0, "groupnorm: C must be a multiple of 32");
  const int NV = C / 8, R = std::max(1, 256 / NV);
  const int threads = std::max(64, cdiv(NV * R, 64) * 64);
  dim3 g(gn_splits(B, HW), B);
  GnSegs sg{};
  sg.nseg = 1; sg.nsplit[0] = (int)g.x; sg.total_hw = HW;
  GnChstat ch{cs1, cs2, rows1, rows2};
  PROF_SCOPE("gn_apply", 0, 2.0 * B * HW * C * 2 + 8.0 * B * C * (HW / rows1), s);
  hipLaunchKernelGGL((gn_apply_kernel<bf16_t, true>), g, dim3(threads), (size_t)2 * C * sizeof(double) + (frag_out ? gn_frag_scratch_bytes(threads) : 0), s,
                     (const bf16_t*)x, (const bf16_t*)x2, C1,
                     (bf16_t*)y, (const double*)nullptr, sg, gamma, beta, eps, silu, HW, C, NV, R, ch, 0, frag_out);
  HIP_TRY(hipGetLastError());
  return DDMI_OK;
}

// =====================================================================================================
// Small-batch GroupNorm: statistics + apply in ONE launch (kernels.h GnSmallArgs).  grid (32 groups, B); a workgroup owns the HW x cpg
// elements of one (sample, group): element e = pixel * cpg + c, consecutive lanes on consecutive elements (runs of cpg bf16 = 14 .. 168
// bytes per pixel: not the 16-byte vectors of the streaming passes above - at these sizes, a few MB per launch and L2-resident, the
// launch floor is what is being removed, not bandwidth).  The thread's (at most NE) elements stay in REGISTERS between the statistics
// and the apply: every load of the slice is issued back to back (the loop is fully unrolled - a rolled loop waits out one L2 round
// trip per element: 30-100 us per launch as first written) and nothing is read twice.  All accesses are buffer instructions with
// 32-bit element offsets (one VGPR per address: 32 values + 32 offsets per thread fit the 128 VGPRs of a 1024-thread workgroup; 64-bit
// pointers spilled 439).  Statistics: per-thread fp32 sums of (v - K), K = the slice's first element (shifted data: no cancellation
// when |mean| >> sigma), combined in fp64 in a fixed order (deterministic).
// Measured and NOT kept (round 6): the same kernel reading the producing convolution's raw split-K slabs (it then WAS the splitk_finish
// pass): with 32 workgroups per sample each slab walk is one dependent round trip and the 28-byte runs of a group pull whole 128-byte
// lines - 55 us per launch at 64 x 64 where the coalesced finish launch + this kernel take 5 + 10.
// =====================================================================================================
struct GnSmallK {
  GnSmallArgs a;
  int cpg;
};

template <int NE, bool CAT>
__global__ void __launch_bounds__(1024) gn_small_kernel(GnSmallK k) {
  __shared__ double red[32];
  __shared__ float mr[2];
  __shared__ float ss[2 * 192];
  const GnSmallArgs& a = k.a;
  const int g = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, T = blockDim.x;
  const int cpg = k.cpg, HW = a.HW, C = a.C, n = HW * cpg;
  const int C1 = CAT ? a.C1 : C, ld2 = C - C1;
  const int c0 = g * cpg;
  const int row0 = b * HW;
  // one descriptor per tensor, based at THIS SAMPLE (offsets stay small)
  auto rs = [](const void* p) { return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, 0x7FFFFFFF, 0x00020000); };
  const auto x_r = rs((const bf16_t*)a.x + (int64_t)row0 * C1);
  const auto x2_r = rs(CAT ? (const bf16_t*)a.x2 + (int64_t)row0 * ld2 : nullptr);
  const auto y_r = rs((bf16_t*)a.y + (int64_t)row0 * C);
  auto ld16 = [](decltype(x_r) r, int elem) -> float {
    const unsigned short u = __builtin_amdgcn_raw_buffer_load_b16(r, elem * 2, 0, 0);
    return __uint_as_float((unsigned)u << 16);
  };
  auto st16 = [](decltype(x_r) r, int elem, float v) {
    const bf16_t h = (bf16_t)v;
    __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, h), r, elem * 2, 0, 0);
  };
  auto value = [&](int pix, int c) -> float {   // element (pixel, group channel c) of the sample
    const int ca = c0 + c;
    if (CAT && ca >= C1) return ld16(x2_r, pix * ld2 + (ca - C1));
    return ld16(x_r, pix * C1 + ca);
  };
  const int dpix = T / cpg, dc = T - dpix * cpg;   // element stride T in (pixel, channel) steps
  const int pix_t = tid / cpg, c_t = tid - pix_t * cpg;
  float v[NE];
  int off[NE];  // sample-relative element offset pixel * C + channel of v[i] in the C-wide output, -1 past the slice
  {
    int pix = pix_t, c = c_t;
#pragma unroll
    for (int i = 0; i < NE; ++i) {
      const bool in = tid + i * T < n;
      off[i] = in ? pix * C + c0 + c : -1;
      v[i] = in ? value(pix, c) : 0.f;
      pix += dpix; c += dc;
      if (c >= cpg) { c -= cpg; ++pix; }
    }
  }
  const float K = value(0, 0);   // the shift of the statistics
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < NE; ++i) {
    const float d = off[i] >= 0 ? v[i] - K : 0.f;
    s1 += d;
    s2 = fmaf(d, d, s2);
  }
  {
    double d1 = (double)s1, d2 = (double)s2;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
      d1 += __shfl_xor(d1, o);
      d2 += __shfl_xor(d2, o);
    }
    if ((tid & 63) == 0) { red[tid >> 6] = d1; red[16 + (tid >> 6)] = d2; }
  }
  __syncthreads();
  if (tid == 0) {
    double d1 = 0.0, d2 = 0.0;
    for (int w = 0; w < ((T + 63) >> 6); ++w) { d1 += red[w]; d2 += red[16 + w]; }   // wave order: deterministic
    const double m = d1 / n;
    double var = d2 / n - m * m;
    if (var < 0) var = 0;
    mr[0] = (float)((double)K + m);
    mr[1] = (float)(1.0 / sqrt(var + (double)a.eps));
  }
  __syncthreads();
  if (tid < cpg) {
    const float sc = mr[1] * a.gamma[c0 + tid];
    ss[2 * tid] = sc;
    ss[2 * tid + 1] = a.beta[c0 + tid] - mr[0] * sc;
  }
  __syncthreads();
  // ---- normalise [+ SiLU], round once, store
  int c = c_t;
#pragma unroll
  for (int i = 0; i < NE; ++i) {
    if (off[i] >= 0) {
      float q = fmaf(v[i], ss[2 * c], ss[2 * c + 1]);
      if (a.silu) q = silu_fast(q);
      st16(y_r, off[i], q);
    }
    c += dc;
    if (c >= cpg) c -= cpg;
  }
}

// a (sample, group) slice stays in the registers of one 1024-thread workgroup: 32 elements per thread (16 for a two-source input: 32 spill)
bool gn_small_ok(int B, int HW, int C, bool two_sources) {
  if (C % GN_GROUPS || B < 1 || HW < 1) return false;
  const int cpg = C / GN_GROUPS;
  return cpg <= 192 && (int64_t)HW * cpg <= (two_sources ? 16384 : 32768) && (int64_t)B * GN_GROUPS <= 1024 &&
         (int64_t)B * HW * C < ((int64_t)1 << 30);
}

int launch_gn_small(const GnSmallArgs& a, hipStream_t s) {
  DDMI_REQUIRE(a.x && a.y && a.gamma && a.beta && gn_small_ok(a.B, a.HW, a.C, a.x2 != nullptr), "groupnorm (small batch): shape not served");
  DDMI_REQUIRE(!a.x2 || (a.C1 > 0 && a.C1 < a.C), "groupnorm (small batch): bad concat boundary");
  GnSmallK k{a, a.C / GN_GROUPS};
  const int n = a.HW * k.cpg;
  const int threads = std::min(1024, std::max(64, cdiv(n, 64) * 64));
  const int per = cdiv(n, threads);
  PROF_SCOPE("gn_small", 0, (double)a.B * a.HW * a.C * 4.0, s);
  const dim3 grid(GN_GROUPS, a.B), block(threads);
  if (a.x2) {
    if (per <= 4) hipLaunchKernelGGL((gn_small_kernel<4, true>), grid, block, 0, s, k);
    else if (per <= 8) hipLaunchKernelGGL((gn_small_kernel<8, true>), grid, block, 0, s, k);
    else hipLaunchKernelGGL((gn_small_kernel<16, true>), grid, block, 0, s, k);
  } else {
    if (per <= 4) hipLaunchKernelGGL((gn_small_kernel<4, false>), grid, block, 0, s, k);
    else if (per <= 8) hipLaunchKernelGGL((gn_small_kernel<8, false>), grid, block, 0, s, k);
    else if (per <= 16) hipLaunchKernelGGL((gn_small_kernel<16, false>), grid, block, 0, s, k);
    else hipLaunchKernelGGL((gn_small_kernel<32, false>), grid, block, 0, s, k);
  }
  HIP_TRY(hipGetLastError());
  return DDMI_OK;
}
