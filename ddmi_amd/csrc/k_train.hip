// Kernels of the LDM training step (tools/ldm/image.py:136-178, diffusion/ddpm.py:383-454): what the
// backward pass needs beyond the forward kernels.  First slice: every op is correct in both numerics modes and MFMA-backed
// where the FLOPs are (both convolution gradients run on the forward implicit-GEMM kernels of k_conv.hip); the HBM-bound
// helpers and the attention / GroupNorm gradients are plain kernels, not yet tuned against their rooflines.
//
//   conv  y = conv(up?(x), W) + b + emb[b] + res
//     dgrad : dX = conv(dY', W^T flipped) - the forward kernel on re-packed weights [Cin][k*k flipped][Cout];
//             stride 2: dY' = dY with zeros inserted (the gradient of a strided gather), nearest-x2 input: 2x2 sum-pool
//     wgrad : dW[co][tap, ci] = sum_p dY[p][co] X_col[p][tap, ci] = the forward kernel as a plain GEMM over the pixel
//             axis (dY^T and the transposed im2col image are both pixel-contiguous), split-K in fp32 slabs
//     dbias / demb : per-(sample, channel) sums of dY
//   GroupNorm(32)+SiLU, QKVAttentionLegacy, the small time-embedding linears: reference formulas, fp32 math
//   q_sample, mixed-prediction MSE loss + its gradient, AdamW
#include "kernels.h"
#include "prof.h"

#define TR_GROUPS 32

template <typename T> __device__ __forceinline__ float tr_ld(const T* p) { return to_f32(*p); }
template <typename T> __device__ __forceinline__ void tr_st(T* p, float v) { *p = from_f32<T>(v); }
static inline int tr_blocks(int64_t n) { return (int)std::min<int64_t>(cdiv64(n, 256), 256 * 16); }

// ---------------------------------------------------------------------------------------------------------------------
// convolution gradients: layout helpers
// ---------------------------------------------------------------------------------------------------------------------
// master fp32 [Co][Ci][k][k] -> packed forward-kernel layout [Ci][k*k][Co] of the TRANSPOSED, spatially flipped filter
template <typename T>
__global__ void pack_dgrad_weight_kernel(const float* __restrict__ w, T* __restrict__ out, int Co, int Ci, int k) {
  const int kk = k * k;
  const int64_t n = (int64_t)Co * Ci * kk;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int co = (int)(i % Co);
    const int tp = (int)((i / Co) % kk);
    const int ci = (int)(i / ((int64_t)Co * kk));
    const int kh = k - 1 - tp / k, kw = k - 1 - tp % k;
    out[i] = from_f32<T>(w[(((int64_t)co * Ci + ci) * k + kh) * k + kw]);
  }
}
// the same through an LDS tile (this runs for every convolution of every training step: 2 GB of master weights): a block owns
// 32 co x TCI ci x kk taps; the source rows (TCI * kk contiguous floats per co) are read in whole runs, the destination is
// written as 32 consecutive co per (ci, tap).  TCI * kk <= 72: grid (ceil(Ci / TCI), ceil(Co / 32)) - small tiles on purpose,
// a 224 x 224 x 9 filter must still spread over the chip (196 blocks; 128-co tiles gave it 56).
template <typename T>
__global__ void __launch_bounds__(256) pack_dgrad_weight_tiled_kernel(const float* __restrict__ w, T* __restrict__ out, int Co, int Ci, int k,
                                                                      int TCI) {
  __shared__ float tile[32][73];
  const int kk = k * k, run = TCI * kk;                  // contiguous floats of one co row inside the tile
  const int ci0 = blockIdx.x * TCI, co0 = blockIdx.y * 32;
  const int nci = min(TCI, Ci - ci0), len = nci * kk;
  for (int i = threadIdx.x; i < 32 * run; i += 256) {
    const int r = i / run, j = i - r * run;
    if (co0 + r < Co && j < len) tile[r][j] = w[((int64_t)(co0 + r) * Ci + ci0) * kk + j];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < run * 32; i += 256) {
    const int j = i >> 5, r = i & 31;                    // j = (ci, destination tap), r = co
    const int ci = j / kk, tp = j - ci * kk;
    if (co0 + r < Co && ci < nci) out[((int64_t)(ci0 + ci) * kk + tp) * Co + co0 + r] = from_f32<T>(tile[r][ci * kk + (kk - 1 - tp)]);
  }
}
// forward layout [Co][k*k][Ci] (launch_pack_conv_weight) through LDS: a block owns one co and up to 256 ci - the source run
// (256 * kk floats) is contiguous, the destination is kk runs of 256 ci
template <typename T>
__global__ void __launch_bounds__(256) pack_fwd_weight_tiled_kernel(const float* __restrict__ w, T* __restrict__ out, int Co, int Ci, int kk) {
  __shared__ float tile[256 * 9 + 8];
  const int co = blockIdx.y, ci0 = blockIdx.x * 256, nci = min(256, Ci - ci0), len = nci * kk;
  const float* src = w + ((int64_t)co * Ci + ci0) * kk;
  for (int i = threadIdx.x; i < len; i += 256) tile[i] = src[i];
  __syncthreads();
  for (int i = threadIdx.x; i < len; i += 256) {
    const int tp = i / nci, ci = i - tp * nci;
    out[((int64_t)co * kk + tp) * Ci + ci0 + ci] = from_f32<T>(tile[ci * kk + tp]);
  }
}

// u[b][2i][2j][c] = dy[b][i][j][c], 0 elsewhere  (u is [B][2Ho][2Wo][C])
template <typename T>
__global__ void zero_insert2x_kernel(const T* __restrict__ dy, T* __restrict__ u, int B, int Ho, int Wo, int C) {
  const int64_t n = (int64_t)B * 4 * Ho * Wo * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const int64_t pix = i / C;
    const int w = (int)(pix % (2 * Wo)), h = (int)((pix / (2 * Wo)) % (2 * Ho));
    const int b = (int)(pix / ((int64_t)4 * Ho * Wo));
    float v = 0.f;
    if (!(h & 1) && !(w & 1)) v = tr_ld(dy + (((int64_t)b * Ho + (h >> 1)) * Wo + (w >> 1)) * C + c);
    tr_st(u + i, v);
  }
}

// dx[b][i][j][c] = sum of the 2x2 block of up[b][2i..2i+1][2j..2j+1][c]  (gradient of nearest x2)
template <typename T>
__global__ void sumpool2x_kernel(const T* __restrict__ up, T* __restrict__ dx, int B, int H, int W, int C) {
  const int64_t n = (int64_t)B * H * W * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const int64_t pix = i / C;
    const int w = (int)(pix % W), h = (int)((pix / W) % H);
    const int b = (int)(pix / ((int64_t)H * W));
    const T* p = up + (((int64_t)b * 2 * H + 2 * h) * 2 * W + 2 * w) * C + c;
    const float v = tr_ld(p) + tr_ld(p + C) + tr_ld(p + (int64_t)2 * W * C) + tr_ld(p + (int64_t)2 * W * C + C);
    tr_st(dx + i, v);
  }
}

// transposed im2col: out[(tap * Ci + ci)][p] = xin[b][oh*stride - pad + kh][ow*stride - pad + kw][ci], xin = nearest-x2(x)
// when ups; p = (b*Ho + oh)*Wo + ow; zero outside the image.  Tiles of 32 pixels x 32 channels go through LDS so both the
// reads (channel-contiguous) and the writes (pixel-contiguous) are coalesced.
template <typename T>
__global__ void __launch_bounds__(256) im2col_t_kernel(const T* __restrict__ x, T* __restrict__ out, int B, int H, int W, int Ci,
                                                      int Ho, int Wo, int k, int stride, int pad, int ups, int64_t ldp) {
  __shared__ float tile[32][33];
  const int64_t P = (int64_t)B * Ho * Wo;
  const int tap = blockIdx.z, kh = tap / k, kw = tap % k;
  const int64_t p0 = (int64_t)blockIdx.x * 32;
  const int c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const int Hin = ups ? 2 * H : H, Win = ups ? 2 * W : W;
  for (int r = ty; r < 32; r += 8) {   // r = pixel within the tile, tx = channel
    const int64_t p = p0 + r;
    float v = 0.f;
    if (p < P && c0 + tx < Ci) {
      const int ow = (int)(p % Wo), oh = (int)((p / Wo) % Ho);
      const int b = (int)(p / ((int64_t)Ho * Wo));
      int ih = oh * stride - pad + kh, iw = ow * stride - pad + kw;
      if (ih >= 0 && ih < Hin && iw >= 0 && iw < Win) {
        if (ups) { ih >>= 1; iw >>= 1; }
        v = tr_ld(x + (((int64_t)b * H + ih) * W + iw) * Ci + c0 + tx);
      }
    }
    tile[r][tx] = v;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {   // r = channel within the tile, tx = pixel
    const int64_t p = p0 + tx;
    if (p < ldp && c0 + r < Ci) tr_st(out + ((int64_t)tap * Ci + c0 + r) * ldp + p, tile[tx][r]);   // columns [P, ldp) are zeros
  }
}

// bf16 form with 16-byte accesses on both sides: a 64-pixel x 64-channel tile per block (one tap).  Reads are bf16x8 along
// the channels of a pixel (NHWC rows), writes bf16x8 along the pixels of a channel (rows of the transposed image); the
// transpose goes through an LDS tile with a 144-byte pitch.  The scalar kernel above moved 2 bytes per access: 0.85 TB/s.
__global__ void __launch_bounds__(256) im2col_t_bf16_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ out, int B, int H, int W,
                                                           int Ci, int Ho, int Wo, int k, int stride, int pad, int ups, int64_t ldp) {
  __shared__ __attribute__((aligned(16))) bf16_t tile[64 * 72];
  const int64_t P = (int64_t)B * Ho * Wo;
  const int tap = blockIdx.z, kh = tap / k, kw = tap % k;
  const int64_t p0 = (int64_t)blockIdx.x * 64;
  const int c0 = blockIdx.y * 64;
  const int Hin = ups ? 2 * H : H, Win = ups ? 2 * W : W;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int idx = threadIdx.x + 256 * j, r = idx >> 3, c8 = (idx & 7) * 8;
    const int64_t p = p0 + r;
    bf16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (bf16_t)0.f;
    if (p < P && c0 + c8 < Ci) {
      const int ow = (int)(p % Wo), oh = (int)((p / Wo) % Ho);
      const int b = (int)(p / ((int64_t)Ho * Wo));
      int ih = oh * stride - pad + kh, iw = ow * stride - pad + kw;
      if (ih >= 0 && ih < Hin && iw >= 0 && iw < Win) {
        if (ups) { ih >>= 1; iw >>= 1; }
        v = *reinterpret_cast<const bf16x8*>(x + (((int64_t)b * H + ih) * W + iw) * Ci + c0 + c8);
      }
    }
    *reinterpret_cast<bf16x8*>(tile + r * 72 + c8) = v;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int idx = threadIdx.x + 256 * j, c = idx >> 3, p8 = (idx & 7) * 8;
    if (c0 + c < Ci && p0 + p8 < ldp) {   // P and ldp are multiples of 8: a vector is wholly inside or outside; [P, ldp) = zeros
      bf16x8 v;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = tile[(p8 + e) * 72 + c];
      *reinterpret_cast<bf16x8*>(out + ((int64_t)tap * Ci + c0 + c) * ldp + p0 + p8) = v;
    }
  }
}

// dw[co][ci][kh][kw] = G[tap * Ci + ci][co]   (G = the fp32 output of the pixel-axis GEMM, [taps * Ci][Co])
// accumulate != 0: dw += ... (the pixel axis is processed in chunks when the im2col image would exceed the 2 GiB descriptor range)
__global__ void wgrad_finish_kernel(const float* __restrict__ G, float* __restrict__ dw, int Co, int Ci, int kk, int accumulate) {
  const int64_t n = (int64_t)Co * Ci * kk;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int tap = (int)(i % kk);
    const int ci = (int)((i / kk) % Ci);
    const int co = (int)(i / ((int64_t)kk * Ci));
    const float v = G[((int64_t)tap * Ci + ci) * Co + co];
    dw[i] = accumulate ? dw[i] + v : v;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// wgrad as a TN implicit GEMM (bf16; 3x3 pad 1 or 1x1, stride 1, W in {8, 16, 32, 64}):
//     dW[co][ci][tap] = sum over pixels p of dY[p][co] * X[p + tap][ci]
// Both operands are PIXEL-major in memory (NHWC) while the MFMA wants 8 consecutive k (= pixels) per lane: the tiles are
// staged row-major in LDS - dY as two [128 pixels][64 co] halves, X as the (TR+2) x (W+2) halo of the same pixels x 64 ci,
// zero outside the image - and the fragments are fetched with ds_read_b64_tr_b16 (the hardware's 4 x 16 transpose: lane
// 4q+p of a 16-lane group supplies row q / columns 4p.., lane i receives column i).  The nine taps read the SAME halo at
// shifted rows, so nothing like the 9x im2col image of the GEMM formulation exists (that image made the small-channel
// layers bandwidth-bound: 74 FLOP per byte).
//   * Workgroup = 8 waves = a 128 co x 64 ci x 9 tap tile of dW for one slice of the pixel axis; wave (hf, wi) = 64 co x
//     16 ci x 9 taps = 36 accumulator fragments: 26 transposing reads per 36 MFMAs keeps the LDS pipe under the MFMA time.
//   * Tiles arrive by LDS-DMA (buffer_load ... lds, 16 B per lane, 1 KiB = 8 rows per wave instruction, zero fill through
//     the descriptor's range check) into a two-stage ring: tile t+1 is in flight while tile t is multiplied, one barrier per
//     tile, no staging registers.  One DMA instruction moves 8 rows x 128 B (whole cache lines) into one KiB of LDS laid
//     out [16-channel unit u][row & 7][32 B] - the lane -> (row, chunk) assignment on the SOURCE side is free - and the k
//     index of a k-step is permuted so that lane group lg owns the 4-pixel blocks lg and lg + 4: a 32-lane half of a
//     transposing read then covers 8 consecutive rows of one unit = 256 contiguous bytes, all 64 banks once.
//   * Pixel slices are summed in slice order by wgrad_tn_finish_kernel (deterministic).
// ---------------------------------------------------------------------------------------------------------------------
struct WgradTnK {
  const bf16_t* dy;   // [B][H][W][Co]
  const bf16_t* x;    // [B][H][W][Ci]
  float* slab;        // [S][Co][Ci][KK]
  int B, H, W, Co, Ci, S, TR, wshift;
  int gx, gy;         // ci / co blocks of the tile grid (the launch is 1-D: wg_block)
  int all_frags;      // diagnostics (DDMI_WGRAD_ALLFRAGS): bit 0 multiply the zero fragments of ragged tiles, bit 1 plain workgroup order (round 2), bit 2 no stagger of the DMA issue
  int ups;            // x is [B][H/2][W/2][Ci] and the convolution ran on its nearest x2 upsampling (Upsample.conv): halo row (iy, ix)
                      // is fetched from source pixel (iy >> 1, ix >> 1) - a change of DMA addresses only
};
#define WG_YB (128 * 128)                 // bytes of one dY half tile: 128 pixels x 64 co
#define WG_XROWS 272                      // halo rows rounded up to whole DMA instructions (>= 4 x 66)
#define WG_XB (WG_XROWS * 128)
#define WG_STAGE (2 * WG_YB + WG_XB)      // 67584 B; two stages = 132 KiB of the CU's 160
#define WG_OOB 0x80000000u

typedef short wg_s16x4 __attribute__((ext_vector_type(4)));
typedef short wg_s16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) wg_s16x4* wg_lds4;
typedef __attribute__((address_space(3))) void wg_lds_void;
__device__ __forceinline__ bf16x8 wg_tr_frag(const unsigned char* lo, const unsigned char* hi) {
  const wg_s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((wg_lds4)lo);
  const wg_s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((wg_lds4)hi);
  const wg_s16x8 both = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, both);   // whole-vector reinterpretation (see k_attention.hip on the element-wise trap)
}
// byte offset inside a [rows][64 channel] tile of (row r, 16-channel unit 0): KiB blocks of 8 rows, [unit][row & 7][32 B]
__device__ __forceinline__ int wg_row(int r) { return (r >> 3) * 768 + r * 32; }
// the same offset without the quarter-rate multiply (WIDE tap loop; in the narrow loop, which computes two addresses per tap, the
// multiply form schedules better: 562 vs 624 us at 896 -> 896 @16x16)
__device__ __forceinline__ int wg_row_s(int r) { return ((r >> 3) << 10) | ((r & 7) << 5); }

// 1-D launch, XCD-aware: workgroups are dealt round-robin to the 8 XCDs (linear id % 8), each with its own L2.  The tile list is
// slice-major - the gx * gy workgroups of one pixel slice read the SAME dY / X tiles - and XCD x takes a contiguous eighth of
// it, so the readers of a slice share one L2 (as a 3-D grid their linear ids were consecutive: eight XCDs, every tile fetched
// from HBM by each of them; 224 -> 224 at 64x64 has exactly 8 workgroups per slice).  All workgroups are resident at once (one
// per CU) and walk their slices in step.  Returns false for the padding of the last eighth.
__device__ __forceinline__ bool wg_block(const WgradTnK& a, int* bx, int* by, int* bz) {
  const int per = a.gx * a.gy, n = per * a.S, chunk = (int)(gridDim.x >> 3);
  const int idx = a.all_frags & 2 ? (int)blockIdx.x : ((int)blockIdx.x & 7) * chunk + ((int)blockIdx.x >> 3);   // (bit 1: plain order)
  if (idx >= n) return false;
  const int z = idx / per, rem = idx - z * per;
  *bz = z;
  *by = rem / a.gx;
  *bx = rem - *by * a.gx;
  return true;
}

// WIDE (W >= 32): the two 4-pixel blocks of a lane's k-step (pixels p and p + 16) lie in one image row, their halo rows 16 apart =
// two KiB blocks of the X image: the second transposing read is the first one's address + 2048 (an instruction offset), which halves
// the address arithmetic of the tap loop (it was ~10 VALU per 4 MFMAs).
template <int KK, bool WIDE>
__global__ void __launch_bounds__(512) wgrad_tn_kernel(WgradTnK a) {
  constexpr int KS3 = KK == 9 ? 3 : 1, PAD = KK == 9 ? 1 : 0;
  extern __shared__ __attribute__((aligned(1024))) unsigned char wg_smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // wave -> (co half hf, ci fragment wi).  Waves w and w + 4 share a SIMD; the pairing (hf 0, wi) + (hf 1, wi ^ 2) puts on every
  // SIMD one wave of each co half AND one wave of each ci half, so a ragged tile (Co = 224: the second co block holds 96 channels,
  // Ci = 224: the last ci block 32) leaves every SIMD the same share of the MFMAs that are left - none are spent on zero fragments
  // (round 2 multiplied them: 23 % of the kernel's MFMAs at 224 channels)
  const int hf = wave >> 2, wi = (wave ^ (hf << 1)) & 3;
  const int lr = lane & 15, lg = lane >> 4, q = lr >> 2, p4 = lr & 3;
  const int W = a.W, H = a.H, TR = a.TR, PT = TR * W;          // pixels per tile (<= 128)
  const int HWd = W + 2 * PAD, NHR = (TR + 2 * PAD) * HWd;     // halo geometry (no halo for 1x1)
  int bx, by, bz;
  if (!wg_block(a, &bx, &by, &bz)) return;   // (workgroup-uniform, before any barrier)
  const int ci0 = bx * 64, co0 = by * 128;
  // co fragments of this wave that hold channels (0..4); a wave whose ci fragment is past Ci only moves tiles
  const int nf_co = min(4, max(0, (a.Co - co0 - 64 * hf + 15) >> 4));
  const int nf = a.all_frags & 1 ? 4 : ci0 + 16 * wi < a.Ci ? nf_co : 0;
  const int tiles_per_sample = (H * W) / PT, ntiles = a.B * tiles_per_sample;
  const int per = (ntiles + a.S - 1) / a.S, t_begin = bz * per, t_end = min(ntiles, t_begin + per);

  // ---- DMA roles (fixed over the tiles): instruction gi = wave + 8 j moves rows 8 gi .. 8 gi + 7; lane = (row, 16-byte slot)
  const int drow = (lane >> 1) & 7, dchunk = 2 * (lane >> 4) + (lane & 1);   // LDS position 16 lane = [unit][row][half]
  unsigned yoff[4];            // dY: 32 instructions (2 halves x 16 row groups)
  int xoff[5], xhy[5];         // X halo: <= 34 instructions; byte offset relative to the tile's first pixel row, halo row - PAD
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int gi = wave + 8 * j, yh = gi >> 4, r = 8 * (gi & 15) + drow;
    const int co = co0 + 64 * yh + 8 * dchunk;
    yoff[j] = (r < PT && co < a.Co) ? (unsigned)((r * a.Co + co) * 2) : WG_OOB;
  }
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    const int gi = wave + 8 * j, r = 8 * gi + drow;
    const int ci = ci0 + 8 * dchunk;
    const int hy = r / HWd, hx = r - hy * HWd;
    const bool ok = r < NHR && ci < a.Ci && hx >= PAD && hx < W + PAD;
    xoff[j] = a.ups ? ((((hy - PAD) >> 1) * (W >> 1) + ((hx - PAD) >> 1)) * a.Ci + ci) * 2 : (((hy - PAD) * W + hx - PAD) * a.Ci + ci) * 2;
    xhy[j] = ok ? hy - PAD : -(1 << 20);
  }
  // (sample, first row) of the tile to fetch next, advanced tile by tile: the division t / tiles_per_sample per tile was ~25
  // instructions of every wave's issue stream
  int nb = t_begin / tiles_per_sample, nty = (t_begin - nb * tiles_per_sample) * TR;
  auto issue = [&](int b, int ty0, int stage) {
    unsigned char* base = wg_smem + stage * WG_STAGE;
    const bf16_t* yg = a.dy + ((int64_t)b * H * W + ty0 * W) * a.Co;
    const int x_px = a.ups ? (H >> 1) * (W >> 1) : H * W;     // pixels of one sample of x
    const bf16_t* xg = a.x + (int64_t)b * x_px * a.Ci;
    auto y_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)yg, 0, (unsigned)(PT * a.Co * 2), 0x00020000);
    auto x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)xg, 0, (unsigned)(x_px * a.Ci * 2), 0x00020000);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int gi = wave + 8 * j;
      if (8 * (gi & 15) < PT)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(y_rsrc, (wg_lds_void*)(base + (gi >> 4) * WG_YB + (gi & 15) * 1024), 16, yoff[j], 0, 0, 0);
    }
    const int tile_off = a.ups ? (ty0 >> 1) * (W >> 1) * a.Ci * 2 : ty0 * W * a.Ci * 2;   // TR is even: ty0 >> 1 is exact
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const int gi = wave + 8 * j;
      if (8 * gi < NHR) {
        const unsigned voff = (unsigned)(ty0 + xhy[j]) < (unsigned)H ? (unsigned)(xoff[j] + tile_off) : WG_OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rsrc, (wg_lds_void*)(base + 2 * WG_YB + gi * 1024), 16, voff, 0, 0, 0);
      }
    }
  };

  f32x4 acc[4][KK];
#pragma unroll
  for (int f = 0; f < 4; ++f)
#pragma unroll
    for (int t = 0; t < KK; ++t) acc[f][t] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (t_begin < t_end) issue(nb, nty, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int stage = 0;
  for (int t = t_begin; t < t_end; ++t) {
    // the other stage was last read before the barrier that ended tile t-1.  (Spreading the nine DMA instructions over the k-steps,
    // the two waves of a SIMD half a tile apart - an LDS-DMA instruction costs its wave 60-180 issue cycles - measured 5-8 %
    // SLOWER on every shape: the pieces' scalar bookkeeping inside the k-loop costs more than the idle matrix pipe it removes.)
    // An LDS-DMA instruction costs its wave 60-180 issue cycles and all eight waves used to issue their nine at the top of the tile,
    // the matrix pipe idle meanwhile (fetching nothing after the first tile - timing only - is 20-28 % faster).  The two waves of a
    // SIMD are now a k-step apart: the hf 1 wave issues behind its first k-step, under the hf 0 wave's MFMAs and vice versa.
    nty += TR;
    if (nty >= H) { nty = 0; ++nb; }
    const bool more = t + 1 < t_end && !(a.all_frags & 8);   // (bit 3, timing experiments only: no tile after the first is fetched)
    // (WIDE only: in the narrow instantiation the second in-loop copy of the issue code spills, 6-12 % slower)
    const bool late = WIDE && more && hf == 1 && nf > 0 && !(a.all_frags & 4);
    if (more && !late) issue(nb, nty, stage ^ 1);
    const unsigned char* Y = wg_smem + stage * WG_STAGE + hf * WG_YB;
    const unsigned char* X = wg_smem + stage * WG_STAGE + 2 * WG_YB;
    // two copies of the k-loop (wave-uniform choice): all four co fragments, or the first two (a ragged co half: nf <= 2); a wave
    // without channels skips it.  (Guarding fragments 2 / 3 inside ONE loop body cut it into basic blocks the scheduler no longer
    // interleaves: +6 % on the full waves.)
    auto k_loop = [&](auto nfc) {
      constexpr int NF = decltype(nfc)::value;
#pragma unroll 1
      for (int ks = 0; ks < PT / 32; ++ks) {
        // k index of the k-step: lane group lg, element e -> pixel 4 lg + e (e < 4), 16 + 4 lg + e - 4 (e >= 4); this lane
        // supplies the address of row q of both blocks
        const int pl0 = ks * 32 + 4 * lg + q, pl1 = pl0 + 16;
        const unsigned char* y0 = Y + wg_row(pl0) + 8 * p4;
        const unsigned char* y1 = Y + wg_row(pl1) + 8 * p4;
        bf16x8 af[NF];
#pragma unroll
        for (int f = 0; f < NF; ++f) af[f] = wg_tr_frag(y0 + 256 * f, y1 + 256 * f);
        if constexpr (WIDE) {
          if (late && ks == 1) issue(nb, nty, stage ^ 1);
        }
        const int r0 = (pl0 >> a.wshift) * HWd + (pl0 & (W - 1)), r1 = (pl1 >> a.wshift) * HWd + (pl1 & (W - 1));
        const unsigned char* xb = X + 256 * wi + 8 * p4;
#pragma unroll
        for (int kh = 0; kh < KS3; ++kh)
#pragma unroll
          for (int kw = 0; kw < KS3; ++kw) {
            const int sh = kh * HWd + kw;
            const unsigned char* xlo = xb + (WIDE ? wg_row_s(r0 + sh) : wg_row(r0 + sh));
            const bf16x8 bf = wg_tr_frag(xlo, WIDE ? xlo + 2048 : xb + wg_row(r1 + sh));
#pragma unroll
            for (int f = 0; f < NF; ++f)
              acc[f][kh * KS3 + kw] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[f], bf, acc[f][kh * KS3 + kw], 0, 0, 0);
          }
      }
    };
    if (nf > 2) k_loop(std::integral_constant<int, 4>{});
    else if (nf > 0) k_loop(std::integral_constant<int, 2>{});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of tile t+1 has landed
    __syncthreads();                                    // ... everyone's has, and everyone is done reading tile t
    stage ^= 1;
  }
  // slab[z][co][ci][tap]: lane (lr = ci, lg -> co rows 4 lg + e) of accumulator fragment f
  const int ci = ci0 + 16 * wi + lr;
  if (ci < a.Ci) {
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int co = co0 + 64 * hf + 16 * f + 4 * lg + e;
        if (co < a.Co) {
          float* dst = a.slab + (((int64_t)bz * a.Co + co) * a.Ci + ci) * KK;
#pragma unroll
          for (int t = 0; t < KK; ++t) dst[t] = acc[f][t][e];
        }
      }
  }
}

// The 1x1 convolutions (attention qkv / proj_out, ResBlock skip connections): dW[co][ci] = sum_p dY[p][co] X[p][ci] over the flat
// pixel list - the same machinery without a halo, but with ONE tap the 64 x 16 wave tile above would fetch ten fragments for
// four MFMAs.  Here a wave owns 128 co x 64 ci (32 accumulator fragments, 12 fragment fetches per 32 MFMAs), the workgroup
// 256 co x 256 ci, a stage is 64 pixels of both operands (4 + 4 panels of [64 px][64 ch], 64 KiB), two stages.
#define WG1_PANEL (64 * 128)            // bytes of one [64 px][64 ch] panel
#define WG1_STAGE (8 * WG1_PANEL)       // 4 dY panels + 4 X panels
__global__ void __launch_bounds__(512) wgrad_tn1_kernel(WgradTnK a) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char wg_smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wi = wave & 3, hf = wave >> 2;
  const int lr = lane & 15, lg = lane >> 4, q = lr >> 2, p4 = lr & 3;
  int bx, by, bz;
  if (!wg_block(a, &bx, &by, &bz)) return;
  const int ci0 = bx * 256, co0 = by * 256;
  const int ntiles = a.B;                         // a.B = number of 64-pixel tiles of the flat pixel list here
  const int per = (ntiles + a.S - 1) / a.S, t_begin = bz * per, t_end = min(ntiles, t_begin + per);
  const int drow = (lane >> 1) & 7, dchunk = 2 * (lane >> 4) + (lane & 1);
  unsigned yoff[4], xoff[4];                       // instruction gi = wave + 8 j: panel gi >> 3, rows 8 (gi & 7) ..
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int gi = wave + 8 * j, panel = gi >> 3, r = 8 * (gi & 7) + drow;
    const int co = co0 + 64 * panel + 8 * dchunk, ci = ci0 + 64 * panel + 8 * dchunk;
    yoff[j] = co < a.Co ? (unsigned)((r * a.Co + co) * 2) : WG_OOB;
    xoff[j] = ci < a.Ci ? (unsigned)((r * a.Ci + ci) * 2) : WG_OOB;
  }
  auto issue = [&](int t, int stage) {
    unsigned char* base = wg_smem + stage * WG1_STAGE;
    auto y_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(a.dy + (int64_t)t * 64 * a.Co), 0, (unsigned)(64 * a.Co * 2), 0x00020000);
    auto x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x + (int64_t)t * 64 * a.Ci), 0, (unsigned)(64 * a.Ci * 2), 0x00020000);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int gi = wave + 8 * j;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(y_rsrc, (wg_lds_void*)(base + gi * 1024), 16, yoff[j], 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rsrc, (wg_lds_void*)(base + 4 * WG1_PANEL + gi * 1024), 16, xoff[j], 0, 0, 0);
    }
  };
  f32x4 acc[8][4];
#pragma unroll
  for (int f = 0; f < 8; ++f)
#pragma unroll
    for (int u = 0; u < 4; ++u) acc[f][u] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (t_begin < t_end) issue(t_begin, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int stage = 0;
  for (int t = t_begin; t < t_end; ++t) {
    if (t + 1 < t_end) issue(t + 1, stage ^ 1);
    const unsigned char* Y = wg_smem + stage * WG1_STAGE + 2 * hf * WG1_PANEL + 8 * p4;
    const unsigned char* X = wg_smem + stage * WG1_STAGE + (4 + wi) * WG1_PANEL + 8 * p4;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int o0 = wg_row(ks * 32 + 4 * lg + q), o1 = wg_row(ks * 32 + 16 + 4 * lg + q);
      bf16x8 bfr[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) bfr[u] = wg_tr_frag(X + o0 + 256 * u, X + o1 + 256 * u);
#pragma unroll
      for (int f = 0; f < 8; ++f) {
        const int fo = (f >> 2) * WG1_PANEL + 256 * (f & 3);
        const bf16x8 af = wg_tr_frag(Y + o0 + fo, Y + o1 + fo);
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[f][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bfr[u], acc[f][u], 0, 0, 0);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    stage ^= 1;
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int ci = ci0 + 64 * wi + 16 * u + lr;
    if (ci >= a.Ci) continue;
#pragma unroll
    for (int f = 0; f < 8; ++f)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int co = co0 + 128 * hf + 16 * f + 4 * lg + e;
        if (co < a.Co) a.slab[((int64_t)bz * a.Co + co) * a.Ci + ci] = acc[f][u][e];
      }
  }
}

__global__ void wgrad_tn_finish_kernel(const float* __restrict__ slab, float* __restrict__ dw, int64_t n, int S, int accumulate) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float acc = 0.f;
    for (int z = 0; z < S; ++z) acc += slab[(int64_t)z * n + i];
    dw[i] = accumulate ? dw[i] + acc : acc;
  }
}
// 16-byte form, four slices in flight per thread; the slices are still added in slice order (bit-identical to the scalar kernel)
__global__ void __launch_bounds__(256) wgrad_tn_finish_vec4_kernel(const float4* __restrict__ slab, float4* __restrict__ dw, int64_t n4, int S,
                                                                   int accumulate) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    int z = 0;
    for (; z + 4 <= S; z += 4) {
      const float4 a = slab[(int64_t)z * n4 + i], b = slab[(int64_t)(z + 1) * n4 + i], c = slab[(int64_t)(z + 2) * n4 + i],
                   d = slab[(int64_t)(z + 3) * n4 + i];
      acc.x = (((acc.x + a.x) + b.x) + c.x) + d.x; acc.y = (((acc.y + a.y) + b.y) + c.y) + d.y;
      acc.z = (((acc.z + a.z) + b.z) + c.z) + d.z; acc.w = (((acc.w + a.w) + b.w) + c.w) + d.w;
    }
    for (; z < S; ++z) {
      const float4 a = slab[(int64_t)z * n4 + i];
      acc.x += a.x; acc.y += a.y; acc.z += a.z; acc.w += a.w;
    }
    if (accumulate) {
      const float4 o = dw[i];
      acc.x = o.x + acc.x; acc.y = o.y + acc.y; acc.z = o.z + acc.z; acc.w = o.w + acc.w;
    }
    dw[i] = acc;
  }
}

// H, W: the grid the convolution ran on (= dY's; with `ups` the input x is [H/2][W/2])
bool wgrad_tn_ok(int dt, int B, int H, int W, int Ci, int Co, int k, int stride, int ups) {
  if (dt != DT_BF16 || stride != 1 || !(k == 1 || k == 3) || Ci % 8 || Co % 8 || (ups && k != 3)) return false;
  if (k == 1) return ((int64_t)B * H * W) % 64 == 0;          // flat pixel list in 64-pixel tiles
  if (!(W == 8 || W == 16 || W == 32 || W == 64)) return false;
  const int PT = std::min(128, H * W);
  if (ups && ((PT / W) & 1)) return false;                    // tiles of an even number of rows map onto whole source rows
  return (H * W) % PT == 0 && PT % 32 == 0 && PT % W == 0;
}
int wgrad_tn_splits(int B, int H, int W, int Ci, int Co, int k) {
  // one workgroup per CU (128+ KiB of LDS): fill the 256 CUs once, keep the fp32 slabs (S x |dW|) small
  static const int target = getenv("DDMI_WGRAD_TN_WGS") ? atoi(getenv("DDMI_WGRAD_TN_WGS")) : 256;
  if (k == 1) {
    const int tiles = cdiv(Ci, 256) * cdiv(Co, 256), ntiles = (int)((int64_t)B * H * W / 64);
    return std::max(1, std::min(std::min(target / tiles, ntiles), 128));
  }
  const int tiles = cdiv(Ci, 64) * cdiv(Co, 128), PT = std::min(128, H * W), ntiles = B * (H * W / PT);
  return std::max(1, std::min(std::min(target / tiles, ntiles), 64));
}
// slab: wgrad_tn_splits(...) * Co * Ci * k*k floats
int launch_wgrad_tn(const void* dy, const void* x, float* dw, float* slab, int B, int H, int W, int Ci, int Co, int k, hipStream_t s,
                    int accumulate, int ups) {
  WgradTnK a{};
  a.dy = (const bf16_t*)dy; a.x = (const bf16_t*)x; a.slab = slab;
  a.B = B; a.H = H; a.W = W; a.Co = Co; a.Ci = Ci; a.ups = ups ? 1 : 0;
  a.S = wgrad_tn_splits(B, H, W, Ci, Co, k);
  static const int all_frags = getenv("DDMI_WGRAD_ALLFRAGS") ? atoi(getenv("DDMI_WGRAD_ALLFRAGS")) : 0;   // diagnostics
  a.all_frags = all_frags;
  const int kk = k * k;
  static bool attr_done = false;
  if (!attr_done) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_tn_kernel<9, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * WG_STAGE));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_tn_kernel<9, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * WG_STAGE));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_tn1_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * WG1_STAGE));
    attr_done = true;
  }
  PROF_SCOPE("wgrad_tn", 2.0 * B * H * W * (double)Co * Ci * kk, ((double)B * H * W * (Co + Ci)) * 2.0, s);
  if (k == 3) {
    a.TR = std::min(128, H * W) / W;
    a.wshift = W == 8 ? 3 : W == 16 ? 4 : W == 32 ? 5 : 6;
    a.gx = cdiv(Ci, 64); a.gy = cdiv(Co, 128);
    dim3 g((unsigned)(cdiv(a.gx * a.gy * a.S, 8) * 8));
    if (W >= 32) hipLaunchKernelGGL((wgrad_tn_kernel<9, true>), g, dim3(512), 2 * WG_STAGE, s, a);
    else hipLaunchKernelGGL((wgrad_tn_kernel<9, false>), g, dim3(512), 2 * WG_STAGE, s, a);
  } else {
    a.B = (int)((int64_t)B * H * W / 64);   // 64-pixel tiles of the flat pixel list
    a.gx = cdiv(Ci, 256); a.gy = cdiv(Co, 256);
    dim3 g((unsigned)(cdiv(a.gx * a.gy * a.S, 8) * 8));
    hipLaunchKernelGGL(wgrad_tn1_kernel, g, dim3(512), 2 * WG1_STAGE, s, a);
  }
  const int64_t n = (int64_t)Co * Ci * kk;
  if (n % 4 == 0 && (((uintptr_t)slab | (uintptr_t)dw) & 15) == 0)
    hipLaunchKernelGGL(wgrad_tn_finish_vec4_kernel, dim3(tr_blocks(n / 4)), dim3(256), 0, s, (const float4*)slab, (float4*)dw, n / 4, a.S, accumulate);
  else
    hipLaunchKernelGGL(wgrad_tn_finish_kernel, dim3(tr_blocks(n)), dim3(256), 0, s, slab, dw, n, a.S, accumulate);
  HIP_TRY(hipGetLastError());
  return DDMI_OK;
}

// out[b][c] = sum over HW of dy[b][hw][c].  grid (ceil(C/64), B, S): S pixel slices per sample so that a launch fills the
// chip (a [B, C/64] grid alone leaves most CUs idle); block 256 = 64 channels x 4 pixel lanes; the S partials of a (b, c)
// are added by sum_slices_kernel in slice order (deterministic).
template <typename T>
__global__ void __launch_bounds__(256) sum_hw_kernel(const T* __restrict__ dy, float* __restrict__ part, int HW, int C, int S) {
  __shared__ float red[4][64];
  const int b = blockIdx.y, c = blockIdx.x * 64 + (threadIdx.x & 63), lane = threadIdx.x >> 6, z = blockIdx.z;
  const int per = (HW + S - 1) / S, p0 = z * per, p1 = min(HW, p0 + per);
  float acc = 0.f;
  if (c < C)
    for (int p = p0 + lane; p < p1; p += 4) acc += tr_ld(dy + ((int64_t)b * HW + p) * C + c);
  red[lane][threadIdx.x & 63] = acc;
  __syncthreads();
  if (lane == 0 && c < C)
    part[((int64_t)z * gridDim.y + b) * C + c] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}
// bf16, C % 8 == 0: 16-byte loads - block 256 = 8 channel chunks (64 channels) x 32 pixel lanes, same grid and slices
__global__ void __launch_bounds__(256) sum_hw_bf16x8_kernel(const bf16_t* __restrict__ dy, float* __restrict__ part, int HW, int C, int S) {
  __shared__ float red[32][65];
  const int b = blockIdx.y, ch = threadIdx.x & 7, lane = threadIdx.x >> 3, z = blockIdx.z;
  const int c = blockIdx.x * 64 + 8 * ch;
  const int per = (HW + S - 1) / S, p0 = z * per, p1 = min(HW, p0 + per);
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (c < C)
    for (int p = p0 + lane; p < p1; p += 32) {
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(dy + ((int64_t)b * HW + p) * C + c);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += (float)v[e];
    }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[lane][8 * ch + e] = acc[e];
  __syncthreads();
  if (threadIdx.x < 64 && blockIdx.x * 64 + threadIdx.x < C) {
    float t = 0.f;
    for (int l = 0; l < 32; ++l) t += red[l][threadIdx.x];   // fixed order: deterministic
    part[((int64_t)z * gridDim.y + b) * C + blockIdx.x * 64 + threadIdx.x] = t;
  }
}
__global__ void sum_slices_kernel(const float* __restrict__ part, float* __restrict__ out, int64_t n, int S) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float acc = 0.f;
  for (int z = 0; z < S; ++z) acc += part[(int64_t)z * n + i];
  out[i] = acc;
}

// out[c] = sum_b in[b][c]: block 256 = 64 channels x 4 batch lanes (a lane adds every 4th row, four loads in flight), the four
// partials are added in lane order (deterministic)
__global__ void __launch_bounds__(256) sum_b_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int C, int accumulate) {
  __shared__ float red[4][64];
  const int cl = threadIdx.x & 63, bl = threadIdx.x >> 6, c = blockIdx.x * 64 + cl;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (c < C) {
    int b = bl;
    for (; b + 12 < B; b += 16) {
      a0 += in[(int64_t)b * C + c];
      a1 += in[(int64_t)(b + 4) * C + c];
      a2 += in[(int64_t)(b + 8) * C + c];
      a3 += in[(int64_t)(b + 12) * C + c];
    }
    for (; b < B; b += 4) a0 += in[(int64_t)b * C + c];
  }
  red[bl][cl] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (bl == 0 && c < C) out[c] = (accumulate ? out[c] : 0.f) + ((red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]));
}

int launch_pack_dgrad_weight(const float* w, void* out, int dt, int Co, int Ci, int k, hipStream_t s) {
  const int64_t n = (int64_t)Co * Ci * k * k;
  if (k == 1 || k == 3) {
    const int TCI = k == 1 ? 64 : 8;
    dim3 g((unsigned)cdiv(Ci, TCI), (unsigned)cdiv(Co, 32));
    if (dt == DT_F32) hipLaunchKernelGGL(pack_dgrad_weight_tiled_kernel<float>, g, dim3(256), 0, s, w, (float*)out, Co, Ci, k, TCI);
    else hipLaunchKernelGGL(pack_dgrad_weight_tiled_kernel<bf16_t>, g, dim3(256), 0, s, w, (bf16_t*)out, Co, Ci, k, TCI);
  } else if (dt == DT_F32) hipLaunchKernelGGL(pack_dgrad_weight_kernel<float>, dim3(tr_blocks(n)), dim3(256), 0, s, w, (float*)out, Co, Ci, k);
  else hipLaunchKernelGGL(pack_dgrad_weight_kernel<bf16_t>, dim3(tr_blocks(n)), dim3(256), 0, s, w, (bf16_t*)out, Co, Ci, k);
  HIP_TRY(hipGetLastError());
  return DDMI_OK;
}
// the training step's forward pack: same result as launch_pack_conv_weight (k_elementwise.hip), LDS-tiled
int launch_pack_fwd_weight(const float* w, void* out, int dt, int Co, int Ci, int k, hipStream_t s) {
  if (k * k > 9) return launch_pack_conv_weight(w, out, dt, Co, Ci, k, k, s);
  dim3 g((unsigned)cdiv(Ci, 256), (unsigned)Co);
  if (dt == DT_F32) hipLaunchKernelGGL(pack_fwd_weight_tiled_kernel<float>, g, dim3(256), 0, s, w, (float*)out, Co, Ci, k * k);
  else hipLaunchKernelGGL(pack_fwd_weight_tiled_kernel<bf16_t>, g, dim3(256), 0, s, w, (bf16_t*)out, Co, Ci, k * k);
  HIP_TRY(hipGetLastError());
  return DDMI_OK;
}
// channel concatenation of two NHWC tensors (the UNet decoder's th.cat([h, hs.pop()], dim=1), openaimodel.py:739) and its inverse, in
// 16-byte vectors: row r of out = [a row r (va vectors) | b row r (vb vectors)].  SPLIT copies the other way (out -> a, b).
template <bool SPLIT>
__global__ void __launch_bounds__(256) cat2_kernel(uint4* __restrict__ a, uint4* __restrict__ b, uint4* __restrict__ out, int64_t rows, int va, int vb) {
  const int vt = va + vb;
  const int64_t n = rows * vt;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / vt;
    const int c = (int)(i - r * vt);
    uint4* side = c < va ? a + r * va + c : b + r * vb + (c - va);
    if (SPLIT) *side = out[i];
    else out[i] = *side;
  }
}
int launch_cat2(const void* a, const void* b, void* out, int dt, int64_t rows, int Ca, int Cb, hipStream_t s, int split) {
  const int vn = 16 / (int)dt_size(dt);
  DDMI_REQUIRE(Ca > 0 && Cb > 0 && Ca % vn == 0 && Cb % vn == 0, "cat2: channel counts must be multiples of the 16-byte vector");
  const int64_t n = rows * ((Ca + Cb) / vn);
  PROF_SCOPE("cat2", 0, 2.0 * n * 16, s);
  if (split) hipLaunchKernelGGL(cat2_kernel<true>, dim3(tr_blocks(n)), dim3(256), 0, s, (uint4*)a, (uint4*)b, (uint4*)out, rows, Ca / vn, Cb / vn);
  else hipLaunchKernelGGL(cat2_kernel<false>, dim3(tr_blocks(n)), dim3(256), 0, s, (uint4*)a, (uint4*)b, (uint4*)out, rows, Ca / vn, Cb / vn);
  HIP_TRY(hipGetLastError());
  return DDMI_OK;
}
int launch_zero_insert2x(const void* dy, void* u, int dt, int B, int Ho, int Wo, int C, hipStream_t s) {
  const int64_t n = (int64_t)B * 4 * Ho * Wo * C;
  PROF_SCOPE("dgrad_layout", 0, 1.25 * n * dt_size(dt), s);
  if (dt == DT_F32) hipLaunchKernelGGL(zero_insert2x_kernel<float>, dim3(tr_blocks(n)), dim3(256), 0, s, (const float*)dy, (float*)u, B, Ho, Wo, C);
  else hipLaunchKernelGGL(zero_insert2x_kernel<bf16_t>, dim3(tr_blocks(n)), dim3(256), 0, s, (const bf16_t*)dy, (bf16_t*)u, B, Ho, Wo, C);
  HIP_TRY(hipGetLastError());
  return DDMI_OK;
}
int launch_sumpool2x(const void* up, void* dx, int dt, int B, int H, int W, int C, hipStream_t s) {
  const int64_t n = (int64_t)B * H * W * C;
  PROF_SCOPE("dgrad_layout", 0, 5.0 * n * dt_size(dt), s);
  if (dt == DT_F32) hipLaunchKernelGGL(sumpool2x_kernel<float>, dim3(tr_blocks(n)), dim3(256), 0, s, (const float*)up, (float*)dx, B, H, W, C);
  else hipLaunchKernelGGL(sumpool2x_kernel<bf16_t>, dim3(tr_blocks(n)), dim3(256), 0, s, (const bf16_t*)up, (bf16_t*)dx, B, H, W, C);
  HIP_TRY(hipGetLastError());
  return DDMI_OK;
}
// out rows have pitch ldp >= P = B*Ho*Wo elements; columns [P, ldp) are written as zeros.  A pitch that is not a large power
// of two matters: the wgrad GEMM walks hundreds of these rows in lock step, and rows 2^k bytes apart fall on the same L2 / HBM
// channels.
int launch_im2col_t(const void* x, void* out, int dt, int B, int H, int W, int Ci, int Ho, int Wo, int k, int stride, int pad, int ups,
                    int64_t ldp, hipStream_t s) {
  const int64_t P = (int64_t)B * Ho * Wo;
  DDMI_REQUIRE(ldp >= P, "im2col_t: row pitch below the row length");
  dim3 g((unsigned)cdiv64(ldp, 32), (unsigned)cdiv(Ci, 32), (unsigned)(k * k));
  PROF_SCOPE("wgrad_im2col_t", 0, 2.0 * k * k * (double)P * Ci * dt_size(dt), s);
  if (dt == DT_F32) hipLaunchKernelGGL(im2col_t_kernel<float>, g, dim3(256), 0, s, (const float*)x, (float*)out, B, H, W, Ci, Ho, Wo, k, stride, pad, ups, ldp);
  else if (Ci % 8 == 0 && P % 8 == 0 && ldp % 8 == 0)
    hipLaunchKernelGGL(im2col_t_bf16_kernel, dim3((unsigned)cdiv64(ldp, 64), (unsigned)cdiv(Ci, 64), (unsigned)(k * k)), dim3(256), 0, s,
                       (const bf16_t*)x, (bf16_t*)out, B, H, W, Ci, Ho, Wo, k, stride, pad, ups, ldp);
  else hipLaunchKernelGGL(im2col_t_kernel<bf16_t>, g, dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)out, B, H, W, Ci, Ho, Wo, k, stride, pad, ups, ldp);
  HIP_TRY(hipGetLastError());
  return DDMI_OK;
}
int launch_wgrad_finish(const float* G, float* dw, int Co, int Ci, int kk, int accumulate, hipStream_t s) {
  hipLaunchKernelGGL(wgrad_finish_kernel, dim3(tr_blocks((int64_t)Co * Ci * kk)), dim3(256), 0, s, G, dw, Co, Ci, kk, accumulate);
  HIP_TRY(hipGetLastError());
  return DDMI_OK;
}
int sum_hw_slices(int B, int HW, int C) {   // pixel slices per sample: ~1024 blocks per launch, at least 64 pixels per slice
  const int base = std::max(1, cdiv(C, 64) * B);
  return std::max(1, std::min(std::min(cdiv(1024, base), HW / 64), 64));
}
// `part` = scratch of sum_hw_slices(B, HW, C) * B * C floats (unused when the slice count is 1)
int launch_sum_hw(const void* dy, float* out, float* part, int dt, int B, int HW, int C, hipStream_t s) {
  const int S = sum_hw_slices(B, HW, C);
  dim3 g((unsigned)cdiv(C, 64), (unsigned)B, (unsigned)S);
  PROF_SCOPE("bias_emb_grad", 0, (double)B * HW * C * dt_size(dt), s);
  float* dst = S == 1 ? out : part;
  if (dt == DT_F32) hipLaunchKernelGGL(sum_hw_kernel<float>, g, dim3(256), 0, s, (const float*)dy, dst, HW, C, S);
  else if (C % 8 == 0) hipLaunchKernelGGL(sum_hw_bf16x8_kernel, g, dim3(256), 0, s, (const bf16_t*)dy, dst, HW, C, S);
  else hipLaunchKernelGGL(sum_hw_kernel<bf16_t>, g, dim3(256), 0, s, (const bf16_t*)dy, dst, HW, C, S);
  if (S > 1) hipLaunchKernelGGL(sum_slices_kernel, dim3((unsigned)cdiv64((int64_t)B * C, 256)), dim3(256), 0, s, part, out, (int64_t)B * C, S);
  HIP_TRY(hipGetLastError());
  return DDMI_OK;
}
int launch_sum_b(const float* in, float* out, int B, int C, hipStream_t s, int accumulate) {
  hipLaunchKernelGGL(sum_b_kernel, dim3(cdiv(C, 64)), dim3(256), 0, s, in, out, B, C, accumulate);
  HIP_TRY(hipGetLastError());
  return DDMI_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// GroupNorm(32) [+ SiLU] backward.  mr = saved [B][32][2] (mean, rstd) of the forward.
//   x^ = (x - mean) rstd ; z = gamma x^ + beta ; y = silu?(z)
//   dz = dy silu'(z) ; dgamma = sum dz x^ ; dbeta = sum dz ; dx^ = dz gamma
//   dx = rstd (dx^ - mean_g(dx^) - x^ mean_g(dx^ x^))
// pass 1, grid (32, B): a thread owns ONE channel of the group for a fixed set of pixels -> per-channel and per-group sums
// reduced through LDS in a fixed order (no atomics: deterministic)
// ---------------------------------------------------------------------------------------------------------------------
// FAST (bf16 storage): v_exp_f32 + v_rcp_f32 as in the forward (k_groupnorm.hip) - the precise expf and division were ~60 VALU
// instructions per element and held the GroupNorm backward's statistics pass at 2.9 TB/s; the fp32 parity mode keeps them
template <bool FAST = false>
__device__ __forceinline__ float tr_dsilu(float z) {
  const float sg = FAST ? rcp_fast(1.0f + __expf(-z)) : 1.0f / (1.0f + expf(-z));
  return sg * (1.0f + z * (1.0f - sg));
}

// 16-byte vector access to NHWC rows (8 bf16 / 4 fp32 channels)
template <typename T> struct TrVec;
template <> struct TrVec<float> {
  static constexpr int N = 4;
  __device__ static void load(const float* p, float* v) { const float4 a = *reinterpret_cast<const float4*>(p); v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; }
  __device__ static void store(float* p, const float* v) { *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]); }
};
template <> struct TrVec<bf16_t> {
  static constexpr int N = 8;
  __device__ static void load(const bf16_t* p, float* v) {
    const bf16x8 a = *reinterpret_cast<const bf16x8*>(p);
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

// pass 1, grid (nsplit, B), block = NV x R threads (NV = C / vector channel-vectors of a pixel, R pixel lanes): whole NHWC rows
// are read with 16-byte loads (the first version gave a block ONE group's 7..84 channels of every pixel: 14-byte pieces of
// 448-byte rows).  Per-thread sums of its channels over its pixels -> LDS -> per-channel partials of this pixel slice:
// part[b][split][c] = (dgamma, dbeta, sum dx^, sum dx^ x^).  Fixed order everywhere: deterministic.
template <typename T>
__global__ void gn_bwd_partial_kernel(const T* __restrict__ x, const T* __restrict__ dy, const float* __restrict__ mr,
                                      const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ part,
                                      int HW, int C, int NV, int R, int silu) {
  constexpr int VN = TrVec<T>::N;
  extern __shared__ float red[];   // [4][R][C]
  __shared__ float smr[TR_GROUPS * 2];
  const int b = blockIdx.y, tid = threadIdx.x, cpg = C / TR_GROUPS;
  if (tid < TR_GROUPS * 2) smr[tid] = mr[(int64_t)b * TR_GROUPS * 2 + tid];
  __syncthreads();
  const int v = tid % NV, r = tid / NV;
  const int per = cdiv(HW, gridDim.x), p0 = blockIdx.x * per, p1 = min(HW, p0 + per);
  if (r < R) {
    float mean[VN], rstd[VN], ga[VN], be[VN], s0[VN], s1[VN], s2[VN], s3[VN];
#pragma unroll
    for (int i = 0; i < VN; ++i) {
      const int c = v * VN + i, g = c / cpg;
      mean[i] = smr[2 * g]; rstd[i] = smr[2 * g + 1];
      ga[i] = gamma[c]; be[i] = beta[c];
      s0[i] = s1[i] = s2[i] = s3[i] = 0.f;
    }
    for (int p = p0 + r; p < p1; p += R) {   // (two pixels per trip - four loads in flight - measured SLOWER: 63 -> 80 us average)
      const int64_t o = ((int64_t)b * HW + p) * C + v * VN;
      float xv[VN], gv[VN];
      TrVec<T>::load(x + o, xv);
      TrVec<T>::load(dy + o, gv);
#pragma unroll
      for (int i = 0; i < VN; ++i) {
        const float xh = (xv[i] - mean[i]) * rstd[i];
        float dz = gv[i];
        if (silu) dz *= tr_dsilu<sizeof(T) == 2>(ga[i] * xh + be[i]);
        const float dxh = dz * ga[i];
        s0[i] += dz * xh; s1[i] += dz; s2[i] += dxh; s3[i] += dxh * xh;
      }
    }
#pragma unroll
    for (int i = 0; i < VN; ++i) {
      const int c = v * VN + i;
      red[(0 * R + r) * C + c] = s0[i]; red[(1 * R + r) * C + c] = s1[i];
      red[(2 * R + r) * C + c] = s2[i]; red[(3 * R + r) * C + c] = s3[i];
    }
  }
  __syncthreads();
  for (int c = tid; c < C; c += blockDim.x) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (int rr = 0; rr < R; ++rr) {
      a0 += red[(0 * R + rr) * C + c]; a1 += red[(1 * R + rr) * C + c];
      a2 += red[(2 * R + rr) * C + c]; a3 += red[(3 * R + rr) * C + c];
    }
    *reinterpret_cast<float4*>(part + (((int64_t)b * gridDim.x + blockIdx.x) * C + c) * 4) = make_float4(a0, a1, a2, a3);
  }
}

// pass 2, grid (B): slices summed in order -> dgp / dbp [B][C] and the group means ab [B][32][2]
__global__ void gn_bwd_finish_kernel(const float* __restrict__ part, float* __restrict__ ab, float* __restrict__ dgp,
                                     float* __restrict__ dbp, int nsplit, int HW, int C) {
  extern __shared__ float ch[];   // [2][C]: per-channel sum dx^, sum dx^ x^
  const int b = blockIdx.x, cpg = C / TR_GROUPS;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (int sidx = 0; sidx < nsplit; ++sidx) {
      const float4 q = *reinterpret_cast<const float4*>(part + (((int64_t)b * nsplit + sidx) * C + c) * 4);
      a0 += q.x; a1 += q.y; a2 += q.z; a3 += q.w;
    }
    dgp[(int64_t)b * C + c] = a0;
    dbp[(int64_t)b * C + c] = a1;
    ch[c] = a2;
    ch[C + c] = a3;
  }
  __syncthreads();
  if (threadIdx.x < TR_GROUPS) {
    const int g = threadIdx.x;
    double a = 0.0, d = 0.0;
    for (int c = g * cpg; c < (g + 1) * cpg; ++c) { a += ch[c]; d += ch[C + c]; }
    const double n = (double)HW * cpg;
    ab[((int64_t)b * TR_GROUPS + g) * 2] = (float)(a / n);
    ab[((int64_t)b * TR_GROUPS + g) * 2 + 1] = (float)(d / n);
  }
}

// pass 3, same grid / block shape as pass 1: dx = rstd (dz gamma - mean_g(dx^) - x^ mean_g(dx^ x^)), 16-byte accesses
template <typename T>
__global__ void gn_bwd_apply_kernel(const T* __restrict__ x, const T* __restrict__ dy, const float* __restrict__ mr,
                                    const float* __restrict__ ab, const float* __restrict__ gamma, const float* __restrict__ beta,
                                    T* __restrict__ dx, int HW, int C, int NV, int R, int silu, const T* __restrict__ dres) {
  // dres: a second gradient of x (the residual / skip path of the block whose first op this GroupNorm is), added here instead of
  // by autograd's own add kernel: one more read in this pass against a read-read-write pass of its own
  constexpr int VN = TrVec<T>::N;
  __shared__ float smr[TR_GROUPS * 2], sab[TR_GROUPS * 2];
  const int b = blockIdx.y, tid = threadIdx.x, cpg = C / TR_GROUPS;
  if (tid < TR_GROUPS * 2) {
    smr[tid] = mr[(int64_t)b * TR_GROUPS * 2 + tid];
    sab[tid] = ab[(int64_t)b * TR_GROUPS * 2 + tid];
  }
  __syncthreads();
  const int v = tid % NV, r = tid / NV;
  if (r >= R) return;
  float mean[VN], rstd[VN], ga[VN], be[VN], mA[VN], mB[VN];
#pragma unroll
  for (int i = 0; i < VN; ++i) {
    const int c = v * VN + i, g = c / cpg;
    mean[i] = smr[2 * g]; rstd[i] = smr[2 * g + 1]; mA[i] = sab[2 * g]; mB[i] = sab[2 * g + 1];
    ga[i] = gamma[c]; be[i] = beta[c];
  }
  const int per = cdiv(HW, gridDim.x), p0 = blockIdx.x * per, p1 = min(HW, p0 + per);
  for (int p = p0 + r; p < p1; p += R) {
    const int64_t o = ((int64_t)b * HW + p) * C + v * VN;
    float xv[VN], gv[VN], ov[VN];
    TrVec<T>::load(x + o, xv);
    TrVec<T>::load(dy + o, gv);
#pragma unroll
    for (int i = 0; i < VN; ++i) {
      const float xh = (xv[i] - mean[i]) * rstd[i];
      float dz = gv[i];
      if (silu) dz *= tr_dsilu<sizeof(T) == 2>(ga[i] * xh + be[i]);
      ov[i] = rstd[i] * (dz * ga[i] - mA[i] - xh * mB[i]);
    }
    if (dres != nullptr) {
      float rv[VN];
      TrVec<T>::load(dres + o, rv);
#pragma unroll
      for (int i = 0; i < VN; ++i) ov[i] += rv[i];
    }
    TrVec<T>::store(dx + o, ov);
  }
}

// scratch: ab [B*64] | dgp [B*C] | dbp [B*C] | part [B * nsplit * C * 4] floats
size_t gn_backward_scratch_floats(int B, int HW, int C) { return (size_t)B * 64 + (size_t)2 * B * C + (size_t)B * gn_nsplit(B, HW) * C * 4; }

int launch_gn_backward(const void* x, const void* dy, const float* meanrstd, const float* gamma, const float* beta, void* dx,
                       float* dgamma, float* dbeta, float* scratch, int dt, int B, int HW, int C, int silu, hipStream_t s, int accumulate,
                       const void* dres) {
  const int vn = dt == DT_F32 ? 4 : 8;
  DDMI_REQUIRE(C % TR_GROUPS == 0 && C % vn == 0, "gn_backward: C must be a multiple of 32 and of the 16-byte vector");
  const int NV = C / vn;
  DDMI_REQUIRE(NV <= 1024, "gn_backward: too many channels");
  const int R = std::max(1, 256 / NV);
  const int threads = std::max(64, cdiv(NV * R, 64) * 64);
  const int ns = gn_nsplit(B, HW);
  float* ab = scratch;
  float* dgp = ab + (size_t)B * 64;
  float* dbp = dgp + (size_t)B * C;
  float* part = dbp + (size_t)B * C;
  dim3 g(ns, B);
  const size_t lds = (size_t)4 * R * C * sizeof(float);
  DDMI_REQUIRE(lds <= 64 * 1024, "gn_backward: channel count exceeds the reduction tile");
  PROF_SCOPE("gn_bwd", 0, 5.0 * (double)B * HW * C * dt_size(dt), s);
  if (dt == DT_F32) {
    hipLaunchKernelGGL(gn_bwd_partial_kernel<float>, g, dim3(threads), lds, s, (const float*)x, (const float*)dy, meanrstd, gamma, beta, part, HW, C, NV, R, silu);
    hipLaunchKernelGGL(gn_bwd_finish_kernel, dim3(B), dim3(256), (size_t)2 * C * sizeof(float), s, part, ab, dgp, dbp, ns, HW, C);
    hipLaunchKernelGGL(gn_bwd_apply_kernel<float>, g, dim3(threads), 0, s, (const float*)x, (const float*)dy, meanrstd, ab, gamma, beta, (float*)dx, HW, C, NV, R, silu, (const float*)dres);
  } else {
    hipLaunchKernelGGL(gn_bwd_partial_kernel<bf16_t>, g, dim3(threads), lds, s, (const bf16_t*)x, (const bf16_t*)dy, meanrstd, gamma, beta, part, HW, C, NV, R, silu);
    hipLaunchKernelGGL(gn_bwd_finish_kernel, dim3(B), dim3(256), (size_t)2 * C * sizeof(float), s, part, ab, dgp, dbp, ns, HW, C);
    hipLaunchKernelGGL(gn_bwd_apply_kernel<bf16_t>, g, dim3(threads), 0, s, (const bf16_t*)x, (const bf16_t*)dy, meanrstd, ab, gamma, beta, (bf16_t*)dx, HW, C, NV, R, silu, (const bf16_t*)dres);
  }
  HIP_TRY(hipGetLastError());
  DDMI_TRY(launch_sum_b(dgp, dgamma, B, C, s, accumulate));
  DDMI_TRY(launch_sum_b(dbp, dbeta, B, C, s, accumulate));
  return DDMI_OK;
}

// saved statistics of the forward: (mean, rstd) per (sample, group) from the partial sums launch_gn_stats wrote
__global__ void gn_meanrstd_kernel(const double* __restrict__ part, float* __restrict__ mr, int nsplit, int maxsplit, double cnt, float eps) {
  const int b = blockIdx.x, g = threadIdx.x;
  if (g >= TR_GROUPS) return;
  const double* p = part + (int64_t)b * maxsplit * (TR_GROUPS * 2) + g * 2;
  double s1 = 0.0, s2 = 0.0;
  for (int i = 0; i < nsplit; ++i) { s1 += p[(int64_t)i * (TR_GROUPS * 2)]; s2 += p[(int64_t)i * (TR_GROUPS * 2) + 1]; }
  const double m = s1 / cnt;
  double var = s2 / cnt - m * m;
  if (var < 0) var = 0;
  mr[((int64_t)b * TR_GROUPS + g) * 2] = (float)m;
  mr[((int64_t)b * TR_GROUPS + g) * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
}
int launch_gn_meanrstd(const double* part, float* mr, int B, int HW, int C, float eps, int nsplit, hipStream_t s) {
  hipLaunchKernelGGL(gn_meanrstd_kernel, dim3(B), dim3(64), 0, s, part, mr, nsplit, 64, (double)HW * (C / TR_GROUPS), eps);
  HIP_TRY(hipGetLastError());
  return DDMI_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// QKVAttentionLegacy backward (openaimodel.py:397-426): per (sample, head), logits s_ij = (q_i . k_j) / sqrt(d),
// p = softmax_j(s), o_i = sum_j p_ij v_j.
//   D_i = dO_i . O_i ; dP_ij = dO_i . v_j ; dS_ij = p_ij (dP_ij - D_i)
//   dQ_i = sum_j dS_ij k_j / sqrt(d) ; dK_j = sum_i dS_ij q_i / sqrt(d) ; dV_j = sum_i p_ij dO_i
// Rows of qkv: [q(d) | k(d) | v(d)] per head (the legacy per-head interleave).  One thread per query row (dq kernel) /
// key row (dkv kernel); the other side's rows are read wave-uniformly.  fp32 math; probabilities are recomputed from the
// row log-sum-exp.
// ---------------------------------------------------------------------------------------------------------------------
template <typename T, int D>
__global__ void __launch_bounds__(128) attn_bwd_dq_kernel(const T* __restrict__ qkv, const T* __restrict__ o, const T* __restrict__ dout,
                                                         T* __restrict__ dqkv, float* __restrict__ lse, float* __restrict__ dsum,
                                                         int Tn, int heads) {
  const int b = blockIdx.z, h = blockIdx.y, i = blockIdx.x * 128 + threadIdx.x;
  const int ld = heads * 3 * D, ldo = heads * D;
  const float scale = rsqrtf((float)D);
  const T* base = qkv + (int64_t)b * Tn * ld + h * 3 * D;
  const bool live = i < Tn;
  const int ii = live ? i : Tn - 1;
  float q[D], go[D], dq[D];
  float dsm = 0.f;
#pragma unroll
  for (int e = 0; e < D; ++e) {
    q[e] = tr_ld(base + (int64_t)ii * ld + e) * scale;
    go[e] = tr_ld(dout + ((int64_t)b * Tn + ii) * ldo + h * D + e);
    dsm += go[e] * tr_ld(o + ((int64_t)b * Tn + ii) * ldo + h * D + e);
    dq[e] = 0.f;
  }
  float m = -INFINITY, l = 0.f;
  for (int j = 0; j < Tn; ++j) {   // online log-sum-exp
    const T* kr = base + (int64_t)j * ld + D;
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < D; ++e) s += q[e] * tr_ld(kr + e);
    const float mn = fmaxf(m, s);
    l = l * expf(m - mn) + expf(s - mn);
    m = mn;
  }
  const float L = m + logf(l);
  for (int j = 0; j < Tn; ++j) {
    const T* kr = base + (int64_t)j * ld + D;
    const T* vr = kr + D;
    float s = 0.f, dp = 0.f;
    float kk[D];
#pragma unroll
    for (int e = 0; e < D; ++e) {
      kk[e] = tr_ld(kr + e);
      s += q[e] * kk[e];
      dp += go[e] * tr_ld(vr + e);
    }
    const float ds = expf(s - L) * (dp - dsm) * scale;
#pragma unroll
    for (int e = 0; e < D; ++e) dq[e] += ds * kk[e];
  }
  if (live) {
    lse[((int64_t)b * heads + h) * Tn + i] = L;
    dsum[((int64_t)b * heads + h) * Tn + i] = dsm;
    T* dst = dqkv + ((int64_t)b * Tn + i) * ld + h * 3 * D;
#pragma unroll
    for (int e = 0; e < D; ++e) tr_st(dst + e, dq[e]);
  }
}

template <typename T, int D>
__global__ void __launch_bounds__(128) attn_bwd_dkv_kernel(const T* __restrict__ qkv, const T* __restrict__ dout, T* __restrict__ dqkv,
                                                          const float* __restrict__ lse, const float* __restrict__ dsum, int Tn, int heads) {
  const int b = blockIdx.z, h = blockIdx.y, j = blockIdx.x * 128 + threadIdx.x;
  const int ld = heads * 3 * D, ldo = heads * D;
  const float scale = rsqrtf((float)D);
  const T* base = qkv + (int64_t)b * Tn * ld + h * 3 * D;
  const bool live = j < Tn;
  const int jj = live ? j : Tn - 1;
  float k[D], v[D], dk[D], dv[D];
#pragma unroll
  for (int e = 0; e < D; ++e) {
    k[e] = tr_ld(base + (int64_t)jj * ld + D + e);
    v[e] = tr_ld(base + (int64_t)jj * ld + 2 * D + e);
    dk[e] = dv[e] = 0.f;
  }
  const float* Lr = lse + ((int64_t)b * heads + h) * Tn;
  const float* Dr = dsum + ((int64_t)b * heads + h) * Tn;
  for (int i = 0; i < Tn; ++i) {
    const T* qr = base + (int64_t)i * ld;
    const T* gr = dout + ((int64_t)b * Tn + i) * ldo + h * D;
    float s = 0.f, dp = 0.f, qq[D], go[D];
#pragma unroll
    for (int e = 0; e < D; ++e) {
      qq[e] = tr_ld(qr + e);
      go[e] = tr_ld(gr + e);
      s += qq[e] * k[e];
      dp += go[e] * v[e];
    }
    const float p = expf(s * scale - Lr[i]);
    const float ds = p * (dp - Dr[i]) * scale;
#pragma unroll
    for (int e = 0; e < D; ++e) {
      dv[e] += p * go[e];
      dk[e] += ds * qq[e];
    }
  }
  if (live) {
    T* dst = dqkv + ((int64_t)b * Tn + j) * ld + h * 3 * D;
#pragma unroll
    for (int e = 0; e < D; ++e) {
      tr_st(dst + D + e, dk[e]);
      tr_st(dst + 2 * D + e, dv[e]);
    }
  }
}

// ---- bf16 / MFMA form (T a multiple of 64, d = 32 = ONE k-step of v_mfma_f32_16x16x32_bf16) -------------------------------
// The query (dq kernel) or the key (dkv kernel) sits on the MFMA lane: S^T = K Q^T (A = key rows from LDS, B = the wave's 16
// query rows in registers) leaves a lane with 16 scores of ITS query per 64-key tile, so the softmax statistics and
// dS = P (dP - D) are per-lane elementwise work.  The accumulation GEMMs (dQ^T += K^T dS^T etc.) take dS / P straight from
// those accumulator registers as the B operand: the k-axis of that product (keys) is consumed in the order the accumulator
// layout hands it out - position 8 lg + e + 4 (f & 1) of chunk f >> 1 is key 16 f + 4 lg + e - and the A operand (K^T, the
// transposed tile) is staged in LDS with its columns in exactly that order.  P and dS enter the products rounded to bf16.
#define AB_TILE 64
#define AB_ROWP 40   // row pitch of the [64][32] row tiles in elements (80 B): conflict-free b128 fragment reads
#define AB_TP 72     // row pitch of the transposed [32][64] tiles in elements (144 B)

__device__ __forceinline__ int ab_pos(int key) {   // column of a tile row in the transposed images (see above)
  const int f = key >> 4, lg = (key >> 2) & 3, e = key & 3;
  return (f >> 1) * 32 + 8 * lg + e + 4 * (f & 1);
}
// stage one [64][32] row tile (rows of `src` with leading dimension ld) into R (row-major, pitch AB_ROWP) and, when TR is
// non-null, its transpose into TR[d][ab_pos(row)] (pitch AB_TP).  256 threads: thread -> (row = tid >> 2, 8 channels).
__device__ __forceinline__ void ab_stage(const bf16_t* __restrict__ src, int ld, bf16_t* R, bf16_t* TR) {
  const int row = threadIdx.x >> 2, c8 = (threadIdx.x & 3) * 8;
  const bf16x8 v = *reinterpret_cast<const bf16x8*>(src + (int64_t)row * ld + c8);
  *reinterpret_cast<bf16x8*>(R + row * AB_ROWP + c8) = v;
  if (TR != nullptr) {
    const int pos = ab_pos(row);
#pragma unroll
    for (int e = 0; e < 8; ++e) TR[(c8 + e) * AB_TP + pos] = v[e];
  }
}

__global__ void __launch_bounds__(256) attn_bwd_dq_mfma_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ o,
                                                              const bf16_t* __restrict__ dout, bf16_t* __restrict__ dqkv,
                                                              float* __restrict__ lse, float* __restrict__ dsum, int Tn, int heads) {
  constexpr int D = 32;
  __shared__ __attribute__((aligned(16))) bf16_t Ks[AB_TILE * AB_ROWP], Vs[AB_TILE * AB_ROWP], KT[D * AB_TP];
  const int b = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * AB_TILE;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lr = lane & 15, lg = lane >> 4;
  const int ld = heads * 3 * D, ldo = heads * D;
  const bf16_t* base = qkv + (int64_t)b * Tn * ld + h * 3 * D;
  const int q = q0 + 16 * wave + lr;                      // this lane's query
  const float scale = 0.17677669529663687f;               // 1 / sqrt(32)
  const float sl2 = scale * 1.4426950408889634f;          // logits in the exp2 domain
  const bf16x8 qf = *reinterpret_cast<const bf16x8*>(base + (int64_t)q * ld + 8 * lg);
  const bf16x8 gof = *reinterpret_cast<const bf16x8*>(dout + ((int64_t)b * Tn + q) * ldo + h * D + 8 * lg);
  float dsm;
  {
    const bf16x8 of = *reinterpret_cast<const bf16x8*>(o + ((int64_t)b * Tn + q) * ldo + h * D + 8 * lg);
    float part = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) part += (float)gof[e] * (float)of[e];
    part += __shfl_xor(part, 16);
    part += __shfl_xor(part, 32);
    dsm = part;                                            // D_q = dO_q . O_q
  }
  const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};
  // ---- pass 1: row log-sum-exp (exp2 domain) ----------------------------------------------------------------------------
  float m = -INFINITY, l = 0.f;
  for (int k0 = 0; k0 < Tn; k0 += AB_TILE) {
    __syncthreads();
    ab_stage(base + (int64_t)k0 * ld + D, ld, Ks, nullptr);
    __syncthreads();
    float sc[16];
    float tm = m;
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      const bf16x8 kf = *reinterpret_cast<const bf16x8*>(Ks + (16 * f + lr) * AB_ROWP + 8 * lg);
      const f32x4 s4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf, zero4, 0, 0, 0);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        sc[4 * f + e] = s4[e] * sl2;
        tm = fmaxf(tm, sc[4 * f + e]);
      }
    }
    float add = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) add += exp2f(sc[j] - tm);
    l = l * exp2f(m - tm) + add;
    m = tm;
  }
  {  // combine the four lanes (lg) that share a query
#pragma unroll
    for (int sh = 16; sh <= 32; sh <<= 1) {
      const float mo = __shfl_xor(m, sh), lo = __shfl_xor(l, sh);
      const float mn = fmaxf(m, mo);
      l = l * exp2f(m - mn) + lo * exp2f(mo - mn);
      m = mn;
    }
  }
  const float L2 = m + log2f(l);                           // log2-domain log-sum-exp
  // ---- pass 2: dQ --------------------------------------------------------------------------------------------------------
  f32x4 dq[2] = {zero4, zero4};
  for (int k0 = 0; k0 < Tn; k0 += AB_TILE) {
    __syncthreads();
    ab_stage(base + (int64_t)k0 * ld + D, ld, Ks, KT);
    ab_stage(base + (int64_t)k0 * ld + 2 * D, ld, Vs, nullptr);
    __syncthreads();
    bf16x8 dsb[2];
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      const bf16x8 kf = *reinterpret_cast<const bf16x8*>(Ks + (16 * f + lr) * AB_ROWP + 8 * lg);
      const bf16x8 vf = *reinterpret_cast<const bf16x8*>(Vs + (16 * f + lr) * AB_ROWP + 8 * lg);
      const f32x4 s4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf, zero4, 0, 0, 0);
      const f32x4 p4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, gof, zero4, 0, 0, 0);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float p = exp2f(s4[e] * sl2 - L2);
        dsb[f >> 1][4 * (f & 1) + e] = (bf16_t)(p * (p4[e] - dsm) * scale);
      }
    }
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const bf16x8 kt = *reinterpret_cast<const bf16x8*>(KT + (16 * r + lr) * AB_TP + 32 * c + 8 * lg);
        dq[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kt, dsb[c], dq[r], 0, 0, 0);
      }
  }
  if (lg == 0) {
    lse[((int64_t)b * heads + h) * Tn + q] = L2 * 0.6931471805599453f;   // natural-log form (what the fp32 kernels store)
    dsum[((int64_t)b * heads + h) * Tn + q] = dsm;
  }
  bf16_t* dst = dqkv + ((int64_t)b * Tn + q) * ld + h * 3 * D;
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    bf16x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (bf16_t)dq[r][e];
    *reinterpret_cast<bf16x4*>(dst + 16 * r + 4 * lg) = v;
  }
}

__global__ void __launch_bounds__(256) attn_bwd_dkv_mfma_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dout,
                                                               bf16_t* __restrict__ dqkv, const float* __restrict__ lse,
                                                               const float* __restrict__ dsum, int Tn, int heads) {
  constexpr int D = 32;
  __shared__ __attribute__((aligned(16))) bf16_t Qs[AB_TILE * AB_ROWP], Gs[AB_TILE * AB_ROWP], QT[D * AB_TP], GT[D * AB_TP];
  __shared__ __attribute__((aligned(16))) float Ls[AB_TILE], Ds[AB_TILE];
  const int b = blockIdx.z, h = blockIdx.y, k0 = blockIdx.x * AB_TILE;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lr = lane & 15, lg = lane >> 4;
  const int ld = heads * 3 * D, ldo = heads * D;
  const bf16_t* base = qkv + (int64_t)b * Tn * ld + h * 3 * D;
  const int key = k0 + 16 * wave + lr;                    // this lane's key
  const float scale = 0.17677669529663687f, sl2 = scale * 1.4426950408889634f;
  const bf16x8 kf = *reinterpret_cast<const bf16x8*>(base + (int64_t)key * ld + D + 8 * lg);
  const bf16x8 vf = *reinterpret_cast<const bf16x8*>(base + (int64_t)key * ld + 2 * D + 8 * lg);
  const float* Lr = lse + ((int64_t)b * heads + h) * Tn;
  const float* Dr = dsum + ((int64_t)b * heads + h) * Tn;
  const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 dk[2] = {zero4, zero4}, dv[2] = {zero4, zero4};
  for (int q0 = 0; q0 < Tn; q0 += AB_TILE) {
    __syncthreads();
    ab_stage(base + (int64_t)q0 * ld, ld, Qs, QT);
    ab_stage(dout + ((int64_t)b * Tn + q0) * ldo + h * D, ldo, Gs, GT);
    if (threadIdx.x < AB_TILE) {
      Ls[threadIdx.x] = Lr[q0 + threadIdx.x] * 1.4426950408889634f;   // back to the exp2 domain
      Ds[threadIdx.x] = Dr[q0 + threadIdx.x];
    }
    __syncthreads();
    bf16x8 pb[2], dsb[2];
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      const bf16x8 qf = *reinterpret_cast<const bf16x8*>(Qs + (16 * f + lr) * AB_ROWP + 8 * lg);
      const bf16x8 gf = *reinterpret_cast<const bf16x8*>(Gs + (16 * f + lr) * AB_ROWP + 8 * lg);
      const f32x4 s4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf, kf, zero4, 0, 0, 0);   // rows = queries 16f + 4lg + e, column = key
      const f32x4 p4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gf, vf, zero4, 0, 0, 0);
      const f32x4 L4 = *reinterpret_cast<const f32x4*>(Ls + 16 * f + 4 * lg);
      const f32x4 D4 = *reinterpret_cast<const f32x4*>(Ds + 16 * f + 4 * lg);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float p = exp2f(s4[e] * sl2 - L4[e]);
        pb[f >> 1][4 * (f & 1) + e] = (bf16_t)p;
        dsb[f >> 1][4 * (f & 1) + e] = (bf16_t)(p * (p4[e] - D4[e]) * scale);
      }
    }
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const bf16x8 gt = *reinterpret_cast<const bf16x8*>(GT + (16 * r + lr) * AB_TP + 32 * c + 8 * lg);
        const bf16x8 qt = *reinterpret_cast<const bf16x8*>(QT + (16 * r + lr) * AB_TP + 32 * c + 8 * lg);
        dv[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gt, pb[c], dv[r], 0, 0, 0);
        dk[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qt, dsb[c], dk[r], 0, 0, 0);
      }
  }
  bf16_t* dst = dqkv + ((int64_t)b * Tn + key) * ld + h * 3 * D;
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    bf16x4 a, c;
#pragma unroll
    for (int e = 0; e < 4; ++e) { a[e] = (bf16_t)dk[r][e]; c[e] = (bf16_t)dv[r][e]; }
    *reinterpret_cast<bf16x4*>(dst + D + 16 * r + 4 * lg) = a;
    *reinterpret_cast<bf16x4*>(dst + 2 * D + 16 * r + 4 * lg) = c;
  }
}

// the plain (one thread per query / key row) pair for a head dim other than 32: the tri-plane UNet's AttentionBlocks (num_head_channels 64)
// and its 16-head inter-plane attention (C / 16 = 16 .. 96 channels per head, openaimodel.py:955-957) on the training path.  Correct and
// slow - O(T^2 d) VALU work per head with 4 d floats of state per thread; the MFMA kernels above and in k_attn_bwd.hip serve d = 32
template <typename T, int D>
static void attn_bwd_plain(const void* qkv, const void* out, const void* dout, void* dqkv, float* lse, float* dsum, int B, int Tn, int heads,
                           hipStream_t s) {
  dim3 g((unsigned)cdiv(Tn, 128), (unsigned)heads, (unsigned)B);
  hipLaunchKernelGGL((attn_bwd_dq_kernel<T, D>), g, dim3(128), 0, s, (const T*)qkv, (const T*)out, (const T*)dout, (T*)dqkv, lse, dsum, Tn, heads);
  hipLaunchKernelGGL((attn_bwd_dkv_kernel<T, D>), g, dim3(128), 0, s, (const T*)qkv, (const T*)dout, (T*)dqkv, lse, dsum, Tn, heads);
}

int launch_attention_legacy_backward(const void* qkv, const void* out, const void* dout, void* dqkv, float* scratch /* 2*B*heads*T */,
                                     int dt, int B, int T, int heads, int d, hipStream_t s) {
  DDMI_REQUIRE(d == 2 || d == 4 || d == 8 || d == 16 || d == 32 || d == 64 || d == 96 || d == 128,
               "attention backward: head dim 2, 4, 8, 16, 32, 64, 96 or 128");
  float* lse = scratch;
  float* dsum = scratch + (size_t)B * heads * T;
  dim3 g((unsigned)cdiv(T, 128), (unsigned)heads, (unsigned)B);
  PROF_SCOPE("attention_legacy_bwd", 2.0 * 5.0 * B * heads * (double)T * T * d, 0, s);
  static const bool no_mfma = getenv("DDMI_ATTN_BWD_VALU") != nullptr;   // diagnostics: the plain kernels in bf16 too
  if (d != 32) {
    const bool f32 = dt == DT_F32;
    switch (d) {   // (2, 4, 8: the toy tri-plane networks of the fixtures - 16 heads of a 32 .. 128-channel tensor)
      case 2: f32 ? attn_bwd_plain<float, 2>(qkv, out, dout, dqkv, lse, dsum, B, T, heads, s) : attn_bwd_plain<bf16_t, 2>(qkv, out, dout, dqkv, lse, dsum, B, T, heads, s); break;
      case 4: f32 ? attn_bwd_plain<float, 4>(qkv, out, dout, dqkv, lse, dsum, B, T, heads, s) : attn_bwd_plain<bf16_t, 4>(qkv, out, dout, dqkv, lse, dsum, B, T, heads, s); break;
      case 8: f32 ? attn_bwd_plain<float, 8>(qkv, out, dout, dqkv, lse, dsum, B, T, heads, s) : attn_bwd_plain<bf16_t, 8>(qkv, out, dout, dqkv, lse, dsum, B, T, heads, s); break;
      case 16: f32 ? attn_bwd_plain<float, 16>(qkv, out, dout, dqkv, lse, dsum, B, T, heads, s) : attn_bwd_plain<bf16_t, 16>(qkv, out, dout, dqkv, lse, dsum, B, T, heads, s); break;
      case 64: f32 ? attn_bwd_plain<float, 64>(qkv, out, dout, dqkv, lse, dsum, B, T, heads, s) : attn_bwd_plain<bf16_t, 64>(qkv, out, dout, dqkv, lse, dsum, B, T, heads, s); break;
      case 96: f32 ? attn_bwd_plain<float, 96>(qkv, out, dout, dqkv, lse, dsum, B, T, heads, s) : attn_bwd_plain<bf16_t, 96>(qkv, out, dout, dqkv, lse, dsum, B, T, heads, s); break;
      default: f32 ? attn_bwd_plain<float, 128>(qkv, out, dout, dqkv, lse, dsum, B, T, heads, s) : attn_bwd_plain<bf16_t, 128>(qkv, out, dout, dqkv, lse, dsum, B, T, heads, s); break;
    }
  } else if (dt == DT_BF16 && T % AB_TILE == 0 && !no_mfma) {
    dim3 gm((unsigned)(T / AB_TILE), (unsigned)heads, (unsigned)B);
    hipLaunchKernelGGL(attn_bwd_dq_mfma_kernel, gm, dim3(256), 0, s, (const bf16_t*)qkv, (const bf16_t*)out, (const bf16_t*)dout, (bf16_t*)dqkv, lse, dsum, T, heads);
    hipLaunchKernelGGL(attn_bwd_dkv_mfma_kernel, gm, dim3(256), 0, s, (const bf16_t*)qkv, (const bf16_t*)dout, (bf16_t*)dqkv, lse, dsum, T, heads);
  } else if (dt == DT_F32) {
    hipLaunchKernelGGL((attn_bwd_dq_kernel<float, 32>), g, dim3(128), 0, s, (const float*)qkv, (const float*)out, (const float*)dout, (float*)dqkv, lse, dsum, T, heads);
    hipLaunchKernelGGL((attn_bwd_dkv_kernel<float, 32>), g, dim3(128), 0, s, (const float*)qkv, (const float*)dout, (float*)dqkv, lse, dsum, T, heads);
  } else {
    hipLaunchKernelGGL((attn_bwd_dq_kernel<bf16_t, 32>), g, dim3(128), 0, s, (const bf16_t*)qkv, (const bf16_t*)out, (const bf16_t*)dout, (bf16_t*)dqkv, lse, dsum, T, heads);
    hipLaunchKernelGGL((attn_bwd_dkv_kernel<bf16_t, 32>), g, dim3(128), 0, s, (const bf16_t*)qkv, (const bf16_t*)dout, (bf16_t*)dqkv, lse, dsum, T, heads);
  }
  HIP_TRY(hipGetLastError());
  return DDMI_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// small fp32 linears of the time-embedding path: y = act(x) W^T + b, act = SiLU when silu_in (launch_linear_small forward)
// ---------------------------------------------------------------------------------------------------------------------
__global__ void linear_bwd_dx_kernel(const float* __restrict__ x, const float* __restrict__ W, const float* __restrict__ dy,
                                     float* __restrict__ dx, int B, int K, int N, int silu_in) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
  if (k >= K) return;
  float acc = 0.f;
  for (int n = 0; n < N; ++n) acc += dy[(int64_t)b * N + n] * W[(int64_t)n * K + k];
  if (silu_in) acc *= tr_dsilu(x[(int64_t)b * K + k]);
  dx[(int64_t)b * K + k] = acc;
}
__global__ void linear_bwd_dw_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dW,
                                     float* __restrict__ db, int B, int K, int N, int silu_in) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x, n = blockIdx.y;
  if (k >= K) return;
  float acc = 0.f, sb = 0.f;
  for (int b = 0; b < B; ++b) {
    float a = x[(int64_t)b * K + k];
    if (silu_in) a = a / (1.0f + expf(-a));
    const float g = dy[(int64_t)b * N + n];
    acc += g * a;
    sb += g;
  }
  dW[(int64_t)n * K + k] = acc;
  if (k == 0 && db) db[n] = sb;
}
int launch_linear_small_backward(const float* x, const float* W, const float* dy, float* dx, float* dW, float* db, int B, int K, int N,
                                 int silu_in, hipStream_t s) {
  if (dx) hipLaunchKernelGGL(linear_bwd_dx_kernel, dim3(cdiv(K, 128), B), dim3(128), 0, s, x, W, dy, dx, B, K, N, silu_in);
  hipLaunchKernelGGL(linear_bwd_dw_kernel, dim3(cdiv(K, 128), N), dim3(128), 0, s, x, dy, dW, db, B, K, N, silu_in);
  HIP_TRY(hipGetLastError());
  return DDMI_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// fp32 GEMM of the time-embedding path, any operand orientation:  C[m][n] = bias[n] + sum_r A[m sam + r sar] B[r sbr + n sbn]
// (forward: A = act, B = W^T; input gradient: A = dy, B = W; weight gradient: A = dy^T, B = act).  64 x 64 tile, 16-deep
// chunks through LDS, 256 threads x 4 x 4 outputs, plain fp32 FMAs in r order (the parity path's arithmetic).  grid.z = S
// slices of the reduction axis write S partial matrices (summed in slice order by sum_slices_kernel) - the input gradient of
// the batched embedding projection reduces over ~12k rows into a 64 x 896 result.
// ---------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) sgemm_kernel(const float* __restrict__ A, int64_t sam, int64_t sar, const float* __restrict__ Bm,
                                                    int64_t sbr, int64_t sbn, float* __restrict__ C, const float* __restrict__ bias, int M,
                                                    int N, int R, int rper) {
  __shared__ float As[16][68], Bs[16][68];
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
  const int r_begin = blockIdx.z * rper, r_end = min(R, r_begin + rper);
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
  const bool a_rfast = sar == 1, b_nfast = sbn == 1;
  for (int r0 = r_begin; r0 < r_end; r0 += 16) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int idx = tid + 256 * j;
      const int ar = a_rfast ? (idx & 15) : (idx >> 6), am = a_rfast ? (idx >> 4) : (idx & 63);
      As[ar][am] = (m0 + am < M && r0 + ar < r_end) ? A[(int64_t)(m0 + am) * sam + (int64_t)(r0 + ar) * sar] : 0.f;
      const int br = b_nfast ? (idx >> 6) : (idx & 15), bn = b_nfast ? (idx & 63) : (idx >> 4);
      Bs[br][bn] = (n0 + bn < N && r0 + br < r_end) ? Bm[(int64_t)(r0 + br) * sbr + (int64_t)(n0 + bn) * sbn] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float4 av = *reinterpret_cast<const float4*>(&As[r][4 * ty]);
      const float4 bv = *reinterpret_cast<const float4*>(&Bs[r][4 * tx]);
      const float a4[4] = {av.x, av.y, av.z, av.w}, b4[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a4[i], b4[j], acc[i][j]);
    }
    __syncthreads();
  }
  float* Cz = C + (int64_t)blockIdx.z * M * N;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + 4 * ty + i;
    if (m >= M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + 4 * tx + j;
      if (n < N) Cz[(int64_t)m * N + n] = acc[i][j] + ((bias && blockIdx.z == 0) ? bias[n] : 0.f);
    }
  }
}
int sgemm_splits(int M, int N, int R) {
  const int tiles = cdiv(M, 64) * cdiv(N, 64);
  return std::max(1, std::min(std::min(256 / std::max(1, tiles), R / 64), 64));
}
// slab: sgemm_splits(M, N, R) * M * N floats when that is > 1 (else unused)
int launch_sgemm(const float* A, int64_t sam, int64_t sar, const float* Bm, int64_t sbr, int64_t sbn, float* C, const float* bias, int M, int N,
                 int R, float* slab, hipStream_t s) {
  const int S = sgemm_splits(M, N, R);
  const int rper = cdiv(cdiv(R, S), 16) * 16;
  PROF_SCOPE("sgemm_f32", 2.0 * M * (double)N * R, 4.0 * ((double)M * R + (double)R * N + (double)M * N), s);
  hipLaunchKernelGGL(sgemm_kernel, dim3((unsigned)cdiv(N, 64), (unsigned)cdiv(M, 64), (unsigned)S), dim3(256), 0, s, A, sam, sar, Bm, sbr, sbn,
                     S > 1 ? slab : C, bias, M, N, R, rper);
  if (S > 1) hipLaunchKernelGGL(sum_slices_kernel, dim3((unsigned)cdiv64((int64_t)M * N, 256)), dim3(256), 0, s, slab, C, (int64_t)M * N, S);
  HIP_TRY(hipGetLastError());
  return DDMI_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// DDPM training arithmetic (ddpm.py:383-386, 409-447) and the optimizer
// ---------------------------------------------------------------------------------------------------------------------
// x_t = sqrt(abar_t) x_0 + sqrt(1 - abar_t) noise, per-sample coefficients  (q_sample)
__global__ void q_sample_kernel(const float* __restrict__ x0, const float* __restrict__ noise, const float* __restrict__ a,
                                const float* __restrict__ sg, float* __restrict__ out, int64_t per, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int b = (int)(i / per);
    out[i] = a[b] * x0[i] + sg[b] * noise[i];
  }
}
int launch_q_sample(const float* x0, const float* noise, const float* sqrt_ac, const float* sqrt_1mac, float* out, int B, int64_t per,
                    hipStream_t s) {
  const int64_t n = (int64_t)B * per;
  hipLaunchKernelGGL(q_sample_kernel, dim3(tr_blocks(n)), dim3(256), 0, s, x0, noise, sqrt_ac, sqrt_1mac, out, per, n);
  HIP_TRY(hipGetLastError());
  return DDMI_OK;
}

// eps^ = (1 - c) sqrt(1-abar_t) x_t + c model_out, c = sigmoid(logit[ch]) (get_mixed_prediction, ddpm.py:469-477);
// loss = mean((eps^ - noise)^2) (mse over [1,2,3] then over the batch: equal sizes -> the global mean, ddpm.py:436-439);
// d_model_out = c g, d_logit[ch] = sum g (model_out - m) c (1 - c), g = 2 (eps^ - noise) / N.
// grid (C, B): one block per (sample, channel) plane; partials are summed in a fixed order by eps_loss_finish_kernel.
__global__ void __launch_bounds__(256) eps_loss_kernel(const float* __restrict__ mo, const float* __restrict__ xt, const float* __restrict__ noise,
                                                      const float* __restrict__ sg, const float* __restrict__ logit, float* __restrict__ dmo,
                                                      float* __restrict__ part /* [B][C][2] */, int C, int HW, float inv_n) {
  __shared__ float r1[256], r2[256];
  const int c = blockIdx.x, b = blockIdx.y;
  const float coef = logit ? 1.0f / (1.0f + expf(-logit[c])) : 1.0f;
  const float s1 = sg[b];
  const int64_t base = ((int64_t)b * C + c) * HW;
  float sl = 0.f, sd = 0.f;
  for (int p = threadIdx.x; p < HW; p += 256) {
    const float m = s1 * xt[base + p], o = mo[base + p];
    const float e = logit ? (1.0f - coef) * m + coef * o : o;
    const float df = e - noise[base + p];
    sl += df * df;
    const float g = 2.0f * df * inv_n;
    dmo[base + p] = coef * g;
    sd += g * (o - m) * coef * (1.0f - coef);
  }
  r1[threadIdx.x] = sl; r2[threadIdx.x] = sd;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if (threadIdx.x < st) { r1[threadIdx.x] += r1[threadIdx.x + st]; r2[threadIdx.x] += r2[threadIdx.x + st]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { part[((int64_t)b * C + c) * 2] = r1[0]; part[((int64_t)b * C + c) * 2 + 1] = r2[0]; }
}
__global__ void eps_loss_finish_kernel(const float* __restrict__ part, float* __restrict__ loss, float* __restrict__ dlogit, int B, int C,
                                       float inv_n) {
  const int c = threadIdx.x;
  __shared__ double tot[1024];
  double l = 0.0;
  float d = 0.f;
  if (c < C)
    for (int b = 0; b < B; ++b) { l += part[((int64_t)b * C + c) * 2]; d += part[((int64_t)b * C + c) * 2 + 1]; }
  tot[c] = l;
  if (c < C && dlogit) dlogit[c] = d;
  __syncthreads();
  if (c == 0) {
    double s = 0.0;
    for (int i = 0; i < C; ++i) s += tot[i];
    *loss = (float)(s * inv_n);
  }
}
int launch_eps_loss(const float* model_out, const float* x_t, const float* noise, const float* sqrt_1mac, const float* logit, float* loss,
                    float* d_model_out, float* d_logit, float* scratch /* 2*B*C */, int B, int C, int HW, hipStream_t s) {
  DDMI_REQUIRE(C <= 1024, "eps_loss: at most 1024 channels");
  const float inv_n = (float)(1.0 / ((double)B * C * HW));
  hipLaunchKernelGGL(eps_loss_kernel, dim3(C, B), dim3(256), 0, s, model_out, x_t, noise, sqrt_1mac, logit, d_model_out, scratch, C, HW, inv_n);
  hipLaunchKernelGGL(eps_loss_finish_kernel, dim3(1), dim3(1024), 0, s, scratch, loss, d_logit, B, C, inv_n);
  HIP_TRY(hipGetLastError());
  return DDMI_OK;
}

// torch.optim.AdamW (decoupled weight decay, bias-corrected moments), one pass over a flat fp32 parameter slab
// one AdamW element (torch.optim.AdamW, decoupled weight decay; the same expression order in both kernels below)
__device__ __forceinline__ void adamw_one(float& p, float g, float& m, float& v, float lr, float b1, float b2, float eps, float wd, float bc1,
                                          float bc2_sqrt, float gscale) {
  const float gi = g * gscale;
  float pi = p * (1.0f - lr * wd);
  const float mi = b1 * m + (1.0f - b1) * gi;
  const float vi = b2 * v + (1.0f - b2) * gi * gi;
  m = mi; v = vi;
  const float denom = sqrtf(vi) / bc2_sqrt + eps;
  pi -= (lr / bc1) * (mi / denom);
  p = pi;
}
__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, int64_t n,
                             float lr, float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt, float gscale) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float pi = p[i], mi = m[i], vi = v[i];
    adamw_one(pi, g[i], mi, vi, lr, b1, b2, eps, wd, bc1, bc2_sqrt, gscale);
    p[i] = pi; m[i] = mi; v[i] = vi;
  }
}
// 16-byte form (n4 = n / 4 vectors; 7 x 16 bytes of traffic per vector: the step is a pure stream of 28 bytes per parameter)
__global__ void __launch_bounds__(256) adamw_vec4_kernel(float4* __restrict__ p, const float4* __restrict__ g, float4* __restrict__ m,
                                                         float4* __restrict__ v, int64_t n4, float lr, float b1, float b2, float eps, float wd,
                                                         float bc1, float bc2_sqrt, float gscale) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    float4 pi = p[i], mi = m[i], vi = v[i];
    const float4 gi = g[i];
    adamw_one(pi.x, gi.x, mi.x, vi.x, lr, b1, b2, eps, wd, bc1, bc2_sqrt, gscale);
    adamw_one(pi.y, gi.y, mi.y, vi.y, lr, b1, b2, eps, wd, bc1, bc2_sqrt, gscale);
    adamw_one(pi.z, gi.z, mi.z, vi.z, lr, b1, b2, eps, wd, bc1, bc2_sqrt, gscale);
    adamw_one(pi.w, gi.w, mi.w, vi.w, lr, b1, b2, eps, wd, bc1, bc2_sqrt, gscale);
    p[i] = pi; m[i] = mi; v[i] = vi;
  }
}
int launch_adamw(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1, float b2, float eps, float wd, int step,
                 float grad_scale, hipStream_t s) {
  DDMI_REQUIRE(step >= 1, "adamw: step counts from 1");
  const float bc1 = 1.0f - powf(b1, (float)step), bc2 = 1.0f - powf(b2, (float)step);
  PROF_SCOPE("adamw", 0, 28.0 * n, s);
  const int64_t n4 = (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0 ? n / 4 : 0;   // 16-byte body, scalar tail
  if (n4 > 0)
    hipLaunchKernelGGL(adamw_vec4_kernel, dim3(tr_blocks(n4)), dim3(256), 0, s, (float4*)p, (const float4*)g, (float4*)m, (float4*)v, n4, lr, b1, b2,
                       eps, wd, bc1, sqrtf(bc2), grad_scale);
  if (n - 4 * n4 > 0)
    hipLaunchKernelGGL(adamw_kernel, dim3(tr_blocks(n - 4 * n4)), dim3(256), 0, s, p + 4 * n4, g + 4 * n4, m + 4 * n4, v + 4 * n4, n - 4 * n4, lr, b1,
                       b2, eps, wd, bc1, sqrtf(bc2), grad_scale);
  HIP_TRY(hipGetLastError());
  return DDMI_OK;
}

// ema_pytorch's update (tools/ldm/image.py:64, 174): ema = ema * decay + param * (1 - decay)  (torch.lerp form)
__global__ void ema_kernel(float* __restrict__ ema, const float* __restrict__ p, int64_t n, float decay) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    ema[i] = ema[i] + (p[i] - ema[i]) * (1.0f - decay);
}
int launch_ema(float* ema, const float* p, int64_t n, float decay, hipStream_t s) {
  hipLaunchKernelGGL(ema_kernel, dim3(tr_blocks(n)), dim3(256), 0, s, ema, p, n, decay);
  HIP_TRY(hipGetLastError());
  return DDMI_OK;
}
