// C ABI of the LDM training step (include/ddmi_hip.h, "Training"): stateless ops on caller-owned NHWC tensors in the
// handle's storage dtype + fp32 master parameters, composed from the forward kernels and k_train.hip.  The handle only owns
// a scratch arena (grown on demand: the first call of a larger shape synchronises the stream once).
#include <cmath>
#include <cstdlib>
#include <vector>
#include "kernels.h"
#include "prof.h"

struct ddmi_train {
  int dt = DT_F32;
  unsigned char* ws = nullptr;
  size_t ws_bytes = 0;
  float* freq = nullptr;   // sinusoidal-embedding frequency table of the last `dim` asked for (device, lives with the handle)
  int freq_dim = 0;
  int acc_next = 0;        // ddmi_train_accumulate_next: parameter-gradient outputs of the NEXT backward call are added in place
  int generation = 0;      // bumped whenever the scratch arena moves: captured hipGraphs hold pointers into the old one
};

namespace {
struct Carve {   // sequential carving of the scratch arena, 256-byte aligned
  ddmi_train* h;
  size_t off = 0;
  void* take(size_t bytes) {
    void* p = h->ws + off;
    off += (bytes + 255) & ~(size_t)255;
    return p;
  }
};
int reserve(ddmi_train* h, size_t bytes, hipStream_t s) {
  bytes += 4096;
  if (bytes <= h->ws_bytes) return DDMI_OK;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(s, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone)
    return ddmi_fail(DDMI_ERR_STATE, "training scratch arena would have to grow inside a stream capture: run the shape eagerly once first");
  HIP_TRY(hipStreamSynchronize(s));   // nothing in flight may still use the old arena
  ++h->generation;                    // graphs captured so far replay into freed memory: the host drops them (ddmi_train_arena_generation)
  if (h->ws) (void)hipFree(h->ws);
  h->ws = nullptr;
  h->ws_bytes = 0;
  size_t want = bytes + bytes / 4;
  HIP_TRY(hipMalloc((void**)&h->ws, want));
  h->ws_bytes = want;
  return DDMI_OK;
}
size_t al(size_t b) { return (b + 255) & ~(size_t)255; }
}  // namespace

extern "C" {

int ddmi_train_create(int numerics, ddmi_train** out) {
  DDMI_REQUIRE(out, "train_create: null argument");
  DDMI_REQUIRE(numerics == DDMI_PARITY_FP32 || numerics == DDMI_PERF_BF16, "train_create: bad numerics");
  ddmi_train* h = new (std::nothrow) ddmi_train();
  if (!h) return ddmi_fail(DDMI_ERR_HIP, "out of memory");
  h->dt = numerics == DDMI_PARITY_FP32 ? DT_F32 : DT_BF16;
  *out = h;
  return DDMI_OK;
}
int ddmi_train_destroy(ddmi_train* h) {
  if (h) {
    if (h->ws) (void)hipFree(h->ws);
    if (h->freq) (void)hipFree(h->freq);
    delete h;
  }
  return DDMI_OK;
}

int ddmi_train_arena_generation(ddmi_train* h, int* generation) {
  DDMI_REQUIRE(h && generation, "train_arena_generation: null argument");
  *generation = h->generation;
  return DDMI_OK;
}

int ddmi_train_accumulate_next(ddmi_train* h, int mask) {
  DDMI_REQUIRE(h && mask >= 0 && mask < 8, "train_accumulate_next: bad arguments");
  h->acc_next = mask;
  return DDMI_OK;
}

int ddmi_train_to_nhwc(ddmi_train* h, const float* x_nchw, void* y_nhwc, int B, int C, int HW, void* stream) {
  DDMI_REQUIRE(h && x_nchw && y_nhwc, "train_to_nhwc: null argument");
  return launch_nchw_to_nhwc(x_nchw, y_nhwc, h->dt, B, C, HW, (hipStream_t)stream);
}
int ddmi_train_to_nchw(ddmi_train* h, const void* x_nhwc, float* y_nchw, int B, int C, int HW, void* stream) {
  DDMI_REQUIRE(h && x_nhwc && y_nchw, "train_to_nchw: null argument");
  return launch_nhwc_to_nchw(x_nhwc, h->dt, y_nchw, B, C, HW, (hipStream_t)stream);
}

// 3x3 / stride-1 convolutions on W in {8..64} have a faster kernel than the heuristic's pick (the halo kernel, k_conv.hip:
// 1.2 vs 0.85 PFLOP/s); the inference networks reach it through the autotuner, the stateless training ops ask for it here.
// launch_conv falls back by itself when the shape is not halo-eligible.
static void prefer_halo(ConvArgs* c, int dt) {
  if (dt == DT_BF16 && c->KH == 1 && c->Cout % 112 == 0 && (int64_t)c->B * c->Hout * c->Wout >= 256 * 64) {
    // 1x1 (qkv / proj / skip, forward and dgrad): the pipe kernel equals or beats the heuristic's ring tile on every such shape
    // (profiles/r3_conv1x1_variants_b512.txt: 992 vs 1 072 us, 789 vs 830, 408 vs 434, 339 vs 355)
    // round 3: the persistent 1x1 kernel where the channel counts allow (4-20 % over the pipe kernel at B = 512); launch_conv
    // falls back to the pipe kernel by itself for the shapes it does not take (Cin <= 128, ragged Cout)
    c->cfg = c->Cout % 224 == 0 ? 25 : 14;   // gemm1x1<224,4,2> : pipe<256,224,4,2>
    c->order = 3;  // paired fragments
    return;
  }
  if (dt == DT_BF16 && c->KH == 1 && c->Cout % 128 == 0 && c->Cin > 128 && c->Cin % 8 == 0 && c->stride == 1 && !c->ups &&
      (int64_t)c->B * c->Hout * c->Wout >= 256 * 64) {
    // the same for the model_channels = 256 networks (the tri-plane UNet's q / k / v / proj_out and skip connections - round 6, video
    // training): the persistent kernel's 256- / 128-wide tiles instead of the heuristic's 128 x 128 ring tile (0.14 of the MFMA peak on
    // these shapes); launch_conv falls back to the pipe kernel where the persistent one does not take the launch
    c->cfg = c->Cout % 256 == 0 ? 29 : 26;   // gemm1x1<256,2,4> : gemm1x1<128,4,2>
    c->order = 3;
    return;
  }
  if (dt != DT_BF16 || c->KH != 3 || c->stride != 1 || c->ups != 0) return;
  const int p224 = cdiv(c->Cout, 224) * 224, p128 = cdiv(c->Cout, 128) * 128;
  if (p224 * 100 <= c->Cout * 115) c->cfg = 20;        // halo<224,4,2>
  else if (c->Cout % 256 == 0) c->cfg = 27;            // halo<256,2,4> (W = 16 / 32; launch_conv takes halo<128,4,2> for other planes)
  else if (p128 * 100 <= c->Cout * 115) c->cfg = 21;   // halo<128,4,2>
}

// wgrad GEMM G[N][M] over a long pixel axis: the software-pipelined 128x224 tile (pipe kernel) instead of the heuristic's ring
// tile, and a split-K that fills whole waves of 256 workgroups (M = Cout = 224 gives only 2 x N/224 tiles: with the generic
// pick 18 tiles x 16 splits = 288 workgroups = one full round + one at 12 %)
static void plan_wgrad_gemm(ConvArgs* g, int dt) {
  if (dt != DT_BF16 || g->Cout % 112 != 0) { g->splitk = conv_pick_splitk(*g, dt); return; }
  g->cfg = 15;   // pipe<128,224,2,2>
  const int tiles = cdiv(g->Win, 128) * cdiv(g->Cout, 224), stages = cdiv(g->Cin, 64);
  int best = 1;
  double best_eff = 0.0;
  for (int sk = 1; sk <= 16 && stages / sk >= 16; ++sk) {
    const double wg = (double)tiles * sk, eff = wg / (std::ceil(wg / 256.0) * 256.0);
    if (eff > best_eff + 0.03) { best = sk; best_eff = eff; }
  }
  g->splitk = best;
}

static void conv_geometry(int H, int W, int k, int stride, int ups, int* Ho, int* Wo) {
  const int hin = ups ? 2 * H : H, win = ups ? 2 * W : W, pad = k / 2;
  *Ho = (hin + 2 * pad - k) / stride + 1;
  *Wo = (win + 2 * pad - k) / stride + 1;
}

int ddmi_train_conv_fwd(ddmi_train* h, const void* x, const float* w, const float* bias, const float* emb, int emb_stride,
                        const void* res, void* y, int B, int H, int W, int Cin, int Cout, int k, int stride, int ups, void* stream) {
  DDMI_REQUIRE(h && x && w && y && (k == 1 || k == 3) && (stride == 1 || stride == 2), "train_conv_fwd: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  const int dt = h->dt;
  int Ho, Wo;
  conv_geometry(H, W, k, stride, ups, &Ho, &Wo);
  ConvArgs a{};
  a.x = x; a.bias = bias; a.emb = emb; a.emb_stride = emb_stride; a.res = res; a.y = y;
  a.B = B; a.Hin = H; a.Win = W; a.Cin = Cin; a.Hout = Ho; a.Wout = Wo; a.Cout = Cout;
  a.KH = a.KW = k; a.stride = stride; a.pad = k / 2; a.ups = ups ? 1 : 0; a.out_scale = 1.f;
  // Upsample.conv (bias only): the parity form of nearest x2 + 3x3 the inference networks use - four 2x2 convolutions on the source
  // grid with pre-summed weights (summed in fp32 from the master weights, every step), 4/9 of the FLOPs, on the 2x2 halo kernel
  static const bool no_up4 = getenv("DDMI_CONV_UP4") && atoi(getenv("DDMI_CONV_UP4")) == 0;
  if (ups && k == 3 && dt == DT_BF16 && !emb && !res && !no_up4) {
    const size_t w4 = al((size_t)16 * Cout * Cin * dt_size(dt));
    DDMI_TRY(reserve(h, w4, s));
    Carve c{h};
    void* wp = c.take(w4);
    DDMI_TRY(launch_pack_conv_up4_f32(w, wp, Cout, Cin, s));
    a.w = wp; a.ups = 3; a.KH = a.KW = 2; a.pad = 1; a.splitk = 1;
    a.cfg = Cout % 224 == 0 ? 20 : 0;   // halo<224,4,2> (2x2 instantiation); launch_conv falls back when the shape is not eligible
    return launch_conv(a, dt, s);
  }
  prefer_halo(&a, dt);
  a.splitk = conv_pick_splitk(a, dt);
  const size_t wbytes = al((size_t)Cout * Cin * k * k * dt_size(dt));
  const size_t sk = a.splitk > 1 ? al((size_t)a.splitk * B * Ho * Wo * Cout * sizeof(float)) : 0;
  DDMI_TRY(reserve(h, wbytes + sk, s));
  Carve c{h};
  void* wp = c.take(wbytes);
  DDMI_TRY(launch_pack_fwd_weight(w, wp, dt, Cout, Cin, k, s));
  a.w = wp;
  if (a.splitk > 1) a.splitk_ws = (float*)c.take(sk);
  return launch_conv(a, dt, s);
}

int ddmi_train_conv_bwd(ddmi_train* h, const void* x, const float* w, const void* dy, void* dx, float* dw, float* dbias,
                        float* demb, int B, int H, int W, int Cin, int Cout, int k, int stride, int ups, void* stream) {
  const int acc = h ? h->acc_next : 0;   // one-shot: bit 0 dw += , bit 1 dbias += ; consumed even when the call is refused below
  if (h) h->acc_next = 0;
  DDMI_REQUIRE(h && x && w && dy && dw && (k == 1 || k == 3) && (stride == 1 || stride == 2), "train_conv_bwd: bad arguments");
  DDMI_REQUIRE(!(stride == 2 && (ups || k != 3)), "train_conv_bwd: stride 2 is the 3x3 Downsample convolution");
  hipStream_t s = (hipStream_t)stream;
  const int dt = h->dt;
  const size_t es = dt_size(dt);
  int Ho, Wo;
  conv_geometry(H, W, k, stride, ups, &Ho, &Wo);
  const int64_t P = (int64_t)B * Ho * Wo;
  const int kk = k * k, N = kk * Cin;
  // wgrad works on chunks of whole samples: the transposed im2col image [N][Pc] (the GEMM's weight-side operand) has to
  // stay inside one 2 GiB buffer descriptor; chunk results are accumulated in fp32 by the finish pass
  static const double chunk_bytes = getenv("DDMI_WGRAD_CHUNK_BYTES") ? atof(getenv("DDMI_WGRAD_CHUNK_BYTES")) : 1.5e9;  // tests
  int bc = B;
  while (bc > 1 && (double)std::max(N, Cout) * ((double)bc * Ho * Wo) * es > chunk_bytes) bc = (bc + 1) / 2;
  const int64_t Pc = (int64_t)bc * Ho * Wo;
  const int WPAD = 64;   // both GEMM operands get rows of Pn + 64 elements (zeros in the tail): see launch_im2col_t
  DDMI_REQUIRE(Pc % (dt == DT_F32 ? 4 : 8) == 0 && P % (dt == DT_F32 ? 4 : 8) == 0,
               "train_conv_bwd: (samples per chunk)*Ho*Wo must be a multiple of the 16-byte vector");
  DDMI_REQUIRE((double)std::max(N, Cout) * Pc * es < 2.1e9, "train_conv_bwd: one sample's im2col image exceeds 2 GiB");
  // ---- scratch plan (the two phases reuse the arena) --------------------------------------------------------
  const int Hg = ups ? 2 * H : (stride == 2 ? 2 * Ho : H), Wg = ups ? 2 * W : (stride == 2 ? 2 * Wo : W);   // dgrad conv grid
  ConvArgs d{};   // dgrad: a stride-1 convolution of dY' with the transposed, flipped filter
  d.B = B; d.Hin = Hg; d.Win = Wg; d.Cin = Cout; d.Hout = Hg; d.Wout = Wg; d.Cout = Cin;
  d.KH = d.KW = k; d.stride = 1; d.pad = k / 2; d.ups = 0; d.out_scale = 1.f;
  prefer_halo(&d, dt);
  d.splitk = dx ? conv_pick_splitk(d, dt) : 1;
  ConvArgs g{};   // wgrad: G[n][co] = sum_p Bm[n][p] A[co][p] as a 1x1 "convolution" over rows of A (one chunk of samples)
  g.B = 1; g.Hin = 1; g.Win = Cout; g.Cin = (int)Pc + WPAD; g.Hout = 1; g.Wout = Cout; g.Cout = N;
  g.KH = g.KW = 1; g.stride = 1; g.pad = 0; g.ups = 0; g.out_scale = 1.f;
  plan_wgrad_gemm(&g, dt);
  const size_t dgrad_bytes = dx ? al((size_t)Cout * Cin * kk * es) + (stride == 2 ? al((size_t)B * Hg * Wg * Cout * es) : 0) +
                                      (ups ? al((size_t)B * Hg * Wg * Cin * es) : 0) +
                                      (d.splitk > 1 ? al((size_t)d.splitk * B * Hg * Wg * Cin * sizeof(float)) : 0)
                                : 0;
  const size_t wgrad_bytes = al((size_t)Cout * (Pc + WPAD) * es) + al((size_t)N * (Pc + WPAD) * es) + al((size_t)N * Cout * sizeof(float)) +
                             (g.splitk > 1 ? al((size_t)g.splitk * Cout * N * sizeof(float)) : 0);
  const size_t red_bytes = al((size_t)B * Cout * sizeof(float)) * (1 + (size_t)sum_hw_slices(B, Ho * Wo, Cout));
  static const bool no_tn = getenv("DDMI_WGRAD_GEMM") != nullptr;   // diagnostics: the im2col + GEMM formulation everywhere
  const bool tn = !no_tn && wgrad_tn_ok(dt, B, Ho, Wo, Cin, Cout, k, stride, ups);   // stride 1: (Ho, Wo) = the grid the conv ran on
  const size_t tn_bytes = tn ? al((size_t)wgrad_tn_splits(B, Ho, Wo, Cin, Cout, k) * Cout * Cin * kk * sizeof(float)) : 0;
  DDMI_TRY(reserve(h, std::max(dgrad_bytes, tn ? tn_bytes : wgrad_bytes) + red_bytes, s));
  // ---- bias / embedding gradients ---------------------------------------------------------------------------
  if (dbias || demb) {
    Carve c{h};
    float* per0 = (float*)c.take(al((size_t)B * Cout * sizeof(float)));
    float* part = (float*)(h->ws + al((size_t)B * Cout * sizeof(float)));
    float* per = demb ? demb : per0;
    DDMI_TRY(launch_sum_hw(dy, per, part, dt, B, Ho * Wo, Cout, s));
    if (dbias) DDMI_TRY(launch_sum_b(per, dbias, B, Cout, s, (acc >> 1) & 1));
  }
  // ---- dgrad ------------------------------------------------------------------------------------------------
  if (dx) {
    Carve c{h};
    c.take(red_bytes);
    void* wp = c.take((size_t)Cout * Cin * kk * es);
    DDMI_TRY(launch_pack_dgrad_weight(w, wp, dt, Cout, Cin, k, s));
    const void* src = dy;
    if (stride == 2) {
      void* u = c.take((size_t)B * Hg * Wg * Cout * es);
      DDMI_TRY(launch_zero_insert2x(dy, u, dt, B, Ho, Wo, Cout, s));
      src = u;
    }
    void* dst = ups ? c.take((size_t)B * Hg * Wg * Cin * es) : dx;
    d.x = src; d.w = wp; d.y = dst;
    if (d.splitk > 1) d.splitk_ws = (float*)c.take((size_t)d.splitk * B * Hg * Wg * Cin * sizeof(float));
    DDMI_TRY(launch_conv(d, dt, s));
    if (ups) DDMI_TRY(launch_sumpool2x(dst, dx, dt, B, H, W, Cin, s));
  }
  // ---- wgrad ------------------------------------------------------------------------------------------------
  if (tn) {   // TN implicit GEMM straight from dY and x (k_train.hip): no transposed copies, no im2col image
    Carve c{h};
    c.take(red_bytes);
    DDMI_TRY(launch_wgrad_tn(dy, x, dw, (float*)c.take(tn_bytes), B, Ho, Wo, Cin, Cout, k, s, acc & 1, ups));
  } else {
    Carve c{h};
    c.take(red_bytes);
    void* A = c.take((size_t)Cout * (Pc + WPAD) * es);
    void* Bm = c.take((size_t)N * (Pc + WPAD) * es);
    float* G = (float*)c.take((size_t)N * Cout * sizeof(float));
    if (g.splitk > 1) g.splitk_ws = (float*)c.take((size_t)g.splitk * Cout * N * sizeof(float));
    for (int b0 = 0; b0 < B; b0 += bc) {
      const int nb = std::min(bc, B - b0);
      const int64_t Pn = (int64_t)nb * Ho * Wo;
      const unsigned char* dyc = (const unsigned char*)dy + (size_t)b0 * Ho * Wo * Cout * es;
      const unsigned char* xc = (const unsigned char*)x + (size_t)b0 * H * W * Cin * es;
      // A[co][p] = dY[p][co]: the transpose is the k = 1 case of the transposed im2col (dY as a [nb, Ho, Wo, Cout] image)
      DDMI_TRY(launch_im2col_t(dyc, A, dt, nb, Ho, Wo, Cout, Ho, Wo, 1, 1, 0, 0, Pn + WPAD, s));
      DDMI_TRY(launch_im2col_t(xc, Bm, dt, nb, H, W, Cin, Ho, Wo, k, stride, k / 2, ups ? 1 : 0, Pn + WPAD, s));
      ConvArgs gc = g;
      gc.Cin = (int)Pn + WPAD;
      gc.x = A; gc.w = Bm; gc.y_nchw = G;
      if (nb != bc) { plan_wgrad_gemm(&gc, dt); if (gc.splitk > g.splitk) gc.splitk = g.splitk; if (gc.splitk <= 1) gc.splitk_ws = nullptr; }
      DDMI_TRY(launch_conv(gc, dt, s));
      DDMI_TRY(launch_wgrad_finish(G, dw, Cout, Cin, kk, b0 > 0 || (acc & 1), s));
    }
  }
  return DDMI_OK;
}

int ddmi_train_gn_fwd(ddmi_train* h, const void* x, const float* gamma, const float* beta, void* y, float* meanrstd, int B, int HW,
                      int C, float eps, int silu, void* stream) {
  DDMI_REQUIRE(h && x && gamma && beta && y && meanrstd, "train_gn_fwd: null argument");
  hipStream_t s = (hipStream_t)stream;
  DDMI_TRY(reserve(h, GN_STATS_DOUBLES(B) * sizeof(double), s));
  double* st = (double*)h->ws;
  DDMI_TRY(launch_gn_stats(x, h->dt, st, B, HW, C, s));
  DDMI_TRY(launch_gn_apply(x, y, h->dt, st, gamma, beta, eps, silu, B, HW, C, s));
  return launch_gn_meanrstd(st, meanrstd, B, HW, C, eps, gn_nsplit(B, HW), s);
}
int ddmi_train_gn_bwd(ddmi_train* h, const void* x, const float* gamma, const float* beta, const float* meanrstd, const void* dy,
                      void* dx, float* dgamma, float* dbeta, int B, int HW, int C, int silu, void* stream) {
  const int acc = h ? h->acc_next : 0;   // one-shot: bit 2 dgamma / dbeta += ; consumed even when the call is refused below
  if (h) h->acc_next = 0;
  DDMI_REQUIRE(h && x && gamma && beta && meanrstd && dy && dx && dgamma && dbeta, "train_gn_bwd: null argument");
  hipStream_t s = (hipStream_t)stream;
  DDMI_TRY(reserve(h, gn_backward_scratch_floats(B, HW, C) * sizeof(float), s));
  return launch_gn_backward(x, dy, meanrstd, gamma, beta, dx, dgamma, dbeta, (float*)h->ws, h->dt, B, HW, C, silu, s, (acc >> 2) & 1);
}
// the same with a second gradient of x added in the last pass: dx = GroupNorm backward(dy) + dres (dres [B,HW,C] in the handle's
// storage type, may alias nothing else; null = ddmi_train_gn_bwd).  For the block input that feeds both the norm and the block's
// residual / skip path (ResBlock, AttentionBlock): the host passes the skip path's gradient here instead of adding the two itself.
int ddmi_train_gn_bwd_res(ddmi_train* h, const void* x, const float* gamma, const float* beta, const float* meanrstd, const void* dy,
                          const void* dres, void* dx, float* dgamma, float* dbeta, int B, int HW, int C, int silu, void* stream) {
  const int acc = h ? h->acc_next : 0;
  if (h) h->acc_next = 0;
  DDMI_REQUIRE(h && x && gamma && beta && meanrstd && dy && dx && dgamma && dbeta, "train_gn_bwd_res: null argument");
  hipStream_t s = (hipStream_t)stream;
  DDMI_TRY(reserve(h, gn_backward_scratch_floats(B, HW, C) * sizeof(float), s));
  return launch_gn_backward(x, dy, meanrstd, gamma, beta, dx, dgamma, dbeta, (float*)h->ws, h->dt, B, HW, C, silu, s, (acc >> 2) & 1, dres);
}

// out [rows, Ca + Cb] = [a | b] along the channel axis of NHWC rows (rows = B*H*W); ddmi_train_split2 is the inverse (the gradient of the
// concatenation): a, b <- the two column blocks of `in`.  Storage type of the handle; Ca, Cb multiples of the 16-byte vector.
int ddmi_train_cat2(ddmi_train* h, const void* a, const void* b, void* out, int64_t rows, int Ca, int Cb, void* stream) {
  DDMI_REQUIRE(h && a && b && out && rows > 0, "train_cat2: bad arguments");
  return launch_cat2(a, b, out, h->dt, rows, Ca, Cb, (hipStream_t)stream, 0);
}
int ddmi_train_split2(ddmi_train* h, const void* in, void* a, void* b, int64_t rows, int Ca, int Cb, void* stream) {
  DDMI_REQUIRE(h && a && b && in && rows > 0, "train_split2: bad arguments");
  return launch_cat2(a, b, const_cast<void*>(in), h->dt, rows, Ca, Cb, (hipStream_t)stream, 1);
}
int ddmi_train_attn_fwd(ddmi_train* h, const void* qkv, void* out, int B, int T, int heads, int d, void* stream) {
  DDMI_REQUIRE(h && qkv && out, "train_attn_fwd: null argument");
  return launch_attention_legacy(qkv, out, h->dt, B, T, heads, d, (hipStream_t)stream);
}
// the same pair with the forward's row log-sum-exp handed over (lse [B][heads][T] fp32, caller-owned): the backward then makes one
// pass over the scores (k_attn_bwd.hip).  bf16 handles, d = 32, T a multiple of 64 (ddmi_train_attn_lse_ok); anything else is refused.
int ddmi_train_attn_lse_ok(ddmi_train* h, int T, int d) { return h && attention_bwd_lse_ok(h->dt, T, d) ? 1 : 0; }
int ddmi_train_attn_fwd_lse(ddmi_train* h, const void* qkv, void* out, float* lse, int B, int T, int heads, int d, void* stream) {
  DDMI_REQUIRE(h && qkv && out && lse, "train_attn_fwd_lse: null argument");
  DDMI_REQUIRE(attention_bwd_lse_ok(h->dt, T, d), "train_attn_fwd_lse: bf16, head dim 32, token count a multiple of 64");
  return launch_attention_legacy(qkv, out, h->dt, B, T, heads, d, (hipStream_t)stream, lse);
}
int ddmi_train_attn_bwd_lse(ddmi_train* h, const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, int B, int T,
                            int heads, int d, void* stream) {
  DDMI_REQUIRE(h && qkv && out && dout && dqkv && lse, "train_attn_bwd_lse: null argument");
  DDMI_REQUIRE(attention_bwd_lse_ok(h->dt, T, d), "train_attn_bwd_lse: bf16, head dim 32, token count a multiple of 64");
  hipStream_t s = (hipStream_t)stream;
  DDMI_TRY(reserve(h, (size_t)B * heads * T * sizeof(float), s));
  return launch_attention_legacy_backward_lse(qkv, out, dout, dqkv, lse, (float*)h->ws, B, T, heads, s);
}
// the same pair for heads of d < 32 channels ZERO-PADDED to 32 by the caller (qkv / out / dout / dqkv all in the padded layout): the scores
// are those of the d real channels, so the softmax scale is the caller's 1 / sqrt(d) instead of 1 / sqrt(32).  Lets the tri-plane UNet's
// 16-head inter-plane attention at 256 channels (d = 16, 2048 tokens: two thirds of the video training step on the plain backward
// kernels) run on the MFMA kernels at twice its own FLOPs.
int ddmi_train_attn_fwd_lse_scaled(ddmi_train* h, const void* qkv, void* out, float* lse, int B, int T, int heads, float scale, void* stream) {
  DDMI_REQUIRE(h && qkv && out && lse && scale > 0.f, "train_attn_fwd_lse_scaled: bad arguments");
  DDMI_REQUIRE(attention_bwd_lse_ok(h->dt, T, 32), "train_attn_fwd_lse_scaled: bf16, token count a multiple of 64");
  return launch_attention_legacy(qkv, out, h->dt, B, T, heads, 32, (hipStream_t)stream, lse, scale);
}
int ddmi_train_attn_bwd_lse_scaled(ddmi_train* h, const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, int B, int T,
                                   int heads, float scale, void* stream) {
  DDMI_REQUIRE(h && qkv && out && dout && dqkv && lse && scale > 0.f, "train_attn_bwd_lse_scaled: bad arguments");
  DDMI_REQUIRE(attention_bwd_lse_ok(h->dt, T, 32), "train_attn_bwd_lse_scaled: bf16, token count a multiple of 64");
  hipStream_t s = (hipStream_t)stream;
  DDMI_TRY(reserve(h, (size_t)B * heads * T * sizeof(float), s));
  return launch_attention_legacy_backward_lse(qkv, out, dout, dqkv, lse, (float*)h->ws, B, T, heads, s, scale);
}
int ddmi_train_attn_bwd(ddmi_train* h, const void* qkv, const void* out, const void* dout, void* dqkv, int B, int T, int heads, int d,
                        void* stream) {
  DDMI_REQUIRE(h && qkv && out && dout && dqkv, "train_attn_bwd: null argument");
  hipStream_t s = (hipStream_t)stream;
  DDMI_TRY(reserve(h, (size_t)2 * B * heads * T * sizeof(float), s));
  return launch_attention_legacy_backward(qkv, out, dout, dqkv, (float*)h->ws, h->dt, B, T, heads, d, s);
}

int ddmi_train_linear_fwd(ddmi_train* h, const float* x, const float* W, const float* b, float* y, int B, int K, int N, int silu_in,
                          void* stream) {
  DDMI_REQUIRE(h && x && W && y, "train_linear_fwd: null argument");
  return launch_linear_small(x, K, W, b, y, N, B, K, N, silu_in, (hipStream_t)stream);
}
int ddmi_train_linear_bwd(ddmi_train* h, const float* x, const float* W, const float* dy, float* dx, float* dW, float* db, int B, int K,
                          int N, int silu_in, void* stream) {
  DDMI_REQUIRE(h && x && W && dy && dW, "train_linear_bwd: null argument");
  return launch_linear_small_backward(x, W, dy, dx, dW, db, B, K, N, silu_in, (hipStream_t)stream);
}

int ddmi_train_matmul(ddmi_train* h, const float* A, int64_t sam, int64_t sar, const float* Bm, int64_t sbr, int64_t sbn, float* C,
                      const float* bias, int M, int N, int R, void* stream) {
  DDMI_REQUIRE(h && A && Bm && C && M > 0 && N > 0 && R > 0, "train_matmul: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  const int S = sgemm_splits(M, N, R);
  float* slab = nullptr;
  if (S > 1) {
    DDMI_TRY(reserve(h, (size_t)S * M * N * sizeof(float), s));
    slab = (float*)h->ws;
  }
  return launch_sgemm(A, sam, sar, Bm, sbr, sbn, C, bias, M, N, R, slab, s);
}

int ddmi_train_timestep_embedding(ddmi_train* h, const int64_t* t, float* out, int B, int dim, void* stream) {
  DDMI_REQUIRE(h && t && out && dim % 2 == 0, "train_timestep_embedding: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  const int half = dim / 2;
  // the table is built once per handle and dimension: no copy, no host synchronisation in the steady state (a per-step sync
  // here would serialise the host's launch queue with the device, and is not capturable in a hipGraph)
  if (h->freq && h->freq_dim == dim) return launch_timestep_embedding(t, h->freq, out, B, dim, s);
  if (h->freq) {
    HIP_TRY(hipStreamSynchronize(s));
    (void)hipFree(h->freq);
    h->freq = nullptr;
    h->freq_dim = 0;
  }
  std::vector<float> fr(half);
  const float neg_log = (float)(-std::log(10000.0));
  for (int i = 0; i < half; ++i) {   // fp32 product, then fp32 division, as torch evaluates util.py:161-163
    volatile float a = neg_log * (float)i;
    volatile float q = a / (float)half;
    fr[i] = std::exp((float)q);
  }
  HIP_TRY(hipMalloc((void**)&h->freq, half * sizeof(float)));
  HIP_TRY(hipMemcpy(h->freq, fr.data(), half * sizeof(float), hipMemcpyHostToDevice));
  h->freq_dim = dim;
  return launch_timestep_embedding(t, h->freq, out, B, dim, s);
}

int ddmi_train_q_sample(const float* x0, const float* noise, const float* sqrt_ac, const float* sqrt_1mac, float* x_t, int B,
                        int64_t per_sample, void* stream) {
  DDMI_REQUIRE(x0 && noise && sqrt_ac && sqrt_1mac && x_t && B > 0 && per_sample > 0, "train_q_sample: bad arguments");
  prof_begin_forward();   // a training step starts here: applies the profiler's sampling period (ddmi_profile_enable)
  return launch_q_sample(x0, noise, sqrt_ac, sqrt_1mac, x_t, B, per_sample, (hipStream_t)stream);
}
int ddmi_train_eps_loss(ddmi_train* h, const float* model_out, const float* x_t, const float* noise, const float* sqrt_1mac,
                        const float* mixing_logit, float* loss, float* d_model_out, float* d_logit, int B, int C, int HW, void* stream) {
  DDMI_REQUIRE(h && model_out && x_t && noise && sqrt_1mac && loss && d_model_out, "train_eps_loss: null argument");
  hipStream_t s = (hipStream_t)stream;
  DDMI_TRY(reserve(h, (size_t)2 * B * C * sizeof(float), s));
  return launch_eps_loss(model_out, x_t, noise, sqrt_1mac, mixing_logit, loss, d_model_out, d_logit, (float*)h->ws, B, C, HW, s);
}
int ddmi_train_adamw(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1, float beta2,
                     float eps, float weight_decay, int step, float grad_scale, void* stream) {
  DDMI_REQUIRE(param && grad && exp_avg && exp_avg_sq && n > 0, "train_adamw: bad arguments");
  return launch_adamw(param, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, step, grad_scale, (hipStream_t)stream);
}

int ddmi_train_ema(float* ema, const float* param, int64_t n, float decay, void* stream) {
  DDMI_REQUIRE(ema && param && n > 0, "train_ema: bad arguments");
  return launch_ema(ema, param, n, decay, (hipStream_t)stream);
}

}  // extern "C"
