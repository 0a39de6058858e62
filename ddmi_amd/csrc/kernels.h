// Host-callable launchers of the HIP kernels (all asynchronous on `stream`, no allocation, graph-capturable).
#pragma once
#include "common.h"

// ---- fused DDPM / DDIM step (a3-a6) -------------------------------------------------------------
struct StepCoefs {
  float sqrt_one_minus_ac;  // sqrt(1 - abar_t)            -> mixing component scale (ddpm.py:476-477)
  float sqrt_recip_ac;      // sqrt(1 / abar_t)            (ddpm.py:238)
  float sqrt_recipm1_ac;    // sqrt(1 / abar_t - 1)        (ddpm.py:239)
  float coef1, coef2;       // posterior_mean_coef1/2[t]   (ddpm.py:250-251)            ANCESTRAL
  float noise_scale;        // [t != 0] * exp(0.5 * posterior_log_variance_clipped[t])  ANCESTRAL (ddpm.py:280-281)
  float sqrt_ac_next;       // sqrt(abar_next)             (ddpm.py:366)                DDIM
  float c_dir;              // sqrt(1 - abar_next - sigma^2) (ddpm.py:361)              DDIM
  float sigma;              // eta * sqrt(...)             (ddpm.py:360)                DDIM
};
enum { STEP_ANCESTRAL = 0, STEP_DDIM = 1, STEP_DDIM_LAST = 2 };

// noise source: `seeds` != null -> drawn in registers (Philox keyed per sample, stream `tag`; needs C*HW % 4 == 0),
// else `noise` != null -> read, else zero
int launch_ddpm_step(const float* x, const float* model_out, const float* noise, const float* mixing_logit,
                     StepCoefs k, int mode, int clip, float* x_out, float* x0_out, int B, int C, int HW,
                     hipStream_t stream, const uint64_t* seeds = nullptr, unsigned tag = 0);
// out[b][e] = the same normal(seeds[b], tag, e) the step kernel draws
int launch_randn_fill(float* out, const uint64_t* seeds, int B, int64_t n_per_sample, unsigned tag, hipStream_t stream);

// ---- layout / packing -----------------------------------------------------------------------------
// fp32 [B, C, ld] (positions off .. off+HW of every channel row) <-> [B, HW, C] in the network dtype; ld = 0 means HW.
// The video path keeps its three planes as token ranges of one [B, C, L] sequence (openaimodel.py:1141-1143).
int launch_nchw_to_nhwc(const float* src, void* dst, int dt, int B, int C, int HW, hipStream_t s, int ld = 0, int off = 0);
int launch_nhwc_to_nchw(const void* src, int dt, float* dst, int B, int C, int HW, hipStream_t s, int ld = 0, int off = 0);
int launch_concat(const void* a, int Ca, const void* b, int Cb, void* out, int dt, int64_t npix, hipStream_t s);
int launch_pack_conv_weight(const float* w_oihw, void* packed, int dt, int Cout, int Cin, int KH, int KW, hipStream_t s);
// bf16x3 numerics: activations fp32 [npix][C] -> bf16 [npix][xh | xl | xh]; weights fp32 OIHW -> bf16 [Cout][tap][wh | wh | wl]
int launch_split_x3(const float* x, void* out, int64_t npix, int C, hipStream_t s);
int launch_pack_conv_weight_x3(const float* w_oihw, void* packed, int Cout, int Cin, int KH, int KW, hipStream_t s);
// bf16 packed [Cout][9][Cin] -> parity form [4][Cout][4][Cin] of (x2 nearest upsample -> 3x3 conv); ConvArgs.ups = 3
int launch_pack_conv_up4(const void* packed9, void* out, int Cout, int Cin, hipStream_t s);
// the same from the raw fp32 [Cout][Cin][3][3] tensor: the tap sums are rounded to bf16 once
int launch_pack_conv_up4_f32(const float* w_oihw, void* out, int Cout, int Cin, hipStream_t s);
int launch_cast_f32(const float* src, void* dst, int dt, int64_t n, hipStream_t s);
// planes [P*Bn][HW][C] <-> channels [Bn][HW][P*C] (reverse=1: channels -> planes); tri-plane decoders
int launch_planes_channels(const void* planes, void* chans, int dt, int Bn, int HW, int C, int P, int reverse, hipStream_t s);
int launch_upsample2x(const void* src, void* dst, int dt, int B, int H, int W, int C, hipStream_t s);

// ---- time embedding + small dense layers (fp32) ----------------------------------------------------
int launch_timestep_embedding(const int64_t* t, const float* freqs, float* out, int B, int dim, hipStream_t s);
// y[b][n] = bias[n] + sum_k act(x[b][k]) * W[n][k];  act = SiLU when silu_in
int launch_linear_small(const float* x, int x_stride, const float* W, const float* bias, float* y, int y_stride,
                        int B, int K, int N, int silu_in, hipStream_t s);

// ---- GroupNorm(32) ----------------------------------------------------------------------------------
// stats: scratch of GN_STATS_DOUBLES(B) doubles ([B][nseg <= 3][64 splits][32 groups][sum, sumsq]); no zero-fill needed.
// A sample's statistics may span up to three tensors ("segments": the video path normalises the xy | xt | yt tokens
// of a sample jointly, autoencoder_unet.py:1066-1071): run launch_gn_stats once per segment (same B, seg = 0..nseg-1),
// then launch_gn_apply on each tensor with the HW of every segment in `seg_hw`.
#define GN_STATS_DOUBLES(B) ((size_t)(B) * 3 * 64 * 64)
// x2 != null: the input is the channel concatenation [x (C1 channels) | x2 (C - C1)] of two tensors, read in place
// (replaces th.cat before a ResBlock's in_layers, openaimodel.py:787); the output is one C-channel tensor.
int launch_gn_stats(const void* x, int dt, double* stats, int B, int HW, int C, hipStream_t s, int seg = 0, int nseg = 1,
                    const void* x2 = nullptr, int C1 = 0);
// GroupNorm statistics from the per-channel sums the producing convolutions wrote (ConvArgs::chstat, `rows` pixels per block);
// cs2 / rows2 / C1: second source of a channel-concatenated input (C1 = channels of the first), else null.  Same `stats` layout
// as launch_gn_stats with one segment.
int launch_gn_chstat_finish(const float* cs1, int rows1, const float* cs2, int rows2, int C1, double* stats, int B, int HW, int C,
                            hipStream_t s);
// apply pass that reduces the producers' channel statistics itself (no finish launch); bf16 tensors
int launch_gn_apply_chstat(const void* x, void* y, const float* cs1, int rows1, const float* cs2, int rows2, const float* gamma,
                           const float* beta, float eps, int silu, int B, int HW, int C, hipStream_t s, const void* x2 = nullptr, int C1 = 0,
                           int frag_out = 0);
int launch_gn_apply(const void* x, void* y, int dt, const double* stats, const float* gamma, const float* beta,
                    float eps, int silu, int B, int HW, int C, hipStream_t s, const int* seg_hw = nullptr, int nseg = 1,
                    const void* x2 = nullptr, int C1 = 0, int x3_out = 0,    // x3_out: fp32 in, [xh | xl | xh] bf16 image out (bf16x3 numerics)
                    int frag_out = 0);   // bf16: output in MFMA B-fragment order [16 tokens][32 channels][lane][8] for k_attnblock.hip

bool net_small_batch_enabled();   // net.hip: the small-batch regime's switch (DDMI_SMALL_BATCH / ddmi_debug_small_batch)
// Small-batch GroupNorm32 [+ SiLU] as ONE launch (bf16 networks, B <= 16 or so): a workgroup per (sample, group) takes the statistics
// and applies them - where the two-launch form (grid = pixel splits x samples) spends two ~5 us launch floors and, at B = 1, runs on 64
// workgroups that wait for each other through the launch boundary.  The input is a bf16 NHWC tensor, optionally the channel
// concatenation [x | x2] read in place.
struct GnSmallArgs {
  const void* x;        // bf16 [B, HW, C1 or C]
  const void* x2;       // bf16 [B, HW, C - C1] or null
  int C1;
  void* y;              // bf16 [B, HW, C]
  const float* gamma;
  const float* beta;
  float eps;
  int silu;
  int B, HW, C;
};
// whether launch_gn_small serves the shape: whole groups, a (sample, group) slice that fits one workgroup's registers (32 Ki elements;
// 16 Ki for a two-source input), at most 1024 workgroups
bool gn_small_ok(int B, int HW, int C, bool two_sources);
int launch_gn_small(const GnSmallArgs& a, hipStream_t s);

// ---- implicit-GEMM convolution (k in {1,3}, stride in {1,2}, optional fused nearest x2 upsample) ------
struct ConvArgs {
  const void* x;      // [B, Hin, Win, Cin]   NHWC, dtype dt
  const void* w;      // [Cout][KH*KW*Cin]    packed (tap-major, channel-minor), dtype dt
  const float* bias;  // [Cout] or null
  const float* emb;   // [B][emb_stride] per-(sample, channel) addend or null (time-embedding projection)
  const void* res;    // [B, Hout, Wout, Cout] residual (dtype dt) or null
  void* y;            // [B, Hout, Wout, Cout] (dtype dt) or null when y_nchw is used
  float* y_nchw;      // optional fp32 NCHW output (model boundary)
  float* splitk_ws;   // fp32 [splitk][M][Cout] partial slabs when splitk > 1
  int B, Hin, Win, Cin, Hout, Wout, Cout;
  int KH, KW, stride, pad, ups;   // ups: fused nearest upsample of the input, 1 = x2 on both axes, 2 = x2 along W only,
                                  // 3 = x2 on both axes in parity form (bf16): Hin/Win = source, KH = KW = 2, pad = 1,
                                  //     w = launch_pack_conv_up4's [4][Cout][4][Cin], 2.25x fewer FLOPs than 1
  int emb_stride;
  int emb_mod;        // > 0: row of `emb` = sample % emb_mod (planes stacked along the batch axis share the embedding)
  int splitk;
  float out_scale;    // y = (acc + bias + emb + res) * out_scale
  int cfg;            // bf16 ring-kernel configuration id + 1 (0 = heuristic), see k_conv.hip kRing
  int order;          // K walk: 0 = heuristic, 1 = tap-major, 2 = channel-major (taps innermost); 3 / 4 = the same walks
                      // on the pipe kernel's paired-fragment instantiation (16-byte epilogue accesses)
  int tile_map;       // blockIdx -> tile: 0 = heuristic, -1 = row-major, 1 / 2 = XCD-contiguous m-major / n-major (k_conv.hip)
  float* chstat;      // optional: per-channel (sum, sum of squares) of the stored output per block of conv_chstat_rows() rows,
                      // [M / rows][Cout][2] fp32 - the consumer GroupNorm's statistics without a pass over the tensor
  // ResBlock skip_connection folded into the K loop (bf16 halo kernel only): y += w2 . [x2 | x3] at the centre tap, i.e. the
  // 1x1 convolution of openaimodel.py:272-275 (`self.skip_connection(x) + h`) as C2 + C3 more K columns of this 3x3 convolution
  const void* x2;     // [B, Hout, Wout, C2] NHWC or null
  const void* x3;     // [B, Hout, Wout, C3] or null: the second source of a channel-concatenated block input
  const void* w2;     // [Cout][C2 + C3] packed 1x1 weights
  int C2, C3;
  float* y32;         // bf16 kernels only, no split-K: fp32 NHWC output [B, Hout, Wout, Cout] (with bias / emb / res32 added in fp32)
  const float* res32; // fp32 residual for y32 / y_nchw outputs
  int force_ws;       // write the raw fp32 accumulators to splitk_ws even when splitk == 1 (bias / emb / res / y untouched): the
                      // bf16x3 mode finishes them in fp32 with launch_splitk_finish(..., DT_F32)
  // 1x1 convolution over the channel concatenation [x (Cin - Ccat channels) | xcat (Ccat channels)] of two NHWC tensors read in place
  // (the ResBlock's skip_connection on th.cat([h, hs.pop()], 1)); w [Cout][Cin]; persistent 1x1 kernel only (conv_cat_ok)
  const void* xcat;
  int Ccat;
  // producer-side GroupNorm (k_conv.hip conv_epilogue_fgn): when gn_gamma is set, y receives [SiLU](GroupNorm32(conv + bias + emb))
  // instead of the raw result.  Only launches conv_fgn_ok() accepts; gn_part / gn_sync from ConvFgnScratch (one per network).
  const float* gn_gamma;
  const float* gn_beta;
  float gn_eps;
  int gn_silu;
  float* gn_part;
  unsigned* gn_sync;
  int gn_any_batch;   // fuse at any batch (tests); default: cohorts that exchange statistics only at batches that are multiples of 8 (conv_fgn_ok)
  unsigned* gn_flag_host;   // host-mapped word a timed-out hand-off also sets (Net peeks it without synchronising), or null
  int gn_no_handoff;  // only shapes whose statistics are complete inside one tile (16x16: tps == 1) may fuse: networks whose owner never
                      // looks at the hand-off's error flag (the decoders), or a device the progress argument was not made for
};
// scratch of the producer-side GroupNorm: granules [max samples][16 tiles][32 groups][4 floats] + sync words {epoch, error flag}
#define CONV_FGN_PART_FLOATS(samples) ((size_t)(samples) * 16 * 32 * 4)
#define CONV_FGN_SYNC_WORDS 4
// whether launch_conv(a) with a.gn_gamma set can run the fused epilogue (bf16 halo-kernel launch without split-K, whole channel tiles
// holding whole groups, 256-row tiles inside one sample ...); the caller runs conv + gn_apply otherwise
bool conv_fgn_ok(const ConvArgs& a, int dt);
bool conv_cat_ok(const ConvArgs& a, int dt);
// the 3x3 / stride 1 shapes the halo kernel (and with it the folded skip connection) serves
bool conv_halo_shape_ok(int H, int W);
// whether a bf16 network may fold the softmax scale into its qkv projection for head dim d (every bf16 attention kernel then takes the
// score as the base-2 exponent: launch_attention sets scale x log2 e = 1; d <= 128: the lazy tiled / resident / streaming kernels,
// 256 / 512: the streaming kernel - the video decoder's deepest inter-plane levels - or the generic one)
inline bool attention_prescale_ok(int dt, int d) { return dt == DT_BF16 && d % 8 == 0 && d <= 512; }
// bump when the meaning of (cfg, order) changes: plan files written for other tables are ignored (net.hip)
#define CONV_PLAN_VERSION 3
struct ConvPlan { int cfg = 0, order = 0, splitk = 0; float us = 0.f; };  // result of conv_autotune (0 = heuristic)
// times the candidate (configuration, K order, split-K) triples on the real buffers and returns the fastest;
// `ws`/`ws_bytes`: split-K slab scratch for the trials.  Synchronises; only called at finalize.
int conv_autotune(const ConvArgs& a, int dt, float* ws, size_t ws_bytes, ConvPlan* best);
int launch_conv(const ConvArgs& a, int dt, hipStream_t s);
// rows per block of ConvArgs::chstat for this launch, 0 when the launch cannot fill it (split-K, fp32, ragged channel tiles ...)
int conv_chstat_rows(const ConvArgs& a, int dt);
// split-K finish: y = (ws + bias + emb + res) * scale
int launch_splitk_finish(const ConvArgs& a, int dt, hipStream_t s);
int conv_pick_splitk(const ConvArgs& a, int dt);
void conv_force_tile(int bm, int bn);  // diagnostics: 0,0 = automatic
void conv_halo_half_override(int mode);  // diagnostics: halo kernel's two-taps-per-stage tail for Cin = 64 k + 32: 0 off, 1 on, -1 environment

// batched plain GEMM  C[b][m][n] = alpha * sum_k A[b][m][k] * Bm[b][n][k]  (both K-contiguous), dtype dt in/out
int launch_gemm_nt(const void* A, const void* Bm, void* C, int dt, int batch, int M, int N, int K, int64_t strideA,
                   int64_t strideB, int64_t strideC, float alpha, hipStream_t s);
// row softmax in place over the last dim, fp32 math
int launch_softmax_rows(void* x, int dt, int64_t rows, int cols, hipStream_t s);
// dst[b][c][r] = src[b][r][c]
int launch_transpose_2d(const void* src, void* dst, int dt, int batch, int R, int Cc, hipStream_t s);

// ---- attention (QKVAttentionLegacy layout, head dim 32) -------------------------------------------------
// qkv [B, T, heads*3*32] (token-major, per-head [q|k|v] channel interleave) -> out [B, T, heads*32]; out += nothing
// general form: per (sample, head) softmax(q k^T * scale) v over the concatenation of up to three row ranges
struct AttnArgs {
  const void* qkv[3];  // per segment: row 0 of sample 0 of a [B * seg_len][ld_qkv] matrix holding q, k and v
  void* out[3];        // per segment: [B * seg_len][ld_out]
  int seg_len[3];      // tokens per sample in each segment (unused segments 0)
  int ld_qkv, ld_out;  // elements per token row
  int q_off, k_off, v_off;  // element offset of head 0's q / k / v inside a row
  int hs_qkv, hs_out;  // head stride inside a row
  int d;               // head dim (1 .. 512)
  int heads, B;
  float scale;
  int vec_ok;          // filled by launch_attention
  int legacy_name;     // profiler label only
  int q_prescaled;     // q already carries softmax scale x log2 e (folded into the qkv projection's q rows, bf16 networks): see
                       // attention_v2_kernel LAZY
  int xcd_map;         // filled by the bf16 launcher: workgroup -> (sample, head, query tile) order (k_attention.hip)
  int x3;              // fp32 tensors of a DDMI_PARITY_BF16X3 handle: both matrix products as three bf16 MFMA products (attention_x3_kernel)
  float* lse;          // optional [B][heads][N]: natural-log row log-sum-exp of the scaled scores, for the training backward
                       // (written by attention_v2_kernel only: launch_attention refuses it on the other paths)
};
int launch_attention(const AttnArgs& a, int dt, hipStream_t s);
// whether the bf16 streaming kernel (k_attention.hip attention_stream_kernel) serves head dim d at N tokens, diagnostics switches included
bool attention_stream_shape_ok(int d, int N);
int launch_attention_legacy(const void* qkv, void* out, int dt, int B, int T, int heads, int d, hipStream_t s, float* lse = nullptr,
                            float scale = 0.f);   // scale > 0: instead of 1 / sqrt(d) (a head of fewer channels zero-padded to d)
// k_attn_bwd.hip: one-pass backward from the forward's log-sum-exp (bf16, d = 32, T a multiple of 64)
int launch_attention_legacy_backward_lse(const void* qkv, const void* out, const void* dout, void* dqkv, const float* lse, float* dsum,
                                         int B, int T, int heads, hipStream_t s, float scale = 0.f);   // scale: as launch_attention_legacy
inline bool attention_bwd_lse_ok(int dt, int T, int d) { return dt == DT_BF16 && d == 32 && T % 64 == 0; }

// ---- fused plane sampling + coordinate MLP (image) ------------------------------------------------------
struct Mlp2dLayer { const void* w; const float* bias; int K; };  // w [256][K] dtype dt (folded), bias [256] fp32
struct Mlp2dArgs {
  const float* planes[3];  // [B,64,P,P] NCHW fp32 (as returned by decode)
  int plane_hw[3];
  const float* coords;     // [N,2] (x,y) normalised
  const void* w;           // packed folded weights, see mlp2d.hip
  const float* fb;         // packed fp32 biases / folded constants
  const float* noise;      // [12][B][N] or null
  const uint64_t* seeds;   // [B] per-sample Philox seeds or null: NoiseInjection drawn in the kernel when noise == null
  const float* noise_w;    // [12]
  float* rgb;              // [B,3,N]
  int B, N;
};
int launch_mlp2d(const Mlp2dArgs& a, int dt, hipStream_t s);
size_t mlp2d_weight_elems();
size_t mlp2d_fb_elems();
// bf16 handles keep their folded weights in MFMA fragment order: row-major [rows][K] -> [rows/16][K/32][64 lanes][8]
void mlp2d_pack_fragment_order(const float* src_row_major, float* dst);
// ... and the activation gains folded into matrices / constants (k_mlp2d.hip); applied BEFORE the fragment packing
void mlp2d_fold_gains(float* W_row_major, float* fb);

// ---- fused tri-plane sampling + occupancy MLP (MLP3D) -----------------------------------------------------
struct Mlp3dArgs {
  const void* planes[9];   // index plane*3 + scale, plane order (xy, yz, xz), NHWC in the storage dtype
  int plane_hw[3];
  const float* pts;        // [B][N][3]
  const void* w;
  const float* fb;
  float* logits;           // [B][N]
  int B, N;
};
int launch_mlp3d(const Mlp3dArgs& a, int dt, hipStream_t s);
size_t mlp3d_weight_elems();
size_t mlp3d_fb_elems();
void mlp3d_pack_fragment_order(const float* src_row_major, float* dst);   // bf16 handles (k_mlp3d.hip, mlp_fc2.h)

// ---- fused tri-plane sampling + video MLP (MLPVideo) -------------------------------------------------------
struct MlpVArgs {
  const void* planes[9];   // index type*3 + scale, type order (xy, yt, xt), NHWC in the storage dtype
  int plane_hw[3];         // P per scale: xy planes [P][P], yt / xt planes [frames][P]
  int frames;
  const float *cxy, *cyt, *cxt;  // coordinate grids [2][H][W], [2][T][H], [2][T][W]
  const void* w;
  const float* fb;
  float* out;              // [B][3][T][H][W]
  int B, T, H, W;
};
int launch_mlpv(const MlpVArgs& a, int dt, hipStream_t s);
size_t mlpv_weight_elems();
size_t mlpv_fb_elems();
void mlpv_pack_fragment_order(const float* src_row_major, float* dst);    // bf16 handles (k_mlpv.hip, mlp_fc2.h)

// ---- LDM training step (k_train.hip): what the backward pass needs beyond the forward kernels -------------------------
int launch_pack_dgrad_weight(const float* w_oihw, void* out, int dt, int Co, int Ci, int k, hipStream_t s);   // -> [Ci][k*k flipped][Co]
int launch_zero_insert2x(const void* dy, void* u, int dt, int B, int Ho, int Wo, int C, hipStream_t s);
int launch_sumpool2x(const void* up, void* dx, int dt, int B, int H, int W, int C, hipStream_t s);
int launch_im2col_t(const void* x, void* out, int dt, int B, int H, int W, int Ci, int Ho, int Wo, int k, int stride, int pad, int ups,
                    int64_t ldp, hipStream_t s);   // out [(tap * Ci + ci)][ldp >= B * Ho * Wo], tail columns zero
int launch_wgrad_finish(const float* G, float* dw, int Co, int Ci, int kk, int accumulate, hipStream_t s);
int sum_hw_slices(int B, int HW, int C);
int launch_sum_hw(const void* dy, float* out, float* part, int dt, int B, int HW, int C, hipStream_t s);   // out [B][C]; part: slices*B*C floats
int launch_sum_b(const float* in, float* out, int B, int C, hipStream_t s, int accumulate = 0);   // accumulate: out += the sum
int launch_gn_meanrstd(const double* part, float* mr, int B, int HW, int C, float eps, int nsplit, hipStream_t s);
int gn_nsplit(int B, int HW);   // the split count launch_gn_stats uses (k_groupnorm.hip)
size_t gn_backward_scratch_floats(int B, int HW, int C);
int launch_gn_backward(const void* x, const void* dy, const float* meanrstd, const float* gamma, const float* beta, void* dx,
                       float* dgamma, float* dbeta, float* scratch, int dt, int B, int HW, int C, int silu, hipStream_t s, int accumulate = 0,
                       const void* dres = nullptr);
int launch_cat2(const void* a, const void* b, void* out, int dt, int64_t rows, int Ca, int Cb, hipStream_t s, int split);
int launch_attention_legacy_backward(const void* qkv, const void* out, const void* dout, void* dqkv, float* scratch, int dt, int B,
                                     int T, int heads, int d, hipStream_t s);
int launch_linear_small_backward(const float* x, const float* W, const float* dy, float* dx, float* dW, float* db, int B, int K, int N,
                                 int silu_in, hipStream_t s);
int launch_q_sample(const float* x0, const float* noise, const float* sqrt_ac, const float* sqrt_1mac, float* out, int B, int64_t per,
                    hipStream_t s);
int launch_eps_loss(const float* model_out, const float* x_t, const float* noise, const float* sqrt_1mac, const float* logit, float* loss,
                    float* d_model_out, float* d_logit, float* scratch, int B, int C, int HW, hipStream_t s);
int launch_adamw(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1, float b2, float eps, float wd, int step,
                 float grad_scale, hipStream_t s);
int launch_ema(float* ema, const float* p, int64_t n, float decay, hipStream_t s);
// wgrad as a TN implicit GEMM with transposing LDS reads (bf16, 3x3 / 1x1 stride 1, W in {8,16,32,64}); slab = splits*Co*Ci*k*k floats
bool wgrad_tn_ok(int dt, int B, int H, int W, int Ci, int Co, int k, int stride, int ups);
int wgrad_tn_splits(int B, int H, int W, int Ci, int Co, int k);
int launch_wgrad_tn(const void* dy, const void* x, float* dw, float* slab, int B, int H, int W, int Ci, int Co, int k, hipStream_t s,
                    int accumulate = 0, int ups = 0);   // H, W: dY's grid; ups: x is [H/2][W/2], upsampled x2 by the convolution
int launch_pack_fwd_weight(const float* w, void* out, int dt, int Co, int Ci, int k, hipStream_t s);
// fp32 GEMM with strided operands: C[m][n] = bias[n] + sum_r A[m sam + r sar] B[r sbr + n sbn]; slab = sgemm_splits * M * N floats
int sgemm_splits(int M, int N, int R);
int launch_sgemm(const float* A, int64_t sam, int64_t sar, const float* Bm, int64_t sbr, int64_t sbn, float* C, const float* bias, int M, int N,
                 int R, float* slab, hipStream_t s);

// ---- fused AttentionBlock body (k_attnblock.hip): GroupNorm output -> qkv projection -> attention in one launch -----------------
// bf16, head dim 32, QKVAttentionLegacy layout, N = H W in {64, 256, 1024}, C a multiple of 64
bool attnblock_shape_ok(int dt, int N, int C, int d, int max_batch);
size_t attnblock_weight_bytes(int C, int heads);
void attnblock_pack_weights(const float* w_qkv, const float* b_qkv, int C, int heads, uint16_t* wp, float* bp);
// x: the normalised activations in FRAGMENT order (launch_gn_apply frag_out) -> out [B N][C] (attention output, head h at channels [32 h, 32 h + 32)); q rows of the packed
// projection carry softmax scale x log2 e
// diagnostics: 16 x 64-bit words per workgroup (start, projection done, wave 0 done, all done, then each wave's end), null = off
void attnblock_debug_ts(unsigned long long* buf);
int launch_attnblock_qkv_attn(const void* x, const void* wp, const float* bp, void* out, int B, int N, int C, int heads, hipStream_t s);

// ---- on-box peak microbenchmarks (k_microbench.hip): the rates bench.py quotes its roofline fractions against besides the nominal ones ----
int bench_peak_mfma(double seconds_target, double* tflops, double* ms_per_launch);   // pure v_mfma_f32_16x16x32_bf16, every CU, 2 waves / SIMD
int bench_peak_copy(double gib, double* tb_per_s);                                    // 16-byte-per-lane copy: (read + written) bytes / s
