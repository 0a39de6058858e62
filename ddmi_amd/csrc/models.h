// Handles behind the C ABI (see include/ddmi_hip.h).
#pragma once
#include "net.h"

struct UNet {
  ddmi_unet_cfg cfg{};
  int size1 = 0, size2 = 0, size3 = 0;   // tri-plane (video) variant: xy = [size1,size2], xt = [size3,size1], yt = [size3,size2]
  Net net;
  int emb_dim = 0;
  int emb_total = 0;             // sum of ResBlock out channels: width of the concatenated emb projection
  bool uniform_t = false;        // caller promises t[0] == t[b] for all b: the embedding path runs for one row
  std::vector<float> emb_w_host, emb_b_host;
  float *freqs = nullptr, *te_w0 = nullptr, *te_b0 = nullptr, *te_w1 = nullptr, *te_b1 = nullptr;
  float *emb_w = nullptr, *emb_b = nullptr;
  float* tbuf = nullptr;         // [B][mc] temb | [B][E] | [B][E] | [B][emb_total]
  // optional hipGraph replay of the whole forward (ddmi_unet_set_graph): ~900 launches of a small-batch forward cost
  // more host + dispatch time than GPU time; a captured graph is keyed by (B, x, t, out, uniform_t) - a sampling loop
  // ping-pongs between two x buffers, so two graphs serve it
  // ... and by whether the fused-GroupNorm kernels were the ones captured (`fuse`: Net::fgn_fusing() x the global mode): a graph captured
  // with them is never replayed after the fusion went off (ddmi_debug_gn_fuse, or a hand-off timeout: unet_async_status drops them all)
  struct GraphEntry { int B; const void* x; const void* t; void* out; bool uniform; int fuse; hipGraphExec_t exec; };
  bool use_graph = false;
  std::vector<GraphEntry> graphs;
  ~UNet();
};
int unet_finalize(UNet& u);
// weight bundles of the blocks shared by the image and video networks (created once, emitted per tensor)
struct ResW { int gn1 = -1, c1 = -1, gn2 = -1, c2 = -1, cs = -1, cs2 = -1, emb_off = -1; float eps = 1e-5f;
              int cs_full = -1;      // cs_full: the un-split 1x1 skip_connection (folded into c2's launch where the shape allows)
              int emb_rep = 1; };    // 4: the embedding rows are repeated for the four pixels of a 2 x 2 image (Net::op_conv, dense form)
struct AttnW { int gn = -1, cqkv = -1, cp = -1, heads = 1, head_dim = 0, legacy = 0; float eps = 1e-5f, scale = 1.f;
               int q_prescaled = 0;      // the projection's q rows carry scale x log2 e (bf16 networks, k_attention.hip LAZY)
               int fused = -1; };        // Net::attnblocks index: qkv projection + attention in one launch where the shape allows
// split > 0: the input will be the channel concat of a `split`-channel tensor and another one (emit_resblock_cat)
// emb_rep = 4: the block will only run on a 2 x 2 image of a bf16 network (Net::px4_ok): its embedding rows are appended four times
int make_unet_resblock(UNet& u, const std::string& pre, int cin, int cout, ResW* w, int split = 0, int emb_rep = 1);
int emit_resblock_cat(Net& n, const ResW& w, int x1, int x2, int* out);
int make_vae_resblock(Net& n, const std::string& pre, int cin, int cout, ResW* w);
int emit_resblock(Net& n, const ResW& w, int x, int* out);
// n_tokens > 0: the block only ever runs on H W = n_tokens (lets a fused block skip the separate qkv convolution's weights)
int make_unet_attnblock(UNet& u, const std::string& pre, int ch, AttnW* w, int n_tokens = 0);
bool attnblock_fused_enabled(int n_tokens, int max_samples = 0);   // max_samples > 0: the small-batch rule applies (net.h)
void attnblock_fused_override(int mode);      // -1: environment, 0: never, 1: wherever the shape allows (networks built afterwards)
//   // DDMI_ATTN_FUSED (0 = off) / DDMI_ATTN_FUSED_N (comma list of token counts)
int make_qkv_attn(Net& n, const std::string& pre, int ch, int heads, bool expand, AttnW* w);
int emit_attn_single(Net& n, const AttnW& w, int x, int* out);
int emit_attn_joint(Net& n, const AttnW& w, int xy, int tp, int* out_xy, int* out_tp);
int unet_time_embed_setup(UNet& u);   // time_embed.* + sinusoid frequencies
int unet_finish(UNet& u);             // concatenated emb projections, scratch, plan()
int unet3_finalize(UNet& u);          // UNetModel_Triplane topology (models_video.hip)
int unet_forward(UNet& u, const float* x, const int64_t* t, float* eps_out, int B, hipStream_t s);
// Net::async_status() of the UNet's network (SYNCHRONISES); on a hand-off timeout the captured graphs are destroyed too (they hold the
// fused kernels) - the next forward runs, and captures, the separate GroupNorm passes
int unet_async_status(UNet& u, const hipStream_t* stream = nullptr);

struct Decoder2d {
  ddmi_decoder2d_cfg cfg{};
  Net net;
  int n_planes = 0;
  int export_ops[8] = {0};
  int plane_hw[8] = {0};
};
int decoder2d_finalize(Decoder2d& d);
int decoder2d_forward(Decoder2d& d, const float* z, void* const* planes, int layout, int B, hipStream_t s);

struct Decoder3d {           // tri-plane decoder (occupancy): same config block as the 2-D decoder + inter_attn_resolutions
  ddmi_decoder2d_cfg cfg{};
  int inter_attn_resolutions[8] = {0};
  int n_inter = 0;
  Net net;
  int n_scales = 0;
  int export_ops[3][4] = {{0}};
  int plane_hw[4] = {0};
};
int decoder3d_finalize(Decoder3d& d);
int decoder3d_forward(Decoder3d& d, const float* z, void* const* planes, int layout, int B, hipStream_t s);

struct Mlp2d {
  ddmi_mlp2d_cfg cfg{};
  int dt = DT_F32;
  bool finalized = false;
  std::map<std::string, HostTensor> host;
  // folded per-`si` weights are rebuilt when si changes (style depends only on si: mlp.py:46-47)
  float cached_si = -1.f;
  bool have_cache = false;
  void* w_dev = nullptr;      // packed folded weights (network dtype)
  float* fb_dev = nullptr;    // biases / folded constants (fp32)
  float* noise_w_dev = nullptr;
  bool noise_nonzero = false;
  std::vector<void*> dev_allocs;
  ~Mlp2d();
};
int mlp2d_finalize(Mlp2d& m);
int mlp2d_forward(Mlp2d& m, const void* const* planes, const int* plane_hw, int layout, const float* coords, float si,
                  const float* noise, const uint64_t* seeds, float* rgb, int B, int N, hipStream_t s);

struct Mlp3d {
  ddmi_mlp2d_cfg cfg{};   // in_ch 3, out_ch 1, ch 256, latent_dim 64
  int dt = DT_F32;
  bool finalized = false;
  std::map<std::string, HostTensor> host;
  void* w_dev = nullptr;
  float* fb_dev = nullptr;
  std::vector<void*> dev_allocs;
  ~Mlp3d();
};
int mlp3d_finalize(Mlp3d& m);
int mlp3d_forward(Mlp3d& m, const void* const* planes, const int* plane_hw, int layout, const float* pts, float* logits,
                  int B, int N, hipStream_t s);

struct DecoderV {            // video decoder (VITAutoencoder.decode + VideoDecoder_light), models_video.hip
  ddmi_decoderv_cfg cfg{};
  Net net;
  int n_scales = 0, out_ch = 0, seq_len = 0;
  int export_ops[3][4] = {{0}};
  int plane_h[2][4] = {{0}}, plane_w[2][4] = {{0}};   // [0] xy planes, [1] yt / xt planes
};
int decoderv_finalize(DecoderV& d);
int decoderv_forward(DecoderV& d, const float* z, void* const* planes, int layout, int B, hipStream_t s);

struct MlpV {
  ddmi_mlp2d_cfg cfg{};   // in_ch 2, out_ch 3, ch 256, latent_dim 64
  int dt = DT_F32;
  bool finalized = false;
  std::map<std::string, HostTensor> host;
  void* w_dev = nullptr;
  float* fb_dev = nullptr;
  std::vector<void*> dev_allocs;
  ~MlpV();
};
int mlpv_finalize(MlpV& m);
int mlpv_forward(MlpV& m, const void* const* planes, const int* plane_hw, int frames, int layout, const float* cxy,
                 const float* cyt, const float* cxt, float* out, int B, int T, int H, int W, hipStream_t s);
