// Handles behind the C ABI (see include/ddmi_hip.h).
#pragma once
#include "net.h"

struct UNet {
  ddmi_unet_cfg cfg{};
  Net net;
  int emb_dim = 0;
  int emb_total = 0;             // sum of ResBlock out channels: width of the concatenated emb projection
  bool uniform_t = false;        // caller promises t[0] == t[b] for all b: the embedding path runs for one row
  std::vector<float> emb_w_host, emb_b_host;
  float *freqs = nullptr, *te_w0 = nullptr, *te_b0 = nullptr, *te_w1 = nullptr, *te_b1 = nullptr;
  float *emb_w = nullptr, *emb_b = nullptr;
  float* tbuf = nullptr;         // [B][mc] temb | [B][E] | [B][E] | [B][emb_total]
};
int unet_finalize(UNet& u);
int unet_forward(UNet& u, const float* x, const int64_t* t, float* eps_out, int B, hipStream_t s);

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
                  const float* noise, float* rgb, int B, int N, hipStream_t s);

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
