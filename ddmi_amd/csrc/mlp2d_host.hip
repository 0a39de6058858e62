// Host side of the fused image coordinate-MLP: pre-folding of the StyleGAN2 modulated 1x1 convs.
//
// The style vector of models/d2c_vae/mlp.py:46-47 depends only on the scalar scale injection `si`, so the
// per-sample modulated / demodulated weights of blocks.py:242-250 are batch-constant: they are computed once per
// distinct `si` here (fp32, ~1 MMAC) and uploaded in the layout k_mlp2d.hip streams.
#include <cmath>
#include <cstring>
#include "models.h"
#include "prof.h"

Mlp2d::~Mlp2d() {
  for (void* p : dev_allocs)
    if (p) (void)hipFree(p);
}

static const HostTensor* need(const Mlp2d& m, const std::string& key, std::string* missing) {
  auto it = m.host.find(key);
  if (it == m.host.end()) {
    if (missing->empty()) *missing = key;
    return nullptr;
  }
  return &it->second;
}

static const char* kBlocks[4] = {"net_res1.", "net_res2.", "net_res3.", "net_res4."};
static const char* kConvs[3] = {"conv1.", "conv2.", "conv3."};

int mlp2d_finalize(Mlp2d& m) {
  DDMI_REQUIRE(m.cfg.in_ch == 2 && m.cfg.out_ch == 3 && m.cfg.ch == 256 && m.cfg.latent_dim == 64,
               "mlp2d: only mlpconfig {in_ch 2, out_ch 3, ch 256, latent_dim 64} is supported");
  std::string missing;
  for (const char* k : {"time_mlp.1.weight", "time_mlp.1.bias", "time_mlp.3.weight", "time_mlp.3.bias", "torgb.bias",
                        "torgb.conv.weight", "torgb.conv.modulation.weight", "torgb.conv.modulation.bias"})
    need(m, k, &missing);
  std::vector<float> nw(12, 0.f);
  for (int b = 0; b < 4; ++b) {
    for (int c = 0; c < 3; ++c) {
      std::string p = std::string(kBlocks[b]) + kConvs[c];
      need(m, p + "conv.weight", &missing);
      need(m, p + "conv.modulation.weight", &missing);
      need(m, p + "conv.modulation.bias", &missing);
      need(m, p + "activate.bias", &missing);
      const HostTensor* n = need(m, p + "noise.weight", &missing);
      if (n) nw[b * 3 + c] = n->data[0];
    }
    if (b < 3) need(m, std::string(kBlocks[b]) + "skip.0.weight", &missing);
  }
  if (!missing.empty()) return ddmi_fail(DDMI_ERR_MISSING_WEIGHT, "missing weight '" + missing + "'");
  m.noise_nonzero = false;
  for (float v : nw) m.noise_nonzero |= (v != 0.f);
  void* p = nullptr;
  HIP_TRY(hipMalloc(&p, mlp2d_weight_elems() * dt_size(m.dt)));
  m.dev_allocs.push_back(p);
  m.w_dev = p;
  HIP_TRY(hipMalloc(&p, mlp2d_fb_elems() * sizeof(float)));
  m.dev_allocs.push_back(p);
  m.fb_dev = (float*)p;
  HIP_TRY(hipMalloc(&p, 12 * sizeof(float)));
  m.dev_allocs.push_back(p);
  m.noise_w_dev = (float*)p;
  HIP_TRY(hipMemcpy(m.noise_w_dev, nw.data(), 12 * sizeof(float), hipMemcpyHostToDevice));
  m.finalized = true;
  m.have_cache = false;
  return DDMI_OK;
}

static inline float gelu_exact(float x) { return 0.5f * x * (1.0f + std::erf(x * 0.70710678118654752f)); }

// W' [out][in] of one ModulatedConv2d (blocks.py:242-250, kernel 1x1) for the given style
static void modulated(const Mlp2d& m, const std::string& pre, const std::vector<float>& style, bool demod,
                      std::vector<float>* out, int* out_ch, int* in_ch) {
  const HostTensor& w = m.host.at(pre + "weight");             // [1, out, in, 1, 1]
  const HostTensor& mw = m.host.at(pre + "modulation.weight");  // [in, style_dim]
  const HostTensor& mb = m.host.at(pre + "modulation.bias");    // [in]
  int oc = (int)w.shape[1], ic = (int)w.shape[2], sd = (int)mw.shape[1];
  std::vector<float> s(ic);
  float mscale = 1.0f / std::sqrt((float)sd);  // EqualLinear scale, lr_mul = 1 (blocks.py:155)
  for (int i = 0; i < ic; ++i) {
    float acc = 0.f;
    for (int j = 0; j < sd; ++j) acc += style[j] * (mw.data[(size_t)i * sd + j] * mscale);
    s[i] = acc + mb.data[i];
  }
  float scale = 1.0f / std::sqrt((float)ic);  // fan_in = in * k^2, k = 1 (blocks.py:224-225)
  out->assign((size_t)oc * ic, 0.f);
  for (int o = 0; o < oc; ++o) {
    float ss = 0.f;
    for (int i = 0; i < ic; ++i) {
      float v = scale * w.data[(size_t)o * ic + i] * s[i];
      (*out)[(size_t)o * ic + i] = v;
      ss += v * v;
    }
    if (demod) {
      float d = 1.0f / std::sqrt(ss + 1e-8f);
      for (int i = 0; i < ic; ++i) (*out)[(size_t)o * ic + i] *= d;
    }
  }
  *out_ch = oc;
  *in_ch = ic;
}

// offsets of the packed matrices (must mirror mlp_w_off in k_mlp2d.hip)
static int w_k(int i) { return (i == 0 || i == 1) ? 64 : (i == 4 || i == 5 || i == 8 || i == 9) ? 320 : 256; }
static int w_rows(int i) { return i == 15 ? 16 : 256; }
static size_t w_off(int i) {
  size_t o = 0;
  for (int j = 0; j < i; ++j) o += (size_t)w_rows(j) * w_k(j);
  return o;
}

static int fold_for_si(Mlp2d& m, float si, hipStream_t s) {
  // ---- style = time_mlp(si): SinusoidalPosEmb(64) -> Linear -> GELU -> Linear (mlp.py:18-25, blocks.py:11-23) ----
  const int dim = m.cfg.ch / 4, half = dim / 2, ch = m.cfg.ch;
  std::vector<float> e(dim), h1(ch), style(ch);
  float step = (float)(std::log(10000.0) / (half - 1));
  for (int i = 0; i < half; ++i) {
    float f = std::exp((float)i * -step);
    float a = si * f;
    e[i] = std::sin(a);
    e[half + i] = std::cos(a);
  }
  const HostTensor &w1 = m.host.at("time_mlp.1.weight"), &b1 = m.host.at("time_mlp.1.bias");
  const HostTensor &w3 = m.host.at("time_mlp.3.weight"), &b3 = m.host.at("time_mlp.3.bias");
  for (int o = 0; o < ch; ++o) {
    float acc = 0.f;
    for (int i = 0; i < dim; ++i) acc += e[i] * w1.data[(size_t)o * dim + i];
    h1[o] = gelu_exact(acc + b1.data[o]);
  }
  for (int o = 0; o < ch; ++o) {
    float acc = 0.f;
    for (int i = 0; i < ch; ++i) acc += h1[i] * w3.data[(size_t)o * ch + i];
    style[o] = acc + b3.data[o];
  }

  std::vector<float> W(mlp2d_weight_elems(), 0.f), fb(mlp2d_fb_elems(), 0.f);
  // matrix index of (block, conv): see k_mlp2d.hip
  const int conv_mat[4][3] = {{0, 2, 3}, {4, 6, 7}, {8, 10, 11}, {12, 13, 14}};
  const int skip_mat[3] = {1, 5, 9};
  for (int b = 0; b < 4; ++b) {
    for (int c = 0; c < 3; ++c) {
      std::string pre = std::string(kBlocks[b]) + kConvs[c];
      std::vector<float> wm;
      int oc, ic;
      modulated(m, pre + "conv.", style, true, &wm, &oc, &ic);
      const HostTensor& ab = m.host.at(pre + "activate.bias");
      int mat = conv_mat[b][c], K = w_k(mat);
      // input channel order of the reference: block1 [f(64), si, si]; block2/3 conv1 [x(256), f(64), si, si]
      DDMI_REQUIRE(ic == K || ic == K + 2, "mlp2d: unexpected ModulatedConv2d width at " + pre);
      float* dst = W.data() + w_off(mat);
      for (int o = 0; o < 256; ++o) {
        std::memcpy(dst + (size_t)o * K, wm.data() + (size_t)o * ic, K * sizeof(float));
        float cst = 0.f;
        if (ic == K + 2) cst = si * wm[(size_t)o * ic + K] + si * wm[(size_t)o * ic + K + 1];
        fb[(b * 3 + c) * 256 + o] = ab.data[o] + cst;
      }
    }
    if (b < 3) {
      const HostTensor& ws = m.host.at(std::string(kBlocks[b]) + "skip.0.weight");  // [256, in, 1, 1]
      int ic = (int)ws.shape[1], mat = skip_mat[b], K = w_k(mat);
      DDMI_REQUIRE(ic == K + 2, "mlp2d: unexpected skip width");
      float sc = 1.0f / std::sqrt((float)ic);  // EqualConv2d scale (blocks.py:111)
      float* dst = W.data() + w_off(mat);
      for (int o = 0; o < 256; ++o) {
        for (int i = 0; i < K; ++i) dst[(size_t)o * K + i] = ws.data[(size_t)o * ic + i] * sc;
        fb[12 * 256 + b * 256 + o] = si * (ws.data[(size_t)o * ic + K] * sc) + si * (ws.data[(size_t)o * ic + K + 1] * sc);
      }
    }
  }
  {
    std::vector<float> wm;
    int oc, ic;
    modulated(m, "torgb.conv.", style, false, &wm, &oc, &ic);  // demodulate=False (blocks.py:398)
    DDMI_REQUIRE(oc == 3 && ic == 256, "mlp2d: unexpected torgb shape");
    std::memcpy(W.data() + w_off(15), wm.data(), (size_t)3 * 256 * sizeof(float));
    const HostTensor& rb = m.host.at("torgb.bias");
    for (int c = 0; c < 3; ++c) fb[15 * 256 + c] = rb.data[c];
  }
  if (m.dt == DT_BF16) {  // the bf16 kernel streams MFMA fragments: [16-row group][k step][lane][8]
    mlp2d_fold_gains(W.data(), fb.data());
    std::vector<float> Wf(W.size());
    mlp2d_pack_fragment_order(W.data(), Wf.data());
    W.swap(Wf);
  }
  // upload (fp32 staging -> cast to the network dtype on device)
  float* stage = nullptr;
  HIP_TRY(hipMalloc((void**)&stage, W.size() * sizeof(float)));
  hipError_t e1 = hipMemcpy(stage, W.data(), W.size() * sizeof(float), hipMemcpyHostToDevice);
  int rc = e1 == hipSuccess ? launch_cast_f32(stage, m.w_dev, m.dt, (int64_t)W.size(), s) : DDMI_ERR_HIP;
  hipError_t e2 = hipStreamSynchronize(s);
  (void)hipFree(stage);
  if (rc != DDMI_OK || e1 != hipSuccess || e2 != hipSuccess) return ddmi_fail(DDMI_ERR_HIP, "mlp2d: weight upload failed");
  HIP_TRY(hipMemcpy(m.fb_dev, fb.data(), fb.size() * sizeof(float), hipMemcpyHostToDevice));
  m.cached_si = si;
  m.have_cache = true;
  return DDMI_OK;
}

int mlp2d_forward(Mlp2d& m, const void* const* planes, const int* plane_hw, int layout, const float* coords, float si,
                  const float* noise, const uint64_t* seeds, float* rgb, int B, int N, hipStream_t s) {
  if (!m.finalized) return ddmi_fail(DDMI_ERR_STATE, "mlp2d not finalized");
  DDMI_REQUIRE(planes && plane_hw && coords && rgb && B > 0 && N > 0, "mlp2d_forward: bad arguments");
  // NoiseInjection (blocks.py:286-297) adds weight * N(0,1) on every call; a checkpoint with non-zero noise.weight must
  // not silently lose that term
  if (m.noise_nonzero && !noise && !seeds)
    return ddmi_fail(DDMI_ERR_INVALID, "mlp2d_forward: this MLP has non-zero NoiseInjection weights - pass the 12 draws "
                                       "(noise) or per-sample seeds (ddmi_hdpe_mlp2d_rng)");
  if (!m.have_cache || m.cached_si != si) DDMI_TRY(fold_for_si(m, si, s));  // once per distinct si (synchronises)
  const void* pl[3] = {planes[0], planes[1], planes[2]};
  std::vector<void*> tmp;
  if (layout == DDMI_LAYOUT_NCHW_F32) {
    // diagnostic path: convert reference-layout planes to channels-last in the handle's dtype (allocates)
    for (int i = 0; i < 3; ++i) {
      void* t = nullptr;
      size_t bytes = (size_t)B * 64 * plane_hw[i] * plane_hw[i] * dt_size(m.dt);
      HIP_TRY(hipMalloc(&t, bytes));
      tmp.push_back(t);
      DDMI_TRY(launch_nchw_to_nhwc((const float*)planes[i], t, m.dt, B, 64, plane_hw[i] * plane_hw[i], s));
      pl[i] = t;
    }
  } else {
    DDMI_REQUIRE(layout == DDMI_LAYOUT_NHWC_NATIVE, "mlp2d_forward: bad layout");
  }
  prof_begin_forward();
  Mlp2dArgs a{};
  for (int i = 0; i < 3; ++i) {
    a.planes[i] = (const float*)pl[i];
    a.plane_hw[i] = plane_hw[i];
  }
  a.coords = coords; a.w = m.w_dev; a.fb = m.fb_dev;
  a.noise = (noise && m.noise_nonzero) ? noise : nullptr;
  a.seeds = (!a.noise && seeds && m.noise_nonzero) ? seeds : nullptr;
  a.noise_w = m.noise_w_dev;
  a.rgb = rgb; a.B = B; a.N = N;
  int rc = launch_mlp2d(a, m.dt, s);
  if (!tmp.empty()) {
    (void)hipStreamSynchronize(s);
    for (void* t : tmp) (void)hipFree(t);
  }
  return rc;
}
