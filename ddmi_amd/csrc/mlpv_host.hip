// Host side of the fused video MLP: packs the MLPVideo state dict (models/d2c_vae/mlp.py:114-126, blocks.py:673-705)
// into the layout k_mlpv.hip streams.
#include <cstring>
#include "models.h"
#include "prof.h"

MlpV::~MlpV() {
  for (void* p : dev_allocs)
    if (p) (void)hipFree(p);
}

static int mv_rows_h(int i) { return i == 0 ? 192 : i == 11 ? 16 : 256; }
static int mv_k_h(int i) { return i <= 2 ? 192 : (i == 3 || i == 5 || i == 6 || i == 8) ? 448 : 256; }
static size_t mv_off_h(int i) {
  size_t o = 0;
  for (int j = 0; j < i; ++j) o += (size_t)mv_rows_h(j) * mv_k_h(j);
  return o;
}

int mlpv_finalize(MlpV& m) {
  DDMI_REQUIRE(m.cfg.out_ch == 3 && m.cfg.ch == 256 && m.cfg.latent_dim == 64,
               "mlpv: only mlpconfig {out_ch 3, ch 256, latent_dim 64} is supported");
  struct Ent { const char* key; int mat; int rows, cols; };
  const Ent mats[] = {
      {"net_res1.fc_0.weight", 0, 192, 192},  {"net_res1.fc_1.weight", 1, 256, 192},  {"net_res1.shortcut.weight", 2, 256, 192},
      {"net_res2.fc_0.weight", 3, 256, 448},  {"net_res2.fc_1.weight", 4, 256, 256},  {"net_res2.shortcut.weight", 5, 256, 448},
      {"net_res3.fc_0.weight", 6, 256, 448},  {"net_res3.fc_1.weight", 7, 256, 256},  {"net_res3.shortcut.weight", 8, 256, 448},
      {"net_res4.fc_0.weight", 9, 256, 256},  {"net_res4.fc_1.weight", 10, 256, 256}, {"net_out.weight", 11, 3, 256}};
  std::vector<float> W(mlpv_weight_elems(), 0.f), fb(mlpv_fb_elems(), 0.f);
  auto get = [&](const char* key, int64_t numel, const HostTensor** out) -> int {
    auto it = m.host.find(key);
    if (it == m.host.end()) return ddmi_fail(DDMI_ERR_MISSING_WEIGHT, std::string("missing weight '") + key + "'");
    DDMI_REQUIRE(it->second.numel() == numel, std::string("size mismatch for ") + key);
    *out = &it->second;
    return DDMI_OK;
  };
  for (const Ent& e : mats) {
    const HostTensor* t;
    DDMI_TRY(get(e.key, (int64_t)e.rows * e.cols, &t));
    DDMI_REQUIRE(e.cols == mv_k_h(e.mat), "mlpv: matrix width mismatch");
    std::memcpy(W.data() + mv_off_h(e.mat), t->data.data(), (size_t)e.rows * e.cols * sizeof(float));
  }
  const struct { const char* key; int off, n; } biases[] = {
      {"net_res1.fc_0.bias", 0, 192},           {"net_res1.fc_1.bias", 192, 256},         {"net_res2.fc_0.bias", 192 + 256, 256},
      {"net_res2.fc_1.bias", 192 + 512, 256},   {"net_res3.fc_0.bias", 192 + 768, 256},   {"net_res3.fc_1.bias", 192 + 1024, 256},
      {"net_res4.fc_0.bias", 192 + 1280, 256},  {"net_res4.fc_1.bias", 192 + 1536, 256},  {"net_out.bias", 192 + 1792, 3}};
  for (auto& b : biases) {
    const HostTensor* t;
    DDMI_TRY(get(b.key, b.n, &t));
    std::memcpy(fb.data() + b.off, t->data.data(), (size_t)b.n * sizeof(float));
  }
  if (m.dt == DT_BF16) {  // the bf16 kernel streams MFMA fragments
    std::vector<float> Wf(W.size());
    mlpv_pack_fragment_order(W.data(), Wf.data());
    W.swap(Wf);
  }
  void* p = nullptr;
  HIP_TRY(hipMalloc(&p, W.size() * dt_size(m.dt)));
  m.dev_allocs.push_back(p);
  m.w_dev = p;
  HIP_TRY(hipMalloc(&p, fb.size() * sizeof(float)));
  m.dev_allocs.push_back(p);
  m.fb_dev = (float*)p;
  float* stage = nullptr;
  HIP_TRY(hipMalloc((void**)&stage, W.size() * sizeof(float)));
  hipError_t e1 = hipMemcpy(stage, W.data(), W.size() * sizeof(float), hipMemcpyHostToDevice);
  int rc = e1 == hipSuccess ? launch_cast_f32(stage, m.w_dev, m.dt, (int64_t)W.size(), 0) : DDMI_ERR_HIP;
  hipError_t e2 = hipDeviceSynchronize();
  (void)hipFree(stage);
  if (rc != DDMI_OK || e1 != hipSuccess || e2 != hipSuccess) return ddmi_fail(DDMI_ERR_HIP, "mlpv: weight upload failed");
  HIP_TRY(hipMemcpy(m.fb_dev, fb.data(), fb.size() * sizeof(float), hipMemcpyHostToDevice));
  m.host.clear();
  m.finalized = true;
  return DDMI_OK;
}

int mlpv_forward(MlpV& m, const void* const* planes, const int* plane_hw, int frames, int layout, const float* cxy,
                 const float* cyt, const float* cxt, float* out, int B, int T, int H, int W, hipStream_t s) {
  if (!m.finalized) return ddmi_fail(DDMI_ERR_STATE, "mlpv not finalized");
  DDMI_REQUIRE(planes && plane_hw && cxy && cyt && cxt && out && B > 0 && T > 0 && H > 0 && W > 0 && frames > 0,
               "mlpv_forward: bad arguments");
  const void* pl[9];
  std::vector<void*> tmp;
  for (int i = 0; i < 9; ++i) pl[i] = planes[i];
  if (layout == DDMI_LAYOUT_NCHW_F32) {  // diagnostic path: reference-layout planes, converted here (allocates)
    for (int i = 0; i < 9; ++i) {
      int P = plane_hw[i % 3];
      int hp = i < 3 ? P : frames;
      void* t = nullptr;
      HIP_TRY(hipMalloc(&t, (size_t)B * 64 * hp * P * dt_size(m.dt)));
      tmp.push_back(t);
      DDMI_TRY(launch_nchw_to_nhwc((const float*)planes[i], t, m.dt, B, 64, hp * P, s));
      pl[i] = t;
    }
  } else {
    DDMI_REQUIRE(layout == DDMI_LAYOUT_NHWC_NATIVE, "mlpv_forward: bad layout");
  }
  prof_begin_forward();
  MlpVArgs a{};
  for (int i = 0; i < 9; ++i) a.planes[i] = pl[i];
  for (int i = 0; i < 3; ++i) a.plane_hw[i] = plane_hw[i];
  a.frames = frames;
  a.cxy = cxy; a.cyt = cyt; a.cxt = cxt;
  a.w = m.w_dev; a.fb = m.fb_dev; a.out = out;
  a.B = B; a.T = T; a.H = H; a.W = W;
  int rc = launch_mlpv(a, m.dt, s);
  if (!tmp.empty()) {
    (void)hipStreamSynchronize(s);
    for (void* t : tmp) (void)hipFree(t);
  }
  return rc;
}
