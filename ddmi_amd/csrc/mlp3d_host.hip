// Host side of the fused occupancy MLP: packs the MLP3D state dict (models/d2c_vae/mlp.py:69-80, blocks.py:673-705)
// into the layout k_mlp3d.hip streams.  No per-call folding is needed (no style modulation in this MLP).
#include <cstring>
#include "models.h"
#include "prof.h"

Mlp3d::~Mlp3d() {
  for (void* p : dev_allocs)
    if (p) (void)hipFree(p);
}

static int m3_rows_h(int i) { return i == 0 ? 64 : i == 11 ? 16 : 256; }
static int m3_k_h(int i) { return i <= 2 ? 64 : (i == 3 || i == 5 || i == 6 || i == 8) ? 320 : 256; }
static size_t m3_off_h(int i) {
  size_t o = 0;
  for (int j = 0; j < i; ++j) o += (size_t)m3_rows_h(j) * m3_k_h(j);
  return o;
}

int mlp3d_finalize(Mlp3d& m) {
  DDMI_REQUIRE(m.cfg.in_ch == 3 && m.cfg.out_ch == 1 && m.cfg.ch == 256 && m.cfg.latent_dim == 64,
               "mlp3d: only mlpconfig {in_ch 3, out_ch 1, ch 256, latent_dim 64} is supported");
  struct Ent { const char* key; int mat; int rows, cols; };
  const Ent mats[] = {
      {"net_res1.fc_0.weight", 0, 64, 64},    {"net_res1.fc_1.weight", 1, 256, 64},  {"net_res1.shortcut.weight", 2, 256, 64},
      {"net_res2.fc_0.weight", 3, 256, 320},  {"net_res2.fc_1.weight", 4, 256, 256}, {"net_res2.shortcut.weight", 5, 256, 320},
      {"net_res3.fc_0.weight", 6, 256, 320},  {"net_res3.fc_1.weight", 7, 256, 256}, {"net_res3.shortcut.weight", 8, 256, 320},
      {"net_res4.fc_0.weight", 9, 256, 256},  {"net_res4.fc_1.weight", 10, 256, 256}, {"net_out.weight", 11, 1, 256}};
  std::vector<float> W(mlp3d_weight_elems(), 0.f), fb(mlp3d_fb_elems(), 0.f);
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
    DDMI_REQUIRE(e.cols == m3_k_h(e.mat), "mlp3d: matrix width mismatch");
    std::memcpy(W.data() + m3_off_h(e.mat), t->data.data(), (size_t)e.rows * e.cols * sizeof(float));
  }
  const struct { const char* key; int off, n; } biases[] = {
      {"net_res1.fc_0.bias", 0, 64},          {"net_res1.fc_1.bias", 64, 256},        {"net_res2.fc_0.bias", 64 + 256, 256},
      {"net_res2.fc_1.bias", 64 + 512, 256},  {"net_res3.fc_0.bias", 64 + 768, 256},  {"net_res3.fc_1.bias", 64 + 1024, 256},
      {"net_res4.fc_0.bias", 64 + 1280, 256}, {"net_res4.fc_1.bias", 64 + 1536, 256}};
  for (auto& b : biases) {
    const HostTensor* t;
    DDMI_TRY(get(b.key, b.n, &t));
    std::memcpy(fb.data() + b.off, t->data.data(), (size_t)b.n * sizeof(float));
  }
  const HostTensor *wp, *bp, *bo;
  DDMI_TRY(get("net_p.weight", 256 * 3, &wp));
  DDMI_TRY(get("net_p.bias", 256, &bp));
  DDMI_TRY(get("net_out.bias", 1, &bo));
  const int WP = 64 + 7 * 256, BP = WP + 1024, BOUT = BP + 256;
  for (int n = 0; n < 256; ++n)
    for (int c = 0; c < 3; ++c) fb[WP + n * 4 + c] = wp->data[n * 3 + c];
  std::memcpy(fb.data() + BP, bp->data.data(), 256 * sizeof(float));
  fb[BOUT] = bo->data[0];

  if (m.dt == DT_BF16) {  // the bf16 kernel streams MFMA fragments
    std::vector<float> Wf(W.size());
    mlp3d_pack_fragment_order(W.data(), Wf.data());
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
  if (rc != DDMI_OK || e1 != hipSuccess || e2 != hipSuccess) return ddmi_fail(DDMI_ERR_HIP, "mlp3d: weight upload failed");
  HIP_TRY(hipMemcpy(m.fb_dev, fb.data(), fb.size() * sizeof(float), hipMemcpyHostToDevice));
  m.host.clear();
  m.finalized = true;
  return DDMI_OK;
}

int mlp3d_forward(Mlp3d& m, const void* const* planes, const int* plane_hw, int layout, const float* pts, float* logits,
                  int B, int N, hipStream_t s) {
  if (!m.finalized) return ddmi_fail(DDMI_ERR_STATE, "mlp3d not finalized");
  DDMI_REQUIRE(planes && plane_hw && pts && logits && B > 0 && N > 0, "mlp3d_forward: bad arguments");
  const void* pl[9];
  std::vector<void*> tmp;
  for (int i = 0; i < 9; ++i) pl[i] = planes[i];
  if (layout == DDMI_LAYOUT_NCHW_F32) {  // diagnostic path: reference-layout planes, converted here (allocates)
    for (int i = 0; i < 9; ++i) {
      int hw = plane_hw[i % 3];
      void* t = nullptr;
      HIP_TRY(hipMalloc(&t, (size_t)B * 64 * hw * hw * dt_size(m.dt)));
      tmp.push_back(t);
      DDMI_TRY(launch_nchw_to_nhwc((const float*)planes[i], t, m.dt, B, 64, hw * hw, s));
      pl[i] = t;
    }
  } else {
    DDMI_REQUIRE(layout == DDMI_LAYOUT_NHWC_NATIVE, "mlp3d_forward: bad layout");
  }
  prof_begin_forward();
  Mlp3dArgs a{};
  for (int i = 0; i < 9; ++i) a.planes[i] = pl[i];
  for (int i = 0; i < 3; ++i) a.plane_hw[i] = plane_hw[i];
  a.pts = pts; a.w = m.w_dev; a.fb = m.fb_dev; a.logits = logits; a.B = B; a.N = N;
  int rc = launch_mlp3d(a, m.dt, s);
  if (!tmp.empty()) {
    (void)hipStreamSynchronize(s);
    for (void* t : tmp) (void)hipFree(t);
  }
  return rc;
}
