// extern "C" surface of libddmi_hip.so (declared in include/ddmi_hip.h).
#include <cstring>
#include <new>
#include "models.h"
#include <cmath>
#include "prof.h"

const char* ddmi_error_cstr();

struct ddmi_unet { UNet u; };
struct ddmi_decoder2d { Decoder2d d; };
struct ddmi_mlp2d { Mlp2d m; };
struct ddmi_decoder3d { Decoder3d d; };
struct ddmi_mlp3d { Mlp3d m; };
struct ddmi_decoderv { DecoderV d; };
struct ddmi_mlpv { MlpV m; };

// DDMI_PARITY_BF16X3: fp32 tensors and fp32 kernels everywhere except the convolutions of the conv networks, which run as three
// bf16 MFMA products (Net::x3); handles without convolutions (the fused MLPs, the single-op diagnostics) take it as fp32
static int numerics_to_dt(int numerics, int* dt, bool* x3 = nullptr) {
  if (x3) *x3 = false;
  if (numerics == DDMI_PARITY_FP32) { *dt = DT_F32; return DDMI_OK; }
  if (numerics == DDMI_PERF_BF16) { *dt = DT_BF16; return DDMI_OK; }
  if (numerics == DDMI_PARITY_BF16X3) { *dt = DT_F32; if (x3) *x3 = true; return DDMI_OK; }
  return ddmi_fail(DDMI_ERR_INVALID, "numerics must be DDMI_PARITY_FP32 (0), DDMI_PERF_BF16 (1) or DDMI_PARITY_BF16X3 (2)");
}

extern "C" {

const char* ddmi_last_error(void) { return ddmi_error_cstr(); }
const char* ddmi_version(void) { return "ddmi_hip 0.1 (gfx950)"; }

int ddmi_init(int device) {
  int n = 0;
  HIP_TRY(hipGetDeviceCount(&n));
  DDMI_REQUIRE(device >= 0 && device < n, "ddmi_init: no such device");
  HIP_TRY(hipSetDevice(device));
  hipDeviceProp_t p;
  HIP_TRY(hipGetDeviceProperties(&p, device));
  if (std::strncmp(p.gcnArchName, "gfx950", 6) != 0)
    return ddmi_fail(DDMI_ERR_INVALID, std::string("ddmi_init: this library is built for gfx950 only, device is ") + p.gcnArchName);
  return DDMI_OK;
}

// ---- live kernel timing (HIP events on the launch stream) -------------------------------------------------
int ddmi_profile_enable(int every_n_forwards) {
  prof_enable(every_n_forwards);
  return DDMI_OK;
}
int ddmi_profile_collect(char* json_out, uint64_t capacity) {
  DDMI_REQUIRE(json_out && capacity > 2, "profile_collect: bad buffer");
  std::string j = prof_collect_json();
  if (j.size() + 1 > capacity) return ddmi_fail(DDMI_ERR_INVALID, "profile_collect: buffer too small");
  std::memcpy(json_out, j.c_str(), j.size() + 1);
  return DDMI_OK;
}

// ---- UNet -----------------------------------------------------------------------------------------------
int ddmi_unet_create(const ddmi_unet_cfg* cfg, ddmi_unet** out) {
  DDMI_REQUIRE(cfg && out, "unet_create: null argument");
  DDMI_REQUIRE(cfg->max_batch >= 1, "unet_create: max_batch must be >= 1");
  int dt;
  bool x3 = false;
  DDMI_TRY(numerics_to_dt(cfg->numerics, &dt, &x3));
  ddmi_unet* h = new (std::nothrow) ddmi_unet();
  if (!h) return ddmi_fail(DDMI_ERR_STATE, "out of host memory");
  h->u.cfg = *cfg;
  h->u.net.dt = dt;
  h->u.net.x3 = x3;
  h->u.net.max_batch = cfg->max_batch;
  *out = h;
  return DDMI_OK;
}
int ddmi_unet_load_weight(ddmi_unet* h, const char* key, const float* data, const int64_t* shape, int ndim) {
  DDMI_REQUIRE(h, "null handle");
  return h->u.net.add_host(key, data, shape, ndim);
}
int ddmi_unet_finalize(ddmi_unet* h) {
  DDMI_REQUIRE(h, "null handle");
  if (h->u.net.finalized) return ddmi_fail(DDMI_ERR_STATE, "already finalized");
  return h->u.size1 > 0 ? unet3_finalize(h->u) : unet_finalize(h->u);
}
int ddmi_unet_set_graph(ddmi_unet* h, int enabled) {
  DDMI_REQUIRE(h, "null handle");
  h->u.use_graph = enabled != 0;
  return DDMI_OK;
}
int ddmi_unet3_create(const ddmi_unet_cfg* cfg, int size1, int size2, int size3, ddmi_unet** out) {
  DDMI_REQUIRE(size1 > 0 && size2 > 0 && size3 > 0, "unet3_create: sizes must be positive");
  DDMI_TRY(ddmi_unet_create(cfg, out));
  (*out)->u.size1 = size1; (*out)->u.size2 = size2; (*out)->u.size3 = size3;
  return DDMI_OK;
}
int ddmi_unet_forward(ddmi_unet* h, const float* x, const int64_t* t, float* eps_out, int B, void* stream) {
  DDMI_REQUIRE(h, "null handle");
  return unet_forward(h->u, x, t, eps_out, B, (hipStream_t)stream);
}
int ddmi_unet_async_status(ddmi_unet* h, int* fused_gn_launches) {
  DDMI_REQUIRE(h, "null handle");
  if (fused_gn_launches) *fused_gn_launches = h->u.net.fgn_launches;
  return unet_async_status(h->u);
}
int ddmi_unet_async_status_on(ddmi_unet* h, void* stream, int* fused_gn_launches) {
  DDMI_REQUIRE(h, "null handle");
  if (fused_gn_launches) *fused_gn_launches = h->u.net.fgn_launches;
  const hipStream_t s = (hipStream_t)stream;
  return unet_async_status(h->u, &s);
}
int ddmi_unet_debug_inject_gn_timeout(ddmi_unet* h) {
  DDMI_REQUIRE(h, "null handle");
  h->u.net.debug_inject_fgn_timeout();
  return DDMI_OK;
}
int ddmi_unet_set_uniform_t(ddmi_unet* h, int enabled) {
  DDMI_REQUIRE(h, "null handle");
  h->u.uniform_t = enabled != 0;
  return DDMI_OK;
}
int ddmi_unet_destroy(ddmi_unet* h) {
  delete h;
  return DDMI_OK;
}
int ddmi_unet_memory(ddmi_unet* h, uint64_t out[3]) {
  DDMI_REQUIRE(h && out, "null argument");
  out[0] = h->u.net.weight_bytes;
  out[1] = h->u.net.arena_bytes;
  out[2] = h->u.net.splitk_bytes;
  return DDMI_OK;
}

// ---- sampler step ---------------------------------------------------------------------------------------
static int ddpm_step_impl(const float* x, const float* model_out, const float* noise, const uint64_t* seeds, unsigned tag,
                          const float* mixing_logit, const ddmi_step_coefs* c, int mode, int clip_denoised, float* x_out,
                          float* x0_out, int B, int C, int HW, void* stream) {
  DDMI_REQUIRE(c, "ddpm_step: null coefs");
  DDMI_REQUIRE(seeds == nullptr || ((int64_t)C * HW) % 4 == 0, "ddpm_step_rng: C*HW must be a multiple of 4");
  StepCoefs k;
  k.sqrt_one_minus_ac = c->sqrt_one_minus_ac;
  k.sqrt_recip_ac = c->sqrt_recip_ac;
  k.sqrt_recipm1_ac = c->sqrt_recipm1_ac;
  k.coef1 = c->coef1;
  k.coef2 = c->coef2;
  k.noise_scale = c->noise_scale;
  k.sqrt_ac_next = c->sqrt_ac_next;
  k.c_dir = c->c_dir;
  k.sigma = c->sigma;
  return launch_ddpm_step(x, model_out, noise, mixing_logit, k, mode, clip_denoised, x_out, x0_out, B, C, HW,
                          (hipStream_t)stream, seeds, tag);
}
int ddmi_ddpm_step(const float* x, const float* model_out, const float* noise, const float* mixing_logit,
                   const ddmi_step_coefs* c, int mode, int clip_denoised, float* x_out, float* x0_out, int B, int C,
                   int HW, void* stream) {
  return ddpm_step_impl(x, model_out, noise, nullptr, 0, mixing_logit, c, mode, clip_denoised, x_out, x0_out, B, C, HW, stream);
}
int ddmi_ddpm_step_rng(const float* x, const float* model_out, const uint64_t* seeds, uint32_t tag,
                       const float* mixing_logit, const ddmi_step_coefs* c, int mode, int clip_denoised, float* x_out,
                       float* x0_out, int B, int C, int HW, void* stream) {
  DDMI_REQUIRE(seeds, "ddpm_step_rng: null seeds");
  return ddpm_step_impl(x, model_out, nullptr, seeds, tag, mixing_logit, c, mode, clip_denoised, x_out, x0_out, B, C, HW, stream);
}
int ddmi_randn_fill(float* out, const uint64_t* seeds, int B, int64_t n_per_sample, uint32_t tag, void* stream) {
  return launch_randn_fill(out, seeds, B, n_per_sample, tag, (hipStream_t)stream);
}

// ---- decoder ----------------------------------------------------------------------------------------------
int ddmi_decoder2d_create(const ddmi_decoder2d_cfg* cfg, ddmi_decoder2d** out) {
  DDMI_REQUIRE(cfg && out, "decoder2d_create: null argument");
  DDMI_REQUIRE(cfg->max_batch >= 1, "decoder2d_create: max_batch must be >= 1");
  int dt;
  bool x3 = false;
  DDMI_TRY(numerics_to_dt(cfg->numerics, &dt, &x3));
  ddmi_decoder2d* h = new (std::nothrow) ddmi_decoder2d();
  if (!h) return ddmi_fail(DDMI_ERR_STATE, "out of host memory");
  h->d.cfg = *cfg;
  h->d.net.dt = dt;
  h->d.net.x3 = x3;
  h->d.net.max_batch = cfg->max_batch;
  *out = h;
  return DDMI_OK;
}
int ddmi_decoder2d_load_weight(ddmi_decoder2d* h, const char* key, const float* data, const int64_t* shape, int ndim) {
  DDMI_REQUIRE(h, "null handle");
  return h->d.net.add_host(key, data, shape, ndim);
}
int ddmi_decoder2d_finalize(ddmi_decoder2d* h) {
  DDMI_REQUIRE(h, "null handle");
  if (h->d.net.finalized) return ddmi_fail(DDMI_ERR_STATE, "already finalized");
  return decoder2d_finalize(h->d);
}
int ddmi_decoder2d_num_planes(ddmi_decoder2d* h, int* n, int plane_hw[8]) {
  DDMI_REQUIRE(h && n && plane_hw, "null argument");
  if (!h->d.net.finalized) return ddmi_fail(DDMI_ERR_STATE, "decoder not finalized");
  *n = h->d.n_planes;
  for (int i = 0; i < h->d.n_planes; ++i) plane_hw[i] = h->d.plane_hw[i];
  return DDMI_OK;
}
int ddmi_decoder2d_forward(ddmi_decoder2d* h, const float* z, void* const* planes, int layout, int B, void* stream) {
  DDMI_REQUIRE(h, "null handle");
  return decoder2d_forward(h->d, z, planes, layout, B, (hipStream_t)stream);
}
int ddmi_decoder2d_destroy(ddmi_decoder2d* h) {
  delete h;
  return DDMI_OK;
}

// ---- fused plane sampling + MLP ---------------------------------------------------------------------------
int ddmi_mlp2d_create(const ddmi_mlp2d_cfg* cfg, ddmi_mlp2d** out) {
  DDMI_REQUIRE(cfg && out, "mlp2d_create: null argument");
  int dt;
  bool x3 = false;
  DDMI_TRY(numerics_to_dt(cfg->numerics, &dt, &x3));
  ddmi_mlp2d* h = new (std::nothrow) ddmi_mlp2d();
  if (!h) return ddmi_fail(DDMI_ERR_STATE, "out of host memory");
  h->m.cfg = *cfg;
  h->m.dt = dt;
  *out = h;
  return DDMI_OK;
}
int ddmi_mlp2d_load_weight(ddmi_mlp2d* h, const char* key, const float* data, const int64_t* shape, int ndim) {
  DDMI_REQUIRE(h && key && data && ndim >= 0 && ndim <= 8, "mlp2d_load_weight: bad arguments");
  HostTensor t;
  t.shape.assign(shape, shape + ndim);
  t.data.assign(data, data + t.numel());
  h->m.host[key] = std::move(t);
  h->m.have_cache = false;
  return DDMI_OK;
}
int ddmi_mlp2d_finalize(ddmi_mlp2d* h) {
  DDMI_REQUIRE(h, "null handle");
  return mlp2d_finalize(h->m);
}
int ddmi_hdpe_mlp2d(ddmi_mlp2d* h, const void* const* planes, const int* plane_hw, int layout, const float* coords,
                    float si, const float* noise, float* rgb_out, int B, int N, void* stream) {
  DDMI_REQUIRE(h, "null handle");
  return mlp2d_forward(h->m, planes, plane_hw, layout, coords, si, noise, nullptr, rgb_out, B, N, (hipStream_t)stream);
}
int ddmi_hdpe_mlp2d_rng(ddmi_mlp2d* h, const void* const* planes, const int* plane_hw, int layout, const float* coords,
                        float si, const uint64_t* seeds, float* rgb_out, int B, int N, void* stream) {
  DDMI_REQUIRE(h && seeds, "mlp2d_rng: null handle / seeds");
  return mlp2d_forward(h->m, planes, plane_hw, layout, coords, si, nullptr, seeds, rgb_out, B, N, (hipStream_t)stream);
}
int ddmi_mlp2d_noise_nonzero(ddmi_mlp2d* h, int* out) {
  DDMI_REQUIRE(h && out, "mlp2d_noise_nonzero: null argument");
  if (!h->m.finalized) return ddmi_fail(DDMI_ERR_STATE, "mlp2d not finalized");
  *out = h->m.noise_nonzero ? 1 : 0;
  return DDMI_OK;
}
int ddmi_mlp2d_destroy(ddmi_mlp2d* h) {
  delete h;
  return DDMI_OK;
}

// ---- occupancy: tri-plane decoder + fused MLP3D ---------------------------------------------------------------
int ddmi_decoder3d_create(const ddmi_decoder2d_cfg* cfg, const int* inter_attn_resolutions, int n_inter, ddmi_decoder3d** out) {
  DDMI_REQUIRE(cfg && out && n_inter >= 0 && n_inter <= 8, "decoder3d_create: bad argument");
  DDMI_REQUIRE(cfg->max_batch >= 1, "decoder3d_create: max_batch must be >= 1");
  int dt;
  bool x3 = false;
  DDMI_TRY(numerics_to_dt(cfg->numerics, &dt, &x3));
  ddmi_decoder3d* h = new (std::nothrow) ddmi_decoder3d();
  if (!h) return ddmi_fail(DDMI_ERR_STATE, "out of host memory");
  h->d.cfg = *cfg;
  h->d.n_inter = n_inter;
  for (int i = 0; i < n_inter; ++i) h->d.inter_attn_resolutions[i] = inter_attn_resolutions[i];
  h->d.net.dt = dt;
  h->d.net.x3 = x3;
  h->d.net.max_batch = cfg->max_batch;
  *out = h;
  return DDMI_OK;
}
int ddmi_decoder3d_load_weight(ddmi_decoder3d* h, const char* key, const float* data, const int64_t* shape, int ndim) {
  DDMI_REQUIRE(h, "null handle");
  return h->d.net.add_host(key, data, shape, ndim);
}
int ddmi_decoder3d_finalize(ddmi_decoder3d* h) {
  DDMI_REQUIRE(h, "null handle");
  if (h->d.net.finalized) return ddmi_fail(DDMI_ERR_STATE, "already finalized");
  return decoder3d_finalize(h->d);
}
int ddmi_decoder3d_num_scales(ddmi_decoder3d* h, int* n, int plane_hw[4]) {
  DDMI_REQUIRE(h && n && plane_hw, "null argument");
  if (!h->d.net.finalized) return ddmi_fail(DDMI_ERR_STATE, "decoder not finalized");
  *n = h->d.n_scales;
  for (int i = 0; i < h->d.n_scales; ++i) plane_hw[i] = h->d.plane_hw[i];
  return DDMI_OK;
}
int ddmi_decoder3d_forward(ddmi_decoder3d* h, const float* z, void* const* planes, int layout, int B, void* stream) {
  DDMI_REQUIRE(h, "null handle");
  return decoder3d_forward(h->d, z, planes, layout, B, (hipStream_t)stream);
}
int ddmi_decoder3d_destroy(ddmi_decoder3d* h) {
  delete h;
  return DDMI_OK;
}

// ---- video decoder / video MLP ---------------------------------------------------------------------------
int ddmi_decoderv_create(const ddmi_decoderv_cfg* cfg, ddmi_decoderv** out) {
  DDMI_REQUIRE(cfg && out, "decoderv_create: null argument");
  DDMI_REQUIRE(cfg->max_batch >= 1, "decoderv_create: max_batch must be >= 1");
  DDMI_REQUIRE(cfg->n_attn >= 0 && cfg->n_attn <= 8 && cfg->n_hdbf >= 0 && cfg->n_hdbf <= 8 && cfg->n_inter >= 0 &&
                   cfg->n_inter <= 8 && cfg->frames >= 1 && cfg->resolution >= 8,
               "decoderv_create: bad configuration");
  int dt;
  bool x3 = false;
  DDMI_TRY(numerics_to_dt(cfg->numerics, &dt, &x3));
  ddmi_decoderv* h = new (std::nothrow) ddmi_decoderv();
  if (!h) return ddmi_fail(DDMI_ERR_STATE, "out of host memory");
  h->d.cfg = *cfg;
  h->d.net.dt = dt;
  h->d.net.x3 = x3;
  h->d.net.max_batch = cfg->max_batch;
  *out = h;
  return DDMI_OK;
}
int ddmi_decoderv_load_weight(ddmi_decoderv* h, const char* key, const float* data, const int64_t* shape, int ndim) {
  DDMI_REQUIRE(h, "null handle");
  return h->d.net.add_host(key, data, shape, ndim);
}
int ddmi_decoderv_finalize(ddmi_decoderv* h) {
  DDMI_REQUIRE(h, "null handle");
  if (h->d.net.finalized) return ddmi_fail(DDMI_ERR_STATE, "already finalized");
  return decoderv_finalize(h->d);
}
int ddmi_decoderv_num_scales(ddmi_decoderv* h, int* n, int plane_hw[4], int* out_ch) {
  DDMI_REQUIRE(h && n && plane_hw && out_ch, "null argument");
  if (!h->d.net.finalized) return ddmi_fail(DDMI_ERR_STATE, "decoder not finalized");
  *n = h->d.n_scales;
  *out_ch = h->d.out_ch;
  for (int i = 0; i < h->d.n_scales; ++i) plane_hw[i] = h->d.plane_w[0][i];
  return DDMI_OK;
}
int ddmi_decoderv_forward(ddmi_decoderv* h, const float* z, void* const* planes, int layout, int B, void* stream) {
  DDMI_REQUIRE(h, "null handle");
  return decoderv_forward(h->d, z, planes, layout, B, (hipStream_t)stream);
}
int ddmi_decoderv_destroy(ddmi_decoderv* h) {
  delete h;
  return DDMI_OK;
}
int ddmi_mlpv_create(const ddmi_mlp2d_cfg* cfg, ddmi_mlpv** out) {
  DDMI_REQUIRE(cfg && out, "mlpv_create: null argument");
  int dt;
  bool x3 = false;
  DDMI_TRY(numerics_to_dt(cfg->numerics, &dt, &x3));
  ddmi_mlpv* h = new (std::nothrow) ddmi_mlpv();
  if (!h) return ddmi_fail(DDMI_ERR_STATE, "out of host memory");
  h->m.cfg = *cfg;
  h->m.dt = dt;
  *out = h;
  return DDMI_OK;
}
int ddmi_mlpv_load_weight(ddmi_mlpv* h, const char* key, const float* data, const int64_t* shape, int ndim) {
  DDMI_REQUIRE(h && key && data && ndim >= 0 && ndim <= 8, "mlpv_load_weight: bad arguments");
  if (h->m.finalized) return ddmi_fail(DDMI_ERR_STATE, "load_weight after finalize");
  HostTensor t;
  t.shape.assign(shape, shape + ndim);
  t.data.assign(data, data + t.numel());
  h->m.host[key] = std::move(t);
  return DDMI_OK;
}
int ddmi_mlpv_finalize(ddmi_mlpv* h) {
  DDMI_REQUIRE(h, "null handle");
  return mlpv_finalize(h->m);
}
int ddmi_hdpe_mlpv(ddmi_mlpv* h, const void* const* planes, const int* plane_hw, int frames, int layout,
                   const float* coords_xy, const float* coords_yt, const float* coords_xt, float* video_out, int B,
                   int T, int H, int W, void* stream) {
  DDMI_REQUIRE(h, "null handle");
  return mlpv_forward(h->m, planes, plane_hw, frames, layout, coords_xy, coords_yt, coords_xt, video_out, B, T, H, W,
                      (hipStream_t)stream);
}
int ddmi_mlpv_destroy(ddmi_mlpv* h) {
  delete h;
  return DDMI_OK;
}

int ddmi_mlp3d_create(const ddmi_mlp2d_cfg* cfg, ddmi_mlp3d** out) {
  DDMI_REQUIRE(cfg && out, "mlp3d_create: null argument");
  int dt;
  bool x3 = false;
  DDMI_TRY(numerics_to_dt(cfg->numerics, &dt, &x3));
  ddmi_mlp3d* h = new (std::nothrow) ddmi_mlp3d();
  if (!h) return ddmi_fail(DDMI_ERR_STATE, "out of host memory");
  h->m.cfg = *cfg;
  h->m.dt = dt;
  *out = h;
  return DDMI_OK;
}
int ddmi_mlp3d_load_weight(ddmi_mlp3d* h, const char* key, const float* data, const int64_t* shape, int ndim) {
  DDMI_REQUIRE(h && key && data && ndim >= 0 && ndim <= 8, "mlp3d_load_weight: bad arguments");
  if (h->m.finalized) return ddmi_fail(DDMI_ERR_STATE, "load_weight after finalize");
  HostTensor t;
  t.shape.assign(shape, shape + ndim);
  t.data.assign(data, data + t.numel());
  h->m.host[key] = std::move(t);
  return DDMI_OK;
}
int ddmi_mlp3d_finalize(ddmi_mlp3d* h) {
  DDMI_REQUIRE(h, "null handle");
  return mlp3d_finalize(h->m);
}
int ddmi_hdpe_mlp3d(ddmi_mlp3d* h, const void* const* planes, const int* plane_hw, int layout, const float* points,
                    float* logits_out, int B, int N, void* stream) {
  DDMI_REQUIRE(h, "null handle");
  return mlp3d_forward(h->m, planes, plane_hw, layout, points, logits_out, B, N, (hipStream_t)stream);
}
int ddmi_mlp3d_destroy(ddmi_mlp3d* h) {
  delete h;
  return DDMI_OK;
}

// ---- diagnostic single-op entry points ----------------------------------------------------------------------
struct Scratch {
  std::vector<void*> p;
  ~Scratch() {
    for (void* q : p) (void)hipFree(q);
  }
  int get(size_t bytes, void** out) {
    void* q = nullptr;
    HIP_TRY(hipMalloc(&q, bytes ? bytes : 16));
    p.push_back(q);
    *out = q;
    return DDMI_OK;
  }
};

// the bf16 network runs (x2 nearest upsample -> 3x3 conv) in its parity form (net.hip op_conv); the diagnostics do too
static int maybe_up4(ConvArgs* a, int dt, Scratch& sc) {
  static const bool up4 = !(getenv("DDMI_CONV_UP4") && atoi(getenv("DDMI_CONV_UP4")) == 0);
  if (!(up4 && dt == DT_BF16 && a->ups == 1 && a->KH == 3 && a->stride == 1 && a->splitk <= 1)) return DDMI_OK;
  void* w4 = nullptr;
  DDMI_TRY(sc.get((size_t)16 * a->Cout * a->Cin * dt_size(dt), &w4));
  DDMI_TRY(launch_pack_conv_up4(a->w, w4, a->Cout, a->Cin, 0));
  a->w = w4; a->ups = 3; a->KH = a->KW = 2; a->pad = 1;
  return DDMI_OK;
}

int ddmi_op_conv2d(const float* x, const float* w, const float* bias, float* y, int B, int Cin, int H, int W, int Cout,
                   int k, int stride, int upsample, int numerics, int splitk) {
  int dt;
  bool x3 = false;
  DDMI_TRY(numerics_to_dt(numerics, &dt, &x3));
  DDMI_REQUIRE(x && w && y, "op_conv2d: null tensor");
  Scratch sc;
  void *xn, *wp, *yn, *ws = nullptr;
  if (x3 && Cin % 8 == 0) {   // DDMI_PARITY_BF16X3: split the input, bf16 kernel over K = 3 Cin into fp32 slabs, fp32 finish (net.hip)
    int hin = upsample == 1 ? 2 * H : H, win = upsample ? 2 * W : W, pad = k / 2;
    int Ho = (hin + 2 * pad - k) / stride + 1, Wo = (win + 2 * pad - k) / stride + 1;
    void *x3in, *w3;
    DDMI_TRY(sc.get((size_t)B * Cin * H * W * 4, &xn));
    DDMI_TRY(sc.get((size_t)B * Cin * H * W * 6, &x3in));
    DDMI_TRY(sc.get((size_t)Cout * Cin * k * k * 6, &w3));
    DDMI_TRY(sc.get((size_t)B * Cout * Ho * Wo * 4, &yn));
    DDMI_TRY(launch_nchw_to_nhwc(x, xn, DT_F32, B, Cin, H * W, 0));
    DDMI_TRY(launch_split_x3((const float*)xn, x3in, (int64_t)B * H * W, Cin, 0));
    DDMI_TRY(launch_pack_conv_weight_x3(w, w3, Cout, Cin, k, k, 0));
    ConvArgs f{};
    f.x = xn; f.bias = bias; f.y = yn;
    f.B = B; f.Hin = H; f.Win = W; f.Cin = Cin; f.Hout = Ho; f.Wout = Wo; f.Cout = Cout;
    f.KH = f.KW = k; f.stride = stride; f.pad = pad; f.ups = upsample; f.out_scale = 1.f;
    ConvArgs b = f;
    b.x = x3in; b.w = w3; b.Cin = 3 * Cin; b.bias = nullptr; b.y = nullptr; b.force_ws = 1;
    b.splitk = splitk > 0 ? splitk : conv_pick_splitk(b, DT_BF16);
    DDMI_TRY(sc.get((size_t)b.splitk * B * Ho * Wo * Cout * sizeof(float), &ws));
    b.splitk_ws = (float*)ws;
    DDMI_TRY(launch_conv(b, DT_BF16, 0));
    f.splitk = b.splitk; f.splitk_ws = (float*)ws;
    DDMI_TRY(launch_splitk_finish(f, DT_F32, 0));
    DDMI_TRY(launch_nhwc_to_nchw(yn, DT_F32, y, B, Cout, Ho * Wo, 0));
    HIP_TRY(hipDeviceSynchronize());
    return DDMI_OK;
  }
  int hin = upsample == 1 ? 2 * H : H, win = upsample ? 2 * W : W, pad = k / 2;
  int Ho = (hin + 2 * pad - k) / stride + 1, Wo = (win + 2 * pad - k) / stride + 1;
  DDMI_TRY(sc.get((size_t)B * Cin * H * W * dt_size(dt), &xn));
  DDMI_TRY(sc.get((size_t)Cout * Cin * k * k * dt_size(dt), &wp));
  DDMI_TRY(sc.get((size_t)B * Cout * Ho * Wo * dt_size(dt), &yn));
  DDMI_TRY(launch_nchw_to_nhwc(x, xn, dt, B, Cin, H * W, 0));
  DDMI_TRY(launch_pack_conv_weight(w, wp, dt, Cout, Cin, k, k, 0));
  ConvArgs a{};
  a.x = xn; a.w = wp; a.bias = bias; a.y = yn;
  a.B = B; a.Hin = H; a.Win = W; a.Cin = Cin; a.Hout = Ho; a.Wout = Wo; a.Cout = Cout;
  a.KH = a.KW = k; a.stride = stride; a.pad = pad; a.ups = upsample; a.out_scale = 1.f;
  a.splitk = splitk > 0 ? splitk : conv_pick_splitk(a, dt);
  DDMI_TRY(maybe_up4(&a, dt, sc));
  if (a.splitk > 1) {
    DDMI_TRY(sc.get((size_t)a.splitk * B * Ho * Wo * Cout * sizeof(float), &ws));
    a.splitk_ws = (float*)ws;
  }
  DDMI_TRY(launch_conv(a, dt, 0));
  DDMI_TRY(launch_nhwc_to_nchw(yn, dt, y, B, Cout, Ho * Wo, 0));
  HIP_TRY(hipDeviceSynchronize());
  return DDMI_OK;
}

// conv3x3(x, w, b) + conv1x1(cat(xs1, xs2), ws, bs): a ResBlock's out_layers convolution plus its skip_connection
// (openaimodel.py:272-275).  bf16: ONE launch, the 1x1 folded into the 3x3 kernel's K loop; fp32: two launches (the parity mode
// keeps the reference's op order: the skip tensor is the second launch's residual).
int ddmi_op_conv2d_skip(const float* x, const float* w, const float* bias, const float* xs1, const float* xs2, const float* ws,
                        const float* bias_s, float* y, int B, int Cin, int H, int W, int Cout, int C2, int C3, int numerics, int splitk) {
  int dt;
  DDMI_TRY(numerics_to_dt(numerics, &dt));
  DDMI_REQUIRE(x && w && bias && xs1 && ws && bias_s && y && C2 > 0 && C3 >= 0 && (C3 == 0 || xs2), "op_conv2d_skip: bad arguments");
  Scratch sc;
  const int Cs = C2 + C3, hw = H * W;
  void *xn, *wp, *yn, *s1, *s2 = nullptr, *w2, *skn = nullptr, *wsk = nullptr;
  float* bsum = nullptr;
  DDMI_TRY(sc.get((size_t)B * Cin * hw * dt_size(dt), &xn));
  DDMI_TRY(sc.get((size_t)Cout * Cin * 9 * dt_size(dt), &wp));
  DDMI_TRY(sc.get((size_t)B * Cout * hw * dt_size(dt), &yn));
  DDMI_TRY(sc.get((size_t)B * C2 * hw * dt_size(dt), &s1));
  if (C3) DDMI_TRY(sc.get((size_t)B * C3 * hw * dt_size(dt), &s2));
  DDMI_TRY(sc.get((size_t)Cout * Cs * dt_size(dt), &w2));
  DDMI_TRY(launch_nchw_to_nhwc(x, xn, dt, B, Cin, hw, 0));
  DDMI_TRY(launch_nchw_to_nhwc(xs1, s1, dt, B, C2, hw, 0));
  if (C3) DDMI_TRY(launch_nchw_to_nhwc(xs2, s2, dt, B, C3, hw, 0));
  DDMI_TRY(launch_pack_conv_weight(w, wp, dt, Cout, Cin, 3, 3, 0));
  DDMI_TRY(launch_pack_conv_weight(ws, w2, dt, Cout, Cs, 1, 1, 0));
  ConvArgs a{};
  a.x = xn; a.w = wp; a.y = yn;
  a.B = B; a.Hin = a.Hout = H; a.Win = a.Wout = W; a.Cin = Cin; a.Cout = Cout;
  a.KH = a.KW = 3; a.stride = 1; a.pad = 1; a.out_scale = 1.f;
  if (dt == DT_BF16) {
    std::vector<float> b1(Cout), b2(Cout);
    HIP_TRY(hipMemcpy(b1.data(), bias, Cout * sizeof(float), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(b2.data(), bias_s, Cout * sizeof(float), hipMemcpyDeviceToHost));
    for (int i = 0; i < Cout; ++i) b1[i] += b2[i];
    DDMI_TRY(sc.get((size_t)Cout * sizeof(float), (void**)&bsum));
    HIP_TRY(hipMemcpy(bsum, b1.data(), Cout * sizeof(float), hipMemcpyHostToDevice));
    a.bias = bsum;
    a.x2 = s1; a.C2 = C2; a.x3 = s2; a.C3 = C3; a.w2 = w2;
  } else {   // skip = conv1x1 over the two column blocks of ws, then the residual of the 3x3 launch
    DDMI_TRY(sc.get((size_t)B * Cout * hw * dt_size(dt), &skn));
    DDMI_TRY(sc.get((size_t)Cout * Cs * dt_size(dt), &wsk));
    std::vector<float> hws((size_t)Cout * Cs), ha((size_t)Cout * C2), hb((size_t)Cout * std::max(C3, 1));
    HIP_TRY(hipMemcpy(hws.data(), ws, hws.size() * sizeof(float), hipMemcpyDeviceToHost));
    for (int o = 0; o < Cout; ++o) {
      std::copy(hws.begin() + (size_t)o * Cs, hws.begin() + (size_t)o * Cs + C2, ha.begin() + (size_t)o * C2);
      if (C3) std::copy(hws.begin() + (size_t)o * Cs + C2, hws.begin() + (size_t)(o + 1) * Cs, hb.begin() + (size_t)o * C3);
    }
    float *da, *db = nullptr;
    void *pa, *pb = nullptr;
    DDMI_TRY(sc.get(ha.size() * sizeof(float), (void**)&da));
    DDMI_TRY(sc.get(ha.size() * dt_size(dt), &pa));
    HIP_TRY(hipMemcpy(da, ha.data(), ha.size() * sizeof(float), hipMemcpyHostToDevice));
    DDMI_TRY(launch_pack_conv_weight(da, pa, dt, Cout, C2, 1, 1, 0));
    ConvArgs k1{};
    k1.x = s1; k1.w = pa; k1.bias = bias_s; k1.y = skn;
    k1.B = B; k1.Hin = k1.Hout = H; k1.Win = k1.Wout = W; k1.Cin = C2; k1.Cout = Cout;
    k1.KH = k1.KW = 1; k1.stride = 1; k1.pad = 0; k1.out_scale = 1.f; k1.splitk = 1;
    DDMI_TRY(launch_conv(k1, dt, 0));
    if (C3) {
      DDMI_TRY(sc.get(hb.size() * sizeof(float), (void**)&db));
      DDMI_TRY(sc.get(hb.size() * dt_size(dt), &pb));
      HIP_TRY(hipMemcpy(db, hb.data(), hb.size() * sizeof(float), hipMemcpyHostToDevice));
      DDMI_TRY(launch_pack_conv_weight(db, pb, dt, Cout, C3, 1, 1, 0));
      ConvArgs k2 = k1;
      k2.x = s2; k2.w = pb; k2.bias = nullptr; k2.res = skn; k2.Cin = C3;
      DDMI_TRY(launch_conv(k2, dt, 0));
    }
    a.bias = bias; a.res = skn;
  }
  a.splitk = splitk > 0 ? splitk : conv_pick_splitk(a, dt);
  void* wsl = nullptr;
  if (a.splitk > 1) {
    DDMI_TRY(sc.get((size_t)a.splitk * B * hw * Cout * sizeof(float), &wsl));
    a.splitk_ws = (float*)wsl;
  }
  DDMI_TRY(launch_conv(a, dt, 0));
  DDMI_TRY(launch_nhwc_to_nchw(yn, dt, y, B, Cout, hw, 0));
  HIP_TRY(hipDeviceSynchronize());
  return DDMI_OK;
}

__global__ void fill_hash_kernel(unsigned* p, int64_t n_words, unsigned seed, int bf16) {
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += stride) {
    unsigned h = (unsigned)i * 2654435761u + seed;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
    if (bf16) {  // two bf16 in [-1, 1): sign + exponent 126/125.. keep it simple: map to +-(1.0 .. 2.0) - 1.5
      unsigned lo = 0x3F00u | (h & 0x807Fu), hi = 0x3F00u | ((h >> 16) & 0x807Fu);
      p[i] = lo | (hi << 16);
    } else {
      p[i] = 0x3F000000u | (h & 0x807FFFFFu);  // +-[0.5, 1)
    }
  }
}

/* Micro-benchmark of one convolution shape on device-resident random data: average ms over `iters` launches. */
int ddmi_bench_conv(int B, int Cin, int H, int W, int Cout, int k, int stride, int upsample, int numerics, int splitk,
                    int force_bm, int force_bn, int iters, double* ms_out, int* splitk_used) {
  int dt;
  DDMI_TRY(numerics_to_dt(numerics, &dt));
  DDMI_REQUIRE(ms_out && iters > 0, "bench_conv: bad arguments");
  Scratch sc;
  void *xn, *wp, *yn, *ws = nullptr;
  int hin = upsample == 1 ? 2 * H : H, win = upsample ? 2 * W : W, pad = k / 2;
  int Ho = (hin + 2 * pad - k) / stride + 1, Wo = (win + 2 * pad - k) / stride + 1;
  size_t xb = (size_t)B * Cin * H * W * dt_size(dt), wb = (size_t)Cout * Cin * k * k * dt_size(dt);
  DDMI_TRY(sc.get(xb, &xn));
  DDMI_TRY(sc.get(wb, &wp));
  DDMI_TRY(sc.get((size_t)B * Cout * Ho * Wo * dt_size(dt), &yn));
  hipLaunchKernelGGL(fill_hash_kernel, dim3(1024), dim3(256), 0, 0, (unsigned*)xn, (int64_t)(xb / 4), 1u, dt == DT_BF16);
  hipLaunchKernelGGL(fill_hash_kernel, dim3(1024), dim3(256), 0, 0, (unsigned*)wp, (int64_t)(wb / 4), 2u, dt == DT_BF16);
  conv_force_tile(force_bm, force_bn);
  ConvArgs a{};
  a.x = xn; a.w = wp; a.y = yn;
  a.B = B; a.Hin = H; a.Win = W; a.Cin = Cin; a.Hout = Ho; a.Wout = Wo; a.Cout = Cout;
  a.KH = a.KW = k; a.stride = stride; a.pad = pad; a.ups = upsample; a.out_scale = 1.f;
  a.splitk = splitk > 0 ? splitk : conv_pick_splitk(a, dt);
  DDMI_TRY(maybe_up4(&a, dt, sc));
  if (splitk_used) *splitk_used = a.splitk;
  int rc = DDMI_OK;
  if (a.splitk > 1) {
    rc = sc.get((size_t)a.splitk * B * Ho * Wo * Cout * sizeof(float), &ws);
    a.splitk_ws = (float*)ws;
  }
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  for (int i = 0; i < 3 && rc == DDMI_OK; ++i) rc = launch_conv(a, dt, 0);
  (void)hipEventRecord(e0, 0);
  for (int i = 0; i < iters && rc == DDMI_OK; ++i) rc = launch_conv(a, dt, 0);
  (void)hipEventRecord(e1, 0);
  hipError_t e = hipDeviceSynchronize();
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  conv_force_tile(0, 0);
  if (rc != DDMI_OK) return rc;
  if (e != hipSuccess) return ddmi_fail(DDMI_ERR_HIP, std::string("bench_conv: ") + hipGetErrorString(e));
  *ms_out = (double)ms / iters;
  return DDMI_OK;
}

int ddmi_debug_conv_config(int cfg_id) {
  if (cfg_id < 0) conv_force_tile(0, 0);
  else conv_force_tile(cfg_id, -1);
  return DDMI_OK;
}

int ddmi_bench_peak_mfma(double seconds_target, double* tflops) {
  DDMI_REQUIRE(tflops && seconds_target > 0 && seconds_target <= 10, "bench_peak_mfma: bad arguments");
  return bench_peak_mfma(seconds_target, tflops, nullptr);
}
int ddmi_bench_peak_copy(double gib, double* tb_per_s) {
  DDMI_REQUIRE(tb_per_s && gib >= 0.25 && gib <= 32, "bench_peak_copy: bad arguments");
  return bench_peak_copy(gib, tb_per_s);
}

int ddmi_debug_halo_half(int mode) {
  conv_halo_half_override(mode);
  return DDMI_OK;
}

int ddmi_debug_small_batch(int mode) {
  net_small_batch_override(mode);
  return DDMI_OK;
}
int ddmi_debug_small_gn_launches(int* count) {
  DDMI_REQUIRE(count, "null argument");
  *count = g_net_small_gn_launches;
  return DDMI_OK;
}

// the one-launch GroupNorm of the small-batch regime (k_groupnorm.hip gn_small_kernel) on its own: GroupNorm32 [+ SiLU] of the channel
// concatenation [x | x2] (x2 may be null), bf16
int ddmi_op_groupnorm_small(const float* x, const float* x2, const float* gamma, const float* beta, float* y, int B, int C1, int C2, int HW,
                            float eps, int silu) {
  DDMI_REQUIRE(x && gamma && beta && y && B > 0 && C1 > 0 && C2 >= 0 && HW > 0 && (x2 != nullptr) == (C2 > 0), "op_groupnorm_small: bad arguments");
  const int C = C1 + C2;
  DDMI_REQUIRE(gn_small_ok(B, HW, C, x2 != nullptr), "op_groupnorm_small: shape not served by the one-launch kernel");
  Scratch sc;
  void *x1n, *x2n = nullptr, *yn;
  DDMI_TRY(sc.get((size_t)B * C1 * HW * 2, &x1n));
  if (x2) DDMI_TRY(sc.get((size_t)B * C2 * HW * 2, &x2n));
  DDMI_TRY(sc.get((size_t)B * C * HW * 2, &yn));
  DDMI_TRY(launch_nchw_to_nhwc(x, x1n, DT_BF16, B, C1, HW, 0));
  if (x2) DDMI_TRY(launch_nchw_to_nhwc(x2, x2n, DT_BF16, B, C2, HW, 0));
  GnSmallArgs a{};
  a.x = x1n; a.x2 = x2n; a.C1 = C1; a.y = yn; a.gamma = gamma; a.beta = beta; a.eps = eps; a.silu = silu; a.B = B; a.HW = HW; a.C = C;
  DDMI_TRY(launch_gn_small(a, 0));
  DDMI_TRY(launch_nhwc_to_nchw(yn, DT_BF16, y, B, C, HW, 0));
  HIP_TRY(hipDeviceSynchronize());
  return DDMI_OK;
}

int ddmi_debug_gn_fuse(int mode) {
  net_gn_fuse_override(mode);
  return DDMI_OK;
}

int ddmi_debug_conv_gn_fusable(int B, int Cin, int H, int W, int Cout, int any_batch, int* fusable, int* tiles_per_sample) {
  DDMI_REQUIRE(fusable && B > 0 && Cin > 0 && H > 0 && W > 0 && Cout > 0, "debug_conv_gn_fusable: bad arguments");
  static float dummy[4];   // the predicate only asks whether the pointers are there
  ConvArgs a{};
  a.B = B; a.Hin = a.Hout = H; a.Win = a.Wout = W; a.Cin = Cin; a.Cout = Cout;
  a.KH = a.KW = 3; a.stride = 1; a.pad = 1; a.out_scale = 1.f; a.splitk = 1;
  a.x = a.w = a.y = dummy; a.bias = dummy;
  a.gn_gamma = a.gn_beta = dummy; a.gn_part = dummy; a.gn_sync = (unsigned*)dummy;
  a.gn_any_batch = (any_batch & 1) ? 1 : 0;
  a.gn_no_handoff = (any_batch & 2) ? 1 : 0;   // bit 1: a network whose owner never looks at the hand-off's flag (the decoders)
  a.cfg = (Cout % 224 == 0 ? 19 : 20) + 1;   // the halo tile Net::run gives a launch without a tuned plan
  *fusable = conv_fgn_ok(a, DT_BF16) ? 1 : 0;
  if (tiles_per_sample) *tiles_per_sample = (H * W) / 256;
  return DDMI_OK;
}

// conv3x3 (+ bias + per-sample embedding row) -> GroupNorm32 [+ SiLU] as the ResBlock builds the pair (openaimodel.py:261-270), through the
// network executor: NCHW in -> OP_CONV -> OP_GN -> NCHW out.  `iters` runs of the same net (the fused form's cross-workgroup hand-off under
// repetition); *fused_used = fused launches of the last run
int ddmi_op_conv_gn(const float* x, const float* w, const float* bias, const float* emb, const float* gamma, const float* beta, float* y,
                    int B, int Cin, int H, int W, int Cout, float eps, int silu, int numerics, int iters, int* fused_used) {
  DDMI_REQUIRE(x && w && bias && gamma && beta && y && B > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0 && iters > 0, "op_conv_gn: bad arguments");
  Net n;
  bool x3 = false;
  DDMI_TRY(numerics_to_dt(numerics, &n.dt, &x3));
  n.x3 = x3;
  n.max_batch = B;
  const int64_t sw[4] = {Cout, Cin, 3, 3}, s1[1] = {Cout};
  DDMI_TRY(n.add_host("c.weight", w, sw, 4));
  DDMI_TRY(n.add_host("c.bias", bias, s1, 1));
  DDMI_TRY(n.add_host("g.weight", gamma, s1, 1));
  DDMI_TRY(n.add_host("g.bias", beta, s1, 1));
  Op in;
  in.kind = OP_NCHW_IN;
  in.ext = 0;
  in.out = n.new_tensor(Cin, H, W);
  n.ops.push_back(in);
  int ci = -1, gi = -1;
  DDMI_TRY(n.make_conv("c.weight", "c.bias", &ci));
  DDMI_TRY(n.make_gn("g.", &gi));
  const int h = n.op_conv(in.out, ci, 1, 0, -1, emb ? 0 : -1);
  const int o = n.op_gn(h, gi, silu, eps);
  Op ex;
  ex.kind = OP_EXPORT;
  ex.in = o;
  ex.ext = 1;
  ex.out_nchw = 1;
  n.ops.push_back(ex);
  n.fgn_handoff = true;   // the flag is checked right below
  DDMI_TRY(n.plan());
  void* ext[2] = {(void*)x, (void*)y};
  for (int i = 0; i < iters; ++i) DDMI_TRY(n.run(B, ext, emb, Cout, 0));
  if (fused_used) *fused_used = n.fgn_launches;
  return n.async_status();
}

// the ResBlock's skip_connection on a channel-concatenated input (openaimodel.py:787, 272-275) through the network executor:
// y = conv1x1([x1 | x2]) + bias, as ONE dual-source launch where the shape allows (*cat_used), else two accumulating launches
int ddmi_op_conv1x1_cat(const float* x1, const float* x2, const float* w, const float* bias, float* y, int B, int C1, int C2, int H, int W,
                        int Cout, int numerics, int* cat_used) {
  DDMI_REQUIRE(x1 && x2 && w && bias && y && B > 0 && C1 > 0 && C2 > 0 && Cout > 0 && H > 0 && W > 0, "op_conv1x1_cat: bad arguments");
  Net n;
  bool x3 = false;
  DDMI_TRY(numerics_to_dt(numerics, &n.dt, &x3));
  n.x3 = x3;
  n.max_batch = B;
  const int64_t sw[2] = {Cout, C1 + C2}, s1[1] = {Cout};
  DDMI_TRY(n.add_host("s.weight", w, sw, 2));
  DDMI_TRY(n.add_host("s.bias", bias, s1, 1));
  int t[2];
  for (int i = 0; i < 2; ++i) {
    Op in;
    in.kind = OP_NCHW_IN;
    in.ext = i;
    in.out = t[i] = n.new_tensor(i ? C2 : C1, H, W);
    n.ops.push_back(in);
  }
  int cfull = -1, sk = -1;
  DDMI_TRY(n.make_conv("s.weight", "s.bias", &cfull));
  const bool cat = n.cat_conv_ok(t[0], t[1], cfull);
  if (cat) {
    sk = n.op_conv_cat(t[0], t[1], cfull);
  } else {   // the column blocks of the weight as two convolutions, the second accumulating (make_unet_resblock's split)
    const HostTensor* wt = n.find("s.weight");
    std::vector<float> wa((size_t)Cout * C1), wb((size_t)Cout * C2), bv(bias, bias + Cout);
    for (int o = 0; o < Cout; ++o) {
      std::copy(wt->data.begin() + (size_t)o * (C1 + C2), wt->data.begin() + (size_t)o * (C1 + C2) + C1, wa.begin() + (size_t)o * C1);
      std::copy(wt->data.begin() + (size_t)o * (C1 + C2) + C1, wt->data.begin() + (size_t)(o + 1) * (C1 + C2), wb.begin() + (size_t)o * C2);
    }
    int ca = -1, cb = -1;
    DDMI_TRY(n.make_conv_host(wa, Cout, C1, 1, &bv, &ca));
    DDMI_TRY(n.make_conv_host(wb, Cout, C2, 1, nullptr, &cb));
    sk = n.op_conv(t[0], ca, 1, 0, -1, -1);
    sk = n.op_conv(t[1], cb, 1, 0, sk, -1);
  }
  if (cat_used) *cat_used = cat ? 1 : 0;
  Op ex;
  ex.kind = OP_EXPORT;
  ex.in = sk;
  ex.ext = 2;
  ex.out_nchw = 1;
  n.ops.push_back(ex);
  DDMI_TRY(n.plan());
  void* ext[3] = {(void*)x1, (void*)x2, (void*)y};
  DDMI_TRY(n.run(B, ext, nullptr, 0, 0));
  HIP_TRY(hipDeviceSynchronize());
  return DDMI_OK;
}

int ddmi_debug_attn_fused(int mode) {
  attnblock_fused_override(mode);
  return DDMI_OK;
}

// one AttentionBlock (openaimodel.py:279-325) as the UNet builds it: NCHW in -> GroupNorm -> qkv -> attention -> proj_out + x -> NCHW
static int build_attention_block_net(UNet& u, const float* gamma, const float* beta, const float* wqkv, const float* bqkv,
                                     const float* wproj, const float* bproj, int C, int H, int W) {
  Net& n = u.net;
  const int64_t s1[1] = {C}, s3[1] = {3 * C}, sq[3] = {3 * C, C, 1}, sp[3] = {C, C, 1};
  DDMI_TRY(n.add_host("blk.norm.weight", gamma, s1, 1));
  DDMI_TRY(n.add_host("blk.norm.bias", beta, s1, 1));
  DDMI_TRY(n.add_host("blk.qkv.weight", wqkv, sq, 3));
  DDMI_TRY(n.add_host("blk.qkv.bias", bqkv, s3, 1));
  DDMI_TRY(n.add_host("blk.proj_out.weight", wproj, sp, 3));
  DDMI_TRY(n.add_host("blk.proj_out.bias", bproj, s1, 1));
  Op in;
  in.kind = OP_NCHW_IN;
  in.ext = 0;
  in.out = n.new_tensor(C, H, W);
  n.ops.push_back(in);
  AttnW w;
  DDMI_TRY(make_unet_attnblock(u, "blk.", C, &w, H * W));
  int y = -1;
  DDMI_TRY(emit_attn_single(n, w, in.out, &y));
  Op ex;
  ex.kind = OP_EXPORT;
  ex.in = y;
  ex.ext = 1;
  ex.out_nchw = 1;
  n.ops.push_back(ex);
  return n.plan();
}

int ddmi_op_attention_block(const float* x, const float* gamma, const float* beta, const float* wqkv, const float* bqkv,
                            const float* wproj, const float* bproj, float* y, int B, int C, int H, int W, int head_channels,
                            int numerics, int* fused_used) {
  DDMI_REQUIRE(x && gamma && beta && wqkv && bqkv && wproj && bproj && y && B > 0 && C > 0 && H > 0 && W > 0 && head_channels > 0,
               "op_attention_block: bad arguments");
  UNet u;
  bool x3 = false;
  DDMI_TRY(numerics_to_dt(numerics, &u.net.dt, &x3));
  u.net.x3 = x3;
  u.net.max_batch = B;
  u.cfg.num_head_channels = head_channels;
  DDMI_TRY(build_attention_block_net(u, gamma, beta, wqkv, bqkv, wproj, bproj, C, H, W));
  if (fused_used) {
    *fused_used = 0;
    for (const Op& o : u.net.ops) *fused_used |= o.kind == OP_ATTN_BLOCK;
  }
  void* ext[2] = {(void*)x, (void*)y};
  DDMI_TRY(u.net.run(B, ext, nullptr, 0, 0));
  HIP_TRY(hipDeviceSynchronize());
  return DDMI_OK;
}

// average device time of one AttentionBlock (random weights and input) as the network runs it; `fused_used` reports the form
int ddmi_bench_attention_block(int B, int C, int H, int W, int iters, double* us_out, int* fused_used) {
  DDMI_REQUIRE(us_out && iters > 0 && B > 0 && C % 32 == 0, "bench_attention_block: bad arguments");
  UNet u;
  u.net.dt = DT_BF16;
  u.net.max_batch = B;
  u.cfg.num_head_channels = 32;
  std::vector<float> gamma(C), beta(C), wqkv((size_t)3 * C * C), bqkv(3 * C), wproj((size_t)C * C), bproj(C);
  uint64_t st = 0x9E3779B97F4A7C15ull;
  auto rnd = [&]() {
    st = st * 6364136223846793005ull + 1442695040888963407ull;
    return (float)((st >> 40) & 0xFFFF) / 32768.0f - 1.0f;
  };
  const float ws = 1.0f / std::sqrt((float)C);
  for (auto& v : gamma) v = 1.0f + 0.1f * rnd();
  for (auto& v : beta) v = 0.1f * rnd();
  for (auto& v : wqkv) v = 1.7f * ws * rnd();
  for (auto& v : bqkv) v = 0.1f * rnd();
  for (auto& v : wproj) v = 1.7f * ws * rnd();
  for (auto& v : bproj) v = 0.1f * rnd();
  DDMI_TRY(build_attention_block_net(u, gamma.data(), beta.data(), wqkv.data(), bqkv.data(), wproj.data(), bproj.data(), C, H, W));
  if (fused_used) {
    *fused_used = 0;
    for (const Op& o : u.net.ops) *fused_used |= o.kind == OP_ATTN_BLOCK;
  }
  Scratch sc;
  void *xin, *yout;
  const size_t nb = (size_t)B * C * H * W * sizeof(float);
  DDMI_TRY(sc.get(nb, &xin));
  DDMI_TRY(sc.get(nb, &yout));
  hipLaunchKernelGGL(fill_hash_kernel, dim3(1024), dim3(256), 0, 0, (unsigned*)xin, (int64_t)(nb / 4), 7u, false);
  void* ext[2] = {xin, yout};
  // the block's own ops only: the NCHW seam launches (first / last op) are timed separately and subtracted
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  int rc = DDMI_OK;
  for (int i = 0; i < 2 && rc == DDMI_OK; ++i) rc = u.net.run(B, ext, nullptr, 0, 0);
  (void)hipEventRecord(e0, 0);
  for (int i = 0; i < iters && rc == DDMI_OK; ++i) rc = u.net.run(B, ext, nullptr, 0, 0);
  (void)hipEventRecord(e1, 0);
  hipError_t e = hipDeviceSynchronize();
  float ms = 0.f, ms_seam = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const TensorDesc& t0 = u.net.tensors[u.net.ops.front().out];
  (void)hipEventRecord(e0, 0);
  for (int i = 0; i < iters && rc == DDMI_OK; ++i) {
    rc = launch_nchw_to_nhwc((const float*)xin, u.net.tptr(u.net.ops.front().out), DT_BF16, B, t0.C, t0.H * t0.W, 0);
    if (rc == DDMI_OK) rc = launch_nhwc_to_nchw(u.net.tptr(u.net.ops.back().in), DT_BF16, (float*)yout, B, t0.C, t0.H * t0.W, 0);
  }
  (void)hipEventRecord(e1, 0);
  hipError_t e2 = hipDeviceSynchronize();
  (void)hipEventElapsedTime(&ms_seam, e0, e1);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  if (rc != DDMI_OK) return rc;
  if (e != hipSuccess || e2 != hipSuccess) return ddmi_fail(DDMI_ERR_HIP, "bench_attention_block: device error");
  prof_begin_forward();   // one more run under the per-launch scopes when the caller enabled them (ddmi_profile_enable)
  if (prof_active()) {
    rc = u.net.run(B, ext, nullptr, 0, 0);
    prof_enable(0);
    if (rc != DDMI_OK) return rc;
  }
  if (getenv("DDMI_ATTNBLOCK_TS") && fused_used && *fused_used) {   // per-workgroup phase timeline of the fused launch
    const int nwg = (int)((cdiv64((int64_t)B * H * W, 1024) + 7) / 8 * 8) * (C / 32);
    void* tsd = nullptr;
    DDMI_TRY(sc.get((size_t)nwg * 128, &tsd));
    HIP_TRY(hipMemset(tsd, 0, (size_t)nwg * 128));
    attnblock_debug_ts((unsigned long long*)tsd);
    rc = u.net.run(B, ext, nullptr, 0, 0);
    attnblock_debug_ts(nullptr);
    HIP_TRY(hipDeviceSynchronize());
    std::vector<unsigned long long> tsh((size_t)nwg * 16);
    HIP_TRY(hipMemcpy(tsh.data(), tsd, (size_t)nwg * 128, hipMemcpyDeviceToHost));
    double pa = 0, pb = 0, pc = 0;
    unsigned long long t_min = ~0ull, t_max = 0;
    int cnt = 0;
    double pw[8] = {0}, cyc_a = 0, cyc_b = 0;
    for (int i = 0; i < nwg; ++i) {
      const unsigned long long* t = &tsh[16 * i];
      if (!t[0]) continue;
      pa += (double)(t[1] - t[0]);
      pb += (double)(t[2] - t[1]);
      pc += (double)(t[3] - t[0]);
      for (int w = 0; w < 8; ++w) pw[w] += (double)(t[4 + w] - t[1]);
      cyc_a += (double)(t[13] - t[12]);
      cyc_b += (double)(t[14] - t[13]);
      t_min = std::min(t_min, t[0]);
      t_max = std::max(t_max, t[3]);
      ++cnt;
    }
    fprintf(stderr, "[attnblock ts] attention phase per wave (us):");
    for (int w = 0; w < 8; ++w) fprintf(stderr, " %.1f", pw[w] / cnt * 0.01);
    fprintf(stderr, "; shader clock: projection %.2f GHz, attention %.2f GHz\n", cyc_a / (pa * 10.0), cyc_b / (pb * 10.0));
    fprintf(stderr, "[attnblock ts] C=%d %dx%d B=%d: %d workgroups, projection %.2f us, attention %.2f us (wave 0), workgroup %.2f us (means), launch span %.1f us\n",
            C, H, W, B, cnt, pa / cnt * 0.01, pb / cnt * 0.01, pc / cnt * 0.01, (double)(t_max - t_min) * 0.01);
  }
  *us_out = 1000.0 * (ms - ms_seam) / iters;
  return DDMI_OK;
}

int ddmi_op_groupnorm(const float* x, const float* gamma, const float* beta, float* y, int B, int C, int HW, float eps,
                      int silu, int numerics) {
  int dt;
  DDMI_TRY(numerics_to_dt(numerics, &dt));
  Scratch sc;
  void *xn, *yn, *st;
  DDMI_TRY(sc.get((size_t)B * C * HW * dt_size(dt), &xn));
  DDMI_TRY(sc.get((size_t)B * C * HW * dt_size(dt), &yn));
  DDMI_TRY(sc.get(GN_STATS_DOUBLES(B) * sizeof(double), &st));
  DDMI_TRY(launch_nchw_to_nhwc(x, xn, dt, B, C, HW, 0));
  DDMI_TRY(launch_gn_stats(xn, dt, (double*)st, B, HW, C, 0));
  DDMI_TRY(launch_gn_apply(xn, yn, dt, (const double*)st, gamma, beta, eps, silu, B, HW, C, 0));
  DDMI_TRY(launch_nhwc_to_nchw(yn, dt, y, B, C, HW, 0));
  HIP_TRY(hipDeviceSynchronize());
  return DDMI_OK;
}

int ddmi_op_attention_legacy_hd(const float* qkv, float* out, int B, int heads, int d, int T, int numerics) {
  int dt;
  DDMI_TRY(numerics_to_dt(numerics, &dt));
  DDMI_REQUIRE(qkv && out && B > 0 && heads > 0 && d > 0 && T > 0, "op_attention_legacy: bad arguments");
  Scratch sc;
  void *qn, *on;
  int C3 = heads * 3 * d, C = heads * d;
  DDMI_TRY(sc.get((size_t)B * C3 * T * dt_size(dt), &qn));
  DDMI_TRY(sc.get((size_t)B * C * T * dt_size(dt), &on));
  DDMI_TRY(launch_nchw_to_nhwc(qkv, qn, dt, B, C3, T, 0));
  DDMI_TRY(launch_attention_legacy(qn, on, dt, B, T, heads, d, 0));
  DDMI_TRY(launch_nhwc_to_nchw(on, dt, out, B, C, T, 0));
  HIP_TRY(hipDeviceSynchronize());
  return DDMI_OK;
}

int ddmi_op_attention_legacy(const float* qkv, float* out, int B, int heads, int T, int numerics) {
  return ddmi_op_attention_legacy_hd(qkv, out, B, heads, 32, T, numerics);
}

static int op_attention_mh_impl(const float* q, const float* k, const float* v, float* out, int B, int heads, int d, int N,
                                int seg0, int seg1, int numerics, int q_prescaled) {
  int dt;
  DDMI_TRY(numerics_to_dt(numerics, &dt));
  DDMI_REQUIRE(q && k && v && out && B > 0 && heads > 0 && d > 0 && N > 0, "op_attention_mh: bad arguments");
  DDMI_REQUIRE(seg0 > 0 && seg1 >= 0 && seg0 + seg1 <= N, "op_attention_mh: bad segment lengths");
  const int C = heads * d, ld = 3 * C, es = (int)dt_size(dt);
  const int len[3] = {seg0, seg1, N - seg0 - seg1};
  Scratch sc;
  void *qn, *kn, *vn, *qk, *qkvn, *on;
  DDMI_TRY(sc.get((size_t)B * N * C * es, &qn));
  DDMI_TRY(sc.get((size_t)B * N * C * es, &kn));
  DDMI_TRY(sc.get((size_t)B * N * C * es, &vn));
  DDMI_TRY(sc.get((size_t)B * N * 2 * C * es, &qk));
  DDMI_TRY(sc.get((size_t)B * N * ld * es, &qkvn));
  DDMI_TRY(sc.get((size_t)B * N * C * es, &on));
  DDMI_TRY(launch_nchw_to_nhwc(q, qn, dt, B, C, N, 0));
  DDMI_TRY(launch_nchw_to_nhwc(k, kn, dt, B, C, N, 0));
  DDMI_TRY(launch_nchw_to_nhwc(v, vn, dt, B, C, N, 0));
  DDMI_TRY(launch_concat(qn, C, kn, C, qk, dt, (int64_t)B * N, 0));
  DDMI_TRY(launch_concat(qk, 2 * C, vn, C, qkvn, dt, (int64_t)B * N, 0));
  // scatter the token rows of every sample into one buffer per segment ([B * len][ld]), as the planes are stored
  AttnArgs a{};
  void* oseg[3] = {nullptr, nullptr, nullptr};
  int off = 0;
  for (int s3 = 0; s3 < 3; ++s3) {
    a.seg_len[s3] = len[s3];
    if (len[s3] == 0) continue;
    void* buf;
    DDMI_TRY(sc.get((size_t)B * len[s3] * ld * es, &buf));
    DDMI_TRY(sc.get((size_t)B * len[s3] * C * es, &oseg[s3]));
    HIP_TRY(hipMemcpy2DAsync(buf, (size_t)len[s3] * ld * es, (const char*)qkvn + (size_t)off * ld * es, (size_t)N * ld * es,
                             (size_t)len[s3] * ld * es, B, hipMemcpyDeviceToDevice, 0));
    a.qkv[s3] = buf;
    a.out[s3] = oseg[s3];
    off += len[s3];
  }
  a.ld_qkv = ld; a.ld_out = C;
  a.q_off = 0; a.k_off = C; a.v_off = 2 * C;
  a.hs_qkv = d; a.hs_out = d;
  a.d = d; a.heads = heads; a.B = B;
  a.scale = 1.0f / sqrtf((float)d);
  a.q_prescaled = q_prescaled;
  DDMI_TRY(launch_attention(a, dt, 0));
  off = 0;
  for (int s3 = 0; s3 < 3; ++s3) {
    if (len[s3] == 0) continue;
    HIP_TRY(hipMemcpy2DAsync((char*)on + (size_t)off * C * es, (size_t)N * C * es, oseg[s3], (size_t)len[s3] * C * es,
                             (size_t)len[s3] * C * es, B, hipMemcpyDeviceToDevice, 0));
    off += len[s3];
  }
  DDMI_TRY(launch_nhwc_to_nchw(on, dt, out, B, C, N, 0));
  HIP_TRY(hipDeviceSynchronize());
  return DDMI_OK;
}

int ddmi_op_attention_mh(const float* q, const float* k, const float* v, float* out, int B, int heads, int d, int N,
                         int seg0, int seg1, int numerics) {
  return op_attention_mh_impl(q, k, v, out, B, heads, d, N, seg0, seg1, numerics, 0);
}

int ddmi_op_attention_mh_prescaled(const float* q, const float* k, const float* v, float* out, int B, int heads, int d, int N,
                                   int seg0, int seg1) {
  return op_attention_mh_impl(q, k, v, out, B, heads, d, N, seg0, seg1, DDMI_PERF_BF16, 1);
}

int ddmi_op_timestep_embedding(const int64_t* t, float* out, int B, int dim) {
  DDMI_REQUIRE(dim >= 2, "timestep_embedding: dim too small");
  int half = dim / 2;
  std::vector<float> fr(half);
  const float neg_log = (float)(-std::log(10000.0));
  for (int i = 0; i < half; ++i) {
    volatile float a = neg_log * (float)i;
    volatile float q = a / (float)half;
    fr[i] = std::exp((float)q);
  }
  Scratch sc;
  void* f;
  DDMI_TRY(sc.get(half * sizeof(float), &f));
  HIP_TRY(hipMemcpy(f, fr.data(), half * sizeof(float), hipMemcpyHostToDevice));
  DDMI_TRY(launch_timestep_embedding(t, (const float*)f, out, B, dim, 0));
  HIP_TRY(hipDeviceSynchronize());
  return DDMI_OK;
}

}  // extern "C"
