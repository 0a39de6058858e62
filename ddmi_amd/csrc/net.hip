// Network executor: weight packing, liveness-planned activation arena, op dispatch (see net.h).
#include "net.h"
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

static thread_local std::string g_last_error;

void ddmi_set_error(const std::string& msg) { g_last_error = msg; }
int ddmi_fail(int code, const std::string& msg) {
  g_last_error = msg;
  return code;
}
const char* ddmi_error_cstr() { return g_last_error.c_str(); }

Net::~Net() {
  for (void* p : dev_allocs)
    if (p) (void)hipFree(p);
  if (fgn_flag_host) (void)hipHostFree(fgn_flag_host);
}

int Net::add_host(const char* key, const float* data, const int64_t* shape, int ndim) {
  DDMI_REQUIRE(key && data && ndim >= 0 && ndim <= 8, "load_weight: bad arguments");
  if (finalized) return ddmi_fail(DDMI_ERR_STATE, "load_weight after finalize");
  HostTensor t;
  t.shape.assign(shape, shape + ndim);
  int64_t n = t.numel();
  t.data.assign(data, data + n);
  host[key] = std::move(t);
  return DDMI_OK;
}

const HostTensor* Net::find(const std::string& key) const {
  auto it = host.find(key);
  return it == host.end() ? nullptr : &it->second;
}

int Net::new_tensor(int C, int H, int W, int dt_override, int bmul) {
  TensorDesc t;
  t.C = C; t.H = H; t.W = W;
  t.elems = (int64_t)C * H * W;
  t.dt_override = dt_override;
  t.bmul = bmul;
  max_bmul = std::max(max_bmul, bmul);
  tensors.push_back(t);
  return (int)tensors.size() - 1;
}

size_t Net::tbytes(int id, int B) const {
  const TensorDesc& t = tensors[id];
  int d = t.dt_override >= 0 ? t.dt_override : dt;
  if (t.x3img) return (size_t)B * t.bmul * t.elems * 6;   // three bf16 per element
  return (size_t)B * t.bmul * t.elems * dt_size(d);
}

// bf16x3: a GroupNorm whose output is read ONLY as the input of x3 convolutions writes the [xh | xl | xh] image directly (the apply
// pass's store), so those convolutions skip their split pass and no fp32 copy of the normalised tensor exists
void Net::mark_x3_images() {
  static const bool off = getenv("DDMI_X3_FUSED_SPLIT") && atoi(getenv("DDMI_X3_FUSED_SPLIT")) == 0;
  if (!x3 || off) return;
  std::vector<int> uses(tensors.size(), 0), conv_uses(tensors.size(), 0);
  auto use = [&](int t) { if (t >= 0) ++uses[t]; };
  for (const Op& o : ops) {
    use(o.in); use(o.in2); use(o.res); use(o.skip_in); use(o.skip_in2); use(o.stat_in); use(o.stat_in2);
    if (o.kind == OP_CONV && o.in >= 0 && convs[o.conv].w_x3 && o.in_boff == 0 && o.bmul == 1) ++conv_uses[o.in];
  }
  for (Op& o : ops)
    if (o.kind == OP_GN && o.bmul == 1 && o.out >= 0 && uses[o.out] > 0 && uses[o.out] == conv_uses[o.out] && tensors[o.out].C % 8 == 0)
      tensors[o.out].x3img = true;
}

int Net::dev_upload(const float* src, int64_t n, float** out) {
  void* p = nullptr;
  HIP_TRY(hipMalloc(&p, (size_t)n * sizeof(float)));
  dev_allocs.push_back(p);
  HIP_TRY(hipMemcpy(p, src, (size_t)n * sizeof(float), hipMemcpyHostToDevice));
  weight_bytes += (size_t)n * sizeof(float);
  *out = (float*)p;
  return DDMI_OK;
}

int Net::make_conv_host(const std::vector<float>& w, int Cout, int Cin, int K, const std::vector<float>* bias, int* idx) {
  ConvW c;
  c.Cout = Cout; c.Cin = Cin; c.K = K;
  int64_t n = (int64_t)Cout * Cin * K * K;
  DDMI_REQUIRE((int64_t)w.size() == n, "conv weight size mismatch");
  float* raw = nullptr;
  HIP_TRY(hipMalloc((void**)&raw, (size_t)n * sizeof(float)));
  hipError_t e = hipMemcpy(raw, w.data(), (size_t)n * sizeof(float), hipMemcpyHostToDevice);
  if (e != hipSuccess) { (void)hipFree(raw); return ddmi_fail(DDMI_ERR_HIP, "weight upload failed"); }
  // bf16x3: the hi / hi / lo image, three bf16 per weight (a Cin that is not a multiple of 8 keeps the fp32 kernel)
  const bool as_x3 = x3 && Cin % 8 == 0;
  const size_t pbytes = as_x3 ? (size_t)n * 3 * 2 : (size_t)n * dt_size(dt);
  void* packed = nullptr;
  e = hipMalloc(&packed, pbytes);
  if (e != hipSuccess) { (void)hipFree(raw); return ddmi_fail(DDMI_ERR_HIP, "weight alloc failed"); }
  dev_allocs.push_back(packed);
  int rc = as_x3 ? launch_pack_conv_weight_x3(raw, packed, Cout, Cin, K, K, 0) : launch_pack_conv_weight(raw, packed, dt, Cout, Cin, K, K, 0);
  hipError_t e2 = hipDeviceSynchronize();
  (void)hipFree(raw);
  if (rc != DDMI_OK) return rc;
  if (e2 != hipSuccess) return ddmi_fail(DDMI_ERR_HIP, "weight pack failed");
  weight_bytes += pbytes;
  if (as_x3) c.w_x3 = packed;
  else c.w = packed;
  if (bias) {
    DDMI_REQUIRE((int)bias->size() == Cout, "bias size mismatch");
    DDMI_TRY(dev_upload(bias->data(), Cout, &c.bias));
  }
  convs.push_back(c);
  *idx = (int)convs.size() - 1;
  return DDMI_OK;
}

int Net::make_conv(const std::string& wkey, const std::string& bkey, int* idx) {
  const HostTensor* w = find(wkey);
  if (!w) return ddmi_fail(DDMI_ERR_MISSING_WEIGHT, "missing weight '" + wkey + "'");
  DDMI_REQUIRE(w->shape.size() >= 2 && w->shape.size() <= 4, "conv weight must be 2-, 3- or 4-D: " + wkey);
  int K = w->shape.size() == 4 ? (int)w->shape[2] : 1;
  if (w->shape.size() == 4) DDMI_REQUIRE(w->shape[2] == w->shape[3], "conv kernel must be square: " + wkey);
  if (w->shape.size() == 3) DDMI_REQUIRE(w->shape[2] == 1, "conv1d kernel must be 1: " + wkey);
  const std::vector<float>* bias = nullptr;
  if (!bkey.empty()) {
    const HostTensor* b = find(bkey);
    if (!b) return ddmi_fail(DDMI_ERR_MISSING_WEIGHT, "missing weight '" + bkey + "'");
    DDMI_REQUIRE(b->numel() == w->shape[0], "bias size mismatch: " + bkey);
    bias = &b->data;
  }
  DDMI_TRY(make_conv_host(w->data, (int)w->shape[0], (int)w->shape[1], K, bias, idx));
  convs[*idx].wkey = wkey;
  return DDMI_OK;
}

int Net::make_attnblock(const std::vector<float>& w, const std::vector<float>& b, int C, int heads, int* idx) {
  DDMI_REQUIRE(dt == DT_BF16 && !x3 && heads * 32 == C && C % 64 == 0, "fused attention block: bf16 networks, head dim 32");
  DDMI_REQUIRE((int64_t)w.size() == (int64_t)3 * C * C && (int)b.size() == 3 * C, "fused attention block: qkv shape mismatch");
  const size_t wbytes = attnblock_weight_bytes(C, heads);
  std::vector<uint16_t> wp(wbytes / 2);
  std::vector<float> bp((size_t)heads * 96);
  attnblock_pack_weights(w.data(), b.data(), C, heads, wp.data(), bp.data());
  AttnBlockW a;
  a.C = C; a.heads = heads;
  HIP_TRY(hipMalloc(&a.w, wbytes));
  dev_allocs.push_back(a.w);
  HIP_TRY(hipMemcpy(a.w, wp.data(), wbytes, hipMemcpyHostToDevice));
  DDMI_TRY(dev_upload(bp.data(), (int64_t)bp.size(), &a.bias));
  weight_bytes += wbytes;
  attnblocks.push_back(a);
  *idx = (int)attnblocks.size() - 1;
  return DDMI_OK;
}

int Net::make_gn(const std::string& prefix, int* idx) {
  const HostTensor* g = find(prefix + "weight");
  const HostTensor* b = find(prefix + "bias");
  if (!g) return ddmi_fail(DDMI_ERR_MISSING_WEIGHT, "missing weight '" + prefix + "weight'");
  if (!b) return ddmi_fail(DDMI_ERR_MISSING_WEIGHT, "missing weight '" + prefix + "bias'");
  DDMI_REQUIRE(g->numel() == b->numel(), "groupnorm affine size mismatch: " + prefix);
  GNW w;
  w.C = (int)g->numel();
  DDMI_TRY(dev_upload(g->data.data(), w.C, &w.gamma));
  DDMI_TRY(dev_upload(b->data.data(), w.C, &w.beta));
  gns.push_back(w);
  *idx = (int)gns.size() - 1;
  return DDMI_OK;
}

int Net::op_gn(int in, int gn, int silu, float eps, int in2) {
  const TensorDesc& t = tensors[in];
  Op o;
  o.kind = OP_GN;
  o.in = in;
  o.in2 = in2;
  o.out = new_tensor(t.C + (in2 >= 0 ? tensors[in2].C : 0), t.H, t.W, -1, t.bmul);
  o.bmul = t.bmul;
  o.gn = gn;
  o.silu = silu;
  o.eps = eps;
  o.gn_slot = n_gn_slots++;
  ops.push_back(o);
  return o.out;
}

bool Net::px4_ok(int in) const {
  static const bool off = getenv("DDMI_CONV_PX4") && atoi(getenv("DDMI_CONV_PX4")) == 0;   // diagnostics: the 3x3 kernels at the 2 x 2 level too
  const TensorDesc& t = tensors[in];
  return !off && dt == DT_BF16 && !x3 && t.H == 2 && t.W == 2 && t.bmul == 1;
}

int Net::op_conv(int in, int conv, int stride, int ups, int res, int emb_off, float scale, bool emb_px4) {
  const TensorDesc& t = tensors[in];
  ConvW& c = convs[conv];
  int pad = c.K / 2;
  // 3x3 on a 2 x 2 image: the dense 1x1 form (ConvW::w_px4).  Needs the raw weights at plan() (a state-dict key) and channel counts
  // the vector epilogues take; an embedding addend only when the caller repeated its rows for the four pixels
  const bool px4 = px4_ok(in) && c.K == 3 && stride == 1 && ups == 0 && !c.wkey.empty() && c.Cin % 8 == 0 && c.Cout % 8 == 0 &&
                   (emb_off < 0 || emb_px4);
  if (px4) c.want_px4 = true;
  // x2 nearest upsample + 3x3 conv in the bf16 network: four 2x2 parity convolutions on the source grid (k_conv.hip).
  // Only marked here - op_conv returns a TENSOR id, so nothing that can fail lives in it; plan() packs the weights.
  static const bool up4 = !(getenv("DDMI_CONV_UP4") && atoi(getenv("DDMI_CONV_UP4")) == 0);
  if (up4 && dt == DT_BF16 && ups == 1 && c.K == 3 && stride == 1 && res < 0 && emb_off < 0) c.want_up4 = true;
  int hin = ups == 1 ? 2 * t.H : t.H, win = ups ? 2 * t.W : t.W;
  int ho = (hin + 2 * pad - c.K) / stride + 1, wo = (win + 2 * pad - c.K) / stride + 1;
  Op o;
  o.kind = OP_CONV;
  o.in = in;
  o.out = new_tensor(c.Cout, ho, wo, -1, t.bmul);
  o.bmul = t.bmul;
  o.conv = conv;
  o.stride = stride;
  o.pad = pad;
  o.ups = ups;
  o.res = res;
  o.emb_off = emb_off;
  o.px4 = px4 ? 1 : 0;
  o.scale = scale;
  ops.push_back(o);
  return o.out;
}

bool Net::cat_conv_ok(int in, int in2, int conv) const {
  static const bool off = getenv("DDMI_CONV_CAT") && atoi(getenv("DDMI_CONV_CAT")) == 0;
  if (off || dt != DT_BF16 || x3 || conv < 0) return false;
  // small-batch regime: the persistent dual-source kernel has no split-K - at B = 1 it runs on 6 workgroups for 70 us where the two
  // accumulating split-K launches take 2 x 14 (measured, round 6: 3.94 -> 3.69 ms per evaluation at B = 1, 5.78 -> 5.45 at B = 8)
  if (net_small_batch_enabled() && max_batch * max_bmul <= net_small_batch_max()) return false;
  const TensorDesc &t1 = tensors[in], &t2 = tensors[in2];
  const ConvW& c = convs[conv];
  if (c.K != 1 || !c.w || c.Cin != t1.C + t2.C || t1.H != t2.H || t1.W != t2.W || t1.bmul != t2.bmul) return false;
  ConvArgs a{};
  a.x = (const void*)1; a.xcat = (const void*)1; a.y = (void*)1;   // shape check only
  a.Cin = c.Cin; a.Ccat = t2.C; a.Cout = c.Cout; a.KH = a.KW = 1; a.stride = 1; a.Hin = a.Hout = t1.H; a.Win = a.Wout = t1.W; a.B = 1;
  a.splitk = 1;
  return conv_cat_ok(a, dt);
}

int Net::op_conv_cat(int in, int in2, int conv) {
  const int out = op_conv(in, conv, 1, 0, -1, -1);
  ops.back().in2 = in2;
  return out;
}

bool Net::can_fold_skip(int in, int conv) const {
  // OFF by default: measured on one box (B = 512, same process order, round 3) the folded launch is SLOWER than conv2 + the separate
  // 1x1 launches (41.9 vs 30.1 ms of a 139 ms UNet step; 32.5 vs ~29.5 for the variant that staged the skip sources as halos):
  // a one-tap stage fills 60 KB of LDS for the MFMAs a 3x3 stage runs on 34 KB, and the extra scalar state of the second block
  // list spills 44 SGPRs in the stage loop.  DDMI_SKIP_FOLD=1 enables it (kept: parity-tested through ddmi_op_conv2d_skip).
  static const bool off = !(getenv("DDMI_SKIP_FOLD") && atoi(getenv("DDMI_SKIP_FOLD")) == 1);
  const TensorDesc& t = tensors[in];
  return !off && dt == DT_BF16 && convs[conv].K == 3 && t.bmul == 1 && conv_halo_shape_ok(t.H, t.W);
}

int Net::op_conv_skip(int in, int conv, int skip_in, int skip_in2, int skip_conv, int* out) {
  const ConvW &c = convs[conv], &cs = convs[skip_conv];
  const int Cs = tensors[skip_in].C + (skip_in2 >= 0 ? tensors[skip_in2].C : 0);
  DDMI_REQUIRE(can_fold_skip(in, conv), "op_conv_skip: shape cannot fold");
  DDMI_REQUIRE(cs.K == 1 && cs.Cin == Cs && cs.Cout == c.Cout && c.bias && cs.bias, "op_conv_skip: skip weights do not match");
  std::vector<float> b1(c.Cout), b2(c.Cout);
  HIP_TRY(hipMemcpy(b1.data(), c.bias, c.Cout * sizeof(float), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(b2.data(), cs.bias, c.Cout * sizeof(float), hipMemcpyDeviceToHost));
  for (int i = 0; i < c.Cout; ++i) b1[i] += b2[i];
  float* bf = nullptr;
  DDMI_TRY(dev_upload(b1.data(), c.Cout, &bf));
  *out = op_conv(in, conv, 1, 0, -1, -1);
  Op& o = ops.back();
  o.skip_in = skip_in; o.skip_in2 = skip_in2; o.skip_conv = skip_conv; o.bias_fused = bf;
  return DDMI_OK;
}

// ------------------------------------------------------------------------------------------------------
static void fill_conv_args(const Net& n, const Op& o, int B, ConvArgs* a) {
  const TensorDesc& ti = n.tensors[o.in];
  const ConvW& c = n.convs[o.conv];
  *a = ConvArgs{};
  a->x = n.tptr(o.in);
  a->w = c.w;
  a->bias = c.bias;
  a->B = B; a->Hin = ti.H; a->Win = ti.W; a->Cin = c.Cin;
  a->Cout = c.Cout;
  a->KH = a->KW = c.K; a->stride = o.stride; a->pad = o.pad; a->ups = o.ups;
  int hin = o.ups == 1 ? 2 * ti.H : ti.H, win = o.ups ? 2 * ti.W : ti.W;
  a->Hout = (hin + 2 * o.pad - c.K) / o.stride + 1;
  a->Wout = (win + 2 * o.pad - c.K) / o.stride + 1;
  if (o.ups == 1 && c.w_up4 && o.res < 0 && o.emb_off < 0) {  // parity form: same output, 2x2 taps on the source grid
    a->w = c.w_up4; a->ups = 3; a->KH = a->KW = 2; a->pad = 1;
  }
  if (o.px4 && c.w_px4) {   // dense form on the 2 x 2 image: [B][1][1][4 Cin] -> [B][1][1][4 Cout] over the same memory
    a->w = c.w_px4; a->bias = c.bias_px4;
    a->Hin = a->Win = a->Hout = a->Wout = 1;
    a->Cin = 4 * c.Cin; a->Cout = 4 * c.Cout;
    a->KH = a->KW = 1; a->pad = 0;
  }
  a->out_scale = o.scale;
  a->splitk = 1;
  if (o.kind == OP_CONV && o.in2 >= 0) {   // [in | in2] read in place (op_conv_cat)
    a->xcat = n.tptr(o.in2);
    a->Ccat = n.tensors[o.in2].C;
  }
  if (o.skip_conv >= 0) {
    a->x2 = n.tptr(o.skip_in); a->C2 = n.tensors[o.skip_in].C;
    if (o.skip_in2 >= 0) { a->x3 = n.tptr(o.skip_in2); a->C3 = n.tensors[o.skip_in2].C; }
    a->w2 = n.convs[o.skip_conv].w;
    a->bias = o.bias_fused;
  }
}

// bf16x3: the bf16 launch that stands for the fp32 convolution `f` - K widened to 3 Cin over the [xh | xl | xh] image of its
// input, raw accumulators to the fp32 slab(s); bias / embedding / residual / store happen in launch_splitk_finish(f, DT_F32)
static ConvArgs x3_launch_args(const ConvArgs& f, const void* x3_in, const void* w_x3, float* ws) {
  ConvArgs a = f;
  a.x = x3_in; a.w = w_x3; a.Cin = 3 * f.Cin;
  a.bias = nullptr; a.emb = nullptr; a.res = nullptr; a.y = nullptr; a.y_nchw = nullptr; a.chstat = nullptr;
  a.out_scale = 1.f;
  a.force_ws = 1; a.splitk_ws = ws;
  return a;
}

// dense 2 x 2 form [4 Cout][4 Cin] of the convolutions op_conv marked (want_px4), from the raw fp32 weights (rounded once)
int Net::pack_px4_weights() {
  for (ConvW& c : convs) {
    if (!c.want_px4 || c.w_px4) continue;
    const HostTensor* raw = c.wkey.empty() ? nullptr : find(c.wkey);
    DDMI_REQUIRE(raw && raw->numel() == (int64_t)c.Cout * c.Cin * 9, "dense 2x2 convolution: the raw 3x3 weight is gone");
    const int Ci = c.Cin, Co = c.Cout;
    std::vector<float> w4((size_t)16 * Co * Ci);
    for (int po = 0; po < 4; ++po)
      for (int co = 0; co < Co; ++co)
        for (int pi = 0; pi < 4; ++pi) {
          const int kh = (pi >> 1) - (po >> 1) + 1, kw = (pi & 1) - (po & 1) + 1;   // input pixel = output pixel + (tap - 1)
          const float* src = raw->data.data() + ((size_t)co * Ci) * 9 + kh * 3 + kw;
          float* dst = w4.data() + ((size_t)(po * Co + co) * 4 + pi) * Ci;
          for (int ci = 0; ci < Ci; ++ci) dst[ci] = src[(size_t)ci * 9];
        }
    float* tmp = nullptr;
    HIP_TRY(hipMalloc((void**)&tmp, w4.size() * sizeof(float)));
    const size_t bytes = w4.size() * dt_size(dt);
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) { (void)hipFree(tmp); return ddmi_fail(DDMI_ERR_HIP, "weight alloc failed"); }
    dev_allocs.push_back(p);
    e = hipMemcpy(tmp, w4.data(), w4.size() * sizeof(float), hipMemcpyHostToDevice);
    int rc = e == hipSuccess ? launch_pack_conv_weight(tmp, p, dt, 4 * Co, 4 * Ci, 1, 1, 0) : ddmi_fail(DDMI_ERR_HIP, "weight upload failed");
    hipError_t e2 = hipDeviceSynchronize();
    (void)hipFree(tmp);
    DDMI_TRY(rc);
    HIP_TRY(e2);
    weight_bytes += bytes;
    c.w_px4 = p;
    if (c.bias) {
      std::vector<float> b(Co), b4((size_t)4 * Co);
      HIP_TRY(hipMemcpy(b.data(), c.bias, Co * sizeof(float), hipMemcpyDeviceToHost));
      for (int po = 0; po < 4; ++po) std::copy(b.begin(), b.end(), b4.begin() + (size_t)po * Co);
      DDMI_TRY(dev_upload(b4.data(), (int64_t)b4.size(), &c.bias_px4));
    }
  }
  return DDMI_OK;
}

// parity-form weights [4][Cout][4][Cin] of the convolutions op_conv marked (want_up4); every failure is returned
int Net::pack_up4_weights() {
  for (ConvW& c : convs) {
    if (!c.want_up4 || c.w_up4) continue;
    const size_t bytes = (size_t)16 * c.Cout * c.Cin * dt_size(dt);
    void* p = nullptr;
    HIP_TRY(hipMalloc(&p, bytes));
    dev_allocs.push_back(p);  // owned by the net from here on, whatever happens below
    const HostTensor* raw = c.wkey.empty() ? nullptr : find(c.wkey);
    if (raw && raw->numel() == (int64_t)c.Cout * c.Cin * 9) {   // sum the taps in fp32, round once
      float* tmp = nullptr;
      HIP_TRY(hipMalloc((void**)&tmp, raw->data.size() * sizeof(float)));
      hipError_t e = hipMemcpy(tmp, raw->data.data(), raw->data.size() * sizeof(float), hipMemcpyHostToDevice);
      int rc = e == hipSuccess ? launch_pack_conv_up4_f32(tmp, p, c.Cout, c.Cin, 0) : ddmi_fail(DDMI_ERR_HIP, "weight upload failed");
      hipError_t e2 = hipDeviceSynchronize();
      (void)hipFree(tmp);  // on every path
      DDMI_TRY(rc);
      HIP_TRY(e2);
    } else {
      DDMI_TRY(launch_pack_conv_up4(c.w, p, c.Cout, c.Cin, 0));
      HIP_TRY(hipDeviceSynchronize());
    }
    weight_bytes += bytes;
    c.w_up4 = p;
  }
  return DDMI_OK;
}

// A GroupNorm's statistics pass re-reads a tensor some convolution has just written.  Where every source of a GN input is the
// plain NHWC output of a bf16 convolution, that convolution gets a small companion tensor ([pixels / 32][C][2] fp32 at most) for
// the per-channel sums of its epilogue (k_conv.hip conv_epilogue_fast); whether a launch can fill it is decided per launch
// (conv_chstat_rows), so run() falls back to the pass whenever a producer could not.
void Net::attach_channel_stats() {
  static const bool off = getenv("DDMI_GN_CHSTAT") && atoi(getenv("DDMI_GN_CHSTAT")) == 0;
  if (off || dt != DT_BF16) return;
  // producer[t] = the ONE op that writes tensor t when that op is a plain convolution over the whole tensor; -2 = written by
  // anything else as well (plane-stacked convolutions write slices of one tensor, other op kinds deliver no sums)
  std::vector<int> producer(tensors.size(), -1);
  auto writes = [&](int t, int op, bool plain_conv) {
    if (t < 0) return;
    producer[t] = (plain_conv && producer[t] == -1) ? op : -2;
  };
  for (size_t i = 0; i < ops.size(); ++i) {
    const Op& o = ops[i];
    // (a plane-stacked convolution with shared weights is ONE launch over bmul x B samples: as plain as any other; only the per-plane
    // launches that write slices of one tensor - out_boff / in_boff - are not)
    writes(o.out, (int)i, o.kind == OP_CONV && !o.out_nchw && o.out_boff == 0 && o.in_boff == 0 && o.bmul == tensors[o.out].bmul);
    writes(o.out2, (int)i, false);
  }
  const size_t n_ops = ops.size();
  for (size_t i = 0; i < n_ops; ++i) {
    if (ops[i].kind != OP_GN) continue;
    const int in = ops[i].in, in2 = ops[i].in2;
    if (producer[in] < 0 || (in2 >= 0 && producer[in2] < 0)) continue;
    if (ops[producer[in]].bmul != ops[i].bmul || (in2 >= 0 && ops[producer[in2]].bmul != ops[i].bmul)) continue;
    const TensorDesc& t = tensors[in];
    if ((t.H * t.W) % 64 != 0) continue;
    auto stat_tensor = [&](int src) {
      Op& p = ops[producer[src]];
      if (p.stat_out < 0) {
        const TensorDesc& ts = tensors[src];
        p.stat_out = new_tensor(2 * ts.C, (ts.H * ts.W) / 32, 1, DT_F32, ts.bmul);
      }
      return p.stat_out;
    };
    ops[i].stat_in = stat_tensor(in);
    if (in2 >= 0) ops[i].stat_in2 = stat_tensor(in2);
  }
}

// Producer-side GroupNorm: a GroupNorm whose single input is written by ONE plain convolution and read by nothing else (the
// ResBlock's out_layers norm on h = conv1 + emb, openaimodel.py:261-270) can leave that convolution's epilogue already normalised.
// Only marked here; whether a launch qualifies is decided per run (conv_fgn_ok: bf16 halo-kernel launch, no split-K, whole groups
// per channel tile ...), and run() keeps the separate pass otherwise.  DDMI_GN_FUSE=0 / ddmi_debug_gn_fuse(0) keep every pass.
static int g_gn_fuse_override = -1;
void net_gn_fuse_override(int mode) { g_gn_fuse_override = mode; }
// 0 = never, 1 = where the launch would run on the halo kernel without split-K anyway, 2 = wherever the SHAPE allows (the launch is
// moved to the halo kernel without split-K: the parity tests' small batches)
static int gn_fuse_mode() {
  if (g_gn_fuse_override >= 0) return g_gn_fuse_override;
  static const int env = getenv("DDMI_GN_FUSE") ? atoi(getenv("DDMI_GN_FUSE")) : 1;
  return env < 0 ? 1 : env;
}
static bool gn_fuse_enabled() { return gn_fuse_mode() != 0; }
int net_gn_fuse_mode() { return gn_fuse_mode(); }
int g_net_small_gn_launches = 0;
static int g_small_batch_override = -1;
void net_small_batch_override(int mode) { g_small_batch_override = mode; }
bool net_small_batch_enabled() {
  if (g_small_batch_override >= 0) return g_small_batch_override != 0;
  static const bool on = !(getenv("DDMI_SMALL_BATCH") && atoi(getenv("DDMI_SMALL_BATCH")) == 0);
  return on;
}
int net_small_batch_max() {
  static const int m = getenv("DDMI_SMALL_BATCH_MAX") ? atoi(getenv("DDMI_SMALL_BATCH_MAX")) : 16;
  return m;
}

void Net::attach_fused_groupnorm() {
  if (dt != DT_BF16 || x3 || !gn_fuse_enabled()) return;
  std::vector<int> producer(tensors.size(), -1), readers(tensors.size(), 0);
  for (size_t i = 0; i < ops.size(); ++i) {
    const Op& o = ops[i];
    auto writes = [&](int t, bool plain_conv) {
      if (t >= 0) producer[t] = (plain_conv && producer[t] == -1) ? (int)i : -2;
    };
    writes(o.out, o.kind == OP_CONV && !o.out_nchw && o.out_boff == 0 && o.in_boff == 0 && o.bmul == tensors[o.out].bmul && o.res < 0 &&
                      o.skip_conv < 0 && !o.px4 && o.ups == 0 && o.stride == 1 && o.scale == 1.f && convs[o.conv].K == 3);
    writes(o.out2, false); writes(o.tmp0, false); writes(o.tmp1, false); writes(o.tmp2, false);
    for (int t : {o.in, o.in2, o.res, o.skip_in, o.skip_in2}) if (t >= 0) ++readers[t];
  }
  for (size_t i = 0; i < ops.size(); ++i) {
    Op& g = ops[i];
    if (g.kind != OP_GN || g.in2 >= 0 || g.frag_out || producer[g.in] < 0 || readers[g.in] != 1) continue;
    Op& c = ops[producer[g.in]];
    if (c.bmul != g.bmul || c.fgn_gn >= 0) continue;
    const TensorDesc& t = tensors[g.in];
    if ((t.H * t.W) % 256 != 0 || t.C % 32 != 0) continue;
    c.fgn_gn = (int)i;
    g.fgn_conv = producer[g.in];
  }
}

int Net::async_status(const hipStream_t* stream) {
  if (!fgn_sync) return DDMI_OK;
  unsigned w[CONV_FGN_SYNC_WORDS] = {0, 0, 0, 0};
  if (stream) {
    // the host-mapped word is all that is looked at: a timed-out workgroup wrote it with a system-scope store before its kernel ended
    HIP_TRY(hipStreamSynchronize(*stream));
  } else {
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(w, fgn_sync, sizeof w, hipMemcpyDeviceToHost));
  }
  if (w[1] || async_error_pending()) {
    if (stream) HIP_TRY(hipDeviceSynchronize());   // the reset below writes the device word: nothing may still be polling it
    HIP_TRY(hipMemset(fgn_sync + 1, 0, sizeof(unsigned)));
    if (fgn_flag_host) *(volatile unsigned*)fgn_flag_host = 0u;
    fgn_enabled = false;   // this network keeps the separate pass from here on: whoever runs it again gets valid results
    ++fgn_timeouts;
    return ddmi_fail(DDMI_ERR_STATE, "a fused GroupNorm launch timed out waiting for a partner tile (workgroup dispatch order not as "
                                     "assumed): the results of that run are invalid; the fusion is now off for this network - run again");
  }
  return DDMI_OK;
}

void Net::debug_inject_fgn_timeout() {
  if (!fgn_sync) return;
  const unsigned one = 1u;
  (void)hipDeviceSynchronize();
  (void)hipMemcpy(fgn_sync + 1, &one, sizeof one, hipMemcpyHostToDevice);
  if (fgn_flag_host) *(volatile unsigned*)fgn_flag_host = 1u;
}

int Net::plan() {
  DDMI_TRY(pack_up4_weights());
  DDMI_TRY(pack_px4_weights());
  attach_fused_groupnorm();
  attach_channel_stats();
  mark_x3_images();
  const int nops = (int)ops.size();
  auto touch = [&](int id, int i, bool write) {
    if (id < 0) return;
    TensorDesc& t = tensors[id];
    if (t.first < 0) t.first = i;
    (void)write;
    t.last = std::max(t.last, i);
  };
  for (int i = 0; i < nops; ++i) {
    const Op& o = ops[i];
    touch(o.in, i, false); touch(o.in2, i, false); touch(o.res, i, false); touch(o.skip_in, i, false); touch(o.skip_in2, i, false);
    touch(o.out, i, true); touch(o.out2, i, true); touch(o.tmp0, i, true); touch(o.tmp1, i, true); touch(o.tmp2, i, true);
    touch(o.stat_out, i, true); touch(o.stat_in, i, false); touch(o.stat_in2, i, false);
    if (o.fgn_gn >= 0) touch(ops[o.fgn_gn].out, i, true);   // a fused launch writes the GroupNorm's output itself
  }
  // first-fit allocator over op order
  struct Blk { size_t off, size; };
  std::vector<Blk> free_list;
  size_t top = 0;
  auto alloc = [&](size_t bytes) -> size_t {
    bytes = (bytes + 255) & ~(size_t)255;
    for (size_t i = 0; i < free_list.size(); ++i) {
      if (free_list[i].size >= bytes) {
        size_t off = free_list[i].off;
        free_list[i].off += bytes;
        free_list[i].size -= bytes;
        if (free_list[i].size == 0) free_list.erase(free_list.begin() + i);
        return off;
      }
    }
    // extend; merge with a trailing free block if it touches the top
    if (!free_list.empty() && free_list.back().off + free_list.back().size == top) {
      size_t off = free_list.back().off;
      top = off + bytes;
      free_list.pop_back();
      return off;
    }
    size_t off = top;
    top += bytes;
    return off;
  };
  auto release = [&](size_t off, size_t bytes) {
    bytes = (bytes + 255) & ~(size_t)255;
    Blk b{off, bytes};
    auto it = std::lower_bound(free_list.begin(), free_list.end(), b, [](const Blk& x, const Blk& y) { return x.off < y.off; });
    it = free_list.insert(it, b);
    // coalesce with neighbours
    size_t i = it - free_list.begin();
    if (i + 1 < free_list.size() && free_list[i].off + free_list[i].size == free_list[i + 1].off) {
      free_list[i].size += free_list[i + 1].size;
      free_list.erase(free_list.begin() + i + 1);
    }
    if (i > 0 && free_list[i - 1].off + free_list[i - 1].size == free_list[i].off) {
      free_list[i - 1].size += free_list[i].size;
      free_list.erase(free_list.begin() + i);
    }
  };
  for (int i = 0; i < nops; ++i) {
    for (size_t t = 0; t < tensors.size(); ++t)
      if (tensors[t].first == i) tensors[t].off = alloc(tbytes((int)t, max_batch));
    for (size_t t = 0; t < tensors.size(); ++t)
      if (tensors[t].last == i && tensors[t].first >= 0) release(tensors[t].off, tbytes((int)t, max_batch));
  }
  arena_bytes = std::max<size_t>(top, 256);
  HIP_TRY(hipMalloc((void**)&arena, arena_bytes));
  dev_allocs.push_back(arena);

  if (n_gn_slots > 0) {
    size_t bytes = GN_STATS_DOUBLES(max_batch * max_bmul) * sizeof(double);  // one scratch shared by every GN op (stream order)
    HIP_TRY(hipMalloc((void**)&gn_stats, bytes));
    dev_allocs.push_back(gn_stats);
  }
  bool any_fgn = false;
  for (const Op& o : ops) any_fgn = any_fgn || o.fgn_gn >= 0;
  if (any_fgn) {
    const size_t pbytes = CONV_FGN_PART_FLOATS((size_t)max_batch * max_bmul) * sizeof(float);
    HIP_TRY(hipMalloc((void**)&fgn_part, pbytes));
    dev_allocs.push_back(fgn_part);
    HIP_TRY(hipMemset(fgn_part, 0, pbytes));   // tag 0 never matches an epoch (the first launch's is 1)
    HIP_TRY(hipMalloc((void**)&fgn_sync, CONV_FGN_SYNC_WORDS * sizeof(unsigned)));
    dev_allocs.push_back(fgn_sync);
    HIP_TRY(hipMemset(fgn_sync, 0, CONV_FGN_SYNC_WORDS * sizeof(unsigned)));
    HIP_TRY(hipHostMalloc((void**)&fgn_flag_host, 64, hipHostMallocMapped));
    *(volatile unsigned*)fgn_flag_host = 0u;
    // the hand-off's progress argument (k_conv.hip) counts on 8 XCDs of 32 CUs with one workgroup per CU: any other device (another
    // gfx9 part, a CU-masked or partitioned MI355X) keeps the fusion to the shapes that need no hand-off
    int dev = 0;
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDevice(&dev));
    HIP_TRY(hipGetDeviceProperties(&prop, dev));
    if (!(prop.multiProcessorCount == 256 && strncmp(prop.gcnArchName, "gfx950", 6) == 0)) fgn_handoff = false;
  }
  // split-K slabs: worst case over batch sizes
  splitk_bytes = 0;
  const int cdt = x3 ? DT_BF16 : dt;   // the dtype the convolution kernels compute in
  for (const Op& o : ops) {
    if (o.kind != OP_CONV) continue;
    const bool ox3 = x3 && convs[o.conv].w_x3;
    for (int B = 1; B <= max_batch; ++B) {
      ConvArgs a;
      fill_conv_args(*this, o, B * o.bmul, &a);
      if (ox3) a = x3_launch_args(a, nullptr, nullptr, nullptr);
      int sk = conv_pick_splitk(a, ox3 ? DT_BF16 : dt);
      if (sk > 1 || ox3) splitk_bytes = std::max(splitk_bytes, (size_t)sk * a.B * a.Hout * a.Wout * a.Cout * sizeof(float));
      if (ox3) x3_in_bytes = std::max(x3_in_bytes, (size_t)a.B * a.Hin * a.Win * a.Cin * 2);
    }
  }
  if (x3_in_bytes) {
    HIP_TRY(hipMalloc(&x3_in, x3_in_bytes));
    dev_allocs.push_back(x3_in);
  }
  if (splitk_bytes) {
    HIP_TRY(hipMalloc((void**)&splitk_ws, splitk_bytes));
    dev_allocs.push_back(splitk_ws);
  }
  // ---- autotune every distinct conv shape at max_batch (bf16 perf mode only) -----------------------------
  const char* env = getenv("DDMI_AUTOTUNE");
  if (env && env[0] == '0') autotune = false;
  if (autotune && cdt == DT_BF16) {
    auto t0 = std::chrono::steady_clock::now();
    size_t tws_bytes = std::max<size_t>(splitk_bytes, (size_t)256 << 20);
    float* tws = nullptr;
    HIP_TRY(hipMalloc((void**)&tws, tws_bytes));
    std::map<std::vector<int>, ConvPlan> cache;
    // plan files: one line per shape "B Hin Win Cin Cout K stride ups skipC cfg order splitk us" after a "v <CONV_PLAN_VERSION>"
    // header (a file written for other kernel tables is ignored).  DDMI_TUNE_FILE = read-write cache (new shapes are
    // appended); DDMI_TUNE_DEFAULT = the read-only plans shipped with the package (ddmi_amd/tuned/), consulted for
    // shapes the cache does not hold.  Timing-based picks differ from run to run; pinned plans make bf16 results
    // bit-identical across processes and GPUs (ddmi_amd/dist.py broadcasts rank 0's file for the same reason).
    auto read_plans = [&](const char* path, bool overwrite) {
      FILE* f = path ? fopen(path, "r") : nullptr;
      if (!f) return;
      int ver = -1;
      if (fscanf(f, " v %d", &ver) == 1 && ver == CONV_PLAN_VERSION) {
        int v[12];
        float us;
        while (fscanf(f, "%d %d %d %d %d %d %d %d %d %d %d %d %f", &v[0], &v[1], &v[2], &v[3], &v[4], &v[5], &v[6], &v[7],
                      &v[8], &v[9], &v[10], &v[11], &us) == 13) {
          ConvPlan p;
          p.cfg = v[9]; p.order = v[10]; p.splitk = v[11]; p.us = us;
          std::vector<int> key(v, v + 9);
          if (overwrite || !cache.count(key)) cache[key] = p;
        }
      }
      fclose(f);
    };
    const char* tune_file = getenv("DDMI_TUNE_FILE");
    read_plans(tune_file, true);
    std::map<std::vector<int>, ConvPlan> own = cache;   // what the read-write file holds
    read_plans(getenv("DDMI_TUNE_DEFAULT"), false);
    std::map<std::vector<int>, ConvPlan> used;   // what this network's shapes resolved to (written back to the cache file)
    int rc = DDMI_OK;
    for (Op& o : ops) {
      if (o.kind != OP_CONV) continue;
      ConvArgs a;
      fill_conv_args(*this, o, max_batch * o.bmul, &a);
      a.y = tptr(o.out);  // timing only: results are overwritten by the real run
      if (!x3) {
        // time what will run: the embedding addend decides which epilogue a tile takes (a 128-row wave tile on an 8x8 plane), the
        // consumer GroupNorm's statistics are epilogue work for the tiles that deliver them and a separate pass for those that do not
        if (o.emb_off >= 0 && (size_t)a.B * a.Cout * sizeof(float) <= tws_bytes) { a.emb = tws; a.emb_stride = a.Cout; }
        if (o.stat_out >= 0) a.chstat = (float*)tptr(o.stat_out);
      }
      if (x3) {
        if (!convs[o.conv].w_x3) continue;   // stays on the fp32 kernel
        a = x3_launch_args(a, x3_in, convs[o.conv].w_x3, tws);   // tuned as the bf16 problem it runs as (slab output)
      }
      std::vector<int> key{a.B, a.Hin, a.Win, a.Cin, a.Cout, a.KH, a.stride, a.ups,
                           a.x2 ? a.C2 + a.C3 : a.xcat ? 1000000 + a.Ccat : 0};   // last: folded skip channels / second source of a concatenated input
      auto it = cache.find(key);
      // a (cfg 0, 0 us) entry is the PLACEHOLDER an older strict run left in a plan file, not a measurement: treated as absent
      if (it != cache.end() && it->second.cfg == 0 && it->second.us == 0.f) { cache.erase(it); own.erase(key); it = cache.end(); }
      if (it == cache.end()) {
        ConvPlan p;
        // DDMI_TUNE_STRICT=1: a shape the plan files do not hold takes the heuristic plan instead of a timing-based pick, so the
        // tile / split-K (and with them the bf16 rounding order) of every convolution is the same on every box and in every run -
        // what the parity tests run under (tests/conftest.py).  That plan never reaches `used`: through it it would land in a
        // read-write DDMI_TUNE_FILE (and in the plan text broadcast to other ranks), where a later non-strict run would take it for
        // a tuned shape and never autotune it
        static const bool strict = getenv("DDMI_TUNE_STRICT") && atoi(getenv("DDMI_TUNE_STRICT")) != 0;
        if (strict) {
          o.tuned = ConvPlan{};
          continue;
        }
        rc = conv_autotune(a, cdt, tws, tws_bytes, &p);
        if (rc != DDMI_OK) break;
        it = cache.emplace(key, p).first;
      }
      used[key] = it->second;
      o.tuned = it->second;
      if (o.tuned.splitk > 1)
        splitk_bytes = std::max(splitk_bytes, (size_t)o.tuned.splitk * a.B * a.Hout * a.Wout * a.Cout * sizeof(float));
    }
    (void)hipDeviceSynchronize();
    (void)hipFree(tws);
    if (rc != DDMI_OK) return rc;
    if (tune_file) {   // the cache file = its own entries + every shape of this network (so it alone reproduces the run)
      const size_t before = own.size();
      for (auto& kv : used) own.emplace(kv.first, kv.second);
      if (own.size() != before) {
        if (FILE* f = fopen(tune_file, "w")) {
          fprintf(f, "v %d\n", CONV_PLAN_VERSION);
          for (auto& kv : own) {
            for (int x : kv.first) fprintf(f, "%d ", x);
            fprintf(f, "%d %d %d %.2f\n", kv.second.cfg, kv.second.order, kv.second.splitk, kv.second.us);
          }
          fclose(f);
        }
      }
    }
    if (splitk_bytes) {  // re-size the slabs for the tuned splits
      if (splitk_ws) {
        (void)hipFree(splitk_ws);
        dev_allocs.erase(std::remove(dev_allocs.begin(), dev_allocs.end(), (void*)splitk_ws), dev_allocs.end());
      }
      HIP_TRY(hipMalloc((void**)&splitk_ws, splitk_bytes));
      dev_allocs.push_back(splitk_ws);
    }
    tune_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  }
  if (cdt == DT_BF16 && autotune) {
    // slabs must also cover smaller batches run with the tuned tile and a heuristic split
    size_t need = splitk_bytes;
    for (const Op& o : ops) {
      if (o.kind != OP_CONV) continue;
      for (int B = 1; B < max_batch; ++B) {
        ConvArgs a;
        fill_conv_args(*this, o, B * o.bmul, &a);
        const bool ox3 = x3 && convs[o.conv].w_x3;
        if (ox3) a = x3_launch_args(a, nullptr, nullptr, nullptr);
        a.cfg = o.tuned.cfg;
        int sk = conv_pick_splitk(a, ox3 ? DT_BF16 : dt);
        if (sk > 1 || ox3) need = std::max(need, (size_t)sk * a.B * a.Hout * a.Wout * a.Cout * sizeof(float));
      }
    }
    if (need > splitk_bytes) {
      if (splitk_ws) {
        (void)hipFree(splitk_ws);
        dev_allocs.erase(std::remove(dev_allocs.begin(), dev_allocs.end(), (void*)splitk_ws), dev_allocs.end());
      }
      splitk_bytes = need;
      HIP_TRY(hipMalloc((void**)&splitk_ws, splitk_bytes));
      dev_allocs.push_back(splitk_ws);
    }
  }
  host.clear();  // raw fp32 copies are no longer needed
  finalized = true;
  return DDMI_OK;
}

int Net::run(int B, void* const* ext, const float* emb, int emb_stride, hipStream_t s) {
  if (!finalized) return ddmi_fail(DDMI_ERR_STATE, "network not finalized");
  DDMI_REQUIRE(B >= 1 && B <= max_batch, "batch exceeds max_batch given at create time");
  const int Bnet = B;
  std::vector<int> stat_rows(tensors.size(), 0);   // rows per block of the channel statistics a conv wrote for its output (0: none)
  std::vector<char> emb_out(tensors.size(), 0);    // diagnostics (DDMI_GN_ABL): tensors written by a convolution that adds a time embedding
  std::vector<char> fgn_done(ops.size(), 0);       // GroupNorm ops whose producer convolution wrote the normalised tensor itself this run
  fgn_launches = 0;
  // small-batch regime (net.h): one-launch GroupNorm
  const bool small = dt == DT_BF16 && !x3 && net_small_batch_enabled() && Bnet * max_bmul <= net_small_batch_max();
  g_net_small_gn_launches = 0;
  size_t op_index = 0;
  for (const Op& o : ops) {
    const size_t oi = op_index++;
    const int B = Bnet * o.bmul;  // samples this op runs on
    auto plane_ptr = [&](int id, int boff) -> void* {
      return (unsigned char*)tptr(id) + (size_t)boff * Bnet * tensors[id].elems * dt_size(dt);
    };
    switch (o.kind) {
      case OP_NCHW_IN: {
        const TensorDesc& t = tensors[o.out];
        DDMI_TRY(launch_nchw_to_nhwc((const float*)ext[o.ext], tptr(o.out), dt, B, t.C, t.H * t.W, s));
        break;
      }
      case OP_GN: {
        if (fgn_done[oi]) break;   // its producer's epilogue wrote the normalised tensor
        const TensorDesc& t = tensors[o.in];
        const GNW& g = gns[o.gn];
        // in2 >= 0: GroupNorm of the channel concatenation [in | in2] without materialising it
        const int C = t.C + (o.in2 >= 0 ? tensors[o.in2].C : 0);
        const void* x2 = o.in2 >= 0 ? tptr(o.in2) : nullptr;
        DDMI_REQUIRE(g.C == C, "groupnorm channel mismatch");
        if (small && !o.frag_out && !tensors[o.out].x3img && gn_small_ok(B, t.H * t.W, C, x2 != nullptr)) {   // one launch: statistics + apply
          GnSmallArgs ga{};
          ga.x = tptr(o.in); ga.x2 = x2; ga.C1 = t.C;
          ga.y = tptr(o.out); ga.gamma = g.gamma; ga.beta = g.beta; ga.eps = o.eps; ga.silu = o.silu;
          ga.B = B; ga.HW = t.H * t.W; ga.C = C;
          ++g_net_small_gn_launches;
          DDMI_TRY(launch_gn_small(ga, s));
          break;
        }
        double* st = gn_stats;
        {  // TIMING-ONLY (DDMI_GN_ABL=1, with a -DGN_ABL build of k_conv.hip and DDMI_CONV_DBG=32): the out_layers GroupNorm passes are
           // skipped, their consumers read stale memory - prices the producer-side GroupNorm before it is built
          static const bool abl = getenv("DDMI_GN_ABL") && atoi(getenv("DDMI_GN_ABL")) == 1;
          if (abl && o.in2 < 0 && emb_out[o.in]) break;
        }
        static const bool fused_finish = !(getenv("DDMI_GN_FINISH_PASS") && atoi(getenv("DDMI_GN_FINISH_PASS")) == 1);   // diagnostics: the round-2 finish launch
        if (fused_finish && o.stat_in >= 0 && stat_rows[o.in] > 0 && (o.in2 < 0 || (o.stat_in2 >= 0 && stat_rows[o.in2] > 0))) {
          DDMI_TRY(launch_gn_apply_chstat(tptr(o.in), tptr(o.out), (const float*)tptr(o.stat_in), stat_rows[o.in],
                                          o.in2 >= 0 ? (const float*)tptr(o.stat_in2) : nullptr, o.in2 >= 0 ? stat_rows[o.in2] : 0, g.gamma, g.beta,
                                          o.eps, o.silu, B, t.H * t.W, C, s, x2, t.C, o.frag_out));
          break;
        }
        if (o.stat_in >= 0 && stat_rows[o.in] > 0 && (o.in2 < 0 || (o.stat_in2 >= 0 && stat_rows[o.in2] > 0)))
          DDMI_TRY(launch_gn_chstat_finish((const float*)tptr(o.stat_in), stat_rows[o.in], o.in2 >= 0 ? (const float*)tptr(o.stat_in2) : nullptr,
                                           o.in2 >= 0 ? stat_rows[o.in2] : 0, t.C, st, B, t.H * t.W, C, s));
        else
          DDMI_TRY(launch_gn_stats(tptr(o.in), dt, st, B, t.H * t.W, C, s, 0, 1, x2, t.C));
        DDMI_TRY(launch_gn_apply(tptr(o.in), tptr(o.out), dt, st, g.gamma, g.beta, o.eps, o.silu, B, t.H * t.W, C, s, nullptr,
                                 1, x2, t.C, tensors[o.out].x3img ? 1 : 0, o.frag_out));
        break;
      }
      case OP_CONV: {
        ConvArgs a;
        fill_conv_args(*this, o, B, &a);
        DDMI_REQUIRE(tensors[o.in].C + (o.in2 >= 0 ? tensors[o.in2].C : 0) == convs[o.conv].Cin, "conv input channel mismatch");
        a.x = plane_ptr(o.in, o.in_boff);
        if (o.out_nchw) a.y_nchw = (float*)ext[o.ext];
        else a.y = plane_ptr(o.out, o.out_boff);
        if (o.res >= 0) a.res = tptr(o.res);
        if (o.emb_off >= 0) {
          emb_out[o.out] = 1;
          DDMI_REQUIRE(emb != nullptr, "conv needs the time embedding");
          a.emb = emb + o.emb_off;
          a.emb_stride = emb_stride;
          a.emb_mod = o.bmul > 1 ? Bnet : 0;
        }
        if (Bnet == max_batch && o.tuned.cfg > 0) {
          a.cfg = o.tuned.cfg; a.order = o.tuned.order; a.splitk = o.tuned.splitk;
        } else {
          a.cfg = o.tuned.cfg;  // keep the tuned tile, re-derive the split for this batch size
          a.splitk = conv_pick_splitk(a, dt);
        }
        if (a.xcat) a.splitk = 1;   // the dual-source 1x1 form has no split-K (persistent kernel)
        a.splitk_ws = splitk_ws;
        if (a.splitk > 1)
          DDMI_REQUIRE((size_t)a.splitk * a.B * a.Hout * a.Wout * a.Cout * sizeof(float) <= splitk_bytes, "split-K workspace too small");
        if (x3 && convs[o.conv].w_x3) {
          // three bf16 products per fp32 product: split the input, run the bf16 kernel over K = 3 Cin into the fp32 slab(s),
          // finish (+ bias + embedding + residual, store) in fp32
          const bool pre_split = tensors[o.in].x3img;   // its GroupNorm wrote the [xh | xl | xh] image already
          if (!pre_split) DDMI_TRY(launch_split_x3((const float*)a.x, x3_in, (int64_t)a.B * a.Hin * a.Win, a.Cin, s));
          ConvArgs b = x3_launch_args(a, pre_split ? a.x : x3_in, convs[o.conv].w_x3, splitk_ws);
          if (!(Bnet == max_batch && o.tuned.cfg > 0)) b.splitk = conv_pick_splitk(b, DT_BF16);
          b.splitk = std::max(1, b.splitk);
          if (b.splitk == 1) {   // one launch: the bf16 kernel's (general) epilogue adds bias / embedding / fp32 residual and stores fp32
            b.force_ws = 0; b.splitk_ws = nullptr;
            b.bias = a.bias; b.emb = a.emb; b.emb_stride = a.emb_stride; b.emb_mod = a.emb_mod; b.out_scale = a.out_scale;
            b.res32 = (const float*)a.res; b.y32 = (float*)a.y; b.y_nchw = a.y_nchw;
            DDMI_TRY(launch_conv(b, DT_BF16, s));
            break;
          }
          DDMI_REQUIRE((size_t)b.splitk * a.B * a.Hout * a.Wout * a.Cout * sizeof(float) <= splitk_bytes, "split-K workspace too small");
          DDMI_TRY(launch_conv(b, DT_BF16, s));
          a.splitk = b.splitk;
          DDMI_TRY(launch_splitk_finish(a, DT_F32, s));
          break;
        }
        if (o.fgn_gn >= 0 && fgn_fusing()) {   // producer-side GroupNorm where this launch qualifies
          const Op& gop = ops[o.fgn_gn];
          const GNW& g = gns[gop.gn];
          ConvArgs f = a;
          f.gn_gamma = g.gamma; f.gn_beta = g.beta; f.gn_eps = gop.eps; f.gn_silu = gop.silu;
          f.gn_part = fgn_part; f.gn_sync = fgn_sync; f.gn_flag_host = fgn_flag_host;
          f.gn_no_handoff = fgn_handoff ? 0 : 1;
          f.gn_any_batch = gn_fuse_mode() == 2;
          f.y = tptr(gop.out);
          if (f.cfg == 0 || gn_fuse_mode() == 2) {   // no tuned tile for this shape (or forced): the halo kernel is the 3x3 kernel of choice
            f.cfg = (a.Cout % 224 == 0 ? 19 : 20) + 1;
            f.splitk = gn_fuse_mode() == 2 ? 1 : conv_pick_splitk(f, dt);
          }
          if (g.C == a.Cout && conv_fgn_ok(f, dt)) {
            DDMI_TRY(launch_conv(f, dt, s));
            fgn_done[o.fgn_gn] = 1;
            ++fgn_launches;
            break;
          }
        }
        if (o.stat_out >= 0 && !small) {
          const int rows = conv_chstat_rows(a, dt);
          if (rows > 0) a.chstat = (float*)tptr(o.stat_out);
          stat_rows[o.out] = rows;
        }
        DDMI_TRY(launch_conv(a, dt, s));
        break;
      }
      case OP_SEG_IN: {
        const TensorDesc &txy = tensors[o.out], &tt = tensors[o.out2];
        const int nxy = txy.H * txy.W, nt = tt.H * tt.W;
        const float* src = (const float*)ext[o.ext];
        DDMI_TRY(launch_nchw_to_nhwc(src, tptr(o.out), dt, Bnet, txy.C, nxy, s, o.seq_len, 0));
        DDMI_TRY(launch_nchw_to_nhwc(src, plane_ptr(o.out2, 0), dt, Bnet, tt.C, nt, s, o.seq_len, nxy));
        DDMI_TRY(launch_nchw_to_nhwc(src, plane_ptr(o.out2, 1), dt, Bnet, tt.C, nt, s, o.seq_len, nxy + nt));
        break;
      }
      case OP_SEG_OUT: {
        const TensorDesc &txy = tensors[o.in], &tt = tensors[o.in2];
        const int nxy = txy.H * txy.W, nt = tt.H * tt.W;
        float* dst = (float*)ext[o.ext];
        DDMI_TRY(launch_nhwc_to_nchw(tptr(o.in), dt, dst, Bnet, txy.C, nxy, s, o.seq_len, 0));
        DDMI_TRY(launch_nhwc_to_nchw(plane_ptr(o.in2, 0), dt, dst, Bnet, tt.C, nt, s, o.seq_len, nxy));
        DDMI_TRY(launch_nhwc_to_nchw(plane_ptr(o.in2, 1), dt, dst, Bnet, tt.C, nt, s, o.seq_len, nxy + nt));
        break;
      }
      case OP_GN_JOINT: {
        const TensorDesc &txy = tensors[o.in], &tt = tensors[o.in2];
        const GNW& g = gns[o.gn];
        DDMI_REQUIRE(g.C == txy.C && g.C == tt.C, "groupnorm channel mismatch");
        const int hw[3] = {txy.H * txy.W, tt.H * tt.W, tt.H * tt.W};
        const void* src[3] = {tptr(o.in), plane_ptr(o.in2, 0), plane_ptr(o.in2, 1)};
        void* dst[3] = {tptr(o.out), plane_ptr(o.out2, 0), plane_ptr(o.out2, 1)};
        for (int i = 0; i < 3; ++i) DDMI_TRY(launch_gn_stats(src[i], dt, gn_stats, Bnet, hw[i], g.C, s, i, 3));
        for (int i = 0; i < 3; ++i)
          DDMI_TRY(launch_gn_apply(src[i], dst[i], dt, gn_stats, g.gamma, g.beta, o.eps, o.silu, Bnet, hw[i], g.C, s, hw, 3));
        break;
      }
      case OP_ATTN_MH: {
        const TensorDesc &ti = tensors[o.in], &to = tensors[o.out];
        AttnArgs a{};
        a.heads = o.heads; a.d = o.head_dim; a.scale = o.scale; a.q_prescaled = o.q_prescaled;
        a.x3 = x3 ? 1 : 0;
        a.legacy_name = o.legacy_layout;
        a.ld_qkv = ti.C; a.ld_out = to.C;
        if (o.legacy_layout) {
          a.q_off = 0; a.k_off = o.head_dim; a.v_off = 2 * o.head_dim; a.hs_qkv = 3 * o.head_dim;
        } else {
          a.q_off = 0; a.k_off = o.heads * o.head_dim; a.v_off = 2 * o.heads * o.head_dim; a.hs_qkv = o.head_dim;
        }
        a.hs_out = o.head_dim;
        a.qkv[0] = tptr(o.in); a.out[0] = tptr(o.out);
        a.seg_len[0] = ti.H * ti.W;
        if (o.in2 >= 0) {  // tokens of a sample: xy | xt | yt
          const TensorDesc& tt = tensors[o.in2];
          DDMI_REQUIRE(tt.C == ti.C && tensors[o.out2].C == to.C, "attention: plane channel mismatch");
          a.qkv[1] = plane_ptr(o.in2, 0); a.qkv[2] = plane_ptr(o.in2, 1);
          a.out[1] = plane_ptr(o.out2, 0); a.out[2] = plane_ptr(o.out2, 1);
          a.seg_len[1] = a.seg_len[2] = tt.H * tt.W;
          a.B = Bnet;
        } else {
          a.B = B;
        }
        DDMI_TRY(launch_attention(a, dt, s));
        break;
      }
      case OP_ATTN_BLOCK: {
        const TensorDesc& t = tensors[o.in];
        const AttnBlockW& w = attnblocks[o.conv];
        DDMI_REQUIRE(w.C == t.C && tensors[o.out].C == t.C, "fused attention block: channel mismatch");
        DDMI_TRY(launch_attnblock_qkv_attn(tptr(o.in), w.w, w.bias, tptr(o.out), B, t.H * t.W, t.C, w.heads, s));
        break;
      }
      case OP_ATTN_LEGACY: {
        const TensorDesc& t = tensors[o.out];
        DDMI_TRY(launch_attention_legacy(tptr(o.in), tptr(o.out), dt, B, t.H * t.W, o.heads, t.C / o.heads, s));
        break;
      }
      case OP_ATTN_FULL: {
        // single-head attention over T = H*W tokens with C channels: q = in, k = in2, v = res
        const TensorDesc& t = tensors[o.out];
        int T = t.H * t.W, C = t.C;
        void *q = tptr(o.in), *k = tptr(o.in2), *v = tptr(o.res), *S = tptr(o.tmp0), *vT = tptr(o.tmp1);
        DDMI_TRY(launch_gemm_nt(q, k, S, dt, B, T, T, C, (int64_t)T * C, (int64_t)T * C, (int64_t)T * T, o.scale, s));
        DDMI_TRY(launch_softmax_rows(S, dt, (int64_t)B * T, T, s));
        DDMI_TRY(launch_transpose_2d(v, vT, dt, B, T, C, s));
        DDMI_TRY(launch_gemm_nt(S, vT, tptr(o.out), dt, B, T, C, T, (int64_t)T * T, (int64_t)C * T, (int64_t)T * C, 1.0f, s));
        break;
      }
      case OP_CONCAT: {
        const TensorDesc &ta = tensors[o.in], &tb = tensors[o.in2];
        DDMI_TRY(launch_concat(tptr(o.in), ta.C, tptr(o.in2), tb.C, tptr(o.out), dt, (int64_t)B * ta.H * ta.W, s));
        break;
      }
      case OP_EXPORT: {
        const TensorDesc& t = tensors[o.in];
        if (ext[o.ext] == nullptr) break;
        const void* src = plane_ptr(o.in, o.in_boff);
        if (o.out_nchw)
          DDMI_TRY(launch_nhwc_to_nchw(src, dt, (float*)ext[o.ext], Bnet, t.C, t.H * t.W, s));
        else
          HIP_TRY(hipMemcpyAsync(ext[o.ext], src, (size_t)Bnet * t.elems * dt_size(dt), hipMemcpyDeviceToDevice, s));
        break;
      }
      case OP_PLANES_TO_CHANNELS: {  // [3*Bn][HW][C] (plane-major) -> [Bn][HW][3C]
        const TensorDesc& t = tensors[o.in];
        DDMI_TRY(launch_planes_channels(tptr(o.in), tptr(o.out), dt, Bnet, t.H * t.W, t.C, 3, 0, s));
        break;
      }
      case OP_CHANNELS_TO_PLANES: {
        const TensorDesc& t = tensors[o.out];
        DDMI_TRY(launch_planes_channels(tptr(o.out), tptr(o.in), dt, Bnet, t.H * t.W, t.C, 3, 1, s));
        break;
      }
      default:
        return ddmi_fail(DDMI_ERR_STATE, "unknown op");
    }
  }
  return DDMI_OK;
}
