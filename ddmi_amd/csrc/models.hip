// Graph builders for the two conv networks on the path:
//   * LDM UNet            - models/ldm/modules/diffusionmodules/openaimodel.py:463-793 (UNetModel)
//   * D2C-VAE image decoder - models/d2c_vae/autoencoder_unet.py:702-829 (Decoder) + :855-858 (Autoencoder.decode)
// Weight names are the reference's state-dict keys, so `ldm-last.pt` tensors can be handed over unchanged.
#include "models.h"
#include "prof.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <algorithm>

// =====================================================================================================
// UNet
// =====================================================================================================
static bool in_list(const int* v, int n, int x) {
  for (int i = 0; i < n; ++i)
    if (v[i] == x) return true;
  return false;
}

// one ResBlock (openaimodel.py:244-276, no up/down, use_scale_shift_norm=False):
//   h = conv3(SiLU(GN(x))) + emb_proj ; h = conv3(SiLU(GN(h))) ; out = skip(x) + h
// Weights are created once (make_*) and may be emitted on several tensors (the video UNet runs every block on the
// xy plane and on the stacked xt / yt planes with the same weights).
int make_unet_resblock(UNet& u, const std::string& pre, int cin, int cout, ResW* w, int split, int emb_rep) {
  Net& n = u.net;
  w->eps = 1e-5f;
  DDMI_TRY(n.make_gn(pre + "in_layers.0.", &w->gn1));
  DDMI_TRY(n.make_conv(pre + "in_layers.2.weight", pre + "in_layers.2.bias", &w->c1));
  DDMI_REQUIRE(n.convs[w->c1].Cin == cin && n.convs[w->c1].Cout == cout, "resblock conv shape mismatch at " + pre);
  // time-embedding projection rows are appended to one big [Etot, emb_dim] matrix
  const HostTensor* ew = n.find(pre + "emb_layers.1.weight");
  const HostTensor* eb = n.find(pre + "emb_layers.1.bias");
  if (!ew || !eb) return ddmi_fail(DDMI_ERR_MISSING_WEIGHT, "missing weight '" + pre + "emb_layers.1.*'");
  DDMI_REQUIRE(ew->shape.size() == 2 && ew->shape[0] == cout && ew->shape[1] == u.emb_dim, "emb_layers shape mismatch at " + pre);
  w->emb_off = (int)u.emb_b_host.size();
  w->emb_rep = emb_rep;
  for (int r = 0; r < emb_rep; ++r) {   // (4: one copy of the rows per pixel of the 2 x 2 image, the dense convolution's channel order)
    u.emb_w_host.insert(u.emb_w_host.end(), ew->data.begin(), ew->data.end());
    u.emb_b_host.insert(u.emb_b_host.end(), eb->data.begin(), eb->data.end());
  }
  DDMI_TRY(n.make_gn(pre + "out_layers.0.", &w->gn2));
  DDMI_TRY(n.make_conv(pre + "out_layers.3.weight", pre + "out_layers.3.bias", &w->c2));
  w->cs = w->cs2 = -1;
  if (cin != cout && split > 0) {
    // the block's input is th.cat([h, skip], 1) (openaimodel.py:787) and is never materialised: the 1x1 skip_connection
    // becomes two accumulating 1x1 convolutions over the column blocks [0, split) and [split, cin) of its weight
    const HostTensor* wt = n.find(pre + "skip_connection.weight");
    const HostTensor* bt = n.find(pre + "skip_connection.bias");
    if (!wt || !bt) return ddmi_fail(DDMI_ERR_MISSING_WEIGHT, "missing weight '" + pre + "skip_connection.*'");
    DDMI_REQUIRE(wt->numel() == (int64_t)cout * cin && bt->numel() == cout, "skip_connection shape mismatch at " + pre);
    std::vector<float> wa((size_t)cout * split), wb((size_t)cout * (cin - split));
    for (int o = 0; o < cout; ++o) {
      std::copy(wt->data.begin() + (size_t)o * cin, wt->data.begin() + (size_t)o * cin + split, wa.begin() + (size_t)o * split);
      std::copy(wt->data.begin() + (size_t)o * cin + split, wt->data.begin() + (size_t)(o + 1) * cin,
                wb.begin() + (size_t)o * (cin - split));
    }
    DDMI_TRY(n.make_conv_host(wa, cout, split, 1, &bt->data, &w->cs));
    DDMI_TRY(n.make_conv_host(wb, cout, cin - split, 1, nullptr, &w->cs2));
    DDMI_TRY(n.make_conv(pre + "skip_connection.weight", pre + "skip_connection.bias", &w->cs_full));
  } else if (cin != cout) {
    DDMI_TRY(n.make_conv(pre + "skip_connection.weight", pre + "skip_connection.bias", &w->cs));
    DDMI_REQUIRE(n.convs[w->cs].K == 1, "only 1x1 skip connections are supported (use_conv=False)");
  }
  return DDMI_OK;
}

// ResBlock on the channel concatenation [x1 | x2] without materialising it: dual-source GroupNorm for in_layers, the
// skip connection as conv(x1) + conv(x2)
int emit_resblock_cat(Net& n, const ResW& w, int x1, int x2, int* out) {
  DDMI_REQUIRE(w.cs >= 0 && w.cs2 >= 0, "concat-free ResBlock needs a split skip connection");
  DDMI_REQUIRE(n.tensors[x1].C + n.tensors[x2].C == n.convs[w.c1].Cin, "resblock input channel mismatch");
  const bool fold = w.cs_full >= 0 && n.can_fold_skip(x1, w.c2);   // skip_connection as extra K columns of conv2 (k_conv.hip)
  int sk = -1;
  if (!fold) {
    if (w.cs_full >= 0 && n.cat_conv_ok(x1, x2, w.cs_full)) {   // ONE launch over [x1 | x2] read in place (k_conv.hip, CAT)
      sk = n.op_conv_cat(x1, x2, w.cs_full);
    } else {
      sk = n.op_conv(x1, w.cs, 1, 0, -1, -1);
      sk = n.op_conv(x2, w.cs2, 1, 0, sk, -1);
    }
  }
  int a = n.op_gn(x1, w.gn1, 1, w.eps, x2);
  int h = n.op_conv(a, w.c1, 1, 0, -1, w.emb_off, 1.f, w.emb_rep == 4);
  int b = n.op_gn(h, w.gn2, 1, w.eps);
  if (fold) return n.op_conv_skip(b, w.c2, x1, x2, w.cs_full, out);
  *out = n.op_conv(b, w.c2, 1, 0, sk, -1);
  return DDMI_OK;
}

int emit_resblock(Net& n, const ResW& w, int x, int* out) {
  DDMI_REQUIRE(n.tensors[x].C == n.convs[w.c1].Cin, "resblock input channel mismatch");
  const bool fold = w.cs >= 0 && w.cs2 < 0 && n.convs[w.cs].K == 1 && n.can_fold_skip(x, w.c2);
  int skip = x;
  if (w.cs >= 0 && !fold) skip = n.op_conv(x, w.cs, 1, 0, -1, -1);
  int a = n.op_gn(x, w.gn1, 1, w.eps);
  int h = n.op_conv(a, w.c1, 1, 0, -1, w.emb_off, 1.f, w.emb_rep == 4);
  int b = n.op_gn(h, w.gn2, 1, w.eps);
  if (fold) return n.op_conv_skip(b, w.c2, x, -1, w.cs, out);
  *out = n.op_conv(b, w.c2, 1, 0, skip, -1);
  return DDMI_OK;
}

static int build_resblock(UNet& u, const std::string& pre, int x, int cout, int* out) {
  ResW w;
  DDMI_TRY(make_unet_resblock(u, pre, u.net.tensors[x].C, cout, &w, 0, u.net.px4_ok(x) ? 4 : 1));
  return emit_resblock(u.net, w, x, out);
}

// AttentionBlock (openaimodel.py:319-325) with QKVAttentionLegacy, head dim = num_head_channels
// Which AttentionBlocks run as the fused qkv + attention launch (k_attnblock.hip).  DDMI_ATTN_FUSED=0 keeps the five-launch form
// everywhere (A/B and the fallback for shapes the fused kernel does not serve); DDMI_ATTN_FUSED_N=1024,256 restricts it to blocks
// with those token counts.
static int g_attn_fused_override = -1;   // ddmi_debug_attn_fused: -1 = the environment decides
void attnblock_fused_override(int mode) { g_attn_fused_override = mode; }
bool attnblock_fused_enabled(int n_tokens, int max_samples) {
  if (g_attn_fused_override >= 0) return g_attn_fused_override != 0;
  static const bool off = getenv("DDMI_ATTN_FUSED") && atoi(getenv("DDMI_ATTN_FUSED")) == 0;
  if (off) return false;
  // small-batch regime (net.h): a workgroup of the fused kernel is one head of 1024 token rows - 14 to 42 workgroups at B = 1, each
  // ~50 us whatever share of its rows is real; the split-K qkv convolution + the attention kernels take 0.4 ms less per evaluation
  // at B = 1 and 0.2 ms at B = 8 (measured, round 6).  DDMI_ATTN_FUSED=1 keeps the fused launch there too
  static const bool forced = getenv("DDMI_ATTN_FUSED") && atoi(getenv("DDMI_ATTN_FUSED")) == 1;
  if (!forced && max_samples > 0 && net_small_batch_enabled() && max_samples <= net_small_batch_max()) return false;
  const char* e = getenv("DDMI_ATTN_FUSED_N");
  if (!e || n_tokens <= 0) return true;
  for (const char* p = e; *p;) {
    if (atoi(p) == n_tokens) return true;
    while (*p && *p != ',') ++p;
    if (*p == ',') ++p;
  }
  return false;
}

int make_unet_attnblock(UNet& u, const std::string& pre, int ch, AttnW* w, int n_tokens) {
  Net& n = u.net;
  DDMI_REQUIRE(u.cfg.num_head_channels > 0 && ch % u.cfg.num_head_channels == 0, "channels must be a multiple of num_head_channels");
  w->eps = 1e-5f;
  DDMI_TRY(n.make_gn(pre + "norm.", &w->gn));
  w->head_dim = u.cfg.num_head_channels;
  w->heads = ch / w->head_dim;
  w->legacy = 1;
  w->scale = 1.0f / std::sqrt((float)w->head_dim);  // (d^-1/4 q) . (d^-1/4 k), openaimodel.py:413-416
  w->q_prescaled = attention_prescale_ok(n.dt, w->head_dim) ? 1 : 0;
  if (w->q_prescaled) {
    // bf16 network: softmax scale x log2 e folded into the q rows of the projection (QKVAttentionLegacy layout: per head
    // [q | k | v] blocks of head_dim rows, openaimodel.py:406-408), in fp32 before the one rounding to bf16 - the attention kernel's
    // base-2 exponent is then the score itself (k_attention.hip, LAZY)
    const HostTensor* wt = n.find(pre + "qkv.weight");
    const HostTensor* bt = n.find(pre + "qkv.bias");
    if (!wt || !bt) return ddmi_fail(DDMI_ERR_MISSING_WEIGHT, "missing weight '" + pre + "qkv.*'");
    DDMI_REQUIRE(wt->numel() == (int64_t)3 * ch * ch && bt->numel() == 3 * ch, "qkv shape mismatch at " + pre);
    std::vector<float> ws = wt->data, bs = bt->data;
    const float f = w->scale * 1.44269504088896341f;
    for (int h = 0; h < w->heads; ++h)
      for (int r = h * 3 * w->head_dim; r < h * 3 * w->head_dim + w->head_dim; ++r) {
        for (int c = 0; c < ch; ++c) ws[(size_t)r * ch + c] *= f;
        bs[r] *= f;
      }
    // fused qkv + attention launch (k_attnblock.hip): its own packing of the same pre-scaled projection; a block whose only token
    // count is served by it needs no separate qkv convolution
    const bool fusable = !n.x3 && w->head_dim == 32 && attnblock_shape_ok(n.dt, 1024, ch, 32, 1) &&
                         attnblock_fused_enabled(n_tokens, n.max_batch);
    if (fusable) DDMI_TRY(n.make_attnblock(ws, bs, ch, w->heads, &w->fused));
    if (!(fusable && n_tokens > 0 && attnblock_shape_ok(n.dt, n_tokens, ch, 32, n.max_batch * n.max_bmul)))
      DDMI_TRY(n.make_conv_host(ws, 3 * ch, ch, 1, &bs, &w->cqkv));
  } else {
    DDMI_TRY(n.make_conv(pre + "qkv.weight", pre + "qkv.bias", &w->cqkv));
  }
  DDMI_TRY(n.make_conv(pre + "proj_out.weight", pre + "proj_out.bias", &w->cp));
  if (w->cqkv >= 0) DDMI_REQUIRE(n.convs[w->cqkv].Cout == 3 * ch && n.convs[w->cqkv].Cin == ch, "qkv shape mismatch at " + pre);
  return DDMI_OK;
}

// attention over the tokens of ONE tensor (every batch entry - sample or stacked plane - on its own)
int emit_attn_single(Net& n, const AttnW& w, int x, int* out) {
  const TensorDesc t = n.tensors[x];
  int a = n.op_gn(x, w.gn, 0, w.eps);
  if (w.fused >= 0 && attnblock_fused_enabled(t.H * t.W, n.max_batch) && attnblock_shape_ok(n.dt, t.H * t.W, t.C, w.head_dim, n.max_batch * t.bmul)) {
    n.ops.back().frag_out = 1;   // the GroupNorm just emitted: its output only feeds the fused launch, which wants whole B fragments
    Op f;
    f.kind = OP_ATTN_BLOCK;
    f.in = a;
    f.out = n.new_tensor(t.C, t.H, t.W, -1, t.bmul);
    f.bmul = t.bmul;
    f.conv = w.fused;
    f.heads = w.heads; f.head_dim = w.head_dim;
    n.ops.push_back(f);
    *out = n.op_conv(f.out, w.cp, 1, 0, x, -1);
    return DDMI_OK;
  }
  DDMI_REQUIRE(w.cqkv >= 0, "attention block: no qkv projection for this token count");
  int qkv = n.op_conv(a, w.cqkv, 1, 0, -1, -1);
  Op o;
  o.kind = OP_ATTN_MH;
  o.in = qkv;
  o.out = n.new_tensor(w.heads * w.head_dim, t.H, t.W, -1, t.bmul);
  o.bmul = t.bmul;
  o.heads = w.heads; o.head_dim = w.head_dim; o.legacy_layout = w.legacy; o.scale = w.scale; o.q_prescaled = w.q_prescaled;
  n.ops.push_back(o);
  *out = n.op_conv(o.out, w.cp, 1, 0, x, -1);
  return DDMI_OK;
}

// attention over the concatenated tokens xy | xt | yt of every sample (openaimodel.py:1148-1157, ae:1066-1075)
int emit_attn_joint(Net& n, const AttnW& w, int xy, int tp, int* out_xy, int* out_tp) {
  const TensorDesc txy = n.tensors[xy], tt = n.tensors[tp];
  DDMI_REQUIRE(txy.bmul == 1 && tt.bmul == 2 && txy.C == tt.C, "joint attention expects an xy tensor and a stacked xt|yt tensor");
  Op g;
  g.kind = OP_GN_JOINT;
  g.in = xy; g.in2 = tp;
  g.out = n.new_tensor(txy.C, txy.H, txy.W, -1, 1);
  g.out2 = n.new_tensor(tt.C, tt.H, tt.W, -1, 2);
  g.gn = w.gn; g.eps = w.eps; g.silu = 0;
  g.gn_slot = n.n_gn_slots++;
  n.ops.push_back(g);
  int q_xy = n.op_conv(g.out, w.cqkv, 1, 0, -1, -1);
  int q_tp = n.op_conv(g.out2, w.cqkv, 1, 0, -1, -1);
  Op o;
  o.kind = OP_ATTN_MH;
  o.in = q_xy; o.in2 = q_tp;
  o.out = n.new_tensor(w.heads * w.head_dim, txy.H, txy.W, -1, 1);
  o.out2 = n.new_tensor(w.heads * w.head_dim, tt.H, tt.W, -1, 2);
  o.heads = w.heads; o.head_dim = w.head_dim; o.legacy_layout = w.legacy; o.scale = w.scale; o.q_prescaled = w.q_prescaled;
  n.ops.push_back(o);
  *out_xy = n.op_conv(o.out, w.cp, 1, 0, xy, -1);
  *out_tp = n.op_conv(o.out2, w.cp, 1, 0, tp, -1);
  return DDMI_OK;
}

// q / k / v stored as three Conv1d/Conv2d(C -> Cq) modules (autoencoder_unet.py:223-242, 279-298, 410-430, 476-496):
// fused into one [3 Cq, C] projection so the normalised activations are read once
int make_qkv_attn(Net& n, const std::string& pre, int ch, int heads, bool expand, AttnW* w) {
  w->eps = 1e-6f;
  DDMI_TRY(n.make_gn(pre + "norm.", &w->gn));
  const int cq = expand ? ch * heads : ch;
  std::vector<float> wcat, bcat;
  for (const char* nm : {"q.", "k.", "v."}) {
    const HostTensor* wt = n.find(pre + nm + "weight");
    const HostTensor* bt = n.find(pre + nm + "bias");
    if (!wt || !bt) return ddmi_fail(DDMI_ERR_MISSING_WEIGHT, "missing weight '" + pre + nm + "*'");
    DDMI_REQUIRE(wt->numel() == (int64_t)cq * ch && bt->numel() == cq, "q/k/v projection shape mismatch at " + pre);
    wcat.insert(wcat.end(), wt->data.begin(), wt->data.end());
    bcat.insert(bcat.end(), bt->data.begin(), bt->data.end());
  }
  w->heads = heads;
  w->head_dim = cq / heads;
  w->legacy = 0;
  w->scale = 1.0f / std::sqrt((float)w->head_dim);  // xformers.ops.memory_efficient_attention default scale
  w->q_prescaled = attention_prescale_ok(n.dt, w->head_dim) ? 1 : 0;
  if (w->q_prescaled) {   // rows [0, cq) are q: softmax scale x log2 e folded in (see make_unet_attnblock)
    const float f = w->scale * 1.44269504088896341f;
    for (size_t i = 0; i < (size_t)cq * ch; ++i) wcat[i] *= f;
    for (int i = 0; i < cq; ++i) bcat[i] *= f;
  }
  DDMI_TRY(n.make_conv_host(wcat, 3 * cq, ch, 1, &bcat, &w->cqkv));
  DDMI_TRY(n.make_conv(pre + "proj_out.weight", pre + "proj_out.bias", &w->cp));
  DDMI_REQUIRE(n.convs[w->cp].Cin == cq && n.convs[w->cp].Cout == ch, "proj_out shape mismatch at " + pre);
  return DDMI_OK;
}

static int build_attnblock(UNet& u, const std::string& pre, int x, int* out) {
  AttnW w;
  DDMI_TRY(make_unet_attnblock(u, pre, u.net.tensors[x].C, &w, u.net.tensors[x].H * u.net.tensors[x].W));
  return emit_attn_single(u.net, w, x, out);
}

int unet_time_embed_setup(UNet& u) {
  Net& n = u.net;
  const ddmi_unet_cfg& c = u.cfg;
  DDMI_REQUIRE(c.num_head_channels >= 1 && c.num_head_channels <= 512, "num_head_channels must be in 1..512");
  DDMI_REQUIRE(c.n_mult >= 1 && c.n_mult <= 8 && c.n_attn >= 0 && c.n_attn <= 8, "bad channel_mult / attention_resolutions");
  DDMI_REQUIRE(c.model_channels % 32 == 0, "model_channels must be a multiple of 32 (GroupNorm32)");
  const int mc = c.model_channels;
  u.emb_dim = 4 * mc;
  // ---- time embedding MLP (openaimodel.py:557-562) -------------------------------------------------
  {
    int half = mc / 2;
    std::vector<float> fr(half);
    const float neg_log = (float)(-std::log(10000.0));
    for (int i = 0; i < half; ++i) {
      volatile float a = neg_log * (float)i;  // fp32 product, then fp32 division, as torch evaluates util.py:161-163
      volatile float q = a / (float)half;
      fr[i] = std::exp((float)q);
    }
    DDMI_TRY(n.dev_upload(fr.data(), half, &u.freqs));
    const char* keys[4] = {"time_embed.0.weight", "time_embed.0.bias", "time_embed.2.weight", "time_embed.2.bias"};
    float** dst[4] = {&u.te_w0, &u.te_b0, &u.te_w1, &u.te_b1};
    for (int i = 0; i < 4; ++i) {
      const HostTensor* t = n.find(keys[i]);
      if (!t) return ddmi_fail(DDMI_ERR_MISSING_WEIGHT, std::string("missing weight '") + keys[i] + "'");
      DDMI_TRY(n.dev_upload(t->data.data(), t->numel(), dst[i]));
    }
    DDMI_REQUIRE(n.find(keys[0])->shape[0] == u.emb_dim && n.find(keys[0])->shape[1] == mc, "time_embed.0 shape mismatch");
  }
  return DDMI_OK;
}

int unet_finalize(UNet& u) {
  Net& n = u.net;
  const ddmi_unet_cfg& c = u.cfg;
  DDMI_TRY(unet_time_embed_setup(u));
  const int mc = c.model_channels;
  // ---- topology (openaimodel.py:567-743) -------------------------------------------------------------
  const int S = c.image_size;
  Op in;
  in.kind = OP_NCHW_IN;
  in.ext = 0;
  in.out = n.new_tensor(c.in_channels, S, S);
  n.ops.push_back(in);
  int ci;
  DDMI_TRY(n.make_conv("input_blocks.0.0.weight", "input_blocks.0.0.bias", &ci));
  int h = n.op_conv(in.out, ci, 1, 0, -1, -1);
  std::vector<int> hs{h};
  int ch = mc, ds = 1, blk = 1;
  for (int level = 0; level < c.n_mult; ++level) {
    for (int r = 0; r < c.num_res_blocks; ++r) {
      std::string pre = "input_blocks." + std::to_string(blk) + ".";
      DDMI_TRY(build_resblock(u, pre + "0.", h, c.channel_mult[level] * mc, &h));
      ch = c.channel_mult[level] * mc;
      if (in_list(c.attention_resolutions, c.n_attn, ds)) DDMI_TRY(build_attnblock(u, pre + "1.", h, &h));
      hs.push_back(h);
      ++blk;
    }
    if (level != c.n_mult - 1) {
      std::string pre = "input_blocks." + std::to_string(blk) + ".0.";
      int cd;
      DDMI_TRY(n.make_conv(pre + "op.weight", pre + "op.bias", &cd));  // Downsample: conv3x3 stride 2 (openaimodel.py:152-154)
      h = n.op_conv(h, cd, 2, 0, -1, -1);
      hs.push_back(h);
      ds *= 2;
      ++blk;
    }
  }
  DDMI_TRY(build_resblock(u, "middle_block.0.", h, ch, &h));
  DDMI_TRY(build_attnblock(u, "middle_block.1.", h, &h));
  DDMI_TRY(build_resblock(u, "middle_block.2.", h, ch, &h));
  blk = 0;
  for (int level = c.n_mult - 1; level >= 0; --level) {
    for (int i = 0; i <= c.num_res_blocks; ++i) {
      std::string pre = "output_blocks." + std::to_string(blk) + ".";
      int skip = hs.back();
      hs.pop_back();
      const TensorDesc ta = n.tensors[h], tb = n.tensors[skip];
      DDMI_REQUIRE(ta.H == tb.H && ta.W == tb.W, "skip connection spatial mismatch");
      const int cout = mc * c.channel_mult[level];
      static const bool no_cat = !(getenv("DDMI_KEEP_CONCAT") && getenv("DDMI_KEEP_CONCAT")[0] == '1');  // diagnostics
      if (ta.C + tb.C != cout && no_cat) {  // th.cat([h, hs.pop()], 1) is read in place (dual-source GN, split skip conv)
        ResW w;
        DDMI_TRY(make_unet_resblock(u, pre + "0.", ta.C + tb.C, cout, &w, ta.C, n.px4_ok(h) ? 4 : 1));
        DDMI_TRY(emit_resblock_cat(n, w, h, skip, &h));
      } else {
        Op cat;
        cat.kind = OP_CONCAT;
        cat.in = h;
        cat.in2 = skip;
        cat.out = n.new_tensor(ta.C + tb.C, ta.H, ta.W);
        n.ops.push_back(cat);
        DDMI_TRY(build_resblock(u, pre + "0.", cat.out, cout, &h));
      }
      int sub = 1;
      if (in_list(c.attention_resolutions, c.n_attn, ds)) {
        DDMI_TRY(build_attnblock(u, pre + std::to_string(sub) + ".", h, &h));
        ++sub;
      }
      if (level && i == c.num_res_blocks) {
        int cu;
        std::string q = pre + std::to_string(sub) + ".conv.";
        DDMI_TRY(n.make_conv(q + "weight", q + "bias", &cu));  // Upsample: nearest x2 + conv3x3 (openaimodel.py:110-120)
        h = n.op_conv(h, cu, 1, 1, -1, -1);
        ds /= 2;
      }
      ++blk;
    }
  }
  int gno, co;
  DDMI_TRY(n.make_gn("out.0.", &gno));
  DDMI_TRY(n.make_conv("out.2.weight", "out.2.bias", &co));
  int a = n.op_gn(h, gno, 1, 1e-5f);
  int y = n.op_conv(a, co, 1, 0, -1, -1);
  n.ops.back().out_nchw = 1;
  n.ops.back().ext = 1;
  (void)y;
  return unet_finish(u);
}

int unet_finish(UNet& u) {
  Net& n = u.net;
  const ddmi_unet_cfg& c = u.cfg;
  const int mc = c.model_channels;
  // ---- concatenated emb projections ------------------------------------------------------------------
  u.emb_total = (int)u.emb_b_host.size();
  DDMI_TRY(n.dev_upload(u.emb_w_host.data(), (int64_t)u.emb_w_host.size(), &u.emb_w));
  DDMI_TRY(n.dev_upload(u.emb_b_host.data(), (int64_t)u.emb_b_host.size(), &u.emb_b));
  u.emb_w_host.clear();
  u.emb_w_host.shrink_to_fit();
  size_t fb = (size_t)c.max_batch * (mc + 2 * u.emb_dim + u.emb_total) * sizeof(float);
  HIP_TRY(hipMalloc((void**)&u.tbuf, fb));
  n.dev_allocs.push_back(u.tbuf);
  n.fgn_handoff = true;   // this handle's owner sees the hand-off's flag: unet_forward peeks it, ddmi_unet_async_status checks it
  return n.plan();
}

UNet::~UNet() {
  for (auto& g : graphs) (void)hipGraphExecDestroy(g.exec);
}

static int unet_forward_launch(UNet& u, const float* x, const int64_t* t, float* eps_out, int B, hipStream_t s);

int unet_async_status(UNet& u, const hipStream_t* stream) {
  const int rc = u.net.async_status(stream);
  if (rc != DDMI_OK) {
    for (auto& g : u.graphs) (void)hipGraphExecDestroy(g.exec);
    u.graphs.clear();
  }
  return rc;
}

int unet_forward(UNet& u, const float* x, const int64_t* t, float* eps_out, int B, hipStream_t s) {
  Net& n = u.net;
  if (!n.finalized) return ddmi_fail(DDMI_ERR_STATE, "unet not finalized");
  DDMI_REQUIRE(x && t && eps_out, "unet_forward: null pointer");
  DDMI_REQUIRE(B >= 1 && B <= n.max_batch, "unet_forward: batch exceeds max_batch");
  // a hand-off of an EARLIER forward has flagged a timeout (host-mapped word: no synchronisation to look at it): this call reports it
  // instead of running - every forward since the last ddmi_unet_async_status may be invalid; the fusion is off afterwards
  if (n.async_error_pending()) return unet_async_status(u);
  prof_begin_forward();
  if (!u.use_graph || prof_active()) return unet_forward_launch(u, x, t, eps_out, B, s);
  // (the small-batch regime's switch decides launches too - net.h: a graph captured under the other setting is not this call's)
  const int fuse = (n.fgn_fusing() ? net_gn_fuse_mode() : 0) | (net_small_batch_enabled() ? 8 : 0);
  for (auto& g : u.graphs)
    if (g.B == B && g.x == x && g.t == t && g.out == eps_out && g.uniform == u.uniform_t && g.fuse == fuse) {
      HIP_TRY(hipGraphLaunch(g.exec, s));
      return DDMI_OK;
    }
  // first call with these buffers: one plain run (lazy one-time setup such as kernel attributes must not happen inside a
  // capture), then capture the same launches and keep the executable graph
  DDMI_TRY(unet_forward_launch(u, x, t, eps_out, B, s));
  if (u.graphs.size() >= 8) return DDMI_OK;  // an unbounded set of buffers: stay on the launch path
  hipGraph_t graph = nullptr;
  if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) != hipSuccess) {
    // the legacy default stream (torch's current stream unless the caller switched) refuses capture: not an error of this
    // call - its result exists already - just no graph for this handle
    (void)hipGetLastError();
    if (getenv("DDMI_GRAPH_DEBUG")) fprintf(stderr, "ddmi: stream capture refused on this stream; staying on the launch path\n");
    u.use_graph = false;
    return DDMI_OK;
  }
  int rc = unet_forward_launch(u, x, t, eps_out, B, s);
  hipError_t e = hipStreamEndCapture(s, &graph);
  if (rc != DDMI_OK || e != hipSuccess || graph == nullptr) {
    if (getenv("DDMI_GRAPH_DEBUG"))
      fprintf(stderr, "ddmi: graph capture failed: rc=%d (%s) hip=%s\n", rc, ddmi_error_cstr(), hipGetErrorString(e));
    if (graph) (void)hipGraphDestroy(graph);
    (void)hipGetLastError();
    u.use_graph = false;  // capture is not possible here (e.g. a legacy-stream restriction): fall back for good
    return DDMI_OK;       // the plain run above already produced this call's result
  }
  hipGraphExec_t exec = nullptr;
  e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  (void)hipGraphDestroy(graph);
  if (e != hipSuccess) {
    if (getenv("DDMI_GRAPH_DEBUG")) fprintf(stderr, "ddmi: graph instantiate failed: %s\n", hipGetErrorString(e));
    u.use_graph = false;
    return DDMI_OK;
  }
  if (getenv("DDMI_GRAPH_DEBUG")) fprintf(stderr, "ddmi: captured a forward graph (B=%d)\n", B);
  u.graphs.push_back(UNet::GraphEntry{B, x, t, eps_out, u.uniform_t, fuse, exec});
  return DDMI_OK;  // the plain run above already produced this call's result
}

static int unet_forward_launch(UNet& u, const float* x, const int64_t* t, float* eps_out, int B, hipStream_t s) {
  Net& n = u.net;
  const int mc = u.cfg.model_channels, E = u.emb_dim;
  float* temb = u.tbuf;                       // [B][mc]
  float* e1 = temb + (size_t)B * mc;          // [B][E]
  float* e2 = e1 + (size_t)B * E;             // [B][E]
  float* eall = e2 + (size_t)B * E;           // [B][emb_total]
  // all chains of a sampling batch share the timestep (ddpm.py:294, 349): one embedding row, broadcast by stride 0
  const int Be = u.uniform_t ? 1 : B;
  DDMI_TRY(launch_timestep_embedding(t, u.freqs, temb, Be, mc, s));
  DDMI_TRY(launch_linear_small(temb, mc, u.te_w0, u.te_b0, e1, E, Be, mc, E, 0, s));
  DDMI_TRY(launch_linear_small(e1, E, u.te_w1, u.te_b1, e2, E, Be, E, E, 1, s));
  // every ResBlock's emb_layers = Linear(SiLU(emb)) (openaimodel.py:219-225) in one launch
  DDMI_TRY(launch_linear_small(e2, E, u.emb_w, u.emb_b, eall, u.emb_total, Be, E, u.emb_total, 1, s));
  void* ext[2] = {(void*)x, (void*)eps_out};
  return n.run(B, ext, eall, u.uniform_t ? 0 : u.emb_total, s);
}

// =====================================================================================================
// D2C-VAE image decoder
// =====================================================================================================
// ResnetBlock (autoencoder_unet.py:129-149) with temb=None
int make_vae_resblock(Net& n, const std::string& pre, int cin, int cout, ResW* w) {
  w->eps = 1e-6f;
  w->emb_off = -1;
  DDMI_TRY(n.make_gn(pre + "norm1.", &w->gn1));
  DDMI_TRY(n.make_conv(pre + "conv1.weight", pre + "conv1.bias", &w->c1));
  DDMI_TRY(n.make_gn(pre + "norm2.", &w->gn2));
  DDMI_TRY(n.make_conv(pre + "conv2.weight", pre + "conv2.bias", &w->c2));
  DDMI_REQUIRE(n.convs[w->c1].Cin == cin && n.convs[w->c1].Cout == cout, "decoder resblock shape mismatch at " + pre);
  w->cs = -1;
  if (cin != cout) DDMI_TRY(n.make_conv(pre + "nin_shortcut.weight", pre + "nin_shortcut.bias", &w->cs));
  return DDMI_OK;
}

static int build_vae_resblock(Net& n, const std::string& pre, int x, int cout, int* out) {
  ResW w;
  DDMI_TRY(make_vae_resblock(n, pre, n.tensors[x].C, cout, &w));
  return emit_resblock(n, w, x, out);
}

// AttnBlock with num_heads=1 (autoencoder_unet.py:187-215; make_attn 'vanilla' :561-567)
static int build_vae_attn(Net& n, const std::string& pre, int x, int* out) {
  const TensorDesc t = n.tensors[x];
  // ONE fused q | k | v projection and the STREAMING attention kernel with a single head of C channels (k_attention.hip: d = 64 .. 512,
  // >= 2048 tokens, whole 128 QF-query tiles: attention_stream_shape_ok) instead of GEMM -> softmax -> transpose -> GEMM over a materialised [T][T] score matrix (33.5 MB
  // per sample at the AFHQ decoder's 4096 tokens: 17 GB of arena and 67.6 GB through `softmax_rows` per launch at B = 512).  Round-3
  // verdict item 9.  Measured twice in round 4: on the generic kernel (S recomputed per 256-channel value slice, d = 512 QK^T from LDS
  // at a fifth of the MFMA rate) the decoder took 683 ms per 512 latents against 432; on the streaming kernel's d = 512 instantiation the
  // decode phase is 439.8 against 460.9 ms.  Default where the streaming kernel serves the shape; DDMI_VAE_ATTN_FLASH=0 keeps the three
  // GEMMs, =1 takes the attention kernels for every shape.
  const char* fe = getenv("DDMI_VAE_ATTN_FLASH");
  const int T0 = t.H * t.W;
  const bool streams = t.C >= 64 && attention_stream_shape_ok(t.C, T0);   // the launcher's own predicate: the flash form only where that kernel runs
  const bool flash = fe ? atoi(fe) == 1 : streams;
  if (flash && n.dt == DT_BF16 && !n.x3 && t.C <= 512 && t.C % 8 == 0) {
    AttnW w;
    DDMI_TRY(make_qkv_attn(n, pre, t.C, 1, false, &w));
    return emit_attn_single(n, w, x, out);
  }
  int gn, cq, ck, cv, cp;
  DDMI_TRY(n.make_gn(pre + "norm.", &gn));
  DDMI_TRY(n.make_conv(pre + "q.weight", pre + "q.bias", &cq));
  DDMI_TRY(n.make_conv(pre + "k.weight", pre + "k.bias", &ck));
  DDMI_TRY(n.make_conv(pre + "v.weight", pre + "v.bias", &cv));
  DDMI_TRY(n.make_conv(pre + "proj_out.weight", pre + "proj_out.bias", &cp));
  int a = n.op_gn(x, gn, 0, 1e-6f);
  int q = n.op_conv(a, cq, 1, 0, -1, -1);
  int k = n.op_conv(a, ck, 1, 0, -1, -1);
  int v = n.op_conv(a, cv, 1, 0, -1, -1);
  int T = t.H * t.W;
  Op o;
  o.kind = OP_ATTN_FULL;
  o.bmul = t.bmul;
  o.in = q; o.in2 = k; o.res = v;
  o.out = n.new_tensor(t.C, t.H, t.W, -1, t.bmul);
  o.tmp0 = n.new_tensor(T, T, 1, -1, t.bmul);        // scores [T][T]
  o.tmp1 = n.new_tensor(t.C, t.H, t.W, -1, t.bmul);  // V^T [C][T]
  o.scale = 1.0f / std::sqrt((float)t.C);  // int(c)**(-0.5), autoencoder_unet.py:203
  n.ops.push_back(o);
  *out = n.op_conv(o.out, cp, 1, 0, x, -1);
  return DDMI_OK;
}

int decoder2d_finalize(Decoder2d& d) {
  Net& n = d.net;
  const ddmi_decoder2d_cfg& c = d.cfg;
  DDMI_REQUIRE(c.n_mult >= 1 && c.n_mult <= 8, "bad ch_mult");
  const int nres = c.n_mult;
  // `curr` is the NOMINAL resolution of a level (what attn_resolutions / hdbf_resolutions are matched against,
  // autoencoder_unet.py:745-747, 790-829); the tensors have the size of the latent that is actually fed (the decoder is
  // fully convolutional: BASELINE config 1 decodes a 16x16 latent with the 256-anchored AFHQ decoder)
  int curr = c.resolution >> (nres - 1);
  DDMI_REQUIRE(c.latent_size >= 1 && c.latent_size <= 4096, "bad latent_size");
  Op in;
  in.kind = OP_NCHW_IN;
  in.ext = 0;
  in.out = n.new_tensor(c.embed_dim, c.latent_size, c.latent_size);
  n.ops.push_back(in);
  int cpq, cin_;
  DDMI_TRY(n.make_conv("post_quant_conv.weight", "post_quant_conv.bias", &cpq));
  DDMI_TRY(n.make_conv("decoder.conv_in.weight", "decoder.conv_in.bias", &cin_));
  int h = n.op_conv(in.out, cpq, 1, 0, -1, -1);
  h = n.op_conv(h, cin_, 1, 0, -1, -1);
  int block_in = c.ch * c.ch_mult[nres - 1];
  DDMI_TRY(build_vae_resblock(n, "decoder.mid.block_1.", h, block_in, &h));
  DDMI_TRY(build_vae_attn(n, "decoder.mid.attn_1.", h, &h));
  DDMI_TRY(build_vae_resblock(n, "decoder.mid.block_2.", h, block_in, &h));
  d.n_planes = 0;
  auto add_export = [&](int tensor) {
    Op e;
    e.kind = OP_EXPORT;
    e.in = tensor;
    e.ext = 1 + d.n_planes;
    n.ops.push_back(e);
    d.export_ops[d.n_planes] = (int)n.ops.size() - 1;
    d.plane_hw[d.n_planes] = n.tensors[tensor].H;
    d.n_planes++;
  };
  for (int lvl = nres - 1; lvl >= 0; --lvl) {
    int block_out = c.ch * c.ch_mult[lvl];
    for (int ib = 0; ib <= c.num_res_blocks; ++ib) {
      std::string pre = "decoder.up." + std::to_string(lvl) + ".block." + std::to_string(ib) + ".";
      DDMI_TRY(build_vae_resblock(n, pre, h, block_out, &h));
      if (in_list(c.attn_resolutions, c.n_attn, curr)) {
        std::string pa = "decoder.up." + std::to_string(lvl) + ".attn." + std::to_string(ib) + ".";
        DDMI_TRY(build_vae_attn(n, pa, h, &h));
      }
    }
    if (in_list(c.hdbf_resolutions, c.n_hdbf, curr)) {
      DDMI_REQUIRE(d.n_planes < 7, "too many HDBF planes");
      std::string ph = "decoder.up." + std::to_string(lvl) + ".hdbf.0.";
      int chd;
      DDMI_TRY(n.make_conv(ph + "weight", ph + "bias", &chd));
      int p = n.op_conv(h, chd, 1, 0, -1, -1);
      add_export(p);
    }
    if (lvl != 0) {
      std::string pu = "decoder.up." + std::to_string(lvl) + ".upsample.conv.";
      int cu;
      DDMI_TRY(n.make_conv(pu + "weight", pu + "bias", &cu));
      h = n.op_conv(h, cu, 1, 1, -1, -1);
      curr *= 2;
    }
  }
  int gno, co;
  DDMI_TRY(n.make_gn("decoder.norm_out.", &gno));
  DDMI_TRY(n.make_conv("decoder.conv_out.weight", "decoder.conv_out.bias", &co));
  int a = n.op_gn(h, gno, 1, 1e-6f);
  int y = n.op_conv(a, co, 1, 0, -1, -1);
  add_export(y);
  return n.plan();
}

int decoder2d_forward(Decoder2d& d, const float* z, void* const* planes, int layout, int B, hipStream_t s) {
  Net& n = d.net;
  if (!n.finalized) return ddmi_fail(DDMI_ERR_STATE, "decoder not finalized");
  DDMI_REQUIRE(z && planes, "decoder2d_forward: null pointer");
  DDMI_REQUIRE(layout == 0 || layout == 1, "decoder2d_forward: layout must be 0 (NCHW fp32) or 1 (NHWC native)");
  prof_begin_forward();
  void* ext[8] = {(void*)z};
  for (int i = 0; i < d.n_planes; ++i) {
    ext[1 + i] = planes[i];
    n.ops[d.export_ops[i]].out_nchw = layout == 0;
  }
  return n.run(B, ext, nullptr, 0, s);
}

// =====================================================================================================
// D2C-VAE tri-plane decoder (occupancy): Autoencoder3D.decode autoencoder_unet.py:907-916 + Decoder_triplane.forward
// :1461-1564.  The three planes share every per-plane weight, and every per-plane op is per-sample, so the planes are
// stacked along the batch axis (3B samples, plane-major: xy | yz | xz in the decoder's call order); the inter-plane
// ResBlock -> Attn -> ResBlock stages run on [B, 3C] after a planes->channels copy and are split back afterwards.
// =====================================================================================================
static int build_inter(Net& n, const std::string& r0, const std::string& at, const std::string& r1, int h3, int* out) {
  const TensorDesc t = n.tensors[h3];
  Op pc;
  pc.kind = OP_PLANES_TO_CHANNELS;
  pc.in = h3;
  pc.out = n.new_tensor(3 * t.C, t.H, t.W, -1, 1);
  n.ops.push_back(pc);
  int h = pc.out;
  DDMI_TRY(build_vae_resblock(n, r0, h, 3 * t.C, &h));
  DDMI_TRY(build_vae_attn(n, at, h, &h));
  DDMI_TRY(build_vae_resblock(n, r1, h, 3 * t.C, &h));
  Op cp;
  cp.kind = OP_CHANNELS_TO_PLANES;
  cp.in = h;
  cp.out = n.new_tensor(t.C, t.H, t.W, -1, 3);
  n.ops.push_back(cp);
  *out = cp.out;
  return DDMI_OK;
}

int decoder3d_finalize(Decoder3d& d) {
  Net& n = d.net;
  const ddmi_decoder2d_cfg& c = d.cfg;
  DDMI_REQUIRE(c.n_mult >= 1 && c.n_mult <= 8, "bad ch_mult");
  const int nres = c.n_mult, e = c.embed_dim;
  int curr = c.resolution >> (nres - 1);
  DDMI_REQUIRE(c.latent_size == curr, "latent_size must equal resolution / 2^(len(ch_mult)-1)");
  Op in;
  in.kind = OP_NCHW_IN;
  in.ext = 0;
  in.out = n.new_tensor(3 * e, curr, curr);
  n.ops.push_back(in);
  // post_quant_conv_{xy,yz,xz}: 1x1 convs on channel blocks (xy | xz | yz) of the latent (autoencoder_unet.py:909-914),
  // realised as 1x1 convs over all 3e channels with zero weights outside the block, each writing one plane slot
  const char* names[3] = {"post_quant_conv_xy.", "post_quant_conv_yz.", "post_quant_conv_xz."};
  const int block_of[3] = {0, 2, 1};  // decoder order (xy, yz, xz) <- latent channel blocks (xy, xz, yz)
  int stacked = n.new_tensor(c.z_channels, curr, curr, -1, 3);
  for (int j = 0; j < 3; ++j) {
    const HostTensor* w = n.find(std::string(names[j]) + "weight");
    const HostTensor* b = n.find(std::string(names[j]) + "bias");
    if (!w || !b) return ddmi_fail(DDMI_ERR_MISSING_WEIGHT, std::string("missing weight '") + names[j] + "*'");
    DDMI_REQUIRE(w->shape.size() == 4 && w->shape[0] == c.z_channels && w->shape[1] == e, "post_quant_conv shape mismatch");
    std::vector<float> full((size_t)c.z_channels * 3 * e, 0.f);
    for (int o = 0; o < c.z_channels; ++o)
      for (int i = 0; i < e; ++i) full[(size_t)o * 3 * e + block_of[j] * e + i] = w->data[(size_t)o * e + i];
    int ci;
    DDMI_TRY(n.make_conv_host(full, c.z_channels, 3 * e, 1, &b->data, &ci));
    int tmp = n.op_conv(in.out, ci, 1, 0, -1, -1);
    (void)tmp;
    Op& o = n.ops.back();  // redirect the output into plane slot j of the stacked tensor
    n.tensors[o.out].first = -1;
    o.out = stacked;
    o.out_boff = j;
  }
  int cin_;
  DDMI_TRY(n.make_conv("decoder.conv_in.weight", "decoder.conv_in.bias", &cin_));
  int h = n.op_conv(stacked, cin_, 1, 0, -1, -1);
  int block_in = c.ch * c.ch_mult[nres - 1];
  DDMI_TRY(build_vae_resblock(n, "decoder.mid.block_1.", h, block_in, &h));
  DDMI_TRY(build_vae_attn(n, "decoder.mid.attn_1.", h, &h));
  DDMI_TRY(build_vae_resblock(n, "decoder.mid.block_2.", h, block_in, &h));
  DDMI_TRY(build_inter(n, "decoder.mid.block_3.", "decoder.mid_attn.", "decoder.mid.block_4.", h, &h));
  d.n_scales = 0;
  auto add_export = [&](int tensor) {
    for (int j = 0; j < 3; ++j) {  // ext slot 1 + plane*3 + scale is fixed up after n_scales is known
      Op ex;
      ex.kind = OP_EXPORT;
      ex.in = tensor;
      ex.in_boff = j;
      n.ops.push_back(ex);
      d.export_ops[j][d.n_scales] = (int)n.ops.size() - 1;
    }
    d.plane_hw[d.n_scales] = n.tensors[tensor].H;
    d.n_scales++;
  };
  for (int lvl = nres - 1; lvl >= 0; --lvl) {
    int block_out = c.ch * c.ch_mult[lvl];
    for (int ib = 0; ib <= c.num_res_blocks; ++ib) {
      std::string pre = "decoder.up." + std::to_string(lvl) + ".block." + std::to_string(ib) + ".";
      DDMI_TRY(build_vae_resblock(n, pre, h, block_out, &h));
      if (in_list(c.attn_resolutions, c.n_attn, curr)) {
        std::string pa = "decoder.up." + std::to_string(lvl) + ".attn." + std::to_string(ib) + ".";
        DDMI_TRY(build_vae_attn(n, pa, h, &h));
      }
    }
    if (in_list(d.inter_attn_resolutions, d.n_inter, curr)) {
      std::string q = "decoder.up." + std::to_string(lvl) + ".inter_attn.";
      DDMI_TRY(build_inter(n, q + "0.", q + "1.", q + "2.", h, &h));
    }
    if (in_list(c.hdbf_resolutions, c.n_hdbf, curr)) {
      DDMI_REQUIRE(d.n_scales < 3, "too many HDBF planes");
      std::string ph = "decoder.up." + std::to_string(lvl) + ".hdbf.0.";
      int chd;
      DDMI_TRY(n.make_conv(ph + "weight", ph + "bias", &chd));
      add_export(n.op_conv(h, chd, 1, 0, -1, -1));
    }
    if (lvl != 0) {
      std::string pu = "decoder.up." + std::to_string(lvl) + ".upsample.conv.";
      int cu;
      DDMI_TRY(n.make_conv(pu + "weight", pu + "bias", &cu));
      h = n.op_conv(h, cu, 1, 1, -1, -1);
      curr *= 2;
    }
  }
  int gno, co;
  DDMI_TRY(n.make_gn("decoder.norm_out.", &gno));
  DDMI_TRY(n.make_conv("decoder.conv_out.weight", "decoder.conv_out.bias", &co));
  int a = n.op_gn(h, gno, 1, 1e-6f);
  DDMI_REQUIRE(d.n_scales < 4, "too many HDBF planes");
  add_export(n.op_conv(a, co, 1, 0, -1, -1));
  for (int j = 0; j < 3; ++j)
    for (int sidx = 0; sidx < d.n_scales; ++sidx) n.ops[d.export_ops[j][sidx]].ext = 1 + j * d.n_scales + sidx;
  return n.plan();
}

int decoder3d_forward(Decoder3d& d, const float* z, void* const* planes, int layout, int B, hipStream_t s) {
  Net& n = d.net;
  if (!n.finalized) return ddmi_fail(DDMI_ERR_STATE, "decoder not finalized");
  DDMI_REQUIRE(z && planes, "decoder3d_forward: null pointer");
  DDMI_REQUIRE(layout == 0 || layout == 1, "decoder3d_forward: layout must be 0 (NCHW fp32) or 1 (NHWC native)");
  prof_begin_forward();
  void* ext[16] = {(void*)z};
  for (int j = 0; j < 3; ++j)
    for (int sidx = 0; sidx < d.n_scales; ++sidx) {
      ext[1 + j * d.n_scales + sidx] = planes[j * d.n_scales + sidx];
      n.ops[d.export_ops[j][sidx]].out_nchw = layout == 0;
    }
  return n.run(B, ext, nullptr, 0, s);
}
