// Graph builders for the video path:
//   * UNetModel_Triplane                      - models/ldm/modules/diffusionmodules/openaimodel.py:796-1221
//   * VITAutoencoder.decode + VideoDecoder_light - models/d2c_vae/autoencoder_vit.py:188-202, autoencoder_unet.py:925-1144
// A sample's latent is a token sequence xy | xt | yt (openaimodel.py:1141-1143).  The xy plane is one NHWC tensor, the
// two t-planes have identical shapes and share every weight, so they are stacked along the batch axis of a second
// tensor (xt samples, then yt samples): every per-plane op is two launches instead of three, and the inter-plane
// attention reads the three row ranges in place through the attention kernel's segment map (k_attention.hip).
#include "models.h"
#include "prof.h"
#include <cmath>

static bool in_list_v(const int* v, int n, int x) {
  for (int i = 0; i < n; ++i)
    if (v[i] == x) return true;
  return false;
}

struct Pair { int xy = -1, tp = -1; };

static int emit_res_pair(Net& n, const ResW& w, Pair x, Pair* out) {
  DDMI_TRY(emit_resblock(n, w, x.xy, &out->xy));
  return emit_resblock(n, w, x.tp, &out->tp);
}
static Pair conv_pair(Net& n, Pair x, int conv, int stride, int ups_xy, int ups_tp) {
  Pair o;
  o.xy = n.op_conv(x.xy, conv, stride, ups_xy, -1, -1);
  o.tp = n.op_conv(x.tp, conv, stride, ups_tp, -1, -1);
  return o;
}
static Pair concat_pair(Net& n, Pair a, Pair b) {
  Pair o;
  const int ids[2][2] = {{a.xy, b.xy}, {a.tp, b.tp}};
  int* outs[2] = {&o.xy, &o.tp};
  for (int i = 0; i < 2; ++i) {
    const TensorDesc ta = n.tensors[ids[i][0]], tb = n.tensors[ids[i][1]];
    Op cat;
    cat.kind = OP_CONCAT;
    cat.in = ids[i][0];
    cat.in2 = ids[i][1];
    cat.bmul = ta.bmul;
    cat.out = n.new_tensor(ta.C + tb.C, ta.H, ta.W, -1, ta.bmul);
    n.ops.push_back(cat);
    *outs[i] = cat.out;
  }
  return o;
}

// =====================================================================================================
// UNetModel_Triplane
// =====================================================================================================
int unet3_finalize(UNet& u) {
  Net& n = u.net;
  const ddmi_unet_cfg& c = u.cfg;
  DDMI_REQUIRE(u.size1 > 0 && u.size2 > 0 && u.size3 > 0, "unet3: size1/size2/size3 must be positive");
  // xt is viewed [size3, size1] and yt [size3, size2] (openaimodel.py:1142-1143); they stack only when size1 == size2,
  // which every reference configuration satisfies (skytimelapse.yaml: 32, 32, 16)
  DDMI_REQUIRE(u.size1 == u.size2, "unet3: size1 must equal size2");
  DDMI_TRY(unet_time_embed_setup(u));
  const int mc = c.model_channels;
  const int L = u.size1 * u.size2 + u.size1 * u.size3 + u.size2 * u.size3;

  Op in;
  in.kind = OP_SEG_IN;
  in.ext = 0;
  in.seq_len = L;
  in.out = n.new_tensor(c.in_channels, u.size1, u.size2, -1, 1);
  in.out2 = n.new_tensor(c.in_channels, u.size3, u.size1, -1, 2);
  n.ops.push_back(in);
  Pair h{in.out, in.out2};

  // make_attn(ch, 'vanilla-1d-multihead') -> MemoryEfficientAttnBlock1D(ch, num_heads=16) (autoencoder_unet.py:585-591)
  auto inter = [&](const std::string& pre, Pair x, Pair* out) -> int {
    AttnW w;
    DDMI_TRY(make_qkv_attn(n, pre, n.tensors[x.xy].C, 16, false, &w));
    return emit_attn_joint(n, w, x.xy, x.tp, &out->xy, &out->tp);
  };
  auto res = [&](const std::string& pre, Pair x, int cout, Pair* out) -> int {
    ResW w;
    DDMI_TRY(make_unet_resblock(u, pre, n.tensors[x.xy].C, cout, &w));
    return emit_res_pair(n, w, x, out);
  };
  auto attn = [&](const std::string& pre, Pair x, Pair* out) -> int {
    AttnW w;
    DDMI_TRY(make_unet_attnblock(u, pre, n.tensors[x.xy].C, &w));
    DDMI_TRY(emit_attn_single(n, w, x.xy, &out->xy));
    return emit_attn_single(n, w, x.tp, &out->tp);
  };

  int ci;
  DDMI_TRY(n.make_conv("input_blocks.0.0.weight", "input_blocks.0.0.bias", &ci));
  h = conv_pair(n, h, ci, 1, 0, 0);  // input_attns[0] is nn.Identity (openaimodel.py:912)
  std::vector<Pair> hs{h};
  int ch = mc, ds = 1, blk = 1;
  for (int level = 0; level < c.n_mult; ++level) {
    for (int r = 0; r < c.num_res_blocks; ++r) {
      std::string pre = "input_blocks." + std::to_string(blk) + ".";
      DDMI_TRY(res(pre + "0.", h, c.channel_mult[level] * mc, &h));
      ch = c.channel_mult[level] * mc;
      if (in_list_v(c.attention_resolutions, c.n_attn, ds)) DDMI_TRY(attn(pre + "1.", h, &h));
      DDMI_TRY(inter("input_attns." + std::to_string(blk) + ".", h, &h));
      hs.push_back(h);
      ++blk;
    }
    if (level != c.n_mult - 1) {
      std::string pre = "input_blocks." + std::to_string(blk) + ".0.";
      int cd;
      DDMI_TRY(n.make_conv(pre + "op.weight", pre + "op.bias", &cd));  // Downsample halves every axis, t included
      h = conv_pair(n, h, cd, 2, 0, 0);
      DDMI_TRY(inter("input_attns." + std::to_string(blk) + ".", h, &h));
      hs.push_back(h);
      ds *= 2;
      ++blk;
    }
  }
  DDMI_TRY(res("middle_block.0.", h, ch, &h));
  DDMI_TRY(attn("middle_block.1.", h, &h));
  DDMI_TRY(res("middle_block.2.", h, ch, &h));
  DDMI_TRY(inter("mid_attn.", h, &h));
  blk = 0;
  for (int level = c.n_mult - 1; level >= 0; --level) {
    for (int i = 0; i <= c.num_res_blocks; ++i) {
      std::string pre = "output_blocks." + std::to_string(blk) + ".";
      Pair skip = hs.back();
      hs.pop_back();
      DDMI_REQUIRE(n.tensors[h.xy].H == n.tensors[skip.xy].H && n.tensors[h.tp].H == n.tensors[skip.tp].H &&
                       n.tensors[h.tp].W == n.tensors[skip.tp].W,
                   "unet3: skip connection shape mismatch (sizes must be divisible by 2^(levels-1))");
      const int cat_c = n.tensors[h.xy].C + n.tensors[skip.xy].C, cout = mc * c.channel_mult[level];
      if (cat_c != cout) {  // th.cat([h, hs.pop()], 1) read in place (dual-source GroupNorm, split skip convolution)
        ResW w;
        DDMI_TRY(make_unet_resblock(u, pre + "0.", cat_c, cout, &w, n.tensors[h.xy].C));
        Pair o;
        DDMI_TRY(emit_resblock_cat(n, w, h.xy, skip.xy, &o.xy));
        DDMI_TRY(emit_resblock_cat(n, w, h.tp, skip.tp, &o.tp));
        h = o;
      } else {
        h = concat_pair(n, h, skip);
        DDMI_TRY(res(pre + "0.", h, cout, &h));
      }
      int sub = 1;
      if (in_list_v(c.attention_resolutions, c.n_attn, ds)) {
        DDMI_TRY(attn(pre + std::to_string(sub) + ".", h, &h));
        ++sub;
      }
      if (level && i == c.num_res_blocks) {
        int cu;
        std::string q = pre + std::to_string(sub) + ".conv.";
        DDMI_TRY(n.make_conv(q + "weight", q + "bias", &cu));
        h = conv_pair(n, h, cu, 1, 1, 1);
        ds /= 2;
      }
      DDMI_TRY(inter("output_attns." + std::to_string(blk) + ".", h, &h));
      ++blk;
    }
  }
  int gno, co;
  DDMI_TRY(n.make_gn("out.0.", &gno));
  DDMI_TRY(n.make_conv("out.2.weight", "out.2.bias", &co));
  Pair a{n.op_gn(h.xy, gno, 1, 1e-5f), n.op_gn(h.tp, gno, 1, 1e-5f)};
  Pair y = conv_pair(n, a, co, 1, 0, 0);
  Op out;
  out.kind = OP_SEG_OUT;
  out.in = y.xy; out.in2 = y.tp;
  out.ext = 1;
  out.seq_len = L;
  n.ops.push_back(out);
  return unet_finish(u);
}

// =====================================================================================================
// VITAutoencoder.decode + VideoDecoder_light
// =====================================================================================================
int decoderv_finalize(DecoderV& d) {
  Net& n = d.net;
  const ddmi_decoderv_cfg& c = d.cfg;
  DDMI_REQUIRE(c.n_mult >= 1 && c.n_mult <= 8, "bad ch_mult");
  DDMI_REQUIRE(c.attn_heads == 1 || c.attn_heads == 16, "attn_heads: 1 ('vanilla') or 16 ('vanilla-multihead')");
  const int nres = c.n_mult;
  int curr = c.resolution >> (nres - 1);      // nominal resolution, only used to match the *_resolutions lists
  const int S = c.resolution / 8, T = c.frames;  // autoencoder_vit.py:189-191 (downsample_factor = 3)
  const int L = S * S + 2 * S * T;
  d.seq_len = L;

  Op in;
  in.kind = OP_SEG_IN;
  in.ext = 0;
  in.seq_len = L;
  in.out = n.new_tensor(c.embed_dim, S, S, -1, 1);
  in.out2 = n.new_tensor(c.embed_dim, T, S, -1, 2);
  n.ops.push_back(in);
  // post_xy / post_xt / post_yt: 1x1 convs (autoencoder_vit.py:142-144, 196-198); xt and yt have their own weights,
  // so each writes one plane slot of the stacked tensor
  int cxy, cxt, cyt;
  DDMI_TRY(n.make_conv("post_xy.weight", "post_xy.bias", &cxy));
  DDMI_TRY(n.make_conv("post_xt.weight", "post_xt.bias", &cxt));
  DDMI_TRY(n.make_conv("post_yt.weight", "post_yt.bias", &cyt));
  DDMI_REQUIRE(n.convs[cxy].Cout == c.z_channels && n.convs[cxy].Cin == c.embed_dim, "post_xy shape mismatch");
  Pair h;
  h.xy = n.op_conv(in.out, cxy, 1, 0, -1, -1);
  h.tp = n.new_tensor(c.z_channels, T, S, -1, 2);
  for (int j = 0; j < 2; ++j) {
    // run on ONE plane of the stacked input (B samples), write plane slot j of the stacked output
    Op o;
    o.kind = OP_CONV;
    o.in = in.out2;
    o.out = h.tp;
    o.conv = j == 0 ? cxt : cyt;
    o.stride = 1; o.pad = 0; o.ups = 0;
    o.bmul = 1;
    o.in_boff = j;
    o.out_boff = j;
    n.ops.push_back(o);
  }
  int cin_;
  DDMI_TRY(n.make_conv("decoder.conv_in.weight", "decoder.conv_in.bias", &cin_));
  h = conv_pair(n, h, cin_, 1, 0, 0);
  int block_in = c.ch * c.ch_mult[nres - 1];

  auto res = [&](const std::string& pre, Pair x, int cout, Pair* out) -> int {
    ResW w;
    DDMI_TRY(make_vae_resblock(n, pre, n.tensors[x.xy].C, cout, &w));
    return emit_res_pair(n, w, x, out);
  };
  auto plane_attn = [&](const std::string& pre, Pair x, Pair* out) -> int {  // MemoryEfficientAttnBlock, ae:273-335
    AttnW w;
    DDMI_TRY(make_qkv_attn(n, pre, n.tensors[x.xy].C, c.attn_heads, false, &w));
    DDMI_TRY(emit_attn_single(n, w, x.xy, &out->xy));
    return emit_attn_single(n, w, x.tp, &out->tp);
  };
  auto inter = [&](const std::string& pre, Pair x, Pair* out) -> int {  // ...1D_expand, 8 heads of dim C, ae:470-534
    AttnW w;
    DDMI_TRY(make_qkv_attn(n, pre, n.tensors[x.xy].C, 8, true, &w));
    return emit_attn_joint(n, w, x.xy, x.tp, &out->xy, &out->tp);
  };
  DDMI_TRY(res("decoder.mid.block_1.", h, block_in, &h));
  DDMI_TRY(plane_attn("decoder.mid.attn_1.", h, &h));
  DDMI_TRY(res("decoder.mid.block_2.", h, block_in, &h));
  DDMI_TRY(inter("decoder.mid_attn.", h, &h));

  d.n_scales = 0;
  auto add_export = [&](Pair p) {
    // reference order of the returned lists: (xy, yt, xt); stacked slot 0 = xt, 1 = yt
    const int tensor[3] = {p.xy, p.tp, p.tp};
    const int boff[3] = {0, 1, 0};
    for (int j = 0; j < 3; ++j) {
      Op ex;
      ex.kind = OP_EXPORT;
      ex.in = tensor[j];
      ex.in_boff = boff[j];
      n.ops.push_back(ex);
      d.export_ops[j][d.n_scales] = (int)n.ops.size() - 1;
    }
    d.plane_h[0][d.n_scales] = n.tensors[p.xy].H; d.plane_w[0][d.n_scales] = n.tensors[p.xy].W;
    d.plane_h[1][d.n_scales] = n.tensors[p.tp].H; d.plane_w[1][d.n_scales] = n.tensors[p.tp].W;
    d.n_scales++;
  };
  for (int lvl = nres - 1; lvl >= 0; --lvl) {
    int block_out = c.ch * c.ch_mult[lvl];
    for (int ib = 0; ib <= c.num_res_blocks; ++ib) {
      std::string pre = "decoder.up." + std::to_string(lvl) + ".block." + std::to_string(ib) + ".";
      DDMI_TRY(res(pre, h, block_out, &h));
      if (in_list_v(c.attn_resolutions, c.n_attn, curr))
        DDMI_TRY(plane_attn("decoder.up." + std::to_string(lvl) + ".attn." + std::to_string(ib) + ".", h, &h));
    }
    if (in_list_v(c.inter_attn_resolutions, c.n_inter, curr))
      DDMI_TRY(inter("decoder.up." + std::to_string(lvl) + ".inter_attn.0.", h, &h));
    if (in_list_v(c.hdbf_resolutions, c.n_hdbf, curr)) {
      DDMI_REQUIRE(d.n_scales < 3, "too many HDBF scales");
      std::string ph = "decoder.up." + std::to_string(lvl) + ".hdbf.0.";
      int chd;
      DDMI_TRY(n.make_conv(ph + "weight", ph + "bias", &chd));
      add_export(conv_pair(n, h, chd, 1, 0, 0));
    }
    if (lvl != 0) {
      std::string pu = "decoder.up." + std::to_string(lvl) + ".upsample.conv.";
      int cu;
      DDMI_TRY(n.make_conv(pu + "weight", pu + "bias", &cu));
      h = conv_pair(n, h, cu, 1, 1, 2);  // xy x2 on both axes, t-planes along W only (ae:1115-1118)
      curr *= 2;
    }
  }
  int gno, co;
  DDMI_TRY(n.make_gn("decoder.norm_out.", &gno));
  DDMI_TRY(n.make_conv("decoder.conv_out.weight", "decoder.conv_out.bias", &co));
  Pair a{n.op_gn(h.xy, gno, 1, 1e-6f), n.op_gn(h.tp, gno, 1, 1e-6f)};
  DDMI_REQUIRE(d.n_scales < 4, "too many HDBF scales");
  add_export(conv_pair(n, a, co, 1, 0, 0));
  d.out_ch = n.convs[co].Cout;
  for (int j = 0; j < 3; ++j)
    for (int sidx = 0; sidx < d.n_scales; ++sidx) n.ops[d.export_ops[j][sidx]].ext = 1 + j * d.n_scales + sidx;
  return n.plan();
}

int decoderv_forward(DecoderV& d, const float* z, void* const* planes, int layout, int B, hipStream_t s) {
  Net& n = d.net;
  if (!n.finalized) return ddmi_fail(DDMI_ERR_STATE, "decoder not finalized");
  DDMI_REQUIRE(z && planes, "decoderv_forward: null pointer");
  DDMI_REQUIRE(layout == 0 || layout == 1, "decoderv_forward: layout must be 0 (NCHW fp32) or 1 (NHWC native)");
  prof_begin_forward();
  void* ext[16] = {(void*)z};
  for (int j = 0; j < 3; ++j)
    for (int sidx = 0; sidx < d.n_scales; ++sidx) {
      ext[1 + j * d.n_scales + sidx] = planes[j * d.n_scales + sidx];
      n.ops[d.export_ops[j][sidx]].out_nchw = layout == 0;
    }
  return n.run(B, ext, nullptr, 0, s);
}
