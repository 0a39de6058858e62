// Static-shape network executor shared by the UNet and the D2C-VAE decoder.
//
// A network is built once (at *_finalize) into a flat list of ops over symbolic tensors; tensors get offsets in one
// activation arena by a liveness-based first-fit allocator (so the working set stays small enough to live in the
// 256 MiB Infinity Cache at moderate batch), weights are packed once into device memory, and `run()` is a plain
// sequence of asynchronous kernel launches on the caller's stream: no allocation, no host sync, hipGraph-capturable.
#pragma once
#include <map>
#include <string>
#include <vector>
#include "kernels.h"

// diagnostics: producer-side GroupNorm for networks planned and run afterwards: 0 off, 1 on, -1 = the DDMI_GN_FUSE environment (net.hip)
void net_gn_fuse_override(int mode);
int net_gn_fuse_mode();   // the effective mode: 0 never, 1 where the launch runs on the halo kernel anyway, 2 wherever the shape allows

// The small-batch regime (round 6; SURVEY 8(d) config 2 at B in {1, 8, 16} - the reference's own gen mode runs test_batch_size 6-10): a UNet
// evaluation there is ~300 dependent launches of 5-20 us, not a bandwidth or MFMA problem.  With it ON (default; DDMI_SMALL_BATCH=0 /
// ddmi_debug_small_batch(0) switch it off - the parity suite does, so that its small batches keep exercising the kernels the benchmark's
// B = 512 runs), a run at B <= net_small_batch_max() takes GroupNorm as ONE launch where a (sample, group) slice fits a workgroup's registers (launch_gn_small),
// and networks BUILT for max_batch <= that bound keep the five-launch AttentionBlock and the two
// accumulating 1x1 launches of a concatenated skip connection (the fused / dual-source forms need hundreds of tiles to pay).
void net_small_batch_override(int mode);   // -1: environment, 0 off, 1 on
bool net_small_batch_enabled();
int net_small_batch_max();                 // DDMI_SMALL_BATCH_MAX, default 16
// diagnostics: one-launch GroupNorms of the last Net::run of this process
extern int g_net_small_gn_launches;

struct HostTensor {
  std::vector<float> data;
  std::vector<int64_t> shape;
  int64_t numel() const {
    int64_t n = 1;
    for (auto s : shape) n *= s;
    return n;
  }
};

struct ConvW {
  void* w = nullptr;      // packed [Cout][taps*Cin], network dtype
  float* bias = nullptr;  // [Cout] fp32 (may be null)
  int Cout = 0, Cin = 0, K = 1;
  void* w_x3 = nullptr;   // bf16x3 numerics: [Cout][tap][wh | wh | wl] bf16 (launch_pack_conv_weight_x3); w stays null then
  void* w_up4 = nullptr;  // bf16 parity form for a fused x2 upsample (launch_pack_conv_up4), made by plan() when an op
  bool want_up4 = false;  // asked for it (op_conv only marks: it returns a tensor id, not a status)
  // bf16 DENSE form of a 3x3 / pad 1 convolution on a 2 x 2 image (the lowest level of the occupancy UNet): every output pixel sees
  // every input pixel through exactly one tap, so the convolution IS the 1x1 convolution [B][4 Cin] -> [B][4 Cout] with
  // W4[(po, co)][(pi, ci)] = W[co][ci][yi - yo + 1][xi - xo + 1] over the same NHWC memory - 32 B Cin Cout FLOPs instead of the 72 the
  // 3x3 form spends (5 of every 9 taps multiply padding).  Marked by op_conv, packed by plan() (pack_px4_weights)
  void* w_px4 = nullptr;
  float* bias_px4 = nullptr;   // the bias repeated for the four pixels
  bool want_px4 = false;
  std::string wkey;       // state-dict key of the raw fp32 weight (make_conv), valid until finalize drops the host copies
};

struct AttnBlockW {   // an AttentionBlock's qkv projection as k_attnblock.hip's per-(head, chunk) LDS images
  void* w = nullptr;
  float* bias = nullptr;
  int C = 0, heads = 0;
};

struct GNW {
  float* gamma = nullptr;
  float* beta = nullptr;
  int C = 0;
};

enum OpKind { OP_NCHW_IN = 0, OP_GN, OP_CONV, OP_ATTN_LEGACY, OP_ATTN_FULL, OP_CONCAT, OP_EXPORT, OP_PLANES_TO_CHANNELS,
              OP_CHANNELS_TO_PLANES,
              // video path: a sample's tokens are xy | xt | yt; the xy plane is one tensor (`out` / `in`), the two
              // t-planes are stacked along the batch axis of a second one (`out2` / `in2`: xt samples, then yt samples)
              OP_SEG_IN,      // ext [B, C, L] fp32 -> out (xy), out2 (xt, yt)
              OP_SEG_OUT,     // in, in2 -> ext [B, C, L] fp32
              OP_GN_JOINT,    // GroupNorm whose statistics run over all tokens of the sample: in, in2 -> out, out2
              OP_ATTN_MH,     // multi-head attention over in (+ in2 when >= 0) -> out (+ out2)
              OP_ATTN_BLOCK };  // qkv projection + attention of an AttentionBlock in one launch (k_attnblock.hip): in = the GroupNorm
                                // output, out = the attention output; `conv` indexes Net::attnblocks

struct Op {
  int kind = 0;
  int in = -1, in2 = -1, out = -1, out2 = -1, res = -1;
  int conv = -1, gn = -1;
  int silu = 0;
  int frag_out = 0;    // OP_GN: output in MFMA B-fragment order (its only consumer is an OP_ATTN_BLOCK)
  float eps = 1e-5f;
  int stride = 1, pad = 0, ups = 0;
  int emb_off = -1;
  int px4 = 0;         // OP_CONV: runs as the dense 1x1 form of a 3x3 convolution on a 2 x 2 image (ConvW::w_px4); emb_off then points at
                       // embedding rows repeated for the four pixels
  int heads = 0;
  int head_dim = 0;    // OP_ATTN_MH
  int q_prescaled = 0;    // OP_ATTN_MH: q already carries softmax scale x log2 e (AttnW::q_prescaled)
  int legacy_layout = 0;  // OP_ATTN_MH: per-head [q|k|v] channel interleave (QKVAttentionLegacy) instead of [q heads|k heads|v heads]
  int seq_len = 0;     // OP_SEG_IN / OP_SEG_OUT: L of the external [B, C, L] tensor
  int out_nchw = 0;   // OP_CONV: write fp32 NCHW to external output slot `ext`
  int ext = -1;       // external pointer slot (inputs / outputs of the network)
  int gn_slot = -1;   // index into the GN statistics arena
  int tmp0 = -1, tmp1 = -1, tmp2 = -1;  // scratch tensors (OP_ATTN_FULL)
  // GroupNorm statistics out of the producer's epilogue (plan()): OP_CONV writes per-channel sums of its output into tensor
  // `stat_out`; OP_GN reads the ones of its input(s) instead of running the statistics pass when every producer delivered
  int stat_out = -1, stat_in = -1, stat_in2 = -1;
  // producer-side GroupNorm (k_conv.hip conv_epilogue_fgn; plan(): attach_fused_groupnorm): OP_CONV whose output feeds ONE single-source
  // OP_GN and nothing else carries that op's index in `fgn_gn`; when the launch qualifies (conv_fgn_ok, decided per run) it writes the
  // normalised tensor - the GN op's `out` - itself and the GN op (`fgn_conv` = the producer's index) is skipped
  int fgn_gn = -1, fgn_conv = -1;
  // OP_CONV with the ResBlock's 1x1 skip_connection folded into its K loop (bf16, halo-kernel shapes): the block input(s)
  // `skip_in` (| `skip_in2`: the second half of a channel-concatenated input), the un-split 1x1 weights `skip_conv`, and the
  // summed bias of both convolutions
  int skip_in = -1, skip_in2 = -1, skip_conv = -1;
  const float* bias_fused = nullptr;
  float scale = 1.f;
  ConvPlan tuned;     // OP_CONV: autotuned launch plan for B == max_batch (cfg 0 = heuristic)
  // tri-plane networks stack the three planes (shared weights) along the batch axis: an op runs on bmul * B samples;
  // out_boff / in_boff select a plane (in units of B samples) inside a stacked tensor
  int bmul = 1, in_boff = 0, out_boff = 0;
};

struct TensorDesc {
  int C = 0, H = 0, W = 0;
  int64_t elems = 0;    // per sample
  int first = -1, last = -1;
  size_t off = 0;       // byte offset in the arena (for max batch)
  int dt_override = -1; // -1: network dtype
  int bmul = 1;         // samples held = bmul * B
  bool x3img = false;   // bf16x3 numerics: held as the [xh | xl | xh] bf16 image (6 bytes per element) its consumer convolutions read
};

struct Net {
  int dt = DT_F32;
  bool x3 = false;                          // DDMI_PARITY_BF16X3: fp32 tensors, every convolution as three bf16 MFMA products (k_elementwise.hip)
  void* x3_in = nullptr;                    // [pixels][xh | xl | xh] bf16 image of the current convolution's input
  size_t x3_in_bytes = 0;
  int max_batch = 1;
  bool finalized = false;
  bool autotune = true;                     // bf16 only; DDMI_AUTOTUNE=0 disables
  double tune_seconds = 0;
  std::map<std::string, HostTensor> host;   // raw state-dict tensors (fp32) until finalize
  std::vector<ConvW> convs;
  std::vector<GNW> gns;
  std::vector<AttnBlockW> attnblocks;
  std::vector<Op> ops;
  std::vector<TensorDesc> tensors;
  std::vector<void*> dev_allocs;            // everything hipMalloc'ed by this net
  unsigned char* arena = nullptr;
  size_t arena_bytes = 0;
  double* gn_stats = nullptr;               // GN_STATS_DOUBLES(max_batch) scratch shared by all GN ops
  int n_gn_slots = 0;
  float* splitk_ws = nullptr;
  size_t splitk_bytes = 0;
  size_t weight_bytes = 0;

  ~Net();
  int add_host(const char* key, const float* data, const int64_t* shape, int ndim);
  const HostTensor* find(const std::string& key) const;

  // ---- build helpers -------------------------------------------------------------------------------
  int new_tensor(int C, int H, int W, int dt_override = -1, int bmul = 1);
  int max_bmul = 1;
  int dev_upload(const float* src, int64_t n, float** out);
  int make_conv(const std::string& wkey, const std::string& bkey, int* idx);   // [Cout,Cin,K,K] or [Cout,Cin,1] / [Cout,Cin]
  int make_gn(const std::string& prefix, int* idx);
  // w [3C][C], b [3C]: QKVAttentionLegacy order, q rows pre-scaled (make_unet_attnblock)
  int make_attnblock(const std::vector<float>& w, const std::vector<float>& b, int C, int heads, int* idx);
  int op_gn(int in, int gn, int silu, float eps, int in2 = -1);  // in2 >= 0: GroupNorm of the channel concat [in | in2]
  // emb_px4: the rows at emb_off are repeated four times (make_unet_resblock emb_rep = 4), as the dense 2 x 2 form needs them
  int op_conv(int in, int conv, int stride, int ups, int res, int emb_off, float scale = 1.f, bool emb_px4 = false);
  // 1x1 convolution over the channel concatenation [in | in2] read in place (k_conv.hip CAT instantiation of the persistent 1x1
  // kernel); cat_conv_ok: this network / shape can run it (bf16, whole 224- / 128-wide channel tiles ...), else the caller emits the
  // two accumulating launches.  DDMI_CONV_CAT=0 keeps those everywhere (A/B)
  bool cat_conv_ok(int in, int in2, int conv) const;
  int op_conv_cat(int in, int in2, int conv);
  bool px4_ok(int in) const;   // a 3x3 / stride 1 convolution on tensor `in` may take the dense 2 x 2 form (bf16, 2 x 2 image, one plane)
  // conv3x3(in) + conv1x1([skip_in | skip_in2]) in one launch; false when this network / shape cannot fold (caller emits the two ops)
  bool can_fold_skip(int in, int conv) const;
  int op_conv_skip(int in, int conv, int skip_in, int skip_in2, int skip_conv, int* out);
  int make_conv_host(const std::vector<float>& w, int Cout, int Cin, int K, const std::vector<float>* bias, int* idx);
  int pack_up4_weights();  // parity-form weights of the marked upsample convolutions (called by plan)
  int pack_px4_weights();  // dense 2 x 2 form of the marked convolutions (called by plan)
  void mark_x3_images();         // bf16x3: GroupNorm outputs consumed only by x3 convolutions are produced as x3 images (called by plan)
  void attach_channel_stats();  // pairs GroupNorm inputs with the convolutions that produce them (called by plan)
  void attach_fused_groupnorm();  // marks (convolution, GroupNorm) pairs for the producer-side GroupNorm epilogue (called by plan)
  float* fgn_part = nullptr;      // granule scratch of the producer-side GroupNorm (CONV_FGN_PART_FLOATS), shared by all such launches
  unsigned* fgn_sync = nullptr;   // {launch epoch, error flag, -, -}
  int fgn_launches = 0;           // fused launches of the last run() (diagnostics)
  unsigned* fgn_flag_host = nullptr;   // host-mapped copy of fgn_sync[1]: a timed-out hand-off sets both, the host peeks this one for free
  bool fgn_enabled = true;        // per network: async_status() clears it for good when a hand-off of THIS network timed out
  // the cross-workgroup hand-off (shapes with more than one 256-row tile per sample) only for networks whose owner looks at the flag - the
  // UNet handles (unet_forward peeks, ddmi_unet_async_status checks) - and only on the device its progress argument was made for (plan());
  // every other network fuses the 16 x 16 shapes only, whose statistics are complete inside one tile (round-5 advisor)
  bool fgn_handoff = false;
  int fgn_timeouts = 0;           // hand-off timeouts seen by async_status() (each one switched the fusion off)
  bool fgn_fusing() const { return fgn_part && fgn_enabled && net_gn_fuse_mode() != 0; }
  // a hand-off has flagged a timeout since the last async_status(): a volatile read of host memory, no synchronisation
  bool async_error_pending() const { return fgn_flag_host && *(volatile unsigned*)fgn_flag_host != 0u; }
  // 0 = fine; DDMI_ERR_STATE when a fused GroupNorm launch timed out waiting for a partner tile: the results since the last call are
  // invalid and the fusion is OFF for this network from here on (the caller runs again).  SYNCHRONISES: the whole device, or - when the
  // caller names the stream its forwards ran on - that stream only (no device-wide barrier for other streams' work)
  int async_status(const hipStream_t* stream = nullptr);
  void debug_inject_fgn_timeout();   // tests: what a timed-out workgroup does to the two flags
  int plan();   // liveness + arena + workspaces
  void* tptr(int id) const { return arena + tensors[id].off; }
  size_t tbytes(int id, int B) const;

  // ---- execution -----------------------------------------------------------------------------------
  // ext: external pointers indexed by Op::ext; emb: [B][emb_stride] fp32 or null
  int run(int B, void* const* ext, const float* emb, int emb_stride, hipStream_t s);
};
