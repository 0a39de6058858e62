/* ddmi_hip.h - C ABI of libddmi_hip.so: the MI355X (gfx950) implementation of the DDMI generation path.
 *
 * The reference (mlvlab/DDMI) has no FFI around its sampling path: the seams are Python call signatures
 * (SURVEY.md 8b).  Each entry point below names the reference function it replaces; INTEGRATION.md shows the
 * ctypes binding a maintainer would add on the reference side.
 *
 * Conventions
 *   - every function returns 0 (DDMI_OK) on success, a non-zero DDMI_ERR_* code otherwise; the message for the
 *     calling thread is available from ddmi_last_error().  No exception crosses this boundary.
 *   - all tensor pointers are DEVICE pointers unless the parameter says "host"; the caller owns every input and
 *     output buffer.  The library owns only packed weights and a per-handle activation workspace sized at
 *     *_finalize for `max_batch`.
 *   - `stream` is a hipStream_t passed as void* (e.g. torch.cuda.current_stream().cuda_stream); all work is
 *     enqueued asynchronously on it: no allocation, no host synchronisation inside forward/step calls, so the
 *     calls are hipGraph-capturable.
 *   - handles are not thread-safe; distinct handles may be used concurrently on distinct streams.
 *   - model-boundary tensors keep the reference layout: NCHW fp32.  Internally activations are NHWC in the
 *     handle's numerics mode (fp32 operands on the exact fp32 MFMA, or bf16 operands with fp32 accumulation).
 */
#ifndef DDMI_HIP_H
#define DDMI_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { DDMI_OK = 0, DDMI_ERR_INVALID = 1, DDMI_ERR_HIP = 2, DDMI_ERR_STATE = 3, DDMI_ERR_MISSING_WEIGHT = 4 };

/* numerics modes: PARITY = fp32 storage + v_mfma_f32_16x16x4_f32 (exact fp32 FMA chains), used to match the
 * reference's CPU fp32 path to 1e-3; PERF = bf16 storage/operands + fp32 accumulate (the reference's GPU path
 * is fp16 autocast, tools/ldm/image.py:32-36,239). */
enum { DDMI_PARITY_FP32 = 0, DDMI_PERF_BF16 = 1,
       /* fp32 tensors; every convolution of the UNets / decoders computed as THREE bf16 MFMA products per fp32 product
        * (x = xh + xl, w = wh + wl: xh.wh + xl.wh + xh.wl, fp32 accumulation), everything else as DDMI_PARITY_FP32: the
        * north star's 1e-3 agreement at about 3/16 instead of 16/16 of the fp32-MFMA cost per product */
       DDMI_PARITY_BF16X3 = 2 };

/* plane / tensor layouts at the decoder -> MLP seam */
enum { DDMI_LAYOUT_NCHW_F32 = 0, DDMI_LAYOUT_NHWC_NATIVE = 1 };

int ddmi_init(int device);                 /* selects the device, checks it is gfx950 */
const char* ddmi_last_error(void);         /* thread-local message of the last failing call */
const char* ddmi_version(void);

/* Live per-kernel timing with HIP events recorded on the launch stream around each kernel (bench.py's roofline
 * figure).  enable(n): time every kernel of every n-th network forward (0 = off, the default); collect():
 * synchronises and writes a JSON object {kernel: {launches, total_ms, flops, bytes}} (algorithmic flops/bytes). */
int ddmi_profile_enable(int every_n_forwards);
int ddmi_profile_collect(char* json_out, uint64_t capacity);

/* ---------------------------------------------------------------------------------------------------------
 * Denoiser: replaces UNetModel.forward(x, timesteps)   models/ldm/modules/diffusionmodules/openaimodel.py:761-793
 * (constructor topology :493-743; the DDPM wrapper calls it as self.model(x, t, cond), diffusion/ddpm.py:258,305)
 * --------------------------------------------------------------------------------------------------------- */
typedef struct ddmi_unet ddmi_unet;
typedef struct {
  int image_size;               /* latent H = W                       (unetconfig.image_size) */
  int in_channels, model_channels, out_channels, num_res_blocks;
  int n_mult;  int channel_mult[8];
  int n_attn;  int attention_resolutions[8];   /* downsample rates with attention */
  int num_head_channels;        /* must be 32 */
  int max_batch;
  int numerics;                 /* DDMI_PARITY_FP32 | DDMI_PERF_BF16 | DDMI_PARITY_BF16X3 */
} ddmi_unet_cfg;

int ddmi_unet_create(const ddmi_unet_cfg* cfg, ddmi_unet** out);
/* hands over one state-dict tensor (reference key names, e.g. "input_blocks.4.0.in_layers.2.weight");
 * `data` is a HOST pointer to contiguous fp32 */
int ddmi_unet_load_weight(ddmi_unet* h, const char* key, const float* data, const int64_t* shape, int ndim);
int ddmi_unet_finalize(ddmi_unet* h);      /* packs weights for MFMA, plans the workspace; frees host copies */
/* x [B,Cin,S,S] fp32, t [B] int64 (device), eps_out [B,Cout,S,S] fp32 */
int ddmi_unet_forward(ddmi_unet* h, const float* x, const int64_t* t, float* eps_out, int B, void* stream);
/* promise that every element of `t` equals t[0] in the following forwards (true for every sampling loop of
 * diffusion/ddpm.py: :294, :349): the time-embedding path then runs for one row instead of B */
int ddmi_unet_set_uniform_t(ddmi_unet* h, int enabled);
/* replay the forward as a hipGraph: the first call with a given (B, x, t, eps_out) runs and captures the launches, later
 * calls with the same buffers are one hipGraphLaunch.  Worth it when the forward is launch-bound (batch <~ 16: ~900
 * kernels of a few microseconds); at large batch it changes nothing.  The stream must not be the legacy null stream. */
int ddmi_unet_set_graph(ddmi_unet* h, int enabled);
/* Deferred errors of the asynchronous forwards (SYNCHRONISES the device): the producer-side GroupNorm of the bf16 networks (k_conv.hip
 * conv_epilogue_fgn: the ResBlock's out_layers norm leaves conv1's epilogue normalised; tiles of a sample exchange their partial
 * statistics in memory) bounds its wait for a partner tile (50 ms) and flags a timeout instead of hanging.  DDMI_ERR_STATE then: the
 * forwards since the last call are invalid, the fusion is switched OFF for this handle for good (captured graphs are dropped) and
 * the caller runs them again - ddmi_amd/ddpm.py's samplers do, once.  A caller that never calls this still hears of a timeout: the
 * first ddmi_unet_forward AFTER the flag went up (a host-mapped word, looked at without synchronising) returns the same error instead
 * of running.  Call it wherever results are copied to the host.  *fused_gn_launches (may be null): how many convolutions of the last
 * forward ran the fused epilogue.  (DDMI_GN_FUSE=0 / ddmi_debug_gn_fuse(0) keep the separate pass everywhere.) */
int ddmi_unet_async_status(ddmi_unet* h, int* fused_gn_launches);
/* the same, synchronising only `stream` (the one the forwards were issued on) instead of the device: other streams' work is not waited for */
int ddmi_unet_async_status_on(ddmi_unet* h, void* stream, int* fused_gn_launches);
/* tests: raises the hand-off's timeout flag of this handle as a timed-out workgroup would (device word + host-mapped word) */
int ddmi_unet_debug_inject_gn_timeout(ddmi_unet* h);
int ddmi_unet_destroy(ddmi_unet* h);
/* bytes held by the handle: [0] packed weights, [1] activation arena, [2] split-K slabs */
int ddmi_unet_memory(ddmi_unet* h, uint64_t out[3]);

/* ---------------------------------------------------------------------------------------------------------
 * Sampler step: replaces p_mean_variance + q_posterior + p_sample (diffusion/ddpm.py:257-281), and
 * model_predictions + the DDIM update (ddpm.py:303-332, 353-368), given the denoiser output.
 * --------------------------------------------------------------------------------------------------------- */
typedef struct {
  float sqrt_one_minus_ac;   /* sqrt_one_minus_alphas_cumprod[t]      ddpm.py:476-477 */
  float sqrt_recip_ac;       /* sqrt_recip_alphas_cumprod[t]          ddpm.py:238 */
  float sqrt_recipm1_ac;     /* sqrt_recipm1_alphas_cumprod[t]        ddpm.py:239 */
  float coef1, coef2;        /* posterior_mean_coef1/2[t]             ddpm.py:250-251   (ANCESTRAL) */
  float noise_scale;         /* [t!=0] * exp(0.5*posterior_log_variance_clipped[t])  ddpm.py:280-281 */
  float sqrt_ac_next;        /* sqrt(alphas_cumprod[t_next])          ddpm.py:366       (DDIM) */
  float c_dir;               /* sqrt(1 - alpha_next - sigma^2)        ddpm.py:361 */
  float sigma;               /* eta * sqrt(...)                       ddpm.py:360 */
} ddmi_step_coefs;
enum { DDMI_STEP_ANCESTRAL = 0, DDMI_STEP_DDIM = 1, DDMI_STEP_DDIM_LAST = 2 };
/* x, model_out, noise (nullable -> 0), x_out, x0_out (nullable): [B,C,HW] fp32 NCHW; mixing_logit [C] or null */
int ddmi_ddpm_step(const float* x, const float* model_out, const float* noise, const float* mixing_logit,
                   const ddmi_step_coefs* coefs, int mode, int clip_denoised, float* x_out, float* x0_out, int B,
                   int C, int HW, void* stream);

/* Same step with the noise drawn INSIDE the kernel: the reference draws `noise_like(x.shape)` / `randn_like(img)` from
 * the global torch generator (ddpm.py:278, 363); here every chain has its own counter-based stream
 * normal(seeds[b], tag, element) (Philox4x32-10 + Box-Muller), so a sample's noise does not depend on the batch or GPU
 * it runs in, and no [B,C,HW] noise tensor is written or read.  `tag`: the step index.  C*HW % 4 == 0. */
int ddmi_ddpm_step_rng(const float* x, const float* model_out, const uint64_t* seeds, uint32_t tag,
                       const float* mixing_logit, const ddmi_step_coefs* coefs, int mode, int clip_denoised, float* x_out,
                       float* x0_out, int B, int C, int HW, void* stream);
/* out[b][e] = normal(seeds[b], tag, e), e < n_per_sample (multiple of 4): the very values ddmi_ddpm_step_rng draws for
 * that tag - x_T of a chain (tag 0xFFFFFFFF by convention), or a replay of a step's noise for parity checks */
int ddmi_randn_fill(float* out, const uint64_t* seeds, int B, int64_t n_per_sample, uint32_t tag, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * D2C-VAE image decoder: replaces Autoencoder.decode (models/d2c_vae/autoencoder_unet.py:855-858) =
 * post_quant_conv + Decoder.forward (:790-829).  Keys: "post_quant_conv.*", "decoder.*".
 * --------------------------------------------------------------------------------------------------------- */
typedef struct ddmi_decoder2d ddmi_decoder2d;
typedef struct {
  int embed_dim;               /* latent channels (model.embed_dim) */
  int z_channels, resolution, out_ch, ch, num_res_blocks;
  int n_mult;  int ch_mult[8];
  int n_attn;  int attn_resolutions[8];
  int n_hdbf;  int hdbf_resolutions[8];
  int latent_size;             /* = resolution / 2^(n_mult-1) */
  int max_batch;
  int numerics;
} ddmi_decoder2d_cfg;
int ddmi_decoder2d_create(const ddmi_decoder2d_cfg* cfg, ddmi_decoder2d** out);
int ddmi_decoder2d_load_weight(ddmi_decoder2d* h, const char* key, const float* data, const int64_t* shape, int ndim);
int ddmi_decoder2d_finalize(ddmi_decoder2d* h);
int ddmi_decoder2d_num_planes(ddmi_decoder2d* h, int* n, int plane_hw[8]);
/* z [B,embed_dim,S,S] fp32 NCHW; planes[i]: coarse -> fine, each [B,out_ch,P_i,P_i] in `layout`
 * (DDMI_LAYOUT_NCHW_F32, or DDMI_LAYOUT_NHWC_NATIVE = channels-last in the handle's storage dtype, the layout the
 * fused plane-sampling kernel wants) */
int ddmi_decoder2d_forward(ddmi_decoder2d* h, const float* z, void* const* planes, int layout, int B, void* stream);
int ddmi_decoder2d_destroy(ddmi_decoder2d* h);

/* ---------------------------------------------------------------------------------------------------------
 * Fused HDPE plane sampling + coordinate MLP (image): replaces MLP.forward (models/d2c_vae/mlp.py:34-66):
 * 3 x F.grid_sample(bilinear, border, align_corners=False) (utils/general_utils.py:122-123) + 4 StyledResBlocks
 * + ToRGB (models/d2c_vae/blocks.py:604-638, 390-412) + FusedLeakyReLU (op/fused_bias_act_kernel.cu:19-49).
 * Keys: the MLP state dict ("time_mlp.1.weight", "net_res1.conv1.conv.weight", ...).
 * --------------------------------------------------------------------------------------------------------- */
typedef struct ddmi_mlp2d ddmi_mlp2d;
typedef struct {
  int in_ch, out_ch, ch, latent_dim;   /* mlpconfig; supported: in_ch 2, out_ch 3, ch 256, latent_dim 64 */
  int numerics;
} ddmi_mlp2d_cfg;
int ddmi_mlp2d_create(const ddmi_mlp2d_cfg* cfg, ddmi_mlp2d** out);
int ddmi_mlp2d_load_weight(ddmi_mlp2d* h, const char* key, const float* data, const int64_t* shape, int ndim);
int ddmi_mlp2d_finalize(ddmi_mlp2d* h);
/* planes[3] coarse->fine in `layout` (see above), plane_hw[3]; coords [N,2] fp32 (x,y in [-1,1]) - the grid
 * convert_to_coord_format_2d builds, flattened row-major; si = scale injection (get_scale_injection);
 * noise: optional [12,B,N] fp32 NoiseInjection draws (blocks.py:286-297), null = none; rgb_out [B,3,N] fp32 */
int ddmi_hdpe_mlp2d(ddmi_mlp2d* h, const void* const* planes, const int* plane_hw, int layout, const float* coords,
                    float si, const float* noise, float* rgb_out, int B, int N, void* stream);
/* The same with the NoiseInjection draws made inside the kernel: `torch.randn`-free replacement of
 * `image.new_empty(batch,1,h,w).normal_()` (blocks.py:293-295).  seeds [B] uint64 on the device (one per sample, the
 * per-chain seeds of ddmi_ddpm_step_rng); draw c (0..11, call order) of query point p of sample b is
 * normal(seeds[b], 0xFFFE0000 + c, p) - what ddmi_randn_fill(out, seeds, B, N, 0xFFFE0000 + c) writes - so a sample's
 * image does not depend on its batch.  A handle whose `noise.weight` tensors are all zero ignores the seeds; a handle
 * with non-zero weights REFUSES ddmi_hdpe_mlp2d(noise = null): the term is never dropped silently. */
int ddmi_hdpe_mlp2d_rng(ddmi_mlp2d* h, const void* const* planes, const int* plane_hw, int layout, const float* coords,
                        float si, const uint64_t* seeds, float* rgb_out, int B, int N, void* stream);
/* *out = 1 when any of the 12 `noise.weight` scalars is non-zero (after finalize) */
int ddmi_mlp2d_noise_nonzero(ddmi_mlp2d* h, int* out);
int ddmi_mlp2d_destroy(ddmi_mlp2d* h);

/* ---------------------------------------------------------------------------------------------------------
 * Occupancy (ShapeNet) path.
 * Tri-plane decoder: replaces Autoencoder3D.decode (models/d2c_vae/autoencoder_unet.py:907-916) = three per-plane
 * post_quant convs + Decoder_triplane.forward (:1461-1564).  Keys: "post_quant_conv_{xy,yz,xz}.*", "decoder.*".
 * z [B, 3*embed_dim, S, S] fp32 NCHW with channel blocks (xy | xz | yz); planes[plane*n_scales + scale], plane order
 * (xy, yz, xz) = the tuple decode() returns, scales coarse -> fine, each [B,out_ch,P,P] in `layout`.
 * --------------------------------------------------------------------------------------------------------- */
typedef struct ddmi_decoder3d ddmi_decoder3d;
int ddmi_decoder3d_create(const ddmi_decoder2d_cfg* cfg, const int* inter_attn_resolutions, int n_inter, ddmi_decoder3d** out);
int ddmi_decoder3d_load_weight(ddmi_decoder3d* h, const char* key, const float* data, const int64_t* shape, int ndim);
int ddmi_decoder3d_finalize(ddmi_decoder3d* h);
int ddmi_decoder3d_num_scales(ddmi_decoder3d* h, int* n, int plane_hw[4]);
int ddmi_decoder3d_forward(ddmi_decoder3d* h, const float* z, void* const* planes, int layout, int B, void* stream);
int ddmi_decoder3d_destroy(ddmi_decoder3d* h);
/* Fused tri-plane sampling + occupancy MLP: replaces MLP3D.forward (models/d2c_vae/mlp.py:82-111): sample_plane_feature
 * + normalize_coordinate (utils/general_utils.py:71-119), triplane_positional_encoding(mode='add') (:126-131, bilinear,
 * align_corners=True, border) at 3 scales, net_p + 4 ResnetBlockFC (blocks.py:673-716) + net_out.
 * points [B,N,3] fp32; logits_out [B,N] fp32 (the Bernoulli logits the reference wraps in dist.Bernoulli). */
typedef struct ddmi_mlp3d ddmi_mlp3d;
int ddmi_mlp3d_create(const ddmi_mlp2d_cfg* cfg, ddmi_mlp3d** out);   /* in_ch 3, out_ch 1, ch 256, latent_dim 64 */
int ddmi_mlp3d_load_weight(ddmi_mlp3d* h, const char* key, const float* data, const int64_t* shape, int ndim);
int ddmi_mlp3d_finalize(ddmi_mlp3d* h);
int ddmi_hdpe_mlp3d(ddmi_mlp3d* h, const void* const* planes, const int* plane_hw, int layout, const float* points,
                    float* logits_out, int B, int N, void* stream);
int ddmi_mlp3d_destroy(ddmi_mlp3d* h);

/* ---------------------------------------------------------------------------------------------------------
 * Video (SkyTimelapse) path.  A latent is a token sequence [B, C, L], L = s1*s2 + s1*s3 + s2*s3, holding the planes
 * xy = [s1,s2] | xt = [s3,s1] | yt = [s3,s2] back to back (models/ldm/modules/diffusionmodules/openaimodel.py:1141-1143).
 *
 * Tri-plane UNet: replaces UNetModel_Triplane.forward (openaimodel.py:1113-1221): the UNetModel blocks on each plane
 * with shared weights + a 16-head 1-D attention over the concatenated tokens after every block
 * (make_attn('vanilla-1d-multihead') -> MemoryEfficientAttnBlock1D, models/d2c_vae/autoencoder_unet.py:404-468, with
 * xformers.ops.memory_efficient_attention = softmax(q k^T / sqrt(head_dim)) v).  Keys: the UNetModel_Triplane state
 * dict ("input_blocks.*", "input_attns.N.{norm,q,k,v,proj_out}.*", "mid_attn.*", "output_attns.N.*", ...).
 * The handle type and every other call (load_weight, finalize, forward, set_uniform_t, memory, destroy) are the
 * ddmi_unet_* ones; forward takes x / eps_out as [B, C, L].  cfg->image_size is ignored; size1 must equal size2.
 * --------------------------------------------------------------------------------------------------------- */
int ddmi_unet3_create(const ddmi_unet_cfg* cfg, int size1, int size2, int size3, ddmi_unet** out);

/* Video decoder: replaces VITAutoencoder.decode (models/d2c_vae/autoencoder_vit.py:188-202) = post_xy / post_yt /
 * post_xt 1x1 convs + VideoDecoder_light.forward (autoencoder_unet.py:1040-1144).  Keys: "post_{xy,yt,xt}.*",
 * "decoder.*".  z [B, embed_dim, L] fp32 with s1 = s2 = resolution/8, s3 = frames.
 * planes[plane*n_scales + scale], plane order (xy, yt, xt) = the tuple decode() returns, scales coarse -> fine;
 * xy planes are [B,out_ch,P,P], yt / xt planes [B,out_ch,frames,P], in `layout`. */
typedef struct ddmi_decoderv ddmi_decoderv;
typedef struct {
  int embed_dim, z_channels, resolution, frames, ch, num_res_blocks;
  int n_mult;   int ch_mult[8];
  int n_attn;   int attn_resolutions[8];
  int n_hdbf;   int hdbf_resolutions[8];
  int n_inter;  int inter_attn_resolutions[8];
  int attn_heads;              /* per-plane attention: 1 = attn_type 'vanilla', 16 = 'vanilla-multihead' */
  int max_batch;
  int numerics;
} ddmi_decoderv_cfg;
int ddmi_decoderv_create(const ddmi_decoderv_cfg* cfg, ddmi_decoderv** out);
int ddmi_decoderv_load_weight(ddmi_decoderv* h, const char* key, const float* data, const int64_t* shape, int ndim);
int ddmi_decoderv_finalize(ddmi_decoderv* h);
/* n scales; plane_hw[0..3] = P of each scale; out_ch of the planes */
int ddmi_decoderv_num_scales(ddmi_decoderv* h, int* n, int plane_hw[4], int* out_ch);
int ddmi_decoderv_forward(ddmi_decoderv* h, const float* z, void* const* planes, int layout, int B, void* stream);
int ddmi_decoderv_destroy(ddmi_decoderv* h);

/* Fused tri-plane sampling + video MLP: replaces MLPVideo.forward (models/d2c_vae/mlp.py:128-157):
 * triplane_positional_encoding(mode='concat') (utils/general_utils.py:134-145: three bilinear grid_samples,
 * align_corners=True, border, broadcast over (t,h,w) and channel-concatenated) at 3 scales + 4 ResnetBlockFC
 * (blocks.py:673-716) + Linear(leaky_relu(x, 0.2)).
 * planes[9] as ddmi_decoderv_forward returns them, plane_hw[3] = P per scale, frames = height of the yt / xt planes;
 * coords_xy [2,H,W], coords_yt [2,T,H], coords_xt [2,T,W] fp32: the grids of convert_to_coord_format_3d (channel 0
 * is handed to grid_sample as the plane's width coordinate, exactly as the reference does);
 * video_out [B, out_ch, T, H, W] fp32. */
typedef struct ddmi_mlpv ddmi_mlpv;
int ddmi_mlpv_create(const ddmi_mlp2d_cfg* cfg, ddmi_mlpv** out);   /* in_ch 2, out_ch 3, ch 256, latent_dim 64 */
int ddmi_mlpv_load_weight(ddmi_mlpv* h, const char* key, const float* data, const int64_t* shape, int ndim);
int ddmi_mlpv_finalize(ddmi_mlpv* h);
int ddmi_hdpe_mlpv(ddmi_mlpv* h, const void* const* planes, const int* plane_hw, int frames, int layout,
                   const float* coords_xy, const float* coords_yt, const float* coords_xt, float* video_out, int B,
                   int T, int H, int W, void* stream);
int ddmi_mlpv_destroy(ddmi_mlpv* h);

/* ---------------------------------------------------------------------------------------------------------
 * Post-path consumers: what the reference does with the path's outputs, on the device.
 * --------------------------------------------------------------------------------------------------------- */
/* Marching cubes on a device-resident logit grid: replaces `libmcubes.marching_cubes(occ_hat_padded, threshold)`
 * (convocc/src/conv_onet/generation.py:162; convocc/src/utils/libmcubes/mcubes.pyx:22-27, pywrapper.cpp:89-127,
 * marchingcubes.h:21-237, marchingcubes.cpp:290-326) with IDENTICAL output: the same vertex numbering and triangle
 * order as the reference's sequential sweep and bit-identical float64 coordinates.
 *   volume      device pointer, C-contiguous [nx][ny][nz], dtype 0 = float32, 1 = float64
 *   pad = 1     treats the grid as np.pad(volume, 1, constant_values=pad_value) (generation.py:160-161) without
 *               materialising the padded copy; pad = 0 takes the array as it is
 *   isovalue    the caller rounds it to C float first, as mcubes.pyx:22 (`float isovalue`) does
 * Two calls, because the sizes of the outputs are data dependent:
 *   ddmi_mcubes_count  classifies the cells and scans the counts into `workspace` (ddmi_mcubes_workspace_bytes, device
 *                      memory), synchronises `stream` and returns the number of vertices and triangles;
 *   ddmi_mcubes_emit   writes vertices [n_vertices][3] float64 and triangles [n_triangles][3] int64 (device pointers).
 *                      normalize != 0 fuses extract_mesh's vertex post-processing (generation.py:165-183, vol_bound
 *                      None): v -= 0.5; v -= 1; v /= (n_axis - 1); v = box_size * (v - 0.5), rounded op by op in
 *                      float64 like numpy, n_axis being the UNPADDED grid size.
 * A volume with fewer than 2 points along an axis yields 0 / 0 (the reference's loops never run). */
typedef struct ddmi_mcubes_grid {
  const void* volume;
  int dtype;
  int nx, ny, nz;
  int pad;
  float pad_value;
  double isovalue;
} ddmi_mcubes_grid;
int ddmi_mcubes_workspace_bytes(const ddmi_mcubes_grid* grid, size_t* bytes);
int ddmi_mcubes_count(const ddmi_mcubes_grid* grid, void* workspace, int64_t* n_vertices, int64_t* n_triangles, void* stream);
int ddmi_mcubes_emit(const ddmi_mcubes_grid* grid, const void* workspace, double* vertices, int64_t* triangles,
                     int normalize, double box_size, void* stream);
/* Image write-out of the FID / sample-saving loop (evals/eval.py:226-233): img [B,C,H,W] fp32 (the MLP's output,
 * unclamped or clamped) -> out [B,H,W,C] uint8 = trunc(((clamp(img,-1,1) + 1) / 2) * 255), each step rounded in fp32
 * exactly as torch / numpy do, so the bytes equal the reference's `(fake_img * 255).astype(np.uint8)`. */
int ddmi_image_to_u8(const float* img, uint8_t* out, int B, int C, int H, int W, void* stream);

/* The reference's compiled operator `fused.fused_bias_act(input, bias, refer, act, grad, alpha, scale)`
 * (models/d2c_vae/op/fused_bias_act.cpp:4-20 -> fused_bias_act_kernel.cu:19-49): out = act'(x + bias[(i / step_b) % size_b]) * scale
 * with act 1 = linear, 3 = leaky ReLU(alpha) and grad 0 = value, 1 = gradient gated by the sign of `refer` (the forward's output),
 * 2 = zeros; bias / refer may be null (the reference passes empty tensors); step_b = product of the dims after the channel dim,
 * size_b = channels; dtype 0 fp32, 1 bf16, 2 fp16 (one type for all four tensors).  Device pointers, asynchronous on `stream`.
 * `ddmi_amd/fused_act.py` wraps it as the module object the reference's fused_act.py builds with torch's JIT loader. */
int ddmi_fused_bias_act(const void* x, const void* bias, const void* refer, void* y, int64_t size_x, int64_t step_b, int size_b,
                        int act, int grad, float alpha, float scale, int dtype, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * Diagnostic single-op entry points (used by the parity tests; reference-layout tensors in, reference-layout
 * tensors out; they allocate temporaries and synchronise - not for the hot path).
 * --------------------------------------------------------------------------------------------------------- */
/* F.conv2d(x, w, b, stride, padding=k/2) optionally preceded by nearest x2 upsampling; k in {1,3} */
int ddmi_op_conv2d(const float* x, const float* w, const float* bias, float* y, int B, int Cin, int H, int W,
                   int Cout, int k, int stride, int upsample, int numerics, int splitk);
/* conv3x3(x, w, bias) + conv1x1(cat([xs1, xs2], 1), ws, bias_s): a ResBlock's last convolution plus its skip_connection
 * (models/ldm/modules/diffusionmodules/openaimodel.py:272-275, `self.skip_connection(x) + h`; xs2 / C3 = 0 when the block
 * input is one tensor, two tensors for the `th.cat([h, hs.pop()], dim=1)` input of :787).  In bf16 mode it is ONE launch:
 * the 1x1 convolution is folded into the 3x3 kernel's K loop (shapes of the halo kernel: W in {8,16,32,64}); fp32 runs the
 * launches the reference's op order implies.  ws is [Cout, C2 + C3] (the skip_connection weight with its 1x1 dims dropped). */
int ddmi_op_conv2d_skip(const float* x, const float* w, const float* bias, const float* xs1, const float* xs2, const float* ws,
                        const float* bias_s, float* y, int B, int Cin, int H, int W, int Cout, int C2, int C3, int numerics,
                        int splitk);
/* micro-benchmark of one convolution shape on device-resident pseudo-random data (average ms per launch);
 * force_bm/force_bn = 0 lets the library pick the tile; splitk = 0 lets it pick the split */
int ddmi_bench_conv(int B, int Cin, int H, int W, int Cout, int k, int stride, int upsample, int numerics, int splitk,
                    int force_bm, int force_bn, int iters, double* ms_out, int* splitk_used);
/* diagnostics: pin the bf16 convolution kernel variant (index into k_conv.hip's kRing table) used by the following
 * ddmi_op_conv2d calls of this process, so each variant can be parity-tested on its own; cfg_id < 0 restores the
 * library's choice.  A variant that does not apply to a shape (the halo kernel outside 3x3/stride 1) falls back. */
int ddmi_debug_conv_config(int cfg_id);
/* diagnostics: the halo kernel's two-taps-per-stage tail for a ragged last channel block (Cin = 64 k + 1..32, no split-K; k_conv.hip
 * HALF instantiation; same products in the same order, so results are bit-identical to the plain form): 0 = off, 1 = on,
 * -1 = the DDMI_HALO_HALF environment (default on). */
int ddmi_debug_halo_half(int mode);
/* GroupNorm(32 groups, eps) [+ SiLU] over [B,C,HW] */
int ddmi_op_groupnorm(const float* x, const float* gamma, const float* beta, float* y, int B, int C, int HW, float eps,
                      int silu, int numerics);
/* QKVAttentionLegacy (openaimodel.py:397-422): qkv [B, heads*3*32, T] -> out [B, heads*32, T] */
int ddmi_op_attention_legacy(const float* qkv, float* out, int B, int heads, int T, int numerics);
/* same with an explicit head dim (UNetModel_Triplane runs num_head_channels = 64): qkv [B, heads*3*d, T] */
int ddmi_op_attention_legacy_hd(const float* qkv, float* out, int B, int heads, int d, int T, int numerics);
/* multi-head attention of the video path's MemoryEfficientAttnBlock* modules (autoencoder_unet.py:273-335, 404-534;
 * xformers.ops.memory_efficient_attention semantics): q, k, v, out [B, heads*d, N] with channel = head*d + j,
 * out = softmax(q^T k / sqrt(d)) v per head.  seg0 / seg1 split the N tokens into up to three row ranges that are
 * held in separate buffers internally (N - seg0 - seg1 is the third), the way the xy | xt | yt planes are. */
int ddmi_op_attention_mh(const float* q, const float* k, const float* v, float* out, int B, int heads, int d, int N,
                         int seg0, int seg1, int numerics);
/* The same in the arrangement the bf16 NETWORKS run (models.hip `make_qkv_attn`: the softmax scale x log2 e is folded into the q rows of
 * the qkv projection at finalize): q already carries d^-1/2 * log2(e), the scores are base-2 exponents and no scale is applied.  bf16
 * only.  With d = 16 / 32, N a multiple of 512 (K + V of a head <= 128 KB) and segments that are multiples of 64 tokens this is the
 * RESIDENT kernel (k_attention.hip `attention_res_kernel`: one workgroup per (sample, head), K / V in LDS once), else the tiled one;
 * DDMI_ATTN_RES=0 in the environment pins the tiled kernel. */
int ddmi_op_attention_mh_prescaled(const float* q, const float* k, const float* v, float* out, int B, int heads, int d, int N,
                                   int seg0, int seg1);
/* One AttentionBlock of the LDM UNet (openaimodel.py:279-325 with QKVAttentionLegacy :397-426), built the way the UNet handle
 * builds it: x + proj_out(attention(qkv(GroupNorm32(x)))); x, y [B, C, H, W]; wqkv [3C, C], wproj [C, C] (Conv1d weights with the
 * kernel dim dropped); head_channels = num_head_channels.  In bf16 mode, head dim 32 and H W in {64, 256, 1024} the qkv projection
 * and the attention are ONE launch (k_attnblock.hip: the [B, T, 3C] qkv tensor is never written); *fused_used reports which form
 * ran.  ddmi_debug_attn_fused pins the choice for networks built afterwards: 0 = five launches, 1 = fused where the shape
 * allows, -1 = the DDMI_ATTN_FUSED / DDMI_ATTN_FUSED_N environment (default: fused). */
int ddmi_op_attention_block(const float* x, const float* gamma, const float* beta, const float* wqkv, const float* bqkv,
                            const float* wproj, const float* bproj, float* y, int B, int C, int H, int W, int head_channels,
                            int numerics, int* fused_used);
int ddmi_debug_attn_fused(int mode);
/* conv3x3 + bias (+ emb [B, Cout] per-sample addend, may be null) -> GroupNorm(32 groups, eps) [+ SiLU] as the LDM ResBlock runs the
 * pair h = in_layers(x) + emb_out; out_layers[0:2](h) (openaimodel.py:261-270): x [B,Cin,H,W], w [Cout,Cin,3,3] -> y [B,Cout,H,W].  In bf16
 * mode, where the launch qualifies (k_conv.hip conv_fgn_ok), the GroupNorm is the convolution's epilogue and the raw tensor is never
 * stored; *fused_used reports it.  `iters` repeats the run.  ddmi_debug_gn_fuse pins the choice for networks built AND run afterwards:
 * 0 = separate pass, 1 = fused where the launch runs on the halo kernel without split-K anyway, 2 = fused wherever the shape allows
 * (the launch is moved to the halo kernel without split-K: small test batches), -1 = the DDMI_GN_FUSE environment (default 1). */
int ddmi_op_conv_gn(const float* x, const float* w, const float* bias, const float* emb, const float* gamma, const float* beta, float* y,
                    int B, int Cin, int H, int W, int Cout, float eps, int silu, int numerics, int iters, int* fused_used);
int ddmi_debug_gn_fuse(int mode);
/* The small-batch regime of the bf16 networks (ddmi_amd/csrc/net.h; the reference's own gen mode runs test_batch_size 6-10,
 * configs/ldm/afhq.yaml:76): at B <= 16 (DDMI_SMALL_BATCH_MAX) a GroupNorm whose (sample, group) slice fits a workgroup's registers is ONE
 * launch, and networks created with max_batch <= 16 keep the five-launch AttentionBlock and the two accumulating 1x1
 * launches of a concatenated skip connection.  mode: 0 = off (every batch runs the kernels of the large-batch path - what the parity
 * suite pins, tests/conftest.py), 1 = on, -1 = the DDMI_SMALL_BATCH environment (default on).  Read when a network is built AND run. */
int ddmi_debug_small_batch(int mode);
/* diagnostics: one-launch GroupNorms of the last network run of this process */
int ddmi_debug_small_gn_launches(int* count);
/* the one-launch GroupNorm on its own (bf16): GroupNorm32(eps) [+ SiLU] of the channel concatenation [x [B,C1,HW] | x2 [B,C2,HW] or null]
 * -> y [B,C1+C2,HW]; gamma / beta [C1+C2] DEVICE pointers like ddmi_op_groupnorm's */
int ddmi_op_groupnorm_small(const float* x, const float* x2, const float* gamma, const float* beta, float* y, int B, int C1, int C2, int HW,
                            float eps, int silu);
/* HOST-ONLY (no device needed): would the pair above - a bf16 3x3 / stride-1 convolution [B,Cin,H,W] -> Cout followed by GroupNorm(32) - run
 * with the GroupNorm in the convolution's epilogue on the halo kernel without split-K (k_conv.hip conv_fgn_ok: whole groups per channel
 * tile, whole 256-row tiles per sample, at most 16 workgroups waiting for each other, statistics exchanged only at batches that are
 * multiples of 8 unless any_batch)?  *fusable = 0 / 1; *tiles_per_sample (may be null) = H W / 256, the workgroups of one channel tile
 * that exchange partial statistics (1 = none).  any_batch bit 1 (value 2): answer for a network WITHOUT the cross-workgroup hand-off (every
 * network but the UNet handles', whose owner checks ddmi_unet_async_status): only tiles_per_sample == 1 shapes fuse there. */
int ddmi_debug_conv_gn_fusable(int B, int Cin, int H, int W, int Cout, int any_batch, int* fusable, int* tiles_per_sample);
/* The ResBlock's skip_connection on a channel-concatenated input (openaimodel.py:787 `th.cat([h, hs.pop()], dim=1)`, :272-275): y =
 * conv1x1([x1 | x2]) + bias with x1 [B,C1,H,W], x2 [B,C2,H,W], w [Cout, C1 + C2] -> y [B,Cout,H,W].  In bf16 mode, for Cout a multiple
 * of 224 or 128, it is ONE launch that reads both tensors in place (k_conv.hip, CAT instantiation of the persistent 1x1 kernel;
 * *cat_used = 1), else two accumulating launches over the column blocks of w.  DDMI_CONV_CAT=0 in the environment keeps the two. */
int ddmi_op_conv1x1_cat(const float* x1, const float* x2, const float* w, const float* bias, float* y, int B, int C1, int C2, int H, int W,
                        int Cout, int numerics, int* cat_used);
/* average device time (us) of one bf16 AttentionBlock with num_head_channels 32 on device-resident pseudo-random data */
int ddmi_bench_attention_block(int B, int C, int H, int W, int iters, double* us_out, int* fused_used);
/* On-box peak microbenchmarks: what this GPU sustains, beside the nominal 2.5 PFLOP/s / 8 TB/s the rooflines are
 * quoted against.  peak_mfma: nothing but v_mfma_f32_16x16x32_bf16 on pseudo-random register operands, every CU, two waves per SIMD,
 * for about `seconds_target` seconds -> dense bf16 TFLOP/s.  peak_copy: a 16-byte-per-lane device copy of `gib` GiB (needs 2 x that
 * much free memory), five repetitions -> (bytes read + bytes written) per second in TB/s. */
int ddmi_bench_peak_mfma(double seconds_target, double* tflops);
int ddmi_bench_peak_copy(double gib, double* tb_per_s);
/* timestep_embedding (util.py:150-170): t [B] int64 -> [B, dim] */
int ddmi_op_timestep_embedding(const int64_t* t, float* out, int B, int dim);

/* ---------------------------------------------------------------------------------------------------------
 * Training: the pieces of LDMTrainer.train's inner step (tools/ldm/image.py:136-178) =
 * DDPM.forward -> p_losses (diffusion/ddpm.py:409-454) -> backward -> AdamW (tools/ldm/image.py:62, 171-172), as stateless
 * ops.  Activations are caller-owned NHWC tensors in the handle's storage dtype (fp32 for DDMI_PARITY_FP32, bf16 for
 * DDMI_PERF_BF16); parameters and their gradients are fp32 in the reference's own layouts (the state-dict tensors), so the
 * host can all-reduce the gradients with RCCL (accelerate's DDP wrapper in the reference, tools/ldm/image.py:96) and keep
 * the optimizer state.  The host side (ddmi_amd/train.py) sequences them with an autograd tape; every FLOP runs here.
 * The handle owns only a scratch arena (grown on demand; a growth synchronises the stream once).
 * --------------------------------------------------------------------------------------------------------- */
typedef struct ddmi_train ddmi_train;
int ddmi_train_create(int numerics, ddmi_train** out);
int ddmi_train_destroy(ddmi_train* h);
/* layout seams: fp32 NCHW <-> NHWC in the storage dtype */
int ddmi_train_to_nhwc(ddmi_train* h, const float* x_nchw, void* y_nhwc, int B, int C, int HW, void* stream);
int ddmi_train_to_nchw(ddmi_train* h, const void* x_nhwc, float* y_nchw, int B, int C, int HW, void* stream);
/* y = conv_kxk(nearest_x2?(x), w) + bias + emb[b] + res: nn.Conv2d / Conv1d(k=1) of openaimodel.py (ResBlock :164-276 with the
 * time-embedding addend, Downsample stride 2 :135-161, Upsample :92-120).  x [B,H,W,Cin]; w fp32 [Cout,Cin,k,k];
 * bias [Cout] / emb [B,emb_stride] fp32 / res [B,Ho,Wo,Cout] nullable. */
int ddmi_train_conv_fwd(ddmi_train* h, const void* x, const float* w, const float* bias, const float* emb, int emb_stride,
                        const void* res, void* y, int B, int H, int W, int Cin, int Cout, int k, int stride, int ups, void* stream);
/* gradients of the above for dy [B,Ho,Wo,Cout]: dx [B,H,W,Cin] (nullable), dw fp32 [Cout,Cin,k,k], dbias [Cout] (nullable),
 * demb [B,Cout] fp32 (nullable; = the per-sample channel sums of dy).  d(res) = dy is the caller's. */
int ddmi_train_conv_bwd(ddmi_train* h, const void* x, const float* w, const void* dy, void* dx, float* dw, float* dbias, float* demb,
                        int B, int H, int W, int Cin, int Cout, int k, int stride, int ups, void* stream);
/* GroupNorm32 (+ SiLU) util.py:198-215; meanrstd [B,32,2] fp32 is saved for the backward */
int ddmi_train_gn_fwd(ddmi_train* h, const void* x, const float* gamma, const float* beta, void* y, float* meanrstd, int B, int HW,
                      int C, float eps, int silu, void* stream);
int ddmi_train_gn_bwd(ddmi_train* h, const void* x, const float* gamma, const float* beta, const float* meanrstd, const void* dy,
                      void* dx, float* dgamma, float* dbeta, int B, int HW, int C, int silu, void* stream);
/* dx = GroupNorm backward(dy) + dres: the block input's second gradient (its residual / skip path, openaimodel.py:272-275, 321-325)
 * added in the backward's last pass instead of by the host's tape; dres [B,HW,C] in the handle's storage type (null: as above) */
int ddmi_train_gn_bwd_res(ddmi_train* h, const void* x, const float* gamma, const float* beta, const float* meanrstd, const void* dy,
                          const void* dres, void* dx, float* dgamma, float* dbeta, int B, int HW, int C, int silu, void* stream);
/* th.cat([h, hs.pop()], dim=1) of the decoder (openaimodel.py:739) on NHWC rows, and its gradient: out [rows, Ca+Cb] = [a | b];
 * split2 copies the two column blocks of `in` back to a and b.  rows = B*H*W; Ca, Cb multiples of the 16-byte vector. */
int ddmi_train_cat2(ddmi_train* h, const void* a, const void* b, void* out, int64_t rows, int Ca, int Cb, void* stream);
int ddmi_train_split2(ddmi_train* h, const void* in, void* a, void* b, int64_t rows, int Ca, int Cb, void* stream);
/* QKVAttentionLegacy openaimodel.py:397-426: qkv [B,T,heads*3*d] (per-head [q|k|v]), out [B,T,heads*d]; backward d in {2, 4, 8, 16, 32, 64, 96, 128}
 * (32: MFMA kernels; the others - the tri-plane UNet's 64-channel heads and its 16-head inter-plane attention, openaimodel.py:955-957 -
 * on the plain one-thread-per-row kernels) */
int ddmi_train_attn_fwd(ddmi_train* h, const void* qkv, void* out, int B, int T, int heads, int d, void* stream);
int ddmi_train_attn_bwd(ddmi_train* h, const void* qkv, const void* out, const void* dout, void* dqkv, int B, int T, int heads, int d,
                        void* stream);
/* the log-sum-exp pair (ddmi_train_attn_fwd_lse / _bwd_lse) for heads of d < 32 channels that the caller ZERO-PADS to 32 (all four tensors in
 * the padded layout, heads * 3 * 32 / heads * 32 channels): `scale` = 1 / sqrt(d) replaces the kernels' 1 / sqrt(32).  bf16, T % 64 == 0. */
int ddmi_train_attn_fwd_lse_scaled(ddmi_train* h, const void* qkv, void* out, float* lse, int B, int T, int heads, float scale, void* stream);
int ddmi_train_attn_bwd_lse_scaled(ddmi_train* h, const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, int B, int T,
                                   int heads, float scale, void* stream);
/* the same pair with the forward's row log-sum-exp saved for the backward (lse [B,heads,T] fp32, natural log of the sum of
 * exp(scaled scores)): the backward then makes one pass over the scores instead of three.  Available when
 * ddmi_train_attn_lse_ok() returns 1 (bf16 handle, d = 32, T a multiple of 64); refused (DDMI_ERR_INVALID) otherwise. */
int ddmi_train_attn_lse_ok(ddmi_train* h, int T, int d);
int ddmi_train_attn_fwd_lse(ddmi_train* h, const void* qkv, void* out, float* lse, int B, int T, int heads, int d, void* stream);
int ddmi_train_attn_bwd_lse(ddmi_train* h, const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, int B, int T,
                            int heads, int d, void* stream);
/* y = act(x) W^T + b (fp32 [B,K] -> [B,N]; act = SiLU when silu_in): time_embed and the ResBlock emb_layers */
int ddmi_train_linear_fwd(ddmi_train* h, const float* x, const float* W, const float* b, float* y, int B, int K, int N, int silu_in,
                          void* stream);
int ddmi_train_linear_bwd(ddmi_train* h, const float* x, const float* W, const float* dy, float* dx, float* dW, float* db, int B, int K,
                          int N, int silu_in, void* stream);
/* Gradient-accumulation fusion (what accelerate's `accumulate` + torch's AccumulateGrad do for the reference, tools/ldm/image.py:160-170):
 * the PARAMETER-gradient outputs of the NEXT ddmi_train_conv_bwd / ddmi_train_gn_bwd call on this handle are added to their
 * destination instead of overwriting it - mask bit 0: dw, bit 1: dbias, bit 2: dgamma and dbeta.  One-shot. */
int ddmi_train_accumulate_next(ddmi_train* h, int mask);
/* Number of times the handle's scratch arena has been re-allocated (it grows on the first call of a larger shape, never
 * inside a stream capture: that call fails with DDMI_ERR_STATE).  A hipGraph captured over these entry points holds
 * pointers into the arena of its time: a caller that replays graphs compares this counter and re-captures when it moved. */
int ddmi_train_arena_generation(ddmi_train* h, int* generation);
/* fp32 GEMM with strided operands, C[m][n] = bias[n] + sum_r A[m*sam + r*sar] * B[r*sbr + n*sbn] (bias may be NULL): the batched
 * ResBlock.emb_layers projection of a training step (forward, input gradient, weight gradient are three orientations of it) */
int ddmi_train_matmul(ddmi_train* h, const float* A, int64_t sam, int64_t sar, const float* B, int64_t sbr, int64_t sbn, float* C,
                      const float* bias, int M, int N, int R, void* stream);
int ddmi_train_timestep_embedding(ddmi_train* h, const int64_t* t, float* out, int B, int dim, void* stream);   /* util.py:150-170 */
/* q_sample ddpm.py:383-386 with per-sample coefficients sqrt(abar_t), sqrt(1 - abar_t) [B] */
int ddmi_train_q_sample(const float* x0, const float* noise, const float* sqrt_ac, const float* sqrt_1mac, float* x_t, int B,
                        int64_t per_sample, void* stream);
/* p_losses ddpm.py:409-447 for parameterization 'eps', loss_type 'l2', l_simple_weight 1, original_elbo_weight 0:
 * loss = mean((mix(model_out) - noise)^2) with the mixed prediction of :469-477; also d loss / d model_out [B,C,HW] fp32 NCHW
 * and d loss / d mixing_logit [C] (mixing_logit nullable = no mixing) */
int ddmi_train_eps_loss(ddmi_train* h, const float* model_out, const float* x_t, const float* noise, const float* sqrt_1mac,
                        const float* mixing_logit, float* loss, float* d_model_out, float* d_logit, int B, int C, int HW, void* stream);
/* torch.optim.AdamW over a flat fp32 slab; grad is multiplied by grad_scale first (1 / world_size after a sum all-reduce,
 * 1 / gradient_accumulate_every) */
int ddmi_train_adamw(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1, float beta2,
                     float eps, float weight_decay, int step, float grad_scale, void* stream);
/* ema_pytorch's parameter average (tools/ldm/image.py:64, 174): ema += (param - ema) * (1 - decay) */
int ddmi_train_ema(float* ema, const float* param, int64_t n, float decay, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DDMI_HIP_H */
