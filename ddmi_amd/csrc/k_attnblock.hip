// Fused AttentionBlock body for the LDM UNet (openaimodel.py:279-325 `AttentionBlock._forward`, :397-426 `QKVAttentionLegacy`):
//     qkv = Conv1d(C, 3C, 1)(GroupNorm(x));  a = softmax(q k^T / sqrt(d)) v  per head
// as ONE launch that never writes the [B, T, 3C] qkv tensor (1.4 GB at 32x32, B = 512: written by the 1x1 convolution and read back
// by the attention kernel in the five-launch form).  `proj_out` + residual stay the persistent 1x1 kernel's (k_conv.hip).
//
// Workgroup = 8 waves = one head of one GROUP of 1024 token rows (1 sample at 32x32, 4 at 16x16, 16 at 8x8).
//   Phase A  q | k | v of the head for the group's 1024 rows = x[1024][C] . Wh^T[C][96] on MFMA, operands swapped (A = weight rows,
//            B = token rows) so a lane ends with 8 consecutive channels of one token:
//              - x goes global -> registers directly as B fragments (each row is used by exactly one wave: nothing to share, no LDS
//                round trip); the GroupNorm in front writes it in fragment order, so a fragment is one contiguous KiB (from the
//                row-major tensor every 16-lane group of a load touched 16 cache lines: 12 B/clk per CU, the projection ran at
//                a quarter of the MFMA rate); wave w owns rows [128 w, 128 w + 128) in two passes of 64;
//              - the head's weights stream through a 2 x 12 KB LDS stage per 64-channel chunk by LDS-DMA from a host-packed image
//                that already IS the swizzled LDS layout (a linear 12 KB copy), one chunk ahead of its MFMAs, one barrier per chunk
//                (measured and dropped: a four-stage ring in the unwritten halves of the K / V rows, three chunks ahead - 24.1 vs 19.7 us
//                per projection; the second pass's first chunk fetched under the first pass's last step - 20.1 vs 19.6; x fragments
//                ROLLING through the two register sets, each k-step's registers refilled with chunk c + 2 right behind its MFMAs -
//                40.3 vs 37.2 us at 16x16 where the loop stayed clean, and hipcc's wait-count pass answers any ragged loop shape
//                with vmcnt(0..7) waits behind the refills or, with the loads hidden in inline asm, with spills);
//              - K and V land in LDS for all 1024 rows (2 x 64 KB, the attention kernel's row layouts); Q never leaves registers:
//                the wave that projected a row is the wave that runs that row's queries (8 B fragments = 32 VGPRs).
//   Phase B  attention_v2_kernel's lazy-maximum loop (k_attention.hip) over the LDS-resident K / V of the query's own sample:
//            no global loads, no staging stores and no barrier inside the loop.
// XCD x owns the groups g = x (mod 8) whole (all heads of a group read the same x rows through one L2).
#include "kernels.h"
#include "prof.h"
#include <cstdlib>
#include <cstring>
#include <math.h>
#include <type_traits>
#include <vector>

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((address_space(3))) void lds_void;

// timing-only ablation builds (tools/build_variant.sh -DAB_DBG=n; results are wrong): 1 no x loads after chunk 0, 2 no weight DMA after
// chunk 0, 4 no per-chunk barrier, 8 no MFMAs in the projection, 16 a per-(sample, channel) affine transform applied to every x fragment
// between its load and its MFMAs - what folding the AttentionBlock's GroupNorm (no SiLU: openaimodel.py:300, 320) into this kernel's
// x-fragment load would add to the projection (round-5 verdict item 1a; DESIGN_r6.md has the numbers)
#ifndef AB_DBG
#define AB_DBG 0
#endif
#ifndef AB_OUT_SC1
#define AB_OUT_SC1 1
#endif
#ifndef AB_PRIO
#define AB_PRIO 1
#endif
#ifndef AB_BOUNDED
#define AB_BOUNDED 1   // 0: always track the running maximum (A/B builds)
#endif
#define AB_ROWS 1024                 // token rows per workgroup
#define AB_KV_BYTES (AB_ROWS * 64)   // one of K / V: [row][32 channels] bf16
#define AB_WCHUNK (96 * 128)         // one 64-channel chunk of a head's [96][C] weights: [96 rows][128 B]

struct AttnBlockK {
  const void* x;
  const void* w;
  const float* bias;
  void* out;
  unsigned x_bytes;
  int rows;        // B * N valid token rows
  int N, C, heads, groups, xcd;
  unsigned long long* ts;   // diagnostics (attnblock_debug_ts): 16 words per workgroup, 100 MHz ticks and shader-clock stamps
};

// K rows: attention_v2's 64-byte-row swizzle (conflict-free ds_read_b128 fragments); V rows: its transposed-read swizzle
__device__ __forceinline__ int ab_kswz(int row, int slot) {
  int r = (row >> 2) & 3;
  return slot ^ ((0x78 >> (r * 2)) & 3);
}
__device__ __forceinline__ int ab_vswz(int key) { return 2 * ((key >> 2) & 1); }

template <int QF>
__global__ void __launch_bounds__(512) attnblock_kernel(AttnBlockK a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ab_smem[];
  unsigned char* const Ks = ab_smem;
  unsigned char* const Vs = ab_smem + AB_KV_BYTES;
  unsigned char* const Wst = ab_smem + 2 * AB_KV_BYTES;   // [2][AB_WCHUNK]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lg = lane >> 4;
  int g, h;
  {
    const int L = blockIdx.x;
    if (a.xcd) {   // XCD x (= L % 8 under round-robin dispatch) owns the groups g = x (mod 8); heads of a group consecutive
      const int slot = L >> 3, gl = slot / a.heads;
      h = slot - gl * a.heads;
      g = gl * 8 + (L & 7);
    } else {
      g = L / a.heads;
      h = L - g * a.heads;
    }
    if (g >= a.groups) return;   // workgroup-uniform, before any barrier
  }
  const int N = a.N, C = a.C, NC = C >> 6;
  const int row0 = g * AB_ROWS;
  unsigned long long* const ts = a.ts && tid == 0 ? a.ts + 16 * blockIdx.x : nullptr;
  if (ts) { ts[0] = wall_clock64(); ts[12] = __builtin_amdgcn_s_memtime(); }
  auto x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
  auto w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)((const unsigned char*)a.w + (size_t)h * NC * AB_WCHUNK), 0, NC * AB_WCHUNK, 0x00020000);
  const float* const biasp = a.bias + h * 96;
  auto o_rsrc = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, a.x_bytes, 0x00020000);

  float* const bias_lds = reinterpret_cast<float*>(Wst + 2 * AB_WCHUNK + 64);   // [96], visible after the first step's barrier
  if (tid < 96) bias_lds[tid] = biasp[tid];
  // ================= Phase A: q | k | v projection ==================================================================
  bf16x8 qb[8];
  // Bound on every score of this wave's queries, |q . k| <= |q| |k| (phase B): a row's |.|^2 is the sum of its four lanes' 8-channel
  // partials, so sum over the four lane groups of (max over rows of the partial) bounds every row's - one reduction per wave at the end
  // instead of two cross-lane sums per fragment (a bound up to 2x looser per factor; the margin it is compared with is 2^100)
  float qmax2 = 0.f, kmax2 = 0.f;
  // the bounded attention tiles (phase B) pay for these norms only where a round has many tiles: 32x32 (16 tiles per round: attention
  // phase 41.4 -> 36.4 us); at 4 tiles or 1 the first, tracked, tile is most of the round and the norms cost more than they return
  const bool bounded = AB_BOUNDED && a.N >= 512;
  const int wsw = ((lr >> 1) & 7);
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int tok0 = wave * 128 + p * 64;   // first LDS row of this pass
    f32x4 acc[4][6];
#pragma unroll
    for (int tf = 0; tf < 4; ++tf)
#pragma unroll
      for (int cf = 0; cf < 6; ++cf) acc[tf][cf] = f32x4{0.f, 0.f, 0.f, 0.f};
    // x is in B-fragment order [16-token block][k-step][lane][8] (launch_gn_apply frag_out): one fragment = one contiguous KiB.
    // Blocks past the last sample start beyond x_bytes: the descriptor's range check returns zeros
    const unsigned xl = (unsigned)((row0 + tok0) >> 4) * (unsigned)(C * 32) + (unsigned)(lane * 16);
    u32x4 xa[4][2], xb[4][2];
    auto load_x = [&](int c, u32x4(&xr)[4][2]) {
#pragma unroll
      for (int tf = 0; tf < 4; ++tf)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
          xr[tf][ks] = __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, xl + (unsigned)(tf * C * 32 + (2 * c + ks) * 1024), 0, 0);
    };
    auto dma_w = [&](int c, int buf) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int piece = wave + 8 * i;
        if (piece < 12)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_void*)(Wst + buf * AB_WCHUNK + piece * 1024), 16,
                                                   (unsigned)(c * AB_WCHUNK + piece * 1024 + lane * 16), 0, 0, 0);
      }
    };
    // Chunk ORDER: head h starts at chunk h % NC and wraps.  All heads of a group run at once, in step, on the same x rows: walking
    // the chunks in the same order made every workgroup wait for the same HBM miss at every chunk (the projection alone took 19 us
    // with its MFMAs removed); staggered, a chunk is first touched by one or two heads and is an L2 hit for the others a step later.
    const int cs0 = h % NC;
    auto chunk_of = [&](int c) { const int t = c + cs0; return t >= NC ? t - NC : t; };
    auto step = [&](int c, u32x4(&cur)[4][2], u32x4(&nxt)[4][2]) {
      // the BUILTIN, not inline asm: hipcc's wait-count pass cannot see into an asm string, keeps the chunk-c loads on its scoreboard and
      // then waits for them again behind the chunk c + 1 loads (vmcnt retires in order: that wait drained the whole prefetch)
      __builtin_amdgcn_s_waitcnt(0x0F70);                // vmcnt(0): this wave's weight pieces and x fragments of chunk c have landed
      if (!(AB_DBG & 4)) __builtin_amdgcn_s_barrier();   // everybody's have; everybody is done reading stage (c + 1) & 1
      if (c + 1 < NC) {
        if (!(AB_DBG & 2)) dma_w(chunk_of(c + 1), (c + 1) & 1);
        if (!(AB_DBG & 1)) load_x(chunk_of(c + 1), nxt);
      }
      const unsigned char* wb = Wst + (c & 1) * AB_WCHUNK + lr * 128;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8 wf[6];
#pragma unroll
        for (int cf = 0; cf < 6; ++cf) wf[cf] = *reinterpret_cast<const bf16x8*>(wb + cf * 2048 + (((4 * ks + lg) ^ wsw) << 4));
#pragma unroll
        for (int tf = 0; tf < 4; ++tf) {
          bf16x8 xf = __builtin_bit_cast(bf16x8, cur[tf][ks]);
#if AB_DBG & 16
          {   // timing only: scale / shift of the lane's 8 channels of this k-step from (any) fp32 memory, 8 conversions + 8 FMAs + 8 roundings
            const float* sp = bias_lds + ((8 * (4 * ks + lg) + 16 * (c & 1)) & 63);
            const f32x4 sa = *reinterpret_cast<const f32x4*>(sp), sb = *reinterpret_cast<const f32x4*>(sp + 4);
            const f32x4 ha = *reinterpret_cast<const f32x4*>(sp + 8), hb = *reinterpret_cast<const f32x4*>(sp + 12);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              xf[e] = (bf16_t)__builtin_fmaf((float)xf[e], sa[e], ha[e]);
              xf[4 + e] = (bf16_t)__builtin_fmaf((float)xf[4 + e], sb[e], hb[e]);
            }
          }
#endif
#pragma unroll
          for (int cf = 0; cf < 6; ++cf)
            if (!(AB_DBG & 8)) acc[tf][cf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[cf], xf, acc[tf][cf], 0, 0, 0);
            else acc[tf][cf][0] += (float)wf[cf][0] + (float)xf[0];
        }
      }
    };
    dma_w(chunk_of(0), 0);
    load_x(chunk_of(0), xa);
#pragma unroll 1
    for (int c = 0; c < NC; c += 2) {
      step(c, xa, xb);
      if (c + 1 < NC) step(c + 1, xb, xa);
    }
    // ---- epilogue: + bias; Q -> B fragments in registers, K / V -> LDS rows (the weight rows were permuted on the host so that the
    //      two fragments of a pair hand a lane 8 consecutive positions) -----------------------------------------------------------
    f32x4 bv[6];   // from the LDS copy made at kernel start: a global load here sits on the critical path of both passes
#pragma unroll
    for (int cf = 0; cf < 6; ++cf) bv[cf] = *reinterpret_cast<const f32x4*>(bias_lds + cf * 16 + 4 * lg);
#pragma unroll
    for (int tf = 0; tf < 4; ++tf) {
      const int tok = tok0 + tf * 16 + lr;
      bf16x8 q8, k8, v8;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        q8[j] = (bf16_t)(acc[tf][0][j] + bv[0][j]);
        q8[4 + j] = (bf16_t)(acc[tf][1][j] + bv[1][j]);
        k8[j] = (bf16_t)(acc[tf][2][j] + bv[2][j]);
        k8[4 + j] = (bf16_t)(acc[tf][3][j] + bv[3][j]);
        v8[j] = (bf16_t)(acc[tf][4][j] + bv[4][j]);
        v8[4 + j] = (bf16_t)(acc[tf][5][j] + bv[5][j]);
      }
      qb[p * 4 + tf] = q8;
      if (bounded) {   // partial square norms of the ROUNDED q and k (what the MFMAs multiply) over the lane's 8 channels: running maxima
        float qs = 0.f, ks2 = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float qv = (float)q8[e], kv = (float)k8[e];
          qs = __builtin_fmaf(qv, qv, qs);
          ks2 = __builtin_fmaf(kv, kv, ks2);
        }
        qmax2 = fmaxf(qmax2, qs);
        kmax2 = fmaxf(kmax2, ks2);
      }
      *reinterpret_cast<bf16x8*>(Ks + tok * 64 + ab_kswz(tok, lg) * 16) = k8;
      *reinterpret_cast<bf16x8*>(Vs + tok * 64 + ((lg ^ ab_vswz(tok)) << 4)) = v8;
    }
    if (p == 1 && bounded) {   // max over the 16 rows of a lane group, then the sum over the four lane groups; the keys' goes to a slot
      auto bound2 = [&](float t) {   // behind the weight stages (they may still be read by slower waves)
        t = fmaxf(t, __shfl_xor(t, 1)); t = fmaxf(t, __shfl_xor(t, 2)); t = fmaxf(t, __shfl_xor(t, 4)); t = fmaxf(t, __shfl_xor(t, 8));
        t += __shfl_xor(t, 16); t += __shfl_xor(t, 32);
        return t;
      };
      qmax2 = bound2(qmax2);
      kmax2 = bound2(kmax2);
      if (lane == 0) reinterpret_cast<float*>(Wst + 2 * AB_WCHUNK)[wave] = kmax2;
    }
    __syncthreads();   // the weight stages are free for the next pass; after the second pass: every K / V row is visible
  }
  float Kmax2 = 0.f;   // over all 1024 key rows of the group (a superset of any sample's keys: the bound only gets looser)
  if (bounded) {
#pragma unroll
    for (int w = 0; w < 8; ++w) Kmax2 = fmaxf(Kmax2, reinterpret_cast<const float*>(Wst + 2 * AB_WCHUNK)[w]);
  }

  if (ts) { ts[1] = wall_clock64(); ts[13] = __builtin_amdgcn_s_memtime(); }
  // ================= Phase B: attention over the LDS-resident K / V (attention_v2_kernel's LAZY loop, D = 32) =============
  constexpr int RB = 64, KT = 64, NDF = 2;
  bf16x8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (bf16_t)1.0f;
  const int tq = lr >> 2, tp = lr & 3;
  const int vkey = 4 * lg + tq;
  const int vsw = ab_vswz(vkey);
  const int vbase = vkey * RB + (tp & 1) * 8;
  const int ntiles = N / KT;
#pragma unroll
  for (int rnd = 0; rnd < 8 / QF; ++rnd) {
#if AB_PRIO
    // The two waves of a SIMD (w and w + 4) run this VALU-bound loop without a barrier between them, and the issue arbiter prefers the
    // OLDER one (MI355X_MICROARCH.md, two waves per SIMD): waves 0-3 finished after 34.5 us, waves 4-7 after 51.1, the SIMD half idle
    // in between.  Alternating the priority by round they end after 43.6 / 47.6 us (workgroup 70.9 -> 67.5 us).
    // The younger wave holds the priority for the first two thirds of its tiles (with the older one running at ~0.5 of its unimpeded
    // rate meanwhile, both end together when the switch sits at 1 / (1 + 0.5) of the work), the older one afterwards.
#endif
    const int qrow0 = wave * 128 + rnd * QF * 16;
    const int keybase = qrow0 & ~(N - 1);   // the query's own sample inside the group (N is a power of two)
    float m_run[QF];
    f32x4 o[QF][NDF], ol[QF];
#pragma unroll
    for (int qf = 0; qf < QF; ++qf) {
      m_run[qf] = 0.f;
      ol[qf] = f32x4{0, 0, 0, 0};
#pragma unroll
      for (int df = 0; df < NDF; ++df) o[qf][df] = f32x4{0, 0, 0, 0};
    }
    // per-lane LDS cursors, advanced by one 64-key tile per trip; everything else in the loop's addresses is an instruction offset
    const unsigned char* kp = Ks + (keybase + lr) * RB + ab_kswz(lr, lg) * 16;   // (the K swizzle only looks at row bits 2-3)
    const unsigned char* vp[NDF];
#pragma unroll
    for (int df = 0; df < NDF; ++df) vp[df] = Vs + keybase * RB + vbase + (((2 * df + (tp >> 1)) ^ vsw) << 4);
    // One 64-key tile.  TRACK = false is the BOUNDED form: when |q| max|k| - the Cauchy-Schwarz bound on every score of a query, from the
    // row norms the projection left behind - is within 2^100 of the reference maximum that the first tile set, no later score can
    // overflow exp2 or the fp32 accumulators (1024 keys x 2^100 x |v|), and any reference is exact (the denominator accumulates the same
    // p): the per-tile maximum, its compare / ballot and the rescale branch (46 of the tile's ~320 VALU instructions) are dropped.
    auto tile = [&](auto TRACK, int kt) {
#if AB_PRIO
      {
        const int prog = rnd * ntiles + kt, sw = (2 * (8 / QF) * ntiles + 1) / 3;
        if (prog == 0 || prog == sw) {
          if ((((wave >> 2) ^ (prog == 0 ? 0 : 1)) & 1) != 0) __builtin_amdgcn_s_setprio(1);
          else __builtin_amdgcn_s_setprio(0);
        }
      }
#endif
      f32x4 sc[QF][4];
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(kp + 16 * f * RB);
#pragma unroll
        for (int qf = 0; qf < QF; ++qf) {
          const float init = -m_run[qf];
          sc[qf][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qb[rnd * QF + qf], f32x4{init, init, init, init}, 0, 0, 0);
        }
      }
      if constexpr (decltype(TRACK)::value) {
        float lmax[QF];
        bool grow = kt == 0;
#pragma unroll
        for (int qf = 0; qf < QF; ++qf) {
          float t = sc[qf][0][0];
#pragma unroll
          for (int f = 0; f < 4; ++f)
#pragma unroll
            for (int j = 0; j < 4; ++j) t = fmaxf(t, sc[qf][f][j]);
          lmax[qf] = t;
          grow |= t > 6.0f;
        }
        if (__builtin_amdgcn_ballot_w64(grow) != 0) {   // move the reference of every query of the wave to its running maximum
#pragma unroll
          for (int qf = 0; qf < QF; ++qf) {
            float t = lmax[qf];
            t = fmaxf(t, __shfl_xor(t, 16));
            t = fmaxf(t, __shfl_xor(t, 32));
            t = fmaxf(t, kt == 0 ? -1e30f : 0.f);
            if (kt != 0) {
              const float alpha = __builtin_amdgcn_exp2f(-t);
#pragma unroll
              for (int df = 0; df < NDF; ++df) o[qf][df] *= alpha;
              ol[qf][0] *= alpha;
            }
            m_run[qf] += t;
#pragma unroll
            for (int f = 0; f < 4; ++f)
#pragma unroll
              for (int j = 0; j < 4; ++j) sc[qf][f][j] -= t;
          }
        }
      }
      bf16x8 pf[QF][2];
#pragma unroll
      for (int qf = 0; qf < QF; ++qf)
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
          for (int j = 0; j < 4; ++j) pf[qf][f >> 1][(f & 1) * 4 + j] = (bf16_t)__builtin_amdgcn_exp2f(sc[qf][f][j]);
#pragma unroll
      for (int u = 0; u < 2; ++u) {
#pragma unroll
        for (int qf = 0; qf < QF; ++qf) ol[qf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, pf[qf][u], ol[qf], 0, 0, 0);
#pragma unroll
        for (int df = 0; df < NDF; ++df) {
          typedef short s16x4 __attribute__((ext_vector_type(4)));
          typedef __attribute__((address_space(3))) s16x4* lds_s16x4;
          s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(vp[df] + (32 * u) * RB));
          s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(vp[df] + (32 * u + 16) * RB));
          typedef short s16x8 __attribute__((ext_vector_type(8)));
          s16x8 both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
          const bf16x8 vf = __builtin_bit_cast(bf16x8, both);   // whole-vector reinterpretation (k_attention.hip: the element-wise one miscompiles)
#pragma unroll
          for (int qf = 0; qf < QF; ++qf) o[qf][df] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[qf][u], o[qf][df], 0, 0, 0);
        }
      }
      kp += KT * RB;
      vp[0] += KT * RB;
      vp[1] += KT * RB;
    };
    tile(std::true_type{}, 0);   // the first tile sets the reference to its true maximum
    bool loose = !bounded;
    if (bounded) {
      const float bound = __builtin_sqrtf(qmax2 * Kmax2);
#pragma unroll
      for (int qf = 0; qf < QF; ++qf) loose |= bound - m_run[qf] > 100.0f;
    }
    if (__builtin_amdgcn_ballot_w64(loose) == 0) {
#pragma unroll 1
      for (int kt = 1; kt < ntiles; ++kt) tile(std::false_type{}, kt);
    } else {
#pragma unroll 1
      for (int kt = 1; kt < ntiles; ++kt) tile(std::true_type{}, kt);
    }
    // ---- output rows: LDS position 16 df + 4 lg + j of V holds channel 8 lg + 4 df + j (host permutation of the v rows), so the
    //      lane's two fragments are 8 consecutive channels = one 16-byte store -----------------------------------------------
#pragma unroll
    for (int qf = 0; qf < QF; ++qf) {
      const int row = row0 + qrow0 + qf * 16 + lr;
      if (row >= a.rows) continue;
      const float inv = 1.0f / ol[qf][0];
      bf16x8 t;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        t[j] = (bf16_t)(o[qf][0][j] * inv);
        t[4 + j] = (bf16_t)(o[qf][1][j] * inv);
      }
#if AB_OUT_SC1
      // sc1 store: written through and DROPPED from this XCD's L2 (MI355X_MICROARCH.md, store flavours) - the 64 KB a workgroup writes
      // would otherwise push the group's x rows, which the other heads still need, out of the 4 MiB L2
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, t), o_rsrc, (unsigned)(row * C * 2 + h * 64 + 16 * lg), 0, 16);
#else
      *reinterpret_cast<bf16x8*>((bf16_t*)a.out + (int64_t)row * C + h * 32 + 8 * lg) = t;
#endif
    }
  }
  if (a.ts) {   // diagnostics only: when each wave, and the slowest, is done (+ shader-clock stamps: cycles / 100 MHz ticks = the clock held)
    if (lane == 0) a.ts[16 * blockIdx.x + 4 + wave] = wall_clock64();
    if (ts) { ts[2] = wall_clock64(); ts[14] = __builtin_amdgcn_s_memtime(); }
    __syncthreads();
    if (ts) ts[3] = wall_clock64();
  }
}

// ---- host side --------------------------------------------------------------------------------------------------------
static inline uint16_t ab_bf16_rne(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7F800000u) == 0x7F800000u && (u & 0x007FFFFFu)) return (uint16_t)((u >> 16) | 0x40);   // NaN stays NaN
  return (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}

bool attnblock_shape_ok(int dt, int N, int C, int d, int max_batch) {
  if (dt != DT_BF16 || d != 32 || C % 64 != 0 || C < 64) return false;
  if (N != 64 && N != 256 && N != 1024) return false;
  const int64_t groups = cdiv64((int64_t)max_batch * N, AB_ROWS);
  return groups * AB_ROWS * C * 2 < ((int64_t)1 << 31);   // 32-bit buffer offsets over the padded row range
}

size_t attnblock_weight_bytes(int C, int heads) { return (size_t)heads * (C / 64) * AB_WCHUNK; }

// w [3C][C] / b [3C] in QKVAttentionLegacy order (per head [q | k | v] blocks of 32 rows, openaimodel.py:406-408), q rows already
// carrying softmax scale x log2 e.  wp: bf16 LDS images [head][chunk][96 rows][128 B]; bp: [head][96] in the images' row order.
void attnblock_pack_weights(const float* w, const float* b, int C, int heads, uint16_t* wp, float* bp) {
  const int NC = C / 64;
  for (int h = 0; h < heads; ++h)
    for (int r = 0; r < 96; ++r) {
      const int cf = r >> 4, rr = r & 15, blk = cf >> 1, half = cf & 1, l = rr >> 2, j = rr & 3;
      // fragment pair (2 blk, 2 blk + 1), lane group l: q / k - channels 8 l + 4 half + j (a lane's 8 values are consecutive channels);
      // v - LDS position 8 l + 4 half + j must hold the channel c with 16 df + 4 l' + j' -> 8 l' + 4 df + j' (see the output store)
      int ch;
      if (blk < 2) ch = 8 * l + 4 * half + j;
      else {
        const int pos = 8 * l + 4 * half + j, df = pos >> 4, l2 = (pos >> 2) & 3, j2 = pos & 3;
        ch = 8 * l2 + 4 * df + j2;
      }
      const int src = h * 96 + blk * 32 + ch;
      bp[h * 96 + r] = b[src];
      for (int c = 0; c < NC; ++c) {
        uint16_t* img = wp + ((size_t)(h * NC + c) * 96 + r) * 64;
        for (int s = 0; s < 8; ++s) {
          const int phys = s ^ ((r >> 1) & 7);
          for (int e = 0; e < 8; ++e) img[phys * 8 + e] = ab_bf16_rne(w[(size_t)src * C + 64 * c + 8 * s + e]);
        }
      }
    }
}

static unsigned long long* g_ab_ts = nullptr;
void attnblock_debug_ts(unsigned long long* buf) { g_ab_ts = buf; }

int launch_attnblock_qkv_attn(const void* x, const void* wp, const float* bp, void* out, int B, int N, int C, int heads, hipStream_t s) {
  DDMI_REQUIRE(x && wp && bp && out && B > 0, "attention block: bad arguments");
  DDMI_REQUIRE(attnblock_shape_ok(DT_BF16, N, C, 32, B) && heads * 32 == C, "attention block: shape not served by the fused kernel");
  AttnBlockK k{};
  k.x = x; k.w = wp; k.bias = bp; k.out = out;
  k.rows = B * N;
  k.x_bytes = (unsigned)((size_t)k.rows * C * 2);
  k.N = N; k.C = C; k.heads = heads;
  k.groups = cdiv(k.rows, AB_ROWS);
  k.xcd = k.groups >= 8 ? 1 : 0;
  static const int xcd_env = getenv("DDMI_ATTNBLOCK_XCD") ? atoi(getenv("DDMI_ATTNBLOCK_XCD")) : -1;   // diagnostics: 0 = plain order
  if (xcd_env == 0) k.xcd = 0;
  k.ts = g_ab_ts;
  const int gpad = k.xcd ? cdiv(k.groups, 8) * 8 : k.groups;
  constexpr int lds = 2 * AB_KV_BYTES + 2 * AB_WCHUNK + 64 + 384;   // K | V, two weight stages, eight key-norm maxima, the head's 96 biases
  static int qf_env = getenv("DDMI_ATTNBLOCK_QF") ? atoi(getenv("DDMI_ATTNBLOCK_QF")) : 4;   // 64 queries per wave and round: 63.0 vs 66.5 us per workgroup at 32x32
  auto k2 = attnblock_kernel<2>;
  auto k4 = attnblock_kernel<4>;
  static bool attr_set = false;
  if (!attr_set) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k2), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k4), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    attr_set = true;
  }
  const double T = (double)B * N;
  PROF_SCOPE("attnblock_qkv_attn", 2.0 * T * C * 3 * C + 4.0 * T * N * C, 2.0 * T * C * 2, s);
  hipLaunchKernelGGL(qf_env == 4 ? k4 : k2, dim3(gpad * heads), dim3(512), lds, s, k);
  HIP_TRY(hipGetLastError());
  return DDMI_OK;
}
