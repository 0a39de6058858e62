// Post-path consumers, both HBM-bound byte/integer work:
//   * marching cubes on the device-resident occupancy logit grid - replaces libmcubes.marching_cubes as
//     Generator3D.extract_mesh calls it (convocc/src/conv_onet/generation.py:146-183), with results IDENTICAL to the
//     reference's sequential sweep: same vertex numbering, same triangle order, same doubles;
//   * clamp -> [0,1] -> uint8 HWC image write-out (evals/eval.py:226-233).
//
// The reference sweeps cells i-outer / k-inner and numbers a vertex when the sweep first meets its edge
// (marchingcubes.h:39-223).  That numbering is a prefix sum: a cell creates the vertices of the (cut) edges it owns -
// 6, 5, 10 - plus, on the low faces of the grid, the ones nobody before it owns, in the fixed order
// 6,5,10,0,1,2,3,4,7,8,9,11.  So: classify every cell (8-bit corner mask), scan the per-cell vertex / triangle-corner
// counts in linear cell order, then let every cut cell write its vertices and look the other edges up at their owners
// (owner's offset + rank of the slot among what the owner created).  No atomics, no ordering dependence: the output is a
// pure function of the grid.
//
// Passes (cells = (NX-1)(NY-1)(NZ-1), one thread per cell, 1024 cells per workgroup in 4 coalesced sweeps):
//   mc_classify   reads the grid (4 B/voxel, neighbours from L1/L2), writes 1 B/cell + one packed count per workgroup
//   mc_scan_wg    one workgroup: exclusive scan of the workgroup counts, totals
//   mc_offsets    1 B/cell in, 8 B/cell out (vertex offset, triangle-corner offset)
//   mc_emit       cut cells only: 8 grid values, neighbours' masks/offsets, writes 24 B/vertex + 8 B/triangle corner
// The np.pad(occ_hat, 1, constant_values=-1e6) of generation.py:160-161 is virtual (pad = 1): out-of-range corners read
// as pad_value, no padded copy is made.
#include "common.h"
#pragma clang fp contract(off)   // the vertex arithmetic must round exactly like the reference's x86 doubles

namespace {

// Lorensen-Cline / Bourke triangulation, one 64-bit word per corner mask: nibble n = edge of the n-th triangle corner,
// top nibble = number of corners (same cases, same order as marchingcubes.cpp:29-287 - tests/test_oracle_mcubes.py
// checks all 256 through the outputs)
__device__ const unsigned long long kTriCases[256] = {
  0x0000000000000000ull, 0x3000000000000380ull, 0x3000000000000910ull, 0x6000000000189381ull,
  0x3000000000000a21ull, 0x6000000000a21380ull, 0x6000000000920a29ull, 0x900000089a8a2382ull,
  0x30000000000002b3ull, 0x60000000000b82b0ull, 0x6000000000b32091ull, 0x9000000b89b912b1ull,
  0x60000000003ab1a3ull, 0x9000000ab8a801a0ull, 0x90000009ab9b3093ull, 0x6000000000b8aa89ull,
  0x3000000000000874ull, 0x6000000000437034ull, 0x6000000000748910ull, 0x9000000137174914ull,
  0x6000000000748a21ull, 0x9000000a21403743ull, 0x9000000748209a29ull, 0xc0004973727929a2ull,
  0x60000000002b3748ull, 0x900000040242b74bull, 0x9000000b32748109ull, 0xc0001292b9b49b74ull,
  0x9000000487ab31a3ull, 0xc0004b7401b41ab1ull, 0xc00030bab9b09874ull, 0x9000000ab99b4b74ull,
  0x3000000000000459ull, 0x6000000000380459ull, 0x6000000000051450ull, 0x9000000513538458ull,
  0x6000000000459a21ull, 0x9000000594a21803ull, 0x9000000204245a25ull, 0xc0008434535235a2ull,
  0x6000000000b32459ull, 0x9000000594b802b0ull, 0x9000000b32510450ull, 0xc000584b82852512ull,
  0x900000045931ab3aull, 0xc000ab81a8180594ull, 0xc00030bab5b05045ull, 0x9000000b8aa85845ull,
  0x6000000000975879ull, 0x9000000375359039ull, 0x9000000751710870ull, 0x6000000000753351ull,
  0x900000021a759879ull, 0xc00037503505921aull, 0xc00025a758528208ull, 0x90000007533525a2ull,
  0x90000002b3987597ull, 0xc000b72029279759ull, 0xc000751871810b32ull, 0x900000051771b12bull,
  0xc000b3a31a758859ull, 0xf0aba010b7905075ull, 0xf07570805a30b0abull, 0x60000000005b75abull,
  0x300000000000056aull, 0x60000000006a5380ull, 0x60000000006a5109ull, 0x90000006a5891381ull,
  0x6000000000162561ull, 0x9000000803621561ull, 0x9000000620609569ull, 0xc000823625285895ull,
  0x600000000056ab32ull, 0x900000056a02b80bull, 0x90000006a5b32910ull, 0xc000b892b92916a5ull,
  0x9000000315356b36ull, 0xc0006b51505b0b80ull, 0xc0009505606306b3ull, 0x900000089bb96956ull,
  0x60000000008746a5ull, 0x9000000a56374034ull, 0x90000007486a5091ull, 0xc00049737179156aull,
  0x9000000874156216ull, 0xc000743403625521ull, 0xc000620560509748ull, 0xf962695923497937ull,
  0x900000056a4872b3ull, 0xc000b720242746a5ull, 0xc0006a5b32874910ull, 0xf6a54b7b492b9129ull,
  0xc0006b51535b3748ull, 0xfb404b7b016b5b15ull, 0xf74836b630560950ull, 0xc0009b7974b96956ull,
  0x6000000000a4694aull, 0x9000000380a946a4ull, 0x900000004606a10aull, 0xc000a16468618138ull,
  0x9000000462421941ull, 0xc000462942921803ull, 0x6000000000624420ull, 0x9000000624428238ull,
  0x900000032b46a94aull, 0xc0006a4a94b82280ull, 0xc000a164606102b3ull, 0xf1b8b12184a16146ull,
  0xc00036b319639469ull, 0xf14641916b0181b8ull, 0x90000004600636b3ull, 0x600000000086b846ull,
  0x9000000a98a876a7ull, 0xc000a76a907a0370ull, 0xc0000818717a176aull, 0x900000037117a76aull,
  0xc000768981861621ull, 0xf937390976192962ull, 0x9000000206607087ull, 0x6000000000276237ull,
  0xc00076898a86ab32ull, 0xf7a9a76790b72702ull, 0xfb32a767a1871081ull, 0xc00017616a71b12bull,
  0xf63136b619768698ull, 0x600000000076b190ull, 0xc00006b0b3607087ull, 0x30000000000006b7ull,
  0x3000000000000b67ull, 0x600000000067b803ull, 0x600000000067b910ull, 0x900000067b138918ull,
  0x60000000007b621aull, 0x90000007b6803a21ull, 0x90000007b69a2092ull, 0xc00089a38a3a27b6ull,
  0x6000000000726327ull, 0x9000000026067807ull, 0x9000000910732672ull, 0xc000678891681261ull,
  0x900000073171a67aull, 0xc000801781a7167aull, 0xc0007a69a0a70730ull, 0x90000009a88a7a67ull,
  0x600000000068b486ull, 0x9000000640603b63ull, 0x9000000109648b68ull, 0xc00063b139369649ull,
  0x90000001a28b6486ull, 0xc000640b60b03a21ull, 0xc0009a2920b648b4ull, 0xf36463b34923a39aull,
  0x9000000264248328ull, 0x6000000000264240ull, 0xc000834642432091ull, 0x9000000642241491ull,
  0xc0001a6648168318ull, 0x900000040660a01aull, 0xf39a9303a6834364ull, 0x60000000004a649aull,
  0x6000000000b67594ull, 0x900000067b594380ull, 0x9000000b67045105ull, 0xc00051345343867bull,
  0x9000000b6721a459ull, 0xc000594380a217b6ull, 0xc000204a24a45b67ull, 0xf67b25a523453843ull,
  0x9000000945267327ull, 0xc000786260680459ull, 0xc000045051673263ull, 0xf851584812786826ull,
  0xc00073167161a459ull, 0xf459078701671a61ull, 0xfa737a6a305a4a04ull, 0xc000a84a458a7a67ull,
  0x900000098b9b6596ull, 0xc000590650360b63ull, 0xc000b65510b508b0ull, 0x90000001355363b6ull,
  0xc00065b8b9b59a21ull, 0xfa21965690b603b0ull, 0xf52025a50865b58bull, 0xc00035a3a25363b6ull,
  0xc000283265825985ull, 0x9000000260069659ull, 0xf826283865081851ull, 0x6000000000612651ull,
  0xf698965683a61631ull, 0xc00006505960a01aull, 0x6000000000a65830ull, 0x300000000000065aull,
  0x6000000000b57a5bull, 0x900000003857ba5bull, 0x9000000091ba57b5ull, 0xc0001381897ba57aull,
  0x900000015717b21bull, 0xc000b27571721380ull, 0xc0007b2209729579ull, 0xf289823295b27257ull,
  0x9000000573532a52ull, 0xc00052a578258028ull, 0xc0002a37353a5109ull, 0xf25752a278129289ull,
  0x6000000000573531ull, 0x9000000571170780ull, 0x9000000735539309ull, 0x6000000000795789ull,
  0x90000008ba8a5485ull, 0xc00003bba50b5405ull, 0xc00054aba8a48910ull, 0xf41314943b54a4baull,
  0xc0008548b2582152ull, 0xfb151b2b543b0b40ull, 0xf58b8545b2950520ull, 0x60000000003b2549ull,
  0xc000483543253a52ull, 0x90000000244252a5ull, 0xf910854583a532a3ull, 0xc0002492914252a5ull,
  0x9000000153358548ull, 0x6000000000501540ull, 0xc000530509358548ull, 0x3000000000000549ull,
  0x9000000ba9b947b4ull, 0xc000ba97b9794380ull, 0xc000b470414b1ba1ull, 0xf4bab474a1843413ull,
  0xc000219b294b97b4ull, 0xf3801b2b197b9479ull, 0x900000004224b47bull, 0xc00042343824b47bull,
  0xc000947732972a92ull, 0xf70207872a4797a9ull, 0xfa040a1a472a3a73ull, 0x60000000004782a1ull,
  0x9000000317714194ull, 0xc000178180714194ull, 0x6000000000347304ull, 0x3000000000000784ull,
  0x60000000008ba8a9ull, 0x9000000a9bb93903ull, 0x9000000ba88a0a10ull, 0x6000000000a3ba13ull,
  0x90000008b99b1b21ull, 0xc0009b2921b93903ull, 0x6000000000b08b20ull, 0x3000000000000b23ull,
  0x900000098aa82832ull, 0x60000000002902a9ull, 0xc0008a1810a82832ull, 0x30000000000002a1ull,
  0x6000000000819831ull, 0x3000000000000190ull, 0x3000000000000830ull, 0x0000000000000000ull,
};

struct McGrid {
  const void* vol;
  int f64;
  int nx, ny, nz;        // the array in memory
  int pad;               // 0 or 1 virtual border cells on every side
  float pad_value;
  int cx, cy, cz;        // cells per axis
  unsigned cells;
  double iso;
};

constexpr int MC_WG = 256, MC_SWEEPS = 4, MC_CELLS_PER_WG = MC_WG * MC_SWEEPS;

// edge e joins corners (e, e+1 mod 4) on the bottom face, the same on the top face, and (e-8, e-4) for the verticals:
// the 12-bit mask of cut edges straight from the corner bits (equals edge_table, marchingcubes.cpp:9-27)
__device__ __forceinline__ unsigned mc_edges(unsigned cube) {
  unsigned lo = cube & 15u, hi = cube >> 4;
  unsigned rlo = ((lo >> 1) | (lo << 3)) & 15u, rhi = ((hi >> 1) | (hi << 3)) & 15u;
  return (lo ^ rlo) | ((hi ^ rhi) << 4) | ((lo ^ hi) << 8);
}
// the edges a cell at (i,j,k) numbers itself (marchingcubes.h:72-217): its own three, plus what has no earlier owner
__device__ __forceinline__ unsigned mc_created_mask(int i, int j, int k) {
  return 0x460u | (k == 0 ? 0x00Fu : 0u) | (j == 0 ? 0x311u : 0u) | (i == 0 ? 0x988u : 0u);
}
__device__ __forceinline__ unsigned mc_counts(unsigned cube, int i, int j, int k) {   // vertices | triangle corners << 16
  unsigned nv = __popc(mc_edges(cube) & mc_created_mask(i, j, k));
  unsigned nt = (unsigned)(kTriCases[cube] >> 60);
  return nv | (nt << 16);
}

template <typename T, int PAD>
__device__ __forceinline__ double mc_value(const McGrid& g, int i, int j, int k) {
  if (PAD) {
    i -= 1; j -= 1; k -= 1;
    if ((unsigned)i >= (unsigned)g.nx || (unsigned)j >= (unsigned)g.ny || (unsigned)k >= (unsigned)g.nz)
      return (double)g.pad_value;
  }
  return (double)((const T*)g.vol)[((size_t)i * g.ny + j) * g.nz + k];
}

__device__ __forceinline__ void mc_cell_ijk(const McGrid& g, unsigned c, int* i, int* j, int* k) {
  unsigned r = c / (unsigned)g.cz;
  *k = (int)(c - r * (unsigned)g.cz);
  unsigned q = r / (unsigned)g.cy;
  *j = (int)(r - q * (unsigned)g.cy);
  *i = (int)q;
}

// exclusive scan over the 256 threads of a workgroup; returns the exclusive prefix, *total = the workgroup's sum
__device__ __forceinline__ unsigned mc_wg_scan(unsigned v, unsigned* total, unsigned* lds /* [4] */) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    unsigned o = __shfl_up(inc, d, 64);
    if (lane >= d) inc += o;
  }
  __syncthreads();   // lds reuse across calls
  if (lane == 63) lds[wave] = inc;
  __syncthreads();
  unsigned base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < MC_WG / 64; ++w) {
    unsigned s = lds[w];
    if (w < wave) base += s;
    tot += s;
  }
  *total = tot;
  return base + inc - v;
}

template <typename T, int PAD>
__global__ __launch_bounds__(MC_WG) void mc_classify(McGrid g, unsigned char* __restrict__ cube_out,
                                                      unsigned* __restrict__ wg_counts /* [nwg][2] */) {
  __shared__ unsigned lds[4];
  unsigned sum_v = 0, sum_t = 0;
#pragma unroll
  for (int s = 0; s < MC_SWEEPS; ++s) {
    unsigned c = blockIdx.x * MC_CELLS_PER_WG + s * MC_WG + threadIdx.x;
    unsigned cnt = 0;
    if (c < g.cells) {
      int i, j, k;
      mc_cell_ijk(g, c, &i, &j, &k);
      unsigned cube = 0;   // corner numbering marchingcubes.h:57-61; bit set when v <= isovalue (:63-66)
      cube |= (mc_value<T, PAD>(g, i, j, k) <= g.iso) ? 1u : 0u;
      cube |= (mc_value<T, PAD>(g, i + 1, j, k) <= g.iso) ? 2u : 0u;
      cube |= (mc_value<T, PAD>(g, i + 1, j + 1, k) <= g.iso) ? 4u : 0u;
      cube |= (mc_value<T, PAD>(g, i, j + 1, k) <= g.iso) ? 8u : 0u;
      cube |= (mc_value<T, PAD>(g, i, j, k + 1) <= g.iso) ? 16u : 0u;
      cube |= (mc_value<T, PAD>(g, i + 1, j, k + 1) <= g.iso) ? 32u : 0u;
      cube |= (mc_value<T, PAD>(g, i + 1, j + 1, k + 1) <= g.iso) ? 64u : 0u;
      cube |= (mc_value<T, PAD>(g, i, j + 1, k + 1) <= g.iso) ? 128u : 0u;
      cube_out[c] = (unsigned char)cube;
      cnt = mc_counts(cube, i, j, k);
    }
    unsigned tot;
    mc_wg_scan(cnt, &tot, lds);
    sum_v += tot & 0xFFFFu;
    sum_t += tot >> 16;
  }
  if (threadIdx.x == 0) {
    wg_counts[2 * blockIdx.x] = sum_v;
    wg_counts[2 * blockIdx.x + 1] = sum_t;
  }
}

// one workgroup: wg_counts[nwg][2] -> exclusive offsets in place, totals[0..1]
__global__ __launch_bounds__(MC_WG) void mc_scan_wg(unsigned* __restrict__ wg_counts, int nwg, unsigned long long* __restrict__ totals) {
  __shared__ unsigned lds[4];
  unsigned carry_v = 0, carry_t = 0;
  for (int b0 = 0; b0 < nwg; b0 += MC_WG) {
    int b = b0 + threadIdx.x;
    unsigned v = b < nwg ? wg_counts[2 * b] : 0u, t = b < nwg ? wg_counts[2 * b + 1] : 0u;
    unsigned tv, tt;
    unsigned ev = mc_wg_scan(v, &tv, lds);
    unsigned et = mc_wg_scan(t, &tt, lds);
    if (b < nwg) {
      wg_counts[2 * b] = carry_v + ev;
      wg_counts[2 * b + 1] = carry_t + et;
    }
    carry_v += tv;
    carry_t += tt;
  }
  if (threadIdx.x == 0) {
    totals[0] = carry_v;
    totals[1] = carry_t;
  }
}

__global__ __launch_bounds__(MC_WG) void mc_offsets(McGrid g, const unsigned char* __restrict__ cube_in,
                                                     const unsigned* __restrict__ wg_offsets, unsigned* __restrict__ voff,
                                                     unsigned* __restrict__ toff) {
  __shared__ unsigned lds[4];
  unsigned base_v = wg_offsets[2 * blockIdx.x], base_t = wg_offsets[2 * blockIdx.x + 1];
#pragma unroll
  for (int s = 0; s < MC_SWEEPS; ++s) {
    unsigned c = blockIdx.x * MC_CELLS_PER_WG + s * MC_WG + threadIdx.x;
    unsigned cnt = 0;
    if (c < g.cells) {
      int i, j, k;
      mc_cell_ijk(g, c, &i, &j, &k);
      cnt = mc_counts(cube_in[c], i, j, k);
    }
    unsigned tot;
    unsigned ex = mc_wg_scan(cnt, &tot, lds);
    if (c < g.cells) {
      voff[c] = base_v + (ex & 0xFFFFu);
      toff[c] = base_t + (ex >> 16);
    }
    base_v += tot & 0xFFFFu;
    base_t += tot >> 16;
  }
}

// marchingcubes.cpp:290-297
__device__ __forceinline__ double mc_cut(double iso, double f1, double f2, double x1, double x2) {
  if (f2 == f1) return (x2 + x1) / 2;
  return (x2 - x1) * (iso - f1) / (f2 - f1) + x1;
}

struct McXform {     // generation.py:165-183 (vol_bound None) fused into the vertex store, one rounded op at a time
  int on;
  double div[3];     // n_axis - 1 of the UNPADDED grid
  double box;        // 1 + padding
};
__device__ __forceinline__ double mc_xf(const McXform& x, double v, int axis) {
  if (!x.on) return v;
  v = v - 0.5;
  v = v - 1.0;
  v = v / x.div[axis];
  return x.box * (v - 0.5);
}

// creation order of marchingcubes.h:72-217 and, for the edges a cell does not own, where the owner sits and which of its
// three slots (its edge 6 / 5 / 10) holds the vertex
__device__ constexpr int kOrder[12] = {6, 5, 10, 0, 1, 2, 3, 4, 7, 8, 9, 11};
__device__ constexpr int kBack[12][3] = {{0, 1, 1}, {0, 0, 1}, {0, 0, 1}, {1, 0, 1}, {0, 1, 0}, {0, 0, 0},
                                         {0, 0, 0}, {1, 0, 0}, {1, 1, 0}, {0, 1, 0}, {0, 0, 0}, {1, 0, 0}};
__device__ constexpr int kSlot[12] = {0, 1, 0, 1, 0, 1, 0, 1, 2, 2, 2, 2};
__device__ constexpr int kCorner[8][3] = {{0, 0, 0}, {1, 0, 0}, {1, 1, 0}, {0, 1, 0}, {0, 0, 1}, {1, 0, 1}, {1, 1, 1}, {0, 1, 1}};

template <typename T, int PAD>
__global__ __launch_bounds__(MC_WG) void mc_emit(McGrid g, const unsigned char* __restrict__ cube_in,
                                                  const unsigned* __restrict__ voff, const unsigned* __restrict__ toff,
                                                  double* __restrict__ vertices, long long* __restrict__ triangles, McXform xf) {
  __shared__ unsigned idx_s[12][MC_WG];   // the cell's 12 edge -> vertex indices (dynamic lookup by the case table)
#pragma unroll 1
  for (int s = 0; s < MC_SWEEPS; ++s) {
    unsigned c = blockIdx.x * MC_CELLS_PER_WG + s * MC_WG + threadIdx.x;
    if (c >= g.cells) continue;
    const unsigned cube = cube_in[c];
    const unsigned edges = mc_edges(cube);
    if (!edges) continue;
    int i, j, k;
    mc_cell_ijk(g, c, &i, &j, &k);
    double v[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) v[m] = mc_value<T, PAD>(g, i + kCorner[m][0], j + kCorner[m][1], k + kCorner[m][2]);
    const unsigned created = edges & mc_created_mask(i, j, k);
    unsigned next = voff[c];
#pragma unroll
    for (int o = 0; o < 12; ++o) {
      const int e = kOrder[o];
      if (!(edges & (1u << e))) continue;
      unsigned id;
      if (created & (1u << e)) {
        id = next++;
        const int a = e < 8 ? e : e - 8, b = e < 4 ? (e + 1) & 3 : e < 8 ? 4 + ((e + 1) & 3) : e - 4;
        const int axis = e >= 8 ? 2 : (e & 1);
        double p[3] = {(double)(i + kCorner[a][0]) + 0.5, (double)(j + kCorner[a][1]) + 0.5, (double)(k + kCorner[a][2]) + 0.5};
        const int ib = (axis == 0 ? i : axis == 1 ? j : k) + kCorner[b][axis];
        p[axis] = mc_cut(g.iso, v[a], v[b], p[axis], (double)ib + 0.5);
        double* dst = vertices + 3 * (size_t)id;
        dst[0] = mc_xf(xf, p[0], 0);
        dst[1] = mc_xf(xf, p[1], 1);
        dst[2] = mc_xf(xf, p[2], 2);
      } else {
        const unsigned nc = c - (unsigned)((kBack[e][0] * g.cy + kBack[e][1]) * g.cz + kBack[e][2]);
        const unsigned ne = mc_edges(cube_in[nc]);
        const unsigned rank = kSlot[e] == 0 ? 0u : kSlot[e] == 1 ? ((ne >> 6) & 1u) : ((ne >> 6) & 1u) + ((ne >> 5) & 1u);
        id = voff[nc] + rank;
      }
      idx_s[e][threadIdx.x] = id;
    }
    const unsigned long long tc = kTriCases[cube];
    const int n = (int)(tc >> 60);
    long long* dst = triangles + toff[c];
    for (int m = 0; m < n; ++m) dst[m] = (long long)idx_s[(tc >> (4 * m)) & 15][threadIdx.x];
  }
}

// evals/eval.py:226-233: fake.clamp(-1,1); (fake+1)/2; numpy fp32 * 255 -> astype(uint8) (truncation), CHW -> HWC
__global__ __launch_bounds__(256) void image_to_u8_kernel(const float* __restrict__ img, unsigned char* __restrict__ out, int C,
                                                           int HW, long long total /* B*HW */) {
  long long p = (long long)blockIdx.x * 256 + threadIdx.x;
  if (p >= total) return;
  long long b = p / HW;
  int hw = (int)(p - b * HW);
  const float* src = img + b * C * (long long)HW + hw;
  unsigned char* dst = out + p * C;
  for (int ch = 0; ch < C; ++ch) {
    float x = src[(long long)ch * HW];
    x = fminf(fmaxf(x, -1.0f), 1.0f);
    float y = (x + 1.0f) / 2.0f;
    float z = y * 255.0f;
    dst[ch] = (unsigned char)(int)z;
  }
}

struct McLayout { size_t totals, wg, cube, voff, toff, bytes; int nwg; };
int mc_grid_of(const ddmi_mcubes_grid* in, McGrid* g, McLayout* l) {
  DDMI_REQUIRE(in, "mcubes: null grid");
  DDMI_REQUIRE(in->dtype == 0 || in->dtype == 1, "mcubes: dtype must be 0 (float32) or 1 (float64)");
  DDMI_REQUIRE(in->nx >= 1 && in->ny >= 1 && in->nz >= 1, "mcubes: empty volume");
  DDMI_REQUIRE(in->pad == 0 || in->pad == 1, "mcubes: pad must be 0 or 1");
  g->vol = in->volume; g->f64 = in->dtype; g->nx = in->nx; g->ny = in->ny; g->nz = in->nz;
  g->pad = in->pad; g->pad_value = in->pad_value; g->iso = in->isovalue;
  g->cx = in->nx + 2 * in->pad - 1; g->cy = in->ny + 2 * in->pad - 1; g->cz = in->nz + 2 * in->pad - 1;
  long long cells = (long long)g->cx * g->cy * g->cz;
  DDMI_REQUIRE(cells < (1ll << 28), "mcubes: more than 2^28 cells (offsets are 32-bit: 15 triangle corners per cell)");
  g->cells = (unsigned)cells;
  l->nwg = (int)((cells + MC_CELLS_PER_WG - 1) / MC_CELLS_PER_WG);
  auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
  l->totals = 0;
  l->wg = 256;
  l->cube = l->wg + up(sizeof(unsigned) * 2 * (size_t)(l->nwg > 0 ? l->nwg : 1));
  l->voff = l->cube + up((size_t)cells);
  l->toff = l->voff + up(sizeof(unsigned) * (size_t)cells);
  l->bytes = l->toff + up(sizeof(unsigned) * (size_t)cells);
  return DDMI_OK;
}

}  // namespace

extern "C" {

int ddmi_mcubes_workspace_bytes(const ddmi_mcubes_grid* grid, size_t* bytes) {
  McGrid g; McLayout l;
  DDMI_TRY(mc_grid_of(grid, &g, &l));
  DDMI_REQUIRE(bytes, "mcubes: null bytes");
  *bytes = l.bytes;
  return DDMI_OK;
}

int ddmi_mcubes_count(const ddmi_mcubes_grid* grid, void* workspace, int64_t* n_vertices, int64_t* n_triangles, void* stream) {
  McGrid g; McLayout l;
  DDMI_TRY(mc_grid_of(grid, &g, &l));
  DDMI_REQUIRE(n_vertices && n_triangles, "mcubes_count: null output");
  *n_vertices = 0; *n_triangles = 0;
  if (g.cells == 0) return DDMI_OK;   // a 1-thick volume has no cells (marchingcubes.h:39 never enters the loop)
  DDMI_REQUIRE(grid->volume && workspace, "mcubes_count: null pointer");
  hipStream_t s = (hipStream_t)stream;
  char* ws = (char*)workspace;
  unsigned char* cube = (unsigned char*)(ws + l.cube);
  unsigned* wg = (unsigned*)(ws + l.wg);
  auto* totals = (unsigned long long*)(ws + l.totals);
#define MC_LAUNCH(K, ...)                                                                    \
  do {                                                                                       \
    if (g.f64) { if (g.pad) K<double, 1><<<l.nwg, MC_WG, 0, s>>>(__VA_ARGS__); else K<double, 0><<<l.nwg, MC_WG, 0, s>>>(__VA_ARGS__); } \
    else       { if (g.pad) K<float, 1><<<l.nwg, MC_WG, 0, s>>>(__VA_ARGS__);  else K<float, 0><<<l.nwg, MC_WG, 0, s>>>(__VA_ARGS__); }  \
  } while (0)
  MC_LAUNCH(mc_classify, g, cube, wg);
  mc_scan_wg<<<1, MC_WG, 0, s>>>(wg, l.nwg, totals);
  mc_offsets<<<l.nwg, MC_WG, 0, s>>>(g, cube, wg, (unsigned*)(ws + l.voff), (unsigned*)(ws + l.toff));
  HIP_TRY(hipGetLastError());
  unsigned long long host[2];
  HIP_TRY(hipMemcpyAsync(host, totals, sizeof(host), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  *n_vertices = (int64_t)host[0];
  *n_triangles = (int64_t)(host[1] / 3);
  return DDMI_OK;
}

int ddmi_mcubes_emit(const ddmi_mcubes_grid* grid, const void* workspace, double* vertices, int64_t* triangles, int normalize,
                     double box_size, void* stream) {
  McGrid g; McLayout l;
  DDMI_TRY(mc_grid_of(grid, &g, &l));
  if (g.cells == 0) return DDMI_OK;
  DDMI_REQUIRE(grid->volume && workspace && vertices && triangles, "mcubes_emit: null pointer");
  hipStream_t s = (hipStream_t)stream;
  const char* ws = (const char*)workspace;
  McXform xf{};
  xf.on = normalize ? 1 : 0;
  // generation.py:155,177: n_x, n_y, n_z are the UNPADDED grid; with pad = 0 the caller handed the padded array
  const int shrink = g.pad ? 0 : 2;
  xf.div[0] = (double)(g.nx - shrink - 1); xf.div[1] = (double)(g.ny - shrink - 1); xf.div[2] = (double)(g.nz - shrink - 1);
  xf.box = box_size;
  const unsigned char* cube = (const unsigned char*)(ws + l.cube);
  const unsigned* voff = (const unsigned*)(ws + l.voff);
  const unsigned* toff = (const unsigned*)(ws + l.toff);
  MC_LAUNCH(mc_emit, g, cube, voff, toff, vertices, (long long*)triangles, xf);
#undef MC_LAUNCH
  HIP_TRY(hipGetLastError());
  return DDMI_OK;
}

int ddmi_image_to_u8(const float* img, uint8_t* out, int B, int C, int H, int W, void* stream) {
  DDMI_REQUIRE(img && out, "image_to_u8: null pointer");
  DDMI_REQUIRE(B >= 1 && C >= 1 && H >= 1 && W >= 1, "image_to_u8: empty image");
  long long total = (long long)B * H * W;
  image_to_u8_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(img, out, C, H * W, total);
  HIP_TRY(hipGetLastError());
  return DDMI_OK;
}
}
