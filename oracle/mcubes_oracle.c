/* TEST INFRASTRUCTURE ONLY - the CPU oracle for the marching-cubes row; nothing in the
 * product path may link or call this file.  Plain-C restatement of the reference's libmcubes as it is reached from
 * convocc/src/conv_onet/generation.py:162 (`libmcubes.marching_cubes(occ_hat_padded, threshold)`):
 *
 *   mcubes.pyx:22-27           `float isovalue` (C float!) -> the CALLER of this oracle rounds the threshold to fp32 first
 *   pywrapper.cpp:89-103       lower = 0, upper = shape - 1, num = shape: unit cells
 *   pywrapper.cpp:78-87        volume functor (int,int,int) called with doubles i + 0.5 -> truncation gives i;
 *                              values widened to double (pyarraymodule.h:41-79)
 *   marchingcubes.h:21-237     the sweep: i outer, j, k inner; corner numbering v0..v7 (:57-61); cube index bit m set when
 *                              v[m] <= isovalue (:63-66); vertices are created in the order edge 6, 5, 10 (the three
 *                              edges a cell owns, :72-89) then 0,1,2,3,4,7,8,9,11 only on the low faces of the grid
 *                              (:91-217), every other edge re-uses the index its owner stored; triangle corners are
 *                              appended in triangle_table order (:219-223)
 *   marchingcubes.cpp:290-326  vertex = corner a of the edge, the edge's axis coordinate replaced by
 *                              (x2-x1)*(iso-f1)/(f2-f1) + x1, or the midpoint when f1 == f2 - all in double
 *
 * Pinned bit-for-bit against the reference itself (oracle/_ref/libmcubes_ref.so, built from the reference's sources by
 * oracle/Makefile) and against tests/golden/mcubes_*.npz generated from it (oracle/gen_golden_mcubes.py).
 *
 * The case table is the classic Lorensen-Cline / Bourke triangulation (identical to marchingcubes.cpp:29-287, which a
 * not-gpu test verifies through the outputs on volumes that hit all 256 cases), stored one 64-bit word per case:
 * nibble n = edge of the n-th triangle corner, top nibble = number of corners.  The 12-bit edge mask is computed from
 * the corner bits (edge e joins corners EDGE_A[e], EDGE_B[e]; it is cut iff they differ), equal to marchingcubes.cpp:9-27.
 */
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static const uint64_t TRI_CASES[256] = {
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

static const int EDGE_A[12] = {0, 1, 2, 3, 4, 5, 6, 7, 0, 1, 2, 3};
static const int EDGE_B[12] = {1, 2, 3, 0, 5, 6, 7, 4, 4, 5, 6, 7};
/* corner offsets (di, dj, dk), marchingcubes.h:57-61 */
static const int CORNER[8][3] = {{0, 0, 0}, {1, 0, 0}, {1, 1, 0}, {0, 1, 0}, {0, 0, 1}, {1, 0, 1}, {1, 1, 1}, {0, 1, 1}};
/* the order in which a cell creates vertices, marchingcubes.h:72-217 */
static const int CREATE_ORDER[12] = {6, 5, 10, 0, 1, 2, 3, 4, 7, 8, 9, 11};
/* for the edges a cell does not own: the owner is the cell at (i,j,k) - OWNER_BACK[e] and the vertex sits in the owner's
 * slot OWNER_SLOT[e] (0 = its edge 6, 1 = its edge 5, 2 = its edge 10).  Without a neighbour on that side (index -1)
 * the cell creates the vertex itself. */
static const int OWNER_BACK[12][3] = {{0, 1, 1}, {0, 0, 1}, {0, 0, 1}, {1, 0, 1}, {0, 1, 0}, {0, 0, 0},
                                      {0, 0, 0}, {1, 0, 0}, {1, 1, 0}, {0, 1, 0}, {0, 0, 0}, {1, 0, 0}};
static const int OWNER_SLOT[12] = {0, 1, 0, 1, 0, 1, 0, 1, 2, 2, 2, 2};

static unsigned edge_mask(unsigned cube) {
  unsigned m = 0;
  for (int e = 0; e < 12; ++e)
    if (((cube >> EDGE_A[e]) & 1u) != ((cube >> EDGE_B[e]) & 1u)) m |= 1u << e;
  return m;
}

typedef struct { double* p; size_t n, cap; } DVec;
typedef struct { int64_t* p; size_t n, cap; } IVec;
static int dpush(DVec* v, double x) {
  if (v->n == v->cap) {
    size_t c = v->cap ? 2 * v->cap : 1024;
    double* q = (double*)realloc(v->p, c * sizeof(double));
    if (!q) return 1;
    v->p = q; v->cap = c;
  }
  v->p[v->n++] = x;
  return 0;
}
static int ipush(IVec* v, int64_t x) {
  if (v->n == v->cap) {
    size_t c = v->cap ? 2 * v->cap : 1024;
    int64_t* q = (int64_t*)realloc(v->p, c * sizeof(int64_t));
    if (!q) return 1;
    v->p = q; v->cap = c;
  }
  v->p[v->n++] = x;
  return 0;
}

static double value_at(const void* vol, int is_f64, long ny, long nz, long i, long j, long k) {
  size_t at = ((size_t)i * ny + j) * nz + k;
  return is_f64 ? ((const double*)vol)[at] : (double)((const float*)vol)[at];
}

/* marchingcubes.cpp:290-297 */
static double cut(double iso, double f1, double f2, double x1, double x2) {
  if (f2 == f1) return (x2 + x1) / 2;
  return (x2 - x1) * (iso - f1) / (f2 - f1) + x1;
}

/* volume: C-contiguous [nx][ny][nz], float32 or float64.  Outputs malloc'ed (free with mcubes_oracle_free):
 * vertices = n_vertex_doubles/3 rows of (x,y,z) doubles, indices = n_indices/3 triangles. */
int mcubes_oracle_run(const void* volume, int is_f64, long nx, long ny, long nz, double isovalue, double** vertices,
                      size_t* n_vertex_doubles, int64_t** indices, size_t* n_indices) {
  const long cx = nx - 1, cy = ny - 1, cz = nz - 1;
  DVec V = {0, 0, 0};
  IVec P = {0, 0, 0};
  int64_t* slot = NULL; /* [cell][3]: index of the vertex on the owner's edge 6 / 5 / 10 */
  if (cx > 0 && cy > 0 && cz > 0) {
    slot = (int64_t*)malloc(sizeof(int64_t) * 3 * (size_t)cx * cy * cz);
    if (!slot) return 1;
  }
  for (long i = 0; i < cx; ++i)
    for (long j = 0; j < cy; ++j)
      for (long k = 0; k < cz; ++k) {
        double v[8], pos[8][3];
        unsigned cube = 0;
        for (int m = 0; m < 8; ++m) {
          long ci = i + CORNER[m][0], cj = j + CORNER[m][1], ck = k + CORNER[m][2];
          v[m] = value_at(volume, is_f64, ny, nz, ci, cj, ck);
          pos[m][0] = (double)ci + 0.5; /* lower + d*i + d/2 with d = 1 (marchingcubes.h:44-55) */
          pos[m][1] = (double)cj + 0.5;
          pos[m][2] = (double)ck + 0.5;
          if (v[m] <= isovalue) cube |= 1u << m;
        }
        const unsigned edges = edge_mask(cube);
        int64_t idx[12];
        const size_t cell = ((size_t)i * cy + j) * cz + k;
        for (int o = 0; o < 12; ++o) {
          const int e = CREATE_ORDER[o];
          if (!(edges & (1u << e))) continue;
          const long oi = i - OWNER_BACK[e][0], oj = j - OWNER_BACK[e][1], ok = k - OWNER_BACK[e][2];
          const int own = (oi == i && oj == j && ok == k);
          if (!own && oi >= 0 && oj >= 0 && ok >= 0) {
            idx[e] = slot[3 * ((((size_t)oi * cy) + oj) * cz + ok) + OWNER_SLOT[e]];
            continue;
          }
          const int a = EDGE_A[e], b = EDGE_B[e];
          const int axis = e >= 8 ? 2 : (e & 1);
          idx[e] = (int64_t)(V.n / 3);
          if (own) slot[3 * cell + OWNER_SLOT[e]] = idx[e];
          double p[3] = {pos[a][0], pos[a][1], pos[a][2]};
          p[axis] = cut(isovalue, v[a], v[b], pos[a][axis], pos[b][axis]);
          if (dpush(&V, p[0]) || dpush(&V, p[1]) || dpush(&V, p[2])) return 1;
        }
        const uint64_t tc = TRI_CASES[cube];
        const int n = (int)(tc >> 60);
        for (int m = 0; m < n; ++m)
          if (ipush(&P, idx[(tc >> (4 * m)) & 15])) return 1;
      }
  free(slot);
  *vertices = V.p ? V.p : (double*)malloc(sizeof(double));
  *indices = P.p ? P.p : (int64_t*)malloc(sizeof(int64_t));
  *n_vertex_doubles = V.n;
  *n_indices = P.n;
  return 0;
}

void mcubes_oracle_free(void* p) { free(p); }
