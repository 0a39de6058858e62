// TEST INFRASTRUCTURE ONLY.  C-callable driver around the REFERENCE's own marching cubes, compiled from the sources where
// they lie (/root/reference/convocc/src/utils/libmcubes/marchingcubes.{h,cpp}) by oracle/Makefile into
// oracle/_ref/libmcubes_ref.so.  Nothing of the reference is copied here: this file only replaces the numpy glue of
// pywrapper.cpp (which needs Python/numpy C headers) with plain pointers.
//
// What it reproduces of the glue (pywrapper.cpp:89-103, 78-87):
//   * lower = (0,0,0), upper = shape-1, num = shape            -> unit cells, coordinates i + 0.5
//   * the volume functor takes (int,int,int): the template calls it with doubles (i + 0.5), which truncate to i
//   * values are widened to double (PyArray_SafeGet<double>, pyarraymodule.h:41-79)
#include <cstddef>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "marchingcubes.h"

namespace {
struct VolumeFunctor {
  const void* data;
  int is_f64;
  long ny, nz;
  double operator()(int x, int y, int z) const {
    size_t at = (static_cast<size_t>(x) * ny + y) * nz + z;
    return is_f64 ? static_cast<const double*>(data)[at] : static_cast<double>(static_cast<const float*>(data)[at]);
  }
};
}  // namespace

extern "C" {

// Runs mc::marching_cubes<double> (marchingcubes.h:21-237) on a C-contiguous [nx][ny][nz] volume.  The outputs are
// malloc'ed; release them with mcubes_ref_free.  n_vertex_doubles = 3 * vertices, n_indices = 3 * triangles.
int mcubes_ref_run(const void* volume, int is_f64, long nx, long ny, long nz, double isovalue, double** vertices,
                   size_t* n_vertex_doubles, size_t** indices, size_t* n_indices) {
  double lower[3] = {0, 0, 0};
  double upper[3] = {static_cast<double>(nx - 1), static_cast<double>(ny - 1), static_cast<double>(nz - 1)};
  std::vector<double> v;
  std::vector<size_t> p;
  VolumeFunctor f{volume, is_f64, ny, nz};
  mc::marching_cubes<double>(lower, upper, static_cast<int>(nx), static_cast<int>(ny), static_cast<int>(nz), f, isovalue, v, p);
  *n_vertex_doubles = v.size();
  *n_indices = p.size();
  *vertices = static_cast<double*>(std::malloc(sizeof(double) * (v.size() + 1)));
  *indices = static_cast<size_t*>(std::malloc(sizeof(size_t) * (p.size() + 1)));
  if (!*vertices || !*indices) return 1;
  if (!v.empty()) std::memcpy(*vertices, v.data(), sizeof(double) * v.size());
  if (!p.empty()) std::memcpy(*indices, p.data(), sizeof(size_t) * p.size());
  return 0;
}

void mcubes_ref_free(void* p) { std::free(p); }
}
