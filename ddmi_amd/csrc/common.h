// Shared declarations for the ddmi_hip library (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <string>

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// storage / operand precision of activations and GEMM operands
enum { DT_F32 = 0, DT_BF16 = 1 };

static inline size_t dt_size(int dt) { return dt == DT_F32 ? 4 : 2; }

// ---- error plumbing (no exceptions cross the C ABI; codes come from the public header) --------
#include "../../include/ddmi_hip.h"

void ddmi_set_error(const std::string& msg);
int ddmi_fail(int code, const std::string& msg);
const char* ddmi_error_cstr();

#define HIP_TRY(expr)                                                                              \
  do {                                                                                             \
    hipError_t _e = (expr);                                                                        \
    if (_e != hipSuccess)                                                                          \
      return ddmi_fail(DDMI_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));           \
  } while (0)

#define DDMI_TRY(expr)                                                                             \
  do {                                                                                             \
    int _c = (expr);                                                                               \
    if (_c != DDMI_OK) return _c;                                                                  \
  } while (0)

#define DDMI_REQUIRE(cond, msg)                                                                    \
  do {                                                                                             \
    if (!(cond)) return ddmi_fail(DDMI_ERR_INVALID, std::string(msg) + " [" #cond "]");            \
  } while (0)

#ifdef __HIPCC__
#define DDMI_HD __host__ __device__
#else
#define DDMI_HD
#endif
DDMI_HD static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
DDMI_HD static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---- device helpers ---------------------------------------------------------------------------
#ifdef __HIPCC__
template <typename T> struct VecIO;  // 8-element (bf16) / 4-element (f32) 16-byte vectors

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(bf16_t v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return (bf16_t)v; }

__device__ __forceinline__ float silu_f(float v) { return v / (1.0f + __expf(-v)); }
// bf16-storage paths: SiLU / sigmoid on the hardware exponential and the hardware RECIPROCAL (v_rcp_f32, 1 ulp).  `__frcp_rn` - what these
// paths used through round 4 - is the correctly ROUNDED reciprocal: v_rcp_f32 plus a v_div_scale / v_fma / v_div_fmas / v_div_fixup
// refinement, eleven more VALU instructions per element (found in the ISA of the GroupNorm epilogue in round 5: 17 VALU per element where
// 6 do the work; the result is rounded to bf16 two instructions later)
__device__ __forceinline__ float rcp_fast(float v) { return __builtin_amdgcn_rcpf(v); }
__device__ __forceinline__ float silu_fast(float v) { return v * rcp_fast(1.0f + __expf(-v)); }
// full-precision variant for the parity path (expf, not the fast intrinsic)
__device__ __forceinline__ float silu_precise(float v) { return v / (1.0f + expf(-v)); }
#endif
