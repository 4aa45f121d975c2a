// Shared host/device helpers for libdeadtrees_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/deadtrees_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

void dt_set_error(const char* fmt, ...);

#define DT_REQUIRE(cond, ...)            \
  do {                                   \
    if (!(cond)) {                       \
      dt_set_error(__VA_ARGS__);         \
      return DT_EINVAL;                  \
    }                                    \
  } while (0)

#define DT_LAUNCH_CHECK()                                                        \
  do {                                                                           \
    hipError_t e__ = hipGetLastError();                                          \
    if (e__ != hipSuccess) {                                                     \
      dt_set_error("%s:%d launch failed: %s", __FILE__, __LINE__, hipGetErrorString(e__)); \
      return DT_EHIP;                                                            \
    }                                                                            \
  } while (0)

static inline int dt_cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// wave64 butterfly sum
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ---- shared two-stage column reduction (elementwise.hip): in [planes][P][N] -> out [planes][ceil(P/RB)][N]
// rows_per_block RB; N % 4 == 0; fp32 partial sums in a fixed order (deterministic).
int dt_reduce_rows_launch(const float* in, float* out, int planes, int P, int N, int RB, hipStream_t st);
static inline int dt_reduce_rows_out(int P, int RB) { return (P + RB - 1) / RB; }

// XCD-aware workgroup id remap (bijective for any grid size): hardware deals consecutive workgroup ids
// round-robin over the 8 XCDs, so ids b and b+8 share an L2.  Returns a logical id such that logical
// neighbours (which share input tiles / weights) run on the same XCD.  Speed only, never correctness.
__device__ __forceinline__ unsigned xcd_remap(unsigned id, unsigned n) {
  const unsigned q = n >> 3, r = n & 7u, xcd = id & 7u, idx = id >> 3;
  const unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

// ---- 16-channel-granular kernels (conv_narrow.hip)
extern "C" int dt_conv2d_n16_supported(const dt_conv_desc* d);
int dt_conv2d_n16_launch(const dt_conv_desc* d, const float* src0, const float* w, float* out, float* stats,
                         const float* in_scale, const float* in_shift, hipStream_t st,
                         const dt_bn_bwd_fuse* fuse = nullptr);
extern "C" int dt_conv2d_wgrad_n16_supported(const dt_conv_desc* d);
int dt_wgrad_n16_cfg(const dt_conv_desc* d, int* ksplit, int* parts);
int dt_wgrad_n16_launch(const dt_conv_desc* d, const float* src0, const float* dy, float* ws, const float* in_scale,
                        const float* in_shift, hipStream_t st);

// ---- lean persistent kernel of the narrow full-resolution decoder layers (conv_narrow.hip, round 3): Cin, Cout in {16, 32}
extern "C" int dt_conv2d_narrow_supported(const dt_conv_desc* d);
int dt_conv2d_narrow_rows(const dt_conv_desc* d);
int dt_conv2d_narrow_subpixel(const dt_conv_desc* d);
extern "C" int dt_conv2d_narrow_affine(const dt_conv_desc* d, const float* src0, const float* w_hwio, float* out, const float* scale,
                                       const float* shift, const float* in_scale, const float* in_shift, void* stream);
int dt_conv2d_narrow_launch(const dt_conv_desc* d, const float* src0, const float* w, float* out, float* stats,
                            const float* in_scale, const float* in_shift, hipStream_t st, const dt_bn_bwd_fuse* fuse,
                            bool affine = false);

int dt_bn_bwd_finish_sums(float* red, int P, int C, float* dgamma, float* dbeta, hipStream_t st);
