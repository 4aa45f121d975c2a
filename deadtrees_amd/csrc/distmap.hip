// Signed Euclidean distance maps for the boundary loss, on the device.
// Replaces the loader-side scipy pass of the reference: data/deadtreedata.py:182-185 calls
// loss/losses.py:159-178 `one_hot2dist(class2one_hot(mask), resolution=[1,1])`, i.e. per class k
//     res[k] = edt(~pos)*~pos - (edt(pos) - 1)*pos      (pos = mask == k; all zero when the class is absent)
// stored into an int32 array (truncation toward zero) and then cast to float32.
// Squared distances on a unit grid are integers, so the transform is computed exactly in int32 and the result
// floor(sqrt(d2)) / 1 - floor(sqrt(d2)) is bit-identical to the reference's.
//
// Two passes of the separable exact EDT:
//   1. columns: per pixel the vertical distance to the nearest pixel inside (gP) and outside (gN) the class,
//   2. rows:    d2(y,x) = min_x' (x-x')^2 + g(y,x')^2 over the OPPOSITE set, searched outward from x and
//               cut off as soon as (x-x')^2 can no longer win.
#include "common.h"

namespace {

constexpr int INF16 = 0xFFFF;
constexpr int INF2 = 1 << 29;

// one thread per column; down scan writes, up scan folds the other direction in.
__global__ __launch_bounds__(64) void edt_columns_kernel(const int64_t* __restrict__ labels, uint16_t* __restrict__ gP,
                                                         uint16_t* __restrict__ gN, int32_t* __restrict__ flags,
                                                         int32_t* __restrict__ err, int K, int H, int W) {
  const int x = blockIdx.x * 64 + threadIdx.x;
  const int bk = blockIdx.y, b = bk / K, k = bk % K;
  if (x >= W) return;
  const int64_t* lab = labels + (int64_t)b * H * W + x;
  uint16_t* p = gP + (int64_t)bk * H * W + x;
  uint16_t* n = gN + (int64_t)bk * H * W + x;
  int lastP = -1, lastN = -1;
  bool bad = false;
#pragma unroll 8
  for (int y = 0; y < H; ++y) {
    const int64_t l = lab[(int64_t)y * W];
    bad |= (l < 0 || l >= K);
    if (l == k) lastP = y; else lastN = y;
    p[(int64_t)y * W] = (uint16_t)(lastP < 0 ? INF16 : y - lastP);
    n[(int64_t)y * W] = (uint16_t)(lastN < 0 ? INF16 : y - lastN);
  }
  if (lastP >= 0) atomicOr(&flags[bk * 2 + 0], 1);
  if (lastN >= 0) atomicOr(&flags[bk * 2 + 1], 1);
  if (bad && k == 0) atomicOr(err, 1);
  int nextP = -1, nextN = -1;
#pragma unroll 8
  for (int y = H - 1; y >= 0; --y) {
    const int dp = p[(int64_t)y * W], dn = n[(int64_t)y * W];
    if (dp == 0) nextP = y;
    if (dn == 0) nextN = y;
    const int up = nextP < 0 ? INF16 : nextP - y, un = nextN < 0 ? INF16 : nextN - y;
    p[(int64_t)y * W] = (uint16_t)min(dp, up);
    n[(int64_t)y * W] = (uint16_t)min(dn, un);
  }
}

__device__ __forceinline__ int isqrt_floor(int v) {
  int r = (int)sqrtf((float)v);
  while ((int64_t)r * r > v) --r;
  while ((int64_t)(r + 1) * (r + 1) <= v) ++r;
  return r;
}

// one workgroup per image row; both squared column-distance rows staged in LDS.
__global__ __launch_bounds__(256) void edt_rows_kernel(const uint16_t* __restrict__ gP, const uint16_t* __restrict__ gN,
                                                       const int32_t* __restrict__ flags, float* __restrict__ dist,
                                                       int H, int W) {
  extern __shared__ int lds[];
  int* sP = lds;
  int* sN = lds + W;
  const int y = blockIdx.x, bk = blockIdx.y;
  const int64_t row = ((int64_t)bk * H + y) * W;
  const bool anyP = flags[bk * 2 + 0] != 0, anyN = flags[bk * 2 + 1] != 0;
  float* out = dist + row;
  if (!anyP) {  // class absent from the tile: the reference leaves the plane blank
    for (int x = threadIdx.x; x < W; x += 256) out[x] = 0.f;
    return;
  }
  if (!anyN) {
    // class covers the whole tile: scipy's distance_transform_edt has no background to measure to and
    // returns the distance to a phantom pixel at (row -1, col 0); the reference inherits that
    // (checked against scipy for several shapes, tests/test_distmap_gpu.py).
    for (int x = threadIdx.x; x < W; x += 256) out[x] = (float)(1 - isqrt_floor((y + 1) * (y + 1) + x * x));
    return;
  }
  for (int x = threadIdx.x; x < W; x += 256) {
    const int a = gP[row + x], c = gN[row + x];
    sP[x] = a == INF16 ? INF2 : a * a;
    sN[x] = c == INF16 ? INF2 : c * c;
  }
  __syncthreads();
  for (int x = threadIdx.x; x < W; x += 256) {
    const bool pos = sP[x] == 0;
    const int* s = pos ? sN : sP;  // distance to the opposite set
    int best = s[x];
    for (int d = 1; d < W; ++d) {
      const int dd = d * d;
      if (dd >= best) break;
      const int xl = x - d, xr = x + d;
      if (xl < 0 && xr >= W) break;
      if (xl >= 0) best = min(best, dd + s[xl]);
      if (xr < W) best = min(best, dd + s[xr]);
    }
    const int r = isqrt_floor(best);
    out[x] = (float)(pos ? 1 - r : r);
  }
}

}  // namespace

extern "C" int64_t dt_signed_distmap_workspace(int B, int K, int H, int W) {
  const int64_t planes = (int64_t)B * K * H * W * 2 * (int64_t)sizeof(uint16_t);
  return planes + (int64_t)B * K * 2 * (int64_t)sizeof(int32_t);
}

extern "C" int dt_signed_distmap(const int64_t* labels, float* dist, void* workspace, int32_t* err_flag, int B, int K,
                                 int H, int W, void* stream) {
  DT_REQUIRE(labels && dist && workspace && err_flag, "dt_signed_distmap: null pointer");
  DT_REQUIRE(B > 0 && K > 0 && H > 0 && W > 0, "dt_signed_distmap: bad shape %dx%dx%dx%d", B, K, H, W);
  DT_REQUIRE(H <= 16384 && W <= 16384, "dt_signed_distmap: H and W must be <= 16384");
  DT_REQUIRE((int64_t)B * K <= 65535, "dt_signed_distmap: B*K must be <= 65535");
  hipStream_t st = (hipStream_t)stream;
  const int64_t plane = (int64_t)B * K * H * W;
  uint16_t* gP = (uint16_t*)workspace;
  uint16_t* gN = gP + plane;
  int32_t* flags = (int32_t*)(gN + plane);
  if (hipMemsetAsync(flags, 0, (size_t)B * K * 2 * sizeof(int32_t), st) != hipSuccess) {
    dt_set_error("dt_signed_distmap: memset failed");
    return DT_EHIP;
  }
  edt_columns_kernel<<<dim3(dt_cdiv(W, 64), B * K), 64, 0, st>>>(labels, gP, gN, flags, err_flag, K, H, W);
  DT_LAUNCH_CHECK();
  edt_rows_kernel<<<dim3(H, B * K), 256, (size_t)2 * W * sizeof(int), st>>>(gP, gN, flags, dist, H, W);
  DT_LAUNCH_CHECK();
  return DT_OK;
}
