// micro-benchmark: sustained v_mfma_f32_32x32x2_f32 rate (a) registers only, (b) with one ds_read_b32 per MFMA
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <bool LDS, bool RANDOM>
__global__ __launch_bounds__(256, 2) void k(float* out, int iters) {
  __shared__ float lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) { unsigned h = (i * 2654435761u) ^ (blockIdx.x * 40503u); h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15; lds[i] = RANDOM ? ((int)(h & 0xffffff) - 0x800000) * (1.0f / 8388608.f) : 0.001f * (i & 63); }
  __syncthreads();
  f32x16 acc[4];
  for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
  float a = 1.0f + threadIdx.x * 1e-3f, b = 0.5f;
  int off = threadIdx.x & 63;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      float a0 = a, a1 = a, b0 = b, b1 = b;
      if (LDS) {
        a0 = lds[off + 64 * ((u * 4) & 31)];
        a1 = lds[off + 64 * ((u * 4 + 1) & 31)];
        b0 = lds[off + 64 * ((u * 4 + 2) & 31) + 2048];
        b1 = lds[off + 64 * ((u * 4 + 3) & 31) + 2048];
      }
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[3], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) s += acc[j][i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
  float* out;
  (void)hipMalloc(&out, 4096 * 256 * sizeof(float));
  const int iters = 60000;
  for (int lds = 0; lds < 3; ++lds)
    for (int grid : {2048}) {
      hipEvent_t e0, e1;
      (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
      for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(e0);
        if (lds == 1) hipLaunchKernelGGL((k<true, false>), dim3(grid), dim3(256), 0, 0, out, iters);
        else if (lds == 2) hipLaunchKernelGGL((k<true, true>), dim3(grid), dim3(256), 0, 0, out, iters);
        else hipLaunchKernelGGL((k<false, false>), dim3(grid), dim3(256), 0, 0, out, iters);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
      }
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      double flops = (double)grid * 4 * iters * 64 * 4096.0;
      printf("lds=%d grid=%d: %.3f ms  %.1f TFLOP/s\n", lds, grid, ms, flops / ms / 1e9);
    }
  return 0;
}
