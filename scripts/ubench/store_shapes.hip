// How fast do the store shapes of the narrow fp32 kernels drain?  1 GiB written once per kernel, hipEvent-timed:
//   b128_stream   16 B per lane, 1 KiB contiguous per wave-instruction
//   b32_seg64     dword per lane; a wave-instruction writes four 64-byte segments 256 B apart (16 channels of the pixels
//                 4 kq + i, kq = 0..3), four instructions (i = 0..3) fill a 1-KiB block of 16 pixels x 16 channels:
//                 the epilogue of conv3x3_f32_narrow_kernel<*, 1, ...> (Cout = 16)
//   b32_seg128    the same with 32 channels (two instructions per 128-byte pixel row): Cout = 32
// and the matching dword LOAD shape (the fused BatchNorm-backward sums read y like that).
// Build: hipcc --offload-arch=gfx950 -O3 scripts/ubench/store_shapes.hip -o scripts/ubench/store_shapes
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void b128_stream(f32x4* __restrict__ p, size_t n16) {
  const f32x4 v = {1.f, 2.f, 3.f, 4.f};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) p[i] = v;
}

template <int C>   // channels per pixel: 16 or 32
__global__ __launch_bounds__(256) void b32_seg(float* __restrict__ p, size_t npix) {
  const int lane = threadIdx.x & 63, m = lane & 15, kq = lane >> 4;
  const size_t wave = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = ((size_t)gridDim.x * 256) >> 6;
  for (size_t blk = wave; (blk + 1) * 16 <= npix; blk += nwaves) {
    float* base = p + (blk * 16 + 4 * kq) * C + m;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int nb = 0; nb < C / 16; ++nb) base[i * C + 16 * nb] = 1.f;
  }
}

template <int C>
__global__ __launch_bounds__(256) void l32_seg(const float* __restrict__ p, size_t npix, float* sink) {
  const int lane = threadIdx.x & 63, m = lane & 15, kq = lane >> 4;
  const size_t wave = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = ((size_t)gridDim.x * 256) >> 6;
  float acc = 0.f;
  for (size_t blk = wave; (blk + 1) * 16 <= npix; blk += nwaves) {
    const float* base = p + (blk * 16 + 4 * kq) * C + m;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int nb = 0; nb < C / 16; ++nb) acc += base[i * C + 16 * nb];
  }
  if (acc == 12345.678f) sink[0] = acc;
}

__global__ __launch_bounds__(256) void l128_stream(const f32x4* __restrict__ p, size_t n16, float* sink) {
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) acc += p[i];
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) sink[0] = acc[0];
}

int main() {
  const size_t bytes = (size_t)1 << 30;
  float *a = nullptr, *sink = nullptr;
  CHECK(hipMalloc(&a, bytes));
  CHECK(hipMalloc(&sink, 256));
  CHECK(hipMemset(a, 0, bytes));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const dim3 g(256 * 8), b(256);
  auto run = [&](const char* name, auto launch) {
    for (int w = 0; w < 2; ++w) launch();
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < 5; ++r) launch();
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-14s %8.1f GB/s\n", name, bytes * 5.0 / (ms * 1e-3) / 1e9);
  };
  run("b128_stream", [&] { hipLaunchKernelGGL(b128_stream, g, b, 0, 0, (f32x4*)a, bytes / 16); });
  run("b32_seg64", [&] { hipLaunchKernelGGL((b32_seg<16>), g, b, 0, 0, a, bytes / 64); });
  run("b32_seg128", [&] { hipLaunchKernelGGL((b32_seg<32>), g, b, 0, 0, a, bytes / 128); });
  run("l128_stream", [&] { hipLaunchKernelGGL(l128_stream, g, b, 0, 0, (const f32x4*)a, bytes / 16, sink); });
  run("l32_seg64", [&] { hipLaunchKernelGGL((l32_seg<16>), g, b, 0, 0, (const float*)a, bytes / 64, sink); });
  run("l32_seg128", [&] { hipLaunchKernelGGL((l32_seg<32>), g, b, 0, 0, (const float*)a, bytes / 128, sink); });
  CHECK(hipFree(a));
  CHECK(hipFree(sink));
  return 0;
}
