// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access shapes the convolution kernels use
// ("1.73x cannot be interpreted because FETCH_SIZE is doubled for every kernel").
//
// MI355X_MICROARCH.md (section HBM) calibrates the two counters for ONE shape each: FETCH_SIZE reports exactly 1/2 of the
// bytes of a wide coalesced read (16 B per lane), WRITE_SIZE reads 16-byte-per-lane streaming stores exactly; "other
// access widths are uncalibrated: calibrate on a known byte count in your own access pattern".  Every kernel below moves
// a KNOWN number of bytes (each byte of a 1 GiB buffer exactly once: far beyond the 256 MiB Infinity Cache) in one of
// the shapes of conv_wino.hip / conv_fwd.hip / conv_bf16_dma.hip:
//   read_b128_stream     16 B per lane, contiguous                      (direct kernels' f32x4 loads; the documented x2 case)
//   read_b64_patch       8 B per lane, 16 segments of 32 B per wave-instruction at a 256-B pixel pitch, the 8 chunks of a
//                        pixel row read by successive instructions      (Winograd patch loads: buffer_load_dwordx2)
//   read_lds_dma_b128    global_load_lds_dwordx4, 1 KiB contiguous per wave-instruction (bf16 LDS-DMA kernels, Winograd weights)
//   write_b128_stream    16 B per lane, contiguous                      (bf16 kernels' staged stores; the documented exact case)
//   write_b32_stream     4 B per lane, 256 B contiguous per wave-instruction
//   write_b32_rows       4 B per lane, two 128-B rows per wave-instruction 8 pixels apart, the 8 rows of a 1-KiB block
//                        written by successive instructions             (Winograd epilogue: buffer_store_dword)
// Build:  hipcc --offload-arch=gfx950 -O3 scripts/ubench/pmc_calib.hip -o scripts/ubench/pmc_calib
// Run (two passes, --pmc only):  rocprofv3 --pmc FETCH_SIZE --output-format csv -d OUT/f -- scripts/ubench/pmc_calib
//                                rocprofv3 --pmc WRITE_SIZE --output-format csv -d OUT/w -- scripts/ubench/pmc_calib
// then scripts/pmc_calib_report.py OUT/f/..._counter_collection.csv OUT/w/..._counter_collection.csv -> profiles/pmc_calibration.json
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef const void __attribute__((address_space(1)))* gptr;
typedef void __attribute__((address_space(3)))* lptr;

#define CHECK(x)                                                                  \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));   \
      exit(1);                                                                    \
    }                                                                             \
  } while (0)

// every kernel: grid-stride over `n16` 16-byte units (or the shape's own block), a data-dependent store that never fires
// keeps the loads alive
__global__ __launch_bounds__(256) void read_b128_stream(const f32x4* __restrict__ p, size_t n16, float* sink) {
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) acc += p[i];
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) sink[0] = acc[0];
}

// 256-B pixel rows (64 fp32 channels); a wave owns 16 consecutive pixels per step: lane l -> pixel l >> 2, channel pair
// l & 3 of the 8-channel chunk; 8 successive instructions read the 8 chunks of the rows (each byte once)
__global__ __launch_bounds__(256) void read_b64_patch(const float* __restrict__ p, size_t npix, float* sink) {
  const int lane = threadIdx.x & 63;
  const size_t wave = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = ((size_t)gridDim.x * 256) >> 6;
  f32x2 acc = {0.f, 0.f};
  for (size_t p0 = wave * 16; p0 + 16 <= npix; p0 += nwaves * 16) {
    const float* row = p + (p0 + (lane >> 2)) * 64 + 2 * (lane & 3);
#pragma unroll
    for (int c = 0; c < 8; ++c) acc += *reinterpret_cast<const f32x2*>(row + 8 * c);
  }
  if (acc[0] + acc[1] == 12345.678f) sink[0] = acc[0];
}

__global__ __launch_bounds__(256) void read_lds_dma_b128(const f32x4* __restrict__ p, size_t n16, float* sink) {
  __shared__ __attribute__((aligned(1024))) f32x4 buf[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
    __builtin_amdgcn_global_load_lds((gptr)(p + i), (lptr)&buf[w][0], 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    acc += buf[w][lane];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) sink[0] = acc[0];
}

__global__ __launch_bounds__(256) void write_b128_stream(f32x4* __restrict__ p, size_t n16) {
  const f32x4 v = {1.f, 2.f, 3.f, 4.f};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) p[i] = v;
}

__global__ __launch_bounds__(256) void write_b32_stream(float* __restrict__ p, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) p[i] = 1.f;
}

// the Winograd epilogue's shape: a wave owns a block of 16 pixels x 32 channels inside 256-B pixel rows (64 channels):
// lanes 0..31 store the 32 channels (128 B) of pixel j, lanes 32..63 those of pixel j + 8; 8 successive instructions
// (j = 0..7) cover the block; two waves (channel halves) cover the 256-B rows.  Every byte is written exactly once.
__global__ __launch_bounds__(256) void write_b32_rows(float* __restrict__ p, size_t npix) {
  const int lane = threadIdx.x & 63;
  const size_t wave = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = ((size_t)gridDim.x * 256) >> 6;
  const int half = (int)(wave & 1);
  for (size_t blk = wave >> 1; (blk + 1) * 16 <= npix; blk += nwaves >> 1) {
    float* base = p + (blk * 16 + 8 * (lane >> 5)) * 64 + 32 * half + (lane & 31);
#pragma unroll
    for (int j = 0; j < 8; ++j) base[(size_t)j * 64] = 1.f;
  }
}

int main() {
  const size_t bytes = (size_t)1 << 30;   // 1 GiB: four times the Infinity Cache
  float *a = nullptr, *sink = nullptr;
  CHECK(hipMalloc(&a, bytes));
  CHECK(hipMalloc(&sink, 256));
  CHECK(hipMemset(a, 0, bytes));
  CHECK(hipDeviceSynchronize());
  const dim3 g(256 * 8), b(256);
  const size_t n16 = bytes / 16, n4 = bytes / 4, npix = bytes / 256;
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(read_b128_stream, g, b, 0, 0, (const f32x4*)a, n16, sink);
    hipLaunchKernelGGL(read_b64_patch, g, b, 0, 0, (const float*)a, npix, sink);
    hipLaunchKernelGGL(read_lds_dma_b128, g, b, 0, 0, (const f32x4*)a, n16, sink);
    hipLaunchKernelGGL(write_b128_stream, g, b, 0, 0, (f32x4*)a, n16);
    hipLaunchKernelGGL(write_b32_stream, g, b, 0, 0, a, n4);
    hipLaunchKernelGGL(write_b32_rows, g, b, 0, 0, a, npix);
    CHECK(hipDeviceSynchronize());
  }
  printf("{\"bytes_per_launch\": %zu, \"kernels\": [\"read_b128_stream\", \"read_b64_patch\", \"read_lds_dma_b128\", "
         "\"write_b128_stream\", \"write_b32_stream\", \"write_b32_rows\"]}\n", bytes);
  CHECK(hipFree(a));
  CHECK(hipFree(sink));
  return 0;
}
