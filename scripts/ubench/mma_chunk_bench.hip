// micro-benchmark of the conv kernels' inner MFMA loop (mma_chunk of conv_fwd.hip) on a resident LDS tile:
// isolates the LDS-read / MFMA issue schedule from staging, barriers and epilogue.
#include "../../deadtrees_amd/csrc/conv_fwd.hip"
#include <vector>
void dt_set_error(const char*, ...) {}
int dt_conv2d_n16_launch(const dt_conv_desc*, const float*, const float*, float*, float*, const float*, const float*, hipStream_t) { return 0; }
extern "C" int dt_conv2d_n16_supported(const dt_conv_desc*) { return 0; }

template <int VARIANT>
__global__ __launch_bounds__(256, 2) void bench_mma(float* out, int iters) {
  constexpr int KS = 3, TW = 32, TN = 64, CK = 16;
  using G = ConvGeom<KS, 1, TW, CK>;
  constexpr int IN_ELEMS = CK * G::PLANE, W_ELEMS = G::TAPS * CK * TN;
  __shared__ __attribute__((aligned(16))) float lds[IN_ELEMS + W_ELEMS];
  for (int i = threadIdx.x; i < IN_ELEMS + W_ELEMS; i += 256) {
    unsigned h = (i * 2654435761u) ^ (blockIdx.x * 40503u); h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
    lds[i] = ((int)(h & 0xffffff) - 0x800000) * (1.0f / 8388608.f);
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, s = lane >> 5, r = lane & 31;
  int abase[2];
  for (int mt = 0; mt < 2; ++mt) {
    const int p = (wave * 2 + mt) * 32 + r;
    abase[mt] = s * G::PLANE + (p / TW) * G::HALO_W + (p % TW);
  }
  const int bbase = s * TN + r;
  f32x16 acc[2][2];
  for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
  for (int it = 0; it < iters; ++it) {
    asm volatile("" ::: "memory");   // LDS content is loop invariant here: forbid hoisting the reads
    if (VARIANT == 0) mma_chunk<KS, TN, CK, G::PLANE, G::HALO_W>(lds, lds + IN_ELEMS, abase, bbase, acc);
    else mma_chunk_v1<KS, TN, CK, G::PLANE, G::HALO_W>(lds, lds + IN_ELEMS, abase, bbase, acc);
  }
  float sum = 0.f;
  for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int i = 0; i < 16; ++i) sum += acc[a][b][i];
  out[blockIdx.x * 256 + threadIdx.x] = sum;
}

int main() {
  float* out;
  (void)hipMalloc(&out, 4096 * 256 * sizeof(float));
  const int iters = 64;
  for (int variant = 0; variant < 2; ++variant)
    for (int grid : {512, 2048}) {
      hipEvent_t e0, e1;
      (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
      float ms = 0;
      for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        if (variant == 0) hipLaunchKernelGGL(bench_mma<0>, dim3(grid), dim3(256), 0, 0, out, iters);
        else hipLaunchKernelGGL(bench_mma<1>, dim3(grid), dim3(256), 0, 0, out, iters);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
      }
      double flops = (double)grid * 4 * iters * 288.0 * 4096.0;
      printf("variant=%d grid=%d: %.3f ms  %.1f TFLOP/s\n", variant, grid, ms, flops / ms / 1e9);
    }
  return 0;
}
