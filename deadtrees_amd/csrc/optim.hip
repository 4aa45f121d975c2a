// Fused global-norm clip + Adam over ONE flat parameter buffer, plus library plumbing.
// Replaces torch.nn.utils.clip_grad_norm_(0.5) (reference configs/trainer/default.yaml:18, applied by
// Lightning) and torch.optim.Adam.step (deadtrees/network/segmodel.py:420-425) — ~47x4 small ATen
// launches per step in the reference — with two HBM-bound passes: read g (norm), then
// read g,m,v,p / write m,v,p.  The clip coefficient and the non-finite-loss skip flag stay on the
// device: no host synchronisation in the step.
#include "common.h"

#include <math.h>
#include <string.h>

static thread_local char g_err[512] = "";

void dt_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* dt_last_error(void) { return g_err; }
extern "C" int dt_version(void) { return 100; }
extern "C" int dt_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return -1;
  return n;
}

#define SUMSQ_PER_WG (256 * 4 * 16)

extern "C" int dt_sumsq_rows(int64_t n) { return dt_cdiv(n, SUMSQ_PER_WG); }

__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, int64_t n,
                                                    double* __restrict__ partial) {
  const int64_t base = (int64_t)blockIdx.x * SUMSQ_PER_WG;
  float s = 0.f;
  for (int it = 0; it < 16; ++it) {
    const int64_t i = base + ((int64_t)it * 256 + threadIdx.x) * 4;
    if (i + 3 < n) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(g + i);
      s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    } else {
      for (int64_t j = i; j < n && j < i + 4; ++j) s += g[j] * g[j];
    }
  }
  __shared__ double sh[4];
  const double w = wave_sum_d((double)s);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = w;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

extern "C" int dt_sumsq(const float* g, int64_t n, double* partial, void* stream) {
  DT_REQUIRE(g && partial && n > 0, "sumsq: bad args");
  DT_REQUIRE((((uintptr_t)g) & 15) == 0, "sumsq: g must be 16-byte aligned");
  hipLaunchKernelGGL(sumsq_kernel, dim3(dt_sumsq_rows(n)), dim3(256), 0, (hipStream_t)stream, g, n, partial);
  DT_LAUNCH_CHECK();
  return DT_OK;
}

__global__ __launch_bounds__(256) void clip_coef_kernel(const double* __restrict__ partial, int rows,
                                                        float max_norm, float gscale, float* norm,
                                                        float* clipcoef, int32_t* skip) {
  __shared__ double sh[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < rows; i += 256) s += partial[i];
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    // gradients in the buffer are gscale * true gradient (e.g. sum over replicas -> gscale = 1/N)
    const float nrm = (float)sqrt(sh[0]) * fabsf(gscale);
    norm[0] = nrm;
    float c = max_norm > 0.f ? max_norm / (nrm + 1e-6f) : 1.f;
    if (c > 1.f) c = 1.f;
    clipcoef[0] = c * gscale;
    // a NaN/Inf anywhere in the gradient (e.g. through BatchNorm statistics of a channel that overflowed while the loss
    // stayed finite) would reach every parameter through the clip coefficient: drop the update instead
    if (skip && !(nrm == nrm && fabsf(nrm) != INFINITY)) skip[0] = 1;
  }
}

extern "C" int dt_clip_coef(const double* partial, int rows, float max_norm, float gscale, float* norm,
                            float* clipcoef, int32_t* skip_flag, void* stream) {
  DT_REQUIRE(partial && norm && clipcoef && rows > 0, "clip_coef: bad args");
  hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partial, rows, max_norm, gscale,
                     norm, clipcoef, skip_flag);
  DT_LAUNCH_CHECK();
  return DT_OK;
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, int64_t n,
                                                   float lr, float b1, float b2, float omb1, float omb2, float eps,
                                                   float bc1, float bc2,
                                                   const float* __restrict__ clipcoef,
                                                   const int32_t* __restrict__ skip,
                                                   const float* __restrict__ hyper) {
  if (skip && skip[0] != 0) return;
  if (hyper) {   // per-step scalars read from the device: a captured HIP graph replays with fresh values
    lr = hyper[0];
    bc1 = hyper[1];
    bc2 = hyper[2];
  }
  const float cc = clipcoef ? clipcoef[0] : 1.f;
  const float step = lr / bc1;
  const float rs2 = 1.f / sqrtf(bc2);
  const int64_t n4 = n >> 2;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const f32x4 gg = reinterpret_cast<const f32x4*>(g)[i] * cc;
    f32x4 mm = reinterpret_cast<f32x4*>(m)[i];
    f32x4 vv = reinterpret_cast<f32x4*>(v)[i];
    f32x4 pp = reinterpret_cast<f32x4*>(p)[i];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      // torch.optim.Adam (single-tensor path): exp_avg.lerp_(grad, 1-b1); exp_avg_sq = b2*v + (1-b2) g^2
      // omb = 1 - beta computed in double on the host like torch's Python scalars (1.f - 0.999f is off by 1.3e-5)
      mm[k] = mm[k] + (gg[k] - mm[k]) * omb1;
      vv[k] = vv[k] * b2 + omb2 * gg[k] * gg[k];
      const float denom = sqrtf(vv[k]) * rs2 + eps;
      pp[k] = pp[k] - step * (mm[k] / denom);
    }
    reinterpret_cast<f32x4*>(m)[i] = mm;
    reinterpret_cast<f32x4*>(v)[i] = vv;
    reinterpret_cast<f32x4*>(p)[i] = pp;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    for (int64_t i = n4 << 2; i < n; ++i) {
      const float gk = g[i] * cc;
      const float mk = m[i] + (gk - m[i]) * omb1;
      const float vk = v[i] * b2 + omb2 * gk * gk;
      m[i] = mk;
      v[i] = vk;
      p[i] = p[i] - step * (mk / (sqrtf(vk) * rs2 + eps));
    }
  }
}

extern "C" int dt_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, double beta1,
                            double beta2, float eps, float bias_c1, float bias_c2, const float* clipcoef,
                            const int32_t* skip_flag, void* stream) {
  DT_REQUIRE(p && g && m && v && n > 0 && bias_c1 > 0.f && bias_c2 > 0.f, "adam: bad args");
  DT_REQUIRE(((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)m) | ((uintptr_t)v)) & 15) == 0,
             "adam: buffers must be 16-byte aligned");
  int64_t grid = ((n >> 2) + 255) / 256;
  if (grid > 256 * 16) grid = 256 * 16;
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr,
                     (float)beta1, (float)beta2, (float)(1.0 - beta1), (float)(1.0 - beta2), eps, bias_c1, bias_c2, clipcoef,
                     skip_flag, (const float*)nullptr);
  DT_LAUNCH_CHECK();
  return DT_OK;
}

// ---- per-step scalars of the optimiser on the device (no ATen algebra, no host sync, HIP-graph capturable)
__global__ void skip_from_loss_kernel(const float* __restrict__ loss, int32_t* __restrict__ skip) {
  const float l = loss[0];
  skip[0] = (l == l && fabsf(l) != INFINITY) ? 0 : 1;
}

extern "C" int dt_skip_from_loss(const float* loss, int32_t* skip_flag, void* stream) {
  DT_REQUIRE(loss && skip_flag, "skip_from_loss: bad args");
  hipLaunchKernelGGL(skip_from_loss_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, loss, skip_flag);
  DT_LAUNCH_CHECK();
  return DT_OK;
}

__global__ void adam_advance_kernel(double* __restrict__ t, const int32_t* __restrict__ skip,
                                    const double* __restrict__ lr, double b1, double b2, float* __restrict__ hyper) {
  double tt = t[0];
  if (!(skip && skip[0] != 0)) tt += 1.0;    // a skipped step leaves the step count alone (torch: no optimizer.step)
  t[0] = tt;
  const double te = tt < 1.0 ? 1.0 : tt;
  hyper[0] = (float)lr[0];
  hyper[1] = (float)(1.0 - pow(b1, te));
  hyper[2] = (float)(1.0 - pow(b2, te));
}

extern "C" int dt_adam_advance(double* t_dev, const int32_t* skip_flag, const double* lr_dev, double beta1, double beta2,
                               float* hyper, void* stream) {
  DT_REQUIRE(t_dev && lr_dev && hyper, "adam_advance: bad args");
  hipLaunchKernelGGL(adam_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, t_dev, skip_flag, lr_dev,
                     beta1, beta2, hyper);
  DT_LAUNCH_CHECK();
  return DT_OK;
}

extern "C" int dt_adam_step_dev(float* p, const float* g, float* m, float* v, int64_t n, const float* hyper,
                                double beta1, double beta2, float eps, const float* clipcoef, const int32_t* skip_flag,
                                void* stream) {
  DT_REQUIRE(p && g && m && v && hyper && n > 0, "adam_dev: bad args");
  DT_REQUIRE(((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)m) | ((uintptr_t)v)) & 15) == 0,
             "adam_dev: buffers must be 16-byte aligned");
  int64_t grid = ((n >> 2) + 255) / 256;
  if (grid > 256 * 16) grid = 256 * 16;
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, 0.f,
                     (float)beta1, (float)beta2, (float)(1.0 - beta1), (float)(1.0 - beta2), eps, 1.f, 1.f, clipcoef, skip_flag,
                     hyper);
  DT_LAUNCH_CHECK();
  return DT_OK;
}
