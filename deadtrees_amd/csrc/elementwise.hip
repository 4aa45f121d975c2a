// HBM-bound NHWC passes of the U-Net hot path: BatchNorm finalize/apply/backward, ReLU, residual add,
// max-pool, nearest-upsample backward, layout shuttles, uint8 normalisation.
// Replaces ATen batch_norm / relu_ / add / max_pool2d / interpolate-backward launched by the smp
// ResNet-34 encoder + UnetDecoder (reference call site deadtrees/network/segmodel.py:214).
// All kernels: 16 B per lane (f32x4) coalesced along the channel-fastest NHWC axis, wave64 shuffle
// reductions, deterministic two-stage sums (no float atomics).
#include "common.h"

#include <math.h>

// ------------------------------------------------------------------ generic row-block reduction
// in [planes][P][N] -> out [planes][PB][N], PB = ceil(P/RB): block (col-block, row-block, plane) sums RB rows.
// 256 threads = Q column-quads x (256/Q) row-lanes, 4 independent accumulators per thread for memory-level
// parallelism, LDS combine across row-lanes in a fixed order.
__global__ __launch_bounds__(256) void reduce_rows_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                          int P, int N, int RB, int Q) {
  __shared__ f32x4 sh[256];
  const int N4 = N >> 2;
  const int t = threadIdx.x;
  const int q = t % Q, rl = t / Q, RL = 256 / Q;
  const int cq = blockIdx.x * Q + q;
  const int r0 = blockIdx.y * RB;
  int r1 = r0 + RB;
  if (r1 > P) r1 = P;
  const int PB = gridDim.y;
  const f32x4* src = reinterpret_cast<const f32x4*>(in) + (size_t)blockIdx.z * P * N4;
  f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
  if (cq < N4) {
    int r = r0 + rl;
    for (; r + 3 * RL < r1; r += 4 * RL) {
      const f32x4 v0 = src[(size_t)r * N4 + cq];
      const f32x4 v1 = src[(size_t)(r + RL) * N4 + cq];
      const f32x4 v2 = src[(size_t)(r + 2 * RL) * N4 + cq];
      const f32x4 v3 = src[(size_t)(r + 3 * RL) * N4 + cq];
      a0 += v0; a1 += v1; a2 += v2; a3 += v3;
    }
    for (; r < r1; r += RL) a0 += src[(size_t)r * N4 + cq];
  }
  sh[t] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (rl == 0 && cq < N4) {
    f32x4 s = sh[q];
    for (int i = 1; i < RL; ++i) s += sh[i * Q + q];
    reinterpret_cast<f32x4*>(out)[((size_t)blockIdx.z * PB + blockIdx.y) * N4 + cq] = s;
  }
}

int dt_reduce_rows_launch(const float* in, float* out, int planes, int P, int N, int RB, hipStream_t st) {
  const int N4 = N / 4;
  int Q = 64;
  while (Q > N4) Q >>= 1;   // N4 >= 1; Q in {1,2,4,...,64} divides 256
  if (Q < 1) Q = 1;
  dim3 grid(dt_cdiv(N4, Q), dt_cdiv(P, RB), planes);
  hipLaunchKernelGGL(reduce_rows_kernel, grid, dim3(256), 0, st, in, out, P, N, RB, Q);
  DT_LAUNCH_CHECK();
  return DT_OK;
}

// ------------------------------------------------------------------ BN finalize
// one workgroup per 16 channels; 256 threads = 16 channels x 16 row-lanes
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ stats, int P, int C,
                                                          double count, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float eps,
                                                          float momentum, float* running_mean,
                                                          float* running_var, float* mean, float* invstd,
                                                          float* scale, float* shift) {
  __shared__ double red[2][16][17];
  const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  double s1 = 0.0, s2 = 0.0;
  if (c < C) {
    // 8 independent loads per round (pure latency: <= 256 rows, one launch per BatchNorm layer)
    double u1 = 0.0, u2 = 0.0, v1 = 0.0, v2 = 0.0, w1 = 0.0, w2 = 0.0;
    int p = rl;
    for (; p + 48 < P; p += 64) {
      const float a0 = stats[(size_t)p * C + c], a1 = stats[(size_t)(p + 16) * C + c];
      const float a2 = stats[(size_t)(p + 32) * C + c], a3 = stats[(size_t)(p + 48) * C + c];
      const float b0 = stats[((size_t)P + p) * C + c], b1 = stats[((size_t)P + p + 16) * C + c];
      const float b2 = stats[((size_t)P + p + 32) * C + c], b3 = stats[((size_t)P + p + 48) * C + c];
      s1 += (double)a0; u1 += (double)a1; v1 += (double)a2; w1 += (double)a3;
      s2 += (double)b0; u2 += (double)b1; v2 += (double)b2; w2 += (double)b3;
    }
    for (; p < P; p += 16) {
      s1 += (double)stats[(size_t)p * C + c];
      s2 += (double)stats[((size_t)P + p) * C + c];
    }
    s1 = (s1 + u1) + (v1 + w1);
    s2 = (s2 + u2) + (v2 + w2);
  }
  red[0][rl][cl] = s1;
  red[1][rl][cl] = s2;
  __syncthreads();
  if (rl == 0 && c < C) {
    double t1 = 0.0, t2 = 0.0;
    for (int i = 0; i < 16; ++i) {
      t1 += red[0][i][cl];
      t2 += red[1][i][cl];
    }
    const double m = t1 / count;
    double var = t2 / count - m * m;
    if (var < 0.0) var = 0.0;
    const double is = 1.0 / sqrt(var + (double)eps);
    mean[c] = (float)m;
    invstd[c] = (float)is;
    const float sc = gamma[c] * (float)is;
    scale[c] = sc;
    shift[c] = beta[c] - (float)m * sc;
    if (running_mean) {
      const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
      running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)m;
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
    }
  }
}

#define BN_STAGE1_MIN_ROWS 256
#define BN_STAGE1_RB 64

extern "C" int64_t dt_bn_stats_floats(int P, int C) {
  return (int64_t)2 * P * C + (int64_t)2 * dt_reduce_rows_out(P, BN_STAGE1_RB) * C;
}

extern "C" int dt_bn_finalize(float* stats, int P, int C, double count, const float* gamma,
                              const float* beta, float eps, float momentum, float* running_mean,
                              float* running_var, float* mean, float* invstd, float* scale, float* shift,
                              void* stream) {
  DT_REQUIRE(stats && gamma && beta && mean && invstd && scale && shift && P > 0 && C > 0 && count > 0,
             "bn_finalize: bad args");
  if (P > BN_STAGE1_MIN_ROWS && (C & 3) == 0) {
    // two-stage: 64-row blocks first (many workgroups), into the scratch tail of the stats buffer
    const int PB = dt_reduce_rows_out(P, BN_STAGE1_RB);
    float* scratch = stats + (size_t)2 * P * C;
    int rc = dt_reduce_rows_launch(stats, scratch, 2, P, C, BN_STAGE1_RB, (hipStream_t)stream);
    if (rc != DT_OK) return rc;
    stats = scratch;
    P = PB;
  }
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(dt_cdiv(C, 16)), dim3(256), 0, (hipStream_t)stream, stats, P, C,
                     count, gamma, beta, eps, momentum, running_mean, running_var, mean, invstd, scale, shift);
  DT_LAUNCH_CHECK();
  return DT_OK;
}

__global__ void bn_eval_affine_kernel(const float* gamma, const float* beta, const float* rm, const float* rv,
                                      float eps, int C, float* scale, float* shift) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C) {
    const float is = 1.f / sqrtf(rv[c] + eps);
    const float sc = gamma[c] * is;
    scale[c] = sc;
    shift[c] = beta[c] - rm[c] * sc;
  }
}

__global__ void bn_eval_stats_kernel(const float* rm, const float* rv, float eps, int C, float* mean, float* invstd) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C) {
    mean[c] = rm[c];
    invstd[c] = 1.f / sqrtf(rv[c] + eps);
  }
}

extern "C" int dt_bn_eval_stats(const float* rm, const float* rv, float eps, int C, float* mean, float* invstd,
                                void* stream) {
  DT_REQUIRE(rm && rv && mean && invstd && C > 0, "bn_eval_stats: bad args");
  hipLaunchKernelGGL(bn_eval_stats_kernel, dim3(dt_cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, rm, rv, eps, C,
                     mean, invstd);
  DT_LAUNCH_CHECK();
  return DT_OK;
}

extern "C" int dt_bn_eval_affine(const float* gamma, const float* beta, const float* rm, const float* rv,
                                 float eps, int C, float* scale, float* shift, void* stream) {
  DT_REQUIRE(gamma && beta && rm && rv && scale && shift && C > 0, "bn_eval_affine: bad args");
  hipLaunchKernelGGL(bn_eval_affine_kernel, dim3(dt_cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, gamma,
                     beta, rm, rv, eps, C, scale, shift);
  DT_LAUNCH_CHECK();
  return DT_OK;
}

// ------------------------------------------------------------------ per-channel sums (bias gradient of a conv)
// dbias[c] = sum_pixels g[p][c] (convolution_backward's bias gradient; the ResUnet decoder's 1x1 identity_conv and
// any other biased convolution): row blocks -> partial rows -> fixed-order second stage (fp64 final), no atomics.
#define CS_RB 256
__global__ __launch_bounds__(256) void channel_sums_kernel(const f32x4* __restrict__ g, float* __restrict__ part,
                                                           int64_t n_pix, int C4) {
  // thread t owns channel quad t % C4 of rows (t / C4) + k * (256 / C4) inside this block's CS_RB rows
  const int q = threadIdx.x % C4, rl = threadIdx.x / C4, RL = 256 / C4;
  const int64_t p0 = (int64_t)blockIdx.x * CS_RB;
  int64_t p1 = p0 + CS_RB;
  if (p1 > n_pix) p1 = n_pix;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (rl < RL)
    for (int64_t p = p0 + rl; p < p1; p += RL) s += g[p * C4 + q];
  __shared__ f32x4 sh[256];
  sh[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x < C4) {
    f32x4 t = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < RL; ++k) t += sh[k * C4 + threadIdx.x];   // fixed order
    reinterpret_cast<f32x4*>(part)[(size_t)blockIdx.x * C4 + threadIdx.x] = t;
  }
}

__global__ void channel_sums_final_kernel(const float* __restrict__ part, int P, int C, float* __restrict__ out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C) {
    double s = 0.0;
    for (int p = 0; p < P; ++p) s += (double)part[(size_t)p * C + c];
    out[c] = (float)s;
  }
}

extern "C" int64_t dt_channel_sums_workspace(int64_t n_pix, int C) {
  const int P = dt_cdiv(n_pix, CS_RB);
  const int P2 = dt_reduce_rows_out(P, 64);
  return ((int64_t)P + P2) * C;
}

extern "C" int dt_channel_sums(const float* g, float* workspace, int64_t n_pix, int C, float* out, void* stream) {
  DT_REQUIRE(g && workspace && out && n_pix > 0 && C > 0, "channel_sums: bad args");
  DT_REQUIRE((C & 3) == 0 && C / 4 <= 256 && 256 % (C / 4) == 0, "channel_sums: C/4 must divide 256 (C=%d)", C);
  hipStream_t st = (hipStream_t)stream;
  const int P = dt_cdiv(n_pix, CS_RB);
  hipLaunchKernelGGL(channel_sums_kernel, dim3(P), dim3(256), 0, st, (const f32x4*)g, workspace, n_pix, C / 4);
  DT_LAUNCH_CHECK();
  const float* rows = workspace;
  int rowsP = P;
  if (P > 64) {
    float* stage = workspace + (size_t)P * C;
    int rc = dt_reduce_rows_launch(workspace, stage, 1, P, C, 64, st);
    if (rc != DT_OK) return rc;
    rows = stage;
    rowsP = dt_reduce_rows_out(P, 64);
  }
  hipLaunchKernelGGL(channel_sums_final_kernel, dim3(dt_cdiv(C, 256)), dim3(256), 0, st, rows, rowsP, C, out);
  DT_LAUNCH_CHECK();
  return DT_OK;
}

// ------------------------------------------------------------------ channel-slice copies (dense decoders)
// torch.cat of the Unet++ dense skip connections (smp UnetPlusPlusDecoder.forward; in-tree twin: reference
// deadtrees/network/extra/efficientunetplusplus/decoder.py:170-177) and its backward: NHWC tensors, channel counts
// multiples of 4.  wide[n, off : off + Cn] = narrow[n, :]   /   narrow[n, :] (+)= wide[n, off : off + Cn]
__global__ __launch_bounds__(256) void channel_slice_kernel(const f32x4* __restrict__ src, f32x4* __restrict__ dst,
                                                            int64_t n4, int Cn4, int Cw4, int off4, int to_wide, int acc) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const int64_t pix = i / Cn4;
    const int c = (int)(i - pix * Cn4);
    const int64_t w = pix * Cw4 + off4 + c;
    if (to_wide) {
      dst[w] = src[i];
    } else {
      f32x4 v = src[w];
      if (acc) v += dst[i];
      dst[i] = v;
    }
  }
}

extern "C" int dt_channel_slice(const float* src, float* dst, int64_t n_pix, int C_narrow, int C_wide, int offset,
                                int to_wide, int accumulate, void* stream) {
  DT_REQUIRE(src && dst && n_pix > 0 && C_narrow > 0 && C_wide >= C_narrow, "channel_slice: bad args");
  DT_REQUIRE((C_narrow & 3) == 0 && (C_wide & 3) == 0 && (offset & 3) == 0 && offset >= 0 && offset + C_narrow <= C_wide,
             "channel_slice: channel counts and offset must be multiples of 4 inside the wide tensor");
  const int64_t n4 = n_pix * (C_narrow / 4);
  const int grid = (int)(n4 / 256 + 1 < 8192 ? n4 / 256 + 1 : 8192);
  hipLaunchKernelGGL(channel_slice_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const f32x4*)src, (f32x4*)dst,
                     n4, C_narrow / 4, C_wide / 4, offset / 4, to_wide, accumulate);
  DT_LAUNCH_CHECK();
  return DT_OK;
}

// ------------------------------------------------------------------ BN apply (+residual) (+ReLU)
__global__ __launch_bounds__(256) void bn_act_kernel(const f32x4* __restrict__ y, const float* __restrict__ scale,
                                                     const float* __restrict__ shift,
                                                     const f32x4* __restrict__ res,
                                                     const float* __restrict__ rscale,
                                                     const float* __restrict__ rshift, f32x4* __restrict__ out,
                                                     int64_t n4, int C4, int relu) {
  // When the grid stride is a multiple of C4 a thread stays on one channel quad: its coefficients are loaded once
  // (they are L1 hits, but every per-iteration 16-byte coefficient load costs the same issue slot as a data load).
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool fixed = (stride % C4) == 0;
  const f32x4 one = {1.f, 1.f, 1.f, 1.f}, zero = {0.f, 0.f, 0.f, 0.f};
  int c4 = (int)(i0 % C4);
  f32x4 sc = reinterpret_cast<const f32x4*>(scale)[c4], sh = reinterpret_cast<const f32x4*>(shift)[c4];
  f32x4 rsc = rscale ? reinterpret_cast<const f32x4*>(rscale)[c4] : one;
  f32x4 rsh = rscale ? reinterpret_cast<const f32x4*>(rshift)[c4] : zero;
  for (int64_t i = i0; i < n4; i += stride) {
    if (!fixed) {
      c4 = (int)(i % C4);
      sc = reinterpret_cast<const f32x4*>(scale)[c4];
      sh = reinterpret_cast<const f32x4*>(shift)[c4];
      if (rscale) {
        rsc = reinterpret_cast<const f32x4*>(rscale)[c4];
        rsh = reinterpret_cast<const f32x4*>(rshift)[c4];
      }
    }
    f32x4 v = y[i] * sc + sh;
    if (relu == 2) {   // ReLU on the main branch only, residual added after it (ResUnet decoder block)
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = v[k] < 0.f ? 0.f : v[k];
    }
    if (res) {
      f32x4 rv = res[i];
      if (rscale) rv = rv * rsc + rsh;
      v += rv;
    }
    if (relu == 1) {
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = v[k] < 0.f ? 0.f : v[k];   // NaN stays NaN, like torch.relu
    }
    out[i] = v;
  }
}

static inline int ew_grid(int64_t n_items) {
  int64_t g = (n_items + 255) / 256;
  const int64_t cap = 256 * 16;  // 16 workgroups per CU, grid-stride beyond
  return (int)(g < cap ? (g > 0 ? g : 1) : cap);
}

extern "C" int dt_bn_act(const float* y, const float* scale, const float* shift, const float* res,
                         const float* rscale, const float* rshift, float* out, int64_t n_pix, int C, int relu,
                         void* stream) {
  DT_REQUIRE(y && scale && shift && out && n_pix > 0 && C > 0 && (C & 3) == 0, "bn_act: bad args (C%%4)");
  DT_REQUIRE((rscale == nullptr) == (rshift == nullptr), "bn_act: rscale/rshift must come together");
  const int64_t n4 = n_pix * C / 4;
  hipLaunchKernelGGL(bn_act_kernel, dim3(ew_grid(n4)), dim3(256), 0, (hipStream_t)stream, (const f32x4*)y, scale,
                     shift, (const f32x4*)res, rscale, rshift, (f32x4*)out, n4, C / 4, relu);
  DT_LAUNCH_CHECK();
  return DT_OK;
}

// ------------------------------------------------------------------ BN backward
// pass 1: per-channel partial sums of g and g*xhat over blocks of BNB_RB pixels.
// 256 threads = Q channel-quads x (256/Q) pixel lanes; 2 pixels in flight per thread.
#define BNB_RB 256
// rows (pixels) per workgroup: 256 for small maps, grown so that a launch has at most ~2048 row blocks
static inline int64_t bnb_rb(int64_t n_pix) {
  int64_t rb = BNB_RB;
  const int64_t want = (n_pix + 2047) / 2048;
  if (want > rb) rb = (want + BNB_RB - 1) / BNB_RB * BNB_RB;
  return rb;
}

extern "C" int dt_bn_bwd_rows(int64_t n_pix, int C) {
  (void)C;
  return dt_cdiv(n_pix, bnb_rb(n_pix));
}
extern "C" int64_t dt_bn_bwd_red_floats(int64_t n_pix, int C) { return dt_bn_stats_floats(dt_bn_bwd_rows(n_pix, C), C); }

__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const f32x4* __restrict__ dout,
                                                            const f32x4* __restrict__ out_act,
                                                            const f32x4* __restrict__ y,
                                                            const float* __restrict__ mean,
                                                            const float* __restrict__ invstd,
                                                            const float* __restrict__ act_scale,
                                                            const float* __restrict__ act_shift,
                                                            float* __restrict__ red, int64_t n_pix, int C4, int Q,
                                                            int P, int64_t RB) {
  __shared__ f32x4 sh[2][256];
  const int t = threadIdx.x;
  const int q = t % Q, rl = t / Q, RL = 256 / Q;
  const int cq = blockIdx.x * Q + q;
  const int64_t p0 = (int64_t)blockIdx.y * RB;
  int64_t p1 = p0 + RB;
  if (p1 > n_pix) p1 = n_pix;
  const f32x4 mu = reinterpret_cast<const f32x4*>(mean)[cq];
  const f32x4 is = reinterpret_cast<const f32x4*>(invstd)[cq];
  // ReLU mask either from the stored activation or recomputed from y (virtual activation: z = relu(y*sc+sh))
  const bool from_y = out_act == nullptr && act_scale != nullptr;
  f32x4 asc = {1.f, 1.f, 1.f, 1.f}, ash = {0.f, 0.f, 0.f, 0.f};
  if (from_y) {
    asc = reinterpret_cast<const f32x4*>(act_scale)[cq];
    ash = reinterpret_cast<const f32x4*>(act_shift)[cq];
  }
  f32x4 sg0 = {0.f, 0.f, 0.f, 0.f}, sx0 = sg0, sg1 = sg0, sx1 = sg0;
  int64_t p = p0 + rl;
  for (; p + RL < p1; p += 2 * RL) {
    const size_t o0 = (size_t)p * C4 + cq, o1 = (size_t)(p + RL) * C4 + cq;
    f32x4 g0 = dout[o0], g1 = dout[o1];
    const f32x4 y0 = y[o0], y1 = y[o1];
    if (out_act) {
      const f32x4 a0 = out_act[o0], a1 = out_act[o1];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        g0[k] = a0[k] > 0.f ? g0[k] : 0.f;
        g1[k] = a1[k] > 0.f ? g1[k] : 0.f;
      }
    } else if (from_y) {
      const f32x4 a0 = y0 * asc + ash, a1 = y1 * asc + ash;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        g0[k] = a0[k] > 0.f ? g0[k] : 0.f;
        g1[k] = a1[k] > 0.f ? g1[k] : 0.f;
      }
    }
    sg0 += g0; sx0 += g0 * ((y0 - mu) * is);
    sg1 += g1; sx1 += g1 * ((y1 - mu) * is);
  }
  for (; p < p1; p += RL) {
    const size_t o0 = (size_t)p * C4 + cq;
    f32x4 g0 = dout[o0];
    if (out_act) {
      const f32x4 a0 = out_act[o0];
#pragma unroll
      for (int k = 0; k < 4; ++k) g0[k] = a0[k] > 0.f ? g0[k] : 0.f;
    } else if (from_y) {
      const f32x4 a0 = y[o0] * asc + ash;
#pragma unroll
      for (int k = 0; k < 4; ++k) g0[k] = a0[k] > 0.f ? g0[k] : 0.f;
    }
    sg0 += g0; sx0 += g0 * ((y[o0] - mu) * is);
  }
  sh[0][t] = sg0 + sg1;
  sh[1][t] = sx0 + sx1;
  __syncthreads();
  // pairwise tree over the row lanes (thread t = rl * Q + q): fixed order -> deterministic
  for (int s = RL >> 1; s >= 1; s >>= 1) {
    if (rl < s) {
      sh[0][t] += sh[0][t + s * Q];
      sh[1][t] += sh[1][t + s * Q];
    }
    __syncthreads();
  }
  if (rl == 0) {
    reinterpret_cast<f32x4*>(red)[(size_t)blockIdx.y * C4 + cq] = sh[0][t];
    reinterpret_cast<f32x4*>(red)[((size_t)P + blockIdx.y) * C4 + cq] = sh[1][t];
  }
}

extern "C" int dt_bn_bwd_reduce(const float* dout, const float* out_act, const float* y, const float* mean,
                                const float* invstd, const float* act_scale, const float* act_shift, float* red,
                                int64_t n_pix, int C, void* stream) {
  DT_REQUIRE(dout && y && mean && invstd && red && n_pix > 0 && C > 0 && (C & 3) == 0, "bn_bwd_reduce: bad args");
  const int C4 = C / 4;
  int Q = 64;
  while (Q > C4) Q >>= 1;
  DT_REQUIRE(C4 % Q == 0, "bn_bwd_reduce: C/4 must be a power of two or a multiple of 64 (C=%d)", C);
  const int P = dt_bn_bwd_rows(n_pix, C);
  hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(C4 / Q, P), dim3(256), 0, (hipStream_t)stream, (const f32x4*)dout,
                     (const f32x4*)out_act, (const f32x4*)y, mean, invstd, act_scale, act_shift, red, n_pix, C4, Q, P,
                     bnb_rb(n_pix));
  DT_LAUNCH_CHECK();
  return DT_OK;
}

// finalize the per-channel sums (fp64, fixed order): sums[0][C] = sum g, sums[1][C] = sum g*xhat
__global__ __launch_bounds__(256) void bn_bwd_sums_kernel(const float* __restrict__ red, int P, int C,
                                                          float* __restrict__ dgamma, float* __restrict__ dbeta) {
  __shared__ double sh[2][16][17];
  const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  double s1 = 0.0, s2 = 0.0;
  if (c < C) {
    // 8 independent loads per round (these kernels are pure latency: <= 256 rows, one launch per BatchNorm layer)
    double u1 = 0.0, u2 = 0.0, v1 = 0.0, v2 = 0.0, w1 = 0.0, w2 = 0.0;
    int p = rl;
    for (; p + 48 < P; p += 64) {
      const float a0 = red[(size_t)p * C + c], a1 = red[(size_t)(p + 16) * C + c];
      const float a2 = red[(size_t)(p + 32) * C + c], a3 = red[(size_t)(p + 48) * C + c];
      const float b0 = red[((size_t)P + p) * C + c], b1 = red[((size_t)P + p + 16) * C + c];
      const float b2 = red[((size_t)P + p + 32) * C + c], b3 = red[((size_t)P + p + 48) * C + c];
      s1 += (double)a0; u1 += (double)a1; v1 += (double)a2; w1 += (double)a3;
      s2 += (double)b0; u2 += (double)b1; v2 += (double)b2; w2 += (double)b3;
    }
    for (; p < P; p += 16) {
      s1 += (double)red[(size_t)p * C + c];
      s2 += (double)red[((size_t)P + p) * C + c];
    }
    s1 = (s1 + u1) + (v1 + w1);
    s2 = (s2 + u2) + (v2 + w2);
  }
  sh[0][rl][cl] = s1;
  sh[1][rl][cl] = s2;
  __syncthreads();
  if (rl == 0 && c < C) {
    double t1 = 0.0, t2 = 0.0;
    for (int i = 0; i < 16; ++i) {
      t1 += sh[0][i][cl];
      t2 += sh[1][i][cl];
    }
    dbeta[c] = (float)t1;
    dgamma[c] = (float)t2;
  }
}

// partial rows red[2][P][C] (+ scratch tail) -> dgamma, dbeta : two-stage, fixed order (shared with the bf16 path)
int dt_bn_bwd_finish_sums(float* red, int P, int C, float* dgamma, float* dbeta, hipStream_t st) {
  if (P > BN_STAGE1_MIN_ROWS) {
    const int PB = dt_reduce_rows_out(P, BN_STAGE1_RB);
    float* scratch = red + (size_t)2 * P * C;
    int rc = dt_reduce_rows_launch(red, scratch, 2, P, C, BN_STAGE1_RB, st);
    if (rc != DT_OK) return rc;
    red = scratch;
    P = PB;
  }
  hipLaunchKernelGGL(bn_bwd_sums_kernel, dim3(dt_cdiv(C, 16)), dim3(256), 0, st, red, P, C, dgamma, dbeta);
  DT_LAUNCH_CHECK();
  return DT_OK;
}

// the gradient and the raw output are read for the LAST time by this pass: nontemporal loads (same-box A/B: fp32 step
// +0.4 %, bf16 neutral)
#define BN_LD(p) __builtin_nontemporal_load(p)
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(
    const f32x4* __restrict__ dout, const f32x4* __restrict__ out_act, const f32x4* __restrict__ y,
    const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ gamma,
    const float* __restrict__ dgamma, const float* __restrict__ dbeta, const float* __restrict__ act_scale,
    const float* __restrict__ act_shift, f32x4* __restrict__ dy, f32x4* __restrict__ dres, int dres_acc, int64_t n4,
    int C4, float inv_count) {
  // one channel quad per thread when the grid stride is a multiple of C4 (see bn_act_kernel): 5-7 coefficient loads
  // per element group become 5-7 per thread
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool fixed = (stride % C4) == 0;
  f32x4 asc = {1.f, 1.f, 1.f, 1.f}, ash = {0.f, 0.f, 0.f, 0.f}, mu, is, gi, kb, kg;
  auto load_coef = [&](int c4) {
    if (act_scale) {
      asc = reinterpret_cast<const f32x4*>(act_scale)[c4];
      ash = reinterpret_cast<const f32x4*>(act_shift)[c4];
    }
    mu = reinterpret_cast<const f32x4*>(mean)[c4];
    is = reinterpret_cast<const f32x4*>(invstd)[c4];
    gi = reinterpret_cast<const f32x4*>(gamma)[c4] * is;
    kb = reinterpret_cast<const f32x4*>(dbeta)[c4] * inv_count;
    kg = reinterpret_cast<const f32x4*>(dgamma)[c4] * inv_count;
  };
  load_coef((int)(i0 % C4));
  for (int64_t i = i0; i < n4; i += stride) {
    if (!fixed) load_coef((int)(i % C4));
    f32x4 g = BN_LD(dout + i);
    const f32x4 yv = BN_LD(y + i);
    if (out_act) {
      const f32x4 a = BN_LD(out_act + i);
#pragma unroll
      for (int k = 0; k < 4; ++k) g[k] = a[k] > 0.f ? g[k] : 0.f;
    } else if (act_scale) {
      const f32x4 a = yv * asc + ash;
#pragma unroll
      for (int k = 0; k < 4; ++k) g[k] = a[k] > 0.f ? g[k] : 0.f;
    }
    if (dres) {
      if (dres_acc)
        dres[i] = dres[i] + g;
      else
        dres[i] = g;
    }
    const f32x4 xh = (yv - mu) * is;
    dy[i] = gi * (g - kb - xh * kg);
  }
}

static int bn_bwd_apply_impl(const float* dout, const float* out_act, const float* y, const float* mean,
                             const float* invstd, const float* gamma, const float* act_scale, const float* act_shift,
                             float* red, int P, float* dgamma, float* dbeta, float* dy, float* dres, int dres_accumulate,
                             int64_t n_pix, int C, void* stream, bool frozen) {
  DT_REQUIRE(dout && y && mean && invstd && gamma && red && dgamma && dbeta && dy && n_pix > 0 && C > 0 &&
                 (C & 3) == 0 && P > 0,
             "bn_bwd_apply: bad args");
  hipStream_t st = (hipStream_t)stream;
  int rc0 = dt_bn_bwd_finish_sums(red, P, C, dgamma, dbeta, st);
  if (rc0 != DT_OK) return rc0;
  const int64_t n4 = n_pix * C / 4;
  // frozen statistics (eval-mode BatchNorm): mean / invstd are constants of the layer, so the two batch-mean terms
  // of the training-mode formula vanish -> inv_count = 0 turns dy into g * gamma * invstd
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(ew_grid(n4)), dim3(256), 0, st, (const f32x4*)dout,
                     (const f32x4*)out_act, (const f32x4*)y, mean, invstd, gamma, dgamma, dbeta, act_scale, act_shift,
                     (f32x4*)dy, (f32x4*)dres, dres_accumulate, n4, C / 4, frozen ? 0.f : (float)(1.0 / (double)n_pix));
  DT_LAUNCH_CHECK();
  return DT_OK;
}

extern "C" int dt_bn_bwd_apply(const float* dout, const float* out_act, const float* y, const float* mean,
                               const float* invstd, const float* gamma, const float* act_scale,
                               const float* act_shift, float* red, int P, float* dgamma, float* dbeta, float* dy,
                               float* dres, int dres_accumulate, int64_t n_pix, int C, void* stream) {
  return bn_bwd_apply_impl(dout, out_act, y, mean, invstd, gamma, act_scale, act_shift, red, P, dgamma, dbeta, dy, dres,
                           dres_accumulate, n_pix, C, stream, false);
}

extern "C" int dt_bn_bwd_apply_frozen(const float* dout, const float* out_act, const float* y, const float* mean,
                                      const float* invstd, const float* gamma, const float* act_scale,
                                      const float* act_shift, float* red, int P, float* dgamma, float* dbeta, float* dy,
                                      float* dres, int dres_accumulate, int64_t n_pix, int C, void* stream) {
  return bn_bwd_apply_impl(dout, out_act, y, mean, invstd, gamma, act_scale, act_shift, red, P, dgamma, dbeta, dy, dres,
                           dres_accumulate, n_pix, C, stream, true);
}

// ------------------------------------------------------------------ max-pool 3x3 / stride 2 / pad 1
__global__ __launch_bounds__(256) void maxpool_kernel(const f32x4* __restrict__ x, f32x4* __restrict__ out,
                                                      uint32_t* __restrict__ amax, int B, int H, int W, int C4,
                                                      int Ho, int Wo) {
  const int64_t total = (int64_t)B * Ho * Wo * C4;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int c4 = (int)(i % C4);
    int64_t r = i / C4;
    const int ox = (int)(r % Wo);
    r /= Wo;
    const int oy = (int)(r % Ho);
    const int b = (int)(r / Ho);
    f32x4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    int bi[4] = {0, 0, 0, 0};
    bool first[4] = {true, true, true, true};
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int iy = 2 * oy - 1 + kh;
      if ((unsigned)iy >= (unsigned)H) continue;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int ix = 2 * ox - 1 + kw;
        if ((unsigned)ix >= (unsigned)W) continue;
        const f32x4 v = x[(((int64_t)b * H + iy) * W + ix) * C4 + c4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          // ATen: update when (val > max) || isnan(val); the first valid element always initialises
          if (first[k] || v[k] > best[k] || v[k] != v[k]) {
            best[k] = v[k];
            bi[k] = kh * 3 + kw;
            first[k] = false;
          }
        }
      }
    }
    out[i] = best;
    if (amax) amax[i] = (uint32_t)bi[0] | ((uint32_t)bi[1] << 8) | ((uint32_t)bi[2] << 16) | ((uint32_t)bi[3] << 24);
  }
}

extern "C" int dt_maxpool3x3s2(const float* x, float* out, uint8_t* argmax, int B, int H, int W, int C,
                               void* stream) {
  DT_REQUIRE(x && out && B > 0 && H > 0 && W > 0 && C > 0 && (C & 3) == 0, "maxpool: bad args");
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const int64_t total = (int64_t)B * Ho * Wo * (C / 4);
  hipLaunchKernelGGL(maxpool_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, (const f32x4*)x,
                     (f32x4*)out, (uint32_t*)argmax, B, H, W, C / 4, Ho, Wo);
  DT_LAUNCH_CHECK();
  return DT_OK;
}

__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const f32x4* __restrict__ dout,
                                                          const uint32_t* __restrict__ amax, f32x4* __restrict__ dx,
                                                          int acc, int B, int H, int W, int C4, int Ho, int Wo) {
  const int64_t total = (int64_t)B * H * W * C4;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int c4 = (int)(i % C4);
    int64_t r = i / C4;
    const int ix = (int)(r % W);
    r /= W;
    const int iy = (int)(r % H);
    const int b = (int)(r / H);
    f32x4 g = {0.f, 0.f, 0.f, 0.f};
    // windows (oy,ox) that contain (iy,ix): 2*oy-1 <= iy <= 2*oy+1
    const int oy_lo = iy >> 1, oy_hi = (iy + 1) >> 1;
    const int ox_lo = ix >> 1, ox_hi = (ix + 1) >> 1;
    for (int oy = oy_lo; oy <= oy_hi; ++oy) {
      if (oy >= Ho) continue;
      const int kh = iy - (2 * oy - 1);
      for (int ox = ox_lo; ox <= ox_hi; ++ox) {
        if (ox >= Wo) continue;
        const int kw = ix - (2 * ox - 1);
        const uint32_t pos = (uint32_t)(kh * 3 + kw);
        const int64_t o = (((int64_t)b * Ho + oy) * Wo + ox) * C4 + c4;
        const uint32_t am = amax[o];
        const f32x4 d = dout[o];
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (((am >> (8 * k)) & 0xffu) == pos) g[k] += d[k];
      }
    }
    if (acc)
      dx[i] = dx[i] + g;
    else
      dx[i] = g;
  }
}

// Even maps (every training / fine-tuning tile): one thread per 2 x 2 block of input pixels.  Rows 2 q, 2 q + 1 and
// columns 2 p, 2 p + 1 lie in the four windows (q, p), (q, p + 1), (q + 1, p), (q + 1, p + 1) and in no other, so each
// window's gradient and index quad is loaded ONCE per block instead of once per pixel — the per-pixel form above reads
// 6 bytes from L2 for every byte it writes (measured: 302 us for a 537 MB output, the L2 rate, not HBM's).  The four
// pixels add their windows in the per-pixel kernel's order (oy, then ox, ascending): bit-identical.
__global__ __launch_bounds__(256) void maxpool_bwd_quad_kernel(const f32x4* __restrict__ dout,
                                                               const uint32_t* __restrict__ amax, f32x4* __restrict__ dx,
                                                               int acc, int B, int H, int W, int C4, int Ho, int Wo) {
  const int64_t total = (int64_t)B * Ho * Wo * C4;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int c4 = (int)(i % C4);
    int64_t r = i / C4;
    const int p = (int)(r % Wo);
    r /= Wo;
    const int q = (int)(r % Ho);
    const int b = (int)(r / Ho);
    const bool q1 = q + 1 < Ho, p1 = p + 1 < Wo;
    const int64_t o00 = i, o01 = i + C4, o10 = i + (int64_t)Wo * C4, o11 = o10 + C4;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    const f32x4 d00 = dout[o00], d01 = p1 ? dout[o01] : z, d10 = q1 ? dout[o10] : z, d11 = (q1 && p1) ? dout[o11] : z;
    // an index no tap has (0xff) for windows outside the map
    const uint32_t a00 = amax[o00], a01 = p1 ? amax[o01] : 0xffffffffu, a10 = q1 ? amax[o10] : 0xffffffffu,
                   a11 = (q1 && p1) ? amax[o11] : 0xffffffffu;
    f32x4 g00 = z, g01 = z, g10 = z, g11 = z;   // pixels (2q, 2p), (2q, 2p+1), (2q+1, 2p), (2q+1, 2p+1)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const uint32_t t00 = (a00 >> (8 * k)) & 0xffu, t01 = (a01 >> (8 * k)) & 0xffu, t10 = (a10 >> (8 * k)) & 0xffu,
                     t11 = (a11 >> (8 * k)) & 0xffu;
      if (t00 == 4u) g00[k] += d00[k];
      if (t00 == 5u) g01[k] += d00[k];
      if (t01 == 3u) g01[k] += d01[k];
      if (t00 == 7u) g10[k] += d00[k];
      if (t10 == 1u) g10[k] += d10[k];
      if (t00 == 8u) g11[k] += d00[k];
      if (t01 == 6u) g11[k] += d01[k];
      if (t10 == 2u) g11[k] += d10[k];
      if (t11 == 0u) g11[k] += d11[k];
    }
    const int64_t x00 = (((int64_t)b * H + 2 * q) * W + 2 * p) * C4 + c4, x10 = x00 + (int64_t)W * C4;
    if (acc) {
      dx[x00] = dx[x00] + g00;
      dx[x00 + C4] = dx[x00 + C4] + g01;
      dx[x10] = dx[x10] + g10;
      dx[x10 + C4] = dx[x10 + C4] + g11;
    } else {
      dx[x00] = g00;
      dx[x00 + C4] = g01;
      dx[x10] = g10;
      dx[x10 + C4] = g11;
    }
  }
}

// the same pass with the BatchNorm-backward sums of the layer whose (virtual) activation was pooled (the stem) taken from the
// gradient it writes: g = dx after the optional join, mask from y * scale + shift, sum g and sum g * xhat per channel; one
// partial row per workgroup (fixed channel quad per thread: the grid stride is a multiple of C / 4) -> dt_bn_bwd_apply
__global__ __launch_bounds__(256) void maxpool_bwd_quad_bn_kernel(const f32x4* __restrict__ dout,
                                                                  const uint32_t* __restrict__ amax, f32x4* __restrict__ dx,
                                                                  const f32x4* __restrict__ y, const float* __restrict__ mean,
                                                                  const float* __restrict__ invstd,
                                                                  const float* __restrict__ act_scale,
                                                                  const float* __restrict__ act_shift, float* __restrict__ red,
                                                                  int acc, int B, int H, int W, int C4, int Ho, int Wo, int P) {
  __shared__ f32x4 sh[2][256];
  const int64_t total = (int64_t)B * Ho * Wo * C4;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int t = threadIdx.x;
  const int64_t i0 = (int64_t)blockIdx.x * blockDim.x + t;
  const int c4 = (int)(i0 % C4);
  const f32x4 mu = reinterpret_cast<const f32x4*>(mean)[c4], is = reinterpret_cast<const f32x4*>(invstd)[c4];
  const f32x4 asc = reinterpret_cast<const f32x4*>(act_scale)[c4], ash = reinterpret_cast<const f32x4*>(act_shift)[c4];
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  f32x4 sg = z, sx = z;
  for (int64_t i = i0; i < total; i += stride) {
    int64_t r = i / C4;
    const int p = (int)(r % Wo);
    r /= Wo;
    const int q = (int)(r % Ho);
    const int b = (int)(r / Ho);
    const bool q1 = q + 1 < Ho, p1 = p + 1 < Wo;
    const int64_t o00 = i, o01 = i + C4, o10 = i + (int64_t)Wo * C4, o11 = o10 + C4;
    const f32x4 d00 = dout[o00], d01 = p1 ? dout[o01] : z, d10 = q1 ? dout[o10] : z, d11 = (q1 && p1) ? dout[o11] : z;
    const uint32_t a00 = amax[o00], a01 = p1 ? amax[o01] : 0xffffffffu, a10 = q1 ? amax[o10] : 0xffffffffu,
                   a11 = (q1 && p1) ? amax[o11] : 0xffffffffu;
    f32x4 g[4] = {z, z, z, z};   // pixels (2q, 2p), (2q, 2p+1), (2q+1, 2p), (2q+1, 2p+1)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const uint32_t t00 = (a00 >> (8 * k)) & 0xffu, t01 = (a01 >> (8 * k)) & 0xffu, t10 = (a10 >> (8 * k)) & 0xffu,
                     t11 = (a11 >> (8 * k)) & 0xffu;
      if (t00 == 4u) g[0][k] += d00[k];
      if (t00 == 5u) g[1][k] += d00[k];
      if (t01 == 3u) g[1][k] += d01[k];
      if (t00 == 7u) g[2][k] += d00[k];
      if (t10 == 1u) g[2][k] += d10[k];
      if (t00 == 8u) g[3][k] += d00[k];
      if (t01 == 6u) g[3][k] += d01[k];
      if (t10 == 2u) g[3][k] += d10[k];
      if (t11 == 0u) g[3][k] += d11[k];
    }
    const int64_t x00 = (((int64_t)b * H + 2 * q) * W + 2 * p) * C4 + c4, x10 = x00 + (int64_t)W * C4;
    const int64_t xs[4] = {x00, x00 + C4, x10, x10 + C4};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      f32x4 v = g[e];
      if (acc) v = dx[xs[e]] + v;
      dx[xs[e]] = v;
      const f32x4 yv = y[xs[e]];
      const f32x4 a = yv * asc + ash;
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = a[k] > 0.f ? v[k] : 0.f;
      sg += v;
      sx += v * ((yv - mu) * is);
    }
  }
  sh[0][t] = sg;
  sh[1][t] = sx;
  __syncthreads();
  const int rl = t / C4, RL = 256 / C4;     // threads t, t + C4, ... share the channel quad
  for (int s2 = RL >> 1; s2 >= 1; s2 >>= 1) {
    if (rl < s2) {
      sh[0][t] += sh[0][t + s2 * C4];
      sh[1][t] += sh[1][t + s2 * C4];
    }
    __syncthreads();
  }
  if (t < C4) {
    reinterpret_cast<f32x4*>(red)[(size_t)blockIdx.x * C4 + t] = sh[0][t];
    reinterpret_cast<f32x4*>(red)[((size_t)P + blockIdx.x) * C4 + t] = sh[1][t];
  }
}

extern "C" int dt_maxpool3x3s2_bwd_bn_rows(int B, int H, int W, int C) {
  if (((H | W) & 1) != 0 || C <= 0 || (C & 3) != 0 || C / 4 > 256 || 256 % (C / 4) != 0) return 0;   // even maps only
  return ew_grid((int64_t)B * (H / 2) * (W / 2) * (C / 4));
}

extern "C" int dt_maxpool3x3s2_bwd_bn(const float* dout, const uint8_t* argmax, float* dx, int accumulate,
                                      const dt_bn_bwd_fuse* fuse, float* red, int B, int H, int W, int C, void* stream) {
  DT_REQUIRE(dout && argmax && dx && fuse && red && fuse->y && fuse->mean && fuse->invstd && fuse->act_scale &&
                 fuse->act_shift && B > 0 && H > 0 && W > 0, "maxpool_bwd_bn: bad args");
  const int P = dt_maxpool3x3s2_bwd_bn_rows(B, H, W, C);
  DT_REQUIRE(P > 0, "maxpool_bwd_bn: even maps, C/4 a divisor of 256 (H=%d W=%d C=%d)", H, W, C);
  hipLaunchKernelGGL(maxpool_bwd_quad_bn_kernel, dim3(P), dim3(256), 0, (hipStream_t)stream, (const f32x4*)dout,
                     (const uint32_t*)argmax, (f32x4*)dx, (const f32x4*)fuse->y, fuse->mean, fuse->invstd, fuse->act_scale,
                     fuse->act_shift, red, accumulate, B, H, W, C / 4, H / 2, W / 2, P);
  DT_LAUNCH_CHECK();
  return DT_OK;
}

extern "C" int dt_maxpool3x3s2_bwd(const float* dout, const uint8_t* argmax, float* dx, int accumulate, int B,
                                   int H, int W, int C, void* stream) {
  DT_REQUIRE(dout && argmax && dx && B > 0 && H > 0 && W > 0 && C > 0 && (C & 3) == 0, "maxpool_bwd: bad args");
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  if (((H | W) & 1) == 0) {
    hipLaunchKernelGGL(maxpool_bwd_quad_kernel, dim3(ew_grid((int64_t)B * Ho * Wo * (C / 4))), dim3(256), 0,
                       (hipStream_t)stream, (const f32x4*)dout, (const uint32_t*)argmax, (f32x4*)dx, accumulate, B, H, W,
                       C / 4, Ho, Wo);
    DT_LAUNCH_CHECK();
    return DT_OK;
  }
  const int64_t total = (int64_t)B * H * W * (C / 4);
  hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream,
                     (const f32x4*)dout, (const uint32_t*)argmax, (f32x4*)dx, accumulate, B, H, W, C / 4, Ho, Wo);
  DT_LAUNCH_CHECK();
  return DT_OK;
}

// ------------------------------------------------------------------ nearest x2 upsample backward (2x2 sum)
__global__ __launch_bounds__(256) void upsample2x_bwd_kernel(const f32x4* __restrict__ dup, f32x4* __restrict__ dx,
                                                             int acc, int B, int H, int W, int C4) {
  const int64_t total = (int64_t)B * H * W * C4;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int W2 = 2 * W;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int c4 = (int)(i % C4);
    int64_t r = i / C4;
    const int x = (int)(r % W);
    r /= W;
    const int y = (int)(r % H);
    const int b = (int)(r / H);
    const int64_t base = (((int64_t)b * 2 * H + 2 * y) * W2 + 2 * x) * C4 + c4;
    f32x4 g = (dup[base] + dup[base + C4]) + (dup[base + (int64_t)W2 * C4] + dup[base + (int64_t)W2 * C4 + C4]);
    if (acc)
      dx[i] = dx[i] + g;
    else
      dx[i] = g;
  }
}

extern "C" int dt_upsample2x_bwd(const float* dup, float* dx, int accumulate, int B, int H, int W, int C,
                                 void* stream) {
  DT_REQUIRE(dup && dx && B > 0 && H > 0 && W > 0 && C > 0 && (C & 3) == 0, "upsample2x_bwd: bad args");
  const int64_t total = (int64_t)B * H * W * (C / 4);
  hipLaunchKernelGGL(upsample2x_bwd_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream,
                     (const f32x4*)dup, (f32x4*)dx, accumulate, B, H, W, C / 4);
  DT_LAUNCH_CHECK();
  return DT_OK;
}

// The same 2x2 sum with the BatchNorm-backward reduction of the layer that produced the (virtual) activation fused in:
// g is the gradient of relu(bn(y)); the pass that writes g also accumulates sum g*mask and sum g*mask*xhat from the
// layer's raw output y, so the separate dt_bn_bwd_reduce over (g, y) disappears.  One partial row per workgroup.
__global__ __launch_bounds__(256) void upsample2x_bwd_bn_kernel(const f32x4* __restrict__ dup, f32x4* __restrict__ dx,
                                                                const f32x4* __restrict__ y,
                                                                const float* __restrict__ mean,
                                                                const float* __restrict__ invstd,
                                                                const float* __restrict__ act_scale,
                                                                const float* __restrict__ act_shift,
                                                                float* __restrict__ red, int B, int H, int W, int C4,
                                                                int P) {
  __shared__ f32x4 sh[2][256];
  const int64_t total = (int64_t)B * H * W * C4;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;   // multiple of C4 (host check): fixed channel quad
  const int W2 = 2 * W, t = threadIdx.x;
  const int64_t i0 = (int64_t)blockIdx.x * blockDim.x + t;
  const int c4 = (int)(i0 % C4);
  const f32x4 mu = reinterpret_cast<const f32x4*>(mean)[c4], is = reinterpret_cast<const f32x4*>(invstd)[c4];
  const f32x4 asc = reinterpret_cast<const f32x4*>(act_scale)[c4], ash = reinterpret_cast<const f32x4*>(act_shift)[c4];
  f32x4 sg = {0.f, 0.f, 0.f, 0.f}, sx = sg;
  for (int64_t i = i0; i < total; i += stride) {
    int64_t r = i / C4;
    const int x = (int)(r % W);
    r /= W;
    const int yy = (int)(r % H);
    const int b = (int)(r / H);
    const int64_t base = (((int64_t)b * 2 * H + 2 * yy) * W2 + 2 * x) * C4 + c4;
    f32x4 g = (dup[base] + dup[base + C4]) + (dup[base + (int64_t)W2 * C4] + dup[base + (int64_t)W2 * C4 + C4]);
    dx[i] = g;
    const f32x4 yv = y[i];
    const f32x4 a = yv * asc + ash;
#pragma unroll
    for (int k = 0; k < 4; ++k) g[k] = a[k] > 0.f ? g[k] : 0.f;
    sg += g;
    sx += g * ((yv - mu) * is);
  }
  sh[0][t] = sg;
  sh[1][t] = sx;
  __syncthreads();
  const int rl = t / C4, RL = 256 / C4;     // threads t, t + C4, ... share the channel quad
  for (int s = RL >> 1; s >= 1; s >>= 1) {
    if (rl < s) {
      sh[0][t] += sh[0][t + s * C4];
      sh[1][t] += sh[1][t + s * C4];
    }
    __syncthreads();
  }
  if (t < C4) {
    reinterpret_cast<f32x4*>(red)[(size_t)blockIdx.x * C4 + t] = sh[0][t];
    reinterpret_cast<f32x4*>(red)[((size_t)P + blockIdx.x) * C4 + t] = sh[1][t];
  }
}

extern "C" int dt_upsample2x_bwd_bn_rows(int B, int H, int W, int C) {
  return ew_grid((int64_t)B * H * W * (C / 4));
}

extern "C" int dt_upsample2x_bwd_bn(const float* dup, float* dx, const dt_bn_bwd_fuse* fuse, float* red, int B, int H,
                                    int W, int C, void* stream) {
  DT_REQUIRE(dup && dx && fuse && red && fuse->y && fuse->mean && fuse->invstd && fuse->act_scale && fuse->act_shift &&
                 B > 0 && H > 0 && W > 0 && C > 0 && (C & 3) == 0,
             "upsample2x_bwd_bn: bad args");
  const int C4 = C / 4;
  DT_REQUIRE(C4 <= 256 && 256 % C4 == 0, "upsample2x_bwd_bn: C/4 must divide 256 (C=%d)", C);
  const int P = dt_upsample2x_bwd_bn_rows(B, H, W, C);
  hipLaunchKernelGGL(upsample2x_bwd_bn_kernel, dim3(P), dim3(256), 0, (hipStream_t)stream, (const f32x4*)dup,
                     (f32x4*)dx, (const f32x4*)fuse->y, fuse->mean, fuse->invstd, fuse->act_scale, fuse->act_shift, red,
                     B, H, W, C4, P);
  DT_LAUNCH_CHECK();
  return DT_OK;
}

// ------------------------------------------------------------------ layout shuttles
// NCHW <-> NHWC for small C (image: 3/4 channels; logits: K): one thread per pixel, planar side coalesced.
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                           int B, int C, int64_t HW) {
  const int64_t total = (int64_t)B * HW;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int64_t b = i / HW, p = i - b * HW;
    for (int c = 0; c < C; ++c) dst[i * C + c] = src[(b * C + c) * HW + p];
  }
}
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                           int B, int C, int64_t HW) {
  const int64_t total = (int64_t)B * HW;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int64_t b = i / HW, p = i - b * HW;
    for (int c = 0; c < C; ++c) dst[(b * C + c) * HW + p] = src[i * C + c];
  }
}

extern "C" int dt_nchw_to_nhwc(const float* src, float* dst, int B, int C, int H, int W, void* stream) {
  DT_REQUIRE(src && dst && B > 0 && C > 0 && H > 0 && W > 0, "nchw_to_nhwc: bad args");
  const int64_t HW = (int64_t)H * W;
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(ew_grid(B * HW)), dim3(256), 0, (hipStream_t)stream, src, dst, B, C,
                     HW);
  DT_LAUNCH_CHECK();
  return DT_OK;
}
extern "C" int dt_nhwc_to_nchw(const float* src, float* dst, int B, int C, int H, int W, void* stream) {
  DT_REQUIRE(src && dst && B > 0 && C > 0 && H > 0 && W > 0, "nhwc_to_nchw: bad args");
  const int64_t HW = (int64_t)H * W;
  hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(ew_grid(B * HW)), dim3(256), 0, (hipStream_t)stream, src, dst, B, C,
                     HW);
  DT_LAUNCH_CHECK();
  return DT_OK;
}

__global__ __launch_bounds__(256) void normalize_u8_kernel(const uint8_t* __restrict__ src, float* __restrict__ dst,
                                                           int64_t n_pix, int Cs, int Cd, f32x4 mean, f32x4 stdv) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_pix; i += stride) {
    for (int c = 0; c < Cd; ++c) {
      // albumentations Normalize(max_pixel_value=255): (x - mean*255) / (std*255), computed in fp32
      const float v = (float)src[i * Cs + c];
      dst[i * Cd + c] = (v - mean[c] * 255.f) * (1.f / (stdv[c] * 255.f));
    }
  }
}

extern "C" int dt_normalize_u8(const uint8_t* src, float* dst, int64_t n_pix, int Csrc, int Cdst,
                               const float* mean, const float* stdv, void* stream) {
  DT_REQUIRE(src && dst && mean && stdv && n_pix > 0 && Cdst > 0 && Cdst <= 4 && Cdst <= Csrc,
             "normalize_u8: bad args");
  f32x4 m = {0, 0, 0, 0}, s = {1, 1, 1, 1};
  for (int c = 0; c < Cdst; ++c) {
    m[c] = mean[c];
    s[c] = stdv[c];
  }
  hipLaunchKernelGGL(normalize_u8_kernel, dim3(ew_grid(n_pix)), dim3(256), 0, (hipStream_t)stream, src, dst, n_pix,
                     Csrc, Cdst, m, s);
  DT_LAUNCH_CHECK();
  return DT_OK;
}

// ------------------------------------------------------------------ tiled inference input
// deployment/tiler.py:121-134 + utils/data_handling.py:9-20 (zero-pad the raster to the tile, cut it into d x d blocks in
// row-major block order) + scripts/inference.py:94-96 (albumentations Normalize per sub-tile) as ONE gather: raster uint8
// [Cs][h][w] (band-major, what rioxarray hands over) -> fp32 NHWC sub-tiles [count][d][d][Cd], blocks first .. first +
// count - 1 of the nbx-wide block grid.  Pixels beyond the raster are the tiler's zero padding, normalised like any
// other zero byte.  One thread per output pixel: Cd byte loads (coalesced along x per band), Cd dword stores.
__global__ __launch_bounds__(256) void split_normalize_u8_kernel(const uint8_t* __restrict__ src, float* __restrict__ dst,
                                                                 int h, int w, int d, int nbx, int first, int64_t n_pix,
                                                                 int Cd, f32x4 mean, f32x4 stdv) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t plane = (int64_t)h * w;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_pix; i += stride) {
    const int x = (int)(i % d), y = (int)((i / d) % d), blk = first + (int)(i / ((int64_t)d * d));
    const int gy = (blk / nbx) * d + y, gx = (blk % nbx) * d + x;
    const bool in = gy < h && gx < w;
    for (int c = 0; c < Cd; ++c) {
      const float v = in ? (float)src[c * plane + (int64_t)gy * w + gx] : 0.f;
      dst[i * Cd + c] = (v - mean[c] * 255.f) * (1.f / (stdv[c] * 255.f));   // the arithmetic of normalize_u8_kernel
    }
  }
}

extern "C" int dt_split_normalize_u8(const uint8_t* raster_chw, float* dst_nhwc, int Csrc, int h, int w, int d, int nbx,
                                     int first_block, int n_blocks, int Cdst, const float* mean, const float* stdv,
                                     void* stream) {
  DT_REQUIRE(raster_chw && dst_nhwc && mean && stdv && h > 0 && w > 0 && d > 0 && nbx > 0 && first_block >= 0 &&
                 n_blocks > 0 && Cdst > 0 && Cdst <= 4 && Cdst <= Csrc,
             "split_normalize_u8: bad args");
  f32x4 m = {0, 0, 0, 0}, s = {1, 1, 1, 1};
  for (int c = 0; c < Cdst; ++c) {
    m[c] = mean[c];
    s[c] = stdv[c];
  }
  const int64_t n_pix = (int64_t)n_blocks * d * d;
  hipLaunchKernelGGL(split_normalize_u8_kernel, dim3(ew_grid(n_pix)), dim3(256), 0, (hipStream_t)stream, raster_chw,
                     dst_nhwc, h, w, d, nbx, first_block, n_pix, Cdst, m, s);
  DT_LAUNCH_CHECK();
  return DT_OK;
}

// scripts/inference.py:60-62 `is_valid_tile`: a raster whose first band holds only 0 / 255 is skipped.  flag[0] (int32,
// zeroed by the caller) becomes 1 as soon as any byte differs from both: a reduction without a host pass over the raster.
__global__ __launch_bounds__(256) void band_has_data_kernel(const uint8_t* __restrict__ band, int64_t n,
                                                            int32_t* __restrict__ flag) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  bool any = false;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const uint8_t v = band[i];
    any = any || (v != 0 && v != 255);
  }
  if (__builtin_amdgcn_ballot_w64(any) != 0 && (threadIdx.x & 63) == 0) flag[0] = 1;   // idempotent store: no atomic
}

extern "C" int dt_band_has_data(const uint8_t* band, int64_t n, int32_t* flag, void* stream) {
  DT_REQUIRE(band && flag && n > 0, "band_has_data: bad args");
  hipLaunchKernelGGL(band_has_data_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, band, n, flag);
  DT_LAUNCH_CHECK();
  return DT_OK;
}

// ------------------------------------------------------------------ training augmentation on the device
// data/deadtreedata.py:128-146 `train_transform`: OneOf(HorizontalFlip, VerticalFlip) -> RandomRotate90 ->
// RandomBrightnessContrast(brightness_by_max=False) -> Normalize -> ToTensorV2, per sample on loader CPUs.
// Here: the host draws the per-sample parameters, one gather pass applies flip + rot90 + the brightness/contrast
// LUT + normalisation while converting uint8 NHWC tiles to the fp32 NHWC stem input; labels take the same
// geometric map.  geo[b] = (flip: 0 none, 1 horizontal, 2 vertical; rot: k of np.rot90, counter-clockwise).
__device__ __forceinline__ void aug_source_pixel(int flip, int rot, int y, int x, int H, int W, int* sy, int* sx) {
  // out = rot90^k(flip(in)):  rot90(m,1)[i][j] = m[j][N-1-i]
  int ry = y, rx = x;
  if (rot == 1) { ry = x; rx = W - 1 - y; }
  else if (rot == 2) { ry = H - 1 - y; rx = W - 1 - x; }
  else if (rot == 3) { ry = H - 1 - x; rx = y; }
  if (flip == 1) rx = W - 1 - rx;
  else if (flip == 2) ry = H - 1 - ry;
  *sy = ry;
  *sx = rx;
}

__global__ __launch_bounds__(256) void u8_image_sums_kernel(const uint8_t* __restrict__ src, int64_t n_per,
                                                            unsigned long long* __restrict__ sums) {
  const int b = blockIdx.y;
  const uint8_t* p = src + (size_t)b * n_per;
  unsigned long long s = 0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_per; i += (int64_t)gridDim.x * 256) s += p[i];
  double d = wave_sum_d((double)s);   // < 2^53: exact
  __shared__ double sh[4];
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = d;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(&sums[b], (unsigned long long)(sh[0] + sh[1] + sh[2] + sh[3]));  // integer: exact
}

__global__ __launch_bounds__(256) void augment_normalize_u8_kernel(const uint8_t* __restrict__ src,
                                                                   float* __restrict__ dst,
                                                                   const int32_t* __restrict__ geo,
                                                                   const float* __restrict__ bc,
                                                                   const unsigned long long* __restrict__ sums, int H,
                                                                   int W, int Cs, int Cd, f32x4 mean, f32x4 stdv) {
  const int b = blockIdx.y;
  const int flip = geo[2 * b], rot = geo[2 * b + 1];
  const float alpha = bc[2 * b], beta = bc[2 * b + 1];
  // RandomBrightnessContrast on uint8 (brightness_by_max=False): lut(v) = uint8(clip(v*alpha + beta*mean(img), 0, 255))
  const float add = beta == 0.f ? 0.f : (float)((double)beta * ((double)sums[b] / ((double)H * W * Cs)));
  const int64_t n_pix = (int64_t)H * W;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_pix; i += (int64_t)gridDim.x * 256) {
    const int y = (int)(i / W), x = (int)(i - (int64_t)y * W);
    int sy, sx;
    aug_source_pixel(flip, rot, y, x, H, W, &sy, &sx);
    const uint8_t* sp = src + (((size_t)b * H + sy) * W + sx) * Cs;
    float* dp = dst + ((size_t)b * n_pix + i) * Cd;
    for (int c = 0; c < Cd; ++c) {
      float v = (float)sp[c];
      if (alpha != 1.f || beta != 0.f) {
        v = v * alpha + add;
        v = v < 0.f ? 0.f : (v > 255.f ? 255.f : v);
        v = floorf(v);
      }
      dp[c] = (v - mean[c] * 255.f) * (1.f / (stdv[c] * 255.f));
    }
  }
}

__global__ __launch_bounds__(256) void augment_labels_kernel(const int64_t* __restrict__ src, int64_t* __restrict__ dst,
                                                             const int32_t* __restrict__ geo, int H, int W) {
  const int b = blockIdx.y;
  const int flip = geo[2 * b], rot = geo[2 * b + 1];
  const int64_t n_pix = (int64_t)H * W;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_pix; i += (int64_t)gridDim.x * 256) {
    const int y = (int)(i / W), x = (int)(i - (int64_t)y * W);
    int sy, sx;
    aug_source_pixel(flip, rot, y, x, H, W, &sy, &sx);
    dst[(size_t)b * n_pix + i] = src[((size_t)b * H + sy) * W + sx];
  }
}

extern "C" int dt_augment_normalize_u8(const uint8_t* src, float* dst, const int32_t* geo, const float* bc,
                                       uint64_t* sums_scratch, int B, int H, int W, int Csrc, int Cdst,
                                       const float* mean, const float* stdv, void* stream) {
  DT_REQUIRE(src && dst && geo && bc && sums_scratch && mean && stdv && B > 0 && H > 0 && W > 0 && Cdst > 0 &&
                 Cdst <= 4 && Cdst <= Csrc,
             "augment_normalize_u8: bad args");
  DT_REQUIRE(B <= 65535, "augment_normalize_u8: B must be <= 65535");
  f32x4 m = {0, 0, 0, 0}, s = {1, 1, 1, 1};
  for (int c = 0; c < Cdst; ++c) {
    m[c] = mean[c];
    s[c] = stdv[c];
  }
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(sums_scratch, 0, (size_t)B * sizeof(uint64_t), st) != hipSuccess) {
    dt_set_error("augment_normalize_u8: memset failed");
    return DT_EHIP;
  }
  const int64_t n_per = (int64_t)H * W * Csrc;
  int gx = dt_cdiv(n_per, 256 * 16);
  if (gx > 256) gx = 256;
  hipLaunchKernelGGL(u8_image_sums_kernel, dim3(gx, B), dim3(256), 0, st, src, n_per, (unsigned long long*)sums_scratch);
  DT_LAUNCH_CHECK();
  int gp = dt_cdiv((int64_t)H * W, 256);
  if (gp > 1024) gp = 1024;
  hipLaunchKernelGGL(augment_normalize_u8_kernel, dim3(gp, B), dim3(256), 0, st, src, dst, geo, bc,
                     (const unsigned long long*)sums_scratch, H, W, Csrc, Cdst, m, s);
  DT_LAUNCH_CHECK();
  return DT_OK;
}

extern "C" int dt_augment_labels(const int64_t* src, int64_t* dst, const int32_t* geo, int B, int H, int W,
                                 void* stream) {
  DT_REQUIRE(src && dst && geo && B > 0 && B <= 65535 && H > 0 && W > 0, "augment_labels: bad args");
  int gp = dt_cdiv((int64_t)H * W, 256);
  if (gp > 1024) gp = 1024;
  hipLaunchKernelGGL(augment_labels_kernel, dim3(gp, B), dim3(256), 0, (hipStream_t)stream, src, dst, geo, H, W);
  DT_LAUNCH_CHECK();
  return DT_OK;
}
