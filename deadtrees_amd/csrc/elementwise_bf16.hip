// HBM-bound passes of the bf16 training path: BatchNorm backward, max-pool (+argmax) and its backward,
// nearest-upsample backward, fp32 <-> bf16 shuttles.  8 bf16 (16 B) per lane; every sum in fp32/fp64 with the
// same deterministic two-stage reductions as the fp32 path (elementwise.hip).
#include "common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define BNB16_RB 256

__device__ __forceinline__ void load8(const bf16x8* p, size_t i, float (&v)[8]) {
  const bf16x8 a = p[i];
#pragma unroll
  for (int k = 0; k < 8; ++k) v[k] = (float)a[k];
}

// streaming form for tensors read for the last time (nontemporal loads, see elementwise.hip)
__device__ __forceinline__ void load8s(const bf16x8* p, size_t i, float (&v)[8]) {
  const bf16x8 a = __builtin_nontemporal_load(p + i);
#pragma unroll
  for (int k = 0; k < 8; ++k) v[k] = (float)a[k];
}

// 8 consecutive per-channel fp32 coefficients (c % 8 == 0) as two 16-byte loads
__device__ __forceinline__ void ldc8(const float* __restrict__ p, int c, float (&v)[8]) {
  const f32x4 a = *reinterpret_cast<const f32x4*>(p + c), b = *reinterpret_cast<const f32x4*>(p + c + 4);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    v[k] = a[k];
    v[4 + k] = b[k];
  }
}

// ------------------------------------------------------------------ BN backward, pass 1
// rows (pixels) per workgroup: 256 for small maps, grown so that a launch has at most ~2048 row blocks — the
// in-workgroup reduction and the second-stage row count then stay small next to the streaming part.
static inline int64_t bnb16_rb(int64_t n_pix) {
  int64_t rb = BNB16_RB;
  const int64_t want = (n_pix + 2047) / 2048;
  if (want > rb) rb = (want + BNB16_RB - 1) / BNB16_RB * BNB16_RB;
  return rb;
}

__global__ __launch_bounds__(256) void bn_bwd_reduce_bf16_kernel(const bf16x8* __restrict__ dout,
                                                                 const bf16x8* __restrict__ out_act,
                                                                 const bf16x8* __restrict__ y,
                                                                 const float* __restrict__ mean,
                                                                 const float* __restrict__ invstd,
                                                                 const float* __restrict__ act_scale,
                                                                 const float* __restrict__ act_shift,
                                                                 float* __restrict__ red, int64_t n_pix, int C8, int Q,
                                                                 int P, int64_t RB) {
  __shared__ float sh[16][256];   // [sum kind * 8 + k][thread]: conflict-free columns
  const int t = threadIdx.x;
  const int q = t % Q, rl = t / Q, RL = 256 / Q;
  const int cq = blockIdx.x * Q + q;
  const int64_t p0 = (int64_t)blockIdx.y * RB;
  int64_t p1 = p0 + RB;
  if (p1 > n_pix) p1 = n_pix;
  float mu[8], is[8], asc[8], ash[8], sg[8], sx[8];
  const bool from_y = out_act == nullptr && act_scale != nullptr;
  ldc8(mean, cq * 8, mu);
  ldc8(invstd, cq * 8, is);
  if (from_y) {
    ldc8(act_scale, cq * 8, asc);
    ldc8(act_shift, cq * 8, ash);
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    if (!from_y) {
      asc[k] = 1.f;
      ash[k] = 0.f;
    }
    sg[k] = sx[k] = 0.f;
  }
  auto masked = [&](float (&g)[8], const float (&yv)[8], size_t o) {
    if (out_act) {
      float a[8];
      load8(out_act, o, a);
#pragma unroll
      for (int k = 0; k < 8; ++k) g[k] = a[k] > 0.f ? g[k] : 0.f;
    } else if (from_y) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        // the consumers saw bf16(relu(y*sc+sh)): the mask is the sign of that rounded value
        const float a = (float)(__bf16)(yv[k] * asc[k] + ash[k]);
        g[k] = a > 0.f ? g[k] : 0.f;
      }
    }
  };
  int64_t p = p0 + rl;
  for (; p + RL < p1; p += 2 * RL) {   // two rows in flight per thread
    const size_t o0 = (size_t)p * C8 + cq, o1 = (size_t)(p + RL) * C8 + cq;
    float g0[8], y0[8], g1[8], y1[8];
    load8(dout, o0, g0);
    load8(dout, o1, g1);
    load8(y, o0, y0);
    load8(y, o1, y1);
    masked(g0, y0, o0);
    masked(g1, y1, o1);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      sg[k] += g0[k];
      sx[k] += g0[k] * ((y0[k] - mu[k]) * is[k]);
      sg[k] += g1[k];
      sx[k] += g1[k] * ((y1[k] - mu[k]) * is[k]);
    }
  }
  for (; p < p1; p += RL) {
    const size_t o = (size_t)p * C8 + cq;
    float g[8], yv[8];
    load8(dout, o, g);
    load8(y, o, yv);
    masked(g, yv, o);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      sg[k] += g[k];
      sx[k] += g[k] * ((yv[k] - mu[k]) * is[k]);
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    sh[k][t] = sg[k];
    sh[8 + k][t] = sx[k];
  }
  __syncthreads();
  // pairwise tree over the row lanes (thread t = rl * Q + q): fixed order -> deterministic
  for (int s = RL >> 1; s >= 1; s >>= 1) {
    if (rl < s) {
#pragma unroll
      for (int k = 0; k < 16; ++k) sh[k][t] += sh[k][t + s * Q];
    }
    __syncthreads();
  }
  if (rl == 0) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      red[(size_t)blockIdx.y * C8 * 8 + cq * 8 + k] = sh[k][t];
      red[((size_t)P + blockIdx.y) * C8 * 8 + cq * 8 + k] = sh[8 + k][t];
    }
  }
}

extern "C" int dt_bn_bwd_rows_bf16(int64_t n_pix) { return dt_cdiv(n_pix, bnb16_rb(n_pix)); }

extern "C" int dt_bn_bwd_reduce_bf16(const void* dout, const void* out_act, const void* y, const float* mean,
                                     const float* invstd, const float* act_scale, const float* act_shift, float* red,
                                     int64_t n_pix, int C, void* stream) {
  DT_REQUIRE(dout && y && mean && invstd && red && n_pix > 0 && C > 0 && (C & 7) == 0, "bn_bwd_reduce_bf16: bad args");
  DT_REQUIRE((((uintptr_t)mean | (uintptr_t)invstd | (uintptr_t)act_scale | (uintptr_t)act_shift) & 15) == 0,
             "bn_bwd_reduce_bf16: per-channel arrays must be 16-byte aligned");
  const int C8 = C / 8;
  int Q = 64;
  while (Q > C8) Q >>= 1;
  DT_REQUIRE(C8 % Q == 0, "bn_bwd_reduce_bf16: C/8 must be a power of two or a multiple of 64 (C=%d)", C);
  const int P = dt_bn_bwd_rows_bf16(n_pix);
  hipLaunchKernelGGL(bn_bwd_reduce_bf16_kernel, dim3(C8 / Q, P), dim3(256), 0, (hipStream_t)stream, (const bf16x8*)dout,
                     (const bf16x8*)out_act, (const bf16x8*)y, mean, invstd, act_scale, act_shift, red, n_pix, C8, Q, P,
                     bnb16_rb(n_pix));
  DT_LAUNCH_CHECK();
  return DT_OK;
}

// ------------------------------------------------------------------ BN backward, pass 2
__global__ __launch_bounds__(256) void bn_bwd_apply_bf16_kernel(
    const bf16x8* __restrict__ dout, const bf16x8* __restrict__ out_act, const bf16x8* __restrict__ y,
    const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ gamma,
    const float* __restrict__ dgamma, const float* __restrict__ dbeta, const float* __restrict__ act_scale,
    const float* __restrict__ act_shift, bf16x8* __restrict__ dy, bf16x8* __restrict__ dres, int dres_acc, int64_t n8,
    int C8, float inv_count) {
  // the grid stride is a multiple of C8 (host check), so a thread stays on ONE channel group: its per-channel
  // coefficients are loaded once instead of 56 scalar loads per 16-byte element group
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int c = (int)(i0 % C8) * 8;
  float mu[8], is[8], gi[8], kb[8], kg[8], asc[8], ash[8];
  {
    // c is a multiple of 8: two 16-byte loads per coefficient array, all issued before the first use
    float ga[8], db[8], dg[8];
    ldc8(mean, c, mu);
    ldc8(invstd, c, is);
    ldc8(gamma, c, ga);
    ldc8(dbeta, c, db);
    ldc8(dgamma, c, dg);
    if (act_scale) {
      ldc8(act_scale, c, asc);
      ldc8(act_shift, c, ash);
    } else {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        asc[k] = 1.f;
        ash[k] = 0.f;
      }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      gi[k] = ga[k] * is[k];
      kb[k] = db[k] * inv_count;
      kg[k] = dg[k] * inv_count;
    }
  }
  for (int64_t i = i0; i < n8; i += stride) {
    float g[8], yv[8];
    load8s(dout, i, g);
    load8s(y, i, yv);
    if (out_act) {
      float a[8];
      load8s(out_act, i, a);
#pragma unroll
      for (int k = 0; k < 8; ++k) g[k] = a[k] > 0.f ? g[k] : 0.f;
    } else if (act_scale) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float a = (float)(__bf16)(yv[k] * asc[k] + ash[k]);
        g[k] = a > 0.f ? g[k] : 0.f;
      }
    }
    if (dres) {
      bf16x8 o;
      if (dres_acc) {
        float prev[8];
        load8(dres, i, prev);
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = (__bf16)(prev[k] + g[k]);
      } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = (__bf16)g[k];
      }
      dres[i] = o;
    }
    bf16x8 o;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float xh = (yv[k] - mu[k]) * is[k];
      o[k] = (__bf16)(gi[k] * (g[k] - kb[k] - xh * kg[k]));
    }
    dy[i] = o;
  }
}

extern "C" int dt_bn_bwd_apply_bf16(const void* dout, const void* out_act, const void* y, const float* mean,
                                    const float* invstd, const float* gamma, const float* act_scale,
                                    const float* act_shift, float* red, int P, float* dgamma, float* dbeta, void* dy,
                                    void* dres, int dres_accumulate, int64_t n_pix, int C, void* stream) {
  DT_REQUIRE(dout && y && mean && invstd && gamma && red && dgamma && dbeta && dy && n_pix > 0 && C > 0 &&
                 (C & 7) == 0 && P > 0,
             "bn_bwd_apply_bf16: bad args");
  DT_REQUIRE(256 % (C / 8) == 0, "bn_bwd_apply_bf16: C/8 must divide 256 (C=%d)", C);
  DT_REQUIRE((((uintptr_t)mean | (uintptr_t)invstd | (uintptr_t)gamma | (uintptr_t)dgamma | (uintptr_t)dbeta |
               (uintptr_t)act_scale | (uintptr_t)act_shift) & 15) == 0,
             "bn_bwd_apply_bf16: per-channel arrays must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  int rc = dt_bn_bwd_finish_sums(red, P, C, dgamma, dbeta, st);
  if (rc != DT_OK) return rc;
  const int64_t n8 = n_pix * C / 8;
  int64_t g = (n8 + 255) / 256;
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(bn_bwd_apply_bf16_kernel, dim3((unsigned)g), dim3(256), 0, st, (const bf16x8*)dout,
                     (const bf16x8*)out_act, (const bf16x8*)y, mean, invstd, gamma, dgamma, dbeta, act_scale, act_shift,
                     (bf16x8*)dy, (bf16x8*)dres, dres_accumulate, n8, C / 8, (float)(1.0 / (double)n_pix));
  DT_LAUNCH_CHECK();
  return DT_OK;
}

// ------------------------------------------------------------------ max-pool with argmax / backward
__global__ __launch_bounds__(256) void maxpool_bf16_amax_kernel(const bf16x8* __restrict__ x, bf16x8* __restrict__ out,
                                                                uint2* __restrict__ amax, int B, int H, int W, int C8,
                                                                int Ho, int Wo) {
  const int64_t total = (int64_t)B * Ho * Wo * C8;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int c8 = (int)(i % C8);
    int64_t rr = i / C8;
    const int ox = (int)(rr % Wo);
    rr /= Wo;
    const int oy = (int)(rr % Ho);
    const int b = (int)(rr / Ho);
    float best[8];
    unsigned bi[8];
    bool first = true;
    for (int kh = 0; kh < 3; ++kh) {
      const int iy = 2 * oy - 1 + kh;
      if ((unsigned)iy >= (unsigned)H) continue;
      for (int kw = 0; kw < 3; ++kw) {
        const int ix = 2 * ox - 1 + kw;
        if ((unsigned)ix >= (unsigned)W) continue;
        float v[8];
        load8(x, (((int64_t)b * H + iy) * W + ix) * C8 + c8, v);
#pragma unroll
        for (int k = 0; k < 8; ++k)
          if (first || v[k] > best[k] || v[k] != v[k]) {
            best[k] = v[k];
            bi[k] = kh * 3 + kw;
          }
        first = false;
      }
    }
    bf16x8 o;
    unsigned lo = 0, hi = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      o[k] = (__bf16)best[k];
      if (k < 4) lo |= bi[k] << (8 * k); else hi |= bi[k] << (8 * (k - 4));
    }
    out[i] = o;
    amax[i] = make_uint2(lo, hi);
  }
}

extern "C" int dt_maxpool3x3s2_bf16_amax(const void* x, void* out, uint8_t* argmax, int B, int H, int W, int C,
                                         void* stream) {
  DT_REQUIRE(x && out && argmax && B > 0 && H > 0 && W > 0 && C > 0 && (C & 7) == 0, "maxpool_bf16_amax: bad args");
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const int64_t total = (int64_t)B * Ho * Wo * (C / 8);
  int64_t g = (total + 255) / 256;
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(maxpool_bf16_amax_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, (const bf16x8*)x,
                     (bf16x8*)out, (uint2*)argmax, B, H, W, C / 8, Ho, Wo);
  DT_LAUNCH_CHECK();
  return DT_OK;
}

__global__ __launch_bounds__(256) void maxpool_bwd_bf16_kernel(const bf16x8* __restrict__ dout,
                                                               const uint2* __restrict__ amax, bf16x8* __restrict__ dx,
                                                               int acc, int B, int H, int W, int C8, int Ho, int Wo) {
  const int64_t total = (int64_t)B * H * W * C8;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int c8 = (int)(i % C8);
    int64_t rr = i / C8;
    const int ix = (int)(rr % W);
    rr /= W;
    const int iy = (int)(rr % H);
    const int b = (int)(rr / H);
    float g[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int oy = iy >> 1; oy <= ((iy + 1) >> 1); ++oy) {
      if (oy >= Ho) continue;
      const int kh = iy - (2 * oy - 1);
      for (int ox = ix >> 1; ox <= ((ix + 1) >> 1); ++ox) {
        if (ox >= Wo) continue;
        const unsigned pos = (unsigned)(kh * 3 + (ix - (2 * ox - 1)));
        const int64_t o = (((int64_t)b * Ho + oy) * Wo + ox) * C8 + c8;
        const uint2 am = amax[o];
        float d[8];
        load8(dout, o, d);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const unsigned a = k < 4 ? (am.x >> (8 * k)) & 0xffu : (am.y >> (8 * (k - 4))) & 0xffu;
          if (a == pos) g[k] += d[k];
        }
      }
    }
    bf16x8 o;
    if (acc) {
      float prev[8];
      load8(dx, i, prev);
#pragma unroll
      for (int k = 0; k < 8; ++k) o[k] = (__bf16)(prev[k] + g[k]);
    } else {
#pragma unroll
      for (int k = 0; k < 8; ++k) o[k] = (__bf16)g[k];
    }
    dx[i] = o;
  }
}

// even maps: one thread per 2 x 2 block of input pixels — each of the (at most four) windows it lies in is loaded once
// (bf16 twin of maxpool_bwd_quad_kernel, elementwise.hip; same per-pixel summation order: bit-identical)
__global__ __launch_bounds__(256) void maxpool_bwd_bf16_quad_kernel(const bf16x8* __restrict__ dout,
                                                                    const uint2* __restrict__ amax, bf16x8* __restrict__ dx,
                                                                    int acc, int B, int H, int W, int C8, int Ho, int Wo) {
  const int64_t total = (int64_t)B * Ho * Wo * C8;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int c8 = (int)(i % C8);
    int64_t rr = i / C8;
    const int p = (int)(rr % Wo);
    rr /= Wo;
    const int q = (int)(rr % Ho);
    const int b = (int)(rr / Ho);
    const bool q1 = q + 1 < Ho, p1 = p + 1 < Wo;
    const int64_t o00 = i, o01 = i + C8, o10 = i + (int64_t)Wo * C8, o11 = o10 + C8;
    float d00[8], d01[8], d10[8], d11[8];
    const uint2 none = make_uint2(0xffffffffu, 0xffffffffu);    // an index no tap has
    load8(dout, o00, d00);
    load8(dout, p1 ? o01 : o00, d01);
    load8(dout, q1 ? o10 : o00, d10);
    load8(dout, (q1 && p1) ? o11 : o00, d11);
    const uint2 a00 = amax[o00], a01 = p1 ? amax[o01] : none, a10 = q1 ? amax[o10] : none,
                a11 = (q1 && p1) ? amax[o11] : none;
    float g00[8], g01[8], g10[8], g11[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int sh = 8 * (k & 3);
      const unsigned t00 = ((k < 4 ? a00.x : a00.y) >> sh) & 0xffu, t01 = ((k < 4 ? a01.x : a01.y) >> sh) & 0xffu,
                     t10 = ((k < 4 ? a10.x : a10.y) >> sh) & 0xffu, t11 = ((k < 4 ? a11.x : a11.y) >> sh) & 0xffu;
      float u00 = 0.f, u01 = 0.f, u10 = 0.f, u11 = 0.f;
      if (t00 == 4u) u00 += d00[k];
      if (t00 == 5u) u01 += d00[k];
      if (t01 == 3u) u01 += d01[k];
      if (t00 == 7u) u10 += d00[k];
      if (t10 == 1u) u10 += d10[k];
      if (t00 == 8u) u11 += d00[k];
      if (t01 == 6u) u11 += d01[k];
      if (t10 == 2u) u11 += d10[k];
      if (t11 == 0u) u11 += d11[k];
      g00[k] = u00; g01[k] = u01; g10[k] = u10; g11[k] = u11;
    }
    const int64_t x00 = (((int64_t)b * H + 2 * q) * W + 2 * p) * C8 + c8, x10 = x00 + (int64_t)W * C8;
    const int64_t xs[4] = {x00, x00 + C8, x10, x10 + C8};
    const float* gs[4] = {g00, g01, g10, g11};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      bf16x8 o;
      if (acc) {
        float prev[8];
        load8(dx, xs[e], prev);
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = (__bf16)(prev[k] + gs[e][k]);
      } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = (__bf16)gs[e][k];
      }
      dx[xs[e]] = o;
    }
  }
}

// with the BatchNorm-backward sums of the pooled layer (the stem) from the ROUNDED gradient it writes and the mask of
// bf16(y * scale + shift), like bn_bwd_reduce_bf16_kernel (fp32 twin: maxpool_bwd_quad_bn_kernel)
__global__ __launch_bounds__(256) void maxpool_bwd_bf16_quad_bn_kernel(const bf16x8* __restrict__ dout,
                                                                       const uint2* __restrict__ amax, bf16x8* __restrict__ dx,
                                                                       const bf16x8* __restrict__ y,
                                                                       const float* __restrict__ mean,
                                                                       const float* __restrict__ invstd,
                                                                       const float* __restrict__ act_scale,
                                                                       const float* __restrict__ act_shift,
                                                                       float* __restrict__ red, int acc, int B, int H, int W,
                                                                       int C8, int Ho, int Wo, int P) {
  __shared__ float sh[16][256];
  const int64_t total = (int64_t)B * Ho * Wo * C8;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;   // multiple of C8 (host check)
  const int t = threadIdx.x;
  const int64_t i0 = (int64_t)blockIdx.x * blockDim.x + t;
  const int c8 = (int)(i0 % C8);
  float mu[8], is[8], asc[8], ash[8], sg[8], sx[8];
  ldc8(mean, c8 * 8, mu);
  ldc8(invstd, c8 * 8, is);
  ldc8(act_scale, c8 * 8, asc);
  ldc8(act_shift, c8 * 8, ash);
#pragma unroll
  for (int k = 0; k < 8; ++k) sg[k] = sx[k] = 0.f;
  const uint2 none = make_uint2(0xffffffffu, 0xffffffffu);
  for (int64_t i = i0; i < total; i += stride) {
    int64_t rr = i / C8;
    const int p = (int)(rr % Wo);
    rr /= Wo;
    const int q = (int)(rr % Ho);
    const int b = (int)(rr / Ho);
    const bool q1 = q + 1 < Ho, p1 = p + 1 < Wo;
    const int64_t o00 = i, o01 = i + C8, o10 = i + (int64_t)Wo * C8, o11 = o10 + C8;
    float d00[8], d01[8], d10[8], d11[8];
    load8(dout, o00, d00);
    load8(dout, p1 ? o01 : o00, d01);
    load8(dout, q1 ? o10 : o00, d10);
    load8(dout, (q1 && p1) ? o11 : o00, d11);
    const uint2 a00 = amax[o00], a01 = p1 ? amax[o01] : none, a10 = q1 ? amax[o10] : none,
                a11 = (q1 && p1) ? amax[o11] : none;
    float g[4][8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int shf = 8 * (k & 3);
      const unsigned t00 = ((k < 4 ? a00.x : a00.y) >> shf) & 0xffu, t01 = ((k < 4 ? a01.x : a01.y) >> shf) & 0xffu,
                     t10 = ((k < 4 ? a10.x : a10.y) >> shf) & 0xffu, t11 = ((k < 4 ? a11.x : a11.y) >> shf) & 0xffu;
      float u00 = 0.f, u01 = 0.f, u10 = 0.f, u11 = 0.f;
      if (t00 == 4u) u00 += d00[k];
      if (t00 == 5u) u01 += d00[k];
      if (t01 == 3u) u01 += d01[k];
      if (t00 == 7u) u10 += d00[k];
      if (t10 == 1u) u10 += d10[k];
      if (t00 == 8u) u11 += d00[k];
      if (t01 == 6u) u11 += d01[k];
      if (t10 == 2u) u11 += d10[k];
      if (t11 == 0u) u11 += d11[k];
      g[0][k] = u00; g[1][k] = u01; g[2][k] = u10; g[3][k] = u11;
    }
    const int64_t x00 = (((int64_t)b * H + 2 * q) * W + 2 * p) * C8 + c8, x10 = x00 + (int64_t)W * C8;
    const int64_t xs[4] = {x00, x00 + C8, x10, x10 + C8};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float prev[8], yv[8];
      if (acc) load8(dx, xs[e], prev);
      load8(y, xs[e], yv);
      bf16x8 o;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        o[k] = (__bf16)(acc ? prev[k] + g[e][k] : g[e][k]);
        const float act = (float)(__bf16)(yv[k] * asc[k] + ash[k]);
        const float gm = act > 0.f ? (float)o[k] : 0.f;
        sg[k] += gm;
        sx[k] += gm * ((yv[k] - mu[k]) * is[k]);
      }
      dx[xs[e]] = o;
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    sh[k][t] = sg[k];
    sh[8 + k][t] = sx[k];
  }
  __syncthreads();
  const int rl = t / C8, RL = 256 / C8;
  for (int s2 = RL >> 1; s2 >= 1; s2 >>= 1) {
    if (rl < s2) {
#pragma unroll
      for (int k = 0; k < 16; ++k) sh[k][t] += sh[k][t + s2 * C8];
    }
    __syncthreads();
  }
  if (t < C8) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      red[(size_t)blockIdx.x * C8 * 8 + t * 8 + k] = sh[k][t];
      red[((size_t)P + blockIdx.x) * C8 * 8 + t * 8 + k] = sh[8 + k][t];
    }
  }
}

extern "C" int dt_maxpool3x3s2_bwd_bn_bf16_rows(int B, int H, int W, int C) {
  if (((H | W) & 1) != 0 || C <= 0 || (C & 7) != 0 || C / 8 > 256 || 256 % (C / 8) != 0) return 0;   // even maps only
  const int64_t g = ((int64_t)B * (H / 2) * (W / 2) * (C / 8) + 255) / 256;
  return (int)(g > 4096 ? 4096 : (g > 0 ? g : 1));
}

extern "C" int dt_maxpool3x3s2_bwd_bn_bf16(const void* dout, const uint8_t* argmax, void* dx, int accumulate,
                                           const dt_bn_bwd_fuse* fuse, float* red, int B, int H, int W, int C, void* stream) {
  DT_REQUIRE(dout && argmax && dx && fuse && red && fuse->y && fuse->mean && fuse->invstd && fuse->act_scale &&
                 fuse->act_shift && B > 0 && H > 0 && W > 0, "maxpool_bwd_bn_bf16: bad args");
  const int P = dt_maxpool3x3s2_bwd_bn_bf16_rows(B, H, W, C);
  DT_REQUIRE(P > 0, "maxpool_bwd_bn_bf16: even maps, C/8 a divisor of 256 (H=%d W=%d C=%d)", H, W, C);
  DT_REQUIRE((((uintptr_t)fuse->mean | (uintptr_t)fuse->invstd | (uintptr_t)fuse->act_scale | (uintptr_t)fuse->act_shift) & 15) == 0,
             "maxpool_bwd_bn_bf16: per-channel arrays must be 16-byte aligned");
  hipLaunchKernelGGL(maxpool_bwd_bf16_quad_bn_kernel, dim3(P), dim3(256), 0, (hipStream_t)stream, (const bf16x8*)dout,
                     (const uint2*)argmax, (bf16x8*)dx, (const bf16x8*)fuse->y, fuse->mean, fuse->invstd, fuse->act_scale,
                     fuse->act_shift, red, accumulate, B, H, W, C / 8, H / 2, W / 2, P);
  DT_LAUNCH_CHECK();
  return DT_OK;
}

extern "C" int dt_maxpool3x3s2_bwd_bf16(const void* dout, const uint8_t* argmax, void* dx, int accumulate, int B, int H,
                                        int W, int C, void* stream) {
  DT_REQUIRE(dout && argmax && dx && B > 0 && H > 0 && W > 0 && C > 0 && (C & 7) == 0, "maxpool_bwd_bf16: bad args");
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  if (((H | W) & 1) == 0) {
    int64_t gq = ((int64_t)B * Ho * Wo * (C / 8) + 255) / 256;
    if (gq > 8192) gq = 8192;
    hipLaunchKernelGGL(maxpool_bwd_bf16_quad_kernel, dim3((unsigned)gq), dim3(256), 0, (hipStream_t)stream,
                       (const bf16x8*)dout, (const uint2*)argmax, (bf16x8*)dx, accumulate, B, H, W, C / 8, Ho, Wo);
    DT_LAUNCH_CHECK();
    return DT_OK;
  }
  const int64_t total = (int64_t)B * H * W * (C / 8);
  int64_t g = (total + 255) / 256;
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(maxpool_bwd_bf16_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, (const bf16x8*)dout,
                     (const uint2*)argmax, (bf16x8*)dx, accumulate, B, H, W, C / 8, Ho, Wo);
  DT_LAUNCH_CHECK();
  return DT_OK;
}

// ------------------------------------------------------------------ per-channel sums of a bf16 gradient
// bias gradient of a convolution without BatchNorm (the 1x1 identity_conv of the ResUnet decoder under AMP): bf16 twin of
// dt_channel_sums (elementwise.hip) — row blocks -> fp32 partial rows -> fixed-order fp64 final, no atomics
#define CSB_RB 256
__global__ __launch_bounds__(256) void channel_sums_bf16_kernel(const bf16x8* __restrict__ g, float* __restrict__ part,
                                                                int64_t n_pix, int C8) {
  const int q = threadIdx.x % C8, rl = threadIdx.x / C8, RL = 256 / C8;
  const int64_t p0 = (int64_t)blockIdx.x * CSB_RB;
  int64_t p1 = p0 + CSB_RB;
  if (p1 > n_pix) p1 = n_pix;
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int64_t p = p0 + rl; p < p1; p += RL) {
    float v[8];
    load8(g, p * C8 + q, v);
#pragma unroll
    for (int k = 0; k < 8; ++k) s[k] += v[k];
  }
  __shared__ float sh[8][256];
#pragma unroll
  for (int k = 0; k < 8; ++k) sh[k][threadIdx.x] = s[k];
  __syncthreads();
  if (threadIdx.x < 8 * C8) {
    const int c = threadIdx.x, qq = c >> 3, k = c & 7;
    float t = 0.f;
    for (int r = 0; r < RL; ++r) t += sh[k][r * C8 + qq];   // fixed order
    part[(size_t)blockIdx.x * (8 * C8) + c] = t;
  }
}

__global__ void channel_sums_bf16_final_kernel(const float* __restrict__ part, int P, int C, float* __restrict__ out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C) {
    double s = 0.0;
    for (int p = 0; p < P; ++p) s += (double)part[(size_t)p * C + c];
    out[c] = (float)s;
  }
}

extern "C" int64_t dt_channel_sums_bf16_workspace(int64_t n_pix, int C) { return (int64_t)dt_cdiv(n_pix, CSB_RB) * C; }

extern "C" int dt_channel_sums_bf16(const void* g, float* workspace, int64_t n_pix, int C, float* out, void* stream) {
  DT_REQUIRE(g && workspace && out && n_pix > 0 && C > 0, "channel_sums_bf16: bad args");
  DT_REQUIRE((C & 7) == 0 && C <= 256 && 256 % (C / 8) == 0, "channel_sums_bf16: C/8 must divide 256, C <= 256 (C=%d)", C);
  hipStream_t st = (hipStream_t)stream;
  const int P = dt_cdiv(n_pix, CSB_RB);
  hipLaunchKernelGGL(channel_sums_bf16_kernel, dim3(P), dim3(256), 0, st, (const bf16x8*)g, workspace, n_pix, C / 8);
  DT_LAUNCH_CHECK();
  hipLaunchKernelGGL(channel_sums_bf16_final_kernel, dim3(dt_cdiv(C, 64)), dim3(64), 0, st, workspace, P, C, out);
  DT_LAUNCH_CHECK();
  return DT_OK;
}

// ------------------------------------------------------------------ nearest x2 upsample backward
__global__ __launch_bounds__(256) void upsample2x_bwd_bf16_kernel(const bf16x8* __restrict__ dup, bf16x8* __restrict__ dx,
                                                                  int acc, int B, int H, int W, int C8) {
  const int64_t total = (int64_t)B * H * W * C8;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int W2 = 2 * W;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int c8 = (int)(i % C8);
    int64_t rr = i / C8;
    const int x = (int)(rr % W);
    rr /= W;
    const int y = (int)(rr % H);
    const int b = (int)(rr / H);
    const int64_t base = (((int64_t)b * 2 * H + 2 * y) * W2 + 2 * x) * C8 + c8;
    float a[8], bq[8], c[8], d[8];
    load8(dup, base, a);
    load8(dup, base + C8, bq);
    load8(dup, base + (int64_t)W2 * C8, c);
    load8(dup, base + (int64_t)W2 * C8 + C8, d);
    bf16x8 o;
    if (acc) {   // a node of a dense decoder collects the gradients of several consumers: one rounding per contribution
      float prev[8];
      load8(dx, i, prev);
#pragma unroll
      for (int k = 0; k < 8; ++k) o[k] = (__bf16)(prev[k] + ((a[k] + bq[k]) + (c[k] + d[k])));
    } else {
#pragma unroll
      for (int k = 0; k < 8; ++k) o[k] = (__bf16)((a[k] + bq[k]) + (c[k] + d[k]));
    }
    dx[i] = o;
  }
}

static int upsample2x_bwd_bf16_launch(const void* dup, void* dx, int acc, int B, int H, int W, int C, void* stream) {
  DT_REQUIRE(dup && dx && B > 0 && H > 0 && W > 0 && C > 0 && (C & 7) == 0, "upsample2x_bwd_bf16: bad args");
  const int64_t total = (int64_t)B * H * W * (C / 8);
  int64_t g = (total + 255) / 256;
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(upsample2x_bwd_bf16_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream,
                     (const bf16x8*)dup, (bf16x8*)dx, acc, B, H, W, C / 8);
  DT_LAUNCH_CHECK();
  return DT_OK;
}

extern "C" int dt_upsample2x_bwd_bf16(const void* dup, void* dx, int B, int H, int W, int C, void* stream) {
  return upsample2x_bwd_bf16_launch(dup, dx, 0, B, H, W, C, stream);
}

// dx (+)= 2x2 sums: accumulate != 0 adds to the bf16 gradient already in dx (Unet++ nodes with several consumers)
extern "C" int dt_upsample2x_bwd_acc_bf16(const void* dup, void* dx, int accumulate, int B, int H, int W, int C, void* stream) {
  return upsample2x_bwd_bf16_launch(dup, dx, accumulate, B, H, W, C, stream);
}

// ------------------------------------------------------------------ channel-slice copies of bf16 tensors (Unet++ under AMP)
// bf16 twin of dt_channel_slice: wide[n, off : off + Cn] = narrow[n, :]  /  narrow[n, :] (+)= wide[n, off : off + Cn];
// channel counts and the offset multiples of 8 (16-byte units); the accumulating form adds in fp32 and rounds once
__global__ __launch_bounds__(256) void channel_slice_bf16_kernel(const bf16x8* __restrict__ src, bf16x8* __restrict__ dst,
                                                                 int64_t n8, int Cn8, int Cw8, int off8, int to_wide, int acc) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += stride) {
    const int64_t pix = i / Cn8;
    const int c = (int)(i - pix * Cn8);
    const int64_t w = pix * Cw8 + off8 + c;
    if (to_wide) {
      dst[w] = src[i];
    } else if (!acc) {
      dst[i] = src[w];
    } else {
      float a[8], b[8];
      load8(src, w, a);
      load8(dst, i, b);
      bf16x8 o;
#pragma unroll
      for (int k = 0; k < 8; ++k) o[k] = (__bf16)(b[k] + a[k]);
      dst[i] = o;
    }
  }
}

extern "C" int dt_channel_slice_bf16(const void* src, void* dst, int64_t n_pix, int C_narrow, int C_wide, int offset,
                                     int to_wide, int accumulate, void* stream) {
  DT_REQUIRE(src && dst && n_pix > 0 && C_narrow > 0 && C_wide >= C_narrow, "channel_slice_bf16: bad args");
  DT_REQUIRE((C_narrow & 7) == 0 && (C_wide & 7) == 0 && (offset & 7) == 0 && offset >= 0 && offset + C_narrow <= C_wide,
             "channel_slice_bf16: channel counts and offset must be multiples of 8 inside the wide tensor");
  const int64_t n8 = n_pix * (C_narrow / 8);
  const int grid = (int)(n8 / 256 + 1 < 8192 ? n8 / 256 + 1 : 8192);
  hipLaunchKernelGGL(channel_slice_bf16_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16x8*)src,
                     (bf16x8*)dst, n8, C_narrow / 8, C_wide / 8, offset / 8, to_wide, accumulate);
  DT_LAUNCH_CHECK();
  return DT_OK;
}

// 2x2 sum + BatchNorm-backward reduction of the producing layer (bf16 twin of upsample2x_bwd_bn_kernel): sums from the
// ROUNDED gradient and the mask of bf16(y*scale+shift), like bn_bwd_reduce_bf16_kernel
__global__ __launch_bounds__(256) void upsample2x_bwd_bn_bf16_kernel(const bf16x8* __restrict__ dup, bf16x8* __restrict__ dx,
                                                                     const bf16x8* __restrict__ y,
                                                                     const float* __restrict__ mean,
                                                                     const float* __restrict__ invstd,
                                                                     const float* __restrict__ act_scale,
                                                                     const float* __restrict__ act_shift,
                                                                     float* __restrict__ red, int B, int H, int W, int C8,
                                                                     int P) {
  __shared__ float sh[16][256];
  const int64_t total = (int64_t)B * H * W * C8;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;   // multiple of C8 (host check)
  const int W2 = 2 * W, t = threadIdx.x;
  const int64_t i0 = (int64_t)blockIdx.x * blockDim.x + t;
  const int c8 = (int)(i0 % C8);
  float mu[8], is[8], asc[8], ash[8], sg[8], sx[8];
  ldc8(mean, c8 * 8, mu);
  ldc8(invstd, c8 * 8, is);
  ldc8(act_scale, c8 * 8, asc);
  ldc8(act_shift, c8 * 8, ash);
#pragma unroll
  for (int k = 0; k < 8; ++k) sg[k] = sx[k] = 0.f;
  for (int64_t i = i0; i < total; i += stride) {
    int64_t rr = i / C8;
    const int x = (int)(rr % W);
    rr /= W;
    const int yy = (int)(rr % H);
    const int b = (int)(rr / H);
    const int64_t base = (((int64_t)b * 2 * H + 2 * yy) * W2 + 2 * x) * C8 + c8;
    float a[8], bq[8], c[8], d[8], yv[8];
    load8(dup, base, a);
    load8(dup, base + C8, bq);
    load8(dup, base + (int64_t)W2 * C8, c);
    load8(dup, base + (int64_t)W2 * C8 + C8, d);
    load8(y, i, yv);
    bf16x8 o;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      o[k] = (__bf16)((a[k] + bq[k]) + (c[k] + d[k]));
      const float act = (float)(__bf16)(yv[k] * asc[k] + ash[k]);
      const float g = act > 0.f ? (float)o[k] : 0.f;
      sg[k] += g;
      sx[k] += g * ((yv[k] - mu[k]) * is[k]);
    }
    dx[i] = o;
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    sh[k][t] = sg[k];
    sh[8 + k][t] = sx[k];
  }
  __syncthreads();
  const int rl = t / C8, RL = 256 / C8;
  for (int s = RL >> 1; s >= 1; s >>= 1) {
    if (rl < s) {
#pragma unroll
      for (int k = 0; k < 16; ++k) sh[k][t] += sh[k][t + s * C8];
    }
    __syncthreads();
  }
  if (t < C8) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      red[(size_t)blockIdx.x * C8 * 8 + t * 8 + k] = sh[k][t];
      red[((size_t)P + blockIdx.x) * C8 * 8 + t * 8 + k] = sh[8 + k][t];
    }
  }
}

extern "C" int dt_upsample2x_bwd_bn_bf16_rows(int B, int H, int W, int C) {
  int64_t g = ((int64_t)B * H * W * (C / 8) + 255) / 256;
  return (int)(g > 4096 ? 4096 : (g > 0 ? g : 1));
}

extern "C" int dt_upsample2x_bwd_bn_bf16(const void* dup, void* dx, const dt_bn_bwd_fuse* fuse, float* red, int B, int H,
                                         int W, int C, void* stream) {
  DT_REQUIRE(dup && dx && fuse && red && fuse->y && fuse->mean && fuse->invstd && fuse->act_scale && fuse->act_shift &&
                 B > 0 && H > 0 && W > 0 && C > 0 && (C & 7) == 0,
             "upsample2x_bwd_bn_bf16: bad args");
  const int C8 = C / 8;
  DT_REQUIRE(C8 <= 256 && 256 % C8 == 0, "upsample2x_bwd_bn_bf16: C/8 must divide 256 (C=%d)", C);
  DT_REQUIRE((((uintptr_t)fuse->mean | (uintptr_t)fuse->invstd | (uintptr_t)fuse->act_scale |
               (uintptr_t)fuse->act_shift) & 15) == 0, "upsample2x_bwd_bn_bf16: per-channel arrays must be 16-byte aligned");
  const int P = dt_upsample2x_bwd_bn_bf16_rows(B, H, W, C);
  hipLaunchKernelGGL(upsample2x_bwd_bn_bf16_kernel, dim3(P), dim3(256), 0, (hipStream_t)stream, (const bf16x8*)dup,
                     (bf16x8*)dx, (const bf16x8*)fuse->y, fuse->mean, fuse->invstd, fuse->act_scale, fuse->act_shift, red,
                     B, H, W, C8, P);
  DT_LAUNCH_CHECK();
  return DT_OK;
}

// ------------------------------------------------------------------ fp32 -> bf16
__global__ __launch_bounds__(256) void f32_to_bf16_kernel(const f32x4* __restrict__ x, bf16x8* __restrict__ out,
                                                          int64_t n8) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += stride) {
    const f32x4 a = x[2 * i], b = x[2 * i + 1];
    bf16x8 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      o[k] = (__bf16)a[k];
      o[4 + k] = (__bf16)b[k];
    }
    out[i] = o;
  }
}

extern "C" int dt_f32_to_bf16(const float* x, void* out, int64_t n, void* stream) {
  DT_REQUIRE(x && out && n > 0 && (n & 7) == 0, "f32_to_bf16: n must be a multiple of 8");
  int64_t g = (n / 8 + 255) / 256;
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(f32_to_bf16_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, (const f32x4*)x,
                     (bf16x8*)out, n / 8);
  DT_LAUNCH_CHECK();
  return DT_OK;
}
